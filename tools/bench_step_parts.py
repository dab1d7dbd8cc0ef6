import sys, time, os
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import squid_amd
if len(sys.argv) > 2:
    import torch; torch.cuda.set_device(0); torch.cuda.synchronize(); print("torch loaded")
pre = sys.argv[1]
bam, chim = pre + ".bam", pre + ".chim.bam"
with squid_amd.Context() as ctx:
    if len(sys.argv) > 3:  # as bench.py: three steps from the file first
        for _ in range(3):
            ctx.clear_records(); ctx.load(bam, chim, threads=256); ctx.build_graph(); ctx.order(); ctx.sv_text()
        print("from-file steps done")
    ctx.stage_bam(bam)
    def step(prt=False):
        t0 = time.perf_counter(); ctx.clear_records()
        t1 = time.perf_counter(); ctx.load(bam, chim, threads=256)
        t2 = time.perf_counter(); ctx.build_graph()
        t3 = time.perf_counter(); ctx.order()
        t4 = time.perf_counter(); text = ctx.sv_text()
        t5 = time.perf_counter()
        with open("/tmp/x_sv.txt", "w") as f: f.write(text)
        t6 = time.perf_counter()
        if prt: print(f"clear {1e3*(t1-t0):.1f} load {1e3*(t2-t1):.1f} build {1e3*(t3-t2):.1f} order {1e3*(t4-t3):.1f} sv {1e3*(t5-t4):.1f} write {1e3*(t6-t5):.1f} total {1e3*(t6-t0):.1f}")
    step(); step()
    ctx.timing_accumulate(True)
    t0 = time.perf_counter()
    for _ in range(6): step(True)
    print("avg", (time.perf_counter() - t0) / 6 * 1e3)
