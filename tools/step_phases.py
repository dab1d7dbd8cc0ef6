"""wall time of the phases of one BAM -> _sv.txt step (chimeric decode, concordant ingest, graph, order, calls). usage: step_phases.py <prefix>"""
import sys, time
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import squid_amd, ctypes as C
pre = sys.argv[1]
with squid_amd.Context() as ctx:
    ctx.stage_bam(f"{pre}.bam")
    for it in range(3):
        tc0 = time.time(); ctx.clear_records(); tc1 = time.time()
        names, lens = squid_amd.read_header(f"{pre}.bam"); ctx.ref_names = names
        arr = (C.c_int32 * len(lens))(*lens)
        t0 = time.time(); ctx.lib.sq_set_references(ctx.h, len(lens), arr); t1 = time.time()
        ctx.lib.sq_ingest_chimeric_file(ctx.h, f"{pre}.chim.bam".encode()); t2 = time.time()
        ctx.lib.sq_ingest_concordant_file(ctx.h, f"{pre}.bam".encode(), 16); t3 = time.time()
        ctx.build_graph(); t4 = time.time(); ctx.order(); t5 = time.time(); txt = ctx.sv_text(); t6 = time.time()
        print(f"clear {1e3*(tc1-tc0):.1f} refs {1e3*(t1-t0):.1f} chim {1e3*(t2-t1):.1f} conc {1e3*(t3-t2):.1f} graph {1e3*(t4-t3):.1f} order {1e3*(t5-t4):.1f} sv {1e3*(t6-t5):.1f} ms")
