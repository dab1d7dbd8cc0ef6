#!/usr/bin/env python3
"""--bwa mode at the size of C3: where does the time go, and is the `_sv.txt` the CPU oracle's?
usage: tools/bwa_probe.py [--star-sample] [--no-oracle] [--steps N]
  default sample: gen_synth_bam --config C3 --bwa (one coordinate-sorted file, split reads as supplementary records);
  --star-sample : the STAR-style C3 file of bench.py read in --bwa mode (no split reads: throughput of the two record loops only)
Prints one JSON line at the end (committed as profiles/r05_bwa_C3.json)."""
import hashlib, json, os, statistics, subprocess, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import squid_amd

star = "--star-sample" in sys.argv
steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 4
pre = "/tmp/squid_bench/C3_s20180003" if star else "/tmp/squid_bench/C3bwa_s20180003"
if not os.path.exists(pre + ".bam"):
    os.makedirs(os.path.dirname(pre), exist_ok=True)
    subprocess.check_call([str(ROOT / "build" / "gen_synth_bam"), "--config", "C3", "--seed", "20180003", "--out", pre, "--threads", "128"] + ([] if star else ["--bwa"]), stdout=subprocess.DEVNULL)
squid_amd.keep_host_memory()
rows = []
with squid_amd.Context(star_mapq=False, min_mapqual=1) as ctx:
    ctx.keep_stage_graphs(False)
    for it in range(steps + 1):
        ctx.clear_records()
        if "--drop" in sys.argv:
            squid_amd.drop_file_cache()  # (as bench.py's steps: nothing kept from earlier reads of the file)
        t0 = time.perf_counter()
        ctx.load_bwa(f"{pre}.bam", threads=16)
        t1 = time.perf_counter()
        ctx.build_graph()
        t2 = time.perf_counter()
        ctx.order_sizes()
        text = ctx.sv_text_fast()
        t3 = time.perf_counter()
        k = ctx.counts()
        sha = hashlib.sha256(text.encode()).hexdigest()
        print(f"step {it}: ingest {1e3*(t1-t0):.0f} ms, graph {1e3*(t2-t1):.0f} ms, order + SV {1e3*(t3-t2):.0f} ms; {k['n_concordant']} records, {len(text.splitlines()) - 1} SV rows, sha {sha[:12]}, "
              f"{k['n_concordant']/(t3-t0)/1e6:.1f} M aln/s", file=sys.stderr, flush=True)
        if it:  # (step 0: warm-up)
            rows.append((t3 - t0, t1 - t0, t2 - t1, t3 - t2, sha))
    stages = {a: round(b["ms"], 1) for a, b in sorted(ctx.timing().items(), key=lambda t: -t[1]["ms"])[:12]}
    launches = {a: b["launches"] for a, b in ctx.timing().items() if "stretches" in a}
    n_rec, gpu_reader = k["n_concordant"], k["chimeric_through_gpu_reader"]
line = {"metric": "alignments/sec, one --bwa BAM file (page cache) -> _sv.txt", "value": n_rec / statistics.median(r[0] for r in rows), "unit": "alignments/s", "n_gpus": 1, "steps": steps,
        "ms_per_step": 1e3 * statistics.median(r[0] for r in rows), "ms_each": [round(1e3 * r[0], 1) for r in rows], "ingest_ms": round(1e3 * statistics.median(r[1] for r in rows), 1),
        "graph_ms": round(1e3 * statistics.median(r[2] for r in rows), 1), "order_sv_ms": round(1e3 * statistics.median(r[3] for r in rows), 1), "records": n_rec,
        "sample": "gen_synth_bam --config C3 (STAR-style file read in --bwa mode)" if star else "gen_synth_bam --config C3 --bwa", "ingest_through_gpu_reader": bool(gpu_reader),
        "steps_identical": len({r[4] for r in rows}) == 1, "sv_sha256": rows[0][4], "sv_rows": len(text.splitlines()) - 1, "last_step_stage_ms": stages, "stretches": launches,
        "what": "sq_ingest_bwa_file (GPU reader: inflate, record parse, QNAMEs kept on the device, one copy back) + BuildNode_BWA / RawEdges on the host threads in stretches + the shared graph kernels, ordering, SV calls"}
if "--no-oracle" not in sys.argv:
    out = "/tmp/squid_bench/bwa_oracle"
    t0 = time.perf_counter()
    subprocess.check_call(["taskset", "-c", "0", str(ROOT / "build" / "squid_oracle"), "--bwa", "-b", f"{pre}.bam", "-o", out], stdout=subprocess.DEVNULL)
    t_cpu = time.perf_counter() - t0
    want = Path(out + "_sv.txt").read_text()
    line["cpu_baseline"] = {"value": n_rec / t_cpu, "unit": "alignments/s", "cores": 1, "kind": "port", "sample": f"the same file, {t_cpu:.1f} s, taskset -c 0", "sv_identical_to_gpu": want == text, "sv_rows": len(want.splitlines()) - 1}
    print(f"CPU oracle --bwa: {t_cpu:.1f} s, _sv.txt identical: {want == text}", file=sys.stderr)
print(json.dumps(line))
