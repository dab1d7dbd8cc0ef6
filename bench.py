#!/usr/bin/env python3
"""bench.py -- BAM -> _sv.txt throughput of the MI355X-native SQUID hot path.

One "step" = one full pass of the hot path (record filters -> segmentation -> edges -> filters -> compression ->
components -> ordering -> breakpoints -> support -> _sv.txt text) over one synthetic sample whose decoded
alignment records are already resident in HBM (ingest = host BGZF/BAM decode + H2D copy happens before the timed
region; the file-inclusive rate is reported separately as `e2e_value`).

N = 1: workload = BASELINE.json configs[1] ("hg38 chr17 only, 1M synthetic paired-end reads, ~20 planted
fusions"), generator config C2.  N > 1, default `--shard sample`: one independent C2 sample per rank (a single
chromosome cannot be sharded by chromosome; sample-parallel, no data-path collective, weak scaling).
`--shard chromosome --workload C3 [--records R]`: ONE full-hg38 sample, rank r holds the records of a contiguous
chromosome range and the library's exchanges travel as RCCL all-gathers (BASELINE.json configs[3]; strong scaling).
Launched by torch.distributed.run, barrier + max-over-ranks timing (DESIGN.md "Multi-GPU").

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
BUILD = ROOT / "build"

GPU_KERNELS_PREFIX = ("k_", "scan_")
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling


def synth(config: str, seed: int, outdir: Path, records: int | None = None) -> Path:
    pre = outdir / (f"{config}_s{seed}" + (f"_r{records}" if records else ""))
    if not Path(f"{pre}.bam").exists():
        tmp = Path(f"{pre}.tmp{os.getpid()}")
        subprocess.check_call([str(BUILD / "gen_synth_bam"), "--config", config, "--seed", str(seed), "--out", str(tmp), "--threads", "8"]
                              + (["--records", str(records), "--level", "1"] if records else []), stdout=subprocess.DEVNULL)
        for ext in (".chim.bam", ".truth.txt", ".bam"):
            os.replace(f"{tmp}{ext}", f"{pre}{ext}")
    return pre


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C2", help="generator config (C2 = BASELINE.json configs[1])")
    ap.add_argument("--records", type=int, default=None, help="override the record count of the workload (generator --records)")
    ap.add_argument("--shard", choices=["sample", "chromosome"], default="sample", help="what the ranks of a multi-GPU run divide (see the module docstring)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workdir", default=None)
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch

    dist = None
    if world > 1:
        import torch.distributed as dist  # noqa: F811

        # SQUID_DIST_BACKEND=gloo lets several ranks share one GPU (functional checks on a one-GPU box)
        backend = os.environ.get("SQUID_DIST_BACKEND", "nccl")
        local_rank %= max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product has no CPU path")
    sharded = a.shard == "chromosome" and world > 1

    import squid_amd

    if not (BUILD / "libsquid_hip.so").exists() or not (BUILD / "gen_synth_bam").exists():
        if rank == 0:
            squid_amd.build()
        if dist:
            dist.barrier()

    work = Path(a.workdir) if a.workdir else Path(tempfile.gettempdir()) / "squid_bench"
    work.mkdir(parents=True, exist_ok=True)
    srank = 0 if sharded else rank  # a sharded run works on ONE sample
    seed = 20180002 + 1000 * srank  # rank 0 of C2 = the generator's default seed for that config
    if a.workload != "C2":
        seed = 20180000 + int(a.workload[1:]) + 1000 * srank if a.workload[1:].isdigit() else 20180007 + 1000 * srank
    if sharded:
        if rank == 0:
            synth(a.workload, seed, work, a.records)
        dist.barrier()
    pre = synth(a.workload, seed, work, a.records)

    def barrier():
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
            torch.cuda.synchronize()

    exchange, plan = None, None
    if sharded:
        from squid_amd.dist import TorchExchange, plan_shards

        _, ref_len = squid_amd.read_header(f"{pre}.bam")
        plan = plan_shards(ref_len, world)  # balanced by reference length (records per chromosome are not known before the decode)
        exchange = TorchExchange(dist, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        ctx = squid_amd.Context(device=local_rank, rank=rank, world_size=world, exchange=exchange)
    else:
        ctx = squid_amd.Context(device=local_rank)
    t_ing0 = time.perf_counter()
    ctx.load(f"{pre}.bam", f"{pre}.chim.bam", threads=max(1, (os.cpu_count() or 8) // max(1, world)), shard=plan[rank] if sharded else None)
    t_ingest = time.perf_counter() - t_ing0
    n_aln = ctx.counts()["n_concordant"] + (ctx.counts()["n_chimeric_records"] if (not sharded or rank == 0) else 0)
    sv_path = work / f"bench_rank{rank}_sv.txt"

    def step() -> str:
        ctx.reset()
        ctx.build_graph()
        ctx.order()
        text = ctx.sv_text()
        sv_path.write_text(text)
        return text

    for _ in range(a.warmup):
        step()
    barrier()
    ctx.timing_accumulate(True)  # the library sums its HIP-event / host timers over the timed steps; read once afterwards
    t0 = time.perf_counter()
    for _ in range(a.steps):
        text = step()
    barrier()
    elapsed = time.perf_counter() - t0
    agg: dict[str, dict] = {k: {"ms": v["ms"], "launches": v["launches"], "bytes": v["bytes"]} for k, v in ctx.timing().items()}
    from squid_amd.dist import reduce_timing

    elapsed, total_aln = reduce_timing(elapsed, float(n_aln), dist, device="cuda")

    if rank != 0:
        ctx.close()
        if dist:
            dist.destroy_process_group()
        return

    value = total_aln * a.steps / elapsed
    # dominant GPU kernel by accumulated HIP-event time on the library stream
    gk = {k: v for k, v in agg.items() if k.startswith(GPU_KERNELS_PREFIX) and v["bytes"] > 0}
    dom = max(gk, key=lambda k: gk[k]["ms"])
    d = gk[dom]
    per_launch_bytes = d["bytes"] / d["launches"]
    per_launch_ms = d["ms"] / d["launches"]
    achieved = per_launch_bytes / (per_launch_ms * 1e-3) / 1e9
    gpu_ms = sum(v["ms"] for k, v in agg.items() if k.startswith(GPU_KERNELS_PREFIX)) / a.steps
    scan_bytes = sum(v["bytes"] for v in gk.values()) / a.steps
    traffic = None
    tfile = ROOT / "profiles" / "pmc_traffic.json"  # written by tools/profile.sh from the rocprofv3 --pmc passes
    if tfile.exists() and a.workload == "C2" and not a.records:  # the counters were collected on the default workload
        try:
            traffic = (json.loads(tfile.read_text()).get(dom) or {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        "metric": "paired-end alignments/sec BAM->_sv.txt (records resident in HBM; bit-exact SV calls vs CPU oracle)",
        "value": value, "unit": "alignments/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None,
        "dtype": "int32", "data": "synthetic",
        "config": {"workload": f"{a.workload}: " + ("hg38 chr17 only, 1M synthetic paired-end records, 20 planted fusions (BASELINE.json configs[1])" if a.workload == "C2" else "generator config " + a.workload),
                   "records_per_gpu": int(n_aln) if not sharded else int(total_aln / world),
                   "parallelism": ("one sample sharded by chromosome, %d all-gathers per step" % (exchange.calls // max(1, a.steps + a.warmup)) if sharded else "1 sample per GPU, no collective") if world > 1 else "single GPU",
                   "ingest": "excluded from value: BGZF/BAM decode + H2D, see e2e_value"},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "bytes_per_launch": per_launch_bytes, "us_per_launch": per_launch_ms * 1e3,
                     "all_scan_kernels": {"ms_per_step": gpu_ms, "algorithmic_bytes_per_step": scan_bytes,
                                          "achieved_GBs": scan_bytes / (gpu_ms * 1e-3) / 1e9 if gpu_ms > 0 else None}},
        "e2e_value": (total_aln if sharded else total_aln / world) / (t_ingest + elapsed / a.steps), "e2e_note": "one sample from the BAM files, first pass included: BGZF/BAM decode (host threads; the GPU reader for files >= 1 GiB) + H2D (rank 0)",
        "stage_ms_per_step": {k: round(v["ms"] / a.steps, 4) for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:(None if os.environ.get("BENCH_ALL_STAGES") else 12)]},
    }
    ctx.close()
    if not a.no_cpu_baseline and world == 1:
        # CPU oracle (a port of the reference, 1 thread) on the same files, timed on this box
        t0 = time.perf_counter()
        subprocess.check_call([str(BUILD / "squid_oracle"), "-b", f"{pre}.bam", "-c", f"{pre}.chim.bam", "-o", str(work / "cpu_baseline")], stdout=subprocess.DEVNULL)
        tc = time.perf_counter() - t0
        same = (work / "cpu_baseline_sv.txt").read_text() == text
        out["cpu_baseline"] = {"value": n_aln / tc, "unit": "alignments/s", "cores": 1, "kind": "port",
                               "sample": f"the full {a.workload} workload once, BAM file -> _sv.txt incl. its three BAM decodes ({tc:.2f} s)", "sv_identical_to_gpu": same}
    if dist:
        dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
