#!/usr/bin/env python3
"""bench.py -- BAM -> _sv.txt throughput of the MI355X-native SQUID hot path, end to end.

One "step" = ONE whole run of the hot path on one synthetic sample: chimeric BAM decode (host), concordant BAM
decode on the GPU (BGZF inflate -> record boundaries -> record parse = the reference's three BamReader passes,
SegmentGraph.cpp:293-296,1570-1577,3126-3129, done once), record filters -> segmentation -> edges -> filters ->
compression -> components -> ordering -> breakpoints -> support -> `_sv.txt` written and closed.  Nothing is kept
from step to step except device buffers (sq_clear_records).

Workload: BASELINE.json configs[2] -- "Full hg38, 50M-read synthetic STAR concordant+chimeric BAM, 1xMI355X" --
generator config C3 (50.8 M concordant records, zlib level 6), the largest single-GPU configuration.
`value` = alignments/s with the compressed BAM bytes resident in HBM when the timed region starts (sq_stage_bam);
`from_file_value` = the same step reading the BAM from the page cache (host->device copy of the file included), measured
after the timed region in a fresh context.
`resident_pass_value` = the graph pass alone over records already decoded in HBM (what round 1 reported).

N > 1 (`--gpus N`; the script launches its own ranks through torch.distributed.run when WORLD_SIZE is not set):
ONE C3 sample sharded by chromosome (BASELINE.json configs[3] layout), rank r decodes and holds the records of a
contiguous RefID range, the library's exchanges travel as RCCL all-gathers; "scaling": "strong".
`--shard sample` instead runs one independent sample per rank (no collective, weak scaling).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
BUILD = ROOT / "build"

GPU_KERNELS_PREFIX = ("k_", "scan_")
INGEST_KERNELS = ("k_inflate_tokens", "k_lz_resolve", "k_rec_boundaries", "k_parse_count", "k_parse_write", "k_inflate")
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling
WORKLOADS = {
    "C2": "hg38 chr17 only, 1M synthetic paired-end records, 20 planted fusions (BASELINE.json configs[1])",
    "C3": "full hg38, 50M-record synthetic STAR concordant+chimeric BAM, 200 planted TSVs, zlib level 6 (BASELINE.json configs[2])",
    "C5": "dense-graph stress (-w 1 -a 50) (BASELINE.json configs[4])",
}


def synth(config: str, seed: int, outdir: Path, records: int | None = None, level: int | None = None) -> Path:
    pre = outdir / (f"{config}_s{seed}" + (f"_r{records}" if records else "") + (f"_l{level}" if level is not None else ""))
    if not Path(f"{pre}.bam").exists():
        tmp = Path(f"{pre}.tmp{os.getpid()}")
        cmd = [str(BUILD / "gen_synth_bam"), "--config", config, "--seed", str(seed), "--out", str(tmp), "--threads", str(max(1, os.cpu_count() or 8))]
        if records:
            cmd += ["--records", str(records)]
        if level is not None:
            cmd += ["--level", str(level)]
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
        for ext in (".chim.bam", ".truth.txt", ".bam.bai", ".bam"):
            os.replace(f"{tmp}{ext}", f"{pre}{ext}")
    return pre


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def self_launch(a) -> None:
    """`bench.py --gpus N` without a launcher: start N ranks as a child process (never exec: nothing here has touched the GPU)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           str(Path(__file__).resolve())] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    # rank 0 prints the result line; anything else on stdout (transport banners of the ranks) goes to stderr
    lines = r.stdout.splitlines()
    result = [l for l in lines if l.startswith('{"metric"')]
    for l in lines:
        if not l.startswith('{"metric"'):
            print(l, file=sys.stderr)
    if result:
        print(result[-1])
    raise SystemExit(r.returncode if r.returncode else (0 if result else 1))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="C3", help="generator config (C3 = BASELINE.json configs[2], the largest single-GPU configuration)")
    ap.add_argument("--records", type=int, default=None, help="override the record count of the workload (generator --records)")
    ap.add_argument("--shard", choices=["sample", "chromosome"], default="chromosome", help="what the ranks of a multi-GPU run divide (see the module docstring)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-records", type=int, default=5_000_000, help="records of the bounded sample the CPU oracle is timed on")
    ap.add_argument("--resident-steps", type=int, default=5, help="extra (untimed for `value`) graph passes over resident records, for the per-kernel roofline figures")
    ap.add_argument("--workdir", default=None)
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)

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
    cfgnum = int(a.workload[1:]) if a.workload[1:].isdigit() else 7
    seed = 20180000 + cfgnum + 1000 * srank  # rank 0 = the generator's default seed for that config
    t_gen0 = time.perf_counter()
    if sharded:
        if rank == 0:
            synth(a.workload, seed, work, a.records)
        dist.barrier()
    pre = synth(a.workload, seed, work, a.records)
    t_gen = time.perf_counter() - t_gen0
    bam, chim = f"{pre}.bam", f"{pre}.chim.bam"

    def barrier():
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
            torch.cuda.synchronize()

    exchange, plan = None, None
    host_threads = max(1, (os.cpu_count() or 8) // max(1, world))
    if sharded:
        from squid_amd.dist import TorchExchange, plan_shards, shard_weights

        _, ref_len = squid_amd.read_header(bam)
        plan = plan_shards(shard_weights(bam, ref_len), world)  # balanced by compressed bytes per chromosome (from the .bai), else by reference length
        exchange = TorchExchange(dist, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        ctx = squid_amd.Context(device=local_rank, rank=rank, world_size=world, exchange=exchange)
    else:
        ctx = squid_amd.Context(device=local_rank)
    sv_path = work / f"bench_rank{rank}_sv.txt"

    def graph_pass() -> str:
        ctx.build_graph()
        ctx.order()
        text = ctx.sv_text()
        if rank == 0 or not sharded:
            with open(sv_path, "w") as f:
                f.write(text)
        return text

    def step() -> str:  # BAM files -> _sv.txt
        ctx.clear_records()
        ctx.load(bam, chim, threads=host_threads, shard=plan[rank] if sharded else None)
        return graph_pass()

    # ---- the timed region: compressed BAM bytes resident in HBM
    ctx.stage_bam(bam)
    for _ in range(a.warmup):
        step()
    barrier()
    ctx.timing_accumulate(True)  # the library sums its HIP-event / host timers over the timed steps; read once afterwards
    t0 = time.perf_counter()
    for _ in range(a.steps):
        text = step()
    barrier()
    elapsed = time.perf_counter() - t0
    e2e: dict[str, dict] = {k: dict(v) for k, v in ctx.timing().items()}
    n_aln = ctx.counts()["n_concordant"] + (ctx.counts()["n_chimeric_records"] if (not sharded or rank == 0) else 0)
    n_conc, n_blk = ctx.counts()["n_concordant"], ctx.counts()["n_blocks"]

    # ---- graph pass alone over the resident records: per-kernel figures for the roofline
    ctx.timing_accumulate(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.resident_steps):
        ctx.reset()
        graph_pass()
    barrier()
    t_res = (time.perf_counter() - t0) / max(1, a.resident_steps)
    agg: dict[str, dict] = {k: dict(v) for k, v in ctx.timing().items()}

    # ---- the same step from the files in the page cache (host -> device copy of the compressed bytes inside the step), in a
    # fresh context and AFTER the timed region: a context that has copied the mapped file to the device runs its later
    # ingests ~20 ms slower (measured; tools/bench_step_parts.py), which is not what `value` is defined on
    ctx.close()
    ctx = squid_amd.Context(device=local_rank, rank=rank, world_size=world, exchange=exchange) if sharded else squid_amd.Context(device=local_rank)
    step()
    barrier()
    t0 = time.perf_counter()
    n_file_steps = 2
    for _ in range(n_file_steps):
        step()
    barrier()
    t_file = (time.perf_counter() - t0) / n_file_steps

    from squid_amd.dist import reduce_timing

    elapsed, total_aln = reduce_timing(elapsed, float(n_aln), dist, device="cuda")
    t_file, total_conc = reduce_timing(t_file, float(n_conc), dist, device="cuda")
    t_res, total_blk = reduce_timing(t_res, float(n_blk), dist, device="cuda")
    if rank != 0:
        ctx.close()
        if dist:
            dist.destroy_process_group()
        return

    value = total_aln * a.steps / elapsed
    R = max(1, a.resident_steps)
    # dominant record-streaming kernel of the graph pass, by accumulated HIP-event time on the library stream
    gk = {k: v for k, v in agg.items() if k.startswith(GPU_KERNELS_PREFIX) and v["bytes"] > 0 and k not in INGEST_KERNELS}
    dom = max(gk, key=lambda k: gk[k]["ms"])
    d = gk[dom]
    per_launch_bytes = d["bytes"] / d["launches"]
    per_launch_ms = d["ms"] / d["launches"]
    achieved = per_launch_bytes / (per_launch_ms * 1e-3) / 1e9
    gpu_ms = sum(v["ms"] for v in gk.values()) / R  # the record-streaming kernels (K1-K5, K10), not the small-graph ones
    scan_bytes = sum(v["bytes"] for v in gk.values()) / R
    # N_c * (80 + 24 * blocks per record) algorithmic bytes over the summed time of the record-streaming kernels
    bbar = total_blk / max(1.0, total_conc)
    sec8d_bytes = total_conc / world * (80.0 + 24.0 * bbar) if sharded else n_conc * (80.0 + 24.0 * bbar)
    traffic = None
    tfile = ROOT / "profiles" / "pmc_traffic.json"  # written by tools/profile_pmc.sh from the rocprofv3 --pmc passes
    if tfile.exists():
        try:
            tj = json.loads(tfile.read_text())
            if tj.get("_workload") == a.workload and not a.records and world == 1:  # the counters were collected on this workload
                traffic = (tj.get(dom) or {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    ing = {k: e2e[k] for k in INGEST_KERNELS if k in e2e and e2e[k]["ms"] > 0}
    out = {
        "metric": "paired-end alignments/sec BAM->_sv.txt (BGZF/BAM decode included; bit-exact SV calls vs CPU oracle)",
        "value": value, "unit": "alignments/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None,
        "dtype": "int32", "data": "synthetic",
        "config": {"workload": f"{a.workload}: " + WORKLOADS.get(a.workload, "generator config " + a.workload) + (f", --records {a.records}" if a.records else ""),
                   "records": int(total_aln), "records_per_gpu": int(total_aln / world), "blocks_per_record": round(bbar, 4),
                   "parallelism": ("one sample sharded by chromosome over %d ranks, %d all-gathers (%.0f bytes) per step" % (world, exchange.calls // max(1, a.steps + a.warmup + n_file_steps + 1 + a.resident_steps), exchange.bytes / max(1, exchange.calls)) if sharded else "1 sample per GPU, no collective") if world > 1 else "single GPU",
                   "step": "chimeric BAM decode (host) + concordant BAM decode on the GPU (BGZF inflate, record boundaries, record parse) + graph + ordering + SV calls + _sv.txt written; compressed BAM bytes resident in HBM at the start of every step"},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "bytes_per_launch": per_launch_bytes, "us_per_launch": per_launch_ms * 1e3,
                     "survey_8d": {"algorithmic_bytes_per_pass": sec8d_bytes, "scan_kernels_ms_per_pass": gpu_ms,
                                   "achieved_GBs": sec8d_bytes / (gpu_ms * 1e-3) / 1e9 if gpu_ms > 0 else None,
                                   "frac": sec8d_bytes / (gpu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if gpu_ms > 0 else None,
                                   "sum_of_kernel_bytes_per_pass": scan_bytes},
                     # every record-streaming kernel of the pass: us per launch, achieved GB/s of its algorithmic bytes, fraction of the HBM peak
                     "record_kernels": {k: {"us": round(v["ms"] / v["launches"] * 1e3, 1), "GBs": round(v["bytes"] / v["ms"] / 1e6, 1), "frac": round(v["bytes"] / v["ms"] / 1e6 / HBM_PEAK_GBS, 3)}
                                        for k, v in sorted(gk.items(), key=lambda kv: -kv[1]["ms"]) if v["ms"] > 0},
                     "note": "dominant record-streaming kernel of the graph pass (SURVEY.md 8(d) passes P1-P3); the BGZF inflate kernels are latency-bound bit-serial decoders, listed under ingest_kernels"},
        "ingest_kernels": {k: {"ms_per_step": round(v["ms"] / a.steps, 3), "GBs": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1)} for k, v in ing.items()},
        "from_file_value": total_aln / t_file, "from_file_note": "same step with the BAM read from the page cache: host->device copy of the compressed bytes inside the step",
        "resident_pass_value": total_aln / t_res, "resident_pass_ms": t_res * 1e3,
        "stage_ms_per_step": {k: round(v["ms"] / a.steps, 4) for k, v in sorted(e2e.items(), key=lambda kv: -kv[1]["ms"])[:(None if os.environ.get("BENCH_ALL_STAGES") else 14)]},
        "resident_stage_ms": {k: round(v["ms"] / R, 4) for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:(None if os.environ.get("BENCH_ALL_STAGES") else 16)]},
        "synth_s": round(t_gen, 1),
    }
    ctx.close()
    if not a.no_cpu_baseline and world == 1:
        # CPU oracle (a port of the reference, 1 thread, pinned) on a bounded sample of the same workload, timed on this box;
        # the GPU path runs the same sample and the two _sv.txt files are compared
        spre = synth(a.workload, seed, work, a.cpu_sample_records)
        pin = ["taskset", "-c", "0"] if shutil.which("taskset") else []
        t0 = time.perf_counter()
        subprocess.check_call(pin + [str(BUILD / "squid_oracle"), "-b", f"{spre}.bam", "-c", f"{spre}.chim.bam", "-o", str(work / "cpu_baseline")], stdout=subprocess.DEVNULL)
        tc = time.perf_counter() - t0
        res = squid_amd.run_pipeline(f"{spre}.bam", f"{spre}.chim.bam", device=local_rank)
        n_s = res["counts"]["n_concordant"] + res["counts"]["n_chimeric_records"]
        same = (work / "cpu_baseline_sv.txt").read_text() == res["sv_text"]
        out["cpu_baseline"] = {"value": n_s / tc, "unit": "alignments/s", "cores": 1, "kind": "port",
                               "sample": f"{a.workload} generated with --records {a.cpu_sample_records} ({n_s} records), BAM files -> _sv.txt incl. its three BAM decodes, {tc:.2f} s, " + ("taskset -c 0" if pin else "unpinned"),
                               "cpu_model": cpu_model(), "host_cpus": os.cpu_count(), "sv_identical_to_gpu": same, "sv_rows": res["sv_text"].count("\n") - 1}
    if dist:
        dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
