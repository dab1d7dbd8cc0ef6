#!/usr/bin/env python3
"""which graph stage of the HIP path first differs from the oracle on a sample, with the differing rows (device filters, then host filters)
usage: filter_debug.py CONFIG [generator args ...] [-- squid flags ...]"""
import os, subprocess, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import oracle_util as ou
import squid_amd

args = sys.argv[1:]
flags = []
if "--" in args:
    k = args.index("--"); flags = args[k + 1:]; args = args[:k]
cfg, gen = args[0], args[1:]
if os.environ.get("FD_CHILD") is None:
    for host in ("", "1"):
        env = dict(os.environ, FD_CHILD="1", SQUID_EXACT_DEPTH="1")
        if host: env["SQUID_HOST_FILTERS"] = "1"
        print(f"==== {'host' if host else 'device'} filters", flush=True)
        subprocess.call([sys.executable, __file__] + sys.argv[1:], env=env)
    sys.exit(0)
td = Path(tempfile.mkdtemp())
pre = td / cfg
subprocess.check_call([str(ROOT / "build" / "gen_synth_bam"), "--config", cfg, "--out", str(pre), *gen], stdout=subprocess.DEVNULL)
sv_path, dump = ou.run_oracle(ROOT / "build", pre, td, *flags)
opts = dict(zip(flags[::2], flags[1::2]))
params = {}
if "-w" in opts: params["min_edge_weight"] = int(opts["-w"])
if "-a" in opts: params["max_allowed_degree"] = int(opts["-a"])
with squid_amd.Context(**params) as ctx:
    ctx.load(f"{pre}.bam", f"{pre}.chim.bam")
    ctx.build_graph()
    for stage, name in ((2, "edges_build.txt"), (3, "edges_weight.txt"), (4, "edges_filter.txt")):
        got = ctx.graph(stage)["edges"]
        want = ou.read_edges(dump / name)
        if stage == 2: want = [e[:5] + (0,) for e in want]
        want = [tuple(w[:6]) for w in want]
        print(f"stage {stage}: got {len(got)} want {len(want)} equal {got == want}")
        if got != want:
            gs, ws = set(got), set(want)
            print("  only HIP:", sorted(gs - ws)[:40]); print("  only oracle:", sorted(ws - gs)[:40])
            break
