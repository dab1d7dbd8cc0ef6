#!/usr/bin/env python3
"""Stage-by-stage comparison of the HIP path with the CPU oracle on synthetic inputs (diagnostic script;
tests/test_gpu_parity.py asserts the same things)."""
import argparse
import subprocess
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle_util as ou  # noqa: E402
import squid_amd  # noqa: E402


def first_diff(a, b, name, limit=5):
    n = 0
    if len(a) != len(b):
        print(f"   {name}: length {len(a)} (hip) vs {len(b)} (oracle)")
        n += 1
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            print(f"   {name}[{i}]: hip={x} oracle={y}")
            n += 1
            if n >= limit:
                break
    return n == 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="C1,T2,C2")
    ap.add_argument("--flags", default="")
    ap.add_argument("--gen", default="", help="extra generator arguments, e.g. '--records 10000000'")
    ap.add_argument("--default-depth", action="store_true", help="do not force the exact depth sweep (node depth columns are then not compared)")
    a = ap.parse_args()
    import os
    if not a.default_depth:
        os.environ["SQUID_EXACT_DEPTH"] = "1"
    B = ROOT / "build"
    ok_all = True
    for cfg in a.configs.split(","):
        with tempfile.TemporaryDirectory() as td:
            pre = Path(td) / cfg
            print(subprocess.check_output([str(B / "gen_synth_bam"), "--config", cfg, "--out", str(pre), "--threads", "16", *a.gen.split()]).decode().strip())
            t0 = time.time()
            sv_path, dump = ou.run_oracle(B, pre, td, *a.flags.split())
            t_or = time.time() - t0
            with squid_amd.Context() as ctx:
                t0 = time.time(); ctx.load(f"{pre}.bam", f"{pre}.chim.bam"); t_load = time.time() - t0
                t0 = time.time(); ctx.build_graph(); t_graph = time.time() - t0
                ok = True
                g1 = ctx.graph(1)
                kk = 3 if a.default_depth else 5
                ok &= first_diff([n[:kk] for n in g1["nodes"]], [n[:kk] for n in ou.read_nodes(dump / "nodes_build.txt")], "nodes_build")
                ok &= first_diff(ctx.graph(2)["edges"][:], [e[:5] + (0,) for e in ou.read_edges(dump / "edges_build.txt")], "edges_build")
                ok &= first_diff(ctx.graph(3)["edges"], ou.read_edges(dump / "edges_weight.txt"), "edges_weight")
                ok &= first_diff(ctx.graph(4)["edges"], ou.read_edges(dump / "edges_filter.txt"), "edges_filter")
                g5 = ctx.graph(5)
                ok &= first_diff([n[:kk] for n in g5["nodes"]], [n[:kk] for n in ou.read_nodes(dump / "nodes_compress.txt")], "nodes_compress")
                g0 = ctx.graph(0)
                ok &= first_diff(g0["nodes"], ou.read_nodes(dump / "nodes_final.txt"), "nodes_final")
                ok &= first_diff(g0["edges"], ou.read_edges(dump / "edges_final.txt"), "edges_final")
                t0 = time.time(); orders = ctx.order(); t_ord = time.time() - t0
                ok &= first_diff(orders, ou.read_orders(dump / "orders.txt"), "orders")
                t0 = time.time(); sv = ctx.sv_text(); t_sv = time.time() - t0
                ok &= first_diff(ctx.breakpoints(), ou.read_breakpoints(dump / "breakpoints.txt"), "breakpoints")
                ok &= first_diff(sv.splitlines(), sv_path.read_text().splitlines(), "_sv.txt")
                cnt = ctx.counts()
                print(f"[{cfg}] {'PARITY OK' if ok else 'MISMATCH'}  records={cnt['n_concordant']} sv_rows={sv.count(chr(10)) - 1} "
                      f"oracle={t_or:.2f}s load={t_load:.2f}s graph={t_graph*1e3:.1f}ms order={t_ord*1e3:.1f}ms sv={t_sv*1e3:.1f}ms")
                for k, v in ctx.timing().items():
                    print(f"      {k:28s} {v['ms']:9.3f} ms x{v['launches']}")
                ok_all &= ok
    sys.exit(0 if ok_all else 1)


if __name__ == "__main__":
    main()
