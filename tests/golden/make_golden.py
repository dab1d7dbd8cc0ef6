#!/usr/bin/env python3
"""Regenerates tests/golden/* : synthetic inputs (seeded generator) -> CPU oracle outputs.
Run from the repo root after `make`:  python tests/golden/make_golden.py"""
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "tests"))
import oracle_util as ou  # noqa: E402

for cfg in ["C1", "T2"]:
    with tempfile.TemporaryDirectory() as td:
        pre = Path(td) / cfg
        subprocess.check_call([str(ROOT / "build" / "gen_synth_bam"), "--config", cfg, "--out", str(pre)], stdout=subprocess.DEVNULL)
        sv, dump = ou.run_oracle(ROOT / "build", pre, td)
        shutil.copy(sv, ROOT / "tests" / "golden" / f"{cfg}_sv.txt")
        for name in ["nodes_build.txt", "edges_build.txt", "edges_filter.txt", "nodes_final.txt", "edges_final.txt", "orders.txt", "breakpoints.txt"]:
            shutil.copy(dump / name, ROOT / "tests" / "golden" / f"{cfg}_{name}")
print("golden files written")

# --bwa mode (one BAM, supplementary alignments): regression pins of the oracle's BuildNode_BWA / RawEdges restatement
for cfg in ["C1", "T2"]:
    with tempfile.TemporaryDirectory() as td:
        pre = Path(td) / cfg
        subprocess.check_call([str(ROOT / "build" / "gen_synth_bam"), "--config", cfg, "--bwa", "--out", str(pre)], stdout=subprocess.DEVNULL)
        dump = Path(td) / "dump"
        dump.mkdir()
        subprocess.check_call([str(ROOT / "build" / "squid_oracle"), "--bwa", "-b", f"{pre}.bam", "-o", str(Path(td) / "oracle"), "--dump", str(dump)], stdout=subprocess.DEVNULL)
        shutil.copy(Path(td) / "oracle_sv.txt", ROOT / "tests" / "golden" / f"{cfg}bwa_sv.txt")
        shutil.copy(dump / "orders.txt", ROOT / "tests" / "golden" / f"{cfg}bwa_orders.txt")
print("golden files of the --bwa mode written")
