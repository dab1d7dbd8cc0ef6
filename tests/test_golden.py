"""CPU: the oracle reproduces the committed golden outputs (tests/golden/, made by tests/golden/make_golden.py).

There is no reference-provided golden vector (the reference has no tests and cannot be built here), so these
files are regression pins of the oracle itself; DESIGN.md states "parity unpinned"."""
from pathlib import Path

import pytest

import oracle_util as ou

GOLD = Path(__file__).resolve().parent / "golden"


@pytest.mark.parametrize("cfg", ["C1", "T2"])
def test_oracle_matches_golden(built, synth, tmp_path, cfg):
    sv, dump = ou.run_oracle(built, synth(cfg), tmp_path)
    assert sv.read_text() == (GOLD / f"{cfg}_sv.txt").read_text()
    for name in ["nodes_build.txt", "edges_build.txt", "edges_filter.txt", "nodes_final.txt", "edges_final.txt", "orders.txt", "breakpoints.txt"]:
        assert (dump / name).read_text() == (GOLD / f"{cfg}_{name}").read_text(), name


def test_oracle_recovers_planted_junctions(built, synth, tmp_path):
    """sanity of the synthetic data: most planted junctions come back with their exact breakpoints"""
    pre = synth("C1")
    sv, _ = ou.run_oracle(built, pre, tmp_path)
    truth = [l.split("\t") for l in Path(f"{pre}.truth.txt").read_text().splitlines()[1:]]
    called = set()
    for l in sv.read_text().splitlines()[1:]:
        f = l.split("\t")
        called.add(frozenset([(f[0], int(f[1])), (f[0], int(f[2])), (f[3], int(f[4])), (f[3], int(f[5]))]))
    hits = sum(any((t[0], int(t[1])) in c and (t[3], int(t[4])) in c for c in called) for t in truth)
    assert hits >= len(truth) - 1
