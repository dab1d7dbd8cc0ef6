"""CPU: build/squid_annotate (counterpart of the reference's utils/AnnotateSQUIDOutput.py) against
outputs of the REAL script.  The fixtures under tests/golden/annotate/ are data: small synthetic GTF / _sv.txt inputs and
what the reference script wrote for them in the authoring container (generator: make_annotate_golden.py, committed next
to them).  Every byte must agree except the ORDER of the pairs in the FusedGenes column, which the script derives from
list(set(...)) -- Python's string hashing -- and is compared as a multiset."""
import subprocess
from collections import Counter

import pytest

import squid_amd

GOLD = squid_amd.ROOT / "tests" / "golden" / "annotate"
CASES = sorted(p.name[: -len("_expected.txt")] for p in GOLD.glob("*_expected.txt"))


def _rows(text):
    out = []
    for line in text.splitlines():
        f = line.split("\t")
        out.append((f[:-1], Counter(f[-1].split(","))))
    return out


@pytest.mark.parametrize("case", CASES)
def test_annotate_matches_the_reference_script(built, tmp_path, case):
    args = (GOLD / f"{case}.args").read_text().split()
    out = tmp_path / "out.txt"
    subprocess.run([str(built / "squid_annotate")] + args + [str(GOLD / f"{case}.gtf"), str(GOLD / f"{case}_sv.txt"), str(out)], check=True)
    got, want = out.read_text(), (GOLD / f"{case}_expected.txt").read_text()
    assert got.count("\n") == want.count("\n")
    assert _rows(got) == _rows(want)
    assert any("\tfusion-gene\t" in l for l in want.splitlines())  # the fixture exercises the join


def test_annotate_usage_and_argument_errors(built):
    assert "squid_annotate [options] <GTFfile> <SquidPrediction> <OutputFile>" in subprocess.run([str(built / "squid_annotate")], capture_output=True, text=True).stdout
    assert "Unknown argument --x" in subprocess.run([str(built / "squid_annotate"), "--x"], capture_output=True, text=True).stdout
    assert "Missing GTFfile" in subprocess.run([str(built / "squid_annotate"), "a.gtf", "b.txt"], capture_output=True, text=True).stdout
