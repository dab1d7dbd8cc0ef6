"""CPU: the flag parsers of the oracle and of the product CLI against the REAL reference parser.

src/Config.cpp is the one translation unit of the reference that builds here without BamTools/GLPK/Boost; it is
compiled from where it lies (oracle/Makefile target `ref` -> oracle/_ref/ref_config, never copied into the repo).
On the GPU box the reference tree is absent and the prebuilt binary travels with the snapshot."""
import subprocess

import pytest

import squid_amd

REF = squid_amd.ROOT / "oracle" / "_ref" / "ref_config"

CASES = [
    ["-b", "a.bam", "-c", "c.bam", "-o", "out"],
    ["-b", "a.bam", "-c", "c.bam", "-o", "out", "-w", "1", "-a", "50", "-r", "1.5", "-dp", "2000", "-di", "3"],
    ["-b", "a.bam", "-c", "c.bam", "-o", "out", "-mq", "7", "-pt", "0", "-pl", "5", "-pm", "10"],
    ["-b", "a.bam", "-o", "out"],                                     # STAR without -c: error
    ["-b", "a.bam", "-o", "out", "--bwa"],                            # bwa: -c not needed, mq stays 1
    ["-b", "a.bam", "-c", "c.bam", "-o", "out", "-G", "1", "-CO", "1", "-TO", "1"],
    ["-b", "a.bam", "-c", "c.bam", "-o", "out", "-G", "2"],           # bad bool
    ["-b", "a.bam", "-c", "c.bam", "-o", "out", "-RG", "1"],          # needs -f
    ["-b", "a.bam", "-c", "c.bam", "-o", "out", "-w", "3", "-w", "9"],  # last occurrence wins (ledger B3)
    ["-b", "a.bam", "-c", "c.bam", "-o", "out", "-w"],                # flag in last position is ignored
    ["-b", "a.bam", "-c", "c.bam", "-o", "out", "-w", "abc", "-r", "x"],  # atoi/atof never fail
    ["-o", "out", "-c", "c.bam"],                                     # no -b
    ["-b", "a.bam", "-c", "c.bam", "-o", "out", "-pl", "70000", "-pm", "300"],  # uint16/uint8 truncation
]


def _line(cmd):
    out = subprocess.run(cmd, capture_output=True, text=True).stdout
    return [l for l in out.splitlines() if l.startswith("ok=")][-1]


@pytest.mark.skipif(not REF.exists(), reason="oracle/_ref/ref_config not built (make -C oracle ref)")
@pytest.mark.parametrize("argv", CASES)
def test_parsers_agree_with_reference_config_cpp(built, argv):
    want = _line([str(REF)] + argv)
    assert _line([str(built / "squid_oracle"), "--print-config"] + argv) == want
    assert _line([str(built / "squid"), "--print-config"] + argv) == want
