"""The speculative token pass of the BGZF reader (squid_amd/csrc/sq_inflate_spec.inc, k_inflate_spec) on the CPU: the kernel source itself,
its 64 lanes as coroutines (sq_wave.h with SQ_WAVE_EMU, tools/inflate_emu.cpp), compared with zlib -- on the BGZF blocks of a synthetic
BAM for every stretch length / table size the tuning entry knows, and on fuzzed streams of every block type (stored, fixed, dynamic, flushes
in the middle, payloads at every offset from a 16-byte boundary), plus damaged streams, which may be flagged but must never write outside
their token slots.  Behind every block's token pass the emulator also runs the resolve the device runs (squid_amd/csrc/sq_resolve.inc, k_lz_resolve5: a round's bytes
staged in LDS) with staging areas of 496, 64 and 16 bytes -- rounds that fit, rounds that take the direct way, matches that overlap their own output -- and
compares its bytes with zlib's.  The GPU suite runs the same sources on the device against the host reader and zlib (SQUID_INFLATE_CHECK)."""
import subprocess

import pytest


@pytest.fixture(scope="module")
def emu(built, tmp_path_factory):
    exe = tmp_path_factory.mktemp("emu") / "inflate_emu"
    root = built.parent
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-DSQ_WAVE_EMU", "-Wno-unknown-pragmas", "-o", str(exe), str(root / "tools" / "inflate_emu.cpp"), "-lz"])
    return exe


def test_emulated_token_pass_inflates_bam_blocks_like_zlib(emu, synth):
    pre = synth("T2")
    for cfg in ("0", "1", "2", "3", "4", "5", "6"):  # (3 = <384, 10>, what the reader runs; 6 = <1024, 11>)
        out = subprocess.run([str(emu), f"{pre}.bam", "12" if cfg == "3" else "6", cfg], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and " 0 flagged, 0 WRONG" in out.stdout, (cfg, out.stdout, out.stderr[-2000:])


def test_emulated_token_pass_on_fuzzed_streams(emu):
    out = subprocess.run([str(emu), "--fuzz", "90", "20261004"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and " 0 flagged, 0 WRONG" in out.stdout, (out.stdout, out.stderr[-2000:])
    assert "accepted where zlib refuses" not in out.stderr


def test_emulated_staged_resolve_on_random_token_lists(emu):
    """token lists made here, not by a token pass -- matches of every length and distance the format allows, distances shorter than the match (overlaps),
    lists that are mostly literals or mostly 258-byte matches, lists that end inside a round -- through sq_resolve.inc with three staging sizes against a plain
    loop over the tokens; every fifth list is damaged (a distance beyond the block's start, a match that overruns the block) and must be refused"""
    out = subprocess.run([str(emu), "--resolve-fuzz", "160", "20261004"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and " 0 WRONG" in out.stdout, (out.stdout, out.stderr[-2000:])
    assert "128 identical to the plain resolve, 32 damaged lists refused" in out.stdout, out.stdout
