"""utils/JunctionSequence.cpp counterpart (second half): `build/squid_junction` = sq_junction_sequences of
the library, host work only (runs without a GPU).  PARITY UNPINNED against the reference (it needs Boost + BamTools); what is here:
a known answer derived by hand from the reference's statements, and the product against the oracle's line-by-line restatement."""
import random
import subprocess
from pathlib import Path

import pytest

import bamwriter as bw


def _run(tool, *args):
    subprocess.check_call([str(tool), *[str(a) for a in args]], stdout=subprocess.DEVNULL)


def test_hand_derived_junction_sequences(built, tmp_path):
    """Two chimeric fragments whose first mates are split chrA -> chrB, and one call that spans them.
    By utils/JunctionSequence.cpp: each split gives a read junction {(chrA, start, end, IsLeft = IsReverse = false), (chrB, start, end,
    IsLeft = !IsReverse = true)} (:112-168); both lie within [-300, +5] / [-5, +300] of the call's ends (:170-200).  The call's first end
    is a right end (strand +): narrowed to the leftmost start of the reads that reach it, max(70, 50) = 70 (:298-311); its second end a
    left end (strand -): min(94, 80) = 80 (:312-325).  Both reads hit both ends within 5 bases: support min(2, 2) (:262-275,341-345).
    The second read ends at 98, not 100: an alternative junction point chrA:70:98 (:347-391).  Relaxed: a right end grows 1000 bases
    to the left (to 0), a left end 1000 to the right (to the chromosome's end, 200) (:461-470).  Right end = forward sequence, left
    second end = forward sequence (:435-439)."""
    rng = random.Random(11)
    seq = {"chrA": "".join(rng.choice("ACGT") for _ in range(200)), "chrB": "".join(rng.choice("ACGT") for _ in range(200))}
    fa = tmp_path / "g.fa"
    fa.write_text("".join(f">{n} some text\n" + "\n".join(s[i:i + 50] for i in range(0, 200, 50)) + "\n" for n, s in seq.items()))
    contigs = [("chrA", 200), ("chrB", 200)]
    chim = [
        # fragment q1: first mate = 30 bases on chrA [70, 100) then 70 bases on chrB [20, 90); second mate on chrB
        bw.record("q1", 0, 70, 255, 0x1 | 0x40, "30M70S"), bw.record("q1", 1, 20, 255, 0x1 | 0x40 | 0x100, "30H70M"),
        bw.record("q1", 1, 100, 255, 0x1 | 0x10 | 0x80, "100M"),
        # fragment q2: 26 bases on chrA [72, 98), 74 on chrB [20, 94)
        bw.record("q2", 0, 72, 255, 0x1 | 0x40, "26M74S"), bw.record("q2", 1, 20, 255, 0x1 | 0x40 | 0x100, "26H74M"),
        bw.record("q2", 1, 100, 255, 0x1 | 0x10 | 0x80, "100M"),
    ]
    bw.write_bam(tmp_path / "c.bam", contigs, chim, sort_order="unsorted")
    sv = tmp_path / "x_sv.txt"
    sv.write_text("# chrom1\tstart1\tend1\tchrom2\tstart2\tend2\tname\tscore\tstrand1\tstrand2\tnum_concordantfrag_bp1\tnum_concordantfrag_bp2\n"
                  "chrA\t50\t100\tchrB\t20\t80\t.\t7\t+\t-\t3\t4\n")
    _run(built / "squid_junction", sv, tmp_path / "c.bam", fa, tmp_path / "k")
    a, b = seq["chrA"], seq["chrB"]

    def fasta(head, s):
        return head + "\n" + "".join(s[i:i + 80] + "\n" for i in range(0, len(s), 80))
    assert (tmp_path / "k_junc_precise.fa").read_text() == fasta(">squid_0 chrA:70:100:+ chrB:20:80:+ 2", a[70:100] + b[20:80])
    assert (tmp_path / "k_junc_relax.fa").read_text() == fasta(">squid_0 chrA:0:100:+ chrB:20:200:+", a[0:100] + b[20:200])
    assert (tmp_path / "k_junc_alt.fa").read_text() == fasta(">squid_0_alt_1 chrA:70:98:+ chrB:20:80:+ 2", a[70:98] + b[20:80])
    # the oracle's restatement says the same
    _run(built / "squid_oracle", "--junction", sv, tmp_path / "c.bam", fa, tmp_path / "o")
    for kind in ("precise", "relax", "alt"):
        assert (tmp_path / f"o_junc_{kind}.fa").read_bytes() == (tmp_path / f"k_junc_{kind}.fa").read_bytes()


@pytest.mark.parametrize("cfg", ["C1", "T2"])
def test_junction_sequences_equal_the_oracle(built, synth, tmp_path, cfg):
    """the calls of a whole synthetic sample (the oracle's _sv.txt), reverse-complemented left first ends and right second ends included;
    a FASTA with lower case, IUPAC codes, N runs, ragged lines and a contig the BAM does not know"""
    from test_gpu_parity import _fasta_for

    pre = synth(cfg)
    subprocess.check_call([str(built / "squid_oracle"), "-b", f"{pre}.bam", "-c", f"{pre}.chim.bam", "-o", str(tmp_path / "s")], stdout=subprocess.DEVNULL)
    fa = tmp_path / "g.fa"
    _fasta_for(pre, fa)
    _run(built / "squid_oracle", "--junction", tmp_path / "s_sv.txt", f"{pre}.chim.bam", fa, tmp_path / "o")
    _run(built / "squid_junction", tmp_path / "s_sv.txt", f"{pre}.chim.bam", fa, tmp_path / "p")
    n = 0
    for kind in ("precise", "relax", "alt"):
        want = (tmp_path / f"o_junc_{kind}.fa").read_bytes()
        assert (tmp_path / f"p_junc_{kind}.fa").read_bytes() == want, kind
        n += want.count(b">")
    assert n >= 6
    heads = [l for l in (tmp_path / "p_junc_precise.fa").read_text().splitlines() if l.startswith(">")]
    assert any(":-" in h.split(" ")[1] for h in heads) or any(h.split(" ")[2].endswith(":-") for h in heads)  # some end was reverse-complemented
