"""CPU tests of the oracle (test infrastructure) against hand-derived known answers.

The reference ships no tests or golden vectors, so these KATs are derived by hand from the
reference source lines they cite; they pin the oracle's reading of those lines."""
import subprocess

import bamwriter as bw
import oracle_util as ou


def _run(built, tmp_path, chim_records, conc_records, contigs=(("chrA", 100000), ("chrB", 50000)), flags=()):
    pre = tmp_path / "kat"
    bw.write_bam(f"{pre}.bam", contigs, conc_records)
    bw.write_bam(f"{pre}.chim.bam", contigs, chim_records, sort_order="unsorted")
    # the hand-made inputs are far too sparse for a graph: the run stops where the reference would assert
    # (SegmentGraph.cpp:2537), after the chimeric-record dump these tests read has been written
    return ou.run_oracle(built, pre, tmp_path / "out", *flags, check=False)


def _chim(dump):
    rows = {}
    for line in (dump / "chimrecord.txt").read_text().splitlines():
        if line.startswith("#"):
            readlen = int(line.split("=")[1])
            continue
        f = line.split("\t")
        blocks = {"F": [], "S": []}
        for part in f[5:]:
            toks = part.split(" ")
            blocks[toks[0]] = [tuple(int(x) for x in t.split(",")) for t in toks[1:]]
        rows[f[0]] = (int(f[1]), int(f[2]), blocks["F"], blocks["S"])
    return readlen, rows


def _background(n=40):
    # concordant proper pairs so that the graph stages have something to chew on
    recs = []
    for i in range(n):
        p = 1000 + 7 * i
        recs.append(bw.record(f"r{i}", 0, p, 255, 0x1 | 0x2 | 0x20 | 0x40, "100M", 0, p + 200))
    for i in range(n):
        p = 1200 + 7 * i
        recs.append(bw.record(f"r{i}", 0, p, 255, 0x1 | 0x2 | 0x10 | 0x80, "100M", 0, p - 200))
    return recs


def test_cigar_to_blocks_and_strand_mirroring(built, tmp_path):
    """ReadRec.cpp:45-87: S/H advance the read offset, a run starting at M extends over I/D/X until S/H/N,
    N advances the reference only, reverse-strand blocks get ReadPos = TotalLen - ReadPos - len."""
    chim = [
        # forward split read: 60M40S at 5000 ; mate piece 60H40M at chrB:7000 ; second mate 100M reverse
        bw.record("q1", 0, 5000, 255, 0x1 | 0x40, "60M40S"),
        bw.record("q1", 1, 7000, 255, 0x1 | 0x40 | 0x100, "60H40M"),
        bw.record("q1", 1, 7300, 255, 0x1 | 0x80 | 0x10, "100M"),
        # spliced + indels on the reverse strand: 10S 20M 5I 10M 3D 15M 1000N 40M
        bw.record("q2", 0, 20000, 255, 0x1 | 0x40 | 0x10, "10S20M5I10M3D15M1000N40M"),
        bw.record("q2", 0, 30000, 255, 0x1 | 0x80, "100M"),
        # padding so that the ReadLen median (first five records) is 100
        bw.record("q3", 0, 40000, 255, 0x1 | 0x40, "100M"),
        bw.record("q3", 0, 40300, 255, 0x1 | 0x80 | 0x10, "100M"),
    ]
    sv, dump = _run(built, tmp_path, chim, _background())
    readlen, rows = _chim(dump)
    assert readlen == 100
    tot1, tot2, f, s = rows["q1"]
    assert (tot1, tot2) == (100, 100)
    # blocks: (RefID, RefPos, ReadPos, MatchRef, MatchRead, IsReverse), sorted by ReadPos (ReadRec.cpp:143-146)
    assert f == [(0, 5000, 0, 60, 60, 0), (1, 7000, 60, 40, 40, 0)]
    assert s == [(1, 7300, 0, 100, 100, 1)]
    tot1, tot2, f, s = rows["q2"]
    # TotalLen = 10+20+5+10+15+40 = 100 (D and N excluded).  First run: read 20+5+10+15=50, ref 20+10+3+15=48;
    # reverse strand => ReadPos = 100-10-50 = 40.  After the N: ref 20000+48+1000 = 21048, read offset 60 -> 100-60-40 = 0
    assert tot1 == 100
    assert f == [(0, 21048, 0, 40, 40, 1), (0, 20000, 40, 48, 50, 1)]


def test_polya_blocks_are_dropped_but_still_advance(built, tmp_path):
    """ReadRec.cpp:62-82 (ledger B6): a block with >= 75 % A (or T) is not stored, positions still move on."""
    seq = "A" * 45 + "C" * 15 + "ACGT" * 10  # first 60-base block is 75 % A -> dropped (not < 0.75)
    chim = [
        bw.record("p1", 0, 5000, 255, 0x1 | 0x40, "60M500N40M", seq=seq),
        bw.record("p1", 0, 9000, 255, 0x1 | 0x80 | 0x10, "100M"),
        bw.record("p2", 0, 40000, 255, 0x1 | 0x40, "100M"),
        bw.record("p2", 0, 40300, 255, 0x1 | 0x80 | 0x10, "100M"),
        bw.record("p3", 0, 41000, 255, 0x1 | 0x40, "100M"),
    ]
    sv, dump = _run(built, tmp_path, chim, _background())
    _, rows = _chim(dump)
    assert rows["p1"][2] == [(0, 5560, 60, 40, 40, 0)]


def test_low_phred_run_and_name_suffix(built, tmp_path):
    """ReadRec.cpp:12-13 strips /1 and /2; :19-44 flags a run of > Max_LowPhred_Len qualities below 33+Min_Phred."""
    lowq = [30] * 20 + [2] * 11 + [30] * 69
    okq = [30] * 20 + [2] * 10 + [30] * 70
    chim = [
        bw.record("n1/1", 0, 5000, 255, 0x1 | 0x40, "100M", qual=lowq),
        bw.record("n1/2", 0, 5400, 255, 0x1 | 0x80 | 0x10, "100M", qual=okq),
        bw.record("n2", 0, 40000, 255, 0x1 | 0x40, "100M"),
        bw.record("n2", 0, 40300, 255, 0x1 | 0x80 | 0x10, "100M"),
        bw.record("n3", 0, 41000, 255, 0x1 | 0x40, "100M"),
    ]
    sv, dump = _run(built, tmp_path, chim, _background())
    text = (dump / "chimrecord.txt").read_text()
    line = [l for l in text.splitlines() if l.startswith("n1\t")][0].split("\t")
    assert line[3:5] == ["1", "0"]  # first mate low-Phred, second not
    assert not any(l.startswith("n1/") for l in text.splitlines())


def test_duplicate_and_unmapped_chimeric_records_are_ignored(built, tmp_path):
    """ReadRec.cpp:344"""
    chim = [
        bw.record("d1", 0, 5000, 255, 0x1 | 0x40 | 0x400, "100M"),
        bw.record("d2", -1, -1, 0, 0x1 | 0x4 | 0x40, "100S"),
        bw.record("k1", 0, 40000, 255, 0x1 | 0x40, "100M"),
        bw.record("k1", 0, 40300, 255, 0x1 | 0x80 | 0x10, "100M"),
    ]
    sv, dump = _run(built, tmp_path, chim, _background())
    _, rows = _chim(dump)
    assert set(rows) == {"k1"}


def test_exact_solvers_agree(built):
    """the brute-force enumerator and the branch-and-bound solver pick the same canonical optimum"""
    out = subprocess.check_output([str(built / "squid_oracle"), "--selftest", "1500"]).decode()
    assert "selftest OK" in out


def test_argument_errors_print_and_exit_zero(built):
    """Config.cpp:213-229 / main.cpp: bad arguments print 'Check your argument.' and the process exits 0"""
    p = subprocess.run([str(built / "squid_oracle"), "-b", "x.bam"], capture_output=True, text=True)
    assert p.returncode == 0 and "Check your argument." in p.stdout


def test_linear_time_bridge_search_agrees_with_the_brute_force(built, synth, tmp_path, monkeypatch):
    """components above 64 nodes use a chain-decomposition bridge search in the oracle; ORACLE_BRIDGE_CHECK makes it
    run the quadratic brute force next to it on every call and abort on any difference (dense parameter set: ~300
    splits of a ~2000-node component)"""
    import subprocess

    pre = synth("C5g", "--records", "200000", "--tsv", "400")
    monkeypatch.setenv("ORACLE_BRIDGE_CHECK", "1")
    out = tmp_path / "chk"
    subprocess.check_call([str(built / "squid_oracle"), "-b", f"{pre}.bam", "-c", f"{pre}.chim.bam", "-o", str(out), "-w", "1", "-a", "50"], stdout=subprocess.DEVNULL)
    monkeypatch.delenv("ORACLE_BRIDGE_CHECK")
    out2 = tmp_path / "plain"
    subprocess.check_call([str(built / "squid_oracle"), "-b", f"{pre}.bam", "-c", f"{pre}.chim.bam", "-o", str(out2), "-w", "1", "-a", "50"], stdout=subprocess.DEVNULL)
    assert (tmp_path / "chk_sv.txt").read_text() == (tmp_path / "plain_sv.txt").read_text()
