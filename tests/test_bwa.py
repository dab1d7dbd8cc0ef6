"""`squid --bwa`: one coordinate-sorted BAM, split reads as supplementary records, no chimeric file.
BuildNode_BWA (src/SegmentGraph.cpp:833-1205) and RawEdges (:1698-1930) replace the STAR node builder and edge generators; everything
behind BuildEdges' sort is shared.  CPU part: the oracle's restatement finds the planted junctions and is pinned by golden files;
GPU part: the HIP library (host loops of sq_bwa.cpp + the graph kernels) against the oracle stage by stage, and the command line."""
import subprocess
from pathlib import Path

import pytest

import oracle_util as ou

GOLD = Path(__file__).resolve().parent / "golden"


def _oracle_bwa(built, pre, outdir, *flags):
    outdir = Path(outdir)
    dump = outdir / "dump"
    dump.mkdir(parents=True, exist_ok=True)
    subprocess.check_call([str(built / "squid_oracle"), "--bwa", "-b", f"{pre}.bam", "-o", str(outdir / "oracle"), "--dump", str(dump), *flags], stdout=subprocess.DEVNULL)
    return outdir / "oracle_sv.txt", dump


def _rows(text):
    return [line.split("\t") for line in text.splitlines() if line and not line.startswith("#")]


@pytest.mark.parametrize("cfg,bwa", [("C1", True), ("T2", True), ("C1", False), ("T2", False)])
def test_oracle_calls_the_planted_junctions(built, synth, tmp_path, cfg, bwa):
    """(both modes: the STAR path gets the same check here)  a known answer that does not come from the restatement itself: the junctions the generator planted (truth.txt: chromosome,
    breakpoint and end type of both sides) -- every SV row of the oracle must sit on one of them, exact to the base where split reads
    support it, and most planted junctions must be found"""
    if bwa:
        pre = synth(cfg, "--bwa")
        sv_path, dump = _oracle_bwa(built, pre, tmp_path)
    else:
        pre = synth(cfg)
        sv_path, dump = ou.run_oracle(built, pre, tmp_path)
    truth = []
    for f in _rows(Path(f"{pre}.truth.txt").read_text()):
        truth.append(((f[0], int(f[1]), f[2]), (f[3], int(f[4]), f[5])))
    rows = _rows(sv_path.read_text())
    assert len(rows) >= max(3, (len(truth) * 2) // 3)
    hit = set()
    for r in rows:
        # a row's side is "-" when the segment's HEAD (its start) is the junction end, "+" when its tail (end) is
        sides = [(r[0], int(r[1]) if r[8] == "-" else int(r[2]), "H" if r[8] == "-" else "T"), (r[3], int(r[4]) if r[9] == "-" else int(r[5]), "H" if r[9] == "-" else "T")]
        match = [k for k, (x, y) in enumerate(truth) if {x, y} == set(sides) or (x == sides[0] and y == sides[1]) or (x == sides[1] and y == sides[0])]
        assert match, f"SV row {r[:6]} {r[8:10]} is not a planted junction"
        hit.add(match[0])
    assert len(hit) == len(rows)  # no junction called twice
    stats = dict(line.split("\t") for line in (dump / "order_stats.txt").read_text().splitlines())
    assert stats["ambiguous"] == "0"


@pytest.mark.parametrize("cfg", ["C1", "T2"])
def test_oracle_bwa_mode_golden_files(built, synth, tmp_path, cfg):
    pre = synth(cfg, "--bwa")
    sv_path, dump = _oracle_bwa(built, pre, tmp_path)
    assert sv_path.read_text() == (GOLD / f"{cfg}bwa_sv.txt").read_text()
    assert (dump / "orders.txt").read_text() == (GOLD / f"{cfg}bwa_orders.txt").read_text()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,extra,flags,params", [
    ("C1", (), (), {}),
    ("T2", (), (), {}),
    ("T2", ("--seed", "4242"), (), {}),
    ("C2", (), (), {}),
    ("T2", (), ("-w", "2", "-a", "20", "-mq", "30"), dict(min_edge_weight=2, max_allowed_degree=20, min_mapqual=30)),
])
def test_stage_parity_bwa(built, synth, tmp_path, cfg, extra, flags, params):
    import squid_amd
    from test_gpu_parity import _compare

    pre = synth(cfg, "--bwa", *extra)
    sv_path, dump = _oracle_bwa(built, pre, tmp_path, *flags)
    kw = dict(min_mapqual=1)
    kw.update(params)
    with squid_amd.Context(star_mapq=False, **kw) as ctx:
        ctx.load_bwa(f"{pre}.bam")
        ctx.build_graph()
        sv = _compare(ctx, dump, sv_path)
        assert sv.count("\n") > 1
        # the fragments RawEdges rebuilt from the partially aligned reads (:1883-1926)
        want_frags = sum(1 for line in (dump / "chimrecord.txt").read_text().splitlines() if not line.startswith("#"))
        assert ctx.counts()["n_chim_fragments"] == want_frags > 0
        ctx.reset()
        ctx.build_graph()
        ctx.order()
        assert ctx.sv_text() == sv


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["T2", "C2"])
def test_bwa_batch_through_the_gpu_reader(built, synth, tmp_path, monkeypatch, cfg):
    """SQUID_BWA_GPU=1 (what a --bwa file of 1 GiB and more gets): the records AND their QNAMEs are inflated and parsed on the device
    (K-1 + K0 with the names kept, as for a large chimeric BAM) and come back as the batch the host decoder makes -- every stage, the
    rebuilt fragments and _sv.txt equal the oracle's, as they do on the host route"""
    import squid_amd
    from test_gpu_parity import _compare

    pre = synth(cfg, "--bwa")
    sv_path, dump = _oracle_bwa(built, pre, tmp_path)
    texts = {}
    for route in ("1", "0"):
        monkeypatch.setenv("SQUID_BWA_GPU", route)
        with squid_amd.Context(star_mapq=False, min_mapqual=1) as ctx:
            ctx.load_bwa(f"{pre}.bam")
            assert ctx.counts()["chimeric_through_gpu_reader"] == int(route)
            ctx.build_graph()
            texts[route] = _compare(ctx, dump, sv_path)
            assert ctx.counts()["n_chim_fragments"] == sum(1 for line in (dump / "chimrecord.txt").read_text().splitlines() if not line.startswith("#"))
    assert texts["0"] == texts["1"] and texts["1"].count("\n") > 1


@pytest.mark.parametrize("cfg,extra", [("T2", ()), ("C2", ()), ("T2", ("--seed", "4242")), ("C3", ("--records", "400000"))])
def test_bwa_record_loops_in_stretches_equal_the_loops_in_one_go(built, synth, tmp_path, cfg, extra):
    """host only (tools/bwa_pieces_check.cpp, no device): BuildNode_BWA's automaton cut at coverage gaps -- every stretch started on a guess
    of the few values that cross a gap, checked in order, run again where the guess was wrong (T2: DiscordantRightmost of chromosome 0
    decides the zero-coverage test on chromosome 1) -- and RawEdges' loop cut behind records that pin LocateRead's position give the
    nodes with their Support / AvgDepth, the raw edges and the rebuilt fragments of the loops run in one go, for several stretch lengths"""
    import squid_amd

    exe = tmp_path / "bwa_pieces_check"
    subprocess.check_call(["hipcc", "-O1", "-std=c++17", "-I", str(squid_amd.ROOT / "include"), "-o", str(exe), str(squid_amd.ROOT / "tools" / "bwa_pieces_check.cpp"),
                           "-L", str(built), "-lsquid_hip", f"-Wl,-rpath,{built}", "-lpthread"], stderr=subprocess.DEVNULL)
    pre = synth(cfg, "--bwa", *extra)
    for piece in ("37", "300", "5000"):
        out = subprocess.run([str(exe), f"{pre}.bam", piece, "5"], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and out.stdout.strip().endswith("same"), (piece, out.stdout[-2000:])


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,piece", [("T2", "300"), ("C2", "1000"), ("C2", "97")])
def test_bwa_record_loops_in_stretches_on_the_host_threads(built, synth, tmp_path, monkeypatch, cfg, piece):
    """SQUID_BWA_PIECE=<records>: the BAM loop of RawEdges cut into stretches that start behind a record whose first block pins
    LocateRead's position, and BuildNode_BWA's loop cut at the zero-coverage gaps its state empties at -- what a sample of tens of
    millions of records gets by itself; every stage, the rebuilt fragments and _sv.txt equal the oracle's"""
    import squid_amd
    from test_gpu_parity import _compare

    pre = synth(cfg, "--bwa")
    sv_path, dump = _oracle_bwa(built, pre, tmp_path)
    monkeypatch.setenv("SQUID_BWA_PIECE", piece)
    with squid_amd.Context(star_mapq=False, min_mapqual=1) as ctx:
        ctx.load_bwa(f"{pre}.bam")
        ctx.build_graph()
        sv = _compare(ctx, dump, sv_path)
        assert sv.count("\n") > 1
        assert ctx.counts()["n_chim_fragments"] == sum(1 for line in (dump / "chimrecord.txt").read_text().splitlines() if not line.startswith("#"))
        t = ctx.timing()
        stretches = (t.get("bwa_raw_edge_stretches", {}).get("launches", 0), t.get("bwa_seed_node_stretches", {}).get("launches", 0))
        assert stretches[0] > 3 and stretches[1] > 3, stretches
        assert "bwa_bp_support_stretches_walked_again" in t  # (ExactBPConcordantSupport's walk took the stretched form too; the breakpoint table was compared above)


@pytest.mark.gpu
def test_bwa_command_line_is_a_drop_in(built, synth, tmp_path):
    pre = synth("T2", "--bwa")
    subprocess.check_call([str(built / "squid_oracle"), "--bwa", "-b", f"{pre}.bam", "-o", str(tmp_path / "o"), "-G", "1", "-CO", "1"], stdout=subprocess.DEVNULL)
    subprocess.check_call([str(built / "squid"), "--bwa", "-b", f"{pre}.bam", "-o", str(tmp_path / "p"), "-G", "1", "-CO", "1"], stdout=subprocess.DEVNULL)
    for suffix in ("_sv.txt", "_graph.txt", "_component_pri.txt"):
        assert (tmp_path / f"p{suffix}").read_bytes() == (tmp_path / f"o{suffix}").read_bytes(), suffix
