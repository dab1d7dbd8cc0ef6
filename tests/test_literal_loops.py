"""Literal restatements of reference loops, written straight from the reference text (src/SegmentGraph.cpp, src/ReadRec.cpp) and
NOT from oracle/: statement-by-statement Python over the records the library holds in HBM (sq_debug_download).  They are slow
and small on purpose.  What they give that the oracle comparison cannot: oracle and product are two restatements by one author;
a shared misreading is invisible to a test that compares the two.  These loops are a third reading, kept as close to the
reference's statements as Python allows, of

* the pass-1 record filter, the mate stub and the consecutive-duplicate drop (`SegmentGraph.cpp:297-337`, `ReadRec_t::Equal`
  `ReadRec.cpp:119-141`) -> ReadsMain / ReadsOther;
* the discordant block list `bamdiscordant` (`:203-264`; `IsEndDiscordant / IsSingleAnchored / IsPairDiscordant`,
  `ReadRec.cpp:171-228`) from the merged chimeric fragments;
* per-node Support / AvgDepth (`:766-826`), three sweeps with their lagging cursors and the final division.

The only inputs taken from elsewhere: the merged chimeric fragments (the oracle's dump of BuildChimericSBamRecord's result), the
node coordinates of stage 1 and the number of kept records the stream loop consumes before its `break` (`:338-339`).
"""
import pytest

import oracle_util as ou
import squid_amd

pytestmark = pytest.mark.gpu


def _read_chimrecord(path):
    frags = []
    for line in open(path):
        if line.startswith("#"):
            continue
        f = line.rstrip("\n").split("\t")
        fr = {"ftl": int(f[1]), "stl": int(f[2]), "flow": int(f[3]), "slow": int(f[4]), "F": [], "S": []}
        for part in f[5:]:
            toks = part.split(" ")
            for b in toks[1:]:
                refid, refpos, readpos, matchref, matchread, rev = (int(x) for x in b.split(","))
                fr[toks[0]].append({"RefID": refid, "RefPos": refpos, "ReadPos": readpos, "MatchRef": matchref, "MatchRead": matchread, "IsReverse": bool(rev)})
        frags.append(fr)
    return frags


# ---- ReadRec.cpp:171-228
def _is_single_anchored(r):
    return len(r["F"]) == 0 or len(r["S"]) == 0  # (MultiFilter is false after the constructor, ReadRec.cpp:14)


def _is_end_discordant(r, first):
    L = r["F"] if first else r["S"]
    if len(L) <= 1:
        return False
    for i in range(len(L) - 1):
        a, b = L[i], L[i + 1]
        if a["RefID"] != b["RefID"] or a["IsReverse"] != b["IsReverse"]:
            return True
        elif not a["IsReverse"] and (a["RefPos"] < b["RefPos"]) != (a["ReadPos"] < b["ReadPos"]):
            return True
        elif a["IsReverse"] and (a["RefPos"] < b["RefPos"]) == (a["ReadPos"] < b["ReadPos"]):
            return True
    return False


def _is_pair_discordant(r):
    F, S = r["F"], r["S"]
    if len(F) == 0 or len(S) == 0:
        return False
    if _is_end_discordant(r, True) or _is_end_discordant(r, False):
        return True
    if F[0]["RefID"] != S[-1]["RefID"] or F[0]["IsReverse"] == S[-1]["IsReverse"]:
        return True
    elif not F[0]["IsReverse"] and F[0]["RefPos"] - F[0]["ReadPos"] > S[-1]["RefPos"] - (r["stl"] - S[-1]["ReadPos"] - S[-1]["MatchRead"]):
        return True
    elif not S[0]["IsReverse"] and S[0]["RefPos"] - S[0]["ReadPos"] > F[-1]["RefPos"] - (r["ftl"] - F[-1]["ReadPos"] - F[-1]["MatchRead"]):
        return True
    return False


# ---- SegmentGraph.cpp:203-264 (the PartAlignPos half of the loop does not reach the depths and is left out)
def _bamdiscordant_literal(chim):
    out = []
    for it in chim:
        if _is_end_discordant(it, True) or _is_end_discordant(it, False) or _is_single_anchored(it) or _is_pair_discordant(it):
            out.extend(it["F"])
            out.extend(it["S"])
        else:
            firstinserted = secondinserted = False
            for L, which in ((it["F"], "F"), (it["S"], "S")):
                previnserted = -1
                if len(L) > 0:
                    for i in range(len(L) - 1):
                        if abs(L[i]["RefPos"] - L[i + 1]["RefPos"]) > 750000:
                            if previnserted != i:
                                out.append(L[i])
                            out.append(L[i + 1])
                            previnserted = i + 1
                            if i + 1 == len(L) - 1:
                                if which == "F":
                                    firstinserted = True
                                else:
                                    secondinserted = True
            if len(it["F"]) > 0 and len(it["S"]) > 0:
                if abs(it["F"][-1]["RefPos"] - it["S"][-1]["RefPos"]) > 750000:
                    if not firstinserted:
                        out.append(it["F"][-1])
                        firstinserted = True
                    if not secondinserted:
                        out.append(it["S"][-1])
                        secondinserted = True
    out.sort(key=lambda b: (b["RefID"], b["RefPos"]))  # SingleBamRec_t::operator< (SingleBamRec.h:39-44); ties do not reach the sums below
    return out


# ---- SegmentGraph.cpp:297-337
def _equal(lhs, rhs):  # ReadRec.cpp:119-141; blocks as (RefID, RefPos, MatchRef)
    same1 = same2 = False
    if len(lhs[0]) == len(rhs[0]) and len(lhs[1]) == len(rhs[1]):
        same1 = True
        for i in range(len(lhs[0])):
            if lhs[0][i] != rhs[0][i]:
                same1 = False
        for i in range(len(lhs[1])):
            if lhs[1][i] != rhs[1][i]:
                same1 = False
    if len(lhs[0]) == len(rhs[1]) and len(lhs[1]) == len(rhs[0]):
        same2 = True
        for i in range(len(lhs[0])):
            if lhs[0][i] != rhs[1][i]:
                same2 = False
        for i in range(len(lhs[1])):
            if lhs[1][i] != rhs[0][i]:
                same2 = False
    return same1 or same2


def _pass1_literal(rec, min_mapq, n_break):
    refid, pos, mref, mpos, flag, mapq, aux, off = (rec[k].tolist() for k in ("refid", "pos", "mate_refid", "mate_pos", "flag", "mapq", "aux", "blk_off"))
    b_refpos, b_matchref, b_readpos = rec["b_refpos"].tolist(), rec["b_matchref"].tolist(), rec["b_readpos"].tolist()
    main, other = [], []
    last = ([], [])
    kept = 0
    for i in range(len(refid)):
        f = flag[i]
        # XAtag || IHtagvalue>1 (aux bit 0) || MapQuality<Min_MapQual || IsDuplicate || !IsMapped || RefID==-1 || binary_search(ChimName, Name) (aux bit 1)
        if aux[i] & 1 or mapq[i] < min_mapq or f & 0x400 or f & 0x4 or refid[i] == -1 or aux[i] & 2:
            continue
        blocks = [(refid[i], b_refpos[k], b_matchref[k], b_readpos[k]) for k in range(off[i], off[i + 1])]  # ReadRec_t(record): all blocks on the record's own mate side
        isfirst = bool(f & 0x40)
        tmp_sorted = [b[:3] for b in sorted(blocks, key=lambda b: b[3])]  # tmpreadrec.SortbyReadPos()
        first, second = (tmp_sorted, []) if isfirst else ([], tmp_sorted)
        matemapped = not (f & 0x8)
        if isfirst and matemapped and mref[i] != -1:
            second = second + [(mref[i], mpos[i], 15)]
        elif not isfirst and matemapped and mref[i] != -1:
            first = first + [(mref[i], mpos[i], 15)]
        tmp = (first, second)
        if _equal(last, tmp):
            continue
        last = tmp
        own = [b[:3] for b in blocks]  # readrec (unsorted: CIGAR order)
        if len(own) != 0:  # (FirstRead for a first mate, else SecondMate: the record's own list either way)
            main.append(own[0])
            other.extend(own[1:])
        kept += 1
        if kept == n_break:  # if(itdisstart==bamdiscordant.cend()) break;  -- the cluster automaton is not restated here
            break
    return main, other, kept


# ---- SegmentGraph.cpp:766-826
def _depth_literal(nodes, bamdiscordant, main, other):
    thresh = 3
    support, depth = [0] * len(nodes), [0.0] * len(nodes)
    itdis = 0
    for i, (chr_, p, ln) in enumerate(nodes):
        count = sumlen = 0
        while itdis != len(bamdiscordant) and bamdiscordant[itdis]["RefID"] == chr_ and bamdiscordant[itdis]["RefPos"] < p + ln:
            d = bamdiscordant[itdis]
            if d["RefPos"] >= p and d["RefPos"] + d["MatchRef"] <= p + ln:
                count += 1
                sumlen += d["MatchRef"]
            itdis += 1
        support[i] = count
        depth[i] = sumlen
    other = sorted(other, key=lambda r: (r[0], r[1]))  # (std::sort, unstable: ties can differ from libstdc++'s -- see the test)
    for reads, divide in ((main, False), (other, True)):
        if len(reads) == 0:
            continue
        it = 0
        for i, (chr_, p, ln) in enumerate(nodes):
            covcount = covsumlen = 0
            while it != len(reads):
                c, rp, ml = reads[it]
                if c == chr_ and rp >= p - thresh and rp + ml <= p + ln + thresh:
                    covcount += 1
                    covsumlen += ml
                elif rp >= p + ln or c != chr_:
                    break
                it += 1
            support[i] += covcount
            depth[i] += covsumlen
            if divide:
                depth[i] = 1.0 * depth[i] / ln
    return support, depth


@pytest.mark.parametrize("cfg", ["C1", "T2"])
def test_pass1_filter_duplicate_drop_and_node_depths_against_literal_loops(built, synth, tmp_path, cfg, monkeypatch):
    monkeypatch.setenv("SQUID_EXACT_DEPTH", "1")
    pre = synth(cfg)
    _, dump = ou.run_oracle(built, pre, tmp_path)  # (only for the merged chimeric fragments of its dump)
    chim = _read_chimrecord(dump / "chimrecord.txt")
    with squid_amd.Context() as ctx:
        ctx.load(f"{pre}.bam", f"{pre}.chim.bam")
        ctx.build_graph()
        rec = ctx.records()
        counts = ctx.counts()
        nodes = ctx.graph(1)["nodes"]
    main, other, kept = _pass1_literal(rec, 255, counts["n_break"])
    assert kept == counts["n_break"] <= counts["n_kept_p1"]
    # the whole stream without the break: the library's count of records that pass the filter and the duplicate drop
    _, _, kept_all = _pass1_literal(rec, 255, -1)
    assert kept_all == counts["n_kept_p1"]
    dis = _bamdiscordant_literal(chim)
    support, depth = _depth_literal([n[:3] for n in nodes], dis, main, other)
    assert [n[3] for n in nodes] == support
    assert [n[4] for n in nodes] == depth  # the same IEEE doubles: integer sums, one division
