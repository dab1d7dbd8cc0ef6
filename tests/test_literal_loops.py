"""Literal restatements of reference loops, written straight from the reference text (src/SegmentGraph.cpp, src/ReadRec.cpp) and
NOT from oracle/: statement-by-statement Python over the records the library holds in HBM (sq_debug_download).  They are slow
and small on purpose.  What they give that the oracle comparison cannot: oracle and product are two restatements by one author;
a shared misreading is invisible to a test that compares the two.  These loops are a third reading, kept as close to the
reference's statements as Python allows, of

* the pass-1 record filter, the mate stub and the consecutive-duplicate drop (`SegmentGraph.cpp:297-337`, `ReadRec_t::Equal`
  `ReadRec.cpp:119-141`) -> ReadsMain / ReadsOther;
* the discordant block list `bamdiscordant` (`:203-264`; `IsEndDiscordant / IsSingleAnchored / IsPairDiscordant`,
  `ReadRec.cpp:171-228`) from the merged chimeric fragments;
* per-node Support / AvgDepth (`:766-826`), three sweeps with their lagging cursors and the final division;
* the whole of stage 2: `LocateRead` with its running cursor and its trimming of the blocks (`:1207-1293`), `RawEdgesChim`
  (`:1394-1555`), `RawEdgesOther` over every concordant record (`:1557-1690`), `Edge_t`'s constructor and order (`BPEdge.h:31-74`),
  `IsDiscordant` (`:181-190`) and the sort/merge of `BuildEdges` (`:1932-1959`) -> the edge list with its weights.

* `BuildChimericSBamRecord` itself (`ReadRec.cpp:329-413`): the sort by name, the merge of a name's records, `SortbyReadPos`, the read
  length, the sort under `FrontSmallerThan` (`:90-117`, not a strict weak order: the permutation is whatever introsort makes of it) and the
  PCR-duplicate pass -> the fragment list every loop above starts from, against the oracle's dump of it.

* `ExactBreakpoint` and `CountTop` (`SegmentGraph.cpp:3019-3081`, `:51-104`): the breakpoint pairs `_sv.txt` prints, per edge of the final
  graph, over the fragments as the literal `RawEdgesChim` has trimmed them.

* `FilterbyWeight` (`:1968-2124`, its slips included) and `FilterEdges` with `GroupConnection` / `GroupSelect` (`:2394-2527`): group
  weights and which edges survive, each from the stage in front of it; `CompressNode` (`:2528-2604`) the same way; and
  `FurtherCompressNode` + `ConnectedComponent` + `MultiplyDisEdges` (`:2693-2892`, `:2911-3003`, `:3005-3010`) -> the final graph;
  `FilterbyInterleaving` (`:2161-2277`) -> KeepEdge (all "keep" on the generator's files: the walks and the long-group rule, not the overlap rule).

* `WriteBEDPE` (`WriteIO.cpp:45-124`) with `Node_NewChr` and `DeMultiplyDisEdges`: the text of `_sv.txt`, byte for byte, from the final graph,
  the component orders and the breakpoints.

The only inputs taken from elsewhere: the node coordinates of stage 1 and the number of kept records the stream loop consumes before its
`break` (`:338-339`); the loops that need the merged chimeric fragments read the oracle's dump of them, which the last item checks.
"""
import gzip
import struct

import numpy as np
import pytest

import oracle_util as ou


# ---- a BAM record the way BamTools hands it to ReadRec_t::ReadRec_t (ReadRec.cpp:10-88), from the SAM/BAM specification: the fields the
# loops below read, and the aligned blocks of the constructor (CIGAR walk :45-87, the poly-A/T test :62-72, strand mirroring :74-75)
def _records_from_bam(path, chim_names):
    data = gzip.open(path, "rb").read()
    l_text = struct.unpack_from("<i", data, 4)[0]
    at = 8 + l_text
    n_ref = struct.unpack_from("<i", data, at)[0]
    at += 4
    for _ in range(n_ref):
        at += 8 + struct.unpack_from("<i", data, at)[0]
    rec = {k: [] for k in ("refid", "pos", "mate_refid", "mate_pos", "end_pos", "flag", "mapq", "aux", "totlen", "blk_off", "b_refpos", "b_matchref", "b_readpos", "b_matchread")}
    rec["blk_off"].append(0)
    while at < len(data):
        bs, refid, pos, lname, mapq, _bin, ncig, flag, lseq, mref, mpos, _tlen = struct.unpack_from("<iiiBBHHHiiii", data, at)
        p = at + 36
        name = data[p:p + lname - 1].decode()
        p += lname
        cig = []
        for k in range(ncig):
            v = struct.unpack_from("<I", data, p + 4 * k)[0]
            cig.append(("MIDNSHP=X"[v & 15], v >> 4))
        p += 4 * ncig
        seq = data[p:p + (lseq + 1) // 2]
        p += (lseq + 1) // 2
        qual = data[p:p + lseq]
        p += lseq
        tags = data[p:at + 4 + bs]
        totlen = sum(ln for t, ln in cig if t in "MSHI=X")
        rev = bool(flag & 0x10)
        readpos, refpos, hard, i = 0, pos, 0, 0
        while i < len(cig):
            t, ln = cig[i]
            if t in "SH":
                readpos += ln
                if t == "H":
                    hard += ln
            elif t in "M=":
                tr = tf = 0
                j = i
                while j < len(cig) and cig[j][0] not in "SHN":
                    if cig[j][0] != "D":
                        tr += cig[j][1]
                    if cig[j][0] != "I":
                        tf += cig[j][1]
                    j += 1
                na = nt = 0
                for q in range(readpos - hard, readpos + tr - hard):
                    code = (seq[q >> 1] >> (4 if q % 2 == 0 else 0)) & 15  # "=ACMGRSVTWYHKDBN": A = 1, T = 8
                    na += code == 1
                    nt += code == 8
                if 1.0 * na / tr < 0.75 and 1.0 * nt / tr < 0.75:
                    rec["b_refpos"].append(refpos)
                    rec["b_matchref"].append(tf)
                    rec["b_readpos"].append(totlen - readpos - tr if rev else readpos)
                    rec["b_matchread"].append(tr)
                readpos += tr
                refpos += tf
                i = j - 1
            elif t == "N":
                refpos += ln
            i += 1
        # aux bits of the library's record layout: 1 = HasTag("XA") || IH > 1, 2 = raw name in ChimName, 4 = low-Phred run > Max_LowPhred_Len (10) below 33 + Min_Phred (4)
        aux, run, longest = 0, 0, 0
        for qv in qual:
            run = run + 1 if qv + 33 < 33 + 4 else 0
            longest = max(longest, run)
        if longest > 10:
            aux |= 4
        tp, has_xa, ih = 0, False, 0
        while tp + 3 <= len(tags):
            tag, ty = tags[tp:tp + 2], chr(tags[tp + 2])
            size = {"A": 1, "c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}.get(ty)
            if tag == b"XA":
                has_xa = True
            if tag == b"IH" and ty in "cCsSi":
                ih = int.from_bytes(tags[tp + 3:tp + 3 + size], "little")
            if size is not None:
                tp += 3 + size
            elif ty in "ZH":
                tp = tags.index(0, tp + 3) + 1
            else:  # B array
                sub = {"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}[chr(tags[tp + 3])]
                tp += 8 + sub * struct.unpack_from("<i", tags, tp + 4)[0]
        if has_xa or ih > 1:
            aux |= 1
        if name in chim_names:
            aux |= 2
        end = pos + sum(ln for t, ln in cig if t in "MDN=X")
        for k, v in (("refid", refid), ("pos", pos), ("mate_refid", mref), ("mate_pos", mpos), ("end_pos", end), ("flag", flag), ("mapq", mapq), ("aux", aux), ("totlen", totlen)):
            rec[k].append(v)
        rec["blk_off"].append(len(rec["b_refpos"]))
        at += 4 + bs
    return {k: np.array(v, dtype=np.int64) for k, v in rec.items()}


# ---- the chimeric BAM the way BuildChimericSBamRecord reads it (ReadRec.cpp:329-413): every mapped, non-duplicate record through the
# constructor (ReadRec.cpp:10-88, restated once more here with the name and the per-block strand kept), std::sort by Qname, the merge of
# equal names, SortbyReadPos, the read length as the median of the first five records, std::sort under FrontSmallerThan (which is not a
# strict weak order: what comes out is what libstdc++'s introsort does with it), and the PCR-duplicate pass
def _chim_readrecs(path):
    data = gzip.open(path, "rb").read()
    l_text = struct.unpack_from("<i", data, 4)[0]
    at = 8 + l_text
    n_ref = struct.unpack_from("<i", data, at)[0]
    at += 4
    for _ in range(n_ref):
        at += 8 + struct.unpack_from("<i", data, at)[0]
    out = []
    while at < len(data):
        bs, refid, pos, lname, mapq, _bin, ncig, flag, lseq, _mref, _mpos, _tlen = struct.unpack_from("<iiiBBHHHiiii", data, at)
        p = at + 36
        name = data[p:p + lname - 1].decode()
        p += lname
        cig = []
        for k in range(ncig):
            v = struct.unpack_from("<I", data, p + 4 * k)[0]
            cig.append(("MIDNSHP=X"[v & 15], v >> 4))
        p += 4 * ncig
        seq = data[p:p + (lseq + 1) // 2]
        p += (lseq + 1) // 2
        qual = data[p:p + lseq]
        at += 4 + bs
        if flag & 0x4 or flag & 0x400:  # record.IsMapped() && !record.IsDuplicate()
            continue
        if len(name) >= 2 and name[-2:] in ("/1", "/2"):
            name = name[:-2]
        totlen = sum(ln for t, ln in cig if t in "MSHI=X")
        run = longest = 0
        for qv in qual:
            run = run + 1 if qv + 33 < 33 + 4 else 0
            longest = max(longest, run)
        low = longest > 10
        rev, first = bool(flag & 0x10), bool(flag & 0x40)
        blocks = []
        readpos, refpos, hard, i = 0, pos, 0, 0
        while i < len(cig):
            t, ln = cig[i]
            if t in "SH":
                readpos += ln
                if t == "H":
                    hard += ln
            elif t in "M=":
                tr = tf = 0
                j = i
                while j < len(cig) and cig[j][0] not in "SHN":
                    if cig[j][0] != "D":
                        tr += cig[j][1]
                    if cig[j][0] != "I":
                        tf += cig[j][1]
                    j += 1
                na = nt = 0
                for q in range(readpos - hard, readpos + tr - hard):
                    code = (seq[q >> 1] >> (4 if q % 2 == 0 else 0)) & 15
                    na += code == 1
                    nt += code == 8
                if 1.0 * na / tr < 0.75 and 1.0 * nt / tr < 0.75:
                    blocks.append({"RefID": refid, "RefPos": refpos, "ReadPos": totlen - readpos - tr if rev else readpos, "MatchRef": tf, "MatchRead": tr, "IsReverse": rev})
                readpos += tr
                refpos += tf
                i = j - 1
            elif t == "N":
                refpos += ln
            i += 1
        out.append({"Qname": name, "F": blocks if first else [], "S": [] if first else blocks, "ftl": totlen if first else 0, "stl": 0 if first else totlen,
                    "flow": low if first else None, "slow": None if first else low})  # (None: the constructor leaves the other side's flag unset)
    return out


def _build_chimeric_literal(path):
    recs = _chim_readrecs(path)
    sample = [max(r["ftl"], r["stl"]) for r in recs[:5]]
    _std_sort(recs, lambda a, b: a["Qname"] < b["Qname"])  # sort(SBamrecord.begin(), SBamrecord.end()): operator< compares the names
    merged = []
    for it in recs:
        if len(merged) == 0 or it["Qname"] != merged[-1]["Qname"]:
            merged.append({k: (list(v) if isinstance(v, list) else v) for k, v in it.items()})
        else:
            back = merged[-1]
            if back["ftl"] == 0 and it["ftl"] != 0:
                back["ftl"], back["flow"] = it["ftl"], it["flow"]
            if back["stl"] == 0 and it["stl"] != 0:
                back["stl"], back["slow"] = it["stl"], it["slow"]
            back["F"].extend(it["F"])
            back["S"].extend(it["S"])
    for r in merged:
        _std_sort(r["F"], lambda a, b: a["ReadPos"] < b["ReadPos"])
        _std_sort(r["S"], lambda a, b: a["ReadPos"] < b["ReadPos"])
    sample.sort()
    read_len = sample[len(sample) // 2]

    def front_smaller(l, r):  # ReadRec.cpp:90-117
        for x, y in (("F", "F"), ("S", "S"), ("F", "S"), ("S", "F")):
            if len(l[x]) != 0 and len(r[y]) != 0:
                a, b = l[x][0], r[y][0]
                return a["RefID"] < b["RefID"] if a["RefID"] != b["RefID"] else a["RefPos"] < b["RefPos"]
        return False

    _std_sort(merged, front_smaller)
    key = lambda r: ([(b["RefID"], b["RefPos"], b["MatchRef"]) for b in r["F"]], [(b["RefID"], b["RefPos"], b["MatchRef"]) for b in r["S"]])
    kept = []
    for it in merged:
        if len(kept) == 0:
            kept.append(it)
        elif len(it["F"]) == 0 or len(kept[-1]["F"]) == 0:
            kept.append(it)
        elif it["F"][0]["RefID"] != kept[-1]["F"][0]["RefID"] or it["F"][0]["RefPos"] != kept[-1]["F"][0]["RefPos"]:
            kept.append(it)
        else:
            isdup = False
            for it2 in reversed(kept):
                if len(it2["F"]) == 0 or it["F"][0]["RefID"] != it2["F"][0]["RefID"] or it["F"][0]["RefPos"] != it2["F"][0]["RefPos"]:
                    break
                if _equal(key(it), key(it2)):
                    isdup = True
                    break
            if not isdup:
                kept.append(it)
    return kept, read_len


def _chim_names(dump):
    names = {""}  # (the sized-then-appended vector of :196-201 also holds "", ledger B9)
    for line in open(dump / "chimrecord.txt"):
        if not line.startswith("#"):
            names.add(line.split("\t")[0])
    return names


def _read_chimrecord(path):
    frags = []
    for line in open(path):
        if line.startswith("#"):
            continue
        f = line.rstrip("\n").split("\t")
        fr = {"Qname": f[0], "ftl": int(f[1]), "stl": int(f[2]), "flow": int(f[3]), "slow": int(f[4]), "F": [], "S": []}
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


# ---- libstdc++'s std::sort (bits/stl_algo.h: __introsort_loop, __unguarded_partition_pivot, __move_median_to_first, __final_insertion_sort),
# statement by statement: the reference sorts ReadsOther with it (:781) under a comparator that only looks at (chr, pos), the sort is not
# stable, and a block of <= 3 bases right behind a node boundary is counted for the node in front or the node behind depending on
# where it ends up among its ties.  (The heap-sort fallback of a recursion that has gone 2*log2(n) deep is not restated: it raises.)
def _std_sort(a, less):
    def median_to_first(result, x, y, z):
        if less(a[x], a[y]):
            pick = y if less(a[y], a[z]) else (z if less(a[x], a[z]) else x)
        else:
            pick = x if less(a[x], a[z]) else (z if less(a[y], a[z]) else y)
        a[result], a[pick] = a[pick], a[result]

    def partition(first, last, pivot):
        while True:
            while less(a[first], a[pivot]):
                first += 1
            last -= 1
            while less(a[pivot], a[last]):
                last -= 1
            if not first < last:
                return first
            a[first], a[last] = a[last], a[first]
            first += 1

    def introsort_loop(first, last, depth):
        while last - first > 16:
            if depth == 0:
                raise NotImplementedError("heap-sort fallback of std::sort")
            depth -= 1
            median_to_first(first, first + 1, first + (last - first) // 2, last - 1)
            cut = partition(first + 1, last, first)
            introsort_loop(cut, last, depth)
            last = cut

    def linear_insert(last, guarded_first=None):
        val = a[last]
        nxt = last - 1
        while less(val, a[nxt]):
            a[last] = a[nxt]
            last = nxt
            nxt -= 1
        a[last] = val

    def insertion_sort(first, last):
        for i in range(first + 1, last):
            if less(a[i], a[first]):
                val = a[i]
                a[first + 1:i + 1] = a[first:i]
                a[first] = val
            else:
                linear_insert(i)

    n = len(a)
    if n == 0:
        return a
    introsort_loop(0, n, 2 * (n.bit_length() - 1))
    if n > 16:
        insertion_sort(0, 16)
        for i in range(16, n):
            linear_insert(i)
    else:
        insertion_sort(0, n)
    return a


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
    other = _std_sort(list(other), lambda x, y: x[0] < y[0] if x[0] != y[0] else x[1] < y[1])  # the lambda of :781, under std::sort
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


# ---- SegmentGraph.cpp:1207-1293: both halves of the function are the same statements over FirstRead and then SecondMate with one
# running cursor `i`; the blocks are trimmed to the node they land in (mutating the caller's record, as the reference does)
def _locate_read(nodes, initialguess, F, S):
    n = len(nodes)
    out = [0] * (len(F) + len(S))
    i, thresh = initialguess, 5

    def inside(i, b):
        chr_, p, ln = nodes[i]
        return chr_ == b["RefID"] and b["RefPos"] >= p - thresh and b["RefPos"] + b["MatchRef"] <= p + ln + thresh

    for base, L in ((0, F), (len(F), S)):
        for k, b in enumerate(L):
            if i < 0 or i >= n:
                i = initialguess
            if not inside(i, b):
                if nodes[i][0] < b["RefID"] or (nodes[i][0] == b["RefID"] and nodes[i][1] <= b["RefPos"]):
                    while i < n and nodes[i][0] <= b["RefID"]:
                        if inside(i, b):
                            break
                        i += 1
                else:
                    while i > -1 and nodes[i][0] >= b["RefID"]:
                        if inside(i, b):
                            break
                        i -= 1
            if i < 0 or i >= n or nodes[i][0] != b["RefID"]:
                out[base + k] = -1
            else:
                out[base + k] = i
                _, p, ln = nodes[i]
                if b["RefPos"] < p:
                    d = p - b["RefPos"]
                    if not b["IsReverse"]:
                        b["ReadPos"] += d
                    b["MatchRef"] -= d
                    b["MatchRead"] -= d
                    b["RefPos"] = p
                if b["RefPos"] + b["MatchRef"] > p + ln:
                    d = b["RefPos"] + b["MatchRef"] - p - ln
                    if b["IsReverse"]:
                        b["ReadPos"] += d
                    b["MatchRef"] -= d
                    b["MatchRead"] -= d
    return out


class _Edges:
    """vEdges plus Edge_t's constructor (BPEdge.h:31-52: the smaller index first, heads swapped with it) and IsDiscordant (:181-190)"""

    def __init__(self, nodes):
        self.nodes, self.v = nodes, []

    @staticmethod
    def make(i1, h1, i2, h2):
        return (i2, i1, h2, h1) if i1 > i2 else (i1, i2, h1, h2)  # (Ind1, Ind2, Head1, Head2): the member order of operator<

    def is_discordant(self, e):
        i1, i2, h1, h2 = e
        if self.nodes[i1][0] != self.nodes[i2][0]:
            return True
        elif self.nodes[i2][1] - self.nodes[i1][1] - self.nodes[i1][2] > 50000 and i2 - i1 > 20:  # Concord_Dist_Pos, Concord_Dist_Idx (Config.cpp:24-25)
            return True
        elif h1 is not False or h2 is not True:
            return True
        return False

    def push(self, e, w=1):
        self.v.append((e, w))

    def unlocated(self, firstfrontindex, b):  # the two cursor walks of :1408-1411 / :1614-1616 for a block LocateRead could not place
        nodes, i = self.nodes, firstfrontindex
        while i < len(nodes) and (nodes[i][0] < b["RefID"] or (nodes[i][0] == b["RefID"] and nodes[i][1] + nodes[i][2] < b["RefPos"])):
            i += 1
        while i > -1 and (nodes[min(i, len(nodes) - 1)][0] > b["RefID"] or (nodes[min(i, len(nodes) - 1)][0] == b["RefID"] and nodes[min(i, len(nodes) - 1)][1] > b["RefPos"])):
            assert i < len(nodes)  # (the reference would read vNodes[size()] here)
            i -= 1
        self.push(self.make(i, False, i + 1, True))


def _pair_overlap(r, rn, i, j):  # :1487-1506 == :1655-1674
    nf = len(r["F"])
    isoverlap = False
    for k in range(nf):
        if j == rn[k]:
            isoverlap = True
    for k in range(len(r["S"])):
        if i == rn[nf + k]:
            isoverlap = True
    if nf > 1:
        if _is_end_discordant(r, True) and ((rn[0] <= j and rn[nf - 1] >= j) or (rn[0] >= j and rn[nf - 1] <= j)):
            isoverlap = True
        elif not _is_end_discordant(r, True) and abs(i - j) < 3:
            isoverlap = True
    if len(r["S"]) > 1:
        if _is_end_discordant(r, False) and ((rn[nf] <= i and rn[-1] >= i) or (rn[nf] >= i and rn[-1] <= i)):
            isoverlap = True
        elif not _is_end_discordant(r, False) and abs(i - j) < 3:
            isoverlap = True
    return isoverlap


def _is_pair_discordant_nocheck(r):  # IsPairDiscordant(false), ReadRec.cpp:209-228 without the needcheck part
    F, S = r["F"], r["S"]
    if len(F) == 0 or len(S) == 0:
        return False
    if F[0]["RefID"] != S[-1]["RefID"] or F[0]["IsReverse"] == S[-1]["IsReverse"]:
        return True
    elif not F[0]["IsReverse"] and F[0]["RefPos"] - F[0]["ReadPos"] > S[-1]["RefPos"] - (r["stl"] - S[-1]["ReadPos"] - S[-1]["MatchRead"]):
        return True
    elif not S[0]["IsReverse"] and S[0]["RefPos"] - S[0]["ReadPos"] > F[-1]["RefPos"] - (r["ftl"] - F[-1]["ReadPos"] - F[-1]["MatchRead"]):
        return True
    return False


# ---- SegmentGraph.cpp:1394-1555.  The breakpoint pairs of a discordant edge are only counted here (:1529-1554 keeps every one of them:
# the group filter is commented out at :1547), so PairBreakpoints is a counter
def _raw_edges_chim(E, chim):
    firstfrontindex = 0
    pair_bp = {}
    for it in chim:
        F, S = it["F"], it["S"]
        if len(F) == 0 and len(S) == 0:
            continue
        rn = _locate_read(E.nodes, firstfrontindex, F, S)
        if rn[0] != -1:
            firstfrontindex = rn[0]
        for k in range(len(rn)):
            if rn[k] == -1:
                E.unlocated(firstfrontindex, F[k] if k < len(F) else S[k - len(F)])
        for base, L in ((0, F), (len(F), S)):
            for k in range(len(L) - 1):
                i, j = rn[base + k], rn[base + k + 1]
                if i != j and i != -1 and j != -1:
                    tmp = E.make(i, bool(L[k]["IsReverse"]), j, not L[k + 1]["IsReverse"])
                    if not E.is_discordant(tmp):
                        E.push(tmp)
                    else:
                        pair_bp[tmp] = pair_bp.get(tmp, 0) + 1
        if len(F) > 0 and len(S) > 0:
            if not _is_single_anchored(it) and not _is_end_discordant(it, True) and not _is_end_discordant(it, False):
                i, j = rn[len(F) - 1], rn[-1]
                if i != j and i != -1 and j != -1 and not _pair_overlap(it, rn, i, j):
                    tmp = E.make(i, bool(F[-1]["IsReverse"]), j, bool(S[-1]["IsReverse"]))
                    if not E.is_discordant(tmp):
                        E.push(tmp)
                    elif _is_pair_discordant_nocheck(it):
                        pair_bp[tmp] = pair_bp.get(tmp, 0) + 1
    for tmp in sorted(pair_bp):  # map<Edge_t, ...> iteration order; (False < True as (int)Head)
        E.push(tmp, pair_bp[tmp])


# ---- SegmentGraph.cpp:1557-1690 over the whole concordant stream
def _raw_edges_other(E, rec, min_mapq):
    refid, pos, mref, mpos, flag, mapq, aux, off, totlen = (rec[k].tolist() for k in ("refid", "pos", "mate_refid", "mate_pos", "flag", "mapq", "aux", "blk_off", "totlen"))
    b_refpos, b_matchref, b_readpos, b_matchread = (rec[k].tolist() for k in ("b_refpos", "b_matchref", "b_readpos", "b_matchread"))
    firstfrontindex = 0
    last = ([], [])
    for r in range(len(refid)):
        f = flag[r]
        if aux[r] & 1 or f & 0x400 or mapq[r] < min_mapq or f & 0x4 or aux[r] & 2:
            continue
        rev = bool(f & 0x10)
        own = [{"RefID": refid[r], "RefPos": b_refpos[k], "ReadPos": b_readpos[k], "MatchRef": b_matchref[k], "MatchRead": b_matchread[k], "IsReverse": rev} for k in range(off[r], off[r + 1])]
        own.sort(key=lambda b: b["ReadPos"])  # readrec.SortbyReadPos() (a handful of blocks: libstdc++ sorts <= 16 elements by insertion, stable)
        isfirst = bool(f & 0x40)
        rr = {"F": own if isfirst else [], "S": [] if isfirst else own, "ftl": totlen[r] if isfirst else 0, "stl": 0 if isfirst else totlen[r]}
        low_own = bool(aux[r] & 4)
        if not (f & 0x8) and mref[r] != -1:
            stub = {"RefID": mref[r], "RefPos": mpos[r], "ReadPos": 0, "MatchRef": 15, "MatchRead": 15, "IsReverse": bool(f & 0x20)}
            (rr["S"] if isfirst else rr["F"]).append(stub)
        key = ([(b["RefID"], b["RefPos"], b["MatchRef"]) for b in rr["F"]], [(b["RefID"], b["RefPos"], b["MatchRef"]) for b in rr["S"]])
        if _equal(last, key):
            continue
        last = key
        F, S = rr["F"], rr["S"]
        whether = False
        if len(F) == 0 or len(S) == 0:
            whether = True
        else:
            # (the LowPhred flag of the side the record is not on was never set by the constructor; that side is the mate stub with
            #  ReadPos 0 whenever this branch is reached, so `||` never gets to it)
            first_ok = F[0]["ReadPos"] <= 15 or (low_own if isfirst else _never())
            second_ok = first_ok and (S[0]["ReadPos"] <= 15 or (low_own if not isfirst else _never()))
            whether = first_ok and second_ok
        if not whether:
            continue
        rn = _locate_read(E.nodes, firstfrontindex, F, S)
        if len(rn) != 0 and rn[0] != -1:
            firstfrontindex = rn[0]
        for k in range(len(rn)):
            if rn[k] == -1:
                E.unlocated(firstfrontindex, F[k] if k < len(F) else S[k - len(F)])
        for base, L in ((0, F), (len(F), S)):
            for k in range(len(L) - 1):
                i, j = rn[base + k], rn[base + k + 1]
                if i != j and i != -1 and j != -1:
                    E.push(E.make(i, bool(L[k]["IsReverse"]), j, not L[k + 1]["IsReverse"]))
        if isfirst and len(F) > 0 and len(S) > 0:
            if not _is_single_anchored(rr) and not _is_end_discordant(rr, True) and not _is_end_discordant(rr, False):
                i, j = rn[len(F) - 1], rn[-1]
                if i != j and i != -1 and j != -1 and not _pair_overlap(rr, rn, i, j):
                    tmp = E.make(i, bool(F[-1]["IsReverse"]), j, bool(S[-1]["IsReverse"]))
                    if _is_pair_discordant_nocheck(rr) == E.is_discordant(tmp):
                        E.push(tmp)


def _never():
    raise AssertionError("the reference would read an uninitialised LowPhred flag here")


# ---- SegmentGraph.cpp:1932-1959: sort, merge equal edges by adding their weights, keep Weight > 0
def _build_edges_literal(nodes, chim, rec, min_mapq):
    E = _Edges([n[:3] for n in nodes])
    _raw_edges_chim(E, chim)
    _raw_edges_other(E, rec, min_mapq)
    E.v.sort(key=lambda ew: ew[0])  # operator< reads (Ind1, Ind2, Head1, Head2) only; the weights of equal edges are summed, so tie order is immaterial
    out = []
    for e, w in E.v:
        if len(out) == 0 or out[-1][0] != e:
            out.append([e, w])
        else:
            out[-1][1] += w
    return [(e[0], int(e[2]), e[1], int(e[3]), w) for e, w in out if w > 0]


def _check(rec, chim, nodes, n_break, kept_total, edges=None):
    main, other, kept = _pass1_literal(rec, 255, n_break)
    assert kept == n_break
    if kept_total is not None:  # the whole stream without the break: records that pass the filter and the duplicate drop
        _, _, kept_all = _pass1_literal(rec, 255, -1)
        assert kept_all == kept_total >= n_break
    support, depth = _depth_literal([n[:3] for n in nodes], _bamdiscordant_literal(chim), main, other)
    assert [n[3] for n in nodes] == support
    assert [n[4] for n in nodes] == depth  # the same IEEE doubles: integer sums, one division
    if edges is not None:
        import copy

        assert _build_edges_literal(nodes, copy.deepcopy(chim), rec, 255) == [tuple(e[:5]) for e in edges]


@pytest.mark.parametrize("cfg", ["C1", "T2"])
def test_oracle_against_the_literal_loops(built, synth, tmp_path, cfg):
    """CPU: the oracle's BuildNode_STAR bookkeeping (kept records, the record the loop breaks at) and its per-node Support / AvgDepth
    against the literal loops over records decoded by the literal constructor above -- a reading of the reference that shares no
    code with oracle/"""
    pre = synth(cfg)
    _, dump = ou.run_oracle(built, pre, tmp_path)
    stats = dict(line.split("\t") for line in (dump / "order_stats.txt").read_text().splitlines())
    rec = _records_from_bam(f"{pre}.bam", _chim_names(dump))
    _check(rec, _read_chimrecord(dump / "chimrecord.txt"), ou.read_nodes(dump / "nodes_build.txt"), int(stats["break_record"]), None, ou.read_edges(dump / "edges_build.txt"))
    assert stats["kept_records"] == stats["break_record"]  # (the oracle's loop counts up to its break)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["C1", "T2"])
def test_hip_path_against_the_literal_loops(built, synth, tmp_path, cfg, monkeypatch):
    """GPU: the records K0 leaves in HBM field by field against the literal constructor, then the library's pass-1 counts and stage-1
    Support / AvgDepth against the literal loops run over ITS records"""
    import squid_amd

    monkeypatch.setenv("SQUID_EXACT_DEPTH", "1")
    pre = synth(cfg)
    _, dump = ou.run_oracle(built, pre, tmp_path)  # (only for the merged chimeric fragments of its dump)
    with squid_amd.Context() as ctx:
        ctx.load(f"{pre}.bam", f"{pre}.chim.bam")
        ctx.build_graph()
        rec = ctx.records()
        counts = ctx.counts()
        nodes, edges = ctx.graph(1)["nodes"], ctx.graph(2)["edges"]
    want = _records_from_bam(f"{pre}.bam", _chim_names(dump))
    for k in want:
        assert np.array_equal(np.asarray(rec[k]).astype(np.int64), want[k]), k
    _check(rec, _read_chimrecord(dump / "chimrecord.txt"), nodes, counts["n_break"], counts["n_kept_p1"], edges)


@pytest.mark.parametrize("cfg", ["C1", "T2", "C2"])
def test_oracle_chimeric_fragments_against_the_literal_build(built, synth, tmp_path, cfg):
    """CPU: the merged, sorted, de-duplicated chimeric fragments the oracle dumps (what every other literal loop takes as given) against
    the literal BuildChimericSBamRecord over the chimeric BAM decoded by the literal constructor -- names, lengths, low-Phred flags of the
    sides that have blocks, every block, the order of the fragments, and the read length"""
    pre = synth(cfg)
    _, dump = ou.run_oracle(built, pre, tmp_path)
    want = _read_chimrecord(dump / "chimrecord.txt")
    want_len = int(open(dump / "chimrecord.txt").readline().split("=")[1])
    got, read_len = _build_chimeric_literal(f"{pre}.chim.bam")
    assert read_len == want_len
    assert len(got) == len(want) > 0
    for g, w in zip(got, want):
        assert g["Qname"] == w["Qname"]
        assert (g["ftl"], g["stl"]) == (w["ftl"], w["stl"])
        assert int(bool(g["flow"])) == w["flow"] or len(g["F"]) == 0  # (the dump prints 0 for a side without blocks)
        assert int(bool(g["slow"])) == w["slow"] or len(g["S"]) == 0
        assert g["F"] == w["F"] and g["S"] == w["S"]


def test_oracle_chimeric_fragments_with_pcr_duplicates_against_the_literal_build(built, synth, tmp_path):
    """the same with every seventh chimeric fragment copied under a new name (exact copies: the PCR-duplicate pass, ReadRec.cpp:387-409,
    drops one of each pair) and a few copies shifted by one base (kept): the branch the generator's own files never reach"""
    import os
    import struct

    import bamwriter as bw
    from test_gpu_parity import _read_bam_records, _rename_record

    pre = synth("T2")
    contigs, _ = _read_bam_records(f"{pre}.bam")
    _, chim = _read_bam_records(f"{pre}.chim.bam")
    by_name = {}
    for refid, pos, raw in chim:
        by_name.setdefault(raw[36:36 + raw[12] - 1].decode(), []).append(raw)
    picked = sorted(by_name)[::7][:40]
    out = [raw for _, _, raw in chim]
    for k, nm in enumerate(picked):
        out += [_rename_record(raw, f"pcrdup{k}") for raw in by_name[nm]]
    for k, nm in enumerate(picked[:10]):  # same front position only when the first block does not move: shift every record by one base instead -> no duplicate
        for raw in by_name[nm]:
            moved = bytearray(_rename_record(raw, f"shifted{k}"))
            struct.pack_into("<i", moved, 8, struct.unpack_from("<i", moved, 8)[0] + 1)
            out.append(bytes(moved))
    new = tmp_path / "dups"
    os.symlink(f"{pre}.bam", f"{new}.bam")
    if os.path.exists(f"{pre}.bam.bai"):
        os.symlink(f"{pre}.bam.bai", f"{new}.bam.bai")
    bw.write_bam(f"{new}.chim.bam", contigs, out, sort_order="unsorted")
    _, dump = ou.run_oracle(built, new, tmp_path)
    want = _read_chimrecord(dump / "chimrecord.txt")
    got, read_len = _build_chimeric_literal(f"{new}.chim.bam")
    n_names = len(by_name) + len(picked) + 10
    assert len(want) < n_names  # (fragments were dropped)
    assert [(g["Qname"], g["ftl"], g["stl"], g["F"], g["S"]) for g in got] == [(w["Qname"], w["ftl"], w["stl"], w["F"], w["S"]) for w in want]
    assert read_len == int(open(dump / "chimrecord.txt").readline().split("=")[1])


# ---- SegmentGraph.cpp:51-104 (CountTop) and :3019-3081 (ExactBreakpoint), over the fragments as RawEdgesChim has left them (LocateRead trims
# the blocks to the nodes of the build stage; ExactBreakpoint locates -- and trims -- them again in the final node table)
def _count_top(e, x):
    x = sorted(x)
    y = sorted(set(x))
    count = [0.0] * len(y)
    for i, yi in enumerate(y):
        for xj in x:
            if yi == xj:
                count[i] += 1
            elif abs(yi[0] - xj[0]) + abs(yi[1] - xj[1]) < 10:
                count[i] += 0.5
    out = []
    while len(out) < 5:
        it = max(range(len(count)), key=lambda i: (count[i], -i))  # max_element: the first of the largest
        if count[it] > 3:
            if all(abs(o[0] - y[it][0]) + abs(o[1] - y[it][1]) >= 50 for o in out):
                out.append(y[it])
        else:
            break
        count[it] = 0
    if len(out) == 0:
        lo1, hi1 = min(p[0] for p in y), max(max(p[0] for p in y), 0)
        lo2, hi2 = min(p[1] for p in y), max(max(p[1] for p in y), 0)
        _, _, head1, head2 = e
        out.append((lo1 if head1 else hi1, lo2 if head2 else hi2))
    return out


def _exact_breakpoints_literal(nodes, chim):
    E = _Edges([n[:3] for n in nodes])
    bp = {}
    firstfrontindex = 0
    for it in chim:
        F, S = it["F"], it["S"]
        if len(F) <= 1 and len(S) <= 1:
            continue
        rn = _locate_read(E.nodes, firstfrontindex, F, S)
        if rn[0] != -1:
            firstfrontindex = rn[0]
        for base, L in ((0, F), (len(F), S)):
            if len(L) > 1:
                for k in range(len(L) - 1):
                    i, j = rn[base + k], rn[base + k + 1]
                    if i != j and i != -1 and j != -1:
                        tmp = E.make(i, bool(L[k]["IsReverse"]), j, not L[k + 1]["IsReverse"])
                        if E.is_discordant(tmp):
                            a, b = L[k], L[k + 1]
                            b1 = a["RefPos"] if a["IsReverse"] else a["RefPos"] + a["MatchRef"]
                            b2 = b["RefPos"] + b["MatchRef"] if b["IsReverse"] else b["RefPos"]
                            if (a["RefID"], a["RefPos"]) > (b["RefID"], b["RefPos"]):  # SingleBamRec_t::operator>
                                b1, b2 = b2, b1
                            bp.setdefault(tmp, []).append((b1, b2))
    return {e: _count_top(e, x) for e, x in bp.items()}


@pytest.mark.parametrize("cfg", ["C1", "T2", "C2"])
def test_oracle_exact_breakpoints_against_the_literal_loop(built, synth, tmp_path, cfg):
    """CPU: the breakpoint pairs of every edge of the final graph (what `_sv.txt` prints as positions) from the literal ExactBreakpoint +
    CountTop, run over the fragments after the literal RawEdgesChim has trimmed them, against the oracle's breakpoints.txt"""
    import copy

    pre = synth(cfg)
    _, dump = ou.run_oracle(built, pre, tmp_path)
    chim = _read_chimrecord(dump / "chimrecord.txt")
    _raw_edges_chim(_Edges([n[:3] for n in ou.read_nodes(dump / "nodes_build.txt")]), chim)  # (trims the blocks in place, as the reference does)
    got = _exact_breakpoints_literal(ou.read_nodes(dump / "nodes_final.txt"), copy.deepcopy(chim))
    edges = ou.read_edges(dump / "edges_final.txt")
    want = ou.read_breakpoints(dump / "breakpoints.txt")
    assert len(edges) == len(want) > 0
    n_exact = 0
    for e, w in zip(edges, want):
        key = (e[0], e[2], bool(e[1]), bool(e[3]))
        g = got.get(key, [])
        if g:
            n_exact += 1
            assert [(b1, b2) for b1, b2, _, _ in w] == g, (e, w, g)
        else:
            assert all(b1 == -1 and b2 == -1 for b1, b2, _, _ in w), (e, w)
    assert n_exact > 0


# ---- SegmentGraph.cpp:2394-2527: GroupConnection, GroupSelect, FilterEdges (and UpdateNodeLink :2894-2910 for the per-node edge lists);
# Min_Edge_Weight 5, MaxAllowedDegree 5, Concord_Dist_Pos 50000, Concord_Dist_Idx 20 (Config.cpp:24-28)
def _filter_edges_literal(nodes, edges, keep, min_w=5, max_deg=5, dist_pos=50000, dist_idx=20):
    chr_, pos, ln, depth = [n[0] for n in nodes], [n[1] for n in nodes], [n[2] for n in nodes], [n[4] for n in nodes]
    head, tail = [[] for _ in nodes], [[] for _ in nodes]
    for e in edges:  # e = (Ind1, Head1, Ind2, Head2, Weight, GroupWeight)
        (head if e[1] else tail)[e[0]].append(e)
        (head if e[3] else tail)[e[2]].append(e)
    weak = lambda e, s: e[5] <= 0.01 * s and e[5] <= min_w
    strong = lambda e, s: e[5] > 0.01 * s or e[5] > min_w

    def group_connection(node, E, sumweight):
        conn = sorted((e[0] if e[0] != node else e[2]) for e in E if strong(e, sumweight))
        label = [-1] * len(conn)
        mindist, index = -1, -1
        for i, cn in enumerate(conn):
            if chr_[cn] == chr_[node] and pos[node] - pos[cn] - ln[cn] <= dist_pos and pos[cn] - pos[node] - ln[node] <= dist_pos:
                if mindist == -1 or mindist > abs(node - cn):
                    mindist, index = abs(node - cn), i
        if index != -1:
            label[index] = 0
            for i in range(index + 1, len(conn)):
                if chr_[conn[i]] == chr_[node] and pos[conn[i]] - pos[conn[i - 1]] - ln[conn[i - 1]] <= dist_pos:
                    label[i] = 0
                else:
                    break
            for i in range(index - 1, -1, -1):
                if chr_[conn[i]] == chr_[node] and pos[conn[i + 1]] - pos[conn[i]] - ln[conn[i]] <= dist_pos:
                    label[i] = 0
                else:
                    break
        count = 0
        if len(label) != 0:
            count = 1 if label[0] == -1 else 0
            if label[0] == -1:
                label[0] = 1
            for i in range(1, len(conn)):
                if label[i] != -1:
                    continue
                elif chr_[conn[i]] != chr_[conn[i - 1]] or pos[conn[i]] - pos[conn[i - 1]] - ln[conn[i - 1]] > dist_pos:
                    count += 1
                label[i] = count
        return count, conn, label

    def group_select(node, E, sumweight, count, conn, label, todelete):
        lw = [0] * (count + 1)
        for e in E:
            if strong(e, sumweight):
                lw[label[conn.index(e[0] if e[0] != node else e[2])]] += e[4]
        maxlabel = 1
        for i in range(1, len(lw)):
            if lw[i] > lw[maxlabel]:
                maxlabel = i
        for e in E:
            if strong(e, sumweight):
                lb = label[conn.index(e[0] if e[0] != node else e[2])]
                if lb != maxlabel and lb != 0:
                    todelete.append(e)

    bad, todelete = [], []
    for i in range(len(nodes)):
        hw, tw = sum(e[4] for e in head[i]), sum(e[4] for e in tail[i])
        s = hw + tw
        todelete += [e for e in head[i] if weak(e, s)] + [e for e in tail[i] if weak(e, s)]
        hc, hconn, hlab = group_connection(i, head[i], s) if head[i] else (0, [], [])
        tc, tconn, tlab = group_connection(i, tail[i], s) if tail[i] else (0, [], [])
        if hc + tc >= max_deg:
            bad.append(i)
        else:
            for cnt, E, w, conn, lab in ((hc, head[i], hw, hconn, hlab), (tc, tail[i], tw, tconn, tlab)):
                if cnt > 1:
                    group_select(i, E, s, cnt, conn, lab, todelete)
                else:
                    todelete += [e for e in E if not weak(e, s) and e[5] < 0.01 * w]
    okey = lambda e: (e[0], e[2], e[1], e[3])  # Edge_t::operator<
    todelete.sort(key=okey)
    bad = set(bad)
    tmp = []
    for i, e in enumerate(edges):
        cond1, cond2 = False, True
        if e[0] not in bad and e[2] not in bad and e[5] > min_w:
            cond1 = True
        elif chr_[e[0]] == chr_[e[2]] and abs(pos[e[2]] - pos[e[0]] - ln[e[0]]) <= dist_pos and e[5] > min_w:
            cond1 = True
        if cond1 and (e[2] - e[0] > dist_idx or e[1] != 0 or e[3] != 1):
            c1, c2 = depth[e[0]], depth[e[2]]
            num, den = (c1, c2) if c1 > c2 else (c2, c1)
            ratio = num / den if den != 0 else (float("inf") if num > 0 else float("nan"))  # (what the double division gives)
            if (e[4] <= min_w + 2 and ratio > 3) or (e[4] > min_w + 2 and ratio > 50):
                cond2 = False
        if keep[i] and cond1 and cond2:
            tmp.append(e)
    tmp.sort(key=okey)
    out, a, b = [], 0, 0  # std::set_difference over the two sorted ranges
    while a < len(tmp):
        if b == len(todelete):
            out.append(tmp[a]); a += 1
        elif okey(tmp[a]) < okey(todelete[b]):
            out.append(tmp[a]); a += 1
        elif okey(todelete[b]) < okey(tmp[a]):
            b += 1
        else:
            a += 1; b += 1
    return out


@pytest.mark.parametrize("cfg,gen,flags", [("C1", (), ()), ("T2", (), ()), ("C2", (), ()), ("T2", (), ("-w", "2")), ("C2", ("--support", "2,6"), ("-w", "1", "-a", "50")),
                                           ("C5", ("--records", "300000", "--tsv", "1500"), ("-w", "1", "-a", "50")), ("C5", ("--records", "300000", "--tsv", "1500", "--support", "2,8"), ())])
def test_oracle_filter_edges_against_the_literal_loop(built, synth, tmp_path, cfg, gen, flags):
    """CPU: the edges that survive FilterEdges (GroupConnection / GroupSelect per node, the depth-ratio rule, the set difference with the
    deleted edges) from the literal loop over the oracle's stage in front of it (nodes with depths, edges with group weights, KeepEdge of
    FilterbyInterleaving), against the oracle's stage behind it"""
    pre = synth(cfg, *gen)
    _, dump = ou.run_oracle(built, pre, tmp_path, *flags)
    nodes = ou.read_nodes(dump / "nodes_build.txt")
    rows = ou.read_edges(dump / "edges_interleave.txt")
    opts = dict(zip(flags[::2], flags[1::2]))
    got = _filter_edges_literal(nodes, [r[:6] for r in rows], [r[6] for r in rows], min_w=int(opts.get("-w", 5)), max_deg=int(opts.get("-a", 5)))
    want = ou.read_edges(dump / "edges_filter.txt")
    assert got == [tuple(w[:6]) for w in want]
    assert 0 < len(want) <= len(rows)


# ---- SegmentGraph.cpp:1968-2124, statement by statement, its slips included (`vEdges[i].Ind1` / `vEdges[i].Ind2` where the neighbour j is
# meant at :2002, :2006, :2021, :2054, and the forward loop's opposite-orientation branch moving the lower ends of its ranges, :2058-2059)
def _filter_by_weight_literal(nodes, edges, min_w=5, dist_pos=50000, dist_idx=20):
    chr_, pos, ln = [n[0] for n in nodes], [n[1] for n in nodes], [n[2] for n in nodes]
    E = [list(e[:6]) for e in edges]  # [Ind1, Head1, Ind2, Head2, Weight, GroupWeight]
    n = len(E)
    p1 = lambda e: pos[e[0]] if e[1] else pos[e[0]] + ln[e[0]]
    p2 = lambda e: pos[e[2]] if e[3] else pos[e[2]] + ln[e[2]]

    def is_disc(e):  # IsDiscordant(int), :159-168
        if chr_[e[0]] != chr_[e[2]]:
            return True
        elif pos[e[2]] - pos[e[0]] - ln[e[0]] > dist_pos and e[2] - e[0] > dist_idx:
            return True
        elif e[1] != 0 or e[3] != 1:
            return True
        return False

    seen = [False] * n
    for i in range(n):
        if seen[i]:
            continue
        ei = E[i]
        chr1, chr2 = chr_[ei[0]], chr_[ei[2]]
        near = [i]
        seen[i] = True
        if ei[1] or not ei[3] or chr1 != chr2:
            I1s, P1s, I2s, P2s = [ei[0], ei[0]], [p1(ei), p1(ei)], [ei[2], ei[2]], [p2(ei), p2(ei)]
            I1o, P1o, I2o, P2o = list(I1s), list(P1s), list(I2s), list(P2s)
            long_group = False
            j = i - 1
            while j > -1 and chr_[E[j][0]] == chr1:
                ej = E[j]
                np1, np2 = p1(ej), p2(ej)
                if ei[0] < min(I1s[0], I1o[0]) - dist_idx or np1 < min(P1s[0], P1o[0]) - dist_pos:
                    break
                if ej[1] == ei[1] and ej[3] == ei[3]:
                    if is_disc(ej) and ej[2] >= I2s[0] - dist_idx and ei[2] <= I2s[1] + dist_idx and np2 >= P2s[0] - dist_pos and np2 <= P2s[1] + dist_pos:
                        near.append(j)
                        I1s[0] = min(I1s[0], ej[0]); P1s[0] = min(P1s[0], np1)
                        I2s[0] = min(I2s[0], ej[2]); I2s[1] = max(I2s[1], ej[2])
                        P2s[0] = min(P2s[0], np2); P2s[1] = max(P2s[1], np2)
                        if I1s[1] >= I2s[0]:
                            long_group = True
                elif ej[1] != ei[1] and ej[3] != ei[3]:
                    if is_disc(ej) and ej[2] >= I2o[0] - dist_idx and ei[2] <= I2o[1] + dist_idx and np2 >= P2o[0] - dist_pos and np2 <= P2o[1] + dist_pos:
                        near.append(j)
                        I1o[0] = min(I1o[0], ej[0]); P1o[0] = min(P1o[0], np1)
                        I2o[0] = min(I2o[0], ej[2]); I2o[1] = max(I2o[1], ej[2])
                        P2o[0] = min(P2o[0], np2); P2o[1] = max(P2o[1], np2)
                        if I1o[1] >= I2o[0]:
                            long_group = True
                j -= 1
            j = i + 1
            while j < n and chr_[E[j][0]] == chr1:
                ej = E[j]
                np1, np2 = p1(ej), p2(ej)
                if ej[0] > max(I1s[1], I1o[1]) + dist_idx or np1 > max(P1s[1], P1o[1]) + dist_pos:
                    break
                if ej[1] == ei[1] and ej[3] == ei[3]:
                    if is_disc(ej) and ej[2] >= I2s[0] - dist_idx and ej[2] <= I2s[1] + dist_idx and np2 >= P2s[0] - dist_pos and np2 <= P2s[1] + dist_pos:
                        near.append(j)
                        I1s[1] = max(I1s[1], ej[0]); P1s[1] = max(P1s[1], np1)
                        I2s[0] = min(I2s[0], ej[2]); I2s[1] = max(I2s[1], ej[2])
                        P2s[0] = min(P2s[0], np2); P2s[1] = max(P2s[1], np2)
                        if I1s[1] >= I2s[0]:
                            long_group = True
                elif ej[1] != ei[1] and ej[3] != ei[3]:
                    if is_disc(ej) and ej[2] >= I2o[0] - dist_idx and ei[2] <= I2o[1] + dist_idx and np2 >= P2o[0] - dist_pos and np2 <= P2o[1] + dist_pos:
                        near.append(j)
                        I1o[0] = min(I1o[0], ej[0]); P1o[0] = min(P1o[0], np1)
                        I2o[0] = min(I2o[0], ej[2]); I2o[1] = max(I2o[1], ej[2])
                        P2o[0] = min(P2o[0], np2); P2o[1] = max(P2o[1], np2)
                        if I1o[1] >= I2o[0]:
                            long_group = True
                j += 1
            near = sorted(set(near))
            if not long_group:
                s = sum(E[k][4] for k in near)
                for k in near:
                    E[k][5] = s if E[k][5] < s else E[k][5]
                    seen[k] = True
            else:
                for k in near:
                    E[k][5] = E[k][4]
                    seen[k] = True
        else:
            pos1, pos2 = p1(ei), p2(ei)
            j = i - 1
            while j > -1 and E[j][0] >= ei[0] - dist_idx and chr_[E[j][0]] == chr1 and pos[E[j][0]] + ln[E[j][0]] >= pos1 - dist_pos:
                ej = E[j]
                if ej[2] > ei[0] and ei[1] == ej[1] and ei[3] == ej[3] and chr_[ej[0]] == chr1 and chr_[ej[2]] == chr2 and abs(ej[2] - ei[2]) <= dist_idx and abs(p1(ej) - pos1) <= dist_pos and abs(p2(ej) - pos2) <= dist_pos:
                    near.append(j)
                j -= 1
            j = i + 1
            while j < n and E[j][0] <= ei[0] + dist_idx and chr_[E[j][0]] == chr1 and pos[E[j][0]] <= pos1 + dist_pos:
                ej = E[j]
                if ej[0] < ei[2] and ei[1] == ej[1] and ei[3] == ej[3] and chr_[ej[0]] == chr1 and chr_[ej[2]] == chr2 and abs(ej[2] - ei[2]) <= dist_idx and abs(p1(ej) - pos1) <= dist_pos and abs(p2(ej) - pos2) <= dist_pos:
                    near.append(j)
                j += 1
            ei[5] = sum(E[k][4] for k in sorted(set(near)))
    return [tuple(e) for e in E if e[5] > min_w - 2]


@pytest.mark.parametrize("cfg,gen,flags", [("C1", (), ()), ("T2", (), ()), ("C2", (), ("-w", "2")), ("C2", ("--support", "2,6"), ("-w", "1", "-a", "50")),
                                           ("C5", ("--records", "300000", "--tsv", "1500"), ("-w", "1", "-a", "50")), ("C5", ("--records", "300000", "--tsv", "1500", "--support", "2,8"), ())])
def test_oracle_filter_by_weight_against_the_literal_loop(built, synth, tmp_path, cfg, gen, flags):
    """CPU: the group weight of every edge and the edges that pass the relaxed threshold, from the literal FilterbyWeight over the oracle's
    BuildEdges stage, against the oracle's stage behind it"""
    pre = synth(cfg, *gen)
    _, dump = ou.run_oracle(built, pre, tmp_path, *flags)
    opts = dict(zip(flags[::2], flags[1::2]))
    rows = ou.read_edges(dump / "edges_build.txt")
    got = _filter_by_weight_literal(ou.read_nodes(dump / "nodes_build.txt"), rows, min_w=int(opts.get("-w", 5)))
    want = [tuple(w[:6]) for w in ou.read_edges(dump / "edges_weight.txt")]
    assert got == want
    assert 0 < len(want) <= len(rows) and any(w[5] != w[4] for w in want)


# ---- SegmentGraph.cpp:2528-2604: nodes no edge touches are merged run by run (never across a chromosome), Support summed, AvgDepth the
# length-weighted mean in the order the reference adds it up; edges re-indexed
def _compress_node_literal(nodes, edges):
    linked = sorted(set([e[0] for e in edges] + [e[2] for e in edges]))
    assert linked
    new, old_new = [], {}

    def merged(lo, hi):  # one node for vNodes[lo .. hi)
        length = nodes[hi - 1][1] + nodes[hi - 1][2] - nodes[lo][1]
        support, depth = 0, 0.0
        for k in range(lo, hi):
            support += nodes[k][3]
            depth += nodes[k][4] * nodes[k][2]
        return (nodes[lo][0], nodes[lo][1], length, support, depth / length)

    def run(startidx, endidx):
        lastinsert = startidx
        for j in range(startidx, endidx):
            if nodes[j][0] != nodes[lastinsert][0]:
                new.append(merged(lastinsert, j))
                lastinsert = j
        if lastinsert != endidx:
            new.append(merged(lastinsert, endidx))

    for i, endidx in enumerate(linked):
        run(0 if i == 0 else linked[i - 1] + 1, endidx)
        new.append(tuple(nodes[endidx][:5]))
        old_new[endidx] = len(new) - 1
    if linked[-1] != len(nodes) - 1:
        run(linked[-1] + 1, len(nodes))
    return new, [(old_new[e[0]], e[1], old_new[e[2]], e[3]) + tuple(e[4:6]) for e in edges]


@pytest.mark.parametrize("cfg,gen,flags", [("C1", (), ()), ("T2", (), ()), ("C2", (), ()), ("C5", ("--records", "300000", "--tsv", "1500", "--support", "2,8"), ())])
def test_oracle_compress_node_against_the_literal_loop(built, synth, tmp_path, cfg, gen, flags):
    """CPU: node table and re-indexed edges behind CompressNode from the literal loop over the oracle's stage in front of it -- the merged
    nodes' AvgDepth as the same IEEE doubles (summed and divided in the reference's order)"""
    pre = synth(cfg, *gen)
    _, dump = ou.run_oracle(built, pre, tmp_path, *flags)
    nodes, edges = _compress_node_literal(ou.read_nodes(dump / "nodes_build.txt"), ou.read_edges(dump / "edges_filter.txt"))
    assert nodes == [tuple(n[:5]) for n in ou.read_nodes(dump / "nodes_compress.txt")]
    assert edges == [tuple(e[:6]) for e in ou.read_edges(dump / "edges_compress.txt")]
    # (on the small samples every node carries an edge and nothing is merged; the filtered dense sample has unlinked runs)
    assert len(nodes) < len(ou.read_nodes(dump / "nodes_build.txt")) or cfg != "C5"


# ---- SegmentGraph.cpp:2693-2892 (FurtherCompressNode), :2911-3003 (DFS, ConnectedComponent), :3005-3010 (MultiplyDisEdges, DiscordantRatio 8)
def _further_compress_literal(nodes, edges, dist_pos=50000, dist_idx=20, ratio=8):
    chr_, pos, ln = [n[0] for n in nodes], [n[1] for n in nodes], [n[2] for n in nodes]
    n = len(nodes)

    def link(nn, E):
        head, tail = [[] for _ in range(nn)], [[] for _ in range(nn)]
        for e in E:
            (head if e[1] else tail)[e[0]].append(e)
            (head if e[3] else tail)[e[2]].append(e)
        return head, tail

    def is_disc(e, c=chr_, p=pos, l=ln):
        if c[e[0]] != c[e[2]]:
            return True
        elif p[e[2]] - p[e[0]] - l[e[0]] > dist_pos and e[2] - e[0] > dist_idx:
            return True
        elif e[1] != 0 or e[3] != 1:
            return True
        return False

    head, tail = link(n, edges)
    close = lambda a, b: abs(a[0] - b[0]) <= dist_idx and abs(a[2] - b[2]) <= dist_idx
    samehd = lambda a, b: a[1] == b[1] and a[3] == b[3]
    samechr = lambda a, b: chr_[a[0]] == chr_[b[0]] and chr_[a[2]] == chr_[b[2]]
    merge = [-1] * n
    cur = rightmost = 0
    unset_uses = 0

    def next_with(i, mindis):
        nxt = []
        j = i + 1
        while j < n and j < i + 20 and j < mindis and chr_[i] == chr_[j]:
            nxt += [e for e in head[j] if is_disc(e)] + [e for e in tail[j] if is_disc(e)]
            if nxt:
                break
            j += 1
        return j, nxt

    def squeeze(V, at=None):  # keeps V[k + 1] unless it belongs to the group of V[k]
        out = [V[0]]
        for k in range(len(V) - 1):
            e1, e2 = V[k], V[k + 1]
            if at is None:
                same = close(e1, e2) and samechr(e1, e2) and samehd(e1, e2)
            else:
                same = ((e1[0] == at and e2[0] == at) or (e1[2] == at and e2[2] == at)) and close(e1, e2) and samechr(e1, e2) and samehd(e1, e2)
            if not same:
                out.append(e2)
        return out

    def cross(A, B):
        a_eq, b_eq = [False] * len(A), [False] * len(B)
        for k, e1 in enumerate(A):
            for l, e2 in enumerate(B):
                if e1[2] > e2[0] and e2[2] > e1[0] and samechr(e1, e2) and close(e1, e2) and samehd(e1, e2):
                    a_eq[k] = b_eq[l] = True
        return all(a_eq) and all(b_eq)

    for i in range(n):
        this = []
        mindis = None
        if i != 0 and chr_[i] != chr_[i - 1] and cur == merge[i - 1]:
            cur += 1
        for e in head[i] + tail[i]:
            if is_disc(e):
                this.append(e)
            else:
                rightmost = max(rightmost, e[0], e[2])
        if this:
            mindis = this[0][2] if this[0][0] == i else i + 20
            tmp = [this[0]]
            for k in range(len(this) - 1):
                e1, e2 = this[k], this[k + 1]
                same = (e1[0] == i and e2[0] == i) or (e1[2] == i and e2[2] == i)
                if not (close(e1, e2) and samehd(e1, e2)):  # (:2721: no chromosome test in this one)
                    same = False
                if not same:
                    tmp.append(e2)
                mindis = min(mindis, e2[2] if e2[0] == i else i + 20)
            this = tmp
        if merge[i] == -1:
            if not this and i < rightmost:
                merge[i] = cur
            elif not this and i == rightmost:
                merge[i] = cur
                cur += 1
                rightmost += 1
            else:
                if mindis is None:  # (the reference reads it unset here: a node without discordant edges behind `rightmost`)
                    unset_uses += 1
                    mindis = i + 20
                if i != 0 and cur == merge[i - 1]:
                    cur += 1
                j, nxt = next_with(i, mindis)
                equivalent = len(nxt) != 0
                if nxt:
                    equivalent = cross(this, squeeze(nxt, at=j))
                if not equivalent:
                    merge[i] = cur
                    cur += 1
                else:
                    for k in range(i, j + 1):
                        merge[k] = cur
                rightmost = i + 1
        elif this:
            j, nxt = next_with(i, mindis)
            equivalent = len(nxt) != 0
            if nxt:
                equivalent = cross(squeeze(this), squeeze(nxt))
            if not equivalent:
                cur += 1
            else:
                for k in range(i, j + 1):
                    merge[k] = cur
            rightmost = i + 1
    for i in range(n - 1):
        assert merge[i] == merge[i + 1] or merge[i] + 1 == merge[i + 1]
    new_nodes = []
    ind = 0
    while ind < n:
        j = ind
        while j < n and merge[j] == merge[ind]:
            j += 1
        new_nodes.append((chr_[ind], pos[ind], pos[j - 1] + ln[j - 1] - pos[ind]))
        ind = j
    raw = []
    for e in edges:
        if merge[e[0]] != merge[e[2]]:
            a, b = merge[e[0]], merge[e[2]]
            raw.append((b, e[3], a, e[1], e[4]) if a > b else (a, e[1], b, e[3], e[4]))  # Edge_t's constructor
    raw.sort(key=lambda e: (e[0], e[2], e[1], e[3]))
    new_edges = []
    for e in raw:
        if not new_edges or new_edges[-1][:4] != list(e[:4]):
            new_edges.append(list(e))
        else:
            new_edges[-1][4] += e[4]
    # ConnectedComponent: labels in the order of the first node of every component (the traversal order does not reach the labels)
    nn = len(new_nodes)
    h2, t2 = link(nn, [tuple(e) for e in new_edges])
    label, cur_label = [-1] * nn, 0
    for start in range(nn):
        if label[start] != -1:
            continue
        stack = [start]
        while stack:
            v = stack.pop()
            if label[v] == -1:
                label[v] = cur_label
                for e in h2[v] + t2[v]:
                    stack.append(e[0] if e[0] != v else e[2])
        cur_label += 1
    c2, p2, l2 = [x[0] for x in new_nodes], [x[1] for x in new_nodes], [x[2] for x in new_nodes]
    for e in new_edges:  # MultiplyDisEdges
        if is_disc(e, c2, p2, l2):
            e[4] = int(ratio) * e[4]
    return new_nodes, label, [tuple(e) for e in new_edges], unset_uses


@pytest.mark.parametrize("cfg,gen,flags", [("C1", (), ()), ("T2", (), ()), ("C2", (), ()), ("C2", ("--support", "2,6"), ("-w", "1", "-a", "50")),
                                           ("C5", ("--records", "300000", "--tsv", "1500"), ("-w", "1", "-a", "50")), ("C5", ("--records", "300000", "--tsv", "1500", "--support", "2,8"), ())])
def test_oracle_final_graph_against_the_literal_loops(built, synth, tmp_path, cfg, gen, flags):
    """CPU: the final node table, the component labels and the final edges (weights of discordant edges multiplied) from the literal
    FurtherCompressNode + ConnectedComponent + MultiplyDisEdges over the oracle's stage behind CompressNode, against the oracle's final
    graph"""
    pre = synth(cfg, *gen)
    _, dump = ou.run_oracle(built, pre, tmp_path, *flags)
    nodes, label, edges, unset = _further_compress_literal(ou.read_nodes(dump / "nodes_compress.txt"), ou.read_edges(dump / "edges_compress.txt"))
    want_nodes = ou.read_nodes(dump / "nodes_final.txt")
    assert nodes == [n[:3] for n in want_nodes]
    assert all(n[3] == 0 and n[4] == 0.0 for n in want_nodes)  # (Support / AvgDepth are reset, ledger B15)
    assert label == [n[5] for n in want_nodes]
    assert edges == [tuple(e[:5]) for e in ou.read_edges(dump / "edges_final.txt")]
    assert len(nodes) < len(ou.read_nodes(dump / "nodes_compress.txt"))


# ---- main.cpp:50-53 (Node_NewChr), SegmentGraph.cpp:3012-3017 (DeMultiplyDisEdges), WriteIO.cpp:45-124 (WriteBEDPE): the text of `_sv.txt`
# from the final graph, the component orders and the breakpoints with their support.  std::sort by weight is not stable: the row order among
# equal weights is what introsort makes of it (`_std_sort`).
def _write_bedpe_literal(refnames, nodes, edges, bps, orders, ratio=8, dist_pos=50000, dist_idx=20):
    chr_, pos, ln = [n[0] for n in nodes], [n[1] for n in nodes], [n[2] for n in nodes]

    def is_disc(e):
        if chr_[e[0]] != chr_[e[2]]:
            return True
        elif pos[e[2]] - pos[e[0]] - ln[e[0]] > dist_pos and e[2] - e[0] > dist_idx:
            return True
        elif e[1] != 0 or e[3] != 1:
            return True
        return False

    E = []
    for e, bp in zip(edges, bps):
        w = int(e[4] / ratio) if is_disc(e) and ratio != 1 else e[4]
        E.append((e[0], e[1], e[2], e[3], w, bp))
    _std_sort(E, lambda a, b: a[4] > b[4])
    where = [None] * len(nodes)
    for i, comp in enumerate(orders):
        for j, v in enumerate(comp):
            where[abs(v) - 1] = (i, j)
    out = ["# chrom1\tstart1\tend1\tchrom2\tstart2\tend2\tname\tscore\tstrand1\tstrand2\tnum_concordantfrag_bp1\tnum_concordantfrag_bp2\n"]
    for i1, h1, i2, h2, w, bp in E:
        flag_chr = chr_[i1] == chr_[i2]
        flag_ori = h1 == 0 and h2 == 1
        flag_dist = pos[i2] - pos[i1] - ln[i1] <= dist_pos or i2 - i1 <= dist_idx
        if flag_chr and flag_ori and flag_dist:
            continue
        p1, p2 = where[i1], where[i2]
        flag = False
        if p1[0] == p2[0] and p1[1] < p2[1] and bool(h1) == (orders[p1[0]][p1[1]] < 0) and bool(h2) == (orders[p2[0]][p2[1]] > 0):
            flag = True
        elif p1[0] == p2[0] and p1[1] > p2[1] and bool(h2) == (orders[p2[0]][p2[1]] < 0) and bool(h1) == (orders[p1[0]][p1[1]] > 0):
            flag = True
        if not flag:
            continue
        if all(b[0] == -1 and b[1] == -1 for b in bp):  # no exact breakpoint: the node ends the edge leaves from
            pairs = [(pos[i1] if h1 else pos[i1] + ln[i1], pos[i2] if h2 else pos[i2] + ln[i2])]
        else:
            pairs = [(b[0], b[1]) for b in bp]
        assert len(pairs) == len(bp)
        for (b1, b2), sup in zip(pairs, bp):
            row = [refnames[chr_[i1]]] + ([str(b1), str(pos[i1] + ln[i1])] if h1 else [str(pos[i1]), str(b1)])
            row += [refnames[chr_[i2]]] + ([str(b2), str(pos[i2] + ln[i2])] if h2 else [str(pos[i2]), str(b2)])
            row += [".", str(w), "-" if h1 else "+", "-" if h2 else "+", str(sup[2]), str(sup[3])]
            out.append("\t".join(row) + "\n")
    return "".join(out)


@pytest.mark.parametrize("cfg,gen,flags", [("C1", (), ()), ("T2", (), ()), ("C2", (), ()), ("C2", ("--support", "2,6"), ("-w", "1", "-a", "50")),
                                           ("C5", ("--records", "300000", "--tsv", "1500"), ("-w", "1", "-a", "50"))])
def test_oracle_sv_text_against_the_literal_writer(built, synth, tmp_path, cfg, gen, flags):
    """CPU: `_sv.txt` byte for byte from the literal WriteBEDPE over the oracle's final graph, component orders and breakpoints (row order
    among equal scores included)"""
    import squid_amd

    pre = synth(cfg, *gen)
    sv_path, dump = ou.run_oracle(built, pre, tmp_path, *flags)
    names, _ = squid_amd.read_header(f"{pre}.bam")
    got = _write_bedpe_literal(names, ou.read_nodes(dump / "nodes_final.txt"), ou.read_edges(dump / "edges_final.txt"), ou.read_breakpoints(dump / "breakpoints.txt"),
                               ou.read_orders(dump / "orders.txt"))
    want = open(sv_path).read()
    assert got == want
    assert want.count("\n") > 1


# ---- SegmentGraph.cpp:2161-2277 (FilterbyInterleaving), slips kept: `vEdges[i].Ind1` / `vEdges[i].Ind2` for the neighbour at :2192 and
# :2194, and the stray `;` of :2267 that lets overlapInd1 be computed from value-initialised (0, 0) ranges when a side has no edge
def _filter_by_interleaving_literal(nodes, edges, dist_pos=50000, dist_idx=20):
    chr_, pos, ln = [n[0] for n in nodes], [n[1] for n in nodes], [n[2] for n in nodes]
    E = edges
    n = len(E)
    p1 = lambda e: pos[e[0]] if e[1] else pos[e[0]] + ln[e[0]]
    p2 = lambda e: pos[e[2]] if e[3] else pos[e[2]] + ln[e[2]]
    seen, keep = [False] * n, [True] * n
    for i in range(n):
        if seen[i]:
            continue
        ei = E[i]
        if ei[2] - ei[0] <= dist_idx or (chr_[ei[0]] == chr_[ei[2]] and abs(pos[ei[0]] - pos[ei[2]]) <= dist_pos):
            seen[i] = True
            keep[i] = True
            continue
        chr1 = chr_[ei[0]]
        minpos1 = maxpos1 = p1(ei)
        minidx1 = maxidx1 = ei[0]
        minpos2 = maxpos2 = p2(ei)
        minidx2 = maxidx2 = ei[2]
        long_group = False
        near = [i]
        j = i - 1
        while j > -1 and chr_[E[j][0]] == chr1:
            ej = E[j]
            np1, np2 = p1(ej), p2(ej)
            if ei[0] < minidx1 - dist_idx or np1 < minpos1 - dist_pos:
                break
            if ej[2] >= minidx2 - dist_idx and ei[2] <= maxidx2 + dist_idx and np2 >= minpos2 - dist_pos and np2 <= maxpos2 + dist_pos:
                near.append(j)
                minidx1 = min(minidx1, ej[0]); minpos1 = min(minpos1, np1)
                minidx2 = min(minidx2, ej[2]); maxidx2 = max(maxidx2, ej[2])
                minpos2 = min(minpos2, np2); maxpos2 = max(maxpos2, np2)
                if maxidx1 >= minidx2:
                    long_group = True
                    break
            j -= 1
        j = i + 1
        while j < n and chr_[E[j][0]] == chr1:
            ej = E[j]
            np1, np2 = p1(ej), p2(ej)
            if ej[0] > maxidx1 + dist_idx or np1 > maxpos1 + dist_pos:
                break
            if ej[2] >= minidx2 - dist_idx and ej[2] <= maxidx2 + dist_idx and np2 >= minpos2 - dist_pos and np2 <= maxpos2 + dist_pos:
                near.append(j)
                maxidx1 = max(maxidx1, ej[0]); maxpos1 = max(maxpos1, np1)
                minidx2 = min(minidx2, ej[2]); maxidx2 = max(maxidx2, ej[2])
                minpos2 = min(minpos2, np2); maxpos2 = max(maxpos2, np2)
                if maxidx1 >= minidx2:
                    long_group = True
                    break
            j += 1
        if long_group:
            for k in near:
                seen[k] = True
            continue
        near.sort()
        g1h, g1t, g2h, g2t = [], [], [], []
        for k in near:
            e = E[k]
            (g1h if e[1] else g1t).append(e[2])
            (g2h if e[3] else g2t).append(e[0])
        rng = lambda v: (min(v), max(v)) if v else (0, 0)
        r1h, r1t, r2h, r2t = rng(g1h), rng(g1t), rng(g2h), rng(g2t)
        overlap1 = min(r1h[1], r1t[1]) >= max(r1h[0], r1t[0])  # (computed whatever the group sizes: the `if` in front of it ends in `;`)
        overlap2 = False
        if g2h and g2t:
            overlap2 = min(r2h[1], r2t[1]) >= max(r2h[0], r2t[0])
        if overlap1 and overlap2:
            for k in near:
                keep[k] = False
        for k in near:
            seen[k] = True
    return keep


@pytest.mark.parametrize("cfg,gen,flags", [("C1", (), ()), ("T2", (), ()), ("C2", ("--support", "2,6"), ("-w", "1", "-a", "50")),
                                           ("C5", ("--records", "300000", "--tsv", "1500"), ("-w", "1", "-a", "50")), ("C5", ("--records", "300000", "--tsv", "1500", "--support", "2,8"), ()),
                                           ("T2", ("--interleave", "3"), ()), ("C2", ("--interleave", "6"), ()), ("C5", ("--records", "300000", "--tsv", "1500", "--interleave", "40"), ("-w", "1", "-a", "50"))])
def test_oracle_filter_by_interleaving_against_the_literal_loop(built, synth, tmp_path, cfg, gen, flags):
    """CPU: KeepEdge of FilterbyInterleaving from the literal loop over the oracle's stage behind FilterbyWeight, against the oracle's.  The
    default generator settings plant no interleaved junction pairs (the answer is "keep" everywhere: those samples pin the walks and the
    long-group rule); `--interleave K` plants K pairs of junctions between the same two exons, head-head and tail-tail, on which the overlap
    rule (:2264-2273) fires -- the test asserts that it does."""
    pre = synth(cfg, *gen)
    _, dump = ou.run_oracle(built, pre, tmp_path, *flags)
    rows = ou.read_edges(dump / "edges_interleave.txt")
    assert [tuple(r[:6]) for r in rows] == [tuple(r[:6]) for r in ou.read_edges(dump / "edges_weight.txt")]  # (the stage only decides KeepEdge)
    got = _filter_by_interleaving_literal(ou.read_nodes(dump / "nodes_build.txt"), [r[:6] for r in rows])
    assert got == [bool(r[6]) for r in rows]
    if "--interleave" in gen:
        k = int(gen[gen.index("--interleave") + 1])
        assert sum(1 for r in rows if not r[6]) >= k, "planted interleaved pairs must lose their edges"
