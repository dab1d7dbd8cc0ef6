"""The segmentation automaton of `BuildNode_STAR` (`src/SegmentGraph.cpp:203-764`), statement by statement in Python -- the one stage of the hot
path that had no third reading: 430 lines of iterator state over the sorted discordant blocks, the two sliding windows
of concordant blocks, the `ConcordRest` heap and the list of nodes emitted so far.  Written from the reference text alone (no code shared with
`oracle/` or the library), over records decoded by `test_literal_loops._records_from_bam` (itself a literal constructor) and the merged chimeric
fragments (whose literal build `test_literal_loops` checks against the same dump).  What the reference leaves to chance is settled the way
SURVEY.md's ledger records it, spelled out where it happens:

* B10  `PartAlignPos.resize(RefLength.size())`: the list starts with one (0, 0) per reference sequence (`:204`);
* B11  `bamdiscordant.back()` is read while the list may be empty (`:257`): here an assertion -- no test input gets there with an empty list;
* B12  the stream loop stops at the first kept record behind the last cluster (`:338-339`): the record has been pushed to ReadsMain / ReadsOther,
       nothing else of it is looked at; `n_break` = how many kept records the loop consumed;
* B21  between the moment the last cluster is consumed and that `break`, `itdisstart == cend()` is dereferenced (`:604-606,620,633,640,644`):
       read as a block (RefID 0, RefPos 0, MatchRef 0) -- what fresh zero pages behind the vector's end hold in a large run;
* B8   `sort(bamdiscordant)` is libstdc++'s introsort under `operator<` (RefID, RefPos): restated (`_std_sort`), because the overlap prefix of a
       cluster (`:401-408`) follows the order of blocks with equal keys.

Compared: the seed nodes right after the loop, the nodes after `NormalizeSeedNodes` + the tiling of the genome (`:706-761`), the number of kept
records and the record of the break -- against the oracle's dumps (CPU suite) and the library's node table (GPU suite), on C1, T2, C2, the
low-support C2 and a C2 with interleaved junction pairs."""
import heapq

import pytest

import oracle_util as ou
import test_literal_loops as ll

SENTINEL = {"RefID": 0, "RefPos": 0, "ReadPos": 0, "MatchRef": 0, "MatchRead": 0, "IsReverse": False, "IsFirstRead": False}  # ledger B21


def _same(a, b):  # SingleBamRec_t::Same, SingleBamRec.h:51-53
    return all(a[k] == b[k] for k in ("RefID", "RefPos", "ReadPos", "MatchRead", "MatchRef", "IsReverse", "IsFirstRead"))


def _less(a, b):  # SingleBamRec_t::operator<, SingleBamRec.h:39-44
    return a["RefID"] < b["RefID"] if a["RefID"] != b["RefID"] else a["RefPos"] < b["RefPos"]


def _discordant_lists(chim, n_ref):  # :203-264
    PartAlignPos = [(0, 0)] * n_ref  # ledger B10
    bamdiscordant = []
    for frag in chim:
        it = dict(frag)
        it["F"] = [dict(b, IsFirstRead=True) for b in frag["F"]]
        it["S"] = [dict(b, IsFirstRead=False) for b in frag["S"]]
        F, S = it["F"], it["S"]
        if ll._is_end_discordant(it, True) or ll._is_end_discordant(it, False) or ll._is_single_anchored(it) or ll._is_pair_discordant(it):
            bamdiscordant.extend(F)
            bamdiscordant.extend(S)
        else:
            firstinserted = secondinserted = False
            previnserted = -1
            if len(F) > 0:
                for i in range(len(F) - 1):
                    if abs(F[i]["RefPos"] - F[i + 1]["RefPos"]) > 750000:
                        if previnserted != i:
                            bamdiscordant.append(F[i])
                        bamdiscordant.append(F[i + 1])
                        previnserted = i + 1
                        if i + 1 == len(F) - 1:
                            firstinserted = True
            previnserted = -1
            if len(S) > 0:
                for i in range(len(S) - 1):
                    if abs(S[i]["RefPos"] - S[i + 1]["RefPos"]) > 750000:
                        if previnserted != i:
                            bamdiscordant.append(S[i])
                        bamdiscordant.append(S[i + 1])
                        previnserted = i + 1
                        if i + 1 == len(S) - 1:
                            secondinserted = True
            if len(F) > 0 and len(S) > 0:
                if abs(F[-1]["RefPos"] - S[-1]["RefPos"]) > 750000:
                    if not firstinserted:
                        bamdiscordant.append(F[-1])
                        firstinserted = True
                    if not secondinserted:
                        bamdiscordant.append(S[-1])
                        secondinserted = True
            if not firstinserted and not secondinserted:
                if len(F) != 0 and F[0]["ReadPos"] > 15 and not it["flow"]:
                    PartAlignPos.append((F[0]["RefID"], F[0]["RefPos"] + F[0]["MatchRef"] if F[0]["IsReverse"] else F[0]["RefPos"]))
                if len(F) != 0 and it["ftl"] - F[-1]["ReadPos"] - F[-1]["MatchRead"] > 15 and not it["flow"]:
                    PartAlignPos.append((F[-1]["RefID"], F[-1]["RefPos"] if F[-1]["IsReverse"] else F[-1]["RefPos"] + F[-1]["MatchRef"]))
                if len(S) != 0 and S[0]["ReadPos"] > 15 and not it["slow"]:
                    PartAlignPos.append((S[0]["RefID"], S[0]["RefPos"] + S[0]["MatchRef"] if S[0]["IsReverse"] else S[0]["RefPos"]))
                if len(S) != 0 and it["stl"] - S[-1]["ReadPos"] - S[-1]["MatchRead"] > 15:
                    assert len(bamdiscordant) > 0, "ledger B11: bamdiscordant.back() of an empty list"
                    if not _same(bamdiscordant[-1], S[-1]) and not it["slow"]:
                        PartAlignPos.append((S[-1]["RefID"], S[-1]["RefPos"] if S[-1]["IsReverse"] else S[-1]["RefPos"] + S[-1]["MatchRef"]))
    PartAlignPos.sort()  # (the comparator orders by both fields: equal elements are identical)
    ll._std_sort(bamdiscordant, _less)  # ledger B8
    return PartAlignPos, bamdiscordant


def _build_node_star_literal(rec, chim, ReadLen, ref_len, min_mapq):
    thresh = 3
    PartAlignPos, BD = _discordant_lists(chim, len(ref_len))
    nBD, nPA = len(BD), len(PartAlignPos)
    D = lambda i: BD[i] if i < nBD else SENTINEL  # ledger B21
    itdisstart = itdisend = 0
    itpartstart = itpartend = 0
    ConcordRest = []  # heap under MinHeapComp: front() is a smallest element by (RefID, RefPos)
    heapseq = 0
    ConcordantCluster, PartialAlignCluster = [], []
    offsetCC = offsetPA = 0
    disChr = otherChr = nextdisChr = 0
    disrightmost = otherrightmost = nextdisrightmost = 0
    markedNodeStart = markedNodeChr = -1
    vNodes = []  # [Chr, Position, Length]
    lastreadrec = ([], [])
    kept = 0
    n_break = -1
    refid, pos, mref, mpos, flag, mapq, aux, off, totlen = (rec[k].tolist() for k in ("refid", "pos", "mate_refid", "mate_pos", "flag", "mapq", "aux", "blk_off", "totlen"))
    b_refpos, b_matchref, b_readpos, b_matchread = (rec[k].tolist() for k in ("b_refpos", "b_matchref", "b_readpos", "b_matchread"))

    def seek_cluster():  # :340-348 == :605-613
        nonlocal disrightmost, disChr, nextdisrightmost, nextdisChr, itdisend
        disrightmost, disChr = nextdisrightmost, nextdisChr
        nextdisrightmost = D(itdisstart)["RefPos"] + D(itdisstart)["MatchRef"]
        itdisend = itdisstart
        while itdisend != nBD and BD[itdisend]["RefID"] == BD[itdisstart]["RefID"] and BD[itdisend]["RefPos"] < nextdisrightmost + ReadLen:
            nextdisrightmost = nextdisrightmost if nextdisrightmost > BD[itdisend]["RefPos"] + BD[itdisend]["MatchRef"] else BD[itdisend]["RefPos"] + BD[itdisend]["MatchRef"]
            nextdisChr = BD[itdisend]["RefID"]
            itdisend += 1

    for r in range(len(refid)):
        f = flag[r]
        IsFirstMate, IsSecondMate, IsMapped, IsMateMapped = bool(f & 0x40), bool(f & 0x80), not (f & 0x4), not (f & 0x8)
        IsReverseStrand, IsMateReverseStrand, IsProperPair = bool(f & 0x10), bool(f & 0x20), bool(f & 0x2)
        # :297-303
        if aux[r] & 1 or mapq[r] < min_mapq or f & 0x400 or not IsMapped or refid[r] == -1 or aux[r] & 2:
            continue
        own = [{"RefID": refid[r], "RefPos": b_refpos[k], "ReadPos": b_readpos[k], "MatchRef": b_matchref[k], "MatchRead": b_matchread[k], "IsReverse": IsReverseStrand, "IsFirstRead": IsFirstMate}
               for k in range(off[r], off[r + 1])]
        low = bool(aux[r] & 4)
        # ReadRec_t readrec(record): every block on the side of the record's own mate flag (anything that is not first mate is second, ledger B4)
        rF, rS = (own, []) if IsFirstMate else ([], own)
        tF, tS = list(rF), list(rS)  # tmpreadrec
        ll._std_sort(tF, lambda a, b: a["ReadPos"] < b["ReadPos"])
        ll._std_sort(tS, lambda a, b: a["ReadPos"] < b["ReadPos"])
        if IsFirstMate and IsMateMapped and mref[r] != -1:
            tS.append({"RefID": mref[r], "RefPos": mpos[r], "ReadPos": 0, "MatchRef": 15, "MatchRead": 15, "IsReverse": IsMateReverseStrand, "IsFirstRead": False})
        elif not IsFirstMate and IsMateMapped and mref[r] != -1:
            tF.append({"RefID": mref[r], "RefPos": mpos[r], "ReadPos": 0, "MatchRef": 15, "MatchRead": 15, "IsReverse": IsMateReverseStrand, "IsFirstRead": False})
        key = ([(b["RefID"], b["RefPos"], b["MatchRef"]) for b in tF], [(b["RefID"], b["RefPos"], b["MatchRef"]) for b in tS])
        if ll._equal(lastreadrec, key):
            continue
        lastreadrec = key
        kept += 1
        # (:320-337 push ReadsMain / ReadsOther: not needed for the nodes)
        if itdisstart == nBD:  # :338-339, ledger B12
            n_break = kept
            break
        if itdisend - itdisstart <= 0:
            seek_cluster()
        record_RefID, record_Position = refid[r], pos[r]
        # :353
        while itdisstart != nBD and (BD[itdisstart]["RefID"] < record_RefID or (BD[itdisstart]["RefID"] == record_RefID and nextdisrightmost < record_Position)):
            curEndPos = curStartPos = 0
            disStartPos = disEndPos = disCount = -1
            isClusternSplit = False
            if markedNodeStart != -1 and BD[itdisstart]["RefID"] != markedNodeChr:
                markedNodeChr = markedNodeStart = -1
            while len(ConcordantCluster) != offsetCC and ConcordantCluster[offsetCC]["RefID"] < BD[itdisstart]["RefID"]:
                offsetCC += 1
            while len(PartialAlignCluster) != offsetPA and PartialAlignCluster[offsetPA]["RefID"] < BD[itdisstart]["RefID"]:
                offsetPA += 1
            if len(ConcordantCluster) != offsetCC and BD[itdisstart]["RefPos"] > ConcordantCluster[-1]["RefPos"] + ConcordantCluster[-1]["MatchRef"] + ReadLen:
                offsetCC = len(ConcordantCluster)
            if len(PartialAlignCluster) != offsetPA and BD[itdisstart]["RefPos"] > PartialAlignCluster[-1]["RefPos"] + PartialAlignCluster[-1]["MatchRef"] + ReadLen:
                offsetPA = len(PartialAlignCluster)
            curStartPos = BD[itdisstart]["RefPos"]
            ittmp = None
            if len(ConcordantCluster) != offsetCC and len(PartialAlignCluster) != offsetPA:
                ittmp = ConcordantCluster[offsetCC] if _less(ConcordantCluster[offsetCC], PartialAlignCluster[offsetPA]) else PartialAlignCluster[offsetPA]
            elif len(ConcordantCluster) != offsetCC:
                ittmp = ConcordantCluster[offsetCC]
            elif len(PartialAlignCluster) != offsetPA:
                ittmp = PartialAlignCluster[offsetPA]
            if (len(ConcordantCluster) != offsetCC or len(PartialAlignCluster) != offsetPA) and (ittmp["RefID"] < BD[itdisstart]["RefID"] or (ittmp["RefID"] == BD[itdisstart]["RefID"] and ittmp["RefPos"] < BD[itdisstart]["RefPos"])):
                curStartPos = ittmp["RefPos"]
            curStartPos = curStartPos if curStartPos > markedNodeStart else markedNodeStart
            while len(ConcordRest) != 0 and (ConcordRest[0][2]["RefID"] < BD[itdisstart]["RefID"] or (ConcordRest[0][2]["RefID"] == BD[itdisstart]["RefID"] and ConcordRest[0][2]["RefPos"] < BD[itdisstart]["RefPos"] - ReadLen)):
                heapq.heappop(ConcordRest)
            while itpartstart != nPA and (PartAlignPos[itpartstart][0] < BD[itdisstart]["RefID"] or (PartAlignPos[itpartstart][0] == BD[itdisstart]["RefID"] and PartAlignPos[itpartstart][1] + ReadLen < BD[itdisstart]["RefPos"])):
                itpartstart += 1
            itpartend = itpartstart
            while itpartend != nPA and PartAlignPos[itpartend][0] == BD[itdisstart]["RefID"] and PartAlignPos[itpartend][1] < nextdisrightmost + ReadLen:
                itpartend += 1
            # :395
            while itdisstart != itdisend:
                ds = BD[itdisstart]
                if itdisstart != 0 and ds["RefID"] != BD[itdisstart - 1]["RefID"] and len(ConcordantCluster) == offsetCC and len(PartialAlignCluster) == offsetPA:
                    curStartPos = ds["RefPos"]
                isClusternSplit = False
                MarginPositions = []
                itdiscurrent = itdisstart
                while itdiscurrent != itdisend:
                    MarginPositions.append(BD[itdiscurrent]["RefPos"])
                    MarginPositions.append(BD[itdiscurrent]["RefPos"] + BD[itdiscurrent]["MatchRef"])
                    curEndPos = curEndPos if curEndPos > MarginPositions[-1] else MarginPositions[-1]
                    if itdiscurrent + 1 != itdisend:
                        if BD[itdiscurrent + 1]["RefPos"] > BD[itdiscurrent]["RefPos"] + BD[itdiscurrent]["MatchRef"]:
                            break
                    itdiscurrent += 1
                disStartPos = max(curStartPos, ds["RefPos"])
                disEndPos = curEndPos
                disCount = itdiscurrent - itdisstart
                if itdiscurrent != itdisend:
                    itdiscurrent += 1
                    while itdiscurrent != itdisend and BD[itdiscurrent]["RefPos"] < curEndPos + thresh:
                        MarginPositions.append(BD[itdiscurrent]["RefPos"])
                        MarginPositions.append(BD[itdiscurrent]["RefPos"] + BD[itdiscurrent]["MatchRef"])
                        itdiscurrent += 1
                itpartcurrent = itpartstart
                while itpartcurrent != itpartend and PartAlignPos[itpartcurrent][1] < curEndPos + thresh:
                    MarginPositions.append(PartAlignPos[itpartcurrent][1])
                    itpartcurrent += 1
                for i in range(offsetPA, len(PartialAlignCluster)):
                    it = PartialAlignCluster[i]
                    if it["RefID"] == ds["RefID"] and it["ReadPos"] > 15 and it["RefPos"] > MarginPositions[0] - thresh and it["RefPos"] < curEndPos + thresh:
                        if it["IsReverse"] and it["RefPos"] + it["MatchRef"] > MarginPositions[0] - thresh and it["RefPos"] + it["MatchRef"] < curEndPos + thresh:
                            MarginPositions.append(it["RefPos"] + it["MatchRef"])
                        elif not it["IsReverse"] and it["RefPos"] > MarginPositions[0] - thresh and it["RefPos"] < curEndPos + thresh:
                            MarginPositions.append(it["RefPos"])
                    elif it["RefID"] == ds["RefID"]:
                        if it["IsReverse"] and it["RefPos"] > MarginPositions[0] - thresh and it["RefPos"] < curEndPos + thresh:
                            MarginPositions.append(it["RefPos"])
                        elif not it["IsReverse"] and it["RefPos"] + it["MatchRef"] > MarginPositions[0] - thresh and it["RefPos"] + it["MatchRef"] < curEndPos + thresh:
                            MarginPositions.append(it["RefPos"] + it["MatchRef"])
                MarginPositions.sort()
                lastCurser, lastSupport = -1, 0
                nMP = len(MarginPositions)
                itbreak = 0
                while itbreak != nMP:
                    B = MarginPositions[itbreak]
                    if len(vNodes) != 0 and vNodes[-1][0] == ds["RefID"] and B - vNodes[-1][1] - vNodes[-1][2] < thresh * 20:
                        itbreak += 1  # (`continue`: the for statement's increment, not the jump over equal values below)
                        continue
                    srsupport = peleftfor = perightrev = 0
                    itbreak2 = 0
                    while itbreak2 != nMP and MarginPositions[itbreak2] < B + thresh:
                        if abs(B - MarginPositions[itbreak2]) < thresh:
                            srsupport += 1
                        itbreak2 += 1
                    for k in range(itdisstart, itdisend):
                        d = BD[k]
                        if d["RefPos"] + d["MatchRef"] < B and d["RefPos"] + d["MatchRef"] > B - ReadLen and not d["IsReverse"]:
                            peleftfor += 1
                        elif d["RefPos"] > B and d["RefPos"] < B + ReadLen and d["IsReverse"]:
                            perightrev += 1
                    if srsupport > 3 or srsupport + peleftfor > 4 or srsupport + perightrev > 4:
                        coverage = 0
                        for i in range(offsetCC, len(ConcordantCluster)):
                            it = ConcordantCluster[i]
                            if it["RefID"] == ds["RefID"] and it["RefPos"] + it["MatchRef"] >= B + thresh and it["RefPos"] < B - thresh:
                                coverage += 1
                        for k in range(itdisstart, itdisend):
                            d = BD[k]
                            if d["RefID"] == ds["RefID"] and d["RefPos"] + d["MatchRef"] >= B + thresh and d["RefPos"] < B - thresh:
                                coverage += 1
                        for i in range(offsetPA, len(PartialAlignCluster)):
                            it = PartialAlignCluster[i]
                            if it["RefID"] == ds["RefID"] and it["RefPos"] + it["MatchRef"] >= B + thresh and it["RefPos"] < B - thresh:
                                coverage += 1
                        if srsupport > max(coverage - srsupport, 0) + 2:
                            for _, _, c in ConcordRest:
                                if c["RefID"] == ds["RefID"] and c["RefPos"] + c["MatchRef"] >= B + thresh and c["RefPos"] < B - thresh:
                                    coverage += 1
                        if srsupport > max(coverage - srsupport, 0) + 2:
                            if lastCurser == -1 and B - curStartPos < thresh * 20:
                                markedNodeStart, markedNodeChr = curStartPos, ds["RefID"]
                            elif (lastCurser == -1 or B - lastCurser < thresh * 20) and max(srsupport + peleftfor, srsupport + perightrev) > lastSupport:
                                lastCurser, lastSupport = B, max(srsupport + peleftfor, srsupport + perightrev)
                            elif B - lastCurser >= thresh * 20:
                                isClusternSplit = True
                                if ds["RefPos"] - curStartPos > thresh * 20 and lastCurser - ds["RefPos"] > thresh * 20:
                                    vNodes.append([ds["RefID"], curStartPos, ds["RefPos"] - curStartPos])
                                    curStartPos = ds["RefPos"]
                                vNodes.append([ds["RefID"], curStartPos, lastCurser - curStartPos])
                                curStartPos = curEndPos = lastCurser
                                markedNodeStart, markedNodeChr = lastCurser, ds["RefID"]
                                lastCurser = B
                    itbreaknext = itbreak
                    while itbreaknext != nMP and MarginPositions[itbreaknext] == B:
                        itbreaknext += 1
                    if itbreaknext != nMP:
                        itbreak = itbreaknext  # (itbreak = itbreaknext; itbreak--; then the for statement's itbreak++)
                    else:
                        break
                if lastCurser != -1 and (not isClusternSplit or vNodes[-1][1] + vNodes[-1][2] != lastCurser):
                    isClusternSplit = True
                    if ds["RefPos"] - curStartPos > thresh * 20 and lastCurser - ds["RefPos"] > thresh * 20:
                        vNodes.append([ds["RefID"], curStartPos, ds["RefPos"] - curStartPos])
                        curStartPos = ds["RefPos"]
                    vNodes.append([ds["RefID"], curStartPos, lastCurser - curStartPos])
                    curStartPos = curEndPos = lastCurser
                    markedNodeStart, markedNodeChr = lastCurser, ds["RefID"]
                if disStartPos != -1 and not isClusternSplit and disCount > min(5.0, 4.0 * (disEndPos - disStartPos) / ReadLen):
                    if len(vNodes) != 0 and vNodes[-1][0] == BD[itdisend - 1]["RefID"] and disEndPos - vNodes[-1][1] - vNodes[-1][2] < thresh * 20:
                        vNodes[-1][2] += disEndPos - vNodes[-1][1] - vNodes[-1][2]
                    else:
                        vNodes.append([BD[itdisend - 1]["RefID"], disStartPos, disEndPos - disStartPos])
                    curStartPos = curEndPos = disEndPos
                    markedNodeStart, markedNodeChr = disEndPos, ds["RefID"]
                while len(ConcordantCluster) != offsetCC and ConcordantCluster[offsetCC]["RefID"] < ds["RefID"]:
                    offsetCC += 1
                while len(PartialAlignCluster) != offsetPA and PartialAlignCluster[offsetPA]["RefID"] < ds["RefID"]:
                    offsetPA += 1
                itdiscurrent = itdisstart
                while itdiscurrent != itdisend and BD[itdiscurrent]["RefPos"] + BD[itdiscurrent]["MatchRef"] <= curEndPos:
                    itdiscurrent += 1
                concord0pos = curStartPos
                while True:  # do { ... } while (either window has elements left), :548-579
                    flag1 = flag2 = False
                    if len(ConcordantCluster) != offsetCC:
                        c = ConcordantCluster[offsetCC]
                        flag1 = True
                        if c["RefID"] > ds["RefID"]:
                            flag1 = False
                        if itdiscurrent != nBD and c["RefID"] == BD[itdiscurrent]["RefID"] and c["RefPos"] + c["MatchRef"] + ReadLen >= BD[itdiscurrent]["RefPos"]:
                            flag1 = False
                        if len(vNodes) != 0 and (c["RefID"] > vNodes[-1][0] or (c["RefID"] == vNodes[-1][0] and c["RefPos"] >= vNodes[-1][1] + vNodes[-1][2])):
                            flag1 = False
                        if flag1:
                            concord0pos = concord0pos if concord0pos > c["RefPos"] + c["MatchRef"] else c["RefPos"] + c["MatchRef"]
                            offsetCC += 1
                    if len(PartialAlignCluster) != offsetPA:
                        c = PartialAlignCluster[offsetPA]
                        flag2 = True
                        if c["RefID"] > ds["RefID"]:
                            flag2 = False
                        if itdiscurrent != nBD and c["RefID"] == BD[itdiscurrent]["RefID"] and c["RefPos"] + c["MatchRef"] + ReadLen >= BD[itdiscurrent]["RefPos"]:
                            flag2 = False
                        if len(vNodes) != 0 and (c["RefID"] > vNodes[-1][0] or (c["RefID"] == vNodes[-1][0] and c["RefPos"] >= vNodes[-1][1] + vNodes[-1][2])):
                            flag2 = False
                        if flag2:
                            concord0pos = concord0pos if concord0pos > c["RefPos"] + c["MatchRef"] else c["RefPos"] + c["MatchRef"]
                            offsetPA += 1
                    if not flag1 and not flag2:
                        break
                    if not (len(ConcordantCluster) != offsetCC or len(PartialAlignCluster) != offsetPA):
                        break
                while True:  # :582-612
                    if (markedNodeStart != -1 and (record_RefID > markedNodeChr or record_Position > concord0pos + ReadLen)
                            and (len(ConcordantCluster) == offsetCC or ConcordantCluster[offsetCC]["RefID"] != markedNodeChr or ConcordantCluster[offsetCC]["RefPos"] > concord0pos + ReadLen)
                            and (len(PartialAlignCluster) == offsetPA or PartialAlignCluster[offsetPA]["RefID"] != markedNodeChr or PartialAlignCluster[offsetPA]["RefPos"] > concord0pos)):
                        if concord0pos > markedNodeStart and concord0pos < markedNodeStart + thresh * 20 and len(vNodes) != 0 and vNodes[-1][0] == markedNodeChr:
                            vNodes[-1][2] += concord0pos - vNodes[-1][1] - vNodes[-1][2]
                        elif concord0pos > markedNodeStart:
                            vNodes.append([markedNodeChr, markedNodeStart, concord0pos - markedNodeStart])
                        curStartPos = concord0pos
                        markedNodeChr = markedNodeStart = -1
                        break
                    flag1 = flag2 = False
                    if len(ConcordantCluster) != offsetCC:
                        c = ConcordantCluster[offsetCC]
                        if itdiscurrent == nBD or c["RefID"] < BD[itdiscurrent]["RefID"] or (c["RefID"] == BD[itdiscurrent]["RefID"] and c["RefPos"] + c["MatchRef"] + ReadLen < BD[itdiscurrent]["RefPos"]):
                            flag1 = True
                        if flag1:
                            concord0pos = concord0pos if concord0pos > c["RefPos"] + c["MatchRef"] else c["RefPos"] + c["MatchRef"]
                            offsetCC += 1
                    if len(PartialAlignCluster) != offsetPA:
                        c = PartialAlignCluster[offsetPA]
                        if itdiscurrent == nBD or c["RefID"] < BD[itdiscurrent]["RefID"] or (c["RefID"] == BD[itdiscurrent]["RefID"] and c["RefPos"] + c["MatchRef"] + ReadLen < BD[itdiscurrent]["RefPos"]):
                            flag2 = True
                        if flag2:
                            concord0pos = concord0pos if concord0pos > c["RefPos"] + c["MatchRef"] else c["RefPos"] + c["MatchRef"]
                            offsetPA += 1
                    if not flag1 and not flag2:
                        break
                    if not (len(ConcordantCluster) != offsetCC or len(PartialAlignCluster) != offsetPA):
                        break
                itdisstart = itdiscurrent
            if itdisend - itdisstart <= 0:
                seek_cluster()  # (with itdisstart == cend(): the sentinel's 0 + 0, ledger B21)
        # :616-636
        currightmost = disrightmost if (disChr > otherChr or (disChr == otherChr and disrightmost > otherrightmost)) else otherrightmost
        curChr = disChr if disChr > otherChr else otherChr
        ds = D(itdisstart)  # ledger B21
        is0coverage = (record_RefID != curChr or record_Position > currightmost + ReadLen) and (curChr < ds["RefID"] or (curChr == ds["RefID"] and currightmost + ReadLen < ds["RefPos"]))
        if is0coverage and markedNodeStart != -1:
            if curChr == markedNodeChr and currightmost > markedNodeStart and currightmost - markedNodeStart < thresh * 20 and len(vNodes) > 0 and markedNodeStart == vNodes[-1][1] + vNodes[-1][2]:
                vNodes[-1][2] += currightmost - markedNodeStart
            elif curChr == markedNodeChr and currightmost > markedNodeStart and currightmost - markedNodeStart >= thresh * 20:
                vNodes.append([markedNodeChr, markedNodeStart, currightmost - markedNodeStart])
            markedNodeStart = markedNodeChr = -1
        if is0coverage and (curChr != ds["RefID"] or currightmost + ReadLen < ds["RefPos"]):
            offsetCC, offsetPA = len(ConcordantCluster), len(PartialAlignCluster)
        else:
            while len(ConcordantCluster) > offsetCC and ConcordantCluster[offsetCC]["RefID"] != record_RefID:
                offsetCC += 1
            while len(ConcordantCluster) > offsetCC and (ConcordantCluster[offsetCC]["RefID"] < ds["RefID"] or (len(vNodes) != 0 and ConcordantCluster[offsetCC]["RefID"] == vNodes[-1][0] and ConcordantCluster[offsetCC]["RefPos"] < vNodes[-1][1] + vNodes[-1][2])):
                offsetCC += 1
            while len(PartialAlignCluster) > offsetPA and PartialAlignCluster[offsetPA]["RefID"] != record_RefID:
                offsetPA += 1
            while len(PartialAlignCluster) > offsetPA and (PartialAlignCluster[offsetPA]["RefID"] < ds["RefID"] or (len(vNodes) != 0 and PartialAlignCluster[offsetPA]["RefID"] == vNodes[-1][0] and PartialAlignCluster[offsetPA]["RefPos"] < vNodes[-1][1] + vNodes[-1][2])):
                offsetPA += 1
        # :650-700
        recordconcordant = recordpartalign = False
        if IsMapped and IsMateMapped and mref[r] != -1 and IsReverseStrand and not IsMateReverseStrand and refid[r] == mref[r] and pos[r] >= mpos[r] and pos[r] - mpos[r] <= 750000 and IsProperPair:
            recordconcordant = True
        elif IsMapped and IsMateMapped and mref[r] != -1 and not IsReverseStrand and IsMateReverseStrand and refid[r] == mref[r] and mpos[r] >= pos[r] and mpos[r] - pos[r] <= 750000 and IsProperPair:
            recordconcordant = True
        if recordconcordant and len(rF) + len(rS) > 0:
            if otherChr == record_RefID and IsFirstMate:
                otherrightmost = otherrightmost if otherrightmost > rF[0]["RefPos"] + rF[0]["MatchRef"] else rF[0]["RefPos"] + rF[0]["MatchRef"]
            elif otherChr == record_RefID and IsSecondMate:
                otherrightmost = otherrightmost if otherrightmost > rS[0]["RefPos"] + rS[0]["MatchRef"] else rS[0]["RefPos"] + rS[0]["MatchRef"]
            elif IsFirstMate:
                otherrightmost, otherChr = rF[0]["RefPos"] + rF[0]["MatchRef"], record_RefID
            elif IsSecondMate:
                otherrightmost, otherChr = rS[0]["RefPos"] + rS[0]["MatchRef"], record_RefID
            if IsFirstMate and tF[0]["ReadPos"] > 15 and not low:
                PartialAlignCluster.append(rF[0])
                recordpartalign = True
            elif IsFirstMate and totlen[r] - tF[-1]["ReadPos"] - tF[-1]["MatchRead"] > 15 and not low:
                PartialAlignCluster.append(rF[0])
                recordpartalign = True
            if IsSecondMate and tS[0]["ReadPos"] > 15 and not low:
                PartialAlignCluster.append(rS[0])
                recordpartalign = True
            elif IsSecondMate and totlen[r] - tS[-1]["ReadPos"] - tS[-1]["MatchRead"] > 15 and not low:
                PartialAlignCluster.append(rS[0])
                recordpartalign = True
            if not recordpartalign:
                ConcordantCluster.append(rF[0] if IsFirstMate else rS[0])
            if IsFirstMate and len(rF) > 1:
                for i in range(1, len(rF)):
                    if itdisstart != nBD and rF[i]["RefPos"] >= BD[itdisstart]["RefPos"] - ReadLen:
                        heapq.heappush(ConcordRest, ((rF[i]["RefID"], rF[i]["RefPos"]), heapseq, rF[i]))
                        heapseq += 1
            if IsSecondMate and len(rS) > 1:
                for i in range(1, len(rS)):
                    if itdisstart != nBD and rS[i]["RefPos"] >= BD[itdisstart]["RefPos"] - ReadLen:
                        heapq.heappush(ConcordRest, ((rS[i]["RefID"], rS[i]["RefPos"]), heapseq, rS[i]))
                        heapseq += 1
    seeds = [tuple(n) for n in vNodes]
    # NormalizeSeedNodes, :19-38
    if len(vNodes) >= 2:
        vNodes.sort()
        normalized = []
        for node in vNodes:
            if len(normalized) == 0 or normalized[-1][0] != node[0] or normalized[-1][1] + normalized[-1][2] <= node[1]:
                normalized.append(list(node))
            else:
                normalized[-1][2] = max(normalized[-1][1] + normalized[-1][2], node[1] + node[2]) - normalized[-1][1]
        vNodes = normalized
    # :718-761, the nodes expanded to cover every reference sequence
    tmpNodes = []
    for i in range(len(vNodes)):
        if len(tmpNodes) == 0 or tmpNodes[-1][0] != vNodes[i][0]:
            if len(tmpNodes) != 0 and tmpNodes[-1][1] + tmpNodes[-1][2] != ref_len[tmpNodes[-1][0]]:
                tmpNodes.append([tmpNodes[-1][0], tmpNodes[-1][1] + tmpNodes[-1][2], ref_len[tmpNodes[-1][0]] - tmpNodes[-1][1] - tmpNodes[-1][2]])
            chrstart = 0 if len(tmpNodes) == 0 else tmpNodes[-1][0] + 1
            while chrstart != vNodes[i][0]:
                tmpNodes.append([chrstart, 0, ref_len[chrstart]])
                chrstart += 1
            if vNodes[i][1] != 0:
                if vNodes[i][1] > 100:
                    tmpNodes.append([vNodes[i][0], 0, vNodes[i][1]])
                else:
                    vNodes[i][2] += vNodes[i][1]
                    vNodes[i][1] = 0
                    tmpNodes.append(vNodes[i])
                    continue
        if tmpNodes[-1][1] + tmpNodes[-1][2] < vNodes[i][1]:
            if vNodes[i][1] - tmpNodes[-1][1] - tmpNodes[-1][2] > 100:
                tmpNodes.append([vNodes[i][0], tmpNodes[-1][1] + tmpNodes[-1][2], vNodes[i][1] - tmpNodes[-1][1] - tmpNodes[-1][2]])
                tmpNodes.append(vNodes[i])
            else:
                vNodes[i][2] += vNodes[i][1] - tmpNodes[-1][1] - tmpNodes[-1][2]
                vNodes[i][1] = tmpNodes[-1][1] + tmpNodes[-1][2]
                tmpNodes.append(vNodes[i])
        else:
            tmpNodes.append(vNodes[i])
    if len(tmpNodes) != 0 and tmpNodes[-1][1] + tmpNodes[-1][2] != ref_len[tmpNodes[-1][0]]:
        tmpNodes.append([tmpNodes[-1][0], tmpNodes[-1][1] + tmpNodes[-1][2], ref_len[tmpNodes[-1][0]] - tmpNodes[-1][1] - tmpNodes[-1][2]])
    for chrstart in range(tmpNodes[-1][0] + 1, len(ref_len)):
        tmpNodes.append([chrstart, 0, ref_len[chrstart]])
    return seeds, [tuple(n) for n in tmpNodes], kept, n_break


CASES = [("C1", (), ()), ("T2", (), ()), ("C2", (), ()), ("C2", ("--support", "2,6"), ("-w", "1", "-a", "50")), ("C2", ("--interleave", "6"), ())]
# (a C2 case is a minute of Python loops: the CPU suite keeps three inputs, the GPU suite four, all five are covered; every case passed in both when the test was written)
CPU_CASES = [CASES[0], CASES[1], CASES[2]]
GPU_CASES = [CASES[0], CASES[1], CASES[3], CASES[4]]


def _inputs(built, synth, tmp_path, cfg, gen, flags):
    import squid_amd

    pre = synth(cfg, *gen)
    sv_path, dump = ou.run_oracle(built, pre, tmp_path, *flags)
    chim = ll._read_chimrecord(dump / "chimrecord.txt")
    rec = ll._records_from_bam(f"{pre}.bam", ll._chim_names(dump))
    _, read_len = ll._build_chimeric_literal(f"{pre}.chim.bam")
    _, ref_len = squid_amd.read_header(f"{pre}.bam")
    return pre, dump, chim, rec, read_len, ref_len


def _dumped_nodes(path):
    out = []
    for line in open(path):
        if not line.startswith("#"):
            f = line.split("\t")
            out.append((int(f[0]), int(f[1]), int(f[2])))
    return out


@pytest.mark.parametrize("cfg,gen,flags", CPU_CASES)
def test_oracle_nodes_against_the_literal_automaton(built, synth, tmp_path, cfg, gen, flags):
    pre, dump, chim, rec, read_len, ref_len = _inputs(built, synth, tmp_path, cfg, gen, flags)
    seeds, nodes, kept, n_break = _build_node_star_literal(rec, chim, read_len, ref_len, min_mapq=255)  # (STAR mode: Min_MapQual 255, ledger B2)
    assert len(seeds) >= 3
    assert seeds == _dumped_nodes(dump / "nodes_seed.txt")
    assert nodes == _dumped_nodes(dump / "nodes_build.txt")
    stats = dict(line.split("\t") for line in (dump / "order_stats.txt").read_text().splitlines())
    assert n_break == int(stats["break_record"]) or (n_break == -1 and kept == int(stats["kept_records"]))


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,gen,flags", GPU_CASES)
def test_hip_path_nodes_against_the_literal_automaton(built, synth, tmp_path, cfg, gen, flags):
    """the library's node table (cluster triggers, zero-coverage candidates and window summaries from the kernels, the event-driven replay on the
    host, tiling) against the same literal reading"""
    import squid_amd

    pre, dump, chim, rec, read_len, ref_len = _inputs(built, synth, tmp_path, cfg, gen, flags)
    _, nodes, _, _ = _build_node_star_literal(rec, chim, read_len, ref_len, min_mapq=255)
    kw = {}
    if flags:
        kw = {"min_edge_weight": 1, "max_allowed_degree": 50}
    with squid_amd.Context(**kw) as ctx:
        ctx.load(f"{pre}.bam", f"{pre}.chim.bam")
        ctx.build_graph()
        got = [(int(n[0]), int(n[1]), int(n[2])) for n in ctx.graph(1)["nodes"]]
    assert got == nodes
