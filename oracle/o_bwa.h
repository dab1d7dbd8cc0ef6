// ORACLE (test infrastructure only) -- never linked, imported or executed by the product path.
// PARITY UNPINNED: see o_bam.h.  CPU restatement of the `--bwa` half of squid v1.5: the single-file
// node builder and edge generator that replace BuildNode_STAR / RawEdgesChim / RawEdgesOther when the aligner wrote split reads as
// supplementary alignments into ONE BAM file.
//   BuildNode_BWA   src/SegmentGraph.cpp:833-1205
//   RawEdges        src/SegmentGraph.cpp:1698-1930  (also rebuilds Chimrecord from the partially aligned reads, :1883-1926)
// Everything downstream (BuildEdges' sort + sum, the filters, compression, components, ordering, ExactBreakpoint,
// ExactBPConcordantSupport, WriteBEDPE) is the code of o_graph.h / o_order.h: the reference shares it between the two modes.
//
// Quirks kept on purpose (ledger):
//   W1  the three sliding windows are std::vectors with a moving front offset; when one reaches its CAPACITY (65536 after the
//       reserve of :844-848, doubled only when a compaction does not free anything) it is compacted by a test that also drops
//       elements from its middle (:1087-1112) -- behaviour depends on the capacity, which is therefore tracked explicitly here;
//   W2  `DiscordantCluster.front()` / `[0]` is read where `[offsetDiscordantCluster]` is meant (:929,953,960,986);
//   W3  the last Qname group of PartialAlign is never flushed (:1886-1925 has no epilogue);
//   W4  LocateRead for the rebuilt fragments starts from whatever hint the BAM loop left (:1893), and the fragments are stored
//       BEFORE LocateRead trims them (:1892);
//   W5  second mates reach the loop only when multi-aligned (:1723-1726); their pair edge carries weight -1 and is added only when
//       the first mate of the same name added a discordant pair edge (:1851-1879);
//   W6  Reads is swept with a cursor that never goes back: a block that starts in front of the current node is passed over
//       (:1180-1200).
#pragma once
#include "o_graph.h"

namespace oracle {

// NormalizeSeedNodes + the seed asserts + whole-genome tiling (:1120-1175; the same statements as :706-761 of the STAR builder)
inline void TileGenomeBWA(std::vector<Node_t>& vNodes, const std::vector<int>& RefLength) {
    if (vNodes.size() >= 2) {
        std::sort(vNodes.begin(), vNodes.end());
        std::vector<Node_t> normalized;
        int mergedCount = 0;
        for (const Node_t& node : vNodes) {
            if (normalized.size() == 0 || normalized.back().Chr != node.Chr || normalized.back().Position + normalized.back().Length <= node.Position) normalized.push_back(node);
            else {
                int mergedEnd = std::max(normalized.back().Position + normalized.back().Length, node.Position + node.Length);
                normalized.back().Length = mergedEnd - normalized.back().Position;
                mergedCount++;
            }
        }
        if (mergedCount > 0) std::cerr << "[SQUID] normalized " << mergedCount << " overlapping seed nodes in BuildNode_BWA.\n";
        vNodes.swap(normalized);
    }
    for (size_t i = 0; i < vNodes.size(); i++) {
        bool ok = vNodes[i].Length > 0 && vNodes[i].Position + vNodes[i].Length <= RefLength[vNodes[i].Chr];
        if (i + 1 < vNodes.size()) ok = ok && ((vNodes[i].Chr != vNodes[i + 1].Chr) || vNodes[i].Position + vNodes[i].Length <= vNodes[i + 1].Position);
        if (!ok) { std::cerr << "oracle: seed-node assertion of SegmentGraph.cpp:1122-1126 fails (reference aborts)\n"; std::exit(4); }
    }
    std::vector<Node_t> T;
    auto endof = [](const Node_t& n) { return n.Position + n.Length; };
    for (size_t i = 0; i < vNodes.size(); i++) {
        Node_t& v = vNodes[i];
        if (T.size() == 0 || T.back().Chr != v.Chr) {
            if (T.size() != 0 && endof(T.back()) != RefLength[T.back().Chr]) T.push_back(Node_t(T.back().Chr, endof(T.back()), RefLength[T.back().Chr] - endof(T.back())));
            for (int chrstart = (T.size() == 0) ? 0 : (T.back().Chr + 1); chrstart != v.Chr; chrstart++) T.push_back(Node_t(chrstart, 0, RefLength[chrstart]));
            if (v.Position != 0) {
                if (v.Position > 100) T.push_back(Node_t(v.Chr, 0, v.Position));
                else { v.Length += v.Position; v.Position = 0; T.push_back(v); continue; }
            }
        }
        if (T.size() != 0 && endof(T.back()) < v.Position) {  // (T empty + a seed at position 0 of reference 0: the reference reads back() of an empty vector; no gap)
            const int gap = v.Position - endof(T.back());
            if (gap > 100) { T.push_back(Node_t(v.Chr, endof(T.back()), gap)); T.push_back(v); }
            else { v.Length += gap; v.Position = endof(T.back()); T.push_back(v); }
        } else T.push_back(v);
    }
    if (T.size() != 0 && endof(T.back()) != RefLength[T.back().Chr]) T.push_back(Node_t(T.back().Chr, endof(T.back()), RefLength[T.back().Chr] - endof(T.back())));
    // (no seed at all: the reference evaluates tmpNodes.back() of an empty vector; every reference becomes one node, as in the STAR restatement)
    for (int chrstart = T.empty() ? 0 : T.back().Chr + 1; chrstart < (int)RefLength.size(); chrstart++) T.push_back(Node_t(chrstart, 0, RefLength[chrstart]));
    vNodes = T;
}

// the record is "concordant" for the node builder (:1037-1040)
inline bool BwaRecordConcordant(const BamAlignment& r) {
    if (!(r.IsMapped() && r.IsMateMapped() && r.MateRefID != -1 && r.RefID == r.MateRefID && r.IsProperPair())) return false;
    if (r.IsReverseStrand() && !r.IsMateReverseStrand() && r.Position >= r.MatePosition && r.Position - r.MatePosition <= 750000) return true;
    if (!r.IsReverseStrand() && r.IsMateReverseStrand() && r.MatePosition >= r.Position && r.MatePosition - r.Position <= 750000) return true;
    return false;
}
// which end of a read is clipped by more than 15 bases (:1050-1065, :1730-1737): 0 none, 1 first read, 2 second mate (first wins per side)
inline bool BwaPartial(const std::vector<SingleBamRec_t>& R, int TotalLen, bool LowPhred) {
    if (R.size() != 0 && R.front().ReadPos > 15 && !LowPhred) return true;
    if (R.size() != 0 && TotalLen - R.back().ReadPos - R.back().MatchRead > 15 && !LowPhred) return true;
    return false;
}

// =====================================================================================================
// BuildNode_BWA (src/SegmentGraph.cpp:833-1205)
// =====================================================================================================
inline void SegmentGraph_t::BuildNode_BWA(const std::vector<int>& RefLength, const std::string& bamfile, uint16_t& ReadLen) {
    BamReader bamreader;
    bamreader.Open(bamfile);
    std::vector<std::pair<int, pii>> Reads;
    int countreadlen = 0;
    const int thresh = 3;
    int prev0CovPos = 0;
    int markedNodeStart = -1, markedNodeChr = -1;
    int disrightmost = 0, otherrightmost = 0;
    // the three windows: vector + front offset + tracked capacity (W1)
    struct Window { std::vector<SingleBamRec_t> v; int off = 0; size_t cap = 65536; bool empty() const { return (int)v.size() == off; } const SingleBamRec_t& front() const { return v[off]; } };
    Window CC, DC, PC;  // ConcordantCluster, DiscordantCluster, PartialAlignCluster
    if (bamreader.IsOpen()) {
        BamAlignment record;
        while (bamreader.GetNextAlignment(record)) {
            if (countreadlen < 5) {  // :857-864
                int tmpreadlen = 0;
                for (const CigarOp& c : record.CigarData)
                    if (c.Type == 'M' || c.Type == 'S' || c.Type == 'H' || c.Type == 'I' || c.Type == '=' || c.Type == 'X') tmpreadlen += c.Length;
                ReadLen = (ReadLen < tmpreadlen) ? (uint16_t)tmpreadlen : ReadLen;
                countreadlen++;
            }
            bool XAtag = record.HasTag("XA"), IHtag = record.HasTag("IH");
            int IHtagvalue = 0;
            if (IHtag) record.GetTagInt("IH", IHtagvalue);
            if (XAtag || IHtagvalue > 1 || record.MapQuality == 0 || record.IsDuplicate() || !record.IsMapped() || record.RefID == -1) continue;
            if ((!DC.empty() && record.RefID != DC.front().RefID) || (!CC.empty() && record.RefID != CC.front().RefID) || (!PC.empty() && record.RefID != PC.front().RefID)) otherrightmost = 0;
            ReadRec_t readrec(record, P);
            if (readrec.FirstRead.size() == 0 && readrec.SecondMate.size() == 0) continue;
            for (const SingleBamRec_t& b : readrec.FirstRead) Reads.push_back(std::make_pair(b.RefID, pii(b.RefPos, b.MatchRef)));
            for (const SingleBamRec_t& b : readrec.SecondMate) Reads.push_back(std::make_pair(b.RefID, pii(b.RefPos, b.MatchRef)));
            if (CC.empty() && PC.empty() && DC.empty()) prev0CovPos = record.Position;
            // ---- segment boundaries once the record no longer overlaps the discordant window (:888-998)
            if (!DC.empty() && (DC.v.back().RefID != record.RefID || disrightmost + ReadLen < record.Position)) {
                int curEndPos = 0, curStartPos = (prev0CovPos > markedNodeStart) ? prev0CovPos : markedNodeStart;
                int disStartPos = -1, disEndPos = -1, disCount = -1;
                bool isClusternSplit = false;
                auto emit = [&](int chr, int start, int end) {  // push a node and move the marks behind it
                    vNodes.push_back(Node_t(chr, start, end - start));
                    curStartPos = end; curEndPos = end;
                    markedNodeStart = end; markedNodeChr = chr;
                };
                while (!DC.empty()) {
                    if (disStartPos != -1 && !isClusternSplit && disCount > std::min(5.0, 4.0 * (disEndPos - disStartPos) / ReadLen)) emit(DC.front().RefID, disStartPos, disEndPos);
                    isClusternSplit = false;
                    std::vector<int> MarginPositions;
                    int i;
                    for (i = DC.off; i < (int)DC.v.size(); i++) {  // the run of overlapping discordant blocks
                        const SingleBamRec_t& it = DC.v[i];
                        MarginPositions.push_back(it.RefPos); MarginPositions.push_back(it.RefPos + it.MatchRef);
                        curEndPos = std::max(curEndPos, MarginPositions.back());
                        if (i + 1 < (int)DC.v.size() && DC.v[i + 1].RefPos > it.RefPos + it.MatchRef) break;
                    }
                    disStartPos = std::max(curStartPos, DC.front().RefPos);
                    disEndPos = curEndPos;
                    disCount = i - DC.off;
                    for (i++; i < (int)DC.v.size() && DC.v[i].RefPos < curEndPos + thresh; i++) { MarginPositions.push_back(DC.v[i].RefPos); MarginPositions.push_back(DC.v[i].RefPos + DC.v[i].MatchRef); }
                    for (i = PC.off; i != (int)PC.v.size(); i++) {
                        const SingleBamRec_t& it = PC.v[i];
                        if (it.RefID == DC.front().RefID && it.ReadPos > 15 && it.RefPos > MarginPositions.front() - thresh && it.RefPos < curEndPos + thresh)
                            MarginPositions.push_back(it.IsReverse ? (it.RefPos + it.MatchRef) : it.RefPos);
                        else if (it.RefID == DC.front().RefID && it.RefPos + it.MatchRef > MarginPositions.front() - thresh && it.RefPos + it.MatchRef < curEndPos + thresh)
                            MarginPositions.push_back(it.IsReverse ? it.RefPos : (it.RefPos + it.MatchRef));
                    }
                    std::sort(MarginPositions.begin(), MarginPositions.end());
                    int lastCurser = -1, lastSupport = 0;
                    for (size_t ib = 0; ib < MarginPositions.size();) {
                        const int brk = MarginPositions[ib];
                        size_t inext = ib;
                        while (inext < MarginPositions.size() && MarginPositions[inext] == brk) inext++;  // (:968-973: on to the next distinct position, or out)
                        if (vNodes.size() != 0 && vNodes.back().Chr == DC.v.front().RefID && brk - vNodes.back().Position - vNodes.back().Length < thresh * 20) { ib++; continue; }  // (W2; `continue` = plain ++)
                        int srsupport = 0, peleftfor = 0, perightrev = 0;
                        for (size_t i2 = 0; i2 < MarginPositions.size() && MarginPositions[i2] < brk + thresh; i2++) if (std::abs(brk - MarginPositions[i2]) < thresh) srsupport++;
                        for (i = DC.off; i < (int)DC.v.size(); i++) {
                            const SingleBamRec_t& it = DC.v[i];
                            if (it.RefPos + it.MatchRef < brk && it.RefPos + it.MatchRef > brk - ReadLen && !it.IsReverse) peleftfor++;
                            else if (it.RefPos > brk && it.RefPos < brk + ReadLen && it.IsReverse) perightrev++;
                        }
                        bool split_here = false;
                        if (srsupport > 3 || srsupport + peleftfor > 4 || srsupport + perightrev > 4) {
                            int coverage = 0;
                            for (i = CC.off; i < (int)CC.v.size(); i++) if (CC.v[i].RefPos + CC.v[i].MatchRef >= brk + thresh && CC.v[i].RefPos < brk - thresh) coverage++;
                            if (srsupport > std::max(coverage - srsupport, 0) + 2) {
                                const int both = std::max(srsupport + peleftfor, srsupport + perightrev);
                                if (lastCurser == -1 && brk - curStartPos < thresh * 20) { markedNodeStart = curStartPos; markedNodeChr = DC.v.front().RefID; }
                                else if ((lastCurser == -1 || brk - lastCurser < thresh * 20) && both > lastSupport) { lastCurser = brk; lastSupport = both; }
                                else if (brk - lastCurser >= thresh * 20) {
                                    isClusternSplit = true;
                                    emit(DC.v.front().RefID, curStartPos, lastCurser);
                                    split_here = true;
                                }
                            }
                        }
                        if (split_here) break;
                        if (inext >= MarginPositions.size()) break;
                        ib = inext;
                    }
                    if (lastCurser != -1 && !isClusternSplit) { isClusternSplit = true; emit(DC.front().RefID, curStartPos, lastCurser); }
                    while (!DC.empty() && DC.front().RefPos + DC.front().MatchRef <= curEndPos) DC.off++;
                }
                if (disStartPos != -1 && !isClusternSplit && disCount > std::min(5.0, 4.0 * (disEndPos - disStartPos) / ReadLen)) emit(DC.v[0].RefID, disStartPos, disEndPos);  // (W2: [0])
                if (DC.empty()) { DC.v.clear(); DC.off = 0; }
                auto prune = [&](Window& W) { while (!W.empty() && (W.front().RefID != record.RefID || W.front().RefPos + W.front().MatchRef + ReadLen < record.Position)) W.off++; };
                prune(CC); prune(PC);
            }
            // ---- zero-coverage position (:1000-1026)
            bool is0coverage = true;
            int currightmost = (disrightmost > otherrightmost) ? disrightmost : otherrightmost, curChr = 0;
            auto lastchr = [&](const Window& W) { for (int i = (int)W.v.size() - 1; i >= W.off && (int)W.v.size() - i < 5; i--) curChr = W.v[i].RefID; };
            lastchr(CC); lastchr(PC); lastchr(DC);
            is0coverage = (record.RefID != curChr || record.Position > currightmost + ReadLen);
            if (is0coverage && markedNodeStart != -1) {
                if (currightmost > markedNodeStart && currightmost - markedNodeStart < thresh * 20 && vNodes.size() > 0 && markedNodeStart == vNodes.back().Position + vNodes.back().Length)
                    vNodes.back().Length += currightmost - markedNodeStart;
                else if (currightmost > markedNodeStart && currightmost - markedNodeStart >= thresh * 20)
                    vNodes.push_back(Node_t(markedNodeChr, markedNodeStart, currightmost - markedNodeStart));
                markedNodeStart = -1; markedNodeChr = -1;
            }
            if (is0coverage) prev0CovPos = record.Position;
            if (DC.empty()) {  // :1028-1033
                auto prune = [&](Window& W) { while (!W.empty() && (W.front().RefID != record.RefID || W.front().RefPos + W.front().MatchRef + ReadLen < record.Position)) W.off++; };
                prune(CC); prune(PC);
            }
            // ---- the record joins one of the windows (:1035-1086)
            const std::vector<SingleBamRec_t>&F = readrec.FirstRead, &S = readrec.SecondMate;
            auto firstend = [&]() { return F.size() != 0 ? F.front().RefPos + F.front().MatchRef : S.front().RefPos + S.front().MatchRef; };  // (one of the two is non-empty here)
            if (BwaRecordConcordant(record)) {
                if (!CC.empty() || !PC.empty()) otherrightmost = std::max(otherrightmost, firstend());
                else otherrightmost = firstend();
                bool recordpartalign = false;
                if (BwaPartial(F, readrec.FirstTotalLen, readrec.FirstLowPhred)) { PC.v.push_back(F.front()); recordpartalign = true; }
                if (BwaPartial(S, readrec.SecondTotalLen, readrec.SecondLowPhred)) { PC.v.push_back(S.front()); recordpartalign = true; }
                if (!recordpartalign) CC.v.push_back(F.size() != 0 ? F.front() : S.front());
            } else {
                if (DC.v.size() != 0) disrightmost = std::max(disrightmost, firstend());  // (:1074: size(), not size() - offset)
                else disrightmost = firstend();
                DC.v.push_back(F.size() != 0 ? F.front() : S.front());
            }
            // ---- a window that has reached its capacity is compacted (:1087-1112, W1)
            auto compact = [&](Window& W) {
                if (W.v.size() != W.cap) return;
                int cChr = record.RefID, cStart = record.Position;
                if (!DC.empty()) cStart = std::min(cStart, DC.front().RefPos);
                std::vector<SingleBamRec_t> tmp;
                for (int i = W.off; i < (int)W.v.size(); i++) if (W.v[i].RefID == cChr && W.v[i].RefPos + W.v[i].MatchRef + ReadLen >= cStart) tmp.push_back(W.v[i]);
                W.v.swap(tmp);
                W.off = 0;
                if (W.v.size() == W.cap) W.cap *= 2;
            };
            compact(CC); compact(PC);
        }
        bamreader.Close();
    }
    seedNodes = vNodes;
    TileGenomeBWA(vNodes, RefLength);
    // ---- reads per node (:1180-1204, W6)
    if (Reads.size() != 0) {
        size_t it = 0;
        for (size_t i = 0; i < vNodes.size(); i++) {
            int covcount = 0, covsumlen = 0;
            for (; it != Reads.size(); it++) {
                const std::pair<int, pii>& r = Reads[it];
                if (r.first == vNodes[i].Chr && r.second.first >= vNodes[i].Position && r.second.first + r.second.second <= vNodes[i].Position + vNodes[i].Length) { covcount++; covsumlen += r.second.second; }
                else if (r.second.first >= vNodes[i].Position + vNodes[i].Length || r.first != vNodes[i].Chr) break;
            }
            vNodes[i].Support = covcount;
            vNodes[i].AvgDepth = 1.0 * covsumlen / vNodes[i].Length;
        }
    }
}

// =====================================================================================================
// RawEdges (src/SegmentGraph.cpp:1698-1930)
// =====================================================================================================
inline void SegmentGraph_t::RawEdges(SBamrecord_t& Chimrecord, const std::string& bamfile) {
    int firstfrontindex = 0;
    const int N = (int)vNodes.size();
    BamReader bamreader;
    bamreader.Open(bamfile);
    std::vector<ReadRec_t> PartialAlign;
    std::vector<std::string> FirstDisInserted, SecondDisMulti;
    std::vector<Edge_t> SecondEdges;
    auto checked = [&](const Edge_t& e) {
        if (!(e.Ind1 >= 0 && e.Ind1 < N && e.Ind2 >= 0 && e.Ind2 < N)) { std::cerr << "oracle: edge index assertion of SegmentGraph.cpp:1760 fails (reference aborts)\n"; std::exit(4); }
        return e;
    };
    auto splitedges = [&](const std::vector<SingleBamRec_t>& R, const std::vector<int>& RN, int base) {  // :1774-1796, :1895-1917
        if (R.size() == 0) return;
        for (int k = 0; k < (int)R.size() - 1; k++) {
            int i = RN[base + k], j = RN[base + k + 1];
            if (i != j && i != -1 && j != -1) vEdges.push_back(checked(Edge_t(i, R[k].IsReverse, j, !R[k + 1].IsReverse, 1)));
        }
    };
    if (bamreader.IsOpen()) {
        BamAlignment record;
        while (bamreader.GetNextAlignment(record)) {
            bool XAtag = record.HasTag("XA"), IHtag = record.HasTag("IH");
            int IHtagvalue = 0;
            if (IHtag) record.GetTagInt("IH", IHtagvalue);
            const bool multi = XAtag || IHtagvalue > 1;
            if (record.IsDuplicate() || !record.IsMapped()) continue;
            else if ((multi || record.MapQuality == 0) && record.IsFirstMate()) continue;
            else if (!multi && !record.IsFirstMate()) continue;
            ReadRec_t readrec(record, P);
            readrec.SortbyReadPos();
            if (!multi) {  // (one push at most: only the record's own side has blocks)
                if (BwaPartial(readrec.FirstRead, readrec.FirstTotalLen, readrec.FirstLowPhred)) PartialAlign.push_back(readrec);
                if (BwaPartial(readrec.SecondMate, readrec.SecondTotalLen, readrec.SecondLowPhred)) PartialAlign.push_back(readrec);
            }
            AppendMateStub(record, readrec);
            const int nf = (int)readrec.FirstRead.size();
            if (record.IsFirstMate() && readrec.FirstRead.size() > 0 && (readrec.FirstRead.front().ReadPos <= 15 || readrec.FirstLowPhred)) {
                std::vector<int> RN = LocateRead(firstfrontindex, readrec);
                if (RN[0] != -1) firstfrontindex = RN[0];
                for (int k = 0; k < (int)RN.size(); k++)
                    if (RN[k] == -1) {
                        const SingleBamRec_t& b = k < nf ? readrec.FirstRead[k] : readrec.SecondMate[k - nf];
                        int i = HomeNode(firstfrontindex, b);
                        vEdges.push_back(checked(Edge_t(i, false, i + 1, true)));
                    }
                splitedges(readrec.FirstRead, RN, 0);
                splitedges(readrec.SecondMate, RN, nf);
                if (readrec.FirstRead.size() > 0 && readrec.SecondMate.size() > 0 && !readrec.IsSingleAnchored() && !readrec.IsEndDiscordant(true) && !readrec.IsEndDiscordant(false)) {
                    int i = RN[nf - 1], j = RN.back();
                    bool isoverlap = detail::PairOverlap(readrec, RN, i, j);
                    if (i != j && i != -1 && j != -1 && !isoverlap) {
                        vEdges.push_back(checked(Edge_t(i, readrec.FirstRead.back().IsReverse, j, readrec.SecondMate.back().IsReverse, 1)));
                        if (IsDiscordant(vEdges.back())) FirstDisInserted.push_back(readrec.Qname);
                    }
                }
            } else if (!record.IsFirstMate() && readrec.SecondMate.size() > 0) {
                readrec.SecondMate.resize(1);
                readrec.SecondMate[0].MatchRef = 15;
                readrec.SecondMate[0].MatchRead = 15;
                std::vector<int> RN = LocateRead(firstfrontindex, readrec);
                if (RN[0] != -1) firstfrontindex = RN[0];
                if (readrec.FirstRead.size() > 0 && readrec.SecondMate.size() > 0 && !readrec.IsSingleAnchored() && !readrec.IsEndDiscordant(true) && !readrec.IsEndDiscordant(false)) {
                    const int nf2 = (int)readrec.FirstRead.size();
                    int i = RN[nf2 - 1], j = RN.back();
                    bool isoverlap = false;  // (:1842-1848: only the two membership loops, not the whole of PairOverlap)
                    for (int k = 0; k < nf2; k++) if (j == RN[k]) isoverlap = true;
                    for (int k = 0; k < (int)readrec.SecondMate.size(); k++) if (i == RN[nf2 + k]) isoverlap = true;
                    if (i != j && i != -1 && j != -1 && !isoverlap) {
                        Edge_t tmp = checked(Edge_t(i, readrec.FirstRead.back().IsReverse, j, readrec.SecondMate.back().IsReverse, -1));
                        if (IsDiscordant(tmp)) { SecondDisMulti.push_back(readrec.Qname); SecondEdges.push_back(tmp); }
                    }
                }
            }
        }
        bamreader.Close();
    }
    std::sort(FirstDisInserted.begin(), FirstDisInserted.end());
    for (size_t i = 0; i < SecondDisMulti.size(); i++)
        if (std::binary_search(FirstDisInserted.begin(), FirstDisInserted.end(), SecondDisMulti[i])) vEdges.push_back(SecondEdges[i]);
    // ---- Chimrecord rebuilt from the partially aligned reads (:1883-1926)
    std::sort(PartialAlign.begin(), PartialAlign.end());
    ReadRec_t merged;
    Chimrecord.clear();
    for (const ReadRec_t& it : PartialAlign) {
        if (merged.FirstRead.size() == 0 && merged.SecondMate.size() == 0) merged = it;
        else if (merged.Qname != it.Qname) {
            merged.SortbyReadPos();
            if (merged.FirstRead.size() > 1 || merged.SecondMate.size() > 1) {
                Chimrecord.push_back(merged);
                std::vector<int> RN = LocateRead(firstfrontindex, merged);  // (W4)
                splitedges(merged.FirstRead, RN, 0);
                splitedges(merged.SecondMate, RN, (int)merged.FirstRead.size());
            }
            merged = it;
        } else {
            merged.FirstRead.insert(merged.FirstRead.end(), it.FirstRead.begin(), it.FirstRead.end());
            merged.SecondMate.insert(merged.SecondMate.end(), it.SecondMate.begin(), it.SecondMate.end());
        }
    }  // (W3: the last group stays in `merged`)
    std::sort(Chimrecord.begin(), Chimrecord.end(), ReadRec_t::FrontSmallerThan);
}

}  // namespace oracle
