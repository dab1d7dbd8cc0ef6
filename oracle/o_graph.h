// ORACLE (test infrastructure only) -- never linked, imported or executed by the product path.
// PARITY UNPINNED (see o_bam.h).  CPU restatement of the squid v1.5 segment-graph construction:
//   src/BPNode.h:26-57, src/BPEdge.h:24-77,
//   src/SegmentGraph.cpp:15-38 (MinHeapComp, NormalizeSeedNodes), :51-102 (CountTop), :104-124 (ctor pipeline),
//   :159-190 (IsDiscordant), :192-831 (BuildNode_STAR), :1207-1293 (LocateRead), :1394-1555 (RawEdgesChim),
//   :1557-1696 (RawEdgesOther), :1932-1966 (BuildEdges), :1968-2123 (FilterbyWeight), :2161-2277
//   (FilterbyInterleaving), :2394-2526 (GroupConnection/GroupSelect/FilterEdges), :2528-2604 (CompressNode),
//   :2693-2892 (FurtherCompressNode), :2894-2935 (UpdateNodeLink, DFS), :2986-3017 (ConnectedComponent,
//   Multiply/DeMultiplyDisEdges), :3019-3221 (ExactBreakpoint, ExactBPConcordantSupport).
// Bug-compatibility ledger entries are reproduced and marked "ledger Bn".
#pragma once
#include <cmath>
#include <cstdlib>
#include <limits>
#include <map>
#include <utility>

#include "o_readrec.h"

namespace oracle {

// src/BPEdge.h:24-77
struct Edge_t {
    int Ind1 = 0, Ind2 = 0;
    bool Head1 = false, Head2 = false;
    int Weight = 0, GroupWeight = 0;
    Edge_t() {}
    Edge_t(int i1, bool h1, int i2, bool h2, int w = 1) : Weight(w), GroupWeight(0) {
        if (i1 > i2) { Ind1 = i2; Head1 = h2; Ind2 = i1; Head2 = h1; }
        else { Ind1 = i1; Head1 = h1; Ind2 = i2; Head2 = h2; }
    }
    bool operator<(const Edge_t& r) const {
        if (Ind1 != r.Ind1) return Ind1 < r.Ind1;
        if (Ind2 != r.Ind2) return Ind2 < r.Ind2;
        if (Head1 != r.Head1) return (int)Head1 < (int)r.Head1;
        if (Head2 != r.Head2) return (int)Head2 < (int)r.Head2;
        return false;
    }
    bool operator==(const Edge_t& r) const { return Ind1 == r.Ind1 && Ind2 == r.Ind2 && Head1 == r.Head1 && Head2 == r.Head2; }
};

// src/BPNode.h:26-57 (adjacency holds edge indices instead of pointers; same iteration order)
struct Node_t {
    int Chr = 0, Position = 0, Length = 0, Support = 0;
    double AvgDepth = 0;
    std::vector<int> HeadEdges, TailEdges;
    Node_t() {}
    Node_t(int c, int p, int l, int s = 0, double d = 0) : Chr(c), Position(p), Length(l), Support(s), AvgDepth(d) {}
    bool operator<(const Node_t& r) const {
        if (Chr != r.Chr) return Chr < r.Chr;
        if (Position != r.Position) return Position < r.Position;
        return Length < r.Length;
    }
};

typedef std::pair<int, int> pii;
typedef std::map<Edge_t, std::vector<pii>> EdgeBPMap;

inline bool pii_less(pii a, pii b) { return a.first != b.first ? a.first < b.first : a.second < b.second; }

// src/SegmentGraph.cpp:51-102
inline void CountTop(const Edge_t& e, std::vector<pii>& x) {
    std::sort(x.begin(), x.end(), pii_less);
    std::vector<pii> y = x;
    y.resize(std::distance(y.begin(), std::unique(y.begin(), y.end())));
    std::vector<double> count(y.size(), 0);
    for (size_t i = 0; i < y.size(); i++)
        for (size_t j = 0; j < x.size(); j++)
            if (y[i] == x[j]) count[i] += 1;
            else if (std::abs(y[i].first - x[j].first) + std::abs(y[i].second - x[j].second) < 10) count[i] += 0.5;
    x.clear();
    while (x.size() < 5) {
        std::vector<double>::iterator it = std::max_element(count.begin(), count.end());
        if ((*it) > 3) {
            const pii& c = y[std::distance(count.begin(), it)];
            bool flag = true;
            for (size_t i = 0; i < x.size(); i++)
                if (std::abs(x[i].first - c.first) + std::abs(x[i].second - c.second) < 50) flag = false;
            if (flag) x.push_back(c);
        } else
            break;
        *it = 0;
    }
    if (x.size() == 0) {
        int maxBP1 = 0, maxBP2 = 0, minBP1 = std::numeric_limits<int>::max(), minBP2 = std::numeric_limits<int>::max();
        for (const pii& p : y) {
            minBP1 = std::min(minBP1, p.first); maxBP1 = std::max(maxBP1, p.first);
            minBP2 = std::min(minBP2, p.second); maxBP2 = std::max(maxBP2, p.second);
        }
        x.push_back(pii(e.Head1 ? minBP1 : maxBP1, e.Head2 ? minBP2 : maxBP2));
    }
}

struct StageSink;  // optional per-stage dump hook (squid_oracle.cpp)

class SegmentGraph_t {
public:
    std::vector<Node_t> vNodes;
    std::vector<Edge_t> vEdges;
    std::vector<int> Label;
    const Params& P;
    std::vector<Node_t> seedNodes;      // vNodes right after the stream loop (before NormalizeSeedNodes); dump only
    std::vector<bool> KeepEdgeDump;     // dump only
    long n_kept_records = 0, n_break_record = -1;  // stream-loop bookkeeping; dump only

    explicit SegmentGraph_t(const Params& P) : P(P) {}

    // ---- src/SegmentGraph.cpp:159-190
    bool IsDiscordant(const Edge_t& e) const {
        int ind1 = e.Ind1, ind2 = e.Ind2;
        if (vNodes[ind1].Chr != vNodes[ind2].Chr) return true;
        else if (vNodes[ind2].Position - vNodes[ind1].Position - vNodes[ind1].Length > P.Concord_Dist_Pos && ind2 - ind1 > P.Concord_Dist_Idx) return true;
        else if (e.Head1 != false || e.Head2 != true) return true;
        return false;
    }
    bool IsDiscordant(int edgeidx) const { return IsDiscordant(vEdges[edgeidx]); }

    // sorted unique stripped names of Chimrecord, with the size-then-push_back bug that also adds ""
    // (src/SegmentGraph.cpp:196-201, ledger B9)
    static std::vector<std::string> BuildChimName(const SBamrecord_t& Chimrecord) {
        std::vector<std::string> ChimName(Chimrecord.size());
        for (const ReadRec_t& r : Chimrecord) ChimName.push_back(r.Qname);
        std::sort(ChimName.begin(), ChimName.end());
        ChimName.resize(std::distance(ChimName.begin(), std::unique(ChimName.begin(), ChimName.end())));
        return ChimName;
    }
    // record filter of the three concordant-BAM passes: :297-303, :1579-1585, :3131-3137
    bool RecordFiltered(const BamAlignment& record, const std::vector<std::string>& ChimName, bool testRefID) const {
        bool XAtag = record.HasTag("XA");
        bool IHtag = record.HasTag("IH");
        int IHtagvalue = 0;
        if (IHtag) record.GetTagInt("IH", IHtagvalue);
        return XAtag || IHtagvalue > 1 || record.MapQuality < P.Min_MapQual || record.IsDuplicate() || !record.IsMapped() ||
               (testRefID && record.RefID == -1) || std::binary_search(ChimName.begin(), ChimName.end(), record.Name);
    }
    // mate stub appended before the duplicate test (:307-314, :1589-1596)
    static void AppendMateStub(const BamAlignment& record, ReadRec_t& r) {
        if (record.IsMateMapped() && record.MateRefID != -1) {
            SingleBamRec_t tmp(record.MateRefID, record.MatePosition, 0, 15, 15, 60, record.IsMateReverseStrand(), false);
            if (record.IsFirstMate()) r.SecondMate.push_back(tmp);
            else r.FirstRead.push_back(tmp);
        }
    }

    // ---- src/SegmentGraph.cpp:104-124
    void Construct(const std::vector<int>& RefLength, SBamrecord_t& Chimrecord, const std::string& bamfile, StageSink* sink, uint16_t* ReadLen = nullptr);

    void BuildNode_STAR(const std::vector<int>& RefLength, SBamrecord_t& Chimrecord, const std::string& bamfile);
    void BuildNode_BWA(const std::vector<int>& RefLength, const std::string& bamfile, uint16_t& ReadLen);  // o_bwa.h
    void RawEdges(SBamrecord_t& Chimrecord, const std::string& bamfile);                                   // o_bwa.h
    std::vector<int> LocateRead(int initialguess, ReadRec_t& ReadRec) const;
    void RawEdgesChim(SBamrecord_t& Chimrecord);
    void RawEdgesOther(SBamrecord_t& Chimrecord, const std::string& bamfile);
    void BuildEdges(SBamrecord_t& Chimrecord, const std::string& bamfile);
    void FilterbyWeight();
    void FilterbyInterleaving(std::vector<bool>& KeepEdge);
    int GroupConnection(int node, const std::vector<int>& Edges, int sumweight, std::vector<int>& Connection, std::vector<int>& Lab) const;
    void GroupSelect(int node, const std::vector<int>& Edges, int sumweight, int count, std::vector<int>& Connection, std::vector<int>& Lab,
                     std::vector<Edge_t>& ToDelete) const;
    void FilterEdges(const std::vector<bool>& KeepEdge);
    void UpdateNodeLink();
    void CompressNode();
    void FurtherCompressNode();
    void ConnectedComponent();
    void MultiplyDisEdges();
    void DeMultiplyDisEdges();
    void ExactBreakpoint(SBamrecord_t& Chimrecord, EdgeBPMap& ExactBP) const;
    void ExactBPConcordantSupport(const std::string& Input_BAM, SBamrecord_t& Chimrecord, const EdgeBPMap& ExactBP, EdgeBPMap& Support) const;

private:
    // helpers shared by RawEdgesChim / RawEdgesOther / ExactBreakpoint
    int HomeNode(int start, const SingleBamRec_t& b) const;
};

// =====================================================================================================
// BuildNode_STAR  (src/SegmentGraph.cpp:192-831)
// =====================================================================================================
inline void SegmentGraph_t::BuildNode_STAR(const std::vector<int>& RefLength, SBamrecord_t& Chimrecord, const std::string& bamfile) {
    const int ReadLen = P.ReadLen;
    std::vector<std::string> ChimName = BuildChimName(Chimrecord);

    // ---- (i) discordant blocks and clip positions from the chimeric fragments (:203-264)
    std::vector<pii> PartAlignPos;
    PartAlignPos.resize(RefLength.size());  // ledger B10: n_ref spurious (0,0) entries
    std::vector<SingleBamRec_t> bamdiscordant;
    for (const ReadRec_t& r : Chimrecord) {
        if (r.IsEndDiscordant(true) || r.IsEndDiscordant(false) || r.IsSingleAnchored() || r.IsPairDiscordant()) {
            for (const SingleBamRec_t& s : r.FirstRead) bamdiscordant.push_back(s);
            for (const SingleBamRec_t& s : r.SecondMate) bamdiscordant.push_back(s);
        } else {
            bool firstinserted = false, secondinserted = false;
            auto farpairs = [&](const std::vector<SingleBamRec_t>& R, bool& inserted) {  // :217-239
                int previnserted = -1;
                if (R.size() > 0)
                    for (int i = 0; i < (int)R.size() - 1; i++)
                        if (std::abs(R[i].RefPos - R[i + 1].RefPos) > 750000) {
                            if (previnserted != i) bamdiscordant.push_back(R[i]);
                            bamdiscordant.push_back(R[i + 1]);
                            previnserted = i + 1;
                            if (i + 1 == (int)R.size() - 1) inserted = true;
                        }
            };
            farpairs(r.FirstRead, firstinserted);
            farpairs(r.SecondMate, secondinserted);
            if (r.FirstRead.size() > 0 && r.SecondMate.size() > 0) {
                if (std::abs(r.FirstRead.back().RefPos - r.SecondMate.back().RefPos) > 750000) {
                    if (!firstinserted) { bamdiscordant.push_back(r.FirstRead.back()); firstinserted = true; }
                    if (!secondinserted) { bamdiscordant.push_back(r.SecondMate.back()); secondinserted = true; }
                }
            }
            if (!firstinserted && !secondinserted) {
                const std::vector<SingleBamRec_t>&F = r.FirstRead, &S = r.SecondMate;
                if (F.size() != 0 && F.front().ReadPos > 15 && !r.FirstLowPhred)
                    PartAlignPos.push_back(pii(F[0].RefID, F[0].IsReverse ? (F[0].RefPos + F[0].MatchRef) : F[0].RefPos));
                if (F.size() != 0 && r.FirstTotalLen - F.back().ReadPos - F.back().MatchRead > 15 && !r.FirstLowPhred)
                    PartAlignPos.push_back(pii(F.back().RefID, F.back().IsReverse ? F.back().RefPos : (F.back().RefPos + F.back().MatchRef)));
                if (S.size() != 0 && S.front().ReadPos > 15 && !r.SecondLowPhred)
                    PartAlignPos.push_back(pii(S[0].RefID, S[0].IsReverse ? (S[0].RefPos + S[0].MatchRef) : S[0].RefPos));
                // ledger B11: bamdiscordant.back() is read even when the vector is empty; the zero-filled
                // allocator convention of the oracle makes that an all-zero block.
                SingleBamRec_t lastdis = bamdiscordant.empty() ? SingleBamRec_t() : bamdiscordant.back();
                if (S.size() != 0 && r.SecondTotalLen - S.back().ReadPos - S.back().MatchRead > 15 && !lastdis.Same(S.back()) && !r.SecondLowPhred)
                    PartAlignPos.push_back(pii(S.back().RefID, S.back().IsReverse ? S.back().RefPos : (S.back().RefPos + S.back().MatchRef)));
            }
        }
    }
    std::sort(PartAlignPos.begin(), PartAlignPos.end(), pii_less);
    std::sort(bamdiscordant.begin(), bamdiscordant.end());  // (RefID,RefPos) only; tie order = introsort (ledger B8)
    const int NDIS = (int)bamdiscordant.size();
    bamdiscordant.push_back(SingleBamRec_t());  // ledger B21: explicit zero sentinel at cend()
    const std::vector<SingleBamRec_t>& D = bamdiscordant;

    // ---- (ii)+(iii) stream the concordant BAM
    int itdisstart = 0, itdisend = 0, itdiscurrent = 0;                   // indices into D; NDIS == cend()
    size_t itpartstart = 0, itpartend = 0, itpartcurrent = 0;
    std::vector<std::pair<int, pii>> ReadsMain, ReadsOther;
    std::vector<SingleBamRec_t> ConcordRest;                              // min-heap via MinHeapComp (:15-17)
    auto MinHeapComp = [](const SingleBamRec_t& l, const SingleBamRec_t& r) { return !(l < r); };
    std::vector<SingleBamRec_t> ConcordantCluster, PartialAlignCluster;
    int offsetConcordantCluster = 0, offsetPartialAlignCluster = 0;
    const int thresh = 3;
    int disChr = 0, otherChr = 0, nextdisChr = 0;
    int disrightmost = 0, otherrightmost = 0, nextdisrightmost = 0;
    int markedNodeStart = -1, markedNodeChr = -1;
    auto CC = [&]() -> std::vector<SingleBamRec_t>& { return ConcordantCluster; };
    auto PC = [&]() -> std::vector<SingleBamRec_t>& { return PartialAlignCluster; };
    auto newcluster = [&]() {  // :341-348 and :604-611
        disrightmost = nextdisrightmost; disChr = nextdisChr;
        nextdisrightmost = D[itdisstart].RefPos + D[itdisstart].MatchRef;
        for (itdisend = itdisstart; itdisend != NDIS && D[itdisend].RefID == D[itdisstart].RefID && D[itdisend].RefPos < nextdisrightmost + ReadLen; itdisend++) {
            nextdisrightmost = std::max(nextdisrightmost, D[itdisend].RefPos + D[itdisend].MatchRef);
            nextdisChr = D[itdisend].RefID;
        }
    };

    ReadRec_t lastreadrec;
    BamReader bamreader;
    bamreader.Open(bamfile);
    if (bamreader.IsOpen()) {
        BamAlignment record;
        while (bamreader.GetNextAlignment(record)) {
            if (RecordFiltered(record, ChimName, true)) continue;
            ReadRec_t readrec(record, P);
            ReadRec_t tmpreadrec = readrec;
            tmpreadrec.SortbyReadPos();
            AppendMateStub(record, tmpreadrec);
            if (ReadRec_t::Equal(lastreadrec, tmpreadrec)) continue;
            else lastreadrec = tmpreadrec;
            n_kept_records++;

            {   // :320-333
                const std::vector<SingleBamRec_t>* R = nullptr;
                if (record.IsFirstMate() && readrec.FirstRead.size() != 0) R = &readrec.FirstRead;
                else if (readrec.SecondMate.size() != 0) R = &readrec.SecondMate;
                if (R) {
                    ReadsMain.push_back(std::make_pair((*R)[0].RefID, pii((*R)[0].RefPos, (*R)[0].MatchRef)));
                    for (size_t i = 1; i < R->size(); i++) ReadsOther.push_back(std::make_pair((*R)[i].RefID, pii((*R)[i].RefPos, (*R)[i].MatchRef)));
                }
            }
            if (itdisstart == NDIS) { n_break_record = n_kept_records; break; }  // :338-339, ledger B12
            if (itdisend - itdisstart <= 0) newcluster();

            // ---- the stream passed the current discordant cluster: segment it (:353-612)
            while (itdisstart != NDIS && (D[itdisstart].RefID < record.RefID || (D[itdisstart].RefID == record.RefID && nextdisrightmost < record.Position))) {
                int curEndPos = 0, curStartPos = 0;
                int disStartPos = -1, disEndPos = -1, disCount = -1;
                bool isClusternSplit = false;
                if (markedNodeStart != -1 && D[itdisstart].RefID != markedNodeChr) { markedNodeChr = -1; markedNodeStart = -1; }

                while ((int)CC().size() != offsetConcordantCluster && CC()[offsetConcordantCluster].RefID < D[itdisstart].RefID) offsetConcordantCluster++;
                while ((int)PC().size() != offsetPartialAlignCluster && PC()[offsetPartialAlignCluster].RefID < D[itdisstart].RefID) offsetPartialAlignCluster++;
                if ((int)CC().size() != offsetConcordantCluster && D[itdisstart].RefPos > CC().back().RefPos + CC().back().MatchRef + ReadLen)
                    offsetConcordantCluster = (int)CC().size();
                if ((int)PC().size() != offsetPartialAlignCluster && D[itdisstart].RefPos > PC().back().RefPos + PC().back().MatchRef + ReadLen)
                    offsetPartialAlignCluster = (int)PC().size();
                curStartPos = D[itdisstart].RefPos;
                {   // :376-385
                    bool hc = (int)CC().size() != offsetConcordantCluster, hp = (int)PC().size() != offsetPartialAlignCluster;
                    SingleBamRec_t ittmp;
                    if (hc && hp) ittmp = (CC()[offsetConcordantCluster] < PC()[offsetPartialAlignCluster]) ? CC()[offsetConcordantCluster] : PC()[offsetPartialAlignCluster];
                    else if (hc) ittmp = CC()[offsetConcordantCluster];
                    else if (hp) ittmp = PC()[offsetPartialAlignCluster];
                    if ((hc || hp) && (ittmp.RefID < D[itdisstart].RefID || (ittmp.RefID == D[itdisstart].RefID && ittmp.RefPos < D[itdisstart].RefPos)))
                        curStartPos = ittmp.RefPos;
                }
                curStartPos = (curStartPos > markedNodeStart) ? curStartPos : markedNodeStart;

                while (ConcordRest.size() != 0 && (ConcordRest.front().RefID < D[itdisstart].RefID ||
                                                   (ConcordRest.front().RefID == D[itdisstart].RefID && ConcordRest.front().RefPos < D[itdisstart].RefPos - ReadLen))) {
                    std::pop_heap(ConcordRest.begin(), ConcordRest.end(), MinHeapComp);
                    ConcordRest.pop_back();
                }
                for (; itpartstart != PartAlignPos.size() && (PartAlignPos[itpartstart].first < D[itdisstart].RefID ||
                                                              (PartAlignPos[itpartstart].first == D[itdisstart].RefID && PartAlignPos[itpartstart].second + ReadLen < D[itdisstart].RefPos));
                     itpartstart++) {}
                for (itpartend = itpartstart; itpartend != PartAlignPos.size() && PartAlignPos[itpartend].first == D[itdisstart].RefID && PartAlignPos[itpartend].second < nextdisrightmost + ReadLen;
                     itpartend++) {}

                while (itdisstart != itdisend) {
                    const int chr = D[itdisstart].RefID;
                    if (itdisstart != 0 && D[itdisstart].RefID != D[itdisstart - 1].RefID && (int)CC().size() == offsetConcordantCluster && (int)PC().size() == offsetPartialAlignCluster)
                        curStartPos = D[itdisstart].RefPos;
                    isClusternSplit = false;
                    // candidate break positions (:400-435)
                    std::vector<int> MarginPositions;
                    for (itdiscurrent = itdisstart; itdiscurrent != itdisend; itdiscurrent++) {
                        MarginPositions.push_back(D[itdiscurrent].RefPos);
                        MarginPositions.push_back(D[itdiscurrent].RefPos + D[itdiscurrent].MatchRef);
                        curEndPos = (curEndPos > MarginPositions.back()) ? curEndPos : MarginPositions.back();
                        if ((itdiscurrent + 1) != itdisend) {
                            if (D[itdiscurrent + 1].RefPos > D[itdiscurrent].RefPos + D[itdiscurrent].MatchRef) break;
                        }
                    }
                    disStartPos = std::max(curStartPos, D[itdisstart].RefPos);
                    disEndPos = curEndPos;
                    disCount = itdiscurrent - itdisstart;
                    if (itdiscurrent != itdisend) {
                        for (itdiscurrent++; itdiscurrent != itdisend && D[itdiscurrent].RefPos < curEndPos + thresh; itdiscurrent++) {
                            MarginPositions.push_back(D[itdiscurrent].RefPos);
                            MarginPositions.push_back(D[itdiscurrent].RefPos + D[itdiscurrent].MatchRef);
                        }
                    }
                    for (itpartcurrent = itpartstart; itpartcurrent != itpartend && PartAlignPos[itpartcurrent].second < curEndPos + thresh; itpartcurrent++)
                        MarginPositions.push_back(PartAlignPos[itpartcurrent].second);
                    for (int i = offsetPartialAlignCluster; i != (int)PC().size(); i++) {
                        const SingleBamRec_t& it = PC()[i];
                        const int front = MarginPositions.front();
                        if (it.RefID == chr && it.ReadPos > 15 && it.RefPos > front - thresh && it.RefPos < curEndPos + thresh) {
                            if (it.IsReverse && it.RefPos + it.MatchRef > front - thresh && it.RefPos + it.MatchRef < curEndPos + thresh) MarginPositions.push_back(it.RefPos + it.MatchRef);
                            else if (!it.IsReverse && it.RefPos > front - thresh && it.RefPos < curEndPos + thresh) MarginPositions.push_back(it.RefPos);
                        } else if (it.RefID == chr) {
                            if (it.IsReverse && it.RefPos > front - thresh && it.RefPos < curEndPos + thresh) MarginPositions.push_back(it.RefPos);
                            else if (!it.IsReverse && it.RefPos + it.MatchRef > front - thresh && it.RefPos + it.MatchRef < curEndPos + thresh) MarginPositions.push_back(it.RefPos + it.MatchRef);
                        }
                    }
                    std::sort(MarginPositions.begin(), MarginPositions.end());

                    // evaluate candidates (:439-504)
                    int lastCurser = -1, lastSupport = 0;
                    auto closenode = [&](int lastC) {  // :483-493 and :506-515
                        isClusternSplit = true;
                        if (D[itdisstart].RefPos - curStartPos > thresh * 20 && lastC - D[itdisstart].RefPos > thresh * 20) {
                            vNodes.push_back(Node_t(chr, curStartPos, D[itdisstart].RefPos - curStartPos));
                            curStartPos = D[itdisstart].RefPos;
                        }
                        vNodes.push_back(Node_t(chr, curStartPos, lastC - curStartPos));
                        curStartPos = lastC; curEndPos = lastC;
                        markedNodeStart = lastC; markedNodeChr = chr;
                    };
                    for (size_t ib = 0; ib < MarginPositions.size(); ib++) {
                        const int brk = MarginPositions[ib];
                        bool skip = vNodes.size() != 0 && vNodes.back().Chr == chr && brk - vNodes.back().Position - vNodes.back().Length < thresh * 20;
                        if (!skip) {
                            int srsupport = 0, peleftfor = 0, perightrev = 0;
                            for (size_t i2 = 0; i2 < MarginPositions.size() && MarginPositions[i2] < brk + thresh; i2++)
                                if (std::abs(brk - MarginPositions[i2]) < thresh) srsupport++;
                            for (int d = itdisstart; d != itdisend; d++) {
                                if (D[d].RefPos + D[d].MatchRef < brk && D[d].RefPos + D[d].MatchRef > brk - ReadLen && !D[d].IsReverse) peleftfor++;
                                else if (D[d].RefPos > brk && D[d].RefPos < brk + ReadLen && D[d].IsReverse) perightrev++;
                            }
                            if (srsupport > 3 || srsupport + peleftfor > 4 || srsupport + perightrev > 4) {
                                auto spans = [&](const SingleBamRec_t& b) { return b.RefID == chr && b.RefPos + b.MatchRef >= brk + thresh && b.RefPos < brk - thresh; };
                                int coverage = 0;
                                for (int i = offsetConcordantCluster; i < (int)CC().size(); i++) if (spans(CC()[i])) coverage++;
                                for (int d = itdisstart; d != itdisend; d++) if (spans(D[d])) coverage++;
                                for (int i = offsetPartialAlignCluster; i != (int)PC().size(); i++) if (spans(PC()[i])) coverage++;
                                if (srsupport > std::max(coverage - srsupport, 0) + 2)
                                    for (size_t i = 0; i < ConcordRest.size(); i++) if (spans(ConcordRest[i])) coverage++;
                                if (srsupport > std::max(coverage - srsupport, 0) + 2) {
                                    int sup = std::max(srsupport + peleftfor, srsupport + perightrev);
                                    if (lastCurser == -1 && brk - curStartPos < thresh * 20) {
                                        markedNodeStart = curStartPos; markedNodeChr = chr;
                                    } else if ((lastCurser == -1 || brk - lastCurser < thresh * 20) && sup > lastSupport) {
                                        lastCurser = brk; lastSupport = sup;
                                    } else if (brk - lastCurser >= thresh * 20) {
                                        closenode(lastCurser);
                                        lastCurser = brk;
                                    }
                                }
                            }
                        }
                        // jump to the last copy of this position (:498-503)
                        size_t nx = ib;
                        while (nx < MarginPositions.size() && MarginPositions[nx] == brk) nx++;
                        if (nx == MarginPositions.size()) break;
                        ib = nx - 1;
                    }
                    if (lastCurser != -1 && (!isClusternSplit || vNodes.back().Position + vNodes.back().Length != lastCurser)) closenode(lastCurser);
                    // dense cluster without an accepted break becomes one node (:518-527)
                    if (disStartPos != -1 && !isClusternSplit && disCount > std::min(5.0, 4.0 * (disEndPos - disStartPos) / ReadLen)) {
                        if (vNodes.size() != 0 && vNodes.back().Chr == D[itdisend - 1].RefID && disEndPos - vNodes.back().Position - vNodes.back().Length < thresh * 20)
                            vNodes.back().Length += disEndPos - vNodes.back().Position - vNodes.back().Length;
                        else
                            vNodes.push_back(Node_t(D[itdisend - 1].RefID, disStartPos, disEndPos - disStartPos));
                        curStartPos = disEndPos; curEndPos = disEndPos;
                        markedNodeStart = disEndPos; markedNodeChr = chr;
                    }
                    // window offsets past the new node; extend to zero coverage (:528-601)
                    while ((int)CC().size() != offsetConcordantCluster && CC()[offsetConcordantCluster].RefID < chr) offsetConcordantCluster++;
                    while ((int)PC().size() != offsetPartialAlignCluster && PC()[offsetPartialAlignCluster].RefID < chr) offsetPartialAlignCluster++;
                    for (itdiscurrent = itdisstart; itdiscurrent != itdisend && D[itdiscurrent].RefPos + D[itdiscurrent].MatchRef <= curEndPos; itdiscurrent++) {}
                    int concord0pos = curStartPos;
                    auto step1 = [&](std::vector<SingleBamRec_t>& V, int& off) -> bool {  // :539-551 / :552-564
                        if ((int)V.size() == off) return false;
                        bool flag = true;
                        const SingleBamRec_t& b = V[off];
                        if (b.RefID > chr) flag = false;
                        if (itdiscurrent != NDIS && b.RefID == D[itdiscurrent].RefID && b.RefPos + b.MatchRef + ReadLen >= D[itdiscurrent].RefPos) flag = false;
                        if (vNodes.size() != 0 && (b.RefID > vNodes.back().Chr || (b.RefID == vNodes.back().Chr && b.RefPos >= vNodes.back().Position + vNodes.back().Length))) flag = false;
                        if (flag) { concord0pos = std::max(concord0pos, b.RefPos + b.MatchRef); off++; }
                        return flag;
                    };
                    do {
                        bool flag1 = step1(CC(), offsetConcordantCluster);
                        bool flag2 = step1(PC(), offsetPartialAlignCluster);
                        if (!flag1 && !flag2) break;
                    } while ((int)CC().size() != offsetConcordantCluster || (int)PC().size() != offsetPartialAlignCluster);
                    auto step2 = [&](std::vector<SingleBamRec_t>& V, int& off) -> bool {  // :583-590 / :591-598
                        if ((int)V.size() == off) return false;
                        const SingleBamRec_t& b = V[off];
                        bool flag = false;
                        if (itdiscurrent == NDIS || b.RefID < D[itdiscurrent].RefID || (b.RefID == D[itdiscurrent].RefID && b.RefPos + b.MatchRef + ReadLen < D[itdiscurrent].RefPos)) flag = true;
                        if (flag) { concord0pos = std::max(concord0pos, b.RefPos + b.MatchRef); off++; }
                        return flag;
                    };
                    do {
                        if (markedNodeStart != -1 && (record.RefID > markedNodeChr || record.Position > concord0pos + ReadLen) &&
                            ((int)CC().size() == offsetConcordantCluster || CC()[offsetConcordantCluster].RefID != markedNodeChr || CC()[offsetConcordantCluster].RefPos > concord0pos + ReadLen) &&
                            ((int)PC().size() == offsetPartialAlignCluster || PC()[offsetPartialAlignCluster].RefID != markedNodeChr || PC()[offsetPartialAlignCluster].RefPos > concord0pos)) {
                            if (concord0pos > markedNodeStart && concord0pos < markedNodeStart + thresh * 20 && vNodes.size() != 0 && vNodes.back().Chr == markedNodeChr)
                                vNodes.back().Length += (concord0pos - vNodes.back().Position - vNodes.back().Length);
                            else if (concord0pos > markedNodeStart)
                                vNodes.push_back(Node_t(markedNodeChr, markedNodeStart, concord0pos - markedNodeStart));
                            curStartPos = concord0pos;
                            markedNodeChr = -1; markedNodeStart = -1;
                            break;
                        }
                        bool flag1 = step2(CC(), offsetConcordantCluster);
                        bool flag2 = step2(PC(), offsetPartialAlignCluster);
                        if (!flag1 && !flag2) break;
                    } while ((int)CC().size() != offsetConcordantCluster || (int)PC().size() != offsetPartialAlignCluster);
                    itdisstart = itdiscurrent;
                }
                if (itdisend - itdisstart <= 0) newcluster();  // at cend() this reads the zero sentinel (ledger B21)
            }

            // ---- zero-coverage test for a pending node end (:616-630)
            int currightmost = (disChr > otherChr || (disChr == otherChr && disrightmost > otherrightmost)) ? disrightmost : otherrightmost;
            int curChr = (disChr > otherChr) ? disChr : otherChr;
            bool is0coverage = ((record.RefID != curChr || record.Position > currightmost + ReadLen) &&
                                (curChr < D[itdisstart].RefID || (curChr == D[itdisstart].RefID && currightmost + ReadLen < D[itdisstart].RefPos)));
            if (is0coverage && markedNodeStart != -1) {
                if (curChr == markedNodeChr && currightmost > markedNodeStart && currightmost - markedNodeStart < thresh * 20 && vNodes.size() > 0 &&
                    markedNodeStart == vNodes.back().Position + vNodes.back().Length)
                    vNodes.back().Length += currightmost - markedNodeStart;
                else if (curChr == markedNodeChr && currightmost > markedNodeStart && currightmost - markedNodeStart >= thresh * 20)
                    vNodes.push_back(Node_t(markedNodeChr, markedNodeStart, currightmost - markedNodeStart));
                markedNodeStart = -1; markedNodeChr = -1;
            }
            // ---- prune the windows (:633-646)
            if (is0coverage && (curChr != D[itdisstart].RefID || currightmost + ReadLen < D[itdisstart].RefPos)) {
                offsetConcordantCluster = (int)CC().size();
                offsetPartialAlignCluster = (int)PC().size();
            } else {
                auto prune = [&](std::vector<SingleBamRec_t>& V, int& off) {
                    while ((int)V.size() > off && V[off].RefID != record.RefID) off++;
                    while ((int)V.size() > off && (V[off].RefID < D[itdisstart].RefID ||
                                                   (vNodes.size() != 0 && V[off].RefID == vNodes.back().Chr && V[off].RefPos < vNodes.back().Position + vNodes.back().Length)))
                        off++;
                };
                prune(CC(), offsetConcordantCluster);
                prune(PC(), offsetPartialAlignCluster);
            }
            // ---- push the record into the windows (:649-700)
            bool recordconcordant = false, recordpartalign = false;
            if (record.IsMapped() && record.IsMateMapped() && record.MateRefID != -1 && record.IsReverseStrand() && !record.IsMateReverseStrand() && record.RefID == record.MateRefID &&
                record.Position >= record.MatePosition && record.Position - record.MatePosition <= 750000 && record.IsProperPair())
                recordconcordant = true;
            else if (record.IsMapped() && record.IsMateMapped() && record.MateRefID != -1 && !record.IsReverseStrand() && record.IsMateReverseStrand() && record.RefID == record.MateRefID &&
                     record.MatePosition >= record.Position && record.MatePosition - record.Position <= 750000 && record.IsProperPair())
                recordconcordant = true;
            if (recordconcordant && (int)readrec.FirstRead.size() + (int)readrec.SecondMate.size() > 0) {
                // NB: the reference dereferences FirstRead/SecondMate by mate flag; a record carrying neither
                // 0x40 nor 0x80 lands in SecondMate but matches no branch below (kept).
                if (otherChr == record.RefID && record.IsFirstMate())
                    otherrightmost = std::max(otherrightmost, readrec.FirstRead.front().RefPos + readrec.FirstRead.front().MatchRef);
                else if (otherChr == record.RefID && record.IsSecondMate())
                    otherrightmost = std::max(otherrightmost, readrec.SecondMate.front().RefPos + readrec.SecondMate.front().MatchRef);
                else if (record.IsFirstMate()) { otherrightmost = readrec.FirstRead.front().RefPos + readrec.FirstRead.front().MatchRef; otherChr = record.RefID; }
                else if (record.IsSecondMate()) { otherrightmost = readrec.SecondMate.front().RefPos + readrec.SecondMate.front().MatchRef; otherChr = record.RefID; }
                if (record.IsFirstMate() && tmpreadrec.FirstRead.front().ReadPos > 15 && !tmpreadrec.FirstLowPhred) {
                    PC().push_back(readrec.FirstRead.front()); recordpartalign = true;
                } else if (record.IsFirstMate() && tmpreadrec.FirstTotalLen - tmpreadrec.FirstRead.back().ReadPos - tmpreadrec.FirstRead.back().MatchRead > 15 && !tmpreadrec.FirstLowPhred) {
                    PC().push_back(readrec.FirstRead.front()); recordpartalign = true;
                }
                if (record.IsSecondMate() && tmpreadrec.SecondMate.front().ReadPos > 15 && !tmpreadrec.SecondLowPhred) {
                    PC().push_back(readrec.SecondMate.front()); recordpartalign = true;
                } else if (record.IsSecondMate() && tmpreadrec.SecondTotalLen - tmpreadrec.SecondMate.back().ReadPos - tmpreadrec.SecondMate.back().MatchRead > 15 && !tmpreadrec.SecondLowPhred) {
                    PC().push_back(readrec.SecondMate.front()); recordpartalign = true;
                }
                if (!recordpartalign) {
                    if (record.IsFirstMate()) CC().push_back(readrec.FirstRead.front());
                    else CC().push_back(readrec.SecondMate.front());
                }
                auto pushrest = [&](const std::vector<SingleBamRec_t>& R) {
                    for (size_t i = 1; i < R.size(); i++)
                        if (itdisstart != NDIS && R[i].RefPos >= D[itdisstart].RefPos - ReadLen) {
                            ConcordRest.push_back(R[i]);
                            std::push_heap(ConcordRest.begin(), ConcordRest.end(), MinHeapComp);
                        }
                };
                if (record.IsFirstMate() && readrec.FirstRead.size() > 1) pushrest(readrec.FirstRead);
                if (record.IsSecondMate() && readrec.SecondMate.size() > 1) pushrest(readrec.SecondMate);
            }
        }
    }
    seedNodes = vNodes;

    // ---- (iv) NormalizeSeedNodes (:19-38,706), sanity asserts (:708-712), whole-genome tiling (:714-761)
    if (vNodes.size() >= 2) {
        std::sort(vNodes.begin(), vNodes.end());
        std::vector<Node_t> normalized;
        int mergedCount = 0;
        for (const Node_t& node : vNodes) {
            if (normalized.size() == 0 || normalized.back().Chr != node.Chr || normalized.back().Position + normalized.back().Length <= node.Position)
                normalized.push_back(node);
            else {
                int mergedEnd = std::max(normalized.back().Position + normalized.back().Length, node.Position + node.Length);
                normalized.back().Length = mergedEnd - normalized.back().Position;
                mergedCount++;
            }
        }
        if (mergedCount > 0) std::cerr << "[SQUID] normalized " << mergedCount << " overlapping seed nodes in BuildNode_STAR.\n";
        vNodes.swap(normalized);
    }
    for (size_t i = 0; i < vNodes.size(); i++) {
        bool ok = vNodes[i].Length > 0 && vNodes[i].Position + vNodes[i].Length <= RefLength[vNodes[i].Chr];
        if (i + 1 < vNodes.size()) ok = ok && ((vNodes[i].Chr != vNodes[i + 1].Chr) || vNodes[i].Position + vNodes[i].Length <= vNodes[i + 1].Position);
        if (!ok) { std::cerr << "oracle: seed-node assertion of SegmentGraph.cpp:708-712 fails (reference aborts)\n"; std::exit(4); }
    }
    std::vector<Node_t> tmpNodes;
    for (size_t i = 0; i < vNodes.size(); i++) {
        if (tmpNodes.size() == 0 || tmpNodes.back().Chr != vNodes[i].Chr) {
            if (tmpNodes.size() != 0 && tmpNodes.back().Position + tmpNodes.back().Length != RefLength[tmpNodes.back().Chr])
                tmpNodes.push_back(Node_t(tmpNodes.back().Chr, tmpNodes.back().Position + tmpNodes.back().Length, RefLength[tmpNodes.back().Chr] - tmpNodes.back().Position - tmpNodes.back().Length));
            int chrstart = (tmpNodes.size() == 0) ? 0 : (tmpNodes.back().Chr + 1);
            for (; chrstart != vNodes[i].Chr; chrstart++) tmpNodes.push_back(Node_t(chrstart, 0, RefLength[chrstart]));
            if (vNodes[i].Position != 0) {
                if (vNodes[i].Position > 100)
                    tmpNodes.push_back(Node_t(vNodes[i].Chr, 0, vNodes[i].Position));
                else {
                    vNodes[i].Length += vNodes[i].Position; vNodes[i].Position = 0;
                    tmpNodes.push_back(vNodes[i]);
                    continue;
                }
            }
        }
        // (empty tmpNodes + seed at chr 0 pos 0: the reference reads back() of an empty vector; treated as no gap)
        if (tmpNodes.size() != 0 && tmpNodes.back().Position + tmpNodes.back().Length < vNodes[i].Position) {
            int gap = vNodes[i].Position - tmpNodes.back().Position - tmpNodes.back().Length;
            if (gap > 100) {
                tmpNodes.push_back(Node_t(vNodes[i].Chr, tmpNodes.back().Position + tmpNodes.back().Length, gap));
                tmpNodes.push_back(vNodes[i]);
            } else {
                vNodes[i].Length += gap;
                vNodes[i].Position = tmpNodes.back().Position + tmpNodes.back().Length;
                tmpNodes.push_back(vNodes[i]);
            }
        } else
            tmpNodes.push_back(vNodes[i]);
    }
    if (tmpNodes.size() != 0 && tmpNodes.back().Position + tmpNodes.back().Length != RefLength[tmpNodes.back().Chr])
        tmpNodes.push_back(Node_t(tmpNodes.back().Chr, tmpNodes.back().Position + tmpNodes.back().Length, RefLength[tmpNodes.back().Chr] - tmpNodes.back().Position - tmpNodes.back().Length));
    // NB: with no seed node at all the reference evaluates tmpNodes.back() on an empty vector (UB); the oracle
    // tiles every chromosome as one node in that case.
    for (int chrstart = tmpNodes.empty() ? 0 : tmpNodes.back().Chr + 1; chrstart < (int)RefLength.size(); chrstart++) tmpNodes.push_back(Node_t(chrstart, 0, RefLength[chrstart]));
    vNodes = tmpNodes;

    // ---- (v) Support / AvgDepth (:766-826)
    int itdis = 0;
    for (size_t i = 0; i < vNodes.size(); i++) {
        int count = 0, sumlen = 0;
        for (; itdis != NDIS && D[itdis].RefID == vNodes[i].Chr && D[itdis].RefPos < vNodes[i].Position + vNodes[i].Length; itdis++)
            if (D[itdis].RefPos >= vNodes[i].Position && D[itdis].RefPos + D[itdis].MatchRef <= vNodes[i].Position + vNodes[i].Length) { count++; sumlen += D[itdis].MatchRef; }
        vNodes[i].Support = count;
        vNodes[i].AvgDepth = sumlen;
    }
    std::sort(ReadsOther.begin(), ReadsOther.end(), [](const std::pair<int, pii>& a, const std::pair<int, pii>& b) {
        if (a.first != b.first) return a.first < b.first;
        return a.second.first < b.second.first;
    });
    auto sweep = [&](const std::vector<std::pair<int, pii>>& Reads, bool normalise) {
        if (Reads.size() == 0) return;
        size_t it = 0;
        for (size_t i = 0; i < vNodes.size(); i++) {
            int covcount = 0, covsumlen = 0;
            for (; it != Reads.size(); it++) {
                const std::pair<int, pii>& r = Reads[it];
                if (r.first == vNodes[i].Chr && r.second.first >= vNodes[i].Position - thresh && r.second.first + r.second.second <= vNodes[i].Position + vNodes[i].Length + thresh) {
                    covcount++; covsumlen += r.second.second;
                } else if (r.second.first >= vNodes[i].Position + vNodes[i].Length || r.first != vNodes[i].Chr)
                    break;
            }
            vNodes[i].Support += covcount;
            vNodes[i].AvgDepth += covsumlen;
            if (normalise) vNodes[i].AvgDepth = 1.0 * vNodes[i].AvgDepth / vNodes[i].Length;  // ledger B13
        }
    };
    sweep(ReadsMain, false);
    sweep(ReadsOther, true);
}

// =====================================================================================================
// LocateRead (src/SegmentGraph.cpp:1207-1293)
// =====================================================================================================
inline std::vector<int> SegmentGraph_t::LocateRead(int initialguess, ReadRec_t& ReadRec) const {
    std::vector<int> tmpRead_Node((int)ReadRec.FirstRead.size() + (int)ReadRec.SecondMate.size(), 0);
    int i = initialguess;
    const int thresh = 5;
    const int N = (int)vNodes.size();
    auto fits = [&](int n, const SingleBamRec_t& b) {
        return vNodes[n].Chr == b.RefID && b.RefPos >= vNodes[n].Position - thresh && b.RefPos + b.MatchRef <= vNodes[n].Position + vNodes[n].Length + thresh;
    };
    auto one = [&](SingleBamRec_t& b) -> int {
        if (i < 0 || i >= N) i = initialguess;
        if (!fits(i, b)) {
            if (vNodes[i].Chr < b.RefID || (vNodes[i].Chr == b.RefID && vNodes[i].Position <= b.RefPos)) {
                for (; i < N && vNodes[i].Chr <= b.RefID; i++) if (fits(i, b)) break;
            } else {
                for (; i > -1 && vNodes[i].Chr >= b.RefID; i--) if (fits(i, b)) break;
            }
        }
        if (i < 0 || i >= N || vNodes[i].Chr != b.RefID) return -1;
        const Node_t& n = vNodes[i];
        if (b.RefPos < n.Position) {  // trim the start (:1229-1239)
            int d = n.Position - b.RefPos;
            if (!b.IsReverse) b.ReadPos += d;
            b.MatchRef -= d; b.MatchRead -= d;
            b.RefPos = n.Position;
        }
        if (b.RefPos + b.MatchRef > n.Position + n.Length) {  // trim the end (:1240-1248)
            int d = b.RefPos + b.MatchRef - n.Position - n.Length;
            if (b.IsReverse) b.ReadPos += d;
            b.MatchRef -= d; b.MatchRead -= d;
        }
        return i;
    };
    for (size_t k = 0; k < ReadRec.FirstRead.size(); k++) tmpRead_Node[k] = one(ReadRec.FirstRead[k]);
    for (size_t k = 0; k < ReadRec.SecondMate.size(); k++) tmpRead_Node[ReadRec.FirstRead.size() + k] = one(ReadRec.SecondMate[k]);
    return tmpRead_Node;
}

// node containing the block start, found the way :1408-1409 / :1614-1615 scan for it
inline int SegmentGraph_t::HomeNode(int start, const SingleBamRec_t& b) const {
    int i = start;
    const int N = (int)vNodes.size();
    for (; i < N && (vNodes[i].Chr < b.RefID || (vNodes[i].Chr == b.RefID && vNodes[i].Position + vNodes[i].Length < b.RefPos)); i++) {}
    if (i >= N) { std::cerr << "oracle: block beyond the last node (reference reads past vNodes, SegmentGraph.cpp:1409)\n"; std::exit(4); }
    for (; i > -1 && (vNodes[i].Chr > b.RefID || (vNodes[i].Chr == b.RefID && vNodes[i].Position > b.RefPos)); i--) {}
    return i;
}

// =====================================================================================================
// RawEdgesChim (src/SegmentGraph.cpp:1394-1555)
// =====================================================================================================
namespace detail {
// breakpoints of a split junction between consecutive blocks a,b of one mate (:1435-1440, :3036-3041)
inline pii SplitBreakpoints(const SingleBamRec_t& a, const SingleBamRec_t& b) {
    int breakpoint1 = a.IsReverse ? a.RefPos : (a.RefPos + a.MatchRef);
    int breakpoint2 = b.IsReverse ? (b.RefPos + b.MatchRef) : b.RefPos;
    if (a > b) std::swap(breakpoint1, breakpoint2);
    return pii(breakpoint1, breakpoint2);
}
// the pair-edge suppression test shared by :1484-1502 and :1658-1676
inline bool PairOverlap(const ReadRec_t& r, const std::vector<int>& RN, int i, int j) {
    const int nf = (int)r.FirstRead.size();
    bool isoverlap = false;
    for (int k = 0; k < nf; k++) if (j == RN[k]) isoverlap = true;
    for (int k = 0; k < (int)r.SecondMate.size(); k++) if (i == RN[nf + k]) isoverlap = true;
    if (r.FirstRead.size() > 1) {
        if (r.IsEndDiscordant(true) && ((RN.front() <= j && RN[nf - 1] >= j) || (RN.front() >= j && RN[nf - 1] <= j))) isoverlap = true;
        else if (!r.IsEndDiscordant(true) && std::abs(i - j) < 3) isoverlap = true;
    }
    if (r.SecondMate.size() > 1) {
        if (r.IsEndDiscordant(false) && ((RN[nf] <= i && RN.back() >= i) || (RN[nf] >= i && RN.back() <= i))) isoverlap = true;
        else if (!r.IsEndDiscordant(false) && std::abs(i - j) < 3) isoverlap = true;
    }
    return isoverlap;
}
}  // namespace detail

inline void SegmentGraph_t::RawEdgesChim(SBamrecord_t& Chimrecord) {
    int firstfrontindex = 0;
    EdgeBPMap PairBreakpoints;
    for (ReadRec_t& r : Chimrecord) {
        if (r.FirstRead.size() == 0 && r.SecondMate.size() == 0) continue;
        std::vector<int> RN = LocateRead(firstfrontindex, r);
        if (RN[0] != -1) firstfrontindex = RN[0];
        const int nf = (int)r.FirstRead.size();
        for (int k = 0; k < (int)RN.size(); k++)
            if (RN[k] == -1) {
                const SingleBamRec_t& b = k < nf ? r.FirstRead[k] : r.SecondMate[k - nf];
                int i = HomeNode(firstfrontindex, b);
                vEdges.push_back(Edge_t(i, false, i + 1, true));  // no bounds assert here in the reference
            }
        auto splitedges = [&](const std::vector<SingleBamRec_t>& R, int base) {  // :1425-1479
            if (R.size() == 0) return;
            for (int k = 0; k < (int)R.size() - 1; k++) {
                int i = RN[base + k], j = RN[base + k + 1];
                if (i != j && i != -1 && j != -1) {
                    Edge_t tmp(i, R[k].IsReverse, j, !R[k + 1].IsReverse, 1);
                    if (!IsDiscordant(tmp)) vEdges.push_back(tmp);
                    else PairBreakpoints[tmp].push_back(detail::SplitBreakpoints(R[k], R[k + 1]));
                }
            }
        };
        splitedges(r.FirstRead, 0);
        splitedges(r.SecondMate, nf);
        if (r.FirstRead.size() > 0 && r.SecondMate.size() > 0) {  // :1481-1526
            if (!r.IsSingleAnchored() && !r.IsEndDiscordant(true) && !r.IsEndDiscordant(false)) {
                int i = RN[nf - 1], j = RN.back();
                bool isoverlap = detail::PairOverlap(r, RN, i, j);
                if (i != j && i != -1 && j != -1 && !isoverlap) {
                    const SingleBamRec_t &fb = r.FirstRead.back(), &sb = r.SecondMate.back();
                    Edge_t tmp(i, fb.IsReverse, j, sb.IsReverse, 1);
                    if (!IsDiscordant(tmp)) vEdges.push_back(tmp);
                    else if (r.IsPairDiscordant(false)) {
                        int breakpoint1 = fb.IsReverse ? fb.RefPos : (fb.RefPos + fb.MatchRef);
                        int breakpoint2 = sb.IsReverse ? sb.RefPos : (sb.RefPos + sb.MatchRef);
                        if (fb > sb) std::swap(breakpoint1, breakpoint2);
                        PairBreakpoints[tmp].push_back(pii(breakpoint1, breakpoint2));
                    }
                }
            }
        }
    }
    // :1529-1554 -- the groupcount loop is dead (its filter is commented out, :1547): Weight = #breakpoints
    for (EdgeBPMap::iterator it = PairBreakpoints.begin(); it != PairBreakpoints.end(); it++) {
        Edge_t tmp = it->first;
        tmp.Weight = (int)it->second.size();
        if (tmp.Weight > 0) vEdges.push_back(tmp);
    }
}

// =====================================================================================================
// RawEdgesOther (src/SegmentGraph.cpp:1557-1696)
// =====================================================================================================
inline void SegmentGraph_t::RawEdgesOther(SBamrecord_t& Chimrecord, const std::string& bamfile) {
    std::vector<std::string> ChimName = BuildChimName(Chimrecord);
    int firstfrontindex = 0;
    ReadRec_t lastreadrec;
    BamReader bamreader;
    bamreader.Open(bamfile);
    if (!bamreader.IsOpen()) return;
    BamAlignment record;
    const int N = (int)vNodes.size();
    while (bamreader.GetNextAlignment(record)) {
        if (RecordFiltered(record, ChimName, false)) continue;
        ReadRec_t readrec(record, P);
        readrec.SortbyReadPos();
        AppendMateStub(record, readrec);
        if (ReadRec_t::Equal(lastreadrec, readrec)) continue;
        else lastreadrec = readrec;
        bool whetherbuildedge = false;
        if (readrec.FirstRead.size() == 0 || readrec.SecondMate.size() == 0) whetherbuildedge = true;
        else if ((readrec.FirstRead.front().ReadPos <= 15 || readrec.FirstLowPhred) && (readrec.SecondMate.front().ReadPos <= 15 || readrec.SecondLowPhred)) whetherbuildedge = true;
        if (!whetherbuildedge) continue;
        std::vector<int> RN = LocateRead(firstfrontindex, readrec);
        if (RN.size() != 0 && RN[0] != -1) firstfrontindex = RN[0];
        const int nf = (int)readrec.FirstRead.size();
        auto checked = [&](const Edge_t& e) {
            if (!(e.Ind1 >= 0 && e.Ind1 < N && e.Ind2 >= 0 && e.Ind2 < N)) { std::cerr << "oracle: edge index assertion of SegmentGraph.cpp:1617 fails (reference aborts)\n"; std::exit(4); }
            vEdges.push_back(e);
        };
        for (int k = 0; k < (int)RN.size(); k++)
            if (RN[k] == -1) {
                const SingleBamRec_t& b = k < nf ? readrec.FirstRead[k] : readrec.SecondMate[k - nf];
                int i = HomeNode(firstfrontindex, b);
                checked(Edge_t(i, false, i + 1, true));
            }
        auto splitedges = [&](const std::vector<SingleBamRec_t>& R, int base) {  // :1631-1653
            if (R.size() == 0) return;
            for (int k = 0; k < (int)R.size() - 1; k++) {
                int i = RN[base + k], j = RN[base + k + 1];
                if (i != j && i != -1 && j != -1) checked(Edge_t(i, R[k].IsReverse, j, !R[k + 1].IsReverse, 1));
            }
        };
        splitedges(readrec.FirstRead, 0);
        splitedges(readrec.SecondMate, nf);
        if (record.IsFirstMate() && readrec.FirstRead.size() > 0 && readrec.SecondMate.size() > 0) {  // :1655-1685
            if (!readrec.IsSingleAnchored() && !readrec.IsEndDiscordant(true) && !readrec.IsEndDiscordant(false)) {
                int i = RN[nf - 1], j = RN.back();
                bool isoverlap = detail::PairOverlap(readrec, RN, i, j);
                if (i != j && i != -1 && j != -1 && !isoverlap) {
                    Edge_t tmp(i, readrec.FirstRead.back().IsReverse, j, readrec.SecondMate.back().IsReverse, 1);
                    if (!(tmp.Ind1 >= 0 && tmp.Ind1 < N && tmp.Ind2 >= 0 && tmp.Ind2 < N)) { std::cerr << "oracle: edge index assertion fails\n"; std::exit(4); }
                    if (readrec.IsPairDiscordant(false) == IsDiscordant(tmp)) vEdges.push_back(tmp);
                }
            }
        }
    }
}

// src/SegmentGraph.cpp:1932-1966
inline void SegmentGraph_t::BuildEdges(SBamrecord_t& Chimrecord, const std::string& bamfile) {
    if (P.UsingSTAR) {
        RawEdgesChim(Chimrecord);
        RawEdgesOther(Chimrecord, bamfile);
    } else RawEdges(Chimrecord, bamfile);
    std::sort(vEdges.begin(), vEdges.end());
    std::vector<Edge_t> tmpEdges;
    for (size_t i = 0; i < vEdges.size(); i++) {
        if (tmpEdges.size() == 0 || !(vEdges[i] == tmpEdges.back())) tmpEdges.push_back(vEdges[i]);
        else tmpEdges.back().Weight += vEdges[i].Weight;
    }
    vEdges.clear();
    for (const Edge_t& e : tmpEdges) if (e.Weight > 0) vEdges.push_back(e);
    UpdateNodeLink();
}

// src/SegmentGraph.cpp:2894-2909
inline void SegmentGraph_t::UpdateNodeLink() {
    for (Node_t& n : vNodes) { n.HeadEdges.clear(); n.TailEdges.clear(); }
    for (size_t i = 0; i < vEdges.size(); i++) {
        const Edge_t& e = vEdges[i];
        (e.Head1 ? vNodes[e.Ind1].HeadEdges : vNodes[e.Ind1].TailEdges).push_back((int)i);
        (e.Head2 ? vNodes[e.Ind2].HeadEdges : vNodes[e.Ind2].TailEdges).push_back((int)i);
    }
}

// =====================================================================================================
// FilterbyWeight (src/SegmentGraph.cpp:1968-2123) -- index typos of ledger B14 kept verbatim
// =====================================================================================================
inline void SegmentGraph_t::FilterbyWeight() {
    const int DI = P.Concord_Dist_Idx, DP = P.Concord_Dist_Pos;
    const int relaxedweight = P.Min_Edge_Weight - 2;
    const int E = (int)vEdges.size();
    std::vector<bool> HasInspected(E, false);
    auto endpos1 = [&](const Edge_t& e) { return e.Head1 ? vNodes[e.Ind1].Position : vNodes[e.Ind1].Position + vNodes[e.Ind1].Length; };
    auto endpos2 = [&](const Edge_t& e) { return e.Head2 ? vNodes[e.Ind2].Position : vNodes[e.Ind2].Position + vNodes[e.Ind2].Length; };
    for (int i = 0; i < E; i++) {
        if (HasInspected[i]) continue;
        const Edge_t ei = vEdges[i];
        int chr1 = vNodes[ei.Ind1].Chr, chr2 = vNodes[ei.Ind2].Chr;
        std::vector<int> NearbyIdx;
        NearbyIdx.push_back(i);
        HasInspected[i] = true;
        if (ei.Head1 || !ei.Head2 || chr1 != chr2) {
            // two orientation classes, each with its own running ranges: [0]=same orientation, [1]=opposite
            struct Range { pii Idx1, Pos1, Idx2, Pos2; } R[2];
            R[0].Idx1 = pii(ei.Ind1, ei.Ind1); R[0].Pos1 = pii(endpos1(ei), endpos1(ei));
            R[0].Idx2 = pii(ei.Ind2, ei.Ind2); R[0].Pos2 = pii(endpos2(ei), endpos2(ei));
            R[1] = R[0];
            bool longconnectiongroup = false;
            auto cls = [&](const Edge_t& ej) { return (ej.Head1 == ei.Head1 && ej.Head2 == ei.Head2) ? 0 : ((ej.Head1 != ei.Head1 && ej.Head2 != ei.Head2) ? 1 : -1); };
            for (int j = i - 1; j > -1 && vNodes[vEdges[j].Ind1].Chr == chr1; j--) {  // :2000-2035
                const Edge_t& ej = vEdges[j];
                int newpos1 = endpos1(ej), newpos2 = endpos2(ej);
                if ((ei.Ind1 < std::min(R[0].Idx1.first, R[1].Idx1.first) - DI) || (newpos1 < std::min(R[0].Pos1.first, R[1].Pos1.first) - DP)) break;  // ei.Ind1: B14
                int c = cls(ej);
                if (c < 0) continue;
                Range& r = R[c];
                if (IsDiscordant(j) && ej.Ind2 >= r.Idx2.first - DI && ei.Ind2 <= r.Idx2.second + DI && newpos2 >= r.Pos2.first - DP && newpos2 <= r.Pos2.second + DP) {  // ei.Ind2: B14
                    NearbyIdx.push_back(j);
                    r.Idx1.first = std::min(r.Idx1.first, ej.Ind1);
                    r.Pos1.first = std::min(r.Pos1.first, newpos1);
                    r.Idx2.first = std::min(r.Idx2.first, ej.Ind2); r.Idx2.second = std::max(r.Idx2.second, ej.Ind2);
                    r.Pos2.first = std::min(r.Pos2.first, newpos2); r.Pos2.second = std::max(r.Pos2.second, newpos2);
                    if (r.Idx1.second >= r.Idx2.first) longconnectiongroup = true;
                }
            }
            for (int j = i + 1; j < E && vNodes[vEdges[j].Ind1].Chr == chr1; j++) {  // :2036-2070
                const Edge_t& ej = vEdges[j];
                int newpos1 = endpos1(ej), newpos2 = endpos2(ej);
                if ((ej.Ind1 > std::max(R[0].Idx1.second, R[1].Idx1.second) + DI) || (newpos1 > std::max(R[0].Pos1.second, R[1].Pos1.second) + DP)) break;
                int c = cls(ej);
                if (c < 0) continue;
                Range& r = R[c];
                // same-orientation class tests ej.Ind2 on both sides (:2042); opposite class tests ei.Ind2 on the upper side (:2056, B14)
                int upperInd2 = (c == 0) ? ej.Ind2 : ei.Ind2;
                if (IsDiscordant(j) && ej.Ind2 >= r.Idx2.first - DI && upperInd2 <= r.Idx2.second + DI && newpos2 >= r.Pos2.first - DP && newpos2 <= r.Pos2.second + DP) {
                    NearbyIdx.push_back(j);
                    if (c == 0) {  // :2045-2046 grow the upper side
                        r.Idx1.second = std::max(r.Idx1.second, ej.Ind1);
                        r.Pos1.second = std::max(r.Pos1.second, newpos1);
                    } else {       // :2060-2061 the opposite class still updates the LOWER side here (kept)
                        r.Idx1.first = std::min(r.Idx1.first, ej.Ind1);
                        r.Pos1.first = std::min(r.Pos1.first, newpos1);
                    }
                    r.Idx2.first = std::min(r.Idx2.first, ej.Ind2); r.Idx2.second = std::max(r.Idx2.second, ej.Ind2);
                    r.Pos2.first = std::min(r.Pos2.first, newpos2); r.Pos2.second = std::max(r.Pos2.second, newpos2);
                    if (r.Idx1.second >= r.Idx2.first) longconnectiongroup = true;
                }
            }
            std::sort(NearbyIdx.begin(), NearbyIdx.end());
            NearbyIdx.resize(std::distance(NearbyIdx.begin(), std::unique(NearbyIdx.begin(), NearbyIdx.end())));
            if (!longconnectiongroup) {
                int sumweight = 0;
                for (int k : NearbyIdx) sumweight += vEdges[k].Weight;
                for (int k : NearbyIdx) { vEdges[k].GroupWeight = std::max(vEdges[k].GroupWeight, sumweight); HasInspected[k] = true; }
            } else {
                for (int k : NearbyIdx) { vEdges[k].GroupWeight = vEdges[k].Weight; HasInspected[k] = true; }
            }
        } else {  // concordant-type edge (:2090-2111)
            int pos1 = endpos1(ei), pos2 = endpos2(ei);
            auto near = [&](const Edge_t& ej) {
                int newchr1 = vNodes[ej.Ind1].Chr, newchr2 = vNodes[ej.Ind2].Chr;
                return ei.Head1 == ej.Head1 && ei.Head2 == ej.Head2 && newchr1 == chr1 && newchr2 == chr2 && std::abs(ej.Ind2 - ei.Ind2) <= DI &&
                       std::abs(endpos1(ej) - pos1) <= DP && std::abs(endpos2(ej) - pos2) <= DP;
            };
            for (int j = i - 1; j > -1 && vEdges[j].Ind1 >= ei.Ind1 - DI && vNodes[vEdges[j].Ind1].Chr == chr1 && vNodes[vEdges[j].Ind1].Position + vNodes[vEdges[j].Ind1].Length >= pos1 - DP; j--)
                if (vEdges[j].Ind2 > ei.Ind1 && near(vEdges[j])) NearbyIdx.push_back(j);
            for (int j = i + 1; j < E && vEdges[j].Ind1 <= ei.Ind1 + DI && vNodes[vEdges[j].Ind1].Chr == chr1 && vNodes[vEdges[j].Ind1].Position <= pos1 + DP; j++)
                if (vEdges[j].Ind1 < ei.Ind2 && near(vEdges[j])) NearbyIdx.push_back(j);
            std::sort(NearbyIdx.begin(), NearbyIdx.end());
            NearbyIdx.resize(std::distance(NearbyIdx.begin(), std::unique(NearbyIdx.begin(), NearbyIdx.end())));
            int sumweight = 0;
            for (int k : NearbyIdx) sumweight += vEdges[k].Weight;
            vEdges[i].GroupWeight = sumweight;
        }
    }
    std::vector<Edge_t> tmpEdges;
    for (const Edge_t& e : vEdges) if (e.GroupWeight > relaxedweight) tmpEdges.push_back(e);
    vEdges = tmpEdges;
    UpdateNodeLink();
}

// =====================================================================================================
// FilterbyInterleaving (src/SegmentGraph.cpp:2161-2277)
// =====================================================================================================
inline void SegmentGraph_t::FilterbyInterleaving(std::vector<bool>& KeepEdge) {
    const int DI = P.Concord_Dist_Idx, DP = P.Concord_Dist_Pos;
    const int E = (int)vEdges.size();
    std::vector<bool> HasInspected(E, false);
    KeepEdge.assign(E, true);
    auto endpos1 = [&](const Edge_t& e) { return e.Head1 ? vNodes[e.Ind1].Position : vNodes[e.Ind1].Position + vNodes[e.Ind1].Length; };
    auto endpos2 = [&](const Edge_t& e) { return e.Head2 ? vNodes[e.Ind2].Position : vNodes[e.Ind2].Position + vNodes[e.Ind2].Length; };
    for (int i = 0; i < E; i++) {
        if (HasInspected[i]) continue;
        const Edge_t& ei = vEdges[i];
        if (ei.Ind2 - ei.Ind1 <= DI || (vNodes[ei.Ind1].Chr == vNodes[ei.Ind2].Chr && std::abs(vNodes[ei.Ind1].Position - vNodes[ei.Ind2].Position) <= DP)) {
            HasInspected[i] = true; KeepEdge[i] = true;
            continue;
        }
        int chr1 = vNodes[ei.Ind1].Chr;
        int minpos1 = endpos1(ei), maxpos1 = minpos1, minidx1 = ei.Ind1, maxidx1 = ei.Ind1;
        int minpos2 = endpos2(ei), maxpos2 = minpos2, minidx2 = ei.Ind2, maxidx2 = ei.Ind2;
        bool longconcordgroup = false;
        std::vector<int> NearbyIdx;
        NearbyIdx.push_back(i);
        for (int j = i - 1; j > -1 && vNodes[vEdges[j].Ind1].Chr == chr1; j--) {
            const Edge_t& ej = vEdges[j];
            int newpos1 = endpos1(ej), newpos2 = endpos2(ej);
            if ((ei.Ind1 < minidx1 - DI) || (newpos1 < minpos1 - DP)) break;  // ei.Ind1: B14
            if (ej.Ind2 >= minidx2 - DI && ei.Ind2 <= maxidx2 + DI && newpos2 >= minpos2 - DP && newpos2 <= maxpos2 + DP) {  // ei.Ind2: B14
                NearbyIdx.push_back(j);
                minidx1 = std::min(minidx1, ej.Ind1); minpos1 = std::min(minpos1, newpos1);
                minidx2 = std::min(minidx2, ej.Ind2); maxidx2 = std::max(maxidx2, ej.Ind2);
                minpos2 = std::min(minpos2, newpos2); maxpos2 = std::max(maxpos2, newpos2);
                if (maxidx1 >= minidx2) { longconcordgroup = true; break; }
            }
        }
        for (int j = i + 1; j < E && vNodes[vEdges[j].Ind1].Chr == chr1; j++) {
            const Edge_t& ej = vEdges[j];
            int newpos1 = endpos1(ej), newpos2 = endpos2(ej);
            if ((ej.Ind1 > maxidx1 + DI) || (newpos1 > maxpos1 + DP)) break;
            if (ej.Ind2 >= minidx2 - DI && ej.Ind2 <= maxidx2 + DI && newpos2 >= minpos2 - DP && newpos2 <= maxpos2 + DP) {
                NearbyIdx.push_back(j);
                maxidx1 = std::max(maxidx1, ej.Ind1); maxpos1 = std::max(maxpos1, newpos1);
                minidx2 = std::min(minidx2, ej.Ind2); maxidx2 = std::max(maxidx2, ej.Ind2);
                minpos2 = std::min(minpos2, newpos2); maxpos2 = std::max(maxpos2, newpos2);
                if (maxidx1 >= minidx2) { longconcordgroup = true; break; }
            }
        }
        if (longconcordgroup) {
            for (int k : NearbyIdx) HasInspected[k] = true;
            continue;
        }
        std::sort(NearbyIdx.begin(), NearbyIdx.end());
        // extreme partner index per (group, end) -- value-initialised pairs are (0,0) when a side is empty
        bool has[4] = {false, false, false, false};
        pii ext[4];  // 0: Ind1 group via head, 1: Ind1 group via tail, 2: Ind2 group via head, 3: Ind2 group via tail
        auto upd = [&](int w, int v) {
            if (!has[w]) { ext[w] = pii(v, v); has[w] = true; }
            else { ext[w].first = std::min(ext[w].first, v); ext[w].second = std::max(ext[w].second, v); }
        };
        for (int k : NearbyIdx) {
            const Edge_t& e = vEdges[k];
            upd(e.Head1 ? 0 : 1, e.Ind2);
            upd(e.Head2 ? 2 : 3, e.Ind1);
        }
        // stray ';' at :2265 makes overlapInd1 unconditional (ledger B14)
        bool overlapInd1 = (std::min(ext[0].second, ext[1].second) >= std::max(ext[0].first, ext[1].first));
        bool overlapInd2 = false;
        if (has[2] && has[3]) overlapInd2 = (std::min(ext[2].second, ext[3].second) >= std::max(ext[2].first, ext[3].first));
        if (overlapInd1 && overlapInd2)
            for (int k : NearbyIdx) KeepEdge[k] = false;
        for (int k : NearbyIdx) HasInspected[k] = true;
    }
}

// =====================================================================================================
// GroupConnection / GroupSelect / FilterEdges (src/SegmentGraph.cpp:2394-2526)
// =====================================================================================================
inline int SegmentGraph_t::GroupConnection(int node, const std::vector<int>& Edges, int sumweight, std::vector<int>& Connection, std::vector<int>& Lab) const {
    const int DP = P.Concord_Dist_Pos;
    Connection.clear();
    int count = 0, mindist = -1, index = -1;
    for (int ei : Edges) {
        const Edge_t& e = vEdges[ei];
        if (e.GroupWeight > 0.01 * sumweight || e.GroupWeight > P.Min_Edge_Weight) Connection.push_back((e.Ind1 != node) ? e.Ind1 : e.Ind2);
    }
    std::sort(Connection.begin(), Connection.end());
    Lab.assign(Connection.size(), -1);
    auto gap = [&](int a, int b) { return vNodes[b].Position - vNodes[a].Position - vNodes[a].Length; };  // from end of a to start of b
    for (int i = 0; i < (int)Connection.size(); i++)
        if (vNodes[Connection[i]].Chr == vNodes[node].Chr && gap(Connection[i], node) <= DP && gap(node, Connection[i]) <= DP) {
            if (mindist == -1 || mindist > std::abs(node - Connection[i])) { mindist = std::abs(node - Connection[i]); index = i; }
        }
    if (index != -1) {
        Lab[index] = 0;
        for (int i = index + 1; i < (int)Connection.size(); i++)
            if (vNodes[Connection[i]].Chr == vNodes[node].Chr && gap(Connection[i - 1], Connection[i]) <= DP) Lab[i] = 0;
            else break;
        for (int i = index - 1; i >= 0; i--)
            if (vNodes[Connection[i]].Chr == vNodes[node].Chr && gap(Connection[i], Connection[i + 1]) <= DP) Lab[i] = 0;
            else break;
    }
    if (Lab.size() != 0) {
        count = (Lab[0] == -1) ? 1 : 0;
        if (Lab[0] == -1) Lab[0] = 1;
        for (int i = 1; i < (int)Connection.size(); i++) {
            if (Lab[i] != -1) continue;
            else if (vNodes[Connection[i]].Chr != vNodes[Connection[i - 1]].Chr || gap(Connection[i - 1], Connection[i]) > DP) count++;
            Lab[i] = count;
        }
    }
    return count;
}

inline void SegmentGraph_t::GroupSelect(int node, const std::vector<int>& Edges, int sumweight, int count, std::vector<int>& Connection, std::vector<int>& Lab,
                                        std::vector<Edge_t>& ToDelete) const {
    std::vector<int> LabelWeight(count + 1, 0);
    auto strong = [&](const Edge_t& e) { return e.GroupWeight > 0.01 * sumweight || e.GroupWeight > P.Min_Edge_Weight; };
    auto labelof = [&](const Edge_t& e) {
        int mateNode = (e.Ind1 != node) ? e.Ind1 : e.Ind2;
        return Lab[std::distance(Connection.begin(), std::find(Connection.begin(), Connection.end(), mateNode))];
    };
    for (int ei : Edges) if (strong(vEdges[ei])) LabelWeight[labelof(vEdges[ei])] += vEdges[ei].Weight;
    int maxLabel = 1;
    for (int i = 1; i < (int)LabelWeight.size(); i++) if (LabelWeight[i] > LabelWeight[maxLabel]) maxLabel = i;
    for (int ei : Edges)
        if (strong(vEdges[ei])) {
            int l = labelof(vEdges[ei]);
            if (l != maxLabel && l != 0) ToDelete.push_back(vEdges[ei]);
        }
}

inline void SegmentGraph_t::FilterEdges(const std::vector<bool>& KeepEdge) {
    const int MEW = P.Min_Edge_Weight;
    std::vector<int> BadNodes;
    std::vector<Edge_t> ToDelete;
    for (int i = 0; i < (int)vNodes.size(); i++) {
        const std::vector<int>&HE = vNodes[i].HeadEdges, &TE = vNodes[i].TailEdges;
        int headweight = 0, tailweight = 0, sumweight = 0;
        for (int e : HE) headweight += vEdges[e].Weight;
        for (int e : TE) tailweight += vEdges[e].Weight;
        sumweight = headweight + tailweight;
        auto weak = [&](const Edge_t& e) { return e.GroupWeight <= 0.01 * sumweight && e.GroupWeight <= MEW; };
        for (int e : HE) if (weak(vEdges[e])) ToDelete.push_back(vEdges[e]);
        for (int e : TE) if (weak(vEdges[e])) ToDelete.push_back(vEdges[e]);
        std::vector<int> HeadConn, TailConn, HeadLabel, TailLabel;
        int headcount = 0, tailcount = 0;
        if (HE.size() != 0) headcount = GroupConnection(i, HE, sumweight, HeadConn, HeadLabel);
        if (TE.size() != 0) tailcount = GroupConnection(i, TE, sumweight, TailConn, TailLabel);
        if (headcount + tailcount >= P.MaxAllowedDegree) BadNodes.push_back(i);
        else {
            if (headcount > 1) GroupSelect(i, HE, sumweight, headcount, HeadConn, HeadLabel, ToDelete);
            else for (int e : HE) if (!weak(vEdges[e]) && vEdges[e].GroupWeight < 0.01 * headweight) ToDelete.push_back(vEdges[e]);
            if (tailcount > 1) GroupSelect(i, TE, sumweight, tailcount, TailConn, TailLabel, ToDelete);
            else for (int e : TE) if (!weak(vEdges[e]) && vEdges[e].GroupWeight < 0.01 * tailweight) ToDelete.push_back(vEdges[e]);
        }
    }
    std::sort(ToDelete.begin(), ToDelete.end());
    std::sort(BadNodes.begin(), BadNodes.end());
    std::vector<Edge_t> tmpEdges;
    for (size_t i = 0; i < vEdges.size(); i++) {
        const Edge_t& e = vEdges[i];
        bool cond1 = false, cond2 = true;
        if (!std::binary_search(BadNodes.begin(), BadNodes.end(), e.Ind1) && !std::binary_search(BadNodes.begin(), BadNodes.end(), e.Ind2) && e.GroupWeight > MEW) cond1 = true;
        else if (vNodes[e.Ind1].Chr == vNodes[e.Ind2].Chr && std::abs(vNodes[e.Ind2].Position - vNodes[e.Ind1].Position - vNodes[e.Ind1].Length) <= P.Concord_Dist_Pos && e.GroupWeight > MEW) cond1 = true;
        if (cond1 && (e.Ind2 - e.Ind1 > P.Concord_Dist_Idx || e.Head1 != false || e.Head2 != true)) {
            double cov1 = vNodes[e.Ind1].AvgDepth, cov2 = vNodes[e.Ind2].AvgDepth;
            double ratio = (cov1 > cov2) ? cov1 / cov2 : cov2 / cov1;  // x/0 -> inf deletes, 0/0 -> NaN keeps
            if ((e.Weight <= MEW + 2 && ratio > 3) || (e.Weight > MEW + 2 && ratio > 50)) cond2 = false;
        }
        if (KeepEdge[i] && cond1 && cond2) tmpEdges.push_back(e);
    }
    std::sort(tmpEdges.begin(), tmpEdges.end());
    std::vector<Edge_t> out(tmpEdges.size());
    out.resize(std::distance(out.begin(), std::set_difference(tmpEdges.begin(), tmpEdges.end(), ToDelete.begin(), ToDelete.end(), out.begin())));
    vEdges = out;
    UpdateNodeLink();
}

// =====================================================================================================
// CompressNode (src/SegmentGraph.cpp:2528-2604)
// =====================================================================================================
inline void SegmentGraph_t::CompressNode() {
    std::vector<int> LinkedNode;
    for (const Edge_t& e : vEdges) { LinkedNode.push_back(e.Ind1); LinkedNode.push_back(e.Ind2); }
    if (LinkedNode.size() == 0) { std::cout << "Error: 0 nodes are connected by edges.\n"; std::cerr << "oracle: reference asserts here (SegmentGraph.cpp:2537)\n"; std::exit(5); }
    std::sort(LinkedNode.begin(), LinkedNode.end());
    LinkedNode.resize(std::distance(LinkedNode.begin(), std::unique(LinkedNode.begin(), LinkedNode.end())));
    std::vector<Node_t> newNodes;
    int count = 0;
    std::map<int, int> LinkedOld_New;
    auto mergerun = [&](int lo, int hi) {  // one merged node for unlinked nodes [lo,hi) on one chromosome
        Node_t tmp(vNodes[lo].Chr, vNodes[lo].Position, vNodes[hi - 1].Position + vNodes[hi - 1].Length - vNodes[lo].Position, 0);
        for (int k = lo; k < hi; k++) { tmp.Support += vNodes[k].Support; tmp.AvgDepth += vNodes[k].AvgDepth * vNodes[k].Length; }
        tmp.AvgDepth /= tmp.Length;
        newNodes.push_back(tmp); count++;
    };
    auto gaprange = [&](int startidx, int endidx) {  // :2547-2569 / :2574-2596
        int lastinsert = startidx;
        for (int j = startidx; j < endidx; j++)
            if (vNodes[j].Chr != vNodes[lastinsert].Chr) { mergerun(lastinsert, j); lastinsert = j; }
        if (lastinsert != endidx) mergerun(lastinsert, endidx);
    };
    for (size_t i = 0; i < LinkedNode.size(); i++) {
        int startidx = (i == 0) ? 0 : (LinkedNode[i - 1] + 1), endidx = LinkedNode[i];
        gaprange(startidx, endidx);
        newNodes.push_back(vNodes[endidx]); count++;
        LinkedOld_New[endidx] = count - 1;
    }
    if (LinkedNode.back() != (int)vNodes.size() - 1) gaprange(LinkedNode.back() + 1, (int)vNodes.size());
    for (Edge_t& e : vEdges) { e.Ind1 = LinkedOld_New[e.Ind1]; e.Ind2 = LinkedOld_New[e.Ind2]; }
    vNodes = newNodes;
    UpdateNodeLink();
}

// =====================================================================================================
// FurtherCompressNode (src/SegmentGraph.cpp:2693-2892)
// =====================================================================================================
inline void SegmentGraph_t::FurtherCompressNode() {
    const int DI = P.Concord_Dist_Idx;
    const int N = (int)vNodes.size();
    std::vector<int> MergeNode(N, -1);
    int curNode = 0, rightmost = 0;
    auto samehd = [](const Edge_t& a, const Edge_t& b) { return a.Head1 == b.Head1 && a.Head2 == b.Head2; };
    auto closeidx = [&](const Edge_t& a, const Edge_t& b) { return std::abs(a.Ind1 - b.Ind1) <= DI && std::abs(a.Ind2 - b.Ind2) <= DI; };
    auto samechr = [&](const Edge_t& a, const Edge_t& b) { return vNodes[a.Ind1].Chr == vNodes[b.Ind1].Chr && vNodes[a.Ind2].Chr == vNodes[b.Ind2].Chr; };
    auto discordantOf = [&](int n, std::vector<Edge_t>& out, bool trackRight) {
        for (int e : vNodes[n].HeadEdges) { if (IsDiscordant(e)) out.push_back(vEdges[e]); else if (trackRight) rightmost = std::max(rightmost, std::max(vEdges[e].Ind1, vEdges[e].Ind2)); }
        for (int e : vNodes[n].TailEdges) { if (IsDiscordant(e)) out.push_back(vEdges[e]); else if (trackRight) rightmost = std::max(rightmost, std::max(vEdges[e].Ind1, vEdges[e].Ind2)); }
    };
    // first following node (same chr, < i+20, before the nearest discordant partner) that has discordant edges
    auto nextwith = [&](int i, int minDisInd2, std::vector<Edge_t>& nextDis) -> int {
        int j = i + 1;
        for (; j < N && j < i + 20 && j < minDisInd2 && vNodes[i].Chr == vNodes[j].Chr; j++) {
            discordantOf(j, nextDis, false);
            if (nextDis.size() != 0) break;
        }
        return j;
    };
    auto crossmatch = [&](const std::vector<Edge_t>& A, const std::vector<Edge_t>& B) -> bool {  // :2770-2786 / :2833-2849
        std::vector<bool> aEQ(A.size(), false), bEQ(B.size(), false);
        for (size_t k = 0; k < A.size(); k++)
            for (size_t l = 0; l < B.size(); l++)
                if (A[k].Ind2 > B[l].Ind1 && B[l].Ind2 > A[k].Ind1 && samechr(A[k], B[l]) && closeidx(A[k], B[l]) && samehd(A[k], B[l])) { aEQ[k] = true; bEQ[l] = true; }
        for (bool b : aEQ) if (!b) return false;
        for (bool b : bEQ) if (!b) return false;
        return true;
    };
    for (int i = 0; i < N; i++) {
        int minDisInd2 = i + 20;  // only meaningful when thisDis is non-empty (the reference leaves it unset otherwise)
        std::vector<Edge_t> thisDis, tmpDis;
        if (i != 0 && vNodes[i].Chr != vNodes[i - 1].Chr && curNode == MergeNode[i - 1]) curNode++;
        discordantOf(i, thisDis, true);
        if (thisDis.size() != 0) {  // :2714-2730
            minDisInd2 = (thisDis[0].Ind1 == i) ? thisDis[0].Ind2 : (i + 20);
            tmpDis.push_back(thisDis[0]);
            for (size_t k = 0; k + 1 < thisDis.size(); k++) {
                const Edge_t &e1 = thisDis[k], &e2 = thisDis[k + 1];
                bool samegroup = ((e1.Ind1 == i && e2.Ind1 == i) || (e1.Ind2 == i && e2.Ind2 == i));
                if (!(closeidx(e1, e2) && samehd(e1, e2))) samegroup = false;
                if (!samegroup) tmpDis.push_back(e2);
                int tmpmin = (e2.Ind1 == i) ? e2.Ind2 : (i + 20);
                if (tmpmin < minDisInd2) minDisInd2 = tmpmin;
            }
            thisDis = tmpDis;
        }
        if (MergeNode[i] == -1) {
            if (thisDis.size() == 0 && i < rightmost) MergeNode[i] = curNode;
            else if (thisDis.size() == 0 && i == rightmost) { MergeNode[i] = curNode; curNode++; rightmost++; }
            else {  // :2740-2797
                if (i != 0 && curNode == MergeNode[i - 1]) curNode++;
                std::vector<Edge_t> nextDis;
                int j = nextwith(i, minDisInd2, nextDis);
                bool equivalent = (nextDis.size() != 0);
                if (nextDis.size() != 0) {
                    tmpDis.clear();
                    tmpDis.push_back(nextDis[0]);
                    for (size_t k = 0; k + 1 < nextDis.size(); k++) {
                        const Edge_t &e1 = nextDis[k], &e2 = nextDis[k + 1];
                        bool samegroup = ((e1.Ind1 == j && e2.Ind1 == j) || (e1.Ind2 == j && e2.Ind2 == j));
                        if (!(closeidx(e1, e2) && samechr(e1, e2) && samehd(e1, e2))) samegroup = false;
                        if (!samegroup) tmpDis.push_back(e2);
                    }
                    nextDis = tmpDis;
                    if (!crossmatch(thisDis, nextDis)) equivalent = false;
                }
                if (!equivalent) { MergeNode[i] = curNode; curNode++; }
                else for (int k = i; k <= j; k++) MergeNode[k] = curNode;
                rightmost = i + 1;
            }
        } else if (thisDis.size() != 0) {  // :2799-2858
            std::vector<Edge_t> nextDis;
            int j = nextwith(i, minDisInd2, nextDis);
            bool equivalent = (nextDis.size() != 0);
            if (nextDis.size() != 0) {
                auto squeeze = [&](std::vector<Edge_t>& V) {
                    std::vector<Edge_t> t;
                    t.push_back(V[0]);
                    for (size_t k = 0; k + 1 < V.size(); k++)
                        if (!(closeidx(V[k], V[k + 1]) && samechr(V[k], V[k + 1]) && samehd(V[k], V[k + 1]))) t.push_back(V[k + 1]);
                    V = t;
                };
                squeeze(thisDis);
                squeeze(nextDis);
                if (!crossmatch(thisDis, nextDis)) equivalent = false;
            }
            if (!equivalent) curNode++;
            else for (int k = i; k <= j; k++) MergeNode[k] = curNode;
            rightmost = i + 1;
        }
    }
    for (int i = 0; i + 1 < N; i++)
        if (!((MergeNode[i] == MergeNode[i + 1]) || (MergeNode[i] + 1 == MergeNode[i + 1]))) { std::cerr << "oracle: MergeNode assertion of SegmentGraph.cpp:2862 fails (reference aborts)\n"; std::exit(4); }
    std::vector<Node_t> newvNodes;
    std::vector<Edge_t> newvEdges;
    int ind = 0;
    while (ind < N) {
        int j = ind;
        for (; j < N && MergeNode[j] == MergeNode[ind]; j++) {}
        newvNodes.push_back(Node_t(vNodes[ind].Chr, vNodes[ind].Position, vNodes[j - 1].Position + vNodes[j - 1].Length - vNodes[ind].Position));  // Support/AvgDepth reset: ledger B15
        ind = j;
    }
    for (const Edge_t& edge : vEdges)
        if (MergeNode[edge.Ind1] != MergeNode[edge.Ind2]) newvEdges.push_back(Edge_t(MergeNode[edge.Ind1], edge.Head1, MergeNode[edge.Ind2], edge.Head2, edge.Weight));
    vNodes = newvNodes;
    vEdges.clear();
    std::sort(newvEdges.begin(), newvEdges.end());
    for (const Edge_t& e : newvEdges) {
        if (vEdges.size() == 0 || !(e == vEdges.back())) vEdges.push_back(e);
        else vEdges.back().Weight += e.Weight;
    }
    UpdateNodeLink();
}

// src/SegmentGraph.cpp:2911-2935, :2986-3003  (labels in order of smallest member node id)
inline void SegmentGraph_t::ConnectedComponent() {
    Label.assign(vNodes.size(), -1);
    int curlabelid = 0;
    for (int s = 0; s < (int)vNodes.size(); s++) {
        if (Label[s] != -1) continue;
        std::vector<int> Unvisited(1, s);
        while (Unvisited.size() != 0) {
            int v = Unvisited.back();
            Unvisited.pop_back();
            if (Label[v] != -1) continue;
            Label[v] = curlabelid;
            for (int e : vNodes[v].HeadEdges) { if (vEdges[e].Ind1 != v) Unvisited.push_back(vEdges[e].Ind1); else if (vEdges[e].Ind2 != v) Unvisited.push_back(vEdges[e].Ind2); }
            for (int e : vNodes[v].TailEdges) { if (vEdges[e].Ind1 != v) Unvisited.push_back(vEdges[e].Ind1); else if (vEdges[e].Ind2 != v) Unvisited.push_back(vEdges[e].Ind2); }
        }
        curlabelid++;
    }
}

// src/SegmentGraph.cpp:3005-3017 (cast quirk: ledger B16)
inline void SegmentGraph_t::MultiplyDisEdges() {
    for (Edge_t& e : vEdges) if (IsDiscordant(e) && P.DiscordantRatio != 1) e.Weight = (int)P.DiscordantRatio * e.Weight;
}
inline void SegmentGraph_t::DeMultiplyDisEdges() {
    for (Edge_t& e : vEdges) if (IsDiscordant(e) && P.DiscordantRatio != 1) e.Weight = (int)(e.Weight / P.DiscordantRatio);
}

// src/SegmentGraph.cpp:3019-3081
inline void SegmentGraph_t::ExactBreakpoint(SBamrecord_t& Chimrecord, EdgeBPMap& ExactBP) const {
    ExactBP.clear();
    int firstfrontindex = 0;
    for (ReadRec_t& r : Chimrecord) {
        if (r.FirstRead.size() <= 1 && r.SecondMate.size() <= 1) continue;
        std::vector<int> RN = LocateRead(firstfrontindex, r);
        if (RN[0] != -1) firstfrontindex = RN[0];
        auto collect = [&](const std::vector<SingleBamRec_t>& R, int base) {
            if (R.size() <= 1) return;
            for (int k = 0; k < (int)R.size() - 1; k++) {
                int i = RN[base + k], j = RN[base + k + 1];
                if (i != j && i != -1 && j != -1) {
                    Edge_t tmp(i, R[k].IsReverse, j, !R[k + 1].IsReverse, 1);
                    if (IsDiscordant(tmp)) ExactBP[tmp].push_back(detail::SplitBreakpoints(R[k], R[k + 1]));
                }
            }
        };
        collect(r.FirstRead, 0);
        collect(r.SecondMate, (int)r.FirstRead.size());
    }
    for (EdgeBPMap::iterator it = ExactBP.begin(); it != ExactBP.end(); it++) CountTop(it->first, it->second);
}

// src/SegmentGraph.cpp:3083-3221
inline void SegmentGraph_t::ExactBPConcordantSupport(const std::string& Input_BAM, SBamrecord_t& Chimrecord, const EdgeBPMap& ExactBP, EdgeBPMap& Support) const {
    Support.clear();
    auto edgeBPs = [&](const Edge_t& e, std::vector<std::pair<pii, pii>>& out) {  // (chr,pos) pairs of one edge, :3093-3107
        EdgeBPMap::const_iterator itmap = ExactBP.find(e);
        if (itmap != ExactBP.end() && itmap->second.size() != 0) {
            for (const pii& p : itmap->second) out.push_back(std::make_pair(pii(vNodes[itmap->first.Ind1].Chr, p.first), pii(vNodes[itmap->first.Ind2].Chr, p.second)));
        } else {
            pii b1(vNodes[e.Ind1].Chr, vNodes[e.Ind1].Position), b2(vNodes[e.Ind2].Chr, vNodes[e.Ind2].Position);
            if (!e.Head1) b1.second += vNodes[e.Ind1].Length;
            if (!e.Head2) b2.second += vNodes[e.Ind2].Length;
            out.push_back(std::make_pair(b1, b2));
        }
    };
    std::vector<pii> BPs;
    for (const Edge_t& e : vEdges) {
        std::vector<std::pair<pii, pii>> v;
        edgeBPs(e, v);
        for (auto& p : v) { BPs.push_back(p.first); BPs.push_back(p.second); }
    }
    std::sort(BPs.begin(), BPs.end(), pii_less);
    std::vector<std::string> ChimName = BuildChimName(Chimrecord);
    std::vector<int> Coverages(BPs.size(), 0);
    size_t indBP = 0;
    BamReader bamreader;
    bamreader.Open(Input_BAM);
    if (bamreader.IsOpen()) {
        BamAlignment record;
        while (bamreader.GetNextAlignment(record)) {
            if (RecordFiltered(record, ChimName, true)) continue;
            if (record.IsMateMapped() && record.MateRefID == record.RefID && record.MatePosition > record.Position) continue;
            else if (record.IsMateMapped() && record.MateRefID == record.RefID && record.MatePosition == record.Position && record.IsSecondMate()) continue;
            if (indBP == BPs.size()) break;
            int alignChr = record.RefID, alignStart = record.Position, alignEnd = record.GetEndPosition();
            if (record.IsMateMapped() && record.MateRefID == record.RefID) alignStart = record.MatePosition;
            if (alignChr > BPs[indBP].first || (alignChr == BPs[indBP].first && alignStart > BPs[indBP].second + P.Concord_Dist_Pos)) indBP++;  // ledger B18
            for (size_t indBP2 = indBP; indBP2 < BPs.size(); indBP2++) {
                if (alignChr == BPs[indBP2].first && alignStart <= BPs[indBP2].second && alignEnd > BPs[indBP2].second) Coverages[indBP2]++;
                else if (alignChr < BPs[indBP2].first || (alignChr == BPs[indBP2].first && alignEnd <= BPs[indBP2].second)) break;
            }
        }
    }
    for (const Edge_t& e : vEdges) {
        std::vector<std::pair<pii, pii>> v;
        edgeBPs(e, v);
        std::vector<pii> supports;
        for (auto& p : v) {
            size_t i1 = std::lower_bound(BPs.begin(), BPs.end(), p.first, pii_less) - BPs.begin();
            size_t i2 = std::lower_bound(BPs.begin(), BPs.end(), p.second, pii_less) - BPs.begin();
            supports.push_back(pii(Coverages[i1], Coverages[i2]));
        }
        Support[e] = supports;
    }
}

}  // namespace oracle
