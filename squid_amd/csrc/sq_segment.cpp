// Genome segmentation (BuildNode_STAR of src/SegmentGraph.cpp:192-761).
//
// Split of work: the GPU (k_pass1, then k_tile_scan / k_zfinal / k_summarise_tiles: sq_pass_kernels.inc) classifies the concordant stream,
// writes 24-byte summaries (record key, first aligned block, classification) of just the stretches the host replays and -- because
// the discordant-cluster list is static -- also finds, tile by tile (dev_pass1, dev_segment_support): the trigger record of every cluster, every
// "zero coverage" record together with the running (otherChr, otherrightmost) pair in front of it, and per cluster
// the non-first blocks that can span one of its break candidates.  At a zero-coverage record the reference flushes
// a pending node end and empties both sliding windows (SegmentGraph.cpp:616-636), so the stream falls into
// stretches that are independent except for the list of nodes emitted so far; only stretches that contain a cluster
// trigger can emit anything.  The host replays just those stretches with the reference's control automaton (its
// ~15 carried variables, SURVEY.md A.2b, live in `Seg`) and never touches the rest of the stream.
#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>

#include "sq_internal.h"
#include "sq_parsort.h"

namespace sq {

namespace {

struct El {  // a window element = first block of a concordant record
    int32_t refid, refpos, matchref, readpos;
    bool rev;
};

struct Cluster { int ds, de, chr, start, right; };
// what the automaton only reads: fixed by the chimeric fragments (and, for rest_by_cluster, by GPU scans)
struct SegStatic {
    std::vector<Blk> D;              // sorted discordant blocks + zero sentinel at [nd] (ledger B21)
    int nd = 0;
    std::vector<std::pair<int, int>> part;  // PartAlignPos
    std::vector<Cluster> clusters;
    // live ConcordRest content per cluster as CSR, sorted by refpos inside a cluster (only ever counted); rest_max = longest block
    std::vector<int32_t> rest_off, rest_pos, rest_len, rest_max;
    // the clusters are fixed by the sorted discordant blocks alone (SegmentGraph.cpp:341-348 / 604-611)
    // (a cluster never crosses a chromosome: the runs of one RefID in the sorted list are walked side by side and strung together)
    void build_clusters(int RL, HostPool* pool = nullptr) {
        auto walk = [&](int s0, int end, std::vector<Cluster>& out) {
            while (s0 != end) {
                int right = D[s0].refpos + D[s0].matchref, e = s0;
                for (; e != end && D[e].refid == D[s0].refid && D[e].refpos < right + RL; ++e) right = std::max(right, D[e].refpos + D[e].matchref);
                out.push_back(Cluster{s0, e, D[s0].refid, D[s0].refpos, right});
                s0 = e;
            }
        };
        if (!pool || nd < 200000) { walk(0, nd, clusters); return; }
        std::vector<int> cut{0};  // first block of every chromosome: binary searches (the list is sorted by RefID first)
        while (cut.back() != nd) {
            const int rid = D[cut.back()].refid;
            int a = cut.back(), b = nd;
            while (a < b) { const int m = a + (b - a) / 2; if (D[m].refid <= rid) a = m + 1; else b = m; }
            cut.push_back(a);
        }
        std::vector<std::vector<Cluster>> per(cut.size() - 1);
        pool->parallel_for((int)per.size(), 1 << 20, [&](int k) { walk(cut[(size_t)k], cut[(size_t)k + 1], per[(size_t)k]); });
        for (const auto& v : per) clusters.insert(clusters.end(), v.begin(), v.end());
    }
};

// Number of blocks (pos[i], len[i]), i in [from, n), pos ascending, that span a break: pos < brk - T and
// pos + len >= brk + T -- for many nearby breaks.  The reference walks the whole list per break (SegmentGraph.cpp:
// 438-470); here the ends of the blocks that can span anything between the smallest and the largest break are sorted
// once, and a count is  #(pos < brk-T) - #(end < brk+T) + #(pos >= brk-T and end < brk+T), the last term over the
// few blocks that start within 2T of the break.
struct SpanIndex {
    const int32_t *pos = nullptr, *len = nullptr;
    int base = 0, lim = 0;
    mutable int cur = 0;           // first block with pos >= brk - T for the last break asked about (breaks come in ascending order)
    long long elo = 0;             // ends are counted per coordinate in [elo, elo + width)
    std::vector<int32_t> below;    // below[x] = #ends < elo + x   (ends behind the last break are never asked about: clamped)
    std::vector<int32_t> ends;     // fallback for very wide clusters: sorted ends
    bool wide = false;
    int lb(long long x, int a, int b) const { while (a < b) { int m = (a + b) >> 1; if (pos[m] < x) a = m + 1; else b = m; } return a; }
    void build(const int32_t* p, const int32_t* l, int from, int n, int maxlen, int minbrk, int maxbrk, int T) {
        pos = p; len = l;
        base = lb((long long)minbrk + T - maxlen, from, n);  // earlier blocks end before minbrk + T
        lim = lb((long long)maxbrk + T, base, n);            // later blocks start behind every break
        cur = base;
        elo = (long long)minbrk + T - maxlen;
        const long long width = (long long)maxbrk + T - elo + 2;
        wide = width > (1 << 16);
        if (wide) {
            ends.resize((size_t)(lim - base));
            for (int i = base; i < lim; ++i) ends[(size_t)(i - base)] = pos[i] + len[i];
            std::sort(ends.begin(), ends.end());
            return;
        }
        below.assign((size_t)width + 1, 0);
        for (int i = base; i < lim; ++i) {
            long long x = (long long)pos[i] + len[i] - elo;  // >= 1: pos >= elo and len >= 1... (a zero-length block still lands at 0)
            if (x < 0) x = 0;
            if (x >= width) x = width - 1;
            below[(size_t)x + 1]++;
        }
        for (size_t x = 1; x < below.size(); ++x) below[x] += below[x - 1];
    }
    int count(int brk, int T) const {
        const int a = brk - T, b = brk + T;
        while (cur < lim && pos[cur] < a) ++cur;
        const int ia = cur;
        int less_b;
        if (wide) less_b = (int)(std::lower_bound(ends.begin(), ends.end(), b) - ends.begin());
        else { long long x = (long long)b - elo; less_b = x <= 0 ? 0 : below[(size_t)std::min<long long>(x, (long long)below.size() - 1)]; }
        int tail = 0;
        for (int i = ia; i < lim && pos[i] < b; ++i) if (pos[i] + len[i] < b) ++tail;
        return (ia - base) - (less_b - tail);
    }
};

struct Seg {
    const sq_ctx* c;
    const StreamRec* recs;  // host copy of the stream summaries of the replayed stretches, one after the other
    int64_t shift = 0;      // kept index of recs[0] for the stretch being replayed
    const StreamRec& rec(int64_t idx) const { return recs[idx - shift]; }
    const int RL;
    static constexpr int T = 3;     // thresh (SegmentGraph.cpp:286)
    static constexpr int NEAR = 60;  // thresh*20
    const std::vector<Blk>& D;
    const int nd;
    const std::vector<std::pair<int, int>>& part;
    const std::vector<Cluster>& clusters;
    const SegStatic& st;
    std::vector<int32_t> cw, pw;     // windows: indices into in.recs (ConcordantCluster / PartialAlignCluster)
    int co = 0, po = 0;              // window offsets
    // The reference walks a whole window for every break candidate.  Here the live elements that can span anything
    // near the cluster are collected once per sub-cluster (one pass over contiguous arrays) and counted through a
    // SpanIndex; `maxm` bounds the block length.
    // per window: the elements whose block starts at the record position -- in stream order they are sorted by that start
    // -- and, apart, the others (first block of a spliced reverse read: an intron further right), which are few
    struct WinEnt { int32_t pos, len, rid, widx; };
    struct WinInfo { int maxm = 0; std::vector<WinEnt> norm, shifted; } ci, pi;
    void win_push(std::vector<int32_t>& W, WinInfo& wi, int32_t idx) {
        const StreamRec& r = rec(idx);
        wi.maxm = std::max(wi.maxm, r.fb_matchref);
        const WinEnt e{r.fb_refpos, r.fb_matchref, r.refid, (int32_t)W.size()};
        W.push_back(idx);
        if (r.fb_refpos == r.pos && (wi.norm.empty() || (wi.norm.back().rid == r.refid && wi.norm.back().pos <= r.pos))) wi.norm.push_back(e); else wi.shifted.push_back(e);
    }
    void win_clear(std::vector<int32_t>& W, WinInfo& wi, int& off) { W.clear(); off = 0; wi.norm.clear(); wi.shifted.clear(); wi.maxm = 0; }
    struct SpanSet { std::vector<std::pair<int32_t, int32_t>> tmp; std::vector<int32_t> pos, len; SpanIndex ix; };
    // blocks of the live window elements (window index >= off) on `chr` that start in [minbrk + T - maxm, maxbrk + T)
    void span_collect(SpanSet& ss, const WinInfo& wi, int off, int chr, int minbrk, int maxbrk) {
        const long long lo = (long long)minbrk + T - wi.maxm, hi = (long long)maxbrk + T;
        ss.tmp.clear();
        auto it = std::lower_bound(wi.norm.begin(), wi.norm.end(), lo, [](const WinEnt& a, long long b) { return a.pos < b; });
        for (; it != wi.norm.end() && it->pos < hi; ++it) if (it->widx >= off && it->rid == chr) ss.tmp.push_back(std::make_pair(it->pos, it->len));
        const size_t sorted_upto = ss.tmp.size();
        for (const WinEnt& e : wi.shifted) if (e.widx >= off && e.rid == chr && e.pos >= lo && e.pos < hi) ss.tmp.push_back(std::make_pair(e.pos, e.len));
        if (ss.tmp.size() != sorted_upto) {
            std::sort(ss.tmp.begin() + sorted_upto, ss.tmp.end());
            std::inplace_merge(ss.tmp.begin(), ss.tmp.begin() + sorted_upto, ss.tmp.end());
        }
        ss.pos.resize(ss.tmp.size()); ss.len.resize(ss.tmp.size());
        for (size_t i = 0; i < ss.tmp.size(); ++i) { ss.pos[i] = ss.tmp[i].first; ss.len[i] = ss.tmp[i].second; }
        ss.ix.build(ss.pos.data(), ss.len.data(), 0, (int)ss.pos.size(), wi.maxm, minbrk, maxbrk, T);
    }
    SpanSet cset, pset;
    const int32_t *rest_p = nullptr, *rest_l = nullptr;  // live ConcordRest content of the current cluster (refpos, matchref), sorted by refpos
    int rest_n = 0;
    std::vector<int> M_buf, fwd_buf, rev_buf;  // scratch of process_cluster
    double tsec[6] = {0, 0, 0, 0, 0, 0}; long long nsec[6] = {0, 0, 0, 0, 0, 0}; bool prof = false;
    int kc = -1;                     // index of the current cluster (ds == clusters[kc].ds)
    int rest_maxm = 0;
    SpanIndex rspan;
    std::vector<Node>& out;
    int ds = 0, de = 0, dcur = 0;    // itdisstart / itdisend / itdiscurrent
    size_t ps = 0, pe = 0;
    int disChr = 0, otherChr = 0, nextdisChr = 0, disright = 0, otherright = 0, nextdisright = 0;
    int markStart = -1, markChr = -1;
    const bool recount = std::getenv("SQUID_REPLAY_CHECK") != nullptr;  // (tests: every candidate counted twice, check_candidate)

    Seg(const sq_ctx* c, const StreamRec* recs, const SegStatic& st, std::vector<Node>& out)
        : c(c), recs(recs), RL(c->read_len), D(st.D), nd(st.nd), part(st.part), clusters(st.clusters), st(st), out(out) {}

    El el(int32_t idx) const {
        const StreamRec& r = rec(idx);
        return El{r.refid, r.fb_refpos, r.fb_matchref, (int32_t)r.fb_readpos, (bool)(r.flags & SR_REV)};
    }
    static bool el_less(const El& a, const El& b) { return a.refid != b.refid ? a.refid < b.refid : a.refpos < b.refpos; }
    void push_node(int chr, int pos, int len) { out.push_back(Node{chr, pos, len, 0, 0.0}); }
    bool have_back() const { return !out.empty(); }
    int back_end() const { return out.back().pos + out.back().len; }

    void new_cluster() {
        ++kc;
        disright = nextdisright; disChr = nextdisChr;
        if (kc < (int)clusters.size()) {
            const Cluster& k = clusters[kc];
            ds = k.ds; de = k.de; nextdisright = k.right; nextdisChr = k.chr;
            rest_p = st.rest_pos.data() + st.rest_off[kc]; rest_l = st.rest_len.data() + st.rest_off[kc];
            rest_n = st.rest_off[kc + 1] - st.rest_off[kc];
            rest_maxm = st.rest_max[kc];
        } else {  // past the last cluster: the reference reads the zero sentinel (ledger B21); nextdisChr keeps its value
            ds = de = nd; nextdisright = 0;
            rest_p = rest_l = nullptr; rest_n = 0;
        }
    }

    void close_node(int chr, int& curStart, int& curEnd, int lastC, bool& split) {  // :483-493 / :506-515
        split = true;
        if (D[ds].refpos - curStart > NEAR && lastC - D[ds].refpos > NEAR) {
            push_node(chr, curStart, D[ds].refpos - curStart);
            curStart = D[ds].refpos;
        }
        push_node(chr, curStart, lastC - curStart);
        curStart = lastC; curEnd = lastC;
        markStart = lastC; markChr = chr;
    }

    // SQUID_REPLAY_CHECK (tests): the counts of one break candidate once more, with the linear passes of SegmentGraph.cpp:445-474 over
    // MarginPositions, the cluster's blocks, both windows from their offsets on and the ConcordRest content, against what the sorted
    // arrays and the span indices gave (sr, pl, pr, and the spanning coverage in both of its stages)
    void check_candidate(int brk, int chr, const std::vector<int>& M, int sr, int pl, int pr) {
        int sr2 = 0, pl2 = 0, pr2 = 0;
        for (size_t k = 0; k < M.size() && M[k] < brk + T; ++k) if (std::abs(brk - M[k]) < T) ++sr2;
        for (int d = ds; d != de; ++d) {
            const int e = D[d].refpos + D[d].matchref;
            if (e < brk && e > brk - RL && !D[d].rev) ++pl2;
            else if (D[d].refpos > brk && D[d].refpos < brk + RL && D[d].rev) ++pr2;
        }
        bool bad = sr2 != sr || pl2 != pl || pr2 != pr;
        if (sr > 3 || sr + pl > 4 || sr + pr > 4) {
            auto spans = [&](int id, int p, int m) { return id == chr && p + m >= brk + T && p < brk - T; };
            int cov_c = 0, cov_d = 0, cov_p = 0, cov_r = 0;
            for (int i = co; i < (int)cw.size(); ++i) { const El it = el(cw[i]); if (spans(it.refid, it.refpos, it.matchref)) ++cov_c; }
            for (int d = ds; d != de; ++d) if (spans(D[d].refid, D[d].refpos, D[d].matchref)) ++cov_d;
            for (int i = po; i != (int)pw.size(); ++i) { const El it = el(pw[i]); if (spans(it.refid, it.refpos, it.matchref)) ++cov_p; }
            for (int i = 0; i < rest_n; ++i) if (rest_p[i] + rest_l[i] >= brk + T && rest_p[i] < brk - T) ++cov_r;
            int d2 = 0;
            const int dmax = [&]() { int m = 0; for (int d = ds; d != de; ++d) m = std::max(m, D[d].matchref); return m; }();
            auto blk_lb = [&](long long x) { int a = ds, b = de; while (a < b) { int m = (a + b) >> 1; if (D[m].refpos < x) a = m + 1; else b = m; } return a; };
            for (int d = blk_lb((long long)brk + T - dmax), d1 = blk_lb((long long)brk - T); d < d1; ++d) if (spans(D[d].refid, D[d].refpos, D[d].matchref)) ++d2;
            if (cset.ix.count(brk, T) != cov_c || pset.ix.count(brk, T) != cov_p || d2 != cov_d) bad = true;
            if (rest_n && rspan.count(brk, T) != cov_r) bad = true;
        }
        c->replay_checked.fetch_add(1, std::memory_order_relaxed);
        if (bad) c->replay_mismatch.fetch_add(1, std::memory_order_relaxed);
    }
    // one discordant cluster has been passed by record (recChr, recPos): SegmentGraph.cpp:354-611
    void process_cluster(int recChr, int recPos) {
        int curEnd = 0, curStart = 0, disStart = -1, disEnd = -1, disCount = -1;
        bool split = false;
        if (markStart != -1 && D[ds].refid != markChr) { markChr = -1; markStart = -1; }
        while ((int)cw.size() != co && el(cw[co]).refid < D[ds].refid) ++co;
        while ((int)pw.size() != po && el(pw[po]).refid < D[ds].refid) ++po;
        if ((int)cw.size() != co) { El b = el(cw.back()); if (D[ds].refpos > b.refpos + b.matchref + RL) co = (int)cw.size(); }
        if ((int)pw.size() != po) { El b = el(pw.back()); if (D[ds].refpos > b.refpos + b.matchref + RL) po = (int)pw.size(); }
        curStart = D[ds].refpos;
        {
            bool hc = (int)cw.size() != co, hp = (int)pw.size() != po;
            El t{};
            if (hc && hp) { El x = el(cw[co]), y = el(pw[po]); t = el_less(x, y) ? x : y; }
            else if (hc) t = el(cw[co]);
            else if (hp) t = el(pw[po]);
            if ((hc || hp) && (t.refid < D[ds].refid || (t.refid == D[ds].refid && t.refpos < D[ds].refpos))) curStart = t.refpos;
        }
        curStart = std::max(curStart, markStart);
        // ConcordRest: `rest` already holds exactly the heap elements that survive the pops of :387-389 and can span a
        // break of this cluster (same chromosome, refpos >= start - ReadLen), see k_rest_candidates
        for (; ps != part.size() && (part[ps].first < D[ds].refid || (part[ps].first == D[ds].refid && part[ps].second + RL < D[ds].refpos)); ++ps) {}
        for (pe = ps; pe != part.size() && part[pe].first == D[ds].refid && part[pe].second < nextdisright + RL; ++pe) {}

        auto tick = [&]() { return prof ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point(); };
        auto tock = [&](int k, std::chrono::steady_clock::time_point t0, long long n) { if (prof) { tsec[k] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); nsec[k] += n; } };
        std::vector<int>& M = M_buf;  // MarginPositions
        std::vector<int>&fwd_ends = fwd_buf, &rev_starts = rev_buf;
        while (ds != de) {
            const int chr = D[ds].refid;
            if (ds != 0 && D[ds].refid != D[ds - 1].refid && (int)cw.size() == co && (int)pw.size() == po) curStart = D[ds].refpos;
            split = false;
            auto tA = tick();
            M.clear();
            for (dcur = ds; dcur != de; ++dcur) {
                M.push_back(D[dcur].refpos);
                M.push_back(D[dcur].refpos + D[dcur].matchref);
                curEnd = std::max(curEnd, M.back());
                if (dcur + 1 != de && D[dcur + 1].refpos > D[dcur].refpos + D[dcur].matchref) break;
            }
            disStart = std::max(curStart, D[ds].refpos);
            disEnd = curEnd;
            disCount = dcur - ds;
            if (dcur != de)
                for (++dcur; dcur != de && D[dcur].refpos < curEnd + T; ++dcur) { M.push_back(D[dcur].refpos); M.push_back(D[dcur].refpos + D[dcur].matchref); }
            for (size_t q = ps; q != pe && part[q].second < curEnd + T; ++q) M.push_back(part[q].second);
            for (int i = po; i != (int)pw.size(); ++i) {
                El it = el(pw[i]);
                if (it.refid != chr) continue;
                const int front = M.front(), e = it.refpos + it.matchref;
                auto inwin = [&](int x) { return x > front - T && x < curEnd + T; };
                if (it.readpos > 15 && inwin(it.refpos)) {
                    if (it.rev && inwin(e)) M.push_back(e);
                    else if (!it.rev) M.push_back(it.refpos);
                } else {
                    if (it.rev && inwin(it.refpos)) M.push_back(it.refpos);
                    else if (!it.rev && inwin(e)) M.push_back(e);
                }
            }
            std::sort(M.begin(), M.end());
            fwd_ends.clear(); rev_starts.clear();
            int dmaxm = 0;
            for (int d = ds; d != de; ++d) { if (D[d].rev) rev_starts.push_back(D[d].refpos); else fwd_ends.push_back(D[d].refpos + D[d].matchref); dmaxm = std::max(dmaxm, D[d].matchref); }
            std::sort(fwd_ends.begin(), fwd_ends.end());  // (rev_starts is in block order, i.e. already sorted)
            // span counters over the two windows and the ConcordRest content (offsets co/po are fixed while the candidates are tested)
            span_collect(cset, ci, co, chr, M.front(), M.back());
            span_collect(pset, pi, po, chr, M.front(), M.back());
            if (rest_n) rspan.build(rest_p, rest_l, 0, rest_n, rest_maxm, M.front(), M.back(), T);
            tock(0, tA, (long long)M.size());
            auto tB = tick();

            int lastC = -1, lastSup = 0;
            size_t m_lo = 0, m_hi = 0, f_lo = 0, f_hi = 0, r_lo = 0, r_hi = 0;
            for (size_t ib = 0; ib < M.size();) {
                const int brk = M[ib];
                size_t nx = ib;
                while (nx < M.size() && M[nx] == brk) ++nx;
                bool skip = have_back() && out.back().chr == chr && brk - back_end() < NEAR;
                if (prof) nsec[3]++;
                if (!skip) {
                    auto tq = tick();
                    // the reference counts these with linear passes over M and over the cluster's blocks for every candidate;
                    // M, fwd_ends and rev_starts are sorted, so the same counts are differences of binary searches
                    // (the candidates come in ascending order: every bound only moves forward)
                    auto adv = [](const std::vector<int>& v, size_t& it, int x) { while (it < v.size() && v[it] < x) ++it; return (int)it; };
                    const int sr = adv(M, m_hi, brk + T) - adv(M, m_lo, brk - T + 1);                        // |brk - M[k]| < T
                    const int pl = adv(fwd_ends, f_hi, brk) - adv(fwd_ends, f_lo, brk - RL + 1);              // forward block ending in (brk-RL, brk)
                    const int pr = adv(rev_starts, r_hi, brk + RL) - adv(rev_starts, r_lo, brk + 1);          // reverse block starting in (brk, brk+RL)
                    tock(3, tq, 0);
                    if (recount) check_candidate(brk, chr, M, sr, pl, pr);
                    if (sr > 3 || sr + pl > 4 || sr + pr > 4) {
                        auto tw = tick();
                        auto spans = [&](int id, int p, int m) { return id == chr && p + m >= brk + T && p < brk - T; };
                        int cov = 0;
                        // spans() needs p < brk - T and p + m >= brk + T
                        cov += cset.ix.count(brk, T);
                        {   // the cluster's own blocks (sorted by refpos, none longer than dmaxm)
                            auto blk_lb = [&](long long x) { int a = ds, b = de; while (a < b) { int m = (a + b) >> 1; if (D[m].refpos < x) a = m + 1; else b = m; } return a; };
                            for (int d = blk_lb((long long)brk + T - dmaxm), d1 = blk_lb((long long)brk - T); d < d1; ++d) if (spans(D[d].refid, D[d].refpos, D[d].matchref)) ++cov;
                        }
                        cov += pset.ix.count(brk, T);
                        tock(4, tw, 1);
                        auto tr = tick();
                        if (sr > std::max(cov - sr, 0) + 2)
                            if (rest_n) cov += rspan.count(brk, T);  // (same chromosome by construction, see k_rest_candidates)
                        tock(5, tr, 1);
                        if (sr > std::max(cov - sr, 0) + 2) {
                            int sup = std::max(sr + pl, sr + pr);
                            if (lastC == -1 && brk - curStart < NEAR) { markStart = curStart; markChr = chr; }
                            else if ((lastC == -1 || brk - lastC < NEAR) && sup > lastSup) { lastC = brk; lastSup = sup; }
                            else if (brk - lastC >= NEAR) { close_node(chr, curStart, curEnd, lastC, split); lastC = brk; }
                        }
                    }
                }
                ib = nx;
            }
            tock(1, tB, (long long)rest_n);
            auto tC = tick();
            if (lastC != -1 && (!split || back_end() != lastC)) close_node(chr, curStart, curEnd, lastC, split);
            if (disStart != -1 && !split && disCount > std::min(5.0, 4.0 * (disEnd - disStart) / RL)) {  // :518-527 (FP64 as in the reference)
                if (have_back() && out.back().chr == D[de - 1].refid && disEnd - back_end() < NEAR) out.back().len += disEnd - back_end();
                else push_node(D[de - 1].refid, disStart, disEnd - disStart);
                curStart = disEnd; curEnd = disEnd;
                markStart = disEnd; markChr = chr;
            }
            while ((int)cw.size() != co && el(cw[co]).refid < chr) ++co;
            while ((int)pw.size() != po && el(pw[po]).refid < chr) ++po;
            for (dcur = ds; dcur != de && D[dcur].refpos + D[dcur].matchref <= curEnd; ++dcur) {}
            int zero = curStart;  // concord0pos
            auto step1 = [&](std::vector<int32_t>& W, int& off) {
                if ((int)W.size() == off) return false;
                El b = el(W[off]);
                bool f = true;
                if (b.refid > chr) f = false;
                if (dcur != nd && b.refid == D[dcur].refid && b.refpos + b.matchref + RL >= D[dcur].refpos) f = false;
                if (have_back() && (b.refid > out.back().chr || (b.refid == out.back().chr && b.refpos >= back_end()))) f = false;
                if (f) { zero = std::max(zero, b.refpos + b.matchref); ++off; }
                return f;
            };
            do {
                bool f1 = step1(cw, co), f2 = step1(pw, po);
                if (!f1 && !f2) break;
            } while ((int)cw.size() != co || (int)pw.size() != po);
            auto step2 = [&](std::vector<int32_t>& W, int& off) {
                if ((int)W.size() == off) return false;
                El b = el(W[off]);
                bool f = dcur == nd || b.refid < D[dcur].refid || (b.refid == D[dcur].refid && b.refpos + b.matchref + RL < D[dcur].refpos);
                if (f) { zero = std::max(zero, b.refpos + b.matchref); ++off; }
                return f;
            };
            do {
                bool cfree = (int)cw.size() == co, pfree = (int)pw.size() == po;
                if (!cfree) { El b = el(cw[co]); cfree = b.refid != markChr || b.refpos > zero + RL; }
                if (!pfree) { El b = el(pw[po]); pfree = b.refid != markChr || b.refpos > zero; }
                if (markStart != -1 && (recChr > markChr || recPos > zero + RL) && cfree && pfree) {
                    if (zero > markStart && zero < markStart + NEAR && have_back() && out.back().chr == markChr) out.back().len += zero - back_end();
                    else if (zero > markStart) push_node(markChr, markStart, zero - markStart);
                    curStart = zero;
                    markChr = -1; markStart = -1;
                    break;
                }
                bool f1 = step2(cw, co), f2 = step2(pw, po);
                if (!f1 && !f2) break;
            } while ((int)cw.size() != co || (int)pw.size() != po);
            tock(2, tC, (long long)(cw.size() - co + pw.size() - po));
            ds = dcur;
        }
        if (de - ds <= 0) new_cluster();
    }
};

}  // namespace

struct SegPlan {
    SegStatic st;
    SegSupport sup;
    std::vector<int> active;
    std::vector<int> first_cluster;   // per active stretch: the first cluster whose trigger lies in it
    std::vector<int64_t> shift;       // per active stretch: kept index that compact[0] would have
    const StreamRec* compact = nullptr;  // page-locked copy of the summaries inside the replayed stretches
    int64_t K = 0;        // kept records of the local stream
    int64_t K_eff = 0;    // + the appended first kept record of the next shard
    int k0 = 0;           // clusters that an earlier shard's closing record has already passed
    std::vector<int32_t> cl_chr, cl_start, cl_right;  // the cluster table as uploaded
    long long other_max_local = INT64_MIN;  // running (otherChr, otherrightmost) pair over the whole local stream
    bool skip_first = false;  // the local stream does not start the global one: its first record only opens a stretch
};

// static part: discordant blocks, clip positions, cluster table; stream scans that need nothing from other shards
// (host only, reads the chimeric fragments: may run next to the record kernels) returns the elapsed milliseconds
double segment_clusters(const sq_ctx* c, std::shared_ptr<SegPlan>& plan, std::vector<Blk>& disc_sorted) {
    plan = std::make_shared<SegPlan>();
    SegStatic& S = plan->st;
    const auto t_begin = std::chrono::steady_clock::now();
    static const bool laps = std::getenv("SQUID_PREP_DEBUG") != nullptr;
    auto lap = [&](const char* what) { if (laps) std::fprintf(stderr, "[clusters] %-22s at %.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count()); };
    // ---- discordant blocks and clip positions of the chimeric fragments (SegmentGraph.cpp:203-264)
    S.part.assign(c->ref_len.size(), std::make_pair(0, 0));  // ledger B10
    std::vector<Blk>& D = S.D;
    // (the fragments are independent of each other here, except for one look at the last discordant block collected so far -- ledger
    // B11 -- : pieces of the list are worked on by the context's host threads, each with its own output, and strung together in order;
    // a piece that needs that block before it has collected one of its own asks for a second, serial visit)
    struct Out { std::vector<Blk> D; std::vector<std::pair<int, int>> part; std::vector<std::pair<size_t, Blk>> ask; /* (position in part, block to compare with the last discordant block) */ };
    const size_t nf = c->frags.size();
    const int pieces = (c->pool && nf > 20000) ? 4 * (c->pool->size() + 1) : 1;
    std::vector<Out> outs((size_t)pieces);
    auto work = [&](int pi) {
        Out& O = outs[(size_t)pi];
        std::vector<Blk>& D = O.D;
        for (size_t fi = nf * (size_t)pi / (size_t)pieces; fi < nf * ((size_t)pi + 1) / (size_t)pieces; ++fi) {
        const Frag& f = c->frags[fi];
        if (frag_end_discordant(f, true) || frag_end_discordant(f, false) || frag_single_anchored(f) || frag_pair_discordant(f, true)) {
            D.insert(D.end(), f.a.begin(), f.a.end());
            D.insert(D.end(), f.b.begin(), f.b.end());
            continue;
        }
        bool ains = false, bins = false;
        auto far = [&](const BlkList& R, bool& ins) {
            int prev = -1;
            for (int i = 0; i + 1 < (int)R.size(); ++i)
                if (std::abs(R[i].refpos - R[i + 1].refpos) > 750000) {
                    if (prev != i) D.push_back(R[i]);
                    D.push_back(R[i + 1]);
                    prev = i + 1;
                    if (i + 1 == (int)R.size() - 1) ins = true;
                }
        };
        far(f.a, ains);
        far(f.b, bins);
        if (!f.a.empty() && !f.b.empty() && std::abs(f.a.back().refpos - f.b.back().refpos) > 750000) {
            if (!ains) { D.push_back(f.a.back()); ains = true; }
            if (!bins) { D.push_back(f.b.back()); bins = true; }
        }
        if (!ains && !bins) {
            auto clip = [&](const Blk& b, bool leftEnd) { return std::make_pair(b.refid, (leftEnd == b.rev) ? b.refpos + b.matchref : b.refpos); };
            if (!f.a.empty() && f.a.front().readpos > 15 && !f.alow) O.part.push_back(clip(f.a.front(), true));
            if (!f.a.empty() && f.atot - f.a.back().readpos - f.a.back().matchread > 15 && !f.alow) O.part.push_back(clip(f.a.back(), false));
            if (!f.b.empty() && f.b.front().readpos > 15 && !f.blow) O.part.push_back(clip(f.b.front(), true));
            if (!f.b.empty() && f.btot - f.b.back().readpos - f.b.back().matchread > 15 && !f.blow) {
                // ledger B11: compared with bamdiscordant.back() even when that vector is empty (zero block there)
                const Blk& s = f.b.back();
                if (D.empty()) { O.ask.push_back(std::make_pair(O.part.size(), s)); O.part.push_back(clip(s, false)); }  // (settled below, once the pieces in front are known)
                else if (!blk_same(D.back(), s)) O.part.push_back(clip(s, false));
            }
        }
        }
    };
    if (pieces > 1) c->pool->parallel_for(pieces, 1 << 20, work); else work(0);
    lap("fragments walked");
    auto par = [&](size_t n, const std::function<void(size_t, size_t)>& f) {  // [lo, hi) pieces on the context's host threads
        const int np = (c->pool && n > 100000) ? 4 * (c->pool->size() + 1) : 1;
        if (np <= 1) { f(0, n); return; }
        c->pool->parallel_for(np, 1 << 20, [&](int k) { f(n * (size_t)k / (size_t)np, n * ((size_t)k + 1) / (size_t)np); });
    };
    // (the pieces are not strung together: the sort below works on (key, index) elements, and the sorted list is gathered straight from
    // the pieces -- d_at[k] = global index of piece k's first block)
    std::vector<size_t> d_at(outs.size() + 1, 0);
    for (size_t k = 0; k < outs.size(); ++k) d_at[k + 1] = d_at[k] + outs[k].D.size();
    const size_t nD = d_at.back();
    {
        // the entries that were waiting for the last discordant block of the pieces in front: drop those that are the `Same` block
        const Blk z{0, 0, 0, 0, 0, false, false};
        const Blk* last = &z;  // the last discordant block of the pieces in front
        for (size_t k = 0; k < outs.size(); ++k) {
            Out& O = outs[k];
            std::vector<char> drop(O.part.size(), 0);
            for (const auto& q : O.ask) if (blk_same(*last, q.second)) drop[q.first] = 1;
            for (size_t i = 0; i < O.part.size(); ++i) if (!drop[i]) S.part.push_back(O.part[i]);
            if (!O.D.empty()) last = &O.D.back();
        }
    }
    lap("pieces strung");
    const int sort_threads = c->pool ? std::min(c->pool->size() + 1, 32) : 1;
    std_sort_parallel(S.part.begin(), S.part.end(), std::less<std::pair<int, int>>(), sort_threads, true);
    lap("clip positions sorted");
    std::future<void> room2;  // (the room for the copy the edge stage reads: made next to the sort as well)
    {   // ledger B8: operator< looks at (RefID, RefPos) only and the sort is not stable.  Sorting 12-byte (key, index) elements with the
        // same comparison takes libstdc++'s introsort through the same decisions, hence to the same permutation, at a fraction of
        // the memory traffic of sorting the blocks themselves
        struct PK { int32_t refid, refpos, idx; };
        struct RawPK { PK* p; explicit RawPK(size_t n) : p((PK*)std::malloc(std::max<size_t>(1, n) * sizeof(PK))) { if (!p) throw std::bad_alloc(); } ~RawPK() { std::free(p); } } pk(nD);  // (every element is written below)
        // the room for the sorted list is made (230 MB of zeroes on the dense config, one thread's work) while the keys are being sorted
        std::future<void> room;
        std::vector<Blk> sorted;
        if (c->pool && nD > 100000) room2 = std::async(std::launch::async, [&disc_sorted, nD]() { disc_sorted.clear(); disc_sorted.resize(nD); });
        if (c->pool && nD > 100000) room = std::async(std::launch::async, [&sorted, nD]() { sorted.reserve(nD + 1); sorted.resize(nD); }); else { sorted.reserve(nD + 1); sorted.resize(nD); }  // (+ 1: the sentinel pushed below must not move the list)  // (a thread of its own: this function may itself be a task of the pool, and a task that waits for another task can starve)
        if (pieces > 1) c->pool->parallel_for(pieces, 1 << 20, [&](int k) { const std::vector<Blk>& P = outs[(size_t)k].D; const size_t at = d_at[(size_t)k]; for (size_t i = 0; i < P.size(); ++i) pk.p[at + i] = PK{P[i].refid, P[i].refpos, (int32_t)(at + i)}; });
        else for (size_t i = 0; i < nD; ++i) pk.p[i] = PK{outs[0].D[i].refid, outs[0].D[i].refpos, (int32_t)i};
        lap("  block keys");
        // (std_sort_parallel, sq_parsort.h: the same introsort with its independent sub-ranges on several threads)
        std_sort_parallel(pk.p, pk.p + nD, [](const PK& x, const PK& y) { return x.refid != y.refid ? x.refid < y.refid : x.refpos < y.refpos; }, sort_threads, true);  // (a strict weak order: the final insertion pass is split too)
        lap("  block keys sorted");
        if (room.valid()) room.get();
        lap("  room for the sorted blocks");
        par(nD, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i) {
                const size_t g = (size_t)pk.p[i].idx;
                const size_t k = (size_t)(std::upper_bound(d_at.begin(), d_at.end(), g) - d_at.begin()) - 1;
                sorted[i] = outs[k].D[g - d_at[k]];
            }
        });
        D.swap(sorted);
        lap("  blocks gathered");
    }
    lap("blocks sorted");
    S.nd = (int)D.size();
    // the copy the edge stage reads; the pieces are freed behind the caller's back (64 unmaps of a few megabytes each)
    if (room2.valid()) {
        room2.get();
        par(nD, [&](size_t lo, size_t hi) { std::copy(D.begin() + (std::ptrdiff_t)lo, D.begin() + (std::ptrdiff_t)hi, disc_sorted.begin() + (std::ptrdiff_t)lo); });
        auto* junk = new std::vector<Out>(std::move(outs));
        std::thread([junk]() { delete junk; }).detach();
    } else disc_sorted = D;
    lap("copy for the edge stage");
    if (laps) std::fprintf(stderr, "[clusters] %zu discordant blocks, %zu clip positions\n", D.size(), S.part.size());

    // ---- static cluster table; everything stream-sized comes from the GPU
    S.build_clusters(c->read_len, c->pool.get());
    lap("clusters built");
    D.push_back(Blk{0, 0, 0, 0, 0, false, false});  // ledger B21 sentinel
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
}
// pass 1 over the records (k_pass1; the cluster table is uploaded here).  A sharded run calls this before it knows the running pair
// of the earlier shards: everything it publishes (stream size, first kept record, running pair, trigger of the last cluster) is
// independent of that seed; segment_prepare repeats the pass with the seed when there is one.
int segment_scan(sq_ctx* c, SegPlan& P, bool fetch, int64_t& trigger_last, long long& other_max, int32_t first_kept[2]) {
    SegStatic& S = P.st;
    const int ncl = (int)S.clusters.size();
    P.cl_chr.resize(ncl); P.cl_start.resize(ncl); P.cl_right.resize(ncl);
    for (int k = 0; k < ncl; ++k) { P.cl_chr[k] = S.clusters[k].chr; P.cl_start[k] = S.clusters[k].start; P.cl_right[k] = S.clusters[k].right; }
    Pass1Result res;
    int rc = dev_pass1(c, P.cl_chr, P.cl_start, P.cl_right, res);
    if (rc) return rc;
    P.K = res.kept;
    P.other_max_local = res.other_max;
    other_max = res.other_max;
    first_kept[0] = res.first_kept[0]; first_kept[1] = res.first_kept[1];
    trigger_last = ncl ? std::min<int64_t>(res.trigger_last, res.kept) : -1;  // (-1: no discordant cluster at all)
    (void)fetch;
    return SQ_OK;
}

// zero-coverage records, triggers, stretches to replay, host copies of the summaries inside them
int segment_prepare(sq_ctx* c, SegPlan& P, int64_t& n_break) {
    SegStatic& S = P.st;
    const Shard& sh = c->shard;
    const int ncl = (int)S.clusters.size(), nd = S.nd;
    SegSupport& sup = P.sup;
    auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char* name) { auto t = std::chrono::steady_clock::now(); c->timer.add(name, std::chrono::duration<double, std::milli>(t - t_begin).count()); t_begin = t; };
    const int64_t K = P.K;
    // (the zero-coverage test looks at the running pair, which earlier shards feed: their pair is the seed)
    int rc = dev_segment_support(c, ncl, sh.on ? sh.other_seed : INT64_MIN, sup);
    if (rc) return rc;
    lap("host_prep_support");
    const bool term = sh.on && sh.has_terminal && K > 0;
    P.K_eff = K + (term ? 1 : 0);
    const int64_t KE = P.K_eff;
    StreamRec term_rec{};
    if (term) {
        // the first kept record of the next shard closes this shard's last stretch: its zero-coverage test (SegmentGraph.cpp:616-620)
        // with the running pair of the whole local stream in front of it
        term_rec.refid = sh.term_refid; term_rec.pos = sh.term_pos;
        int Kt = 0;
        while (Kt < ncl && (S.clusters[Kt].chr < sh.term_refid || (S.clusters[Kt].chr == sh.term_refid && S.clusters[Kt].right < sh.term_pos))) ++Kt;
        const int disChr = Kt > 0 ? S.clusters[Kt - 1].chr : 0, disRight = Kt > 0 ? S.clusters[Kt - 1].right : 0;
        const int dnChr = Kt < ncl ? S.clusters[Kt].chr : 0, dnPos = Kt < ncl ? S.clusters[Kt].start : 0;  // zero sentinel after the last cluster (ledger B21)
        long long ob = std::max<long long>(P.other_max_local, sh.other_seed);
        if (ob < 0) ob = 0;
        const int oChr = (int)(ob >> 32), oRight = (int)(ob & 0xffffffffll);
        const bool disLead = disChr > oChr || (disChr == oChr && disRight > oRight);
        const int curRight = disLead ? disRight : oRight, curChr = std::max(disChr, oChr);
        const int RLt = c->read_len;
        if ((sh.term_refid != curChr || sh.term_pos > curRight + RLt) && (curChr < dnChr || (curChr == dnChr && curRight + RLt < dnPos))) {
            sup.zidx.push_back((int32_t)K); sup.z_ochr.push_back(oChr); sup.z_oright.push_back(oRight);
        }
    }
    if (sh.on) {
        // triggers were computed on the local stream: a cluster nobody here has passed is passed by the appended record
        // of the next shard if that lies beyond it
        for (int k = 0; k < ncl; ++k)
            if (sup.trigger[k] >= K) {
                const Cluster& cl = S.clusters[k];
                bool beyond = term && (cl.chr < sh.term_refid || (cl.chr == sh.term_refid && cl.right < sh.term_pos));
                sup.trigger[k] = (int32_t)(beyond ? K : KE);
            }
    }
    {   // group the candidates by cluster (counting sort), then order each group by refpos
        const size_t cnt = sup.rest_cluster.size();
        S.rest_off.assign((size_t)ncl + 1, 0); S.rest_max.assign((size_t)ncl, 0);
        for (size_t i = 0; i < cnt; ++i) S.rest_off[sup.rest_cluster[i] + 1]++;
        for (int k = 0; k < ncl; ++k) S.rest_off[k + 1] += S.rest_off[k];
        std::vector<std::pair<int32_t, int32_t>> tmp(cnt);
        std::vector<int32_t> fill(S.rest_off.begin(), S.rest_off.end() - 1);
        for (size_t i = 0; i < cnt; ++i) tmp[(size_t)fill[sup.rest_cluster[i]]++] = std::make_pair(sup.rest_pos[i], sup.rest_len[i]);
        S.rest_pos.resize(cnt); S.rest_len.resize(cnt);
        auto one = [&](int k) {
            auto b = tmp.begin() + S.rest_off[k], e = tmp.begin() + S.rest_off[k + 1];
            if (e - b > 1) std::sort(b, e);
            int mx = 0;
            for (auto it = b; it != e; ++it) { const size_t i = (size_t)(it - tmp.begin()); S.rest_pos[i] = it->first; S.rest_len[i] = it->second; mx = std::max(mx, it->second); }
            S.rest_max[k] = mx;
        };
        // (the clusters' groups are independent: sorted side by side on a few host threads once there is enough to sort)
        const int pieces = (c->pool && cnt > 20000 && ncl > 64) ? 32 : 1;
        if (pieces == 1) for (int k = 0; k < ncl; ++k) one(k);
        else c->pool->parallel_for(pieces, 15, [&](int p) { for (int k = (int)((int64_t)ncl * p / pieces); k < (int)((int64_t)ncl * (p + 1) / pieces); ++k) one(k); });
    }
    lap("host_prep_rest");
    // ReadsMain/ReadsOther stop growing at the first record after the last cluster's trigger (SegmentGraph.cpp:338-339, B12)
    if (!sh.on) {
        if (ncl == 0) n_break = std::min<int64_t>(K, 1);
        else n_break = std::min<int64_t>(K, (int64_t)sup.trigger[ncl - 1] + 2);
    } else n_break = std::max<int64_t>(0, std::min<int64_t>(K, sh.n_break_global - sh.kept_before));
    // (the plan object is kept across passes, sq_capi.cpp: every per-pass field is put back before the early way out)
    P.active.clear(); P.first_cluster.clear(); P.shift.clear(); P.compact = nullptr; P.skip_first = false; P.k0 = 0;
    if (nd == 0 || K == 0) return SQ_OK;

    // ---- stretches between zero-coverage records; a stretch j covers: the push step of its first record lo (a
    // zero-coverage record, or -1 for the head of the stream), full steps of lo+1 .. hi-1, and the cluster events
    // plus the zero-coverage step of its last record hi (hi == K_eff: the stream ends inside the stretch)
    const std::vector<int32_t>& Z = sup.zidx;
    const int nz = (int)Z.size();
    auto stretch_of = [&](int64_t t) {  // stretch whose (lo, hi] contains t
        return (int)(std::lower_bound(Z.begin(), Z.end(), (int32_t)t) - Z.begin());
    };
    // a shard that does not start the global stream: its first record was the closing record of an earlier shard's
    // last stretch, which has processed every cluster that record passes
    P.skip_first = sh.on && sh.prior_kept;
    P.k0 = 0;
    if (P.skip_first) {
        const StreamRec* none = nullptr; (void)none;
        while (P.k0 < ncl && sup.trigger[P.k0] == 0) ++P.k0;
        if (P.k0 < ncl && (nz == 0 || Z[0] != 0)) return fail(c, SQ_E_ARG, "internal: first record of a shard is not a zero-coverage record");
    }
    std::vector<int>& active = P.active;
    for (int k = P.k0; k < ncl; ++k) {
        if (sup.trigger[k] >= KE) break;  // never passed by a record: never segmented (the reference leaves its loop first)
        int j = stretch_of(sup.trigger[k]);
        if (active.empty() || active.back() != j) { active.push_back(j); P.first_cluster.push_back(k); }
    }
    std::vector<std::pair<int64_t, int64_t>> ranges;
    std::vector<int> range_of;
    for (int j : active) {
        int64_t lo = j == 0 ? 0 : Z[j - 1], hi = j < nz ? (int64_t)Z[j] + 1 : KE;
        if (!ranges.empty() && ranges.back().second >= lo) ranges.back().second = std::max(ranges.back().second, hi);
        else ranges.push_back(std::make_pair(lo, hi));
        range_of.push_back((int)ranges.size() - 1);
    }
    lap("host_prep_active");
    std::vector<int64_t> range_off;
    rc = dev_fetch_stream(c, ranges, P.compact, range_off, term ? &term_rec : nullptr);
    if (rc) return rc;
    P.shift.resize(active.size());
    for (size_t a = 0; a < active.size(); ++a) P.shift[a] = ranges[range_of[a]].first - range_off[range_of[a]];
    return SQ_OK;
}

// order-dependent part: replay the reference's control automaton over the stretches that contain cluster triggers
// Replays the active stretches [a_begin, a_end).  `virtual_back`: a node has been emitted before this range and it lies
// on an earlier chromosome than anything the range looks at (then only its existence matters); `sens` collects the
// pending node starts that were compared with that node's end without a chromosome test (SegmentGraph.cpp:623).
static int replay_range(sq_ctx* c, SegPlan& plan, size_t a_begin, size_t a_end, std::vector<Node>& seeds, bool virtual_back, std::vector<int32_t>* sens, std::string& err,
                        const Node* seed = nullptr) {
    seeds.clear();
    std::vector<Node> sink;
    if (seed) sink.push_back(*seed);  // the real node in front (the result keeps it, possibly extended)
    else if (virtual_back) sink.push_back(Node{-1, 0, 0, 0, 0.0});
    Seg S(c, plan.compact, plan.st, sink);
    S.prof = std::getenv("SQUID_REPLAY_PROF") != nullptr;
    SegSupport& sup = plan.sup;
    const std::vector<Blk>& D = S.D;
    const int nd = S.nd, RL = c->read_len;
    const int64_t K = plan.K_eff;
    const std::vector<int32_t>& Z = sup.zidx;
    const int nz = (int)Z.size();
    const std::vector<int>& active = plan.active;
    if (nd == 0 || plan.K == 0 || a_begin >= a_end) return SQ_OK;

    static const bool prof = std::getenv("SQUID_REPLAY_PROF") != nullptr;
    long long n_pushed = 0, n_clusters = 0;
    double t_cluster = 0;
    auto push_step = [&](int64_t i) {  // SegmentGraph.cpp:649-700 (ConcordRest pushes are covered by rest_by_cluster)
        const StreamRec& r = S.rec(i);
        if (!(r.flags & SR_CONC)) return;
        const int e = r.fb_refpos + r.fb_matchref;
        if (r.flags & SR_MATE) {  // the reference keys these updates on IsFirstMate()/IsSecondMate()
            if (S.otherChr == r.refid) S.otherright = std::max(S.otherright, e);
            else { S.otherright = e; S.otherChr = r.refid; }
        }
        if (r.flags & SR_PART) S.win_push(S.pw, S.pi, (int32_t)i); else S.win_push(S.cw, S.ci, (int32_t)i);
    };
    // window pruning as of record (refid): SegmentGraph.cpp:637-646.  Everything it tests except the record's chromosome
    // only changes when a cluster is processed, so between two cluster triggers it is enough to apply it once, right
    // before the next trigger, with the chromosome of the record in front of that trigger.
    auto prune_all = [&](int refid) {
        const Blk& dn = D[S.ds];
        auto prune = [&](std::vector<int32_t>& W, int& off) {
            while ((int)W.size() > off && S.el(W[off]).refid != refid) ++off;
            while ((int)W.size() > off) {
                El b = S.el(W[off]);
                if (b.refid < dn.refid || (S.have_back() && b.refid == S.out.back().chr && b.refpos < S.back_end())) ++off; else break;
            }
        };
        prune(S.cw, S.co);
        prune(S.pw, S.po);
    };
    // events + zero-coverage test + window pruning of record i; returns false when the reference has left its loop
    auto head_step = [&](int64_t i, bool& zerocov) -> bool {
        const StreamRec& r = S.rec(i);
        if (S.ds == nd) return false;  // :338-339
        if (prof) {
            auto t0 = std::chrono::steady_clock::now();
            while (S.ds != nd && (D[S.ds].refid < r.refid || (D[S.ds].refid == r.refid && S.nextdisright < r.pos))) { S.process_cluster(r.refid, r.pos); ++n_clusters; }
            t_cluster += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
        while (S.ds != nd && (D[S.ds].refid < r.refid || (D[S.ds].refid == r.refid && S.nextdisright < r.pos))) S.process_cluster(r.refid, r.pos);
        const bool disLead = S.disChr > S.otherChr || (S.disChr == S.otherChr && S.disright > S.otherright);
        const int curRight = disLead ? S.disright : S.otherright, curChr = std::max(S.disChr, S.otherChr);
        const Blk& dn = D[S.ds];  // the zero sentinel once every cluster is consumed
        zerocov = (r.refid != curChr || r.pos > curRight + RL) && (curChr < dn.refid || (curChr == dn.refid && curRight + RL < dn.refpos));
        if (zerocov && S.markStart != -1) {  // :621-630
            // the first test compares with the end of the last node WITHOUT looking at its chromosome: if that node came from
            // an earlier shard the caller must check the recorded value against its real end
            if (sens && virtual_back && sink.size() == 1 && curChr == S.markChr && curRight > S.markStart && curRight - S.markStart < Seg::NEAR) sens->push_back(S.markStart);
            if (curChr == S.markChr && curRight > S.markStart && curRight - S.markStart < Seg::NEAR && S.have_back() && S.markStart == S.back_end()) S.out.back().len += curRight - S.markStart;
            else if (curChr == S.markChr && curRight > S.markStart && curRight - S.markStart >= Seg::NEAR) S.push_node(S.markChr, S.markStart, curRight - S.markStart);
            S.markStart = -1; S.markChr = -1;
        }
        if (zerocov) { S.co = (int)S.cw.size(); S.po = (int)S.pw.size(); }  // :633-636 (the extra condition there is implied by zerocov)
        else prune_all(r.refid);
        return true;
    };
    const int ncl = (int)S.clusters.size();
    const auto t_begin = std::chrono::steady_clock::now();
    S.new_cluster();  // the reference does this at its first kept record (:341)
    // clusters consumed before this range (by earlier stretches, or by the closing record of an earlier shard)
    for (int k = 0; k < plan.first_cluster[a_begin]; ++k) S.new_cluster();
    for (size_t a = a_begin; a < a_end; ++a) {
        const int j = active[a];
        S.shift = plan.shift[a];
        const int64_t lo = j == 0 ? -1 : Z[j - 1], hi = j < nz ? Z[j] : K;
        // a zero-coverage record empties the windows and clears the pending node end; the running
        // (otherChr, otherrightmost) in front of it comes from the GPU scan
        S.win_clear(S.cw, S.ci, S.co); S.win_clear(S.pw, S.pi, S.po);
        if (lo >= 0) {
            if (S.markStart != -1) { err = "internal: pending node end at a zero-coverage record"; return SQ_E_ARG; }
            S.otherChr = sup.z_ochr[j - 1]; S.otherright = sup.z_oright[j - 1];
            push_step(lo);
        }
        bool alive = true, z = false;
        int64_t i = lo + 1;
        while (i < hi && alive) {
            if (S.ds == nd) { alive = false; break; }
            // records in front of the current cluster's trigger only push (no event, no zero coverage by construction);
            // the head step of the last of them is the pruning that the trigger record finds
            const int64_t t = S.kc < ncl ? std::min<int64_t>(sup.trigger[S.kc], hi) : hi;
            if (i < t) {
                n_pushed += t - i;
                for (; i < t - 1; ++i) push_step(i);
                prune_all(S.rec(i).refid);
                push_step(i);
                ++i;
            }
            if (i >= hi) break;
            alive = head_step(i, z);
            if (alive) {
                if (z) { err = "internal: zero-coverage record inside a replayed stretch"; return SQ_E_ARG; }
                push_step(i);
                ++i;
            }
        }
        if (alive && hi < K) alive = head_step(hi, z);  // its push step opens the next stretch
        if (!alive) break;
    }
    if (prof) std::fprintf(stderr, "[replay] M-build %.3f ms (sumM %lld)  brk-loop %.3f ms (sum rest %lld)  tail %.3f ms (win left %lld)\n", S.tsec[0], S.nsec[0], S.tsec[1], S.nsec[1], S.tsec[2], S.nsec[2]);
    if (prof) std::fprintf(stderr, "[replay] brks %lld  counts %.3f ms  win+blk spans %.3f ms (%lld)  rest %.3f ms (%lld)\n", S.nsec[3], S.tsec[3], S.tsec[4], S.nsec[4], S.tsec[5], S.nsec[5]);
    if (prof) std::fprintf(stderr, "[replay] range %zu..%zu stretches=%zu pushed=%lld clusters=%lld t_cluster=%.3f ms total=%.3f ms\n", a_begin, a_end, active.size(), n_pushed, n_clusters, t_cluster,
                           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    if (virtual_back) sink.erase(sink.begin());
    seeds = sink;
    return SQ_OK;
}

// order-dependent part: replay the reference's control automaton over the stretches that contain cluster triggers.
// Stretches only depend on each other through the last node emitted so far, and when all records of a group of
// stretches lie on later chromosomes than everything before, only the existence of that node matters (replay_range):
// such groups are replayed concurrently under that assumption, checked afterwards, and redone one after the other if
// the check fails.
int segment_replay(sq_ctx* c, SegPlan& plan, std::vector<Node>& seeds, bool virtual_back, std::vector<int32_t>* sens, const Node* seed) {
    seeds.clear();
    const std::vector<int>& active = plan.active;
    const size_t na = active.size();
    std::string err;
    auto sequential = [&]() {
        int rc = replay_range(c, plan, 0, na, seeds, virtual_back, sens, err, seed);
        return rc ? fail(c, rc, err) : SQ_OK;
    };
    if (na == 0) { if (seed) seeds.assign(1, *seed); return SQ_OK; }
    if (seed) return sequential();  // (rare repair path of a sharded run)
    static const bool serial_only = std::getenv("SQUID_REPLAY_SERIAL") != nullptr;
    // group boundaries: the stretch opens with a zero-coverage record on a later chromosome than the last record of the
    // stretch before (which is then that record itself or an earlier one)
    const std::vector<int32_t>& Z = plan.sup.zidx;
    const int nz = (int)Z.size();
    auto rec_at = [&](size_t a, int64_t idx) -> const StreamRec& { return plan.compact[idx - plan.shift[a]]; };
    std::vector<size_t> starts(1, 0);
    std::vector<int> first_chr(1, -1);
    for (size_t a = 1; a < na && !serial_only; ++a) {
        const int j = active[a], jp = active[a - 1];
        const int64_t lo = Z[j - 1];                                   // j > 0 here
        // the last record the stretch before really works on: the one in front of its closing zero-coverage record.  (At that record
        // the stretch only flushes its pending node end, SegmentGraph.cpp:621-636, which lies where its own records lie; the first record
        // of a chromosome is such a closing record, so comparing with IT would never see a chromosome change.  Whatever a group
        // assumes about the node in front of it is checked below.)
        const int64_t hip = jp < nz ? std::max<int64_t>((int64_t)Z[jp] - 1, 0) : plan.K_eff - 1;
        if (rec_at(a, lo).refid > rec_at(a - 1, hip).refid) { starts.push_back(a); first_chr.push_back(rec_at(a, lo).refid); }
    }
    const size_t ng = starts.size();
    if (ng < 3) return sequential();
    starts.push_back(na);
    struct Out { std::vector<Node> seeds; std::vector<int32_t> sens; int rc = 0; std::string err; };
    std::vector<Out> res(ng);
    c->pool->parallel_for((int)ng, 1 << 20, [&](int gi) {
        const size_t g = (size_t)gi;
        res[g].rc = replay_range(c, plan, starts[g], starts[g + 1], res[g].seeds, g == 0 ? virtual_back : true, &res[g].sens, res[g].err);
    });
    // check the assumption group by group: a node must exist before the group, on an earlier chromosome than the group's
    // first record, and none of the recorded comparisons may hit its end
    bool ok = true, local_have = false;
    int last_chr = -1, last_end = 0;
    std::vector<int32_t> up;  // comparisons against a node of an earlier shard: the caller checks them
    for (size_t g = 0; g < ng && ok; ++g) {
        if (res[g].rc) { ok = false; break; }
        if (g == 0 || !local_have) {
            // nothing emitted here yet: the node in front is the caller's (an earlier shard's), if any
            if (g > 0 && !virtual_back) ok = false;  // the group assumed a node where there is none
            up.insert(up.end(), res[g].sens.begin(), res[g].sens.end());
        } else {
            if (last_chr >= first_chr[g]) ok = false;
            for (int32_t v : res[g].sens) if (v == last_end) ok = false;
        }
        if (!res[g].seeds.empty()) { local_have = true; last_chr = res[g].seeds.back().chr; last_end = res[g].seeds.back().pos + res[g].seeds.back().len; }
    }
    if (!ok) { seeds.clear(); if (sens) sens->clear(); return sequential(); }
    for (size_t g = 0; g < ng; ++g) seeds.insert(seeds.end(), res[g].seeds.begin(), res[g].seeds.end());
    if (sens) *sens = up;
    return SQ_OK;
}

// NormalizeSeedNodes + sanity checks + whole-genome tiling (SegmentGraph.cpp:19-38,706-761)
int tile_genome(sq_ctx* c, std::vector<Node>& seeds, std::vector<Node>& out) {
    const std::vector<int32_t>& RL = c->ref_len;
    if (seeds.size() >= 2) {
        std::sort(seeds.begin(), seeds.end(), [](const Node& a, const Node& b) {
            if (a.chr != b.chr) return a.chr < b.chr;
            if (a.pos != b.pos) return a.pos < b.pos;
            return a.len < b.len;
        });
        std::vector<Node> norm;
        for (const Node& n : seeds) {
            if (norm.empty() || norm.back().chr != n.chr || norm.back().pos + norm.back().len <= n.pos) norm.push_back(n);
            else norm.back().len = std::max(norm.back().pos + norm.back().len, n.pos + n.len) - norm.back().pos;
        }
        seeds.swap(norm);
    }
    for (size_t i = 0; i < seeds.size(); ++i) {
        bool ok = seeds[i].chr >= 0 && seeds[i].chr < (int)RL.size() && seeds[i].len > 0 && seeds[i].pos + seeds[i].len <= RL[seeds[i].chr];
        if (ok && i + 1 < seeds.size()) ok = seeds[i].chr != seeds[i + 1].chr || seeds[i].pos + seeds[i].len <= seeds[i + 1].pos;
        if (!ok) return fail(c, SQ_E_ASSERT, "seed segment violates the node sanity assert (SegmentGraph.cpp:708-712)");
    }
    out.clear();
    auto whole = [&](int chr) { out.push_back(Node{chr, 0, RL[chr], 0, 0.0}); };
    auto fill_tail = [&]() {  // rest of the current chromosome after the last tiled node
        const Node& b = out.back();
        if (b.pos + b.len != RL[b.chr]) out.push_back(Node{b.chr, b.pos + b.len, RL[b.chr] - b.pos - b.len, 0, 0.0});
    };
    for (Node n : seeds) {
        if (out.empty() || out.back().chr != n.chr) {
            if (!out.empty()) fill_tail();
            for (int k = out.empty() ? 0 : out.back().chr + 1; k != n.chr; ++k) whole(k);
            if (n.pos != 0) {
                if (n.pos > 100) out.push_back(Node{n.chr, 0, n.pos, 0, 0.0});
                else { n.len += n.pos; n.pos = 0; out.push_back(n); continue; }
            }
        }
        // (with nothing tiled yet and a seed at position 0 the reference reads tmpNodes.back() of an empty vector)
        const int end = out.empty() ? n.pos : out.back().pos + out.back().len;
        if (end < n.pos) {
            if (n.pos - end > 100) out.push_back(Node{n.chr, end, n.pos - end, 0, 0.0});
            else { n.len += n.pos - end; n.pos = end; }
        }
        out.push_back(n);
    }
    if (!out.empty()) fill_tail();
    for (int k = out.empty() ? 0 : out.back().chr + 1; k < (int)RL.size(); ++k) whole(k);
    return SQ_OK;
}

}  // namespace sq
