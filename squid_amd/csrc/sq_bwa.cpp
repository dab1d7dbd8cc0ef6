// `squid --bwa`: the single-file mode of the reference.  With BWA-MEM input there is no separate
// chimeric BAM: split reads are supplementary records of the one coordinate-sorted file, and the reference swaps three functions --
//   BuildNode_BWA             src/SegmentGraph.cpp:833-1205   (for BuildNode_STAR)
//   RawEdges                  src/SegmentGraph.cpp:1698-1930  (for RawEdgesChim + RawEdgesOther; it also REBUILDS Chimrecord from the
//                                                              partially aligned reads, :1883-1926)
//   the BAM loops' filters    MapQuality == 0 instead of < Min_MapQual (:871,1723), first mates only (:1723-1726)
// -- and shares everything from BuildEdges' sort on.  Division of labour here: the records are decoded WITH their QNAMEs (RawEdges
// sorts, merges and matches reads by name) -- by the GPU reader for files of 1 GiB and more (K-1 + K0 with sq_ctx::capture_names: the
// names kept next to the records on the device and copied back with them, round 5; 0.3-0.6 s for 50.8 M records where sixteen host
// threads took 3.3-4.5 s), on host threads otherwise --, the two order-dependent loops run on the host over that batch, and the graph from the edge reduction on takes the same kernels as the STAR
// path (filters, compression, components, ordering).  The breakpoint support (ExactBPConcordantSupport, mode-independent in the
// reference) is counted on the host, over the same batch.  Not the path BASELINE.json measures; built for drop-in completeness and
// held to the same parity bar (tests/test_bwa.py).
//
// Quirks of the reference reproduced (ledger W1-W6): capacity-driven window compaction, front() for [offset],
// the last Qname group never flushed, LocateRead hint left over from the BAM loop, weight -1 edges of multi-aligned second mates, the
// one-way depth cursor.
#include <algorithm>
#include <cstdint>
#include <chrono>
#include <cstring>

#include "sq_internal.h"
#include "sq_parsort.h"

namespace sq {

namespace {

struct RecRef {  // one record of the host batch, in the reference's vocabulary
    const HostBatch& hb; size_t i;
    int refid() const { return hb.refid[i]; }
    int pos() const { return hb.pos[i]; }
    int mrefid() const { return hb.mrefid[i]; }
    int mpos() const { return hb.mpos[i]; }
    int flag() const { return hb.flag[i]; }
    bool mapped() const { return !(flag() & 0x4); }
    bool mate_mapped() const { return !(flag() & 0x8); }
    bool rev() const { return flag() & 0x10; }
    bool mate_rev() const { return flag() & 0x20; }
    bool first() const { return flag() & 0x40; }
    bool proper() const { return flag() & 0x2; }
    bool dup() const { return flag() & 0x400; }
    bool multi() const { return hb.aux[i] & SQ_AUX_MULTI; }       // HasTag("XA") || IH > 1
    bool lowphred() const { return hb.aux[i] & SQ_AUX_LOWPHRED; } // of the record's own mate side
    int totlen() const { return hb.totlen[i]; }
    size_t nblk() const { return hb.blk_off[i + 1] - hb.blk_off[i]; }
    Blk blk(size_t k) const {
        const size_t b = hb.blk_off[i] + k;
        return Blk{refid(), hb.b_refpos[b], hb.b_readpos[b], hb.b_matchref[b], hb.b_matchread[b], rev(), first()};
    }
    std::string raw_name() const { return std::string(hb.names.data() + hb.name_off[i], hb.names.data() + hb.name_off[i + 1]); }
    std::string qname() const {  // ReadRec.cpp:11-13
        std::string q = raw_name();
        if (q.size() >= 2 && (q.compare(q.size() - 2, 2, "/1") == 0 || q.compare(q.size() - 2, 2, "/2") == 0)) q.resize(q.size() - 2);
        return q;
    }
    // the pair is "concordant" for the node builder (:1037-1040)
    bool pair_concordant() const {
        if (!(mapped() && mate_mapped() && mrefid() != -1 && refid() == mrefid() && proper())) return false;
        if (rev() && !mate_rev()) return pos() >= mpos() && pos() - mpos() <= 750000;
        if (!rev() && mate_rev()) return mpos() >= pos() && mpos() - pos() <= 750000;
        return false;
    }
};

// more than 15 bases of the read hang over at either end of its aligned blocks, and its qualities are fine (:1050-1065, :1730-1737)
inline bool clipped_end(const BlkList& r, int totlen, bool low) {
    if (r.empty() || low) return false;
    return r.front().readpos > 15 || totlen - r.back().readpos - r.back().matchread > 15;
}

// a sliding window of BuildNode_BWA: storage, front offset and the capacity the reference's compaction test looks at (W1)
struct Window {
    std::vector<Blk> v;
    int off = 0;
    size_t cap = 65536;
    bool none() const { return off == (int)v.size(); }
    const Blk& head() const { return v[(size_t)off]; }
    int end_of(size_t k) const { return v[k].refpos + v[k].matchref; }
    void drop_left_of(int refid, int pos, int read_len) { while (!none() && (head().refid != refid || head().refpos + head().matchref + read_len < pos)) ++off; }
    // chromosome of the oldest of the (at most four) newest elements, if there is one (:1002-1013)
    void vote_chr(int& chr) const { for (int i = (int)v.size() - 1; i >= off && (int)v.size() - i < 5; --i) chr = v[(size_t)i].refid; }
};

}  // namespace

// ---- BuildNode_BWA up to the seed nodes, plus the Reads list of :878-881
// The loop (:855-1114) is one automaton over all records, but everything it carries from record to record dies at a gap in the coverage:
// a record r that starts more than  Lmax + RL + 64  bases behind the start of the passing record in front of it (Lmax = the longest
// first block of the batch; or r opens a chromosome) finds -- after the work it triggers on the windows BEFORE it -- all three windows
// empty, the discordant run flushed, no pending mark, and every seed emitted so far more than 60 bases (thresh * 20, the only distance
// a seed is ever compared over) in front of anything it or its successors will look at.  So the stream is cut at such records: a
// stretch runs from its gap record with the state a fresh automaton has, and ends by letting the NEXT stretch's first record close it
// (the flush of the discordant window and the zero-coverage rule, :888-1026, without the record joining a window).  The one thing a
// fresh stretch does not reproduce is the fill of the windows' storage, which decides WHEN the capacity-driven compaction (W1) runs --
// not what the loop computes: the compaction only removes elements that end more than RL in front of the current record (and of the
// discordant run's head), which no later test can see: the clip-position and cover counts look at positions behind that head, emptiness
// and the voted chromosome do not change while an element that is NOT removable is still in front of them.
// SQUID_BWA_PIECE=<records> sets the stretch length (tests); samples of 200 k records and more are cut into 4 x threads stretches.
namespace {
struct SeedRun {
    int RL = 0, counted = 0, prev0 = 0, mark_start = -1, mark_chr = -1, dis_right = 0, other_right = 0;
    // what a stretch that was started on a guess must report (bwa_seed_nodes): were the two "rightmost" values assigned inside it, the
    // smallest position at which the zero-coverage test came out true by its position half while each was still the caller's, and the
    // outcome of that test for the record that closed the stretch
    bool dis_set = false, oth_set = false, closing_zero = true;
    int minpos_dis = INT32_MAX, minpos_oth = INT32_MAX;
    Window conc, dis, part;
    std::vector<Node> seeds;
    std::vector<Blk> reads;
    std::vector<int> margins;
};
inline bool seed_record_passes(const HostBatch& hb, size_t ri) {
    const RecRef r{hb, ri};
    return !(r.multi() || hb.mapq[ri] == 0 || r.dup() || !r.mapped() || r.refid() == -1) && r.nblk() != 0;
}
// one turn of the loop for record ri; `closing`: the record only closes the stretch in front of it (it belongs to the next one)
// opening_zero >= 0: the record's turn up to the zero-coverage rule was the closing turn of the stretch in front (which found `zero` = opening_zero)
void seed_step(SeedRun& S, const HostBatch& hb, size_t ri, bool closing, int opening_zero = -1) {
    const int thresh = 3;
    int &RL = S.RL, &prev0 = S.prev0, &mark_start = S.mark_start, &mark_chr = S.mark_chr, &dis_right = S.dis_right, &other_right = S.other_right;
    Window &conc = S.conc, &dis = S.dis, &part = S.part;
    std::vector<Node>& seeds = S.seeds;
    std::vector<int>& margins = S.margins;
    auto push_node = [&](int chr, int from, int to, int& cur_start, int& cur_end) {
        seeds.push_back(Node{chr, from, to - from, 0, 0.0});
        cur_start = to; cur_end = to; mark_start = to; mark_chr = chr;
    };
    const RecRef r{hb, ri};
    if (S.counted < 5 && !closing) { RL = std::max(RL, r.totlen()); ++S.counted; }  // :857-864 (over ALL records, in front of the filter)
    if (r.multi() || hb.mapq[ri] == 0 || r.dup() || !r.mapped() || r.refid() == -1) return;
    const bool opening = opening_zero >= 0;
    if (!opening && ((!dis.none() && r.refid() != dis.head().refid) || (!conc.none() && r.refid() != conc.head().refid) || (!part.none() && r.refid() != part.head().refid))) { other_right = 0; S.oth_set = true; }
    const size_t nb = r.nblk();
    if (nb == 0) return;
    if (!closing) for (size_t k = 0; k < nb; ++k) S.reads.push_back(r.blk(k));
    const Blk b0 = r.blk(0), blast = r.blk(nb - 1);
    if (!opening && conc.none() && part.none() && dis.none()) prev0 = r.pos();
    if (!opening && !dis.none() && (dis.v.back().refid != r.refid() || dis_right + RL < r.pos())) {
        // the discordant window is complete: decide the segment boundaries inside it (:888-998)
        int cur_end = 0, cur_start = std::max(prev0, mark_start);
        int d_start = -1, d_end = -1, d_count = -1;
        bool split = false;
        auto dense = [&]() { return d_start != -1 && !split && d_count > std::min(5.0, 4.0 * (d_end - d_start) / RL); };
        while (!dis.none()) {
            if (dense()) push_node(dis.head().refid, d_start, d_end, cur_start, cur_end);
            split = false;
            margins.clear();
            size_t i = (size_t)dis.off;
            for (; i < dis.v.size(); ++i) {  // the leading run of blocks that touch each other
                margins.push_back(dis.v[i].refpos); margins.push_back(dis.end_of(i));
                cur_end = std::max(cur_end, margins.back());
                if (i + 1 < dis.v.size() && dis.v[i + 1].refpos > dis.end_of(i)) break;
            }
            d_start = std::max(cur_start, dis.head().refpos);
            d_end = cur_end;
            d_count = (int)i - dis.off;
            for (++i; i < dis.v.size() && dis.v[i].refpos < cur_end + thresh; ++i) { margins.push_back(dis.v[i].refpos); margins.push_back(dis.end_of(i)); }
            const int m0 = margins.front(), dchr = dis.head().refid;
            for (size_t k = (size_t)part.off; k < part.v.size(); ++k) {  // clip positions of the partially aligned reads next to the run
                const Blk& p = part.v[k];
                if (p.refid != dchr) continue;
                const int pe = p.refpos + p.matchref;
                if (p.readpos > 15 && p.refpos > m0 - thresh && p.refpos < cur_end + thresh) margins.push_back(p.rev ? pe : p.refpos);
                else if (pe > m0 - thresh && pe < cur_end + thresh) margins.push_back(p.rev ? p.refpos : pe);
            }
            std::sort(margins.begin(), margins.end());
            int last_cursor = -1, last_support = 0;
            const int chr0 = dis.v.front().refid;  // (W2: element 0 of the storage, not the window's head)
            for (size_t at = 0; at < margins.size();) {
                const int x = margins[at];
                if (!seeds.empty() && seeds.back().chr == chr0 && x - seeds.back().pos - seeds.back().len < thresh * 20) { ++at; continue; }
                int sr = 0, left_fwd = 0, right_rev = 0;
                for (size_t q = 0; q < margins.size() && margins[q] < x + thresh; ++q) sr += std::abs(x - margins[q]) < thresh;
                for (size_t q = (size_t)dis.off; q < dis.v.size(); ++q) {
                    const int e = dis.end_of(q);
                    if (e < x && e > x - RL && !dis.v[q].rev) ++left_fwd;
                    else if (dis.v[q].refpos > x && dis.v[q].refpos < x + RL && dis.v[q].rev) ++right_rev;
                }
                bool cut_here = false;
                if (sr > 3 || sr + left_fwd > 4 || sr + right_rev > 4) {
                    int cover = 0;
                    for (size_t q = (size_t)conc.off; q < conc.v.size(); ++q) cover += conc.end_of(q) >= x + thresh && conc.v[q].refpos < x - thresh;
                    if (sr > std::max(cover - sr, 0) + 2) {
                        const int strength = sr + std::max(left_fwd, right_rev);
                        if (last_cursor == -1 && x - cur_start < thresh * 20) { mark_start = cur_start; mark_chr = chr0; }
                        else if ((last_cursor == -1 || x - last_cursor < thresh * 20) && strength > last_support) { last_cursor = x; last_support = strength; }
                        else if (x - last_cursor >= thresh * 20) { split = true; push_node(chr0, cur_start, last_cursor, cur_start, cur_end); cut_here = true; }
                    }
                }
                if (cut_here) break;
                size_t nx = at;
                while (nx < margins.size() && margins[nx] == x) ++nx;  // on to the next distinct position
                if (nx >= margins.size()) break;
                at = nx;
            }
            if (last_cursor != -1 && !split) { split = true; push_node(dis.head().refid, cur_start, last_cursor, cur_start, cur_end); }
            while (!dis.none() && dis.head().refpos + dis.head().matchref <= cur_end) ++dis.off;
        }
        if (dense()) push_node(dis.v[0].refid, d_start, d_end, cur_start, cur_end);  // (W2)
        if (dis.none()) { dis.v.clear(); dis.off = 0; }
        conc.drop_left_of(r.refid(), r.pos(), RL); part.drop_left_of(r.refid(), r.pos(), RL);
    }
    // zero coverage in front of this record (:1000-1026)
    const int rightmost = std::max(dis_right, other_right);
    int cur_chr = 0;
    conc.vote_chr(cur_chr); part.vote_chr(cur_chr); dis.vote_chr(cur_chr);
    const bool zero = opening ? opening_zero != 0 : (r.refid() != cur_chr || r.pos() > rightmost + RL);
    if (!opening && zero && r.refid() == cur_chr) {  // (true by its position half alone: a larger value handed in by the caller could have turned it)
        if (!S.dis_set) S.minpos_dis = std::min(S.minpos_dis, r.pos());
        if (!S.oth_set) S.minpos_oth = std::min(S.minpos_oth, r.pos());
    }
    if (closing) S.closing_zero = zero;
    if (!opening && zero && mark_start != -1) {
        if (rightmost > mark_start && rightmost - mark_start < thresh * 20 && !seeds.empty() && mark_start == seeds.back().pos + seeds.back().len) seeds.back().len += rightmost - mark_start;
        else if (rightmost > mark_start && rightmost - mark_start >= thresh * 20) seeds.push_back(Node{mark_chr, mark_start, rightmost - mark_start, 0, 0.0});
        mark_start = -1; mark_chr = -1;
    }
    if (closing) return;
    if (zero) prev0 = r.pos();
    if (dis.none()) { conc.drop_left_of(r.refid(), r.pos(), RL); part.drop_left_of(r.refid(), r.pos(), RL); }
    // the record joins a window (:1035-1086)
    const int e0 = b0.refpos + b0.matchref;
    if (r.pair_concordant()) {
        S.oth_set = true;  // (the first such record of a stretch finds both windows empty: assigned, not compared)
        other_right = (!conc.none() || !part.none()) ? std::max(other_right, e0) : e0;
        const bool clipped = !r.lowphred() && (b0.readpos > 15 || r.totlen() - blast.readpos - blast.matchread > 15);
        (clipped ? part : conc).v.push_back(b0);
    } else {
        S.dis_set = true;
        dis_right = !dis.v.empty() ? std::max(dis_right, e0) : e0;
        dis.v.push_back(b0);
    }
    // compaction at capacity (:1087-1112, W1)
    for (Window* w : {&conc, &part}) {
        if (w->v.size() != w->cap) continue;
        const int from = !dis.none() ? std::min(r.pos(), dis.head().refpos) : r.pos();
        std::vector<Blk> kept;
        for (size_t q = (size_t)w->off; q < w->v.size(); ++q) if (w->v[q].refid == r.refid() && w->end_of(q) + RL >= from) kept.push_back(w->v[q]);
        w->v.swap(kept); w->off = 0;
        if (w->v.size() == w->cap) w->cap *= 2;
    }
}
}  // namespace
static int bwa_seed_nodes(sq_ctx* c, const HostBatch& hb, std::vector<Node>& seeds, std::vector<std::vector<Blk>>& reads) {
    const size_t nrec = hb.size();
    // ReadLen as the loop leaves it (:857-864): the chimeric file's value, raised by the first five records
    int RL_final = c->read_len;
    for (size_t ri = 0; ri < nrec && ri < 5; ++ri) RL_final = std::max(RL_final, (int)hb.totlen[ri]);
    const long piece_env = std::getenv("SQUID_BWA_PIECE") ? std::atol(std::getenv("SQUID_BWA_PIECE")) : 0;
    const int threads = c->pool ? c->pool->size() + 1 : 1;
    std::vector<size_t> cut{0};
    if (threads > 1 && nrec > 16 && (piece_env > 0 || nrec >= 200000)) {
        // even pieces; per piece the largest (RefID, end of the first block) of its passing records -- the records are sorted, so the
        // running maximum of that key in front of a record is "the farthest a window element of the current chromosome reaches" --,
        // then every piece looks for its first passing record that starts more than RL + 64 behind that reach (or on a later chromosome)
        const int np0 = (int)std::min<size_t>(piece_env > 0 ? std::max<size_t>(1, nrec / (size_t)piece_env) : (size_t)(4 * threads), nrec / 8);
        auto key_of = [&](size_t ri) { const uint32_t b = hb.blk_off[ri]; return ((long long)hb.refid[ri] << 32) | (uint32_t)(hb.b_refpos[b] + hb.b_matchref[b]); };
        auto lo_of = [&](int k) { return nrec * (size_t)k / (size_t)np0; };
        std::vector<long long> reach((size_t)np0 + 1, -1);
        c->pool->parallel_for(np0, 1 << 20, [&](int k) {
            long long m = -1;
            for (size_t ri = lo_of(k); ri < lo_of(k + 1); ++ri) if (seed_record_passes(hb, ri)) m = std::max(m, key_of(ri));
            reach[(size_t)k + 1] = m;
        });
        for (int k = 0; k < np0; ++k) reach[(size_t)k + 1] = std::max(reach[(size_t)k + 1], reach[(size_t)k]);  // reach[k]: of everything in front of piece k
        std::vector<size_t> found((size_t)np0, 0);
        c->pool->parallel_for(np0, 1 << 20, [&](int k) {
            if (k == 0 || reach[(size_t)k] < 0) return;  // (a stretch starts behind the first five records, and behind a passing record)
            long long run = reach[(size_t)k];
            for (size_t ri = std::max<size_t>(lo_of(k), 8); ri < lo_of(k + 1); ++ri) {
                if (!seed_record_passes(hb, ri)) continue;
                const int chr = (int)(run >> 32), end = (int)(uint32_t)run;
                if (hb.refid[ri] != chr || (long long)hb.pos[ri] > (long long)end + RL_final + 64) { found[(size_t)k] = ri; return; }
                run = std::max(run, key_of(ri));
            }
        });
        for (int k = 1; k < np0; ++k) if (found[(size_t)k] > cut.back()) cut.push_back(found[(size_t)k]);
    }
    cut.push_back(nrec);
    const int np = (int)cut.size() - 1;
    c->timer.add("bwa_seed_node_stretches", 0.0, 0.0, np);
    // A stretch is started on a GUESS of the values that do cross a gap: the outcome of the zero-coverage test for its first record (guess:
    // true -- then prev0 and the mark are reset by that record itself) and the two "rightmost" values the test compares with (the
    // discordant one: computed in advance, above; the other one: 0).
    // They are stale whenever they matter -- the end of the last discordant / concordant run in front, possibly on an EARLIER CHROMOSOME:
    // the reference compares positions of different chromosomes there (DiscordantRightmost outlives the chromosome, :1002-1013) -- and
    // neither depends on anything the test decides, so every stretch reports what it assigned, the stretches are then walked in order
    // with the real values, and a stretch whose guess was wrong where it counted is run again from the real state.
    struct Carry { int zero, prev0, mark_start, mark_chr, dis_right, other_right; };
    // DiscordantRightmost is known in advance: it follows a rule of its own -- a discordant record assigns it (or raises it while the
    // discordant window holds something), and the window is emptied by the first record on another chromosome or more than RL behind it --
    // that a pass over (filter, pair type, end of the first block) reproduces, stretch by stretch, each from an empty window (a gap
    // record finds it empty or empties it).  A stretch without a discordant record hands on what it was given.
    std::vector<int> dis_in((size_t)np + 1, 0);
    if (np > 1) {
        std::vector<std::pair<bool, int>> sum((size_t)np, std::make_pair(false, 0));
        c->pool->parallel_for(np, 1 << 20, [&](int k) {
            bool nonempty = false, has = false;
            int dr = 0, last = -1, RL = k == 0 ? c->read_len : RL_final;
            for (size_t ri = cut[(size_t)k]; ri < cut[(size_t)k + 1]; ++ri) {
                if (k == 0 && ri < 5) RL = std::max(RL, (int)hb.totlen[ri]);
                if (!seed_record_passes(hb, ri)) continue;
                const RecRef r{hb, ri};
                if (nonempty && (last != r.refid() || dr + RL < r.pos())) nonempty = false;
                if (!r.pair_concordant()) { const Blk b0 = r.blk(0); const int e0 = b0.refpos + b0.matchref; dr = nonempty ? std::max(dr, e0) : e0; nonempty = true; last = r.refid(); has = true; }
            }
            sum[(size_t)k] = std::make_pair(has, dr);
        });
        for (int k = 0; k < np; ++k) dis_in[(size_t)k + 1] = sum[(size_t)k].first ? sum[(size_t)k].second : dis_in[(size_t)k];
    }
    std::vector<SeedRun> runs((size_t)np);
    auto work = [&](int k, const Carry* real) {
        SeedRun& S = runs[(size_t)k];
        S = SeedRun();
        S.RL = k == 0 ? c->read_len : RL_final;
        S.counted = k == 0 ? 0 : 5;
        S.dis_right = dis_in[(size_t)k];
        if (real) { S.prev0 = real->prev0; S.mark_start = real->mark_start; S.mark_chr = real->mark_chr; S.dis_right = real->dis_right; S.other_right = real->other_right; }
        size_t nblk = 0;
        for (size_t ri = cut[(size_t)k]; ri < cut[(size_t)k + 1]; ++ri) nblk += hb.blk_off[ri + 1] - hb.blk_off[ri];
        S.reads.reserve(nblk);  // (at most every block of the stretch: one allocation)
        for (size_t ri = cut[(size_t)k]; ri < cut[(size_t)k + 1]; ++ri) seed_step(S, hb, ri, false, (k > 0 && ri == cut[(size_t)k]) ? (real ? real->zero : 1) : -1);
        if (k + 1 < np) seed_step(S, hb, cut[(size_t)k + 1], true);  // (the next stretch's gap record closes this one)
    };
    if (np > 1) c->pool->parallel_for(np, 1 << 20, [&](int k) { work(k, nullptr); }); else work(0, nullptr);
    int again = 0;
    for (int k = 1; k < np; ++k) {
        const SeedRun& P = runs[(size_t)k - 1];  // (final: run from the real state, or checked)
        const Carry real{P.closing_zero ? 1 : 0, P.prev0, P.mark_start, P.mark_chr, P.dis_right, P.other_right};
        SeedRun& S = runs[(size_t)k];
        // with `zero` true the first record resets prev0 and the mark by itself; the rightmost values only enter through the position half
        // of later tests, and only until the stretch assigns them (smaller than every position a test came out true at: same outcomes)
        const bool fine = real.zero && real.dis_right == dis_in[(size_t)k] &&
                          (real.other_right == 0 || S.minpos_oth == INT32_MAX || (long)real.other_right + RL_final < (long)S.minpos_oth);
        if (fine) {  // (what the stretch did not assign stays the caller's)
            if (!S.oth_set) S.other_right = real.other_right;
            continue;
        }
        if (std::getenv("SQUID_BWA_DEBUG")) std::fprintf(stderr, "stretch %d (%d, %d) again: zero %d, rightmost values %d / %d against first true tests at %d / %d\n", k, hb.refid[cut[(size_t)k]], hb.pos[cut[(size_t)k]], real.zero, real.dis_right, real.other_right, S.minpos_dis, S.minpos_oth);
        work(k, &real);
        ++again;
    }
    c->timer.add("bwa_seed_node_stretches_run_again", 0.0, 0.0, again);
    if (std::getenv("SQUID_BWA_DEBUG")) {
        for (int k = 0; k < np; ++k) {
            std::fprintf(stderr, "stretch %d: records [%zu, %zu) first (%d, %d); seeds:", k, cut[(size_t)k], cut[(size_t)k + 1], hb.refid[cut[(size_t)k]], hb.pos[cut[(size_t)k]]);
            for (const Node& n : runs[(size_t)k].seeds) std::fprintf(stderr, " (%d %d %d)", n.chr, n.pos, n.len);
            std::fprintf(stderr, "\n");
        }
    }
    reads.clear();
    for (SeedRun& S : runs) { seeds.insert(seeds.end(), S.seeds.begin(), S.seeds.end()); reads.push_back(std::move(S.reads)); }
    c->read_len = runs.back().RL;
    return SQ_OK;
}

// Support / AvgDepth of the tiled nodes (:1180-1204): one pass, the cursor never goes back (W6)
static void bwa_node_depth(std::vector<Node>& N, const std::vector<std::vector<Blk>>& parts) {
    size_t pi = 0, it = 0;
    auto skip_empty = [&]() { while (pi < parts.size() && it == parts[pi].size()) { ++pi; it = 0; } };
    skip_empty();
    if (pi == parts.size()) return;  // (no read at all: the reference leaves Support / AvgDepth as constructed)
    for (Node& n : N) {
        int cnt = 0, sum = 0;
        for (; pi < parts.size(); ++it, skip_empty()) {
            const Blk& b = parts[pi][it];
            if (b.refid == n.chr && b.refpos >= n.pos && b.refpos + b.matchref <= n.pos + n.len) { ++cnt; sum += b.matchref; }
            else if (b.refpos >= n.pos + n.len || b.refid != n.chr) break;
        }
        n.support = cnt;
        n.depth = 1.0 * sum / n.len;
        n.depth_lo = n.depth_hi = n.depth;
    }
}

static int bwa_home_node(const std::vector<Node>& N, int start, const Blk& b) {  // the two walks of :1757-1758 as binary searches
    const int n = (int)N.size();
    const int j0 = (int)(std::partition_point(N.begin(), N.end(), [&](const Node& x) { return x.chr < b.refid || (x.chr == b.refid && x.pos + x.len < b.refpos); }) - N.begin());
    const int i = std::max(start, j0);
    if (i >= n) return -2;
    const int j1 = (int)(std::partition_point(N.begin(), N.end(), [&](const Node& x) { return x.chr < b.refid || (x.chr == b.refid && x.pos <= b.refpos); }) - N.begin()) - 1;
    return std::min(i, j1);
}

// ---- RawEdges (:1698-1930): the edges of the BAM loop, the -1 edges of multi-aligned second mates, the fragments rebuilt from the
// partially aligned reads and their split edges
static int bwa_raw_edges(sq_ctx* c, const HostBatch& hb, std::vector<Edge>& raw) {
    const std::vector<Node>& N = c->nodes;
    const int nn = (int)N.size();
    auto in_range = [&](int i) { return i >= 0 && i < nn; };
    auto discordant = [&](const Edge& e) { return edge_discordant(c, N, e); };
    // What the BAM loop (:1712-1880) leaves behind, per stretch of records: the loop carries ONE thing from record to record -- the
    // position LocateRead starts from (`hint`: the node of the last located first block) -- and appends to lists whose order is either
    // the record order (PartialAlign: sorted by name afterwards with an unstable sort, so the order going in counts) or does not matter
    // (the edges are sorted and summed, the names of FirstDisInserted are sorted, the -1 edges are looked up one by one).  A stretch can
    // therefore start behind any record whose first block lies deep inside ONE node -- LocateRead ends there from any start (sq_graph.cpp,
    // frag_first_block_pins) --, and the stretches are worked on side by side and strung together in order.
    struct Piece {
        int hint = 0;
        std::vector<Edge> raw, second_edges;
        std::vector<Frag> partial;
        std::vector<std::string> first_dis, second_names;
        int rc = SQ_OK; const char* err = nullptr;
    };
    // the fragment of record ri as the loop builds it up to its LocateRead call: 0 = the record is skipped or locates nothing,
    // 1 = first mate (:1745-1806), 2 = multi-aligned second mate (:1807-1860); `part`: it also goes to PartialAlign
    auto prepare = [&](size_t ri, Frag& f, bool& part) -> int {
        const RecRef r{hb, ri};
        part = false;
        if (r.dup() || !r.mapped()) return 0;
        if (r.first() ? (r.multi() || hb.mapq[ri] == 0) : !r.multi()) return 0;  // :1723-1726 (W5)
        BlkList& own = r.first() ? f.a : f.b;  // (f.name: given by the caller where a list keeps it -- two std::strings per record otherwise, most of this loop's time)
        for (size_t k = 0; k < r.nblk(); ++k) own.push_back(r.blk(k));
        std::sort(own.begin(), own.end(), blk_less_readpos);
        (r.first() ? f.atot : f.btot) = r.totlen();
        (r.first() ? f.alow : f.blow) = r.lowphred();
        part = !r.multi() && (clipped_end(f.a, f.atot, f.alow) || clipped_end(f.b, f.btot, f.blow));
        return r.first() ? ((!f.a.empty() && (f.a.front().readpos <= 15 || f.alow)) ? 1 : 0) : (!f.b.empty() ? 2 : 0);
    };
    // (the mate stub and, for a second mate, the shortened own block: what LocateRead sees -- after the copy for PartialAlign was taken)
    auto finish_prepare = [&](size_t ri, Frag& f, int kind) {
        const RecRef r{hb, ri};
        if (r.mate_mapped() && r.mrefid() != -1) (r.first() ? f.b : f.a).push_back(Blk{r.mrefid(), r.mpos(), 0, 15, 15, r.mate_rev(), false});
        if (kind == 2) { f.b.resize(1); f.b[0].matchref = 15; f.b[0].matchread = 15; }
    };
    auto run = [&](size_t lo, size_t hi, Piece& P) {
        std::vector<int> rn;
        int& hint = P.hint;
        auto add = [&](int i, bool hi_, int j, bool hj, int w) -> bool {
            if (!in_range(i) || !in_range(j)) { P.rc = SQ_E_ASSERT; P.err = "an edge would leave the node table (the reference asserts, SegmentGraph.cpp:1760)"; return false; }
            P.raw.push_back(make_edge(i, hi_, j, hj, w));
            return true;
        };
        auto split_edges = [&](const BlkList& r, size_t base) -> bool {
            for (size_t k = 0; k + 1 < r.size(); ++k) {
                const int i = rn[base + k], j = rn[base + k + 1];
                if (i != j && i != -1 && j != -1 && !add(i, r[k].rev, j, !r[k + 1].rev, 1)) return false;
            }
            return true;
        };
        Frag f;  // (one object for the stretch, emptied per record: its two block lists keep their storage -- a fresh Frag per record was two or three allocations per record on every thread)
        for (size_t ri = lo; ri < hi; ++ri) {
            f.a.clear(); f.b.clear(); f.name.clear(); f.atot = 0; f.btot = 0; f.alow = false; f.blow = false;
            bool part;
            const int kind = prepare(ri, f, part);
            if (part) { f.name = RecRef{hb, ri}.qname(); P.partial.push_back(f); }
            if (kind == 0) continue;  // (a record that locates nothing leaves nothing else behind: its mate stub is only looked at by LocateRead)
            finish_prepare(ri, f, kind);
            const size_t na = f.a.size();
            if (kind == 1) {
                locate_fragment(N, hint, f, rn);
                if (rn[0] != -1) hint = rn[0];
                for (size_t k = 0; k < rn.size(); ++k)
                    if (rn[k] == -1) {
                        const int i = bwa_home_node(N, hint, k < na ? f.a[k] : f.b[k - na]);
                        if (i == -2) { P.rc = SQ_E_ASSERT; P.err = "a block lies behind the last node (the reference reads past its node table, SegmentGraph.cpp:1757)"; return; }
                        if (!add(i, false, i + 1, true, 1)) return;
                    }
                if (!split_edges(f.a, 0) || !split_edges(f.b, na)) return;
                if (!f.b.empty() && !frag_end_discordant(f, true) && !frag_end_discordant(f, false)) {
                    const int i = rn[na - 1], j = rn.back();
                    if (i != j && i != -1 && j != -1 && !pair_overlap(f, rn, i, j)) {
                        if (!add(i, f.a.back().rev, j, f.b.back().rev, 1)) return;
                        if (discordant(P.raw.back())) P.first_dis.push_back(RecRef{hb, ri}.qname());
                    }
                }
            } else {
                locate_fragment(N, hint, f, rn);
                if (rn[0] != -1) hint = rn[0];
                if (!f.a.empty() && !frag_end_discordant(f, true)) {
                    const int i = rn[f.a.size() - 1], j = rn.back();
                    bool overlap = false;
                    for (size_t k = 0; k < f.a.size(); ++k) overlap |= j == rn[k];
                    overlap |= i == rn[f.a.size()];
                    if (i != j && i != -1 && j != -1 && !overlap) {
                        if (!in_range(i) || !in_range(j)) { P.rc = SQ_E_ASSERT; P.err = "an edge would leave the node table (the reference asserts, SegmentGraph.cpp:1852)"; return; }
                        const Edge e = make_edge(i, f.a.back().rev, j, f.b.back().rev, -1);
                        if (discordant(e)) { P.second_names.push_back(RecRef{hb, ri}.qname()); P.second_edges.push_back(e); }
                    }
                }
            }
        }
    };
    const auto t_re0 = std::chrono::steady_clock::now();
    auto re_lap = [&](const char* what) { if (std::getenv("SQUID_BWA_DEBUG")) std::fprintf(stderr, "RawEdges: %-28s at %8.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_re0).count()); };
    // stretch boundaries: behind the nearest record in front of an even cut that pins the position (looked for among the 4096 records
    // in front of the cut; none there: that cut is left out).  SQUID_BWA_PIECE=<records> sets the stretch length (tests: small inputs)
    const long piece_env = std::getenv("SQUID_BWA_PIECE") ? std::atol(std::getenv("SQUID_BWA_PIECE")) : 0;
    const size_t nrec = hb.size();
    const int threads = c->pool ? c->pool->size() + 1 : 1;
    std::vector<size_t> cut{0};
    std::vector<int> start{0};
    if (threads > 1 && (piece_env > 0 || nrec >= 200000)) {
        const size_t want = piece_env > 0 ? std::max<size_t>(1, nrec / (size_t)piece_env) : (size_t)(4 * threads);
        for (size_t k = 1; k < want; ++k) {
            const size_t at = nrec * k / want;
            if (at <= cut.back()) continue;
            for (size_t q = at; q-- > cut.back() && at - q <= 4096;) {
                Frag f;
                bool part;
                const int kind = prepare(q, f, part);
                if (kind == 0) continue;
                finish_prepare(q, f, kind);
                int node = -1;
                if (frag_first_block_pins(N, f, node)) { cut.push_back(q + 1); start.push_back(node); break; }
            }
        }
    }
    cut.push_back(nrec);
    re_lap("stretches planned");
    const int np = (int)cut.size() - 1;
    c->timer.add("bwa_raw_edge_stretches", 0.0, 0.0, np);  // (how many stretches the loop ran in: tests)
    std::vector<Piece> pieces((size_t)np);
    for (int k = 0; k < np; ++k) pieces[(size_t)k].hint = start[(size_t)k];
    // (the edges of a stretch are summed per key before they are strung together -- BuildEdges sorts the list and adds the weights of equal
    // keys up, :1943-1957, so neither the order nor the grouping of the addends shows; tens of millions of unit edges shrink to the
    // distinct keys of the stretch.  Nothing is dropped here: a sum <= 0 is only final when every stretch and the -1 edges are in)
    std::vector<int64_t> emitted((size_t)np, 0);
    auto presum = [](std::vector<Edge>& v) {
        std::sort(v.begin(), v.end(), edge_key_less);
        size_t o = 0;
        for (size_t i = 0; i < v.size(); ++i) { if (o && edge_key_eq(v[i], v[o - 1])) v[o - 1].w += v[i].w; else v[o++] = v[i]; }
        v.resize(o);
    };
    if (np > 1) c->pool->parallel_for(np, 1 << 20, [&](int k) { run(cut[(size_t)k], cut[(size_t)k + 1], pieces[(size_t)k]); emitted[(size_t)k] = (int64_t)pieces[(size_t)k].raw.size(); if (!pieces[(size_t)k].rc) presum(pieces[(size_t)k].raw); });
    else { run(0, nrec, pieces[0]); emitted[0] = (int64_t)pieces[0].raw.size(); }
    int64_t n_emitted = 0;
    for (int64_t e : emitted) n_emitted += e;
    re_lap("record loop");
    int hint = 0;
    std::vector<Frag> partial;
    std::vector<std::string> first_dis, second_names;
    std::vector<Edge> second_edges;
    std::vector<int> rn;
    for (Piece& P : pieces) {
        if (P.rc) return fail(c, P.rc, P.err);  // (the first stretch in record order that ran into one: what the loop in one go would have hit first)
        raw.insert(raw.end(), P.raw.begin(), P.raw.end());
        partial.insert(partial.end(), std::make_move_iterator(P.partial.begin()), std::make_move_iterator(P.partial.end()));
        first_dis.insert(first_dis.end(), std::make_move_iterator(P.first_dis.begin()), std::make_move_iterator(P.first_dis.end()));
        second_names.insert(second_names.end(), std::make_move_iterator(P.second_names.begin()), std::make_move_iterator(P.second_names.end()));
        second_edges.insert(second_edges.end(), P.second_edges.begin(), P.second_edges.end());
        hint = P.hint;
    }
    const size_t raw_strung = raw.size();
    re_lap("stretches strung");
    auto add = [&](int i, bool hi, int j, bool hj, int w) -> int {
        if (!in_range(i) || !in_range(j)) return fail(c, SQ_E_ASSERT, "an edge would leave the node table (the reference asserts, SegmentGraph.cpp:1760)");
        raw.push_back(make_edge(i, hi, j, hj, w));
        return SQ_OK;
    };
    auto split_edges = [&](const BlkList& r, size_t base) -> int {
        for (size_t k = 0; k + 1 < r.size(); ++k) {
            const int i = rn[base + k], j = rn[base + k + 1];
            if (i != j && i != -1 && j != -1) { const int rc = add(i, r[k].rev, j, !r[k + 1].rev, 1); if (rc) return rc; }
        }
        return SQ_OK;
    };
    std::sort(first_dis.begin(), first_dis.end());
    for (size_t k = 0; k < second_names.size(); ++k)
        if (std::binary_search(first_dis.begin(), first_dis.end(), second_names[k])) raw.push_back(second_edges[k]);
    // the fragments of the partially aligned reads: grouped by name (the sort of :1883 is libstdc++'s introsort on the names, ledger B8),
    // merged, stored untrimmed, located from the hint the loop above left behind (W4); the last group is dropped (W3)
    // (what is sorted is the index of every fragment with the same comparison: introsort takes the same decisions, hence the same order, without
    // moving 100-byte objects around; `name < name` is a strict weak order, so the threaded form with the split final pass applies, sq_parsort.h)
    std::vector<uint32_t> by_name(partial.size());
    for (size_t i = 0; i < by_name.size(); ++i) by_name[i] = (uint32_t)i;
    std_sort_parallel(by_name.begin(), by_name.end(), [&](uint32_t x, uint32_t y) { return partial[x].name < partial[y].name; }, c->pool ? std::min(c->pool->size() + 1, 32) : 1, true);
    re_lap("partial reads sorted by name");
    std::vector<Frag> rebuilt;
    Frag cur;
    for (const uint32_t pi : by_name) {
        const Frag& p = partial[pi];
        if (cur.a.empty() && cur.b.empty()) { cur = p; continue; }
        if (cur.name == p.name) { cur.a.insert(cur.a.end(), p.a.begin(), p.a.end()); cur.b.insert(cur.b.end(), p.b.begin(), p.b.end()); continue; }
        std::sort(cur.a.begin(), cur.a.end(), blk_less_readpos);
        std::sort(cur.b.begin(), cur.b.end(), blk_less_readpos);
        if (cur.a.size() > 1 || cur.b.size() > 1) {
            rebuilt.push_back(cur);
            locate_fragment(N, hint, cur, rn);
            int rc = split_edges(cur.a, 0);
            if (!rc) rc = split_edges(cur.b, cur.a.size());
            if (rc) return rc;
        }
        cur = p;
    }
    // ReadRec_t::FrontSmallerThan (ReadRec.cpp:90-117; not a strict weak order, kept as it is)
    std::sort(rebuilt.begin(), rebuilt.end(), [](const Frag& x, const Frag& y) {
        const Blk* p = !x.a.empty() && !y.a.empty() ? &x.a.front() : !x.b.empty() && !y.b.empty() ? &x.b.front() : !x.a.empty() && !y.b.empty() ? &x.a.front() : !x.b.empty() && !y.a.empty() ? &x.b.front() : nullptr;
        const Blk* q = !x.a.empty() && !y.a.empty() ? &y.a.front() : !x.b.empty() && !y.b.empty() ? &y.b.front() : !x.a.empty() && !y.b.empty() ? &y.b.front() : !x.b.empty() && !y.a.empty() ? &y.a.front() : nullptr;
        return p && q && blk_less_pos(*p, *q);
    });
    c->frags = rebuilt;
    c->frags0 = c->frags;
    // the name set ExactBPConcordantSupport tests raw QNAMEs against (:3113-3118; the sized-then-appended vector also holds "", ledger B9)
    c->chim_names.clear();
    if (!rebuilt.empty()) c->chim_names.push_back(std::string());
    for (const Frag& f : rebuilt) c->chim_names.push_back(f.name);
    std::sort(c->chim_names.begin(), c->chim_names.end());
    c->chim_names.erase(std::unique(c->chim_names.begin(), c->chim_names.end()), c->chim_names.end());
    c->counts.n_chim_fragments = (int64_t)rebuilt.size();
    re_lap("fragments rebuilt");
    c->counts.n_raw_edges = n_emitted + (int64_t)(raw.size() - raw_strung);  // (as the loops emitted them: the stretches' edges arrive summed)
    return SQ_OK;
}

// BuildNode_BWA + RawEdges: c->nodes, c->frags and the raw edge list
int bwa_nodes_and_edges(sq_ctx* c, std::vector<Edge>& raw) {
    if (!c->bwa) return fail(c, SQ_E_ARG, "sq_ingest_bwa_file first");
    const HostBatch& hb = *c->bwa;
    std::vector<Node> seeds;
    std::vector<std::vector<Blk>> reads;  // (the Reads list, stretch by stretch)
    int rc;
    { HostClock hc(c, "host_bwa_seed_nodes"); rc = bwa_seed_nodes(c, hb, seeds, reads); }
    if (rc) return rc;
    c->counts.read_len = c->read_len;
    { HostClock hc(c, "host_tile_genome"); rc = tile_genome(c, seeds, c->nodes); }
    if (rc) return rc;
    { HostClock hc(c, "host_bwa_node_depth"); bwa_node_depth(c->nodes, reads); }
    c->snap[1].take(c->nodes, std::vector<Edge>(), nullptr);
    raw.clear();
    { HostClock hc(c, "host_bwa_raw_edges"); rc = bwa_raw_edges(c, hb, raw); }
    return rc;
}

// ExactBPConcordantSupport's counting loop (:3129-3166) over the host batch: `bps` sorted (chr, pos)
int bwa_breakpoint_support(sq_ctx* c, const std::vector<std::pair<int, int>>& bps, std::vector<int32_t>& cov) {
    if (!c->bwa) return fail(c, SQ_E_ARG, "sq_ingest_bwa_file first");
    const HostBatch& hb = *c->bwa;
    HostClock hc(c, "host_bwa_bp_support");
    cov.assign(bps.size(), 0);
    if (bps.empty()) return SQ_OK;
    // which records the loop looks at at all is decided per record -- its own flags, its mate's place, its QNAME against the name set of the
    // rebuilt fragments -- before the loop touches the one thing it carries (the cursor): decided side by side on the host threads (the
    // name test, a binary search per record, was 6.5 s of one thread for C3's 50.8 M records), and the loop then walks the survivors
    const std::vector<std::string>& names = c->chim_names;
    auto in_names = [&](size_t ri) {  // std::binary_search(names, raw name) without building the std::string
        const char* p = hb.names.data() + hb.name_off[ri];
        const size_t L = hb.name_off[ri + 1] - hb.name_off[ri];
        size_t lo = 0, hi = names.size();
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            const std::string& m = names[mid];
            const int cmp = std::memcmp(m.data(), p, std::min(m.size(), L));
            if (cmp < 0 || (cmp == 0 && m.size() < L)) lo = mid + 1; else hi = mid;
        }
        return lo < names.size() && names[lo].size() == L && std::memcmp(names[lo].data(), p, L) == 0;
    };
    // (in front of the binary search: one bit per name hash -- 15 k names in 4 M bits; all but one record in a few thousand end here)
    auto hash_of = [](const char* p, size_t L) { unsigned long long h = 1469598103934665603ull; for (size_t i = 0; i < L; ++i) { h ^= (unsigned char)p[i]; h *= 1099511628211ull; } h ^= h >> 29; return h; };
    const size_t nbits = (size_t)1 << 22;
    std::vector<uint64_t> maybe(nbits / 64, 0);
    for (const std::string& nm : names) { const unsigned long long h = hash_of(nm.data(), nm.size()) & (nbits - 1); maybe[h >> 6] |= 1ull << (h & 63); }
    auto in_names_fast = [&](size_t ri) {
        const char* p = hb.names.data() + hb.name_off[ri];
        const unsigned long long h = hash_of(p, hb.name_off[ri + 1] - hb.name_off[ri]) & (nbits - 1);
        return ((maybe[h >> 6] >> (h & 63)) & 1) && in_names(ri);
    };
    const size_t nrec = hb.size();
    std::vector<uint8_t> look(nrec, 0);
    auto decide = [&](size_t lo, size_t hi) {
        for (size_t ri = lo; ri < hi; ++ri) {
            const RecRef r{hb, ri};
            if (r.multi() || (int)hb.mapq[ri] < c->P.min_mapqual || r.dup() || !r.mapped() || r.refid() == -1) continue;
            const bool same_chr_mate = r.mate_mapped() && r.mrefid() == r.refid();
            if (same_chr_mate && (r.mpos() > r.pos() || (r.mpos() == r.pos() && (r.flag() & 0x80)))) continue;  // only the right-hand record of a pair
            if (!names.empty() && in_names_fast(ri)) continue;
            look[ri] = 1;
        }
    };
    if (c->pool && nrec > 100000) { const int np = 8 * (c->pool->size() + 1); c->pool->parallel_for(np, 1 << 20, [&](int k) { decide(nrec * (size_t)k / (size_t)np, nrec * ((size_t)k + 1) / (size_t)np); }); }
    else decide(0, nrec);
    // The walk itself: the cursor moves one entry per record at most and never back (:3157), so where it stands depends on every record
    // before -- but it is a small number that a walk from 0 over a few ten thousand records in front of a stretch usually reproduces (two
    // cursors fed the same records never cross, and meet for good once the one behind has caught up).  Every stretch walks from such a
    // guess into counts of its own; the stretches are then checked in order -- a guess that is not the cursor the stretch in front ended
    // with: that stretch is walked again from the real one -- and the counts added up.
    auto walk = [&](size_t lo, size_t hi, size_t cur, std::vector<int32_t>* into) -> size_t {
        for (size_t ri = lo; ri < hi; ++ri) {
            if (!look[ri]) continue;
            if (cur == bps.size()) break;
            const RecRef r{hb, ri};
            const bool same_chr_mate = r.mate_mapped() && r.mrefid() == r.refid();
            const int chr = r.refid(), start = same_chr_mate ? r.mpos() : r.pos(), end = hb.endpos[ri];
            if (chr > bps[cur].first || (chr == bps[cur].first && start > bps[cur].second + c->P.concord_dist_pos)) ++cur;
            if (!into) continue;
            for (size_t k = cur; k < bps.size(); ++k) {
                if (chr == bps[k].first && start <= bps[k].second && end > bps[k].second) ++(*into)[k];
                else if (chr < bps[k].first || (chr == bps[k].first && end <= bps[k].second)) break;
            }
        }
        return cur;
    };
    const long piece_env = std::getenv("SQUID_BWA_PIECE") ? std::atol(std::getenv("SQUID_BWA_PIECE")) : 0;  // (tests: short stretches and a short warm-up on small inputs)
    const int np = !c->pool ? 1 : piece_env > 0 ? (int)std::min<size_t>(4096, std::max<size_t>(1, nrec / (size_t)piece_env)) : (nrec > 400000 ? 4 * (c->pool->size() + 1) : 1);
    const size_t warm_len = piece_env > 0 ? (size_t)piece_env : 65536;
    if (np == 1) { walk(0, nrec, 0, &cov); return SQ_OK; }
    auto lo_of = [&](int k) { return nrec * (size_t)k / (size_t)np; };
    struct Part { size_t guess = 0, end = 0; std::vector<int32_t> cov; };
    std::vector<Part> parts((size_t)np);
    c->pool->parallel_for(np, 1 << 20, [&](int k) {
        Part& P = parts[(size_t)k];
        const size_t lo = lo_of(k), warm = lo > warm_len ? lo - warm_len : 0;
        P.guess = k == 0 ? 0 : walk(warm, lo, 0, nullptr);
        P.cov.assign(bps.size(), 0);
        P.end = walk(lo, lo_of(k + 1), P.guess, &P.cov);
    });
    int again = 0;
    for (int k = 1; k < np; ++k) {
        Part& P = parts[(size_t)k];
        const size_t real = parts[(size_t)k - 1].end;
        if (P.guess == real) continue;
        P.cov.assign(bps.size(), 0);
        P.end = walk(lo_of(k), lo_of(k + 1), real, &P.cov);
        ++again;
    }
    c->timer.add("bwa_bp_support_stretches_walked_again", 0.0, 0.0, again);
    for (const Part& P : parts) for (size_t k = 0; k < bps.size(); ++k) cov[k] += P.cov[k];
    return SQ_OK;
}

}  // namespace sq
