// Chimeric-fragment assembly on the host (small input, order-sensitive).
// Counterpart of BuildChimericSBamRecord (src/ReadRec.cpp:329-413) and of the ReadRec_t predicates
// (src/ReadRec.cpp:119-141,171-228).  The three std::sort calls whose tie order leaks into the result
// (ledger B8) are issued on index arrays with the same comparators, so libstdc++'s introsort takes the same
// decisions as it does on the reference's objects.
#include <algorithm>
#include <cstring>

#include "sq_internal.h"

namespace sq {

bool frag_single_anchored(const Frag& f) { return f.a.empty() || f.b.empty(); }  // MultiFilter is never set (ReadRec.cpp:14)

bool frag_end_discordant(const Frag& f, bool first) {  // ReadRec.cpp:178-209
    const std::vector<Blk>& R = first ? f.a : f.b;
    if (R.size() <= 1) return false;
    for (size_t i = 0; i + 1 < R.size(); ++i) {
        const Blk &x = R[i], &y = R[i + 1];
        if (x.refid != y.refid || x.rev != y.rev) return true;
        bool refup = x.refpos < y.refpos, readup = x.readpos < y.readpos;
        if (!x.rev && refup != readup) return true;
        if (x.rev && refup == readup) return true;
    }
    return false;
}

bool frag_pair_discordant(const Frag& f, bool needcheck) {  // ReadRec.cpp:211-228
    if (f.a.empty() || f.b.empty()) return false;
    if (needcheck && (frag_end_discordant(f, true) || frag_end_discordant(f, false))) return true;
    const Blk &af = f.a.front(), &ab = f.a.back(), &bf = f.b.front(), &bb = f.b.back();
    if (af.refid != bb.refid || af.rev == bb.rev) return true;
    if (!af.rev && af.refpos - af.readpos > bb.refpos - (f.btot - bb.readpos - bb.matchread)) return true;
    if (!bf.rev && bf.refpos - bf.readpos > ab.refpos - (f.atot - ab.readpos - ab.matchread)) return true;
    return false;
}

bool frag_equal(const Frag& x, const Frag& y) {  // ReadRec.cpp:119-141
    auto same = [](const std::vector<Blk>& p, const std::vector<Blk>& q) {
        if (p.size() != q.size()) return false;
        for (size_t i = 0; i < p.size(); ++i)
            if (p[i].refid != q[i].refid || p[i].refpos != q[i].refpos || p[i].matchref != q[i].matchref) return false;
        return true;
    };
    return (same(x.a, y.a) && same(x.b, y.b)) || (same(x.a, y.b) && same(x.b, y.a));
}

static bool front_smaller(const Frag& l, const Frag& r) {  // ReadRec.cpp:90-117 (not a strict weak order; kept)
    auto lt = blk_less_pos;
    if (!l.a.empty() && !r.a.empty()) return lt(l.a.front(), r.a.front());
    if (!l.b.empty() && !r.b.empty()) return lt(l.b.front(), r.b.front());
    if (!l.a.empty() && !r.b.empty()) return lt(l.a.front(), r.b.front());
    if (!l.b.empty() && !r.a.empty()) return lt(l.b.front(), r.a.front());
    return false;
}

int build_fragments(sq_ctx* c, const sq_aln_batch* b) {
    if (!b->name_off || !b->name_blob) return fail(c, SQ_E_ARG, "chimeric batch needs names");
    // one single-record fragment per usable record (mapped, not duplicate: ReadRec.cpp:344)
    std::vector<Frag> recs;
    std::vector<uint16_t> sample;
    for (int64_t i = 0; i < b->n_rec; ++i) {
        int flag = b->flag[i];
        if ((flag & 0x4) || (flag & 0x400)) continue;
        Frag f;
        f.name.assign(b->name_blob + b->name_off[i], b->name_blob + b->name_off[i + 1]);
        size_t L = f.name.size();
        if (L >= 2 && f.name[L - 2] == '/' && (f.name[L - 1] == '1' || f.name[L - 1] == '2')) f.name.resize(L - 2);
        bool first = flag & 0x40, rev = flag & 0x10;
        std::vector<Blk>& dst = first ? f.a : f.b;
        for (uint32_t k = b->blk_off[i]; k < b->blk_off[i + 1]; ++k)
            dst.push_back(Blk{b->refid[i], b->b_refpos[k], (int32_t)b->b_readpos[k], b->b_matchref[k], (int32_t)b->b_matchread[k], rev, first});
        bool low = b->aux[i] & SQ_AUX_LOWPHRED;
        if (first) { f.atot = b->totlen[i]; f.alow = low; }
        else { f.btot = b->totlen[i]; f.blow = low; }
        if (sample.size() < 5) sample.push_back((uint16_t)std::max(f.atot, f.btot));
        recs.push_back(std::move(f));
    }
    c->n_chim_records = b->n_rec;
    if (sample.empty()) return fail(c, SQ_E_EMPTYCHIM, "chimeric input has no mapped, non-duplicate record");
    // sort by QNAME (ReadRec.cpp:354): same comparator on an index array => same permutation
    std::vector<int> idx(recs.size());
    for (size_t i = 0; i < idx.size(); ++i) idx[i] = (int)i;
    // (the first 16 bytes of every name as two big-endian words next to the index: most comparisons are decided without touching the
    // strings; the comparison results, and with them the permutation introsort produces, are those of `name < name`)
    struct NameKey { uint64_t hi, lo; };
    std::vector<NameKey> nk(recs.size());
    for (size_t i = 0; i < recs.size(); ++i) {
        unsigned char buf[16] = {0};
        std::memcpy(buf, recs[i].name.data(), std::min<size_t>(16, recs[i].name.size()));
        uint64_t h = 0, l = 0;
        for (int k = 0; k < 8; ++k) { h = (h << 8) | buf[k]; l = (l << 8) | buf[8 + k]; }
        nk[i] = NameKey{h, l};
    }
    std::sort(idx.begin(), idx.end(), [&](int x, int y) {
        const NameKey &a = nk[(size_t)x], &b = nk[(size_t)y];
        if (a.hi != b.hi) return a.hi < b.hi;
        if (a.lo != b.lo) return a.lo < b.lo;
        return recs[x].name < recs[y].name;
    });
    // merge equal names (ReadRec.cpp:356-373)
    std::vector<Frag> merged;
    for (int id : idx) {
        Frag& r = recs[id];
        if (merged.empty() || r.name != merged.back().name) merged.push_back(std::move(r));
        else {
            Frag& m = merged.back();
            if (m.atot == 0 && r.atot != 0) { m.atot = r.atot; m.alow = r.alow; }
            if (m.btot == 0 && r.btot != 0) { m.btot = r.btot; m.blow = r.blow; }
            m.a.insert(m.a.end(), r.a.begin(), r.a.end());
            m.b.insert(m.b.end(), r.b.begin(), r.b.end());
        }
    }
    auto by_readpos = blk_less_readpos;
    for (Frag& m : merged) {  // SortbyReadPos (ReadRec.cpp:143-146)
        std::sort(m.a.begin(), m.a.end(), by_readpos);
        std::sort(m.b.begin(), m.b.end(), by_readpos);
    }
    std::sort(sample.begin(), sample.end());
    c->read_len = sample[sample.size() / 2];  // ReadRec.cpp:378-379
    // sort by front position (ReadRec.cpp:382)
    idx.resize(merged.size());
    for (size_t i = 0; i < idx.size(); ++i) idx[i] = (int)i;
    std::sort(idx.begin(), idx.end(), [&](int x, int y) { return front_smaller(merged[x], merged[y]); });
    // PCR duplicate removal (ReadRec.cpp:387-409)
    c->frags.clear();
    std::vector<Frag>& out = c->frags;
    for (int id : idx) {
        Frag& f = merged[id];
        bool keep;
        if (out.empty()) keep = true;
        else if (f.a.empty() || out.back().a.empty()) keep = true;
        else if (f.a.front().refid != out.back().a.front().refid || f.a.front().refpos != out.back().a.front().refpos) keep = true;
        else {
            keep = true;
            for (size_t k = out.size(); k-- > 0;) {
                const Frag& g = out[k];
                if (g.a.empty() || f.a.front().refid != g.a.front().refid || f.a.front().refpos != g.a.front().refpos) break;
                if (frag_equal(f, g)) { keep = false; break; }
            }
        }
        if (keep) out.push_back(std::move(f));
    }
    // ChimName: Chimrecord.size() empty strings + every Qname, sorted unique (SegmentGraph.cpp:196-201, ledger B9)
    c->chim_names.clear();
    if (!out.empty()) c->chim_names.push_back("");
    for (const Frag& f : out) c->chim_names.push_back(f.name);
    std::sort(c->chim_names.begin(), c->chim_names.end());
    c->chim_names.erase(std::unique(c->chim_names.begin(), c->chim_names.end()), c->chim_names.end());
    c->chim_set.clear();
    c->chim_set.insert(c->chim_names.begin(), c->chim_names.end());
    c->counts.n_chimeric_records = b->n_rec;
    c->counts.n_chim_fragments = (int64_t)out.size();
    c->counts.read_len = c->read_len;
    return SQ_OK;
}

}  // namespace sq
