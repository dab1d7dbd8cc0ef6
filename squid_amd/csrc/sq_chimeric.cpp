// Chimeric-fragment assembly on the host (small input, order-sensitive).
// Counterpart of BuildChimericSBamRecord (src/ReadRec.cpp:329-413) and of the ReadRec_t predicates
// (src/ReadRec.cpp:119-141,171-228).  The three std::sort calls whose tie order leaks into the result
// (ledger B8) are issued on index arrays with the same comparators, so libstdc++'s introsort takes the same
// decisions as it does on the reference's objects.
#include <algorithm>
#include <cstdlib>
#include <functional>
#include <new>
#include <type_traits>
#include <cstring>
#include <functional>

#include "sq_internal.h"
#include <atomic>
#include <future>
#include "sq_parsort.h"

namespace sq {

bool frag_single_anchored(const Frag& f) { return f.a.empty() || f.b.empty(); }  // MultiFilter is never set (ReadRec.cpp:14)

bool frag_end_discordant(const Frag& f, bool first) {  // ReadRec.cpp:178-209
    const BlkList& R = first ? f.a : f.b;
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
    auto same = [](const BlkList& p, const BlkList& q) {
        if (p.size() != q.size()) return false;
        for (size_t i = 0; i < p.size(); ++i)
            if (p[i].refid != q[i].refid || p[i].refpos != q[i].refpos || p[i].matchref != q[i].matchref) return false;
        return true;
    };
    return (same(x.a, y.a) && same(x.b, y.b)) || (same(x.a, y.b) && same(x.b, y.a));
}

[[maybe_unused]] static bool front_smaller(const Frag& l, const Frag& r) {  // ReadRec.cpp:90-117 (not a strict weak order; kept)
    auto lt = blk_less_pos;
    if (!l.a.empty() && !r.a.empty()) return lt(l.a.front(), r.a.front());
    if (!l.b.empty() && !r.b.empty()) return lt(l.b.front(), r.b.front());
    if (!l.a.empty() && !r.b.empty()) return lt(l.a.front(), r.b.front());
    if (!l.b.empty() && !r.a.empty()) return lt(l.b.front(), r.a.front());
    return false;
}

// Scratch of build_fragments without the serial part of a std::vector: `std::vector<T> v(n)` fills n elements on the calling thread --
// for the dense config's 9.6 M records that is ~0.9 GB of zeroes and page faults on ONE thread before the parallel loop that overwrites
// them starts (measured: a third of the 0.9 s this function takes).  Raw<T> is malloc'ed and left alone: the pages are touched by the
// threads that write them first.
namespace {
template <class T> struct Raw {
    T* p; size_t n;
    explicit Raw(size_t n_) : p((T*)std::malloc(std::max<size_t>(1, n_) * sizeof(T))), n(n_) { static_assert(std::is_trivially_copyable<T>::value, "scratch of plain values"); if (!p) throw std::bad_alloc(); }
    ~Raw() { std::free(p); }
    Raw(const Raw&) = delete; Raw& operator=(const Raw&) = delete;
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
    T* begin() { return p; } T* end() { return p + n; }
    const T* begin() const { return p; } const T* end() const { return p + n; }
    size_t size() const { return n; }
};
}  // namespace
int build_fragments(sq_ctx* c, const sq_aln_batch* b) {
    if (!b->name_off || !b->name_blob) return fail(c, SQ_E_ARG, "chimeric batch needs names");
    HostPool* pool = c->pool.get();
    auto par = [&](int64_t n, const std::function<void(int64_t, int64_t)>& f) {  // [lo, hi) pieces on the context's host threads
        const int pieces = (int)std::min<int64_t>(std::max<int64_t>(1, n / 4096), pool ? 4 * (pool->size() + 1) : 1);
        if (pieces <= 1 || !pool) { f(0, n); return; }
        pool->parallel_for(pieces, 15, [&](int k) { f(n * k / pieces, n * (k + 1) / pieces); });  // (allocation-heavy loops: measured on 9.6 M records, 'merge runs' 277 ms with 16 threads, 410 ms with 64 -- glibc's arenas)
    };
    static const bool prof = std::getenv("SQUID_CHIM_PROF") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!prof) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "build_fragments: %-22s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    // one single-record fragment per usable record (mapped, not duplicate: ReadRec.cpp:344)
    // (the indices i in [0, n) with pred(i), ascending: counted and written piece by piece on the host threads)
    auto select = [&](int64_t n, const std::function<bool(int64_t)>& pred, std::vector<int64_t>& out, size_t extra = 0) {
        const int pieces = (int)std::min<int64_t>(std::max<int64_t>(1, n / 65536), pool ? 4 * (pool->size() + 1) : 1);
        std::vector<size_t> at((size_t)pieces + 1, 0);
        auto count = [&](int k) { size_t cnt = 0; for (int64_t i = n * k / pieces; i < n * (k + 1) / pieces; ++i) cnt += pred(i) ? 1 : 0; at[(size_t)k + 1] = cnt; };
        if (pieces > 1) pool->parallel_for(pieces, 15, count); else count(0);
        for (int k = 0; k < pieces; ++k) at[(size_t)k + 1] += at[(size_t)k];
        out.reserve(at.back() + extra);
        out.resize(at.back());
        auto fill = [&](int k) { size_t o = at[(size_t)k]; for (int64_t i = n * k / pieces; i < n * (k + 1) / pieces; ++i) if (pred(i)) out[o++] = i; };
        if (pieces > 1) pool->parallel_for(pieces, 15, fill); else fill(0);
    };
    std::vector<int64_t> usable;
    select(b->n_rec, [&](int64_t i) { return !(b->flag[i] & 0x4) && !(b->flag[i] & 0x400); }, usable);
    c->n_chim_records = b->n_rec;
    if (usable.empty()) return fail(c, SQ_E_EMPTYCHIM, "chimeric input has no mapped, non-duplicate record");
    std::vector<uint16_t> sample;  // ReadLen = median of max(FirstTotalLen, SecondTotalLen) of the first five (ReadRec.cpp:336,347-348)
    for (size_t k = 0; k < usable.size() && k < 5; ++k) sample.push_back((uint16_t)b->totlen[usable[k]]);
    const size_t nr = usable.size();
    // sort by QNAME (ReadRec.cpp:354).  What is sorted is (first 16 bytes of the name as two big-endian words, index): most comparisons
    // are decided inside the 24-byte elements, without touching the names; the comparison results, and with them the permutation
    // libstdc++'s introsort produces (ledger B8), are those of `name < name` on the reference's std::string objects (bytes compared as
    // unsigned values, then the lengths).  The names stay where the batch has them: (offset, length) with a trailing /1 or /2 cut off
    // (ReadRec.cpp:62-66); no per-record fragment object is built -- the merged fragments are put together from the sorted order.
    struct NameKey { uint64_t hi, lo; int32_t idx; uint32_t len; };  // (names of up to 16 bytes are compared without leaving the element)
    Raw<NameKey> nk(nr);
    Raw<uint32_t> nlen(nr);
    auto name_ptr = [&](size_t k) { return b->name_blob + b->name_off[usable[k]]; };
    par((int64_t)nr, [&](int64_t lo, int64_t hi) {
        for (int64_t k = lo; k < hi; ++k) {
            const int64_t i = usable[(size_t)k];
            const char* nm = b->name_blob + b->name_off[i];
            size_t L = b->name_off[i + 1] - b->name_off[i];
            if (L >= 2 && nm[L - 2] == '/' && (nm[L - 1] == '1' || nm[L - 1] == '2')) L -= 2;
            nlen[(size_t)k] = (uint32_t)L;
            unsigned char buf[16] = {0};
            std::memcpy(buf, nm, std::min<size_t>(16, L));
            uint64_t h = 0, l = 0;
            for (int q = 0; q < 8; ++q) { h = (h << 8) | buf[q]; l = (l << 8) | buf[8 + q]; }
            nk[(size_t)k] = NameKey{h, l, (int32_t)k, (uint32_t)L};
        }
    });
    auto name_cmp = [&](int32_t x, int32_t y) {  // <0, 0, >0 like std::string::compare
        const size_t lx = nlen[(size_t)x], ly = nlen[(size_t)y];
        const int r = std::memcmp(name_ptr((size_t)x), name_ptr((size_t)y), std::min(lx, ly));
        return r ? r : (lx < ly ? -1 : (lx > ly ? 1 : 0));
    };
    lap("name keys");
    // (std_sort_parallel: libstdc++'s introsort with its independent sub-ranges on several threads -- same comparisons, same result)
    const int sort_threads = pool ? std::min(pool->size() + 1, 32) : 1;
    std_sort_parallel(nk.begin(), nk.end(), [&](const NameKey& x, const NameKey& y) {
        if (x.hi != y.hi) return x.hi < y.hi;
        if (x.lo != y.lo) return x.lo < y.lo;
        // equal first 16 bytes (zero padded; a name holds no NUL): two names of at most 16 bytes are then the same name
        if (x.len <= 16 && y.len <= 16) return false;
        return name_cmp(x.idx, y.idx) < 0;
    }, sort_threads, true);  // (`name < name` is a strict weak order: the final insertion pass is split as well; FrontSmallerThan below is not one)
    lap("name sort");
    // what the merge below reads of a record and of its blocks, packed: the runs visit the records in NAME order -- random order in the
    // batch -- and nine arrays cost nine cache misses per record where two packed ones cost two
    struct PRec { int32_t refid; uint32_t b0, nb; uint16_t flag, tot; uint8_t low; };
    struct PBlk { int32_t refpos, matchref; uint16_t readpos, matchread; };
    Raw<PRec> prec(nr);
    const size_t nblk_all = b->n_rec ? (size_t)b->blk_off[b->n_rec] : 0;
    Raw<PBlk> pblk(nblk_all);
    par((int64_t)nr, [&](int64_t lo, int64_t hi) {
        for (int64_t k = lo; k < hi; ++k) {
            const int64_t i = usable[(size_t)k];
            prec[(size_t)k] = PRec{b->refid[i], b->blk_off[i], b->blk_off[i + 1] - b->blk_off[i], b->flag[i], b->totlen[i], (uint8_t)((b->aux[i] & SQ_AUX_LOWPHRED) ? 1 : 0)};
        }
    });
    par((int64_t)nblk_all, [&](int64_t lo, int64_t hi) { for (int64_t q = lo; q < hi; ++q) pblk[(size_t)q] = PBlk{b->b_refpos[q], b->b_matchref[q], b->b_readpos[q], b->b_matchread[q]}; });
    // merge equal names (ReadRec.cpp:356-373): runs of equal names in the sorted order; the runs are independent of each other
    std::vector<int64_t> run_start;
    {
        Raw<uint8_t> starts(nr);
        par((int64_t)nr, [&](int64_t lo, int64_t hi) {
            for (int64_t k = lo; k < hi; ++k) {
                const NameKey& x = nk[(size_t)k];
                const NameKey& w = nk[(size_t)k == 0 ? 0 : (size_t)k - 1];
                starts[(size_t)k] = k == 0 || x.hi != w.hi || x.lo != w.lo || ((x.len > 16 || w.len > 16) && name_cmp(x.idx, w.idx) != 0);
            }
        });
        select((int64_t)nr, [&](int64_t k) { return starts[(size_t)k] != 0; }, run_start, 1);
    }
    run_start.push_back((int64_t)nr);
    const size_t nm = run_start.size() - 1;
    // (the merged fragments: raw storage, every element constructed by the thread that fills it and destroyed side by side at the end)
    struct FragStore {
        Frag* p; size_t n; const std::function<void(int64_t, const std::function<void(int64_t, int64_t)>&)>* par_; std::vector<uint8_t> made;
        ~FragStore() { if (par_) (*par_)((int64_t)n, [&](int64_t lo, int64_t hi) { for (int64_t j = lo; j < hi; ++j) if (made[(size_t)j]) p[j].~Frag(); }); std::free(p); }
    };
    const std::function<void(int64_t, const std::function<void(int64_t, int64_t)>&)> par_fn = par;
    FragStore store{(Frag*)std::malloc(std::max<size_t>(1, nm) * sizeof(Frag)), nm, &par_fn, std::vector<uint8_t>(nm, 0)};
    if (!store.p) throw std::bad_alloc();
    Frag* merged = store.p;
    auto by_readpos = blk_less_readpos;
    par((int64_t)nm, [&](int64_t lo, int64_t hi) {
        for (int64_t j = lo; j < hi; ++j) {
            // the reference merges the records of a run into the first one, in the sorted order: blocks are appended mate by mate, and a
            // mate's total length / low-quality flag come from the first record (whatever they are) unless its length is 0 and a later
            // record brings one
            if ((size_t)j + 8 < nm) __builtin_prefetch(&prec[(size_t)nk[(size_t)run_start[(size_t)j + 8]].idx]);  // (the records of the runs ahead: random places)
            if ((size_t)j + 3 < nm) __builtin_prefetch(&pblk[prec[(size_t)nk[(size_t)run_start[(size_t)j + 3]].idx].b0]);
            Frag& m = *new (&merged[(size_t)j]) Frag();
            store.made[(size_t)j] = 1;
            const size_t k0 = (size_t)run_start[(size_t)j], k1 = (size_t)run_start[(size_t)j + 1];
            m.name.assign(name_ptr((size_t)nk[k0].idx), nlen[(size_t)nk[k0].idx]);
            size_t na = 0, nb = 0;
            for (size_t k = k0; k < k1; ++k) { const PRec& r = prec[(size_t)nk[k].idx]; ((r.flag & 0x40) ? na : nb) += r.nb; }
            m.a.reserve(na); m.b.reserve(nb);
            for (size_t k = k0; k < k1; ++k) {
                const PRec& r = prec[(size_t)nk[k].idx];
                const int flag = r.flag;
                const bool first = flag & 0x40, rev = flag & 0x10, low = r.low != 0;
                const int tot = r.tot;
                BlkList& dst = first ? m.a : m.b;
                for (uint32_t q = r.b0; q < r.b0 + r.nb; ++q)
                    dst.push_back(Blk{r.refid, pblk[q].refpos, (int32_t)pblk[q].readpos, pblk[q].matchref, (int32_t)pblk[q].matchread, rev, first});
                if (k == k0) { if (first) { m.atot = tot; m.alow = low; } else { m.btot = tot; m.blow = low; } }
                else if (first) { if (m.atot == 0 && tot != 0) { m.atot = tot; m.alow = low; } }
                else if (m.btot == 0 && tot != 0) { m.btot = tot; m.blow = low; }
            }
            std::sort(m.a.begin(), m.a.end(), by_readpos);  // SortbyReadPos (ReadRec.cpp:143-146)
            std::sort(m.b.begin(), m.b.end(), by_readpos);
        }
    });
    lap("merge runs");
    std::sort(sample.begin(), sample.end());
    c->read_len = sample[sample.size() / 2];  // ReadRec.cpp:378-379
    // sort by front position (ReadRec.cpp:382; FrontSmallerThan is not a strict weak order, kept): again on small elements that
    // carry everything the comparator looks at
    struct FrontKey { Blk a, b; bool has_a, has_b; int32_t idx; };
    Raw<FrontKey> fk(nm);
    std::atomic<long> without_a{0};
    par((int64_t)nm, [&](int64_t lo, int64_t hi) {
        long none = 0;
        for (int64_t j = lo; j < hi; ++j) {
            const Frag& m = merged[(size_t)j];
            FrontKey k{};
            k.has_a = !m.a.empty(); k.has_b = !m.b.empty(); k.idx = (int32_t)j;
            if (k.has_a) k.a = m.a.front();
            if (k.has_b) k.b = m.b.front();
            none += !k.has_a;
            fk[(size_t)j] = k;
        }
        without_a += none;
    });
    // FrontSmallerThan (ReadRec.cpp:90-117) is not a strict weak order in general -- which pair of blocks it compares depends on which mates
    // BOTH fragments have -- but when every fragment has a first-mate block it IS the comparison of those blocks by (RefID, RefPos), a
    // strict weak order: the threaded introsort may then split its final insertion pass (sq_parsort.h)
    const bool front_is_strict = without_a.load() == 0;
    std_sort_parallel(fk.begin(), fk.end(), [](const FrontKey& l, const FrontKey& r) {  // == front_smaller(merged[l.idx], merged[r.idx])
        if (l.has_a && r.has_a) return blk_less_pos(l.a, r.a);
        if (l.has_b && r.has_b) return blk_less_pos(l.b, r.b);
        if (l.has_a && r.has_b) return blk_less_pos(l.a, r.b);
        if (l.has_b && r.has_a) return blk_less_pos(l.b, r.a);
        return false;
    }, sort_threads, front_is_strict);
    lap(front_is_strict ? "front sort (strict)" : "front sort");
    // PCR duplicate removal (ReadRec.cpp:387-409)
    // The loop of the reference keeps a fragment unless an equal one was kept among the fragments right in front of it that share its
    // first first-in-pair block position; the backward search stops at the first kept fragment without such a block or with another
    // position.  So the decisions fall into groups -- maximal runs, in this order, of fragments with a first-in-pair block at one
    // position -- whose first member is always kept (what was kept before it lies in another group, or has no such block) and whose
    // other members are compared with kept members of the same group only.  Groups are decided side by side.
    par((int64_t)c->frags.size(), [&](int64_t lo, int64_t hi) {  // (the fragments of an earlier ingest: freed side by side, not one by one)
        for (int64_t k = lo; k < hi; ++k) { Frag& f = c->frags[(size_t)k]; f.a.release(); f.b.release(); std::string().swap(f.name); }
    });
    c->frags.clear();
    std::vector<Frag>& out = c->frags;
    std::vector<uint8_t> kept(nm, 0);     // by merged index
    std::vector<uint8_t> keep_at(nm, 0);  // by position in the sorted order
    std::vector<uint8_t> named(nm, 0);    // by merged index: kept, and the name is not empty
    std::vector<int32_t> where(nm, -1);   // by merged index: position in the output
    {
        Raw<uint8_t> gstart(nm);
        par((int64_t)nm, [&](int64_t lo, int64_t hi) {
            for (int64_t p = lo; p < hi; ++p) {
                const FrontKey& k = fk[(size_t)p];
                bool st = p == 0 || !k.has_a;
                if (!st) { const FrontKey& q = fk[(size_t)p - 1]; st = !q.has_a || q.a.refid != k.a.refid || q.a.refpos != k.a.refpos; }
                gstart[(size_t)p] = st;
            }
        });
        std::vector<int64_t> groups;
        select((int64_t)nm, [&](int64_t p) { return gstart[(size_t)p] != 0; }, groups, 1);
        groups.push_back((int64_t)nm);
        par((int64_t)groups.size() - 1, [&](int64_t lo, int64_t hi) {
            std::vector<size_t> mine;  // kept members of the group so far
            for (int64_t g = lo; g < hi; ++g) {
                mine.clear();
                for (size_t p = (size_t)groups[(size_t)g]; p < (size_t)groups[(size_t)g + 1]; ++p) {
                    const Frag& f = merged[(size_t)fk[p].idx];
                    bool keep = true;
                    for (size_t q = mine.size(); q-- > 0 && keep;) if (frag_equal(f, merged[(size_t)fk[mine[q]].idx])) keep = false;
                    if (keep) { mine.push_back(p); keep_at[p] = 1; }
                }
            }
        });
        // output position of every kept fragment: a prefix sum over the sorted order, piece by piece
        const int pieces = (int)std::min<int64_t>(std::max<int64_t>(1, (int64_t)nm / 65536), pool ? 4 * (pool->size() + 1) : 1);
        std::vector<size_t> at((size_t)pieces + 1, 0);
        auto piece_of = [&](int k) { return std::make_pair(nm * (size_t)k / (size_t)pieces, nm * ((size_t)k + 1) / (size_t)pieces); };
        auto count = [&](int k) { size_t cnt = 0; for (size_t p = piece_of(k).first; p < piece_of(k).second; ++p) cnt += keep_at[p]; at[(size_t)k + 1] = cnt; };
        if (pieces > 1) pool->parallel_for(pieces, 15, count); else count(0);
        for (int k = 0; k < pieces; ++k) at[(size_t)k + 1] += at[(size_t)k];
        const size_t total = at.back();
        // (the room for the output -- one default-constructed Frag per kept fragment, a third of a gigabyte on the dense config and one
        // thread's work -- is made while the other threads settle who goes where)
        std::future<void> room;
        if (pool && total > 100000) room = std::async(std::launch::async, [&out, total]() { out.resize(total); }); else out.resize(total);
        auto place = [&](int k) {
            size_t o = at[(size_t)k];
            for (size_t p = piece_of(k).first; p < piece_of(k).second; ++p) if (keep_at[p]) {
                const size_t j = (size_t)fk[p].idx;
                kept[j] = 1; where[j] = (int32_t)o++; named[j] = !merged[j].name.empty();
            }
        };
        if (pieces > 1) pool->parallel_for(pieces, 15, place); else place(0);
        // the names that leave the reference's ChimName with their fragment (a name belongs to one merged fragment), in name order
        c->chim_dead.clear();
        {
            std::vector<int64_t> dead;
            select((int64_t)nm, [&](int64_t j) { return !kept[(size_t)j] && !merged[(size_t)j].name.empty(); }, dead);
            c->chim_dead.resize(dead.size());
            par((int64_t)dead.size(), [&](int64_t lo, int64_t hi) { for (int64_t q = lo; q < hi; ++q) c->chim_dead[(size_t)q] = merged[(size_t)dead[(size_t)q]].name; });
        }
        if (room.valid()) room.get();
        par((int64_t)nm, [&](int64_t lo, int64_t hi) {
            for (int64_t j = lo; j < hi; ++j) if (kept[(size_t)j]) out[(size_t)where[(size_t)j]] = std::move(merged[(size_t)j]);
        });
    }
    lap("duplicate removal");
    // ChimName: Chimrecord.size() empty strings + every Qname, sorted unique (SegmentGraph.cpp:196-201, ledger B9).  `merged` is in name
    // order and holds every name once, so the kept names in that order are already the sorted unique list (the names were moved into
    // `out`: read them back through the map from merged index to output position)
    c->chim_names.clear();
    if (!out.empty()) {
        std::vector<int64_t> with_name;  // merged indices, ascending = name order
        select((int64_t)nm, [&](int64_t j) { return named[(size_t)j] != 0; }, with_name);
        c->chim_names.resize(with_name.size() + 1);  // slot 0: "" (sorts in front of everything; a fragment with an empty name falls together with it)
        par((int64_t)with_name.size(), [&](int64_t lo, int64_t hi) {
            for (int64_t q = lo; q < hi; ++q) c->chim_names[(size_t)q + 1] = out[(size_t)where[(size_t)with_name[(size_t)q]]].name;
        });
    }
    lap("names");
    c->chim_set.clear();  // (only the host parser looks names up in a set: built there)
    c->counts.n_chimeric_records = b->n_rec;
    c->counts.n_chim_fragments = (int64_t)out.size();
    c->counts.read_len = c->read_len;
    return SQ_OK;
}

}  // namespace sq
