// Internal declarations of libsquid_hip.so (host side).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <thread>
#include <mutex>
#include <deque>
#include <exception>
#include <new>
#include <condition_variable>
#include <atomic>
#include <cstdint>
#include <functional>
#include <future>
#include <type_traits>
#include <cstring>
#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/squid_hip.h"

namespace sq {

// ---------------------------------------------------------------------------------------------- host model
// one aligned block of a chimeric fragment (host side is AoS: N_x is ~1 % of N_c)
struct Blk {
    int32_t refid, refpos, readpos, matchref, matchread;
    bool rev;
    bool first;  // IsFirstRead of the record the block came from (only used by the B11 `Same` quirk)
};
// SingleBamRec_t's comparators (src/SingleBamRec.h:39-58) on the host block type; every sort / comparison of blocks in
// the library goes through these, and tests/test_ref_pin.py checks them against the reference header compiled in place
inline bool blk_less_pos(const Blk& x, const Blk& y) { return x.refid != y.refid ? x.refid < y.refid : x.refpos < y.refpos; }       // operator<
inline bool blk_greater_pos(const Blk& x, const Blk& y) { return x.refid != y.refid ? x.refid > y.refid : x.refpos > y.refpos; }    // operator>
inline bool blk_eq_pos(const Blk& x, const Blk& y) { return x.refid == y.refid && x.refpos == y.refpos; }                           // operator==
inline bool blk_less_readpos(const Blk& x, const Blk& y) { return x.readpos < y.readpos; }                                          // CompReadPos
inline bool blk_same(const Blk& x, const Blk& y) {                                                                                  // Same
    return x.refid == y.refid && x.refpos == y.refpos && x.readpos == y.readpos && x.matchread == y.matchread && x.matchref == y.matchref && x.rev == y.rev && x.first == y.first;
}
// The block list of one mate of a fragment: nearly always one or two blocks (a split read has two), so two live inside the object and
// only a longer list goes to the heap.  Millions of fragments with two std::vectors each were millions of small heap blocks -- made by
// one thread, freed by another (glibc then takes the maker's arena lock), chased through a pointer by every loop over the fragments; on
// the dense config the pairing, the cluster walk, the chimeric edges and the frees all paid for that layout.
// The subset of std::vector's interface the library uses, for trivially copyable T; iterators are pointers.
template <class T, int N>
class SmallVec {
    static_assert(std::is_trivially_copyable<T>::value, "plain values");
    T* p_;
    uint32_t n_, cap_;
    alignas(T) unsigned char in_[N * sizeof(T)];
    T* inl() { return reinterpret_cast<T*>(in_); }
    void grow(size_t c) {
        T* q = (T*)std::malloc(c * sizeof(T));
        if (!q) throw std::bad_alloc();  // (the object is unchanged)
        std::memcpy((void*)q, (const void*)p_, (size_t)n_ * sizeof(T));
        if (p_ != inl()) std::free(p_);
        p_ = q; cap_ = (uint32_t)c;
    }
public:
    typedef T value_type; typedef T* iterator; typedef const T* const_iterator;
    SmallVec() : p_(inl()), n_(0), cap_(N) {}
    SmallVec(const SmallVec& o) : SmallVec() { insert(end(), o.begin(), o.end()); }
    SmallVec(SmallVec&& o) noexcept : SmallVec() { take(o); }
    SmallVec& operator=(const SmallVec& o) { if (this != &o) { n_ = 0; insert(end(), o.begin(), o.end()); } return *this; }
    SmallVec& operator=(SmallVec&& o) noexcept { if (this != &o) { release(); take(o); } return *this; }
    ~SmallVec() { if (p_ != inl()) std::free(p_); }
    void release() { if (p_ != inl()) std::free(p_); p_ = inl(); n_ = 0; cap_ = N; }  // back to the empty state, heap block returned
    void take(SmallVec& o) {  // (*this is empty and inline)
        if (o.p_ != o.inl()) { p_ = o.p_; n_ = o.n_; cap_ = o.cap_; o.p_ = o.inl(); o.n_ = 0; o.cap_ = N; }
        else { std::memcpy((void*)inl(), (const void*)o.inl(), (size_t)o.n_ * sizeof(T)); n_ = o.n_; o.n_ = 0; }
    }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    T* begin() { return p_; } T* end() { return p_ + n_; }
    const T* begin() const { return p_; } const T* end() const { return p_ + n_; }
    T& front() { return p_[0]; } const T& front() const { return p_[0]; }
    T& back() { return p_[n_ - 1]; } const T& back() const { return p_[n_ - 1]; }
    T& operator[](size_t i) { return p_[i]; } const T& operator[](size_t i) const { return p_[i]; }
    void reserve(size_t c) { if (c > cap_) grow(c); }
    void clear() { n_ = 0; }
    void push_back(const T& v) { if (n_ == cap_) { const T keep = v; grow((size_t)cap_ * 2 > 4 ? (size_t)cap_ * 2 : 4); p_[n_++] = keep; } else p_[n_++] = v; }
    void resize(size_t n) { reserve(n); for (size_t i = n_; i < n; ++i) p_[i] = T(); n_ = (uint32_t)n; }
    template <class It> void insert(const T* at, It first, It last) {  // (only at the end: what the library does)
        (void)at;
        const size_t k = (size_t)(last - first);
        if (n_ + k > cap_) grow(std::max<size_t>(n_ + k, (size_t)cap_ * 2));
        for (; first != last; ++first) p_[n_++] = *first;
    }
};
typedef SmallVec<Blk, 2> BlkList;
struct Frag {  // merged chimeric fragment = ReadRec_t after BuildChimericSBamRecord
    std::string name;
    BlkList a, b;  // first-in-pair blocks, second-in-pair blocks (sorted by read offset)
    int atot = 0, btot = 0;
    bool alow = false, blow = false;
};
struct Node {
    int32_t chr, pos, len, support;
    double depth;
    // bounds of `depth` over every tie order the reference's unstable ReadsOther sort could produce (== depth when the
    // value is exact); FilterEdges checks that its coverage-ratio decisions do not depend on where inside they fall
    double depth_lo = 0, depth_hi = 0;
};
struct Edge {
    int32_t a, b;  // a <= b
    uint8_t ha, hb;
    int32_t w, gw;
};
inline bool edge_key_less(const Edge& x, const Edge& y) {
    if (x.a != y.a) return x.a < y.a;
    if (x.b != y.b) return x.b < y.b;
    if (x.ha != y.ha) return x.ha < y.ha;
    return x.hb < y.hb;
}
inline bool edge_key_eq(const Edge& x, const Edge& y) { return x.a == y.a && x.b == y.b && x.ha == y.ha && x.hb == y.hb; }
inline Edge make_edge(int i, bool hi, int j, bool hj, int w = 1) {
    Edge e;
    if (i > j) { e.a = j; e.ha = hj; e.b = i; e.hb = hi; }
    else { e.a = i; e.ha = hi; e.b = j; e.hb = hj; }
    e.w = w;
    e.gw = 0;
    return e;
}
inline uint64_t edge_pack(const Edge& e) { return ((uint64_t)(uint32_t)e.a << 32) | ((uint64_t)(uint32_t)e.b << 2) | ((uint64_t)e.ha << 1) | e.hb; }

struct GraphSnap {  // flattened copy for sq_graph_view
    std::vector<int32_t> chr, pos, len, support, label, ind1, ind2, weight, gweight;
    std::vector<double> depth;
    std::vector<uint8_t> h1, h2;
    void take(const std::vector<Node>& N, const std::vector<Edge>& E, const std::vector<int32_t>* lab);
    void view(sq_graph* g) const;
};

// per-record summary of the kept pass-1 stream, produced on the GPU for the host segmentation automaton
struct StreamRec {
    int32_t refid, pos;      // record.RefID / record.Position
    int32_t fb_refpos, fb_matchref;  // first aligned block in CIGAR order (ReadsMain / window element)
    uint16_t fb_readpos;
    uint8_t flags;           // SR_* bits
    uint8_t nrest;           // number of further blocks (saturated at 255)
    uint32_t rest_off;       // offset of the further blocks in the rest arrays
};
enum : uint8_t { SR_HASBLK = 1, SR_CONC = 2, SR_PART = 4, SR_REV = 8, SR_MATE = 16 /* 0x40 or 0x80 set */ };

struct Timer {
    std::vector<const char*> names;
    std::vector<double> ms, bytes, busy;  // busy: time during which at least one launch of that name was running (kernels of one name on several streams overlap; ms is the sum of their durations)
    std::vector<int64_t> launches;
    int slot(const char* name);
    void add(const char* name, double ms_, double bytes_ = 0, int64_t n = 1);
    void add_busy(const char* name, double ms_);
    void clear();
};

// Host threads of a context, started once (sq_create): single tasks that run next to the GPU work of the calling thread
// (submit) and index loops (parallel_for; the caller takes part, so a loop started from a pooled task cannot starve).
class HostPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    bool stop = false;

public:
    explicit HostPool(int n) {
        for (int i = 0; i < n; ++i) th.emplace_back([this]() {
            for (;;) {
                std::function<void()> f;
                { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return stop || !q.empty(); }); if (stop && q.empty()) return; f = std::move(q.front()); q.pop_front(); }
                f();
            }
        });
    }
    ~HostPool() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); for (auto& t : th) t.join(); }
    int size() const { return (int)th.size(); }
    template <class F> auto submit(F f) -> std::future<decltype(f())> {
        auto task = std::make_shared<std::packaged_task<decltype(f())()>>(std::move(f));
        auto fut = task->get_future();
        if (th.empty()) { (*task)(); return fut; }
        { std::lock_guard<std::mutex> lk(mu); q.emplace_back([task]() { (*task)(); }); }
        cv.notify_one();
        return fut;
    }
    // The caller takes part and returns when every index has been worked on -- NOT when every helper task has had its turn: a helper that
    // is still queued when the work is done finds nothing to do whenever it runs (it shares the loop state, never `f`).  So a loop started
    // from a pooled task cannot starve, even on a pool of one thread that is the caller itself (round 5 waited for the helpers' futures:
    // with eight ranks sharing the box's 16 CPUs a context has ONE helper thread, and the cluster table, a task of that thread, waited
    // for a helper that only that thread could have run).
    void parallel_for(int count, int max_helpers, const std::function<void(int)>& f) {
        if (count <= 0) return;
        // An exception thrown by `f` (the bodies allocate: std::bad_alloc) is caught where it is thrown -- on a helper thread it would end the
        // process, on the calling thread it would unwind past helpers that still hold `fp` and the body's by-reference captures --, the first
        // one is kept, the indices that are left are claimed without being run, and the caller rethrows it once every index is accounted for.
        struct Loop { std::atomic<int> next{0}, done{0}; std::atomic<bool> failed{false}; std::mutex m; std::condition_variable cv; std::exception_ptr first; };
        auto st = std::make_shared<Loop>();
        const std::function<void(int)>* fp = &f;
        auto body = [st, count, fp]() {
            int mine = 0;
            for (int i; (i = st->next.fetch_add(1)) < count;) {
                if (!st->failed.load(std::memory_order_relaxed)) {
                    try { (*fp)(i); }
                    catch (...) { std::lock_guard<std::mutex> lk(st->m); if (!st->first) st->first = std::current_exception(); st->failed.store(true); }
                }
                ++mine;
            }
            if (mine && st->done.fetch_add(mine) + mine == count) { std::lock_guard<std::mutex> lk(st->m); st->cv.notify_all(); }
        };
        const int nh = std::min(std::min(count - 1, max_helpers), size());
        if (nh > 0) {
            { std::lock_guard<std::mutex> lk(mu); for (int h = 0; h < nh; ++h) q.emplace_back(body); }
            cv.notify_all();
        }
        body();
        std::unique_lock<std::mutex> lk(st->m);
        st->cv.wait(lk, [&]() { return st->done.load() == count; });
        if (st->first) std::rethrow_exception(st->first);
    }
};

// CPUs this process can really use (affinity mask, cgroup CPU quota: the GPU boxes of this pool show 256 logical CPUs and allow 16 CPUs of
// time -- more runnable threads than that are not faster, they are THROTTLED for the rest of the scheduler period); host_workers: helper
// threads of a context's pool = the rank's share of those CPUs minus the calling thread, between 1 and 63
int usable_cpus();
// (the share is by the processes that run side by side on this host, which is not always the shard world: N independent samples on N GPUs are N
// contexts of world size 1 -- torchrun tells through LOCAL_WORLD_SIZE)
inline int local_process_count(int world_size) {
    static const int env = []() { const char* v = std::getenv("LOCAL_WORLD_SIZE"); const int n = v ? std::atoi(v) : 0; return n > 0 ? n : 1; }();
    return world_size > env ? world_size : env;
}
inline int host_workers(int world_size) { const int n = usable_cpus() / local_process_count(world_size) - 1; return n < 1 ? 1 : (n > 63 ? 63 : n); }
struct DeviceRecords;  // HBM-resident SoA + scratch (sq_kernels.hip)
struct HostBatch;      // decoded records on the host (below)

// Chromosome sharding: rank r holds the concordant records of a contiguous RefID range and
// everything that is small (chimeric fragments, cluster table, node and edge tables) is replicated.  The fields
// below are what a rank learns about the other shards from the exchanges (see build_graph in sq_capi.cpp).
struct Shard {
    bool on = false;
    int first_ref = 0, end_ref = 0;   // owns RefIDs [first_ref, end_ref); the last rank also owns RefID -1
    int dedup_mask = 0;               // bit0 / bit1: a pass-1 / pass-2 record precedes this shard and its lists are not empty
    bool prior_kept = false;          // an earlier rank has kept pass-1 records
    int64_t kept_before = 0, kept_total = 0;
    bool has_terminal = false;        // a later rank has kept records; its first one closes this shard's last stretch
    int32_t term_refid = 0, term_pos = 0;
    long long other_seed = INT64_MIN; // running (otherChr << 32 | otherrightmost) of the earlier ranks
    int64_t n_break_global = 0;
};
struct GraphBuild;  // locals of sq_build_graph that survive an exchange (sq_capi.cpp)
struct SegPlan;     // sq_segment.cpp
struct RcclTransport;  // sq_exchange.cpp
void exchange_release(sq_ctx* c);
struct SvBuild;     // same for sq_call_sv

}  // namespace sq

struct sq_ctx {
    sq_params P;
    std::string err;
    hipStream_t stream = nullptr;
    std::vector<int32_t> ref_len;
    // chimeric side (host)
    int read_len = 0;
    std::vector<sq::Frag> frags, frags0;        // frags0: untrimmed copy restored by sq_reset
    std::vector<std::string> chim_names;        // sorted unique, incl. "" (ledger B9)
    std::unordered_set<std::string> chim_set;
    int64_t n_chim_records = 0;
    // concordant side (device)
    sq::DeviceRecords* dev = nullptr;
    // --bwa (sq_ingest_bwa_file): every record of the one BAM file on the host, with its QNAME (sq_bwa.cpp)
    std::shared_ptr<sq::HostBatch> bwa, bwa_spare;  // bwa_spare: the storage of the last --bwa batch, kept by sq_clear_records for the next one
    std::shared_ptr<sq::HostBatch> chim_decoded;  // the chimeric records as the GPU reader decoded them (sq_ingest_files), kept for the next call's pages
    // graph state (host, small)
    std::vector<sq::Node> nodes;
    std::vector<sq::Edge> edges;
    std::vector<int32_t> label;
    sq::GraphSnap snap[6];
    bool graph_built = false, ordered = false;
    // SQUID_REPLAY_CHECK: every break candidate the segmentation replay tests (SegmentGraph.cpp:440-481) is counted a second time with the
    // reference's linear passes over the same windows and compared with the binary-search / span-index counts the replay uses
    mutable std::atomic<long long> replay_checked{0}, replay_mismatch{0};
    std::atomic<bool> chim_pairing_running{false};  // sq_ingest_files: BuildChimericSBamRecord of a large chimeric BAM is busy on the host threads (the file feeder takes fewer readers)
    bool keep_stages = true;  // sq_keep_stage_graphs: flat copies of the graph after BuildNode / BuildEdges / the filters / CompressNode for sq_graph_view (inspection; the result does not need them)
    bool capture_names = false;  // the records being parsed by K0 are the chimeric BAM's: no name-set lookups, their QNAMEs are kept on the device (dev_download_names)
    bool ablated = false;  // a timing-only switch (SQUID_P1_ABLATE / SQUID_EDGES_ABLATE) cut a kernel short: sq_build_graph refuses to return a graph
    bool depth_bounds = false;      // node depths are canonical values with [depth_lo, depth_hi] bounds
    bool depth_ambiguous = false;   // a FilterEdges decision depends on the position inside the bounds
    std::vector<int32_t> ord_off, ord_nodes;
    std::vector<int32_t> tot_off, tot_nodes;  // sq_total_order: the components stitched into whole new chromosomes
    // sv output
    std::vector<int32_t> sv_cols[9];
    std::vector<uint8_t> sv_s1, sv_s2;
    std::vector<int32_t> bp_off, bp1, bp2, bsup1, bsup2;
    sq::Timer timer;
    bool timer_keep = false;  // sq_timing_accumulate
    sq_counts counts{};
    // chromosome-sharded runs: sq_build_graph / sq_call_sv return SQ_NEED_EXCHANGE with `xbuf` filled; the caller
    // all-gathers it and hands the result back through sq_exchange_unpack before calling the same function again
    uint64_t source_size = 0, source_mtime = 0;  // identity of the concordant BAM (sq_set_source / sq_ingest_concordant_file); 0,0 = unknown
    std::string staged_path;               // sq_stage_bam: the file whose compressed bytes are resident in HBM (DeviceRecords::staged)
    size_t staged_bytes = 0;
    uint64_t staged_ino = 0, staged_mtime = 0;  // (size, inode and mtime of the staged file: a rewrite in between is refused)
    const uint8_t* ingest_dfile = nullptr; // device copy of the file being ingested (set for the duration of the call)
    size_t ingest_total_bytes = 0, ingest_seen_bytes = 0;  // file ingest in progress: inflated bytes in the file / handed to the GPU so far
    std::unique_ptr<sq::HostPool> pool;  // host threads of this context
    sq::Shard shard;
    // ExactBreakpoint (host, chimeric fragments only) runs on a second thread from the end of sq_build_graph, next to
    // sq_order; sq_call_sv collects it
    std::future<int> bp_future;
    std::string chim_err;          // (its error text; moved into `err` by chim_join)
    std::promise<int> chim_names_promise;  // set by the helper once the device holds the table of all usable chimeric QNAMEs
    std::future<int> chim_names_future;
    std::vector<std::string> chim_dead;    // QNAMEs of the fragments the PCR-duplicate removal dropped (they leave the set: dev_chim_finalize)
    std::future<int> chim_future;  // sq_ingest_files: the chimeric decode running next to the concordant ingest (chim_join)
    std::shared_ptr<std::map<uint64_t, std::vector<std::pair<int, int>>>> bp_early;
    double bp_early_ms = 0;
    // the discordant-cluster table (segment_clusters) only needs the chimeric fragments: sq_ingest_files builds it on the chimeric helper
    // thread while the concordant file is still being read, and the first sq_build_graph after the ingest takes it (sq_reset drops it: a
    // graph pass over resident records builds its own)
    std::shared_ptr<sq::SegPlan> plan_early;
    std::vector<sq::Blk> disc_early;
    double clusters_early_ms = -1;  // < 0: none
    std::shared_ptr<sq::GraphBuild> gb;
    std::shared_ptr<sq::SvBuild> svb;
    std::vector<uint8_t> xbuf;                 // this rank's contribution
    std::vector<std::vector<uint8_t>> xgot;    // what every rank contributed (by rank), set by sq_exchange_unpack
    bool x_pending = false, x_ready = false;
    // sq_exchange: the installed all-gather (RCCL or the caller's)
    sq_allgather_fn x_allgather = nullptr;
    void* x_user = nullptr;
    std::shared_ptr<sq::RcclTransport> rccl;
    int64_t x_collectives = 0, x_bytes = 0;
};

namespace sq {

int fail(sq_ctx* c, int code, const std::string& msg);
struct ErrSink { explicit ErrSink(std::string* to); ~ErrSink(); };  // while it lives, fail() on this thread writes to *to instead of c->err (sq_capi.cpp)
struct HostClock {  // wall clock of a host stage into the context's timing table
    sq_ctx* c; const char* name; std::chrono::steady_clock::time_point t0;
    HostClock(sq_ctx* c, const char* name) : c(c), name(name), t0(std::chrono::steady_clock::now()) {}
    ~HostClock() {
        const auto t1 = std::chrono::steady_clock::now();
        c->timer.add(name, std::chrono::duration<double, std::milli>(t1 - t0).count());
        static const bool trace = std::getenv("SQUID_HOST_TRACE") != nullptr;  // one line per stage: start (since the first stage of the process) and length
        if (trace) {
            static const std::chrono::steady_clock::time_point origin = t0;
            std::fprintf(stderr, "[host %10.3f ms] %-28s %8.3f ms\n", std::chrono::duration<double, std::milli>(t0 - origin).count(), name, std::chrono::duration<double, std::milli>(t1 - t0).count());
        }
    }
};
// waits for the chimeric decode started by sq_ingest_files and uploads its QNAME set; called in front of the first record parse
int chim_join(sq_ctx* c);
// waits only for the QNAME table the helper builds right after decoding the chimeric BAM (what the record parse needs)
int chim_join_names(sq_ctx* c);

// ---- sq_bam.cpp
struct HostBatch {  // owning storage behind an sq_aln_batch
    std::vector<int32_t> refid, pos, mrefid, mpos, endpos, b_refpos, b_matchref;
    std::vector<uint16_t> flag, totlen, b_readpos, b_matchread;
    std::vector<uint8_t> mapq, aux;
    std::vector<uint32_t> blk_off, name_off;
    std::vector<char> names;
    void clear();
    void append(const HostBatch& o);  // concatenates (block / name offsets are rebased)
    void append_parts(const std::vector<HostBatch>& parts, int count);  // the same for parts[0 .. count), copied side by side
    void view(sq_aln_batch* b, bool with_names) const;
    size_t size() const { return refid.size(); }
};
void drop_file_cache();
void drop_whole_file_scratch();  // host scratch of whole-file reads (sq_release_reader_buffers)
int read_bam_header(const char* path, std::vector<std::string>& names, std::vector<int32_t>& lens, std::string& err);
// streams the file; calls sink(batch) every `batch_records` records.  inchim may be null.
struct ParseOpts { int phred_type, min_phred, max_lowphred_len; bool keep_names; const std::unordered_set<std::string>* inchim; };
int parse_bam_file(const char* path, const ParseOpts& o, size_t batch_records, int n_threads, std::string& err,
                   const std::function<int(const HostBatch&)>& sink);
// raw mode: inflate + record-boundary walk on the host, hand each chunk (bytes, record offsets) to `sink`
struct BgzfRange { unsigned long long coff; uint32_t clen, isize; unsigned long long uoff; };  // one BGZF block: payload offset/length in the file, inflated size/offset
struct RefRange { int first_ref, end_ref; bool with_unplaced; };  // chromosome shard: only the blocks that can hold its records are inflated
// the GPU reader: (file, blocks, first block, end block or npos, offset of the first record, starts on a record, n_ref,
// index_more, file bytes).  index_more (may be null) appends the next blocks of the file to `blocks` and returns false at
// the end of the file: the block index is then built batch by batch, while the GPU works on the batches before
typedef std::function<bool(std::vector<BgzfRange>&)> IndexMore;
// When the compressed bytes are not in HBM yet, the GPU reader streams the file itself (FileFeeder, sq_kernels.hip: pread into
// page-locked buffers, copies that run ahead of the token pass) and walks the BGZF headers in the bytes it has just read, from
// `walk_p` (file offset of the next header, `walk_total` inflated bytes in front of it) up to the last block that starts at or
// before `stop`; on return the three say where that walk ended.
struct GpuFileSrc { const char* path; size_t walk_p, walk_total, stop; bool bad, streamed; };
typedef std::function<int(const uint8_t*, std::vector<BgzfRange>&, size_t, size_t, size_t, bool, int, const IndexMore&, size_t, GpuFileSrc*)> GpuIngest;
int scan_bam_file(const char* path, int n_threads, std::string& err, const std::function<int(const uint8_t*, size_t, const unsigned long long*, int64_t)>& sink,
                  const std::function<void(size_t)>& on_total = nullptr, const RefRange* only = nullptr,
                  const GpuIngest& gpu = nullptr, bool force_gpu = false, bool allow_bai = true, bool gpu_streams = false);

// ---- sq_chimeric.cpp
int build_fragments(sq_ctx* c, const sq_aln_batch* b);
bool frag_end_discordant(const Frag& f, bool first);
bool frag_pair_discordant(const Frag& f, bool needcheck);
bool frag_single_anchored(const Frag& f);
bool frag_equal(const Frag& x, const Frag& y);

// ---- sq_segment.cpp  (host control of K2; counting data comes from the GPU summaries)
struct SegPlan;  // sq_segment.cpp
// static part: cluster table from the chimeric fragments (host only; returns its elapsed milliseconds) ...
double segment_clusters(const sq_ctx* c, std::shared_ptr<SegPlan>& plan, std::vector<Blk>& bamdiscordant_sorted);
// ... and the stream scans that need nothing from other shards
int segment_scan(sq_ctx* c, SegPlan& plan, bool fetch, int64_t& trigger_last, long long& other_max, int32_t first_kept[2]);
// zero-coverage records, stretches to replay, host copies of those stretches; n_break = consumed prefix of the LOCAL stream
int segment_prepare(sq_ctx* c, SegPlan& plan, int64_t& n_break);
// `virtual_back`: an earlier shard has emitted a node (it lies on an earlier chromosome); `sens` collects the pending
// node starts that were compared with that node's end without a chromosome test (SegmentGraph.cpp:623)
// `seed`: the real last node emitted before this shard (it may get extended: the result then starts with it)
int segment_replay(sq_ctx* c, SegPlan& plan, std::vector<Node>& seeds, bool virtual_back, std::vector<int32_t>* sens, const Node* seed);
int tile_genome(sq_ctx* c, std::vector<Node>& seeds, std::vector<Node>& out);

// ---- sq_graph.cpp
struct Located { std::vector<int> node; };
bool edge_discordant(const sq_ctx* c, const std::vector<Node>& N, const Edge& e);
int locate_fragment(const std::vector<Node>& N, int hint, Frag& f, std::vector<int>& out);  // trims f in place
bool frag_first_block_pins(const std::vector<Node>& N, const Frag& f, int& node);  // the first block lies deep inside ONE node: LocateRead ends there from any start
int chimeric_edges(sq_ctx* c, std::vector<Edge>& raw);
bool pair_overlap(const Frag& f, const std::vector<int>& rn, int i, int j);  // the pair-edge suppression test of :1484-1502 / :1801-1819
// ---- sq_junction.cpp (utils/JunctionSequence.cpp)
int chimeric_fragments_host(sq_ctx* c, const char* path, int threads);
int junction_sequences(sq_ctx* c, const std::vector<std::string>& ref_names, const char* bedpe, const char* fasta, const char* out_prefix);
// ---- sq_bwa.cpp (`squid --bwa`)
int bwa_nodes_and_edges(sq_ctx* c, std::vector<Edge>& raw);  // BuildNode_BWA + RawEdges over the host batch: c->nodes (+ snapshot 1), c->frags, raw edges
int bwa_breakpoint_support(sq_ctx* c, const std::vector<std::pair<int, int>>& bps, std::vector<int32_t>& cov);
void reduce_edges(std::vector<Edge>& raw, std::vector<Edge>& out, int threads = 1);
void filter_by_weight(sq_ctx* c);
void filter_by_interleaving(sq_ctx* c, std::vector<uint8_t>& keep);
void filter_edges(sq_ctx* c, const std::vector<uint8_t>& keep);
int compress_nodes(sq_ctx* c);
int further_compress(sq_ctx* c);
void multiply_discordant(sq_ctx* c, bool undo);
typedef std::map<uint64_t, std::vector<std::pair<int, int>>> BPMap;
int exact_breakpoints(sq_ctx* c, BPMap& bp);

// ---- sq_order.cpp
int order_components(sq_ctx* c);
int total_order(sq_ctx* c);  // sq_post.cpp
int order_problem_debug(sq_ctx* c, int n, const std::vector<int32_t>& edges5, bool use_gpu, int32_t& mask, std::vector<int32_t>& order, int64_t& value);

// ---- sq_kernels.hip (device side; every function enqueues on c->stream and records HIP-event timings)
int dev_create(sq_ctx* c);
void dev_destroy(sq_ctx* c);
void dev_flush_timers(sq_ctx* c);
int dev_token_bench(sq_ctx* c, const char* path, int variant, int max_blocks, int reps, int check, double* out);
int dev_append_records(sq_ctx* c, const sq_aln_batch* b);
void dev_clear_records(sq_ctx* c);
int dev_release_reader(sq_ctx* c);
int dev_stage_file(sq_ctx* c, const uint8_t* bytes, size_t n, const uint8_t** dptr);
int dev_upload_chim_names(sq_ctx* c);
int dev_parse_append(sq_ctx* c, const uint8_t* bam, size_t nbytes, const unsigned long long* rec_off, int64_t n_rec);
int dev_ingest_bgzf(sq_ctx* c, const uint8_t* file, std::vector<BgzfRange>& blocks, size_t b0, size_t b1, size_t begin, bool synced, int nref, const IndexMore& index_more, size_t file_bytes, GpuFileSrc* src);
struct HostBatch;
int dev_download_records(sq_ctx* c, HostBatch& hb);
int dev_download_names(sq_ctx* c, HostBatch& hb);
int dev_chim_begin_captured(sq_ctx* c);
int dev_chim_begin(sq_ctx* c, const char* blob, size_t blob_bytes, const uint32_t* off, const uint32_t* len, size_t n);
int dev_chim_finalize(sq_ctx* c, const std::vector<std::string>& dead_names);
struct SegSupport {
    std::vector<int32_t> trigger, zidx, z_ochr, z_oright, rest_cluster, rest_pos, rest_len;
};
// pass 1 over the resident records (k_pass1): scalar results; the lists stay on the device until dev_segment_support
struct Pass1Result { int64_t kept = 0, trigger_last = 0; int32_t n_rest = 0; int32_t first_kept[2] = {0, 0}; long long other_max = INT64_MIN; };
int dev_pass1(sq_ctx* c, const std::vector<int32_t>& cl_chr, const std::vector<int32_t>& cl_start, const std::vector<int32_t>& cl_right, Pass1Result& out);
int dev_segment_support(sq_ctx* c, int ncl, long long seed, SegSupport& out);
int dev_fetch_stream(sq_ctx* c, const std::vector<std::pair<int64_t, int64_t>>& ranges, const StreamRec*& compact, std::vector<int64_t>& range_off, const StreamRec* term);
int dev_classify(sq_ctx* c, int32_t last_info[4]);  // last_info (may be null): {has pass-1, its lists empty, has pass-2, its lists empty}
int dev_gather_other(sq_ctx* c, int64_t n_break, bool& has_tiny, std::vector<int32_t>& other_chr, std::vector<int32_t>& other_pos, std::vector<int32_t>& other_len, bool always_fetch);
int dev_upload_nodes(sq_ctx* c, const std::vector<Node>& nodes);  // node table + coarse position index, once per graph build
int dev_node_depth(sq_ctx* c, const std::vector<Node>& nodes, int64_t n_break, std::vector<int32_t>& support, std::vector<int64_t>& sumlen,
                   bool& need_exact_other, std::vector<int32_t>& amb_plus, std::vector<int32_t>& amb_minus, std::vector<int32_t>& unused);
int dev_concordant_edges(sq_ctx* c, const std::vector<Node>& nodes, std::vector<Edge>& unique_edges);
// K6 / K7 on the device (sq_graph_kernels.inc); the host versions in sq_graph.cpp stay as cross-checks (SQUID_HOST_FILTERS=1)
int dev_filter_by_weight(sq_ctx* c);
int dev_filter_by_interleaving(sq_ctx* c, std::vector<uint8_t>& keep);
int dev_filter_edges(sq_ctx* c, const std::vector<uint8_t>& keep);
int dev_compress_nodes(sq_ctx* c);
int dev_further_compress(sq_ctx* c);  // 2: capacity exceeded, take the host version
int dev_connected_components(sq_ctx* c, int n_nodes, const std::vector<Edge>& edges, std::vector<int32_t>& label);
struct SmallProblem { int n; int eoff, ecount; };  // edges: local u,v,hu,hv,w packed as 5 ints each
int dev_order_small(sq_ctx* c, const std::vector<SmallProblem>& probs, const std::vector<int32_t>& edges5, std::vector<int32_t>& out_mask,
                    std::vector<int32_t>& out_order, int nmax);
constexpr int ORDER_MID_NMAX = 19;  // k_order_mid: components of 9..19 nodes (out_order has this stride)
int dev_order_mid(sq_ctx* c, const std::vector<SmallProblem>& probs, const std::vector<int32_t>& edges5, std::vector<int32_t>& out_mask, std::vector<int32_t>& out_order,
                  std::vector<int32_t>& out_value, std::vector<int32_t>& out_status, bool own_stream = false);
// cur_prev: cursor position left by the records of earlier shards
struct BpBoundary {  // what the next shard needs to know about the breakpoint cursor (SegmentGraph.cpp:3157)
    int cur_end = 0;          // cursor after this shard's records, given the cur_prev it started from
    bool has_p3 = false, has_event = false;
    int64_t n_p3 = 0, absorb = 0;  // pass-3 records; those in front of the first record that moves the cursor beyond cur_prev
};
int dev_breakpoint_support(sq_ctx* c, const std::vector<std::pair<int, int>>& bps_sorted, std::vector<int32_t>& coverage, int cur_prev = 0, BpBoundary* bb = nullptr,
                           bool raw_diff = false);

}  // namespace sq
