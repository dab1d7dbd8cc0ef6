// HIP kernels of the SQUID hot path for gfx950 (MI355X) and their launch wrappers.
//
// HBM layout (DeviceRecords): the concordant alignment stream as structure-of-arrays, 32 B per record
// (refid,pos,mate_refid,mate_pos,end_pos i32; flag,totlen u16; mapq,aux u8; blk_off u32) plus 12 B per aligned
// block (refpos,matchref i32; readpos,matchread u16) -- SURVEY.md section 8(d).  Every kernel below is a
// coalesced scan over those arrays with one thread per record; the node table (a few thousand entries) and the
// breakpoint table are binary-searched out of L2.  All work is integer/index work: no MFMA anywhere.
//
// Kernel <-> reference loop:
//   k_classify      record filters + concordant/partial classification   SegmentGraph.cpp:297-303,651-683,1579-1585,3131-3142
//   k_dedup         consecutive-duplicate drop (ReadRec_t::Equal)         SegmentGraph.cpp:304-318,1587-1600 ; ReadRec.cpp:119-141
//   k_summarise     window elements for the segmentation automaton        SegmentGraph.cpp:320-337,668-699
//   k_depth_*       per-node Support / AvgDepth                           SegmentGraph.cpp:781-826
//   k_block0/k_edges LocateRead chain + raw edge emission + weight count  SegmentGraph.cpp:1207-1293,1601-1686,1943-1949
//   k_bp_*          breakpoint concordant-fragment support                SegmentGraph.cpp:3129-3166
//   k_cc_*          connected components (union-find)                     SegmentGraph.cpp:2911-2935,2986-3003
//   k_order_small   exact ordering of small components, one per workgroup SegmentGraph.cpp:3271-3314,3763-3983
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <condition_variable>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <zlib.h>

#include "sq_internal.h"
#include "sq_inflate_spec.inc"
#include "sq_resolve.inc"

#define HIPCHK(call)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess) return sq::fail(c, SQ_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

namespace sq {

// ------------------------------------------------------------------------------------------------ device state
struct RecView {  // raw device pointers, passed by value to kernels
    int64_t n, nb;
    const int32_t *refid, *pos, *mrefid, *mpos, *endpos;
    const uint16_t *flag, *totlen;
    const uint8_t *mapq, *aux;
    const uint32_t* blk_off;
    const int32_t *b_refpos, *b_matchref;
    const uint16_t *b_readpos, *b_matchread;
    const int4* b_pack;  // the same block as ONE 16-byte load: refpos, matchref, readpos | matchread << 16, 0 (k_pack_blocks; used by k_edges)
    // the fixed part of a record (32 bytes) as TWO 16-byte words side by side, read only by k_pass1, which wants every
    // field: {refid, pos, mate refid, mate pos} {flag | totlen << 16, mapq | aux << 8 | own blocks << 16, blk_off, end} (k_pack_records)
    const int4* r_pack;
};
struct NodeView {
    int32_t n, n_ref;
    const int32_t *chr, *pos, *len;
    const int32_t* chr_start;  // n_ref+1: first node index of each chromosome
    // position index: fine[fine_off[c] + (p >> NODE_FINE_SHIFT)] = node that contains the first base of that 1 KiB stretch of
    // chromosome c, and one more entry per chromosome = its last node.  The node of p lies between the entries of its
    // stretch and the next one: two loads side by side and, inside a gene, a bisection of one to three steps (round 1:
    // a 16 KiB index, a gallop and a bisection, a dozen dependent loads)
    const int32_t *fine, *fine_off;
    const int32_t* bucket_off;  // geometry of the 16 KiB index the breakpoint table uses (BPView)
    const int4* pack;  // chr, pos, len, 0 of a node as one 16-byte load (k_node_pack)
};
constexpr int NODE_BUCKET_SHIFT = 14, NODE_FINE_SHIFT = 10;

template <typename T>
struct DBuf {
    T* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 8 + 64;
        hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    hipError_t grow_keep(size_t used, size_t n, hipStream_t s) {  // keep the first `used` elements
        if (n <= cap) return hipSuccess;
        size_t want = std::max(n + n / 2, (size_t)1 << 16);
        T* q = nullptr;
        hipError_t e = hipMalloc((void**)&q, want * sizeof(T));
        if (e != hipSuccess) return e;
        if (used && p) e = hipMemcpyAsync(q, p, used * sizeof(T), hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (p) (void)hipFree(p);
        p = q;
        cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// one BGZF block for the inflate kernels: payload in the compressed bytes, inflated size and place, and where its tokens go (toff, in
// tokens from the start of the batch's token buffer: t2_tok_cap(isize) slots per block)
struct InflBlock { unsigned long long coff; uint32_t clen, isize; unsigned long long uoff, toff; };
// page-locked host staging for the small device->host results of a stage: the copies are queued without blocking and
// one stream synchronisation makes all of them visible.  A slice stays valid until the next reset().
struct Pinned {
    std::vector<std::pair<uint8_t*, size_t>> chunks;
    size_t used = 0, want = 0;
    void* take(size_t bytes) {
        bytes = (bytes + 63) & ~(size_t)63;
        want += bytes;
        if (chunks.empty() || used + bytes > chunks.back().second) {
            size_t cap = std::max<size_t>(bytes, (size_t)4 << 20);
            uint8_t* p = nullptr;
            if (hipHostMalloc((void**)&p, cap, hipHostMallocDefault) != hipSuccess) return nullptr;
            chunks.push_back(std::make_pair(p, cap));
            used = 0;
        }
        void* r = chunks.back().first + used;
        used += bytes;
        return r;
    }
    template <typename T> T* take_n(size_t n) { return (T*)take(std::max<size_t>(n, 1) * sizeof(T)); }
    void reset() {  // one chunk big enough for everything the last round asked for
        if (chunks.size() > 1) {
            for (auto& ch : chunks) (void)hipHostFree(ch.first);
            chunks.clear();
            uint8_t* p = nullptr;
            size_t cap = want + want / 2;
            if (hipHostMalloc((void**)&p, cap, hipHostMallocDefault) == hipSuccess) chunks.push_back(std::make_pair(p, cap));
        }
        used = 0; want = 0;
    }
    void release() { for (auto& ch : chunks) (void)hipHostFree(ch.first); chunks.clear(); used = 0; }
};

// one tile of k_summarise_tiles: kept records [lo, hi) of the tile at rec_base (rank_base kept records in front of it); output index = dst + (rank - lo)
struct TilePart { unsigned long long ob; long long mx; int cnt, first; };  // of 1024 tiles: kept records, running pair, largest key, first tile with a kept record
struct SumItem { int64_t rec_base; int32_t rank_base, lo, hi; int64_t dst; };

// Threads that live as long as the context and run one job at a time, job(t) on thread t = 0 .. count-1 (FileFeeder's copy threads and
// its header walker).  A thread's first HIP call sets up per-thread state inside the runtime, a few milliseconds each and one after
// the other: sixteen fresh threads per ingest had the last of them queue its first piece 30-50 ms after the first (measured), and the
// token pass of the second batch waited for exactly that piece.
struct FeedPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv, done_cv;
    std::function<void(int)> job;
    int generation = 0, count = 0, running = 0;
    bool stop = false;
    explicit FeedPool(int n) {
        for (int t = 0; t < n; ++t) th.emplace_back([this, t]() {
            int seen = 0;
            for (;;) {
                std::function<void(int)> f;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&]() { return stop || generation != seen; });
                    if (stop) return;
                    seen = generation;
                    if (t >= count) continue;
                    f = job;
                }
                f(t);
                { std::lock_guard<std::mutex> lk(mu); if (--running == 0) done_cv.notify_all(); }
            }
        });
    }
    ~FeedPool() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); for (auto& t : th) t.join(); }
    int size() const { return (int)th.size(); }
    void start(int n, std::function<void(int)> f) { { std::lock_guard<std::mutex> lk(mu); job = std::move(f); count = std::min(n, size()); running = count; ++generation; } cv.notify_all(); }
    void wait() { std::unique_lock<std::mutex> lk(mu); done_cv.wait(lk, [&]() { return running == 0; }); }
};

struct DeviceRecords {
    int64_t n = 0, nb = 0;
    Pinned pin;
    DBuf<int32_t> refid, pos, mrefid, mpos, endpos, b_refpos, b_matchref;
    DBuf<uint16_t> flag, totlen, b_readpos, b_matchread;
    DBuf<uint8_t> mapq, aux;
    DBuf<uint32_t> blk_off;
    DBuf<int4> b_pack, n_pack, r_pack;
    int64_t r_pack_n = 0;  // records [0, r_pack_n) of r_pack are up to date
    // derived
    DBuf<uint8_t> cls, keep;
    DBuf<int32_t> prev1, prev2, rank1, restoff, scratch_a, scratch_b, scratch_c, spine;
    DBuf<int32_t> part_prev, part_next, b0_a, b0_b, b0_home, fc_seg;
    DBuf<StreamRec> srec;
    DBuf<int32_t> rest_refpos, rest_matchref;
    // node table
    DBuf<int32_t> n_chr, n_bucket;  // n_chr: packed node table chr | pos | len | chr_start | bucket_off | fine_off; n_bucket: the fine index
    NodeView nv{};  // the node table of the current graph build (dev_upload_nodes)
    DBuf<int32_t> acc_a, acc_b, acc_c;  // per-node accumulator block / small tables of the later stages
    // edge hash
    DBuf<unsigned long long> h_key, okey;
    DBuf<uint32_t> h_val, oval;
    uint32_t h_slots = 1u << 16;
    DBuf<int32_t> ord_e, ord_o, ord_v;  // ordering kernel: packed input, packed output, values
    DBuf<int32_t> ord_me, ord_mo;       // k_order_mid: packed input, packed output
    DBuf<unsigned long long> tok_prof;  // SQUID_TOK_PROF
    DBuf<int32_t> g_i, g_x;             // K6 / K7 (sq_graph_kernels.inc): graph ints + scratch, CSR / neighbour scratch
    DBuf<double> g_d;
    DBuf<uint8_t> g_b;
    DBuf<long long> other64, spine64, okey64;
    DBuf<uint8_t> bam_chunk, bgzf_out, bgzf_carry;
    DBuf<uint8_t> staged;  // sq_stage_bam: the compressed bytes of a whole BAM file (+ padding for the input rings' read-ahead)
    DBuf<long long> rec_sync, rec_end;
    // GPU ingest, up to il_depth batches in flight: compressed bytes + block table + tokens of a batch
    struct InflSet { DBuf<uint8_t> in; DBuf<InflBlock> tab; DBuf<uint32_t> tok, lens; DBuf<int32_t> ntok, flags; hipEvent_t ready = nullptr, freed = nullptr, copied = nullptr; std::vector<InflBlock> host_tab; const uint8_t* src = nullptr; /* compressed bytes of the batch: `in`, or inside the staged file */ };
    static constexpr int IL_DEPTH_MAX = 8;
    int il_depth = 5;  // buffer sets in use (SQUID_IL_DEPTH, 3..8): batch k is resolved / parsed, the ones behind it are in the token pass, the last is being copied
    InflSet il_set[IL_DEPTH_MAX];
    hipStream_t il_stream[IL_DEPTH_MAX] = {};  // one per set: its host->device copies and its token pass
    // everything behind the token pass, per buffer set as well (round 6: the resolve follows the token pass on the set's stream, into the set's
    // own inflated buffer; what runs one batch after the other is only the boundary search on the library stream and the parse on the parse stream)
    struct PostSet { DBuf<uint8_t> out, big; DBuf<long long> rec_sync, rec_end; DBuf<int32_t> rec_cnt, rec_base, flags, spine; DBuf<unsigned long long> bam_off; hipEvent_t carried = nullptr; /* the tail of `out` has been copied in front of the next batch */ };
    PostSet il_post[IL_DEPTH_MAX];
    hipStream_t il_parse_stream = nullptr;
    hipStream_t order_stream = nullptr;     // k_order_mid beside k_order_small (dev_order_mid)
    int32_t* il_host = nullptr;            // page-locked: the small results of the two sets (32 ints each)
    // host -> device copies of file bytes: four threads stage 16 MiB pieces through page-locked buffers (h2d_parallel)
    static constexpr int H2D_THREADS = 4;
    // streamed read of a file (FileFeeder): the compressed bytes of the range being ingested, every piece with the event of its copy
    DBuf<uint8_t> stream_file;
    static constexpr int FEED_THREADS_MAX = 32;
    uint8_t* feed_pin[FEED_THREADS_MAX][2] = {};
    size_t feed_pin_bytes = 0;
    hipStream_t feed_stream[FEED_THREADS_MAX] = {};
    hipEvent_t feed_buf_ev[FEED_THREADS_MAX][2] = {};
    std::vector<hipEvent_t> feed_piece_ev;
    std::unique_ptr<FeedPool> feed_pool;  // FEED_THREADS_MAX copy threads + the header walker
    uint8_t* h2d_pin[H2D_THREADS][2] = {};
    hipStream_t h2d_stream[H2D_THREADS] = {};
    hipEvent_t h2d_ev[H2D_THREADS][2] = {};
    std::mutex h2d_mu;
    DBuf<int32_t> rec_cnt, rec_base;
    DBuf<unsigned long long> bam_off, chim_hash;
    DBuf<uint32_t> chim_off, chim_len;
    DBuf<char> chim_blob;
    uint32_t chim_mask = 0;
    // the table built from the chimeric records themselves (dev_chim_begin): per-slot dead flags, per-record slot of the match, inputs
    DBuf<uint8_t> chim_dead;
    DBuf<int32_t> chim_slot_of;
    DBuf<uint32_t> chim_in_off, chim_in_len;
    bool chim_provisional = false;   // the table holds every usable chimeric QNAME; dev_chim_finalize has not run yet
    hipStream_t chim_stream = nullptr;
    DBuf<int32_t> parse_nblk, parse_rel;
    DBuf<int4> parse_first2;  // k_parse_records: the first two blocks of every record of the chunk being parsed
    DBuf<int32_t> calib;
    DBuf<uint8_t> zflag;
    DBuf<int32_t> cl_chr, trig, cl_bucket;  // cl_chr: packed cluster table chr | start | right; cl_bucket: bucket_off | position index of the cluster table
    DBuf<int32_t> bp_ev, bp_before, bp_end, bp_valid, bp_bucket, stripes;
    DBuf<TilePart> tile_part;                   // k_tile_partial: sums / maxima of every 1024 tiles of k_pass1
    DBuf<unsigned int> depth_tiles;             // k_depth2: per tile its largest early node / cursor at its first record / list of tiles to correct
    DBuf<char> nm_blob; DBuf<uint32_t> nm_off; size_t nm_bytes = 0;  // QNAMEs of the records parsed while sq_ctx::capture_names is set (the chimeric BAM through K0)
    DBuf<int32_t> g_win;                        // windows of the sorted edge list for the group filters (edge_windows)
    DBuf<uint32_t> p2_list;                     // k_pass2w: the work list of k_edges (kept from the depth stage to the edge stage of the same pass)
    DBuf<int32_t> p2_words;                     // k_pass2w: [0] length of that list, [1] tiles on the list of the general depth sweep, [2..] those tiles
    int64_t p2_valid_n = -1;                    // records the list was made for (-1: none)
    DBuf<unsigned long long> bp_key, bp_front;  // breakpoint cursor: largest (chromosome, fragment start) per 256 records (k_edges_near) / in front of every tile of k_bp2
    int64_t bp_key_n = -1;                      // record count the keys were made for
    // pass 1 (k_pass1): look-back status words, kept records in front of every tile, tile sort keys, the three lists, scalars
    DBuf<int32_t> tile_cnt, tile_K, tile_zcnt2, zc_v, zc_K, zc_refid, zc_pos;
    DBuf<unsigned long long> tile_ob, zc_ob;  // tile_ob: [ntiles] pair of every tile | [ntiles] pair in front of every tile
    DBuf<int32_t> tile_rank, tile_zbase, tile_zcnt, z_idx, z_chr, z_right, rc_cluster, rc_pos, rc_len, p1_sc;
    DBuf<long long> tile_first, tile_max, r_break;
    DBuf<SumItem> sum_items;
    size_t zcap = (size_t)1 << 20, rc_cap = (size_t)1 << 18;
    int p1_zc = 0, p1_rest = 0, p1_ntiles = 0;
    std::vector<int32_t> h_tile_rank;  // host copy of tile_rank (dev_segment_support)
    DBuf<int32_t> flags;  // small device flag/counter block
    struct Pending { const char* name; double bytes; int slot; };
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    std::vector<Pending> ev_pending;
    size_t ev_used = 0;
    std::mutex ev_mu;  // (the GPU reader's planner thread brackets its kernel too)
    int64_t k1 = 0;  // kept pass-1 records
    int cl_n = 0;    // clusters in the packed table cl_chr
    RecView view() const {
        RecView v;
        v.n = n; v.nb = nb;
        v.refid = refid.p; v.pos = pos.p; v.mrefid = mrefid.p; v.mpos = mpos.p; v.endpos = endpos.p;
        v.flag = flag.p; v.totlen = totlen.p; v.mapq = mapq.p; v.aux = aux.p; v.blk_off = blk_off.p;
        v.b_refpos = b_refpos.p; v.b_matchref = b_matchref.p; v.b_readpos = b_readpos.p; v.b_matchread = b_matchread.p; v.b_pack = b_pack.p; v.r_pack = r_pack.p;
        return v;
    }
};

// classification bits (cls)
enum : uint8_t { C_P2 = 1, C_P1 = 2, C_P3 = 4, C_CONC = 8, C_PART = 16, C_HASSTUB = 32 };
// keep bits
enum : uint8_t { K_1 = 1, K_2 = 2, K_BUILD = 4, K_P3 = 8 /* copy of the class bit C_P3: the edge stage makes the breakpoint-cursor keys without reading the class byte */ };

// ------------------------------------------------------------------------------------------------ device helpers
struct DBlk { int32_t refid, refpos, matchref, readpos, matchread; bool rev; };

// own blocks in read-offset order (SortbyReadPos): CIGAR order on the forward strand, reversed otherwise
__device__ __forceinline__ DBlk own_block_sorted(const RecView& R, int64_t r, int k, int nblk, bool rev) {
    uint32_t b = R.blk_off[r] + (rev ? (uint32_t)(nblk - 1 - k) : (uint32_t)k);
    DBlk x;
    const int4 q = R.b_pack[b];  // refpos, matchref, readpos | matchread << 16: one 16-byte load instead of four
    x.refid = R.refid[r]; x.refpos = q.x; x.matchref = q.y;
    x.readpos = (int)((uint32_t)q.z & 0xffffu); x.matchread = (int)((uint32_t)q.z >> 16); x.rev = rev;
    return x;
}
__device__ __forceinline__ bool has_stub(const RecView& R, int64_t r) { return !(R.flag[r] & 0x8) && R.mrefid[r] != -1; }

// element k of list L (0 = FirstRead, 1 = SecondMate) of the stub-augmented, sorted record (tmpreadrec)
struct ListRec {
    int nown; bool first, rev, stub;
    __device__ int size(int L) const { bool own = (L == 0) == first; return own ? nown : (stub ? 1 : 0); }
};
__device__ __forceinline__ ListRec list_rec(const RecView& R, int64_t r) {
    ListRec l;
    l.nown = (int)(R.blk_off[r + 1] - R.blk_off[r]);
    l.first = R.flag[r] & 0x40;
    l.rev = R.flag[r] & 0x10;
    l.stub = has_stub(R, r);
    return l;
}
__device__ __forceinline__ void list_key(const RecView& R, int64_t r, const ListRec& l, int L, int k, int& id, int& p, int& m) {
    bool own = (L == 0) == l.first;
    if (own) { DBlk b = own_block_sorted(R, r, k, l.nown, l.rev); id = b.refid; p = b.refpos; m = b.matchref; }
    else { id = R.mrefid[r]; p = R.mpos[r]; m = 15; }
}
// ReadRec_t::Equal on two records; q < 0 stands for the empty initial lastreadrec
__device__ bool rec_equal(const RecView& R, int64_t q, int64_t r) {
    ListRec lr = list_rec(R, r);
    ListRec lq;
    if (q >= 0) lq = list_rec(R, q);
    else { lq.nown = 0; lq.first = true; lq.rev = false; lq.stub = false; }
    for (int swap = 0; swap < 2; ++swap) {
        int q0 = lq.size(swap ? 1 : 0), q1 = lq.size(swap ? 0 : 1);
        if (q0 != lr.size(0) || q1 != lr.size(1)) continue;
        bool same = true;
        for (int L = 0; L < 2 && same; ++L) {
            int n = lr.size(L);
            for (int k = 0; k < n; ++k) {
                int a0, a1, a2, b0, b1, b2;
                list_key(R, r, lr, L, k, a0, a1, a2);
                list_key(R, q, lq, swap ? 1 - L : L, k, b0, b1, b2);
                if (a0 != b0 || a1 != b1 || a2 != b2) { same = false; break; }
            }
        }
        if (same) return true;
    }
    return false;
}

// last node with (chr,pos) <= (c,p); nodes tile every chromosome, so this is the node containing p
__device__ __forceinline__ int node_home_search(const NodeView& N, int c, int p) {
    int lo = N.chr_start[c], hi = N.chr_start[c + 1];  // [lo,hi)
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (N.pos[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
}
// same result through the fine index
__device__ __forceinline__ int node_home(const NodeView& N, int c, int p) {
    const int first = N.fine_off[c], last = N.fine_off[c + 1] - 2;  // last stretch of c (the entry behind it is the chromosome's last node)
    int b = first + (p < 0 ? 0 : (p >> NODE_FINE_SHIFT));
    if (b > last) b = last;  // a position behind the reference end
    int lo = N.fine[b], hi = N.fine[b + 1];
    while (hi > lo) {
        const int mid = (lo + hi + 1) >> 1;
        if (N.pos[mid] <= p) lo = mid; else hi = mid - 1;
    }
    return lo;
}
// the same with the chromosome's index geometry (fine_off[c], fine_off[c + 1]) already in registers
__device__ __forceinline__ int node_home_geo(const NodeView& N, int first, int next, int p) {
    int b = first + (p < 0 ? 0 : (p >> NODE_FINE_SHIFT));
    if (b > next - 2) b = next - 2;
    int lo = N.fine[b], hi = N.fine[b + 1];
    while (hi > lo) {
        const int mid = (lo + hi + 1) >> 1;
        if (N.pos[mid] <= p) lo = mid; else hi = mid - 1;
    }
    return lo;
}
__global__ void k_node_buckets(NodeView N, int total, int32_t* fine) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    int lo = 0, hi = N.n_ref;  // chromosome of this entry: last c with fine_off[c] <= g
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (N.fine_off[mid] <= g) lo = mid; else hi = mid; }
    fine[g] = g == N.fine_off[lo + 1] - 1 ? N.chr_start[lo + 1] - 1 : node_home_search(N, lo, (g - N.fine_off[lo]) << NODE_FINE_SHIFT);
}

__global__ void k_node_pack(int n, const int32_t* chr, const int32_t* pos, const int32_t* len, int4* pack) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pack[i] = make_int4(chr[i], pos[i], len[i], 0);
}
__global__ void k_pack_records(int64_t from, int64_t to, RecView R, int4* pack) {
    const int64_t r = from + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= to) return;
    const uint32_t bo = R.blk_off[r], nblk = R.blk_off[r + 1] - bo;  // (at most 256 own blocks: k_parse_records refuses more)
    pack[2 * r] = make_int4(R.refid[r], R.pos[r], R.mrefid[r], R.mpos[r]);
    pack[2 * r + 1] = make_int4((int)((uint32_t)R.flag[r] | ((uint32_t)R.totlen[r] << 16)), (int)((uint32_t)R.mapq[r] | ((uint32_t)R.aux[r] << 8) | (nblk << 16)), (int)bo, R.endpos[r]);
}
__global__ void k_pack_blocks(int64_t from, int64_t to, const int32_t* refpos, const int32_t* matchref, const uint16_t* readpos, const uint16_t* matchread, int4* pack) {
    const int64_t b = from + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < to) pack[b] = make_int4(refpos[b], matchref[b], (int)((uint32_t)readpos[b] | ((uint32_t)matchread[b] << 16)), 0);
}

// ------------------------------------------------------------------------------------------------ scans
// Device scans over int32 values produced by a functor (record parse, component labels, debug paths); the scans that run along the
// record stream live inside the fused kernels of sq_pass_kernels.inc.
#ifndef SQ_SCAN_ITEMS
#define SQ_SCAN_ITEMS 16
#endif
constexpr int SCAN_THREADS = 256, SCAN_ITEMS = SQ_SCAN_ITEMS, SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

struct OpSum { typedef int T; static __device__ __forceinline__ int id() { return 0; } static __device__ __forceinline__ int op(int a, int b) { return a + b; } };
struct OpMax { typedef int T; static __device__ __forceinline__ int id() { return INT_MIN; } static __device__ __forceinline__ int op(int a, int b) { return a > b ? a : b; } };
struct OpMin { typedef int T; static __device__ __forceinline__ int id() { return INT_MAX; } static __device__ __forceinline__ int op(int a, int b) { return a < b ? a : b; } };
struct OpMax64 {
    typedef long long T;
    static __device__ __forceinline__ long long id() { return LLONG_MIN; }
    static __device__ __forceinline__ long long op(long long a, long long b) { return a > b ? a : b; }
};

template <typename Op>
__device__ __forceinline__ typename Op::T block_scan_excl(typename Op::T v, typename Op::T& total, typename Op::T* lds) {  // exclusive scan, one value per thread
    typedef typename Op::T T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T x = v;
    for (int d = 1; d < 64; d <<= 1) {
        T y = __shfl_up(x, d, 64);
        if (lane >= d) x = Op::op(x, y);
    }
    if (lane == 63) lds[wave] = x;
    __syncthreads();
    T wprefix = Op::id();
    T tot = Op::id();
    for (int w = 0; w < SCAN_THREADS / 64; ++w) {
        T t = lds[w];
        if (w < wave) wprefix = Op::op(wprefix, t);
        tot = Op::op(tot, t);
    }
    __syncthreads();
    total = tot;
    T incl = Op::op(wprefix, x);
    T prev = __shfl_up(incl, 1, 64);
    if (lane == 0) prev = wprefix;
    return prev;
}

// Single-pass scan with decoupled look-back (Merrill & Garland) for the 32-bit operators: every tile takes a ticket (tiles start
// in ticket order, so a tile only ever waits for tiles that are already running), reads its 2048 inputs ONCE, publishes its
// aggregate, looks back over the status words of the tiles in front of it -- one wave, 64 tiles per look -- until it meets an
// inclusive prefix, publishes its own inclusive prefix and writes its outputs (inputs and outputs cross LDS, so that global
// memory is always touched with consecutive lanes on consecutive elements).  One launch and one read of the input instead of
// reduce + spine + down-sweep (three launches, two reads).  A status word is (epoch << 2 | state) : value in 64 bits, written
// and read with agent-scope atomics (the L2s of the XCDs are not coherent with each other); the epoch grows with every scan,
// so the words never need clearing, and the tile that takes the last ticket puts the ticket counter back to zero.
static std::atomic<uint32_t> g_scan_epoch{0};
template <typename Op, bool EXCL, typename F>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_lookback(int64_t n, F f, int* out, unsigned long long* state /* [0]: ticket counter, [1 + tile]: status */, uint32_t epoch, int ntiles, int* grand) {
    __shared__ int lds[SCAN_THREADS / 64];
    __shared__ int s_tile, s_prefix;
    if (threadIdx.x == 0) {
        const int t = atomicAdd((int*)state, 1);
        if (t == ntiles - 1) __hip_atomic_store((int*)state, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_tile = t;
    }
    __syncthreads();
    const int tile = s_tile;
    unsigned long long* st = state + 1;
    // the inputs are evaluated with consecutive lanes on consecutive elements (coalesced, whatever the functor reads) and turned
    // into the blocked order of the scan through LDS; one padding word per 32 keeps both access shapes free of bank conflicts
    __shared__ int tilebuf[SCAN_TILE + SCAN_TILE / 32];
    auto pad = [](int i) { return i + (i >> 5); };
    const int64_t tbase = (int64_t)tile * SCAN_TILE;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int64_t idx = tbase + i * SCAN_THREADS + threadIdx.x;
        tilebuf[pad(i * SCAN_THREADS + (int)threadIdx.x)] = idx < n ? (int)f(idx) : Op::id();
    }
    __syncthreads();
    int v[SCAN_ITEMS];
    int acc = Op::id();
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = tilebuf[pad((int)threadIdx.x * SCAN_ITEMS + i)];
        acc = Op::op(acc, v[i]);
    }
    int total;
    const int ex = block_scan_excl<Op>(acc, total, lds);
    const unsigned long long tag = (unsigned long long)epoch << 34;  // state 1: aggregate, state 2: inclusive prefix
    if (threadIdx.x < 64) {  // wave 0 publishes and looks back
        const int lane = threadIdx.x;
        int prefix = Op::id();
        if (tile == 0) { if (lane == 0) __hip_atomic_store(&st[0], tag | (2ull << 32) | (uint32_t)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else {
            if (lane == 0) __hip_atomic_store(&st[tile], tag | (1ull << 32) | (uint32_t)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int look = tile - 1; look >= 0; look -= 64) {
                const int j = look - lane;  // this lane's tile, nearest first
                unsigned long long w = 0;
                for (;;) {
                    if (j >= 0) w = __hip_atomic_load(&st[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const bool ready = j < 0 || ((w >> 34) == epoch && ((w >> 32) & 3));
                    if (__all(ready)) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                const bool incl = j >= 0 && ((w >> 32) & 3) == 2;
                const unsigned long long im = __ballot(incl);
                const int stop = im ? __ffsll((long long)im) - 1 : 63;  // lanes 0..stop contribute (stop holds an inclusive prefix, or the window ends)
                int x = (j >= 0 && lane <= stop) ? (int)(uint32_t)w : Op::id();
                for (int d = 32; d >= 1; d >>= 1) x = Op::op(x, __shfl_xor(x, d, 64));
                prefix = Op::op(x, prefix);
                if (im) break;
            }
            if (lane == 0) __hip_atomic_store(&st[tile], tag | (2ull << 32) | (uint32_t)Op::op(prefix, total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) { s_prefix = prefix; if (grand && tile == ntiles - 1) *grand = Op::op(prefix, total); }
    }
    __syncthreads();
    int run = Op::op(s_prefix, ex);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {  // (every thread read its inputs before the barriers of the block scan: the buffer is free)
        int o;
        if (EXCL) { o = run; run = Op::op(run, v[i]); }
        else { run = Op::op(run, v[i]); o = run; }
        tilebuf[pad((int)threadIdx.x * SCAN_ITEMS + i)] = o;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int64_t idx = tbase + i * SCAN_THREADS + threadIdx.x;
        if (idx < n) out[idx] = tilebuf[pad(i * SCAN_THREADS + (int)threadIdx.x)];
    }
}

template <typename Op, bool EXCL, typename F>
static hipError_t device_scan(hipStream_t s, int64_t n, F f, typename Op::T* out, DBuf<typename Op::T>& spine, typename Op::T* grand) {
    if (n <= 0) { if (grand) return hipMemsetAsync(grand, 0, sizeof(typename Op::T), s); return hipSuccess; }
    int ntiles = (int)((n + SCAN_TILE - 1) / SCAN_TILE);
    static_assert(sizeof(typename Op::T) == 4, "the look-back scan carries 32-bit values");
    const size_t cap_before = spine.cap;
    hipError_t e = spine.reserve(2 * (size_t)ntiles + 4);
    if (e != hipSuccess) return e;
    if (spine.cap != cap_before) { e = hipMemsetAsync(spine.p, 0, spine.cap * sizeof(typename Op::T), s); if (e != hipSuccess) return e; }  // fresh memory: no stale status words, ticket counter at zero
    const uint32_t epoch = (g_scan_epoch.fetch_add(1) + 1) & 0x3fffffffu;
    hipLaunchKernelGGL((k_scan_lookback<Op, EXCL, F>), dim3(ntiles), dim3(SCAN_THREADS), 0, s, n, f, (int*)out, (unsigned long long*)spine.p, epoch, ntiles, (int*)grand);
    return hipGetLastError();
}

// calibration kernel for the rocprofv3 FETCH_SIZE counter (MI355X_MICROARCH.md, HBM section): a plain coalesced
// 4-byte-per-lane read of a known number of bytes, the access shape of the record scans
__global__ void k_calib_read4(const int32_t* a, int64_t n, int32_t* out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int acc = 0;
    for (; i < n; i += (int64_t)gridDim.x * blockDim.x) acc += a[i];
    if (acc == 0x7fffffff) out[0] = acc;
}

// ------------------------------------------------------------------------------------------------ K0: BAM record parse
// One thread per BAM record of an inflated chunk resident in HBM (record offsets come from the host's boundary
// walk).  Restates the per-record part of the reference's ingest -- BamTools field decode plus ReadRec_t::ReadRec_t
// (src/ReadRec.cpp:10-88): TotalLen, longest low-Phred run, CIGAR -> aligned blocks with the poly-A/T filter and
// strand-mirrored read offsets, GetEndPosition(), XA / IH tags (src/SegmentGraph.cpp:297-301) and the QNAME lookup in
// the chimeric name set (:302) -- and writes the SoA layout directly.  Two passes: count blocks, scan, write.
struct ChimSetView { uint32_t mask; const unsigned long long* hash; const uint32_t *off, *len; const char* blob; const uint8_t* dead; /* per slot: the name was dropped from the set afterwards (may be null) */ };
struct ParseParams { int qual_thr, max_lowphred_len, min_mapq, chim; /* chim: the records are the chimeric BAM's (the violated-assert rule of BuildChimericSBamRecord's reader) */ };
__device__ __forceinline__ int ld32(const uint8_t* p) { return (int)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24)); }
__device__ __forceinline__ int ld16(const uint8_t* p) { return (int)p[0] | ((int)p[1] << 8); }
__device__ __forceinline__ char cig_type(uint32_t v) {  // "MIDNSHP=X", packed into registers (an indexed local array becomes a memory load per op)
    const unsigned long long lo = 0x3d5048534e44494dull;  // 'M' 'I' 'D' 'N' 'S' 'H' 'P' '='
    const uint32_t k = v & 0xf;
    return k < 8 ? (char)(lo >> (8 * k)) : (k == 8 ? 'X' : '?');
}
// slot of a name hash: FNV-1a of short, nearly sequential names ("r1234567") leaves its middle bits clustered -- probe chains
// hundreds of slots long -- so the hash goes through a 64-bit finaliser first
__host__ __device__ __forceinline__ uint32_t chim_slot(unsigned long long h, uint32_t mask) {
    h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 33;
    return (uint32_t)h & mask;
}
__device__ __forceinline__ unsigned long long chim_hash_of(const uint8_t* name, int n) {
    unsigned long long h = 1469598103934665603ull;
    for (int i = 0; i < n; ++i) { h ^= name[i]; h *= 1099511628211ull; }
    return h ? h : 1;
}
// slot of `name` in the table (the first one on its probe chain: a table made from the records of the chimeric BAM holds a name once
// per record), or -1
__device__ int chim_find(const ChimSetView& C, const uint8_t* name, int n) {
    if (!C.hash) return -1;
    const unsigned long long h = chim_hash_of(name, n);
    for (uint32_t s = chim_slot(h, C.mask), probes = 0; probes <= C.mask; s = (s + 1) & C.mask, ++probes) {
        unsigned long long e = C.hash[s];
        if (e == 0) return -1;
        if (e == h && (int)C.len[s] == n) {
            bool same = true;
            for (int i = 0; i < n; ++i) if (C.blob[C.off[s] + i] != (char)name[i]) { same = false; break; }
            if (same) return (int)s;
        }
    }
    return -1;
}
// The table of the chimeric QNAMEs, built on the device from the decoded chimeric records as soon as they are decoded (sq_ingest_files:
// the record parse of the concordant BAM only waits for THIS, not for the pairing of the chimeric records).  The reference's set holds
// the names of the fragments that survive its PCR-duplicate removal (SegmentGraph.cpp:196-201); the names of the fragments dropped
// there are marked dead afterwards (k_chim_mark_dead) and the records that matched one lose their bit again (k_chim_fixup).
__global__ void k_chim_insert(const char* blob, const uint32_t* in_off, const uint32_t* in_len, int n, uint32_t mask, unsigned long long* hash, uint32_t* off, uint32_t* len) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t o = in_off[i], l = in_len[i];
    if (l == 0xffffffffu) return;  // (k_chim_entries: a record that is not usable)
    const unsigned long long h = chim_hash_of((const uint8_t*)blob + o, (int)l);
    for (uint32_t s = chim_slot(h, mask);; s = (s + 1) & mask)  // (every record takes a slot of its own: no name compare, nobody reads off / len in here)
        if (atomicCAS(&hash[s], 0ull, h) == 0ull) { off[s] = o; len[s] = l; return; }
}
// the same entries from the records of the chimeric BAM where K0 left them (sq_ctx::capture_names): one per usable record -- mapped, not a
// duplicate (ReadRec.cpp:344) --, the name with a trailing /1 or /2 cut off (ReadRec.cpp:62-66); entry n is the empty name the reference's
// set always holds (SegmentGraph.cpp:196-201, ledger B9)
__global__ void k_chim_entries(const char* blob, const uint32_t* nm_off, uint32_t nm_end, const uint16_t* flag, int64_t n, uint32_t* in_off, uint32_t* in_len) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) { in_off[i] = 0; in_len[i] = 0; return; }
    const uint32_t o = nm_off[i], e = i + 1 < n ? nm_off[i + 1] : nm_end;
    uint32_t L = e - o;
    if (L >= 2 && blob[o + L - 2] == '/' && (blob[o + L - 1] == '1' || blob[o + L - 1] == '2')) L -= 2;
    const int f = flag[i];
    in_off[i] = o;
    in_len[i] = ((f & 0x4) || (f & 0x400)) ? 0xffffffffu : L;
}
__global__ void k_chim_mark_dead(const char* dblob, const uint32_t* d_off, const uint32_t* d_len, int n, ChimSetView C, uint8_t* dead) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* name = (const uint8_t*)dblob + d_off[i];
    const int l = (int)d_len[i];
    const unsigned long long h = chim_hash_of(name, l);
    for (uint32_t s = chim_slot(h, C.mask), probes = 0; probes <= C.mask; s = (s + 1) & C.mask, ++probes) {  // every slot that holds the name
        const unsigned long long e = C.hash[s];
        if (e == 0) return;
        if (e == h && (int)C.len[s] == l) {
            bool same = true;
            for (int k = 0; k < l; ++k) if (C.blob[C.off[s] + k] != (char)name[k]) { same = false; break; }
            if (same) dead[s] = 1;
        }
    }
}
// chim_slot[r]: 0 = the record's QNAME matched nothing; s + 1 = it matched slot s; -(s + 1) = it matched slot s and the reference would
// have asserted on the record (ReadRec.cpp:64) had it not been filtered out by exactly that match
__global__ void k_chim_fixup(int64_t n, const int32_t* chim_slot_of, const uint8_t* dead, uint8_t* aux, int32_t* flags) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int32_t cs = chim_slot_of[r];
    if (cs == 0) return;
    const int32_t slot = (cs < 0 ? -cs : cs) - 1;
    if (!dead[slot]) return;
    aux[r] = (uint8_t)(aux[r] & ~SQ_AUX_INCHIM);
    if (cs < 0) atomicOr(&flags[0], 256);
}
// Byte reads of the record parse, from LDS (the staged records of a workgroup) or from global memory (a range that does not fit):
// typed by address space, so the staged path is ds_read and not a flat load that has to find out where it goes, and up to eight bytes
// per load at any alignment (gfx950 takes unaligned LDS and global accesses; a read may run up to 7 bytes past what it needs:
// the staging buffer and the chunk buffer are padded)
typedef wv::lds_u8 lds_u8;
typedef uint64_t __attribute__((aligned(1))) u64_any;
typedef uint32_t __attribute__((aligned(1))) u32_any;
typedef uint16_t __attribute__((aligned(1))) u16_any;
__device__ __forceinline__ uint32_t rd8(const uint8_t* p) { return *p; }
__device__ __forceinline__ uint32_t rd8(const lds_u8* p) { return *p; }
__device__ __forceinline__ uint32_t rd16(const uint8_t* p) { return *(const u16_any*)p; }
__device__ __forceinline__ uint32_t rd16(const lds_u8* p) { return *(const __attribute__((address_space(3))) u16_any*)p; }
__device__ __forceinline__ uint32_t rd32(const uint8_t* p) { return *(const u32_any*)p; }
__device__ __forceinline__ uint32_t rd32(const lds_u8* p) { return *(const __attribute__((address_space(3))) u32_any*)p; }
__device__ __forceinline__ unsigned long long rd64(const uint8_t* p) { return *(const u64_any*)p; }
__device__ __forceinline__ unsigned long long rd64(const lds_u8* p) { return *(const __attribute__((address_space(3))) u64_any*)p; }
// CIGAR -> aligned blocks (ReadRec.cpp:23-60), 16 bytes each as b_pack holds them (refpos, matchref, readpos | matchread << 16, 0).
// PLACE = false: the first two kept blocks are returned in q0 / q1; PLACE = true: kept block k < cap goes to S's arrays at b0 + k.
// Returns the block count, or -1 when the reference's assert(ReadPos >= HardClipOffset && ...) (ReadRec.cpp:64) would fire.
// The poly-A/T filter counts the 4-bit base codes 1 (A) and 8 (T) of the block's stretch of the read sixteen bases per load.
struct BlockSink { int32_t *refpos, *matchref; uint16_t *readpos, *matchread; int4* pack;
    __device__ __forceinline__ void put(uint32_t at, int4 q) const { refpos[at] = q.x; matchref[at] = q.y; readpos[at] = (uint16_t)((uint32_t)q.z & 0xffffu); matchread[at] = (uint16_t)((uint32_t)q.z >> 16); pack[at] = q; } };
template <bool PLACE, class PTR>
__device__ __forceinline__ int parse_blocks(PTR cg, int ncig, PTR seq, int lseq, int pos, bool rev, int totlen, int4& q0, int4& q1, const BlockSink& S, uint32_t b0, int cap) {
    int readpos = 0, refpos = pos, hardclip = 0, nb = 0;
    for (int ic = 0; ic < ncig; ++ic) {
        uint32_t v = rd32(cg + 4 * ic);
        char t = cig_type(v);
        int len = (int)(v >> 4);
        if (t == 'S' || t == 'H') {
            readpos += len;
            if (t == 'H') hardclip += len;
        } else if (t == 'M' || t == '=') {
            int tr = 0, tf = 0, ic2;
            for (ic2 = ic; ic2 < ncig; ++ic2) {
                uint32_t v2 = rd32(cg + 4 * ic2);
                char t2 = cig_type(v2);
                if (t2 == 'S' || t2 == 'H' || t2 == 'N') break;
                if (t2 != 'D') tr += (int)(v2 >> 4);
                if (t2 != 'I') tf += (int)(v2 >> 4);
            }
            int s0 = readpos - hardclip, s1 = readpos + tr - hardclip;
            if (!(readpos >= hardclip && s1 <= lseq)) return -1;
            int na = 0, nt = 0;
            // base i sits in byte i >> 1, high nibble first: swapping the nibbles of every byte of a little-endian word puts base
            // wb + k at nibble k, and a nibble equals c when (word ^ c c c ...) has no bit set in it
            for (int wb = s0 & ~1; wb < s1; wb += 16) {
                unsigned long long w = rd64(seq + (wb >> 1));
                w = ((w & 0x0f0f0f0f0f0f0f0full) << 4) | ((w >> 4) & 0x0f0f0f0f0f0f0f0full);
                const int a = s0 > wb ? s0 - wb : 0, b = s1 - wb < 16 ? s1 - wb : 16;
                const unsigned long long lo = (1ull << (4 * a)) - 1, hi = b == 16 ? ~0ull : (1ull << (4 * b)) - 1;
                const unsigned long long ones = 0x1111111111111111ull & hi & ~lo;
                unsigned long long x = w ^ 0x1111111111111111ull, y = w ^ 0x8888888888888888ull;
                x |= x >> 1; x |= x >> 2;
                y |= y >> 1; y |= y >> 2;
                na += __popcll(~x & ones);
                nt += __popcll(~y & ones);
            }
            if (4 * na < 3 * tr && 4 * nt < 3 * tr) {
                const int4 q = make_int4(refpos, tf, (int)((uint32_t)(uint16_t)(rev ? totlen - readpos - tr : readpos) | ((uint32_t)(uint16_t)tr << 16)), 0);
                if (PLACE) { if (nb < cap) S.put(b0 + (uint32_t)nb, q); }
                else { if (nb == 0) q0 = q; if (nb == 1) q1 = q; }
                ++nb;
            }
            readpos += tr;
            refpos += tf;
            ic = ic2 - 1;
        } else if (t == 'N')
            refpos += len;
    }
    return nb;
}
// The records of a workgroup are one contiguous byte range of the chunk (64 records, ~17 KB).  Parsing walks them per lane, 260 bytes
// apart from the neighbouring lane: straight from global memory every load touches 64 cache lines and the chunk is fetched from
// HBM ~20 times over.  So the range is first copied into LDS with coalesced 16-byte loads and the lanes parse from there (a range
// that does not fit -- very long records -- is parsed in place).
#ifndef SQ_PARSE_THREADS
#define SQ_PARSE_THREADS 64
#define SQ_PARSE_LDS 18432
#endif
constexpr int PARSE_THREADS = SQ_PARSE_THREADS, PARSE_LDS = SQ_PARSE_LDS;  // 4 workgroups = 8 waves per CU (64 threads x 32 KB gave 5)
// Copies the workgroup's records into `lds` when they fit; returns whether they did (uniform).  For its record every lane gets the
// offset into the staged range / the chunk and `avail`, the bytes the record may occupy: up to the next record's offset, the end of the
// chunk and (when staged) the end of the staged range; -1 when the offsets handed in by the caller are not ascending or lie outside the chunk.
template <int LDS_BYTES>
__device__ __forceinline__ bool stage_records(const uint8_t* bam, size_t nbytes, const unsigned long long* rec_off, int64_t n, uint8_t* lds, unsigned long long& at, long long& avail) {
    const int64_t r0 = (int64_t)blockIdx.x * blockDim.x, r1 = r0 + blockDim.x < n ? r0 + blockDim.x : n;
    const unsigned long long first = rec_off[r0], lo = first & ~15ull;
    unsigned long long hi = r1 < n ? rec_off[r1] : (unsigned long long)nbytes;
    if (hi > (unsigned long long)nbytes) hi = (unsigned long long)nbytes;
    bool fits = first <= hi && hi - lo <= (unsigned long long)LDS_BYTES;  // uniform over the workgroup
    const int64_t r = r0 + threadIdx.x;
    avail = -1; at = 0;
    bool mine = true;  // my record lies inside the staged range
    unsigned long long o = 0, e = 0;
    if (r < n) {
        o = rec_off[r]; e = r + 1 < n ? rec_off[r + 1] : (unsigned long long)nbytes;
        mine = o >= first && o <= hi;
    }
    fits = __syncthreads_and(fits && mine) != 0;
    if (fits) {
        const uint4* src = (const uint4*)(bam + lo);  // (the chunk buffer is 256-byte aligned and padded by 64 bytes)
        uint4* dst = (uint4*)lds;
        const int words = (int)((hi - lo + 15) >> 4);
        // (every load is in flight before the first store: written as a loop the copy waits for memory once per 16 bytes and lane,
        // eighteen round trips to HBM one after the other -- four fifths of the kernel's time in round 6's first form)
        // (up to 23 loads per lane and pass: the 18 and 22 KB staging sizes in one pass, the larger ones -- longer reads -- in two or three)
        constexpr int PER_ALL = (LDS_BYTES / 16 + PARSE_THREADS - 1) / PARSE_THREADS + 1, PER = PER_ALL < 23 ? PER_ALL : 23;
        for (int w0 = 0; w0 < words; w0 += PER * PARSE_THREADS) {
            uint4 v[PER];
#pragma unroll
            for (int k = 0; k < PER; ++k) { const int i = w0 + (int)threadIdx.x + k * PARSE_THREADS; v[k] = src[i < words ? i : words - 1]; }  // (no branch around a load: the array stays in registers)
#pragma unroll
            for (int k = 0; k < PER; ++k) asm volatile("" : "+v"(v[k].x), "+v"(v[k].y), "+v"(v[k].z), "+v"(v[k].w));  // (and the loads stay in front of the stores)
#pragma unroll
            for (int k = 0; k < PER; ++k) { const int i = w0 + (int)threadIdx.x + k * PARSE_THREADS; if (i < words) dst[i] = v[k]; }
        }
    }
    __syncthreads();
    if (r >= n) return fits;
    if (o > (unsigned long long)nbytes || e > (unsigned long long)nbytes || e < o) return fits;  // (avail stays -1: the caller flags the record)
    const unsigned long long lim = fits ? hi - o : (unsigned long long)nbytes - o;
    avail = (long long)(e - o < lim ? e - o : lim);
    at = fits ? o - lo : o;
    return fits;
}
// block_size and the fixed fields must fit into `avail`, the variable-length fields into block_size (a malformed record would
// otherwise send the cigar / sequence / quality / tag walks past the chunk or the staging buffer)
template <class PTR>
__device__ __forceinline__ bool rec_header_ok(PTR rec, long long avail) {
    if (avail < 36) return false;
    const long long bs = (int)rd32(rec);
    if (bs < 32 || 4 + bs > avail) return false;
    const long long lname = rd8(rec + 12), ncig = rd16(rec + 16), lseq = (int)rd32(rec + 20);
    if (lseq < 0) return false;
    return 32 + lname + 4 * ncig + (lseq + 1) / 2 + lseq <= bs;
}
// slot of `name` (n bytes at `name`) in the table of the chimeric QNAMEs, as chim_find; the name is hashed eight bytes per load
template <class PTR>
__device__ __forceinline__ int chim_find_at(const ChimSetView& C, PTR name, int n) {
    if (!C.hash) return -1;
    unsigned long long h = 1469598103934665603ull;
    for (int i = 0; i < n; i += 8) {
        unsigned long long w = rd64(name + i);
        const int m = n - i < 8 ? n - i : 8;
        for (int j = 0; j < m; ++j) { h ^= w & 0xff; h *= 1099511628211ull; w >>= 8; }
    }
    if (!h) h = 1;
    for (uint32_t s = chim_slot(h, C.mask), probes = 0; probes <= C.mask; s = (s + 1) & C.mask, ++probes) {
        unsigned long long e = C.hash[s];
        if (e == 0) return -1;
        if (e == h && (int)C.len[s] == n) {
            bool same = true;
            for (int i = 0; i < n; ++i) if (C.blob[C.off[s] + i] != (char)rd8(name + i)) { same = false; break; }
            if (same) return (int)s;
        }
    }
    return -1;
}
// Everything of one record except where its blocks go: the record's SoA fields, the number of kept blocks, and the first two of them
// (16 bytes each, as b_pack holds them) in a scratch array -- k_parse_place moves them behind a scan of the counts and parses the
// CIGAR of a record with more blocks a second time.  One pass over the inflated bytes instead of the two (count, write) of rounds 1-5.
struct ParseOut {
    int32_t *refid, *pos, *mrefid, *mpos, *endpos; uint16_t *flag, *totlen; uint8_t *mapq, *aux; int32_t* chimslot;
    int32_t* nblk; int4* first2;
};
template <class PTR>
__device__ __forceinline__ void parse_record(PTR rec, long long avail, int64_t r, const ChimSetView& C, const ParseParams& P, const ParseOut& O, int32_t* flags) {
    if (!rec_header_ok(rec, avail)) { atomicOr(&flags[0], 128); O.nblk[r] = 0; return; }
    PTR p = rec + 4;
    const int bs = (int)rd32(rec);
    PTR pend = p + bs;
    const int refid = (int)rd32(p), pos = (int)rd32(p + 4), lname = (int)rd8(p + 8), mapq = (int)rd8(p + 9), ncig = (int)rd16(p + 12), flag = (int)rd16(p + 14), lseq = (int)rd32(p + 16),
              mrefid = (int)rd32(p + 20), mpos = (int)rd32(p + 24);
    PTR name = p + 32;
    PTR cg = name + lname;
    PTR seq = cg + 4 * (size_t)ncig;
    PTR qual = seq + (lseq + 1) / 2;
    PTR aux = qual + lseq;
    if (aux > pend) { atomicOr(&flags[0], 128); O.nblk[r] = 0; return; }
    int totlen = 0, endpos = pos;
    for (int i = 0; i < ncig; ++i) {
        uint32_t v = rd32(cg + 4 * i);
        char t = cig_type(v);
        int len = (int)(v >> 4);
        if (t == 'M' || t == 'S' || t == 'H' || t == 'I' || t == '=' || t == 'X') totlen += len;
        if (t == 'M' || t == 'D' || t == 'N' || t == '=' || t == 'X') endpos += len;
    }
    int lowrun = 0, run = 0;
    for (int i = 0; i < lseq; i += 8) {
        unsigned long long w = rd64(qual + i);
        const int m = lseq - i < 8 ? lseq - i : 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int c = (signed char)(((uint32_t)(w >> (8 * j)) + 33) & 0xff);
            run = (j < m && c < P.qual_thr) ? run + 1 : 0;
            if (run > lowrun) lowrun = run;
        }
    }
    // aux: XA present, first IH value (integer typed)
    bool has_xa = false, has_ih = false, bad = false;
    int ih = 0;
    for (PTR q = aux; q + 3 <= pend;) {
        const uint32_t hd = rd32(q);  // tag, type and the first byte of the value (or of whatever follows a value-less end: checked below)
        const uint8_t t0 = hd & 0xff, t1 = (hd >> 8) & 0xff, ty = (hd >> 16) & 0xff;
        PTR v = q + 3;
        size_t sz;
        if (ty == 'A' || ty == 'c' || ty == 'C') sz = 1;
        else if (ty == 's' || ty == 'S') sz = 2;
        else if (ty == 'i' || ty == 'I' || ty == 'f') sz = 4;
        else if (ty == 'Z' || ty == 'H') {
            // the terminating NUL, eight bytes per load
            PTR z = v;
            bool found = false;
            while (z < pend) {
                const unsigned long long w = rd64(z);
                const unsigned long long zero = (w - 0x0101010101010101ull) & ~w & 0x8080808080808080ull;
                if (zero) { z += (__ffsll((long long)zero) - 1) >> 3; found = z < pend; break; }
                z += 8;
            }
            if (!found) { bad = true; break; }
            sz = (size_t)(z - v) + 1;
        } else if (ty == 'B') {
            if (v + 5 > pend) { bad = true; break; }
            const uint32_t et = rd8(v);
            size_t es = (et == 'c' || et == 'C') ? 1 : ((et == 's' || et == 'S') ? 2 : 4);
            sz = 5 + es * (size_t)rd32(v + 1);
        } else { bad = true; break; }
        if (v + sz > pend) { bad = true; break; }
        if (t0 == 'X' && t1 == 'A') has_xa = true;
        if (t0 == 'I' && t1 == 'H' && !has_ih) {
            has_ih = true;
            if (ty == 'c' || ty == 'C' || ty == 'A') ih = (int)rd8(v);
            else if (ty == 's' || ty == 'S') ih = (int)rd16(v);
            else if (ty == 'i') ih = (int)rd32(v);  // (BamTools' GetTag<int> refuses a UINT32 value: IHtagvalue stays 0)
        }
        q = v + sz;
    }
    if (bad) { atomicOr(&flags[0], 128); O.nblk[r] = 0; return; }
    uint8_t ax = 0;
    if (has_xa || ih > 1) ax |= SQ_AUX_MULTI;
    if (lowrun > P.max_lowphred_len) ax |= SQ_AUX_LOWPHRED;
    const int cslot = chim_find_at(C, name, lname > 0 ? lname - 1 : 0);
    if (cslot >= 0 && !(C.dead && C.dead[cslot])) ax |= SQ_AUX_INCHIM;
    int32_t cs_out = (ax & SQ_AUX_INCHIM) ? cslot + 1 : 0;
    int4 q0 = make_int4(0, 0, 0, 0), q1 = q0;
    const int nb = parse_blocks<false>(cg, ncig, seq, lseq, pos, flag & 0x10, totlen, q0, q1, BlockSink{}, 0u, 0);
    // the SoA keeps TotalLen and the read offsets in 16 bits and the segmentation summary counts a record's further blocks in
    // 8 bits: longer reads / more blocks are refused instead of wrapping silently
    if (totlen > 65535 || nb > 256) atomicOr(&flags[0], 2048);
    if (nb < 0) {
        // the reference constructs a ReadRec_t only for records that pass its filters; for those the assert is live
        bool filtered = (ax & (SQ_AUX_MULTI | SQ_AUX_INCHIM)) || (flag & 0x400) || (flag & 0x4) || mapq < P.min_mapq;
        if (P.chim) filtered = (flag & 0x400) || (flag & 0x4);  // (BuildChimericSBamRecord constructs a ReadRec_t for every mapped non-duplicate, SegmentGraph.cpp:196-201)
        if (!filtered) atomicOr(&flags[0], 256);
        const bool but_for_the_name = !(ax & SQ_AUX_MULTI) && !(flag & 0x400) && !(flag & 0x4) && mapq >= P.min_mapq;
        if (filtered && but_for_the_name) cs_out = -cs_out;  // (k_chim_fixup raises the flag should the name leave the set)
    }
    O.nblk[r] = nb < 0 ? 0 : nb;  // (a record that trips the assert owns no slots: its leading blocks are not written)
    O.first2[2 * r] = q0; O.first2[2 * r + 1] = q1;
    O.chimslot[r] = cs_out;
    O.refid[r] = refid; O.pos[r] = pos; O.mrefid[r] = mrefid; O.mpos[r] = mpos; O.endpos[r] = endpos;
    O.flag[r] = (uint16_t)flag; O.totlen[r] = (uint16_t)totlen; O.mapq[r] = (uint8_t)mapq; O.aux[r] = ax;
}
// LDS_BYTES: the staging size -- 18 KB holds 64 records of up to 288 bytes (reads of 100-150 bases), the larger sizes the records of longer reads (the host picks by
// the batch's mean record length; tools/parse_probe.py: records of 437 bytes took 4.6 times as long as records of 212 through the in-place path)
template <int LDS_BYTES>
__global__ __launch_bounds__(PARSE_THREADS) void k_parse_records(const uint8_t* bam, size_t nbytes, const unsigned long long* rec_off, int64_t n, ChimSetView C, ParseParams P, ParseOut O, int32_t* flags) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[LDS_BYTES + 32];
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long at;
    long long avail;
    const bool staged = stage_records<LDS_BYTES>(bam, nbytes, rec_off, n, lds, at, avail);
    if (r >= n) return;
    if (staged) parse_record((const lds_u8*)lds + at, avail, r, C, P, O, flags);
    else parse_record(bam + at, avail, r, C, P, O, flags);
}
// the blocks of the records to their places behind the scan of the counts (blk_rel), in both layouts (the four arrays and b_pack), and
// the records' block offsets.  A record with more than two blocks has its CIGAR parsed again, from the chunk.
struct FArrN { const int32_t* a; __device__ int operator()(int64_t i) const { return a[i]; } };
__global__ void k_parse_place(const uint8_t* bam, const unsigned long long* rec_off, int64_t n, const int32_t* blk_cnt, const int32_t* blk_rel, const int4* first2, uint32_t blk_base, const uint16_t* o_totlen,
                              uint32_t* o_blkoff, int32_t* b_refpos, int32_t* b_matchref, uint16_t* b_readpos, uint16_t* b_matchread, int4* b_pack) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int nb = blk_cnt[r];
    const uint32_t b0 = blk_base + (uint32_t)blk_rel[r];
    o_blkoff[r] = b0;
    const BlockSink S{b_refpos, b_matchref, b_readpos, b_matchread, b_pack};
    if (nb == 0) return;
    if (nb <= 2) {
        S.put(b0, first2[2 * r]);
        if (nb == 2) S.put(b0 + 1, first2[2 * r + 1]);
        return;
    }
    const uint8_t* p = bam + rec_off[r] + 4;  // (k_parse_records has checked the record)
    const int pos = (int)rd32(p + 4), lname = (int)rd8(p + 8), ncig = (int)rd16(p + 12), flag = (int)rd16(p + 14), lseq = (int)rd32(p + 16);
    const uint8_t* cg = p + 32 + lname;
    int4 q0, q1;
    parse_blocks<true>(cg, ncig, cg + 4 * (size_t)ncig, lseq, pos, flag & 0x10, (int)o_totlen[r], q0, q1, S, b0, nb);
}

// ------------------------------------------------------------------------------------------------ K1: record keys (the filters, the duplicate drop and the stream scans are k_pass1, sq_pass_kernels.inc)
// ReadRec_t::Equal (ReadRec.cpp:119-141) on two records whose fixed fields are already in registers: the loads of both records are
// issued together and up front (rec_equal above re-reads them field by field inside its loops, one dependent round trip each)
struct RecKey { int nown, rid, mrid, mp; uint32_t bo; bool first, rev, stub; int4 e0; };  // e0: the first own block in read order
__device__ __forceinline__ RecKey rec_key(const RecView& R, int64_t r) {
    RecKey k;
    k.e0 = make_int4(0, 0, 0, 0);
    if (r < 0) { k.nown = 0; k.rid = 0; k.mrid = 0; k.mp = 0; k.bo = 0; k.first = true; k.rev = false; k.stub = false; return k; }  // the empty initial lastreadrec
    const int flag = R.flag[r];
    const uint32_t b0 = R.blk_off[r], b1 = R.blk_off[r + 1];
    k.rid = R.refid[r]; k.mrid = R.mrefid[r]; k.mp = R.mpos[r];
    k.nown = (int)(b1 - b0); k.bo = b0; k.first = flag & 0x40; k.rev = flag & 0x10; k.stub = !(flag & 0x8) && k.mrid != -1;
    if (k.nown > 0) k.e0 = R.b_pack[b0 + (k.rev ? (uint32_t)(k.nown - 1) : 0u)];
    return k;
}
__device__ __forceinline__ int key_size(const RecKey& k, int L) { const bool own = (L == 0) == k.first; return own ? k.nown : (k.stub ? 1 : 0); }
__device__ __forceinline__ void key_elem(const RecView& R, const RecKey& k, int L, int e, int& id, int& p, int& m) {
    const bool own = (L == 0) == k.first;
    if (own) { const int4 q = e == 0 ? k.e0 : R.b_pack[k.bo + (k.rev ? (uint32_t)(k.nown - 1 - e) : (uint32_t)e)]; id = k.rid; p = q.x; m = q.y; }
    else { id = k.mrid; p = k.mp; m = 15; }
}
__device__ bool key_equal(const RecView& R, const RecKey& kq, const RecKey& kr) {
    for (int swap = 0; swap < 2; ++swap) {
        const int q0 = key_size(kq, swap ? 1 : 0), q1 = key_size(kq, swap ? 0 : 1);
        if (q0 != key_size(kr, 0) || q1 != key_size(kr, 1)) continue;
        bool same = true;
        for (int L = 0; L < 2 && same; ++L) {
            const int n = key_size(kr, L);
            for (int e = 0; e < n; ++e) {
                int a0, a1, a2, b0, b1, b2;
                key_elem(R, kr, L, e, a0, a1, a2);
                key_elem(R, kq, swap ? 1 - L : L, e, b0, b1, b2);
                if (a0 != b0 || a1 != b1 || a2 != b2) { same = false; break; }
            }
        }
        if (same) return true;
    }
    return false;
}
// ------------------------------------------------------------------------------------------------ K2 support
// The discordant-cluster list is static (it depends only on the sorted chimeric blocks), so everything the
// segmentation automaton needs from the N_c-sized stream can be computed by scans over the kept records:
//   * trigger(k): first kept record that has passed cluster k (SegmentGraph.cpp:353),
//   * the "zero coverage" records (:616-620) -- the only places where a pending node end is flushed and where the
//     sliding windows are emptied (:621-636); between two of them nothing but window pushes happens unless a cluster
//     trigger falls there, so the host only replays the stretches that contain triggers,
//   * the running (otherChr, otherrightmost) pair before each such record (:655-667): with a coordinate-sorted
//     stream it is a plain 64-bit max-scan of (refid << 32 | first-block end) over concordant records,
//   * per cluster, the non-first blocks of concordant records that can span one of its break candidates
//     (the live content of the ConcordRest heap, :387-389,471-473,690-699).
struct ClusterView { int32_t n; const int32_t *chr, *start, *right; int32_t n_ref; const int32_t *bucket_off, *bucket; /* position index (k_cluster_buckets): 16 KiB stretches, the geometry of NodeView::bucket_off */ };
__device__ __forceinline__ int clusters_passed(const ClusterView& C, int refid, int pos) {  // #k with (chr_k,right_k) < (refid,pos)
    int lo = 0, hi = C.n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (C.chr[mid] < refid || (C.chr[mid] == refid && C.right[mid] < pos)) lo = mid + 1; else hi = mid; }
    return lo;
}
// ------------------------------------------------------------------------------------------------ K3: node depth
// Node at which the reference's monotone cursor (SegmentGraph.cpp:787-799 / :809-821) would first accept a
// block starting at p: normally the node containing p; a block of <= 3 bases that starts right behind a node
// boundary still "fits" the previous node(s) because of the +-3 slack.
__device__ __forceinline__ int depth_early(const NodeView& N, int c, int p, int len, int& home) {
    home = node_home(N, c, p);
    int j = home;
    if (len <= 3) {
        int lo = N.chr_start[c];
        while (j - 1 >= lo && N.pos[j - 1] + N.len[j - 1] + 3 >= p + len) --j;
    }
    return j;
}
// statistics counters: one atomic per wave, spread over NSTRIPE addresses (a single hot address serialises in L2:
// 800 k waves adding to one word cost more than the rest of the kernel); the host sums the stripes
constexpr int NSTRIPE = 256;
__device__ __forceinline__ void stripe_add(int32_t* stripes, int v) {
    atomicAdd(&stripes[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (NSTRIPE - 1)], v);
}

// add (1, len) to node `at` for every lane with `valid`; lanes of a wave that hit the same node are combined first
// (a wave covers 64 neighbouring records of the sorted stream, which almost always share their node).
// Must be called by ALL lanes of the wave (the shuffles read every lane).
// The accumulators exist NODE_STRIPES times (stripe = wave id mod NODE_STRIPES, stride nn): neighbouring waves mostly hit
// the same node, and atomics onto one word serialise in L2 (on C2 -- one chromosome, a few huge nodes -- they were most
// of the kernel); k_fold_stripes sums the copies.
constexpr int NODE_STRIPES = 16;
__device__ __forceinline__ void node_add(int32_t* cnt, int32_t* sum, int nn, bool valid, int at, int len) {
    unsigned long long active = __ballot(valid);
    const int lane = threadIdx.x & 63;
    const size_t so = (size_t)((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (NODE_STRIPES - 1)) * (size_t)nn;
    cnt += so; sum += so;
    if (!valid) { at = -1; len = 0; }
    while (active) {
        int leader = __ffsll((long long)active) - 1;
        int k = __shfl(at, leader, 64);
        unsigned long long same = __ballot(at == k) & active;
        int v = (at == k) ? len : 0;
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        if (lane == leader) { atomicAdd(&cnt[k], (int)__popcll(same)); atomicAdd(&sum[k], v); }
        active &= ~same;
    }
}
// the same through a small per-workgroup table in LDS (4 nodes per accumulator pair: the 256 neighbouring records of a workgroup
// rarely touch more); what does not fit goes to global memory directly.  k_depth flushes the table at its end: a quarter of the
// global atomics, and those were what its waves queued for (all waves in flight add to the same few nodes).
struct NodeAcc { int key[4], cnt[4], sum[4]; };
__device__ __forceinline__ void node_add_lds(NodeAcc& A, int32_t* cnt, int32_t* sum, int nn, bool valid, int at, int len) {
    unsigned long long active = __ballot(valid);
    const int lane = threadIdx.x & 63;
    if (!valid) { at = -1; len = 0; }
    while (active) {
        int leader = __ffsll((long long)active) - 1;
        int k = __shfl(at, leader, 64);
        unsigned long long same = __ballot(at == k) & active;
        int v = (at == k) ? len : 0;
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        if (lane == leader) {
            bool done = false;
            for (int q = 0; q < 4 && !done; ++q) {
                const int old = atomicCAS(&A.key[q], -1, k);
                if (old == -1 || old == k) { atomicAdd(&A.cnt[q], (int)__popcll(same)); atomicAdd(&A.sum[q], v); done = true; }
            }
            if (!done) {
                const size_t so = (size_t)((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (NODE_STRIPES - 1)) * (size_t)nn;
                atomicAdd(&cnt[so + k], (int)__popcll(same)); atomicAdd(&sum[so + k], v);
            }
        }
        active &= ~same;
    }
}
// one lane adds (n records, sum of lengths) for node k
__device__ __forceinline__ void node_add_bulk(NodeAcc& A, int32_t* cnt, int32_t* sum, int nn, int k, int n, int v) {
    for (int q = 0; q < 4; ++q) {
        const int old = atomicCAS(&A.key[q], -1, k);
        if (old == -1 || old == k) { atomicAdd(&A.cnt[q], n); atomicAdd(&A.sum[q], v); return; }
    }
    const size_t so = (size_t)((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (NODE_STRIPES - 1)) * (size_t)nn;
    atomicAdd(&cnt[so + k], n); atomicAdd(&sum[so + k], v);
}
__global__ void k_fold_stripes(int nn, const int32_t* a, const int32_t* b, const int32_t* c2, const int32_t* d, int32_t* out /* 4 x nn */) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nn) return;
    int sa = 0, sb = 0, sc = 0, sd = 0;
    for (int s = 0; s < NODE_STRIPES; ++s) { const size_t o = (size_t)s * nn + i; sa += a[o]; sb += b[o]; sc += c2[o]; sd += d[o]; }
    out[i] = sa; out[(size_t)nn + i] = sb; out[2 * (size_t)nn + i] = sc; out[3 * (size_t)nn + i] = sd;
}
// number of non-first blocks of a consumed kept record (|ReadsOther| contributions), for the ordered gather
struct FOtherCount {
    RecView R; const uint8_t* keep; const long long* r_break;  // records at or behind *r_break are not consumed (ledger B12)
    __device__ int operator()(int64_t r) const {
        if (!(keep[r] & K_1) || r >= *r_break) return 0;
        int nb = (int)(R.blk_off[r + 1] - R.blk_off[r]);
        return nb > 1 ? nb - 1 : 0;
    }
};
__global__ void k_gather_other(RecView R, const uint8_t* keep, const long long* r_break, const int32_t* off, int32_t* o_chr, int32_t* o_pos, int32_t* o_len, int32_t* flags) {
    // materialise ReadsOther in stream order (offsets from an exclusive scan); flag blocks of <= 3 bases, the only
    // ones whose node attribution can depend on the tie order of the reference's unstable sort
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R.n || !(keep[r] & K_1) || r >= *r_break) return;
    uint32_t b0 = R.blk_off[r];
    int nblk = (int)(R.blk_off[r + 1] - b0);
    for (int k = 1; k < nblk; ++k) {
        int slot = off[r] + k - 1;
        o_chr[slot] = R.refid[r]; o_pos[slot] = R.b_refpos[b0 + k]; o_len[slot] = R.b_matchref[b0 + k];
        if (R.b_matchref[b0 + k] <= 3) atomicOr(&flags[0], 64);
    }
}

// ------------------------------------------------------------------------------------------------ K4/K5: edges

// fitting range [a,b] of nodes for a block (LocateRead's +-5 test, SegmentGraph.cpp:1213) and its home node
// The nodes tile every chromosome (tile_genome: the first starts at 0, each starts where the one before ends), so both ends of
// the range lie next to the home node and are reached by short walks from it -- for a block inside its home node without a
// single further load (round 1: two bisections over the whole chromosome, two dozen dependent loads).
__device__ __forceinline__ void fit_around(const NodeView& N, int c, int p, int end, int home, const int4& nh, int& a, int& b) {
    const int lo = N.chr_start[c], hi = N.chr_start[c + 1];
    // b = last node on c with pos - 5 <= p (home qualifies; the nodes behind it start at the running end)
    b = home;
    for (int nxt = nh.y + nh.z; b + 1 < hi && nxt - 5 <= p;) { ++b; nxt += N.len[b]; }
    // a = first node on c with pos + len + 5 >= end (a == hi: none)
    if (nh.y + nh.z + 5 >= end) {
        a = home;
        for (int pe = nh.y; a - 1 >= lo && pe + 5 >= end;) { --a; pe = N.pos[a]; }  // pe = end of node a - 1 = start of node a
    } else {
        a = home + 1;
        for (int e = nh.y + nh.z; a < hi; ++a) { e += N.len[a]; if (e + 5 >= end) break; }
    }
}
__device__ __forceinline__ void fit_range(const NodeView& N, int c, int p, int end, int& a, int& b, int& home, int4& nh) {
    home = node_home(N, c, p);
    nh = N.pack[home];
    fit_around(N, c, p, end, home, nh, a, b);
}
__device__ __forceinline__ void fit_range(const NodeView& N, int c, int p, int end, int& a, int& b, int& home) {
    int4 nh;
    fit_range(N, c, p, end, a, b, home, nh);
}
// LocateRead for one block with running index i; returns node or -1 and updates i exactly like the scan loops
__device__ __forceinline__ int locate_one(const NodeView& N, int c, int p, int end, int& i, int initialguess) {
    if (i < 0 || i >= N.n) i = initialguess;
    if (c < 0 || c >= N.n_ref) {  // no node can match: the scan runs off one end of the table
        if (N.pack[i].x < c) i = N.n; else i = -1;
        return -1;
    }
    // the running node usually still fits (+-5 test of SegmentGraph.cpp:1213): no search needed
    const int4 ni = N.pack[i];  // chr, pos, len in one load
    if (ni.x == c && p >= ni.y - 5 && end <= ni.y + ni.z + 5) return i;
    int a, b, home;
    fit_range(N, c, p, end, a, b, home);
    bool nonempty = a <= b;
    bool up = ni.x < c || (ni.x == c && ni.y <= p);
    if (up) {
        if (nonempty && a >= i) { i = a; return a; }
        i = N.chr_start[c + 1];  // first node of a later chromosome, or N.n
        return -1;
    }
    if (nonempty && b <= i) { i = b; return b; }
    i = N.chr_start[c] - 1;
    return -1;
}

struct FPart { const uint8_t* keep; __device__ int operator()(int64_t i) const { return (keep[i] & K_BUILD) ? (int)i : -1; } };

// block 0 of the stub-augmented record (the block whose node becomes the next record's hint)
__device__ __forceinline__ bool rec_block0(const RecView& R, int64_t r, int& c, int& p, int& end) {
    ListRec l = list_rec(R, r);
    if (l.size(0) > 0) {
        int m; list_key(R, r, l, 0, 0, c, p, m); end = p + m; return true;
    }
    if (l.size(1) > 0) { int m; list_key(R, r, l, 1, 0, c, p, m); end = p + m; return true; }
    return false;
}
// block 0 of record r against the node table: fitting range [a, b] (a > b: none), home node, and `deep`: block 0 lies deep inside
// its home node -- then, whatever hint arrives, LocateRead puts it into `home` (from a node below the walk goes up to the first
// fitting node, from one above it comes down to the last, and `home` is the only one), so the record needs no incoming hint
__device__ __forceinline__ void block0_fit(const RecView& R, const NodeView& N, int64_t r, int& a, int& b, int& home, bool& deep) {
    int c, p, end;
    a = 1; b = 0; home = -1; deep = false;
    if (rec_block0(R, r, c, p, end) && c >= 0 && c < N.n_ref) {
        home = node_home(N, c, p);
        const int4 nh = N.pack[home];
        const int hp = nh.y, he = hp + nh.z;
        if (end > hp + 5 && p < he - 5 && end <= he + 5) { a = home; b = home; deep = true; }  // deep inside its node: no neighbour can fit
        else fit_around(N, c, p, end, home, nh, a, b);
    }
}
// hint transfer of one record: x -> node of its block 0 if located, else x (SegmentGraph.cpp:1607-1609)
__device__ __forceinline__ int hint_step(int x, int a, int b, int home) {
    if (a > b) return x;
    if (x >= a && x <= b) return x;
    if (x < a) return x <= home ? a : x;  // x in (home, a): the downward scan finds nothing (only with nodes < 5 bp)
    return b;
}

// wave-aggregated insert of (key,+w) into the global open-addressing table
constexpr uint32_t HASH_PROBES = 256;
__device__ __forceinline__ void hash_add(unsigned long long* hk, uint32_t* hv, uint32_t mask, unsigned long long key, uint32_t w, int32_t* flags) {
    uint32_t h = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 40) & mask;
    // (at most HASH_PROBES slots are tried: a table that needs more is too full and the host repeats the pass with four times the
    // slots -- without the limit every insert into a full table walked all of it, 26 s for the first pass over a dense sample)
    for (uint32_t probe = 0; probe <= mask && probe < HASH_PROBES; ++probe) {
        unsigned long long cur = hk[h];
        if (cur == key) { atomicAdd(&hv[h], w); return; }
        if (cur == ~0ull) {
            unsigned long long old = atomicCAS(&hk[h], ~0ull, key);
            if (old == ~0ull || old == key) { atomicAdd(&hv[h], w); return; }
        }
        h = (h + 1) & mask;
    }
    atomicOr(&flags[0], 4);  // table full
}
__device__ __forceinline__ void emit_edge(unsigned long long* hk, uint32_t* hv, uint32_t mask, int i, bool hi, int j, bool hj, int32_t* flags, int32_t* stripes, int nnodes) {
    if (i < 0 || j < 0 || i >= nnodes || j >= nnodes) { atomicOr(&flags[0], 8); return; }  // reference: assert(...) aborts
    int a = i, b = j; bool ha = hi, hb = hj;
    if (i > j) { a = j; ha = hj; b = i; hb = hi; }
    unsigned long long key = ((unsigned long long)(uint32_t)a << 32) | ((unsigned long long)(uint32_t)b << 2) | ((unsigned long long)ha << 1) | (unsigned long long)hb;
    // combine equal keys inside the wave before touching the table (neighbouring records emit the same edge)
    unsigned long long active = __ballot(1);
    int lane = threadIdx.x & 63;
    if (lane == __ffsll((long long)active) - 1) stripe_add(stripes, (int)__popcll(active));  // raw edge count
    while (active) {
        int leader = __ffsll((long long)active) - 1;
        unsigned long long k = __shfl(key, leader, 64);
        unsigned long long same = __ballot(key == k) & active;
        if (lane == leader) hash_add(hk, hv, mask, key, (uint32_t)__popcll(same), flags);
        if (key == k) break;
        active &= ~same;
    }
}

struct EdgeParams { int dp, di; };
__device__ __forceinline__ bool dev_edge_discordant(const NodeView& N, const EdgeParams& P, int i, bool hi, int j, bool hj) {
    int a = i, b = j; bool ha = hi, hb = hj;
    if (i > j) { a = j; ha = hj; b = i; hb = hi; }
    const int4 na = N.pack[a], nb = N.pack[b];
    if (na.x != nb.x) return true;
    if (nb.y - na.y - na.z > P.dp && b - a > P.di) return true;
    if (ha != false || hb != true) return true;
    return false;
}

struct EdgeParams2 { int dp, di, ablate; };
// Pass 1 of the edge stage: the records that cannot emit anything are recognised and dropped here, the others go on a work list
// for k_edges.  A record emits no edge, sets no flag and needs no incoming hint when block 0 of its stub-augmented form lies deep
// inside its home node h (block0_fit: every hint leads to h, which becomes the running node) and every other element -- own
// blocks and mate stub -- passes LocateRead's +-5 test against h (SegmentGraph.cpp:1213): all blocks are then located in h, so
// there is no unlocatable block (:1612-1618), no pair of consecutive blocks in different nodes (:1631-1653) and the pair rule
// sees both mates in one node (:1655-1685).  That is 99 % of the records of an RNA-seq sample.  This kernel is a plain
// scan -- fixed fields, first and last own block, one node lookup -- in 40-odd registers, where the full rule set below
// needs 90 and runs five waves per SIMD.
// It also leaves, per workgroup of 256 records, what the breakpoint-support kernel of sq_call_sv would otherwise have to scan the stream
// for: the largest (chromosome, fragment start) among its pass-3 records (bp_block_key; SegmentGraph.cpp:3147-3158 -- the cursor's bound
// in front of a record is a monotone function of the largest such pair seen so far).  The fields are here anyway, except pos and class.
__device__ __forceinline__ unsigned long long bp_record_key(uint8_t cls, int flag, int rid, int p, int mrid, int mp) {  // 0: not a pass-3 record
    if (!(cls & C_P3)) return 0ull;
    const int st = (!(flag & 0x8) && mrid == rid) ? mp : p;  // SegmentGraph.cpp:3147-3150
    return (((unsigned long long)(uint32_t)rid << 32) | (uint32_t)((uint32_t)st ^ 0x80000000u)) + 1ull;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    for (int d = 32; d >= 1; d >>= 1) { const unsigned long long o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
    return v;
}
__global__ __launch_bounds__(256) void k_edges_near(RecView R, NodeView N, const uint8_t* keep, unsigned long long* bp_block_key, uint32_t* list, int32_t* count, int all /* SQUID_EDGES_ALL: clear nothing, every record takes the full rule set (cross-check) */) {
    __shared__ unsigned long long s_key[4];
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool need = false;
    unsigned long long bkey = 0;
    if (r < R.n) {
        const uint8_t kp = keep[r];  // (the loads below do not wait for it: one memory round trip)
        const uint8_t cl = (kp & K_P3) ? C_P3 : 0;
        const int flag = R.flag[r], rid = R.refid[r], mrid = R.mrefid[r], mp = R.mpos[r], rpos = R.pos[r];
        const uint32_t bo = R.blk_off[r], bo1 = R.blk_off[r + 1];
        bkey = bp_record_key(cl, flag, rid, rpos, mrid, mp);
        if (kp & K_BUILD) {
            const int nown = (int)(bo1 - bo);
            const bool first = flag & 0x40, rev = flag & 0x10, stub = !(flag & 0x8) && mrid != -1;
            if (nown + (stub ? 1 : 0) > 0) {
                need = true;
                int4 qa = make_int4(0, 0, 0, 0), qb = qa;  // first and last own block in CIGAR order
                if (nown > 0) { qa = R.b_pack[bo]; qb = R.b_pack[bo1 - 1]; }
                const bool own0 = first ? nown > 0 : !stub;  // is block 0 an own block?  (rec_block0)
                const int4 q0 = rev ? qb : qa;               // the first own block in read-offset order
                const int c0 = own0 ? rid : mrid, p0 = own0 ? q0.x : mp, e0 = own0 ? q0.x + q0.y : mp + 15;
                if (c0 >= 0 && c0 < N.n_ref) {
                    const int4 nh = N.pack[node_home(N, c0, p0)];
                    const int hp = nh.y, he = hp + nh.z;
                    auto fits = [&](int c, int p, int end) { return c == nh.x && p >= hp - 5 && end <= he + 5; };
                    if (!all && e0 > hp + 5 && p0 < he - 5 && e0 <= he + 5) {  // deep inside its node
                        bool ok = !stub || fits(mrid, mp, mp + 15);
                        if (nown > 0) ok = ok && fits(rid, qa.x, qa.x + qa.y) && fits(rid, qb.x, qb.x + qb.y);
                        for (int k = 1; ok && k + 1 < nown; ++k) { const int4 q = R.b_pack[bo + (uint32_t)k]; ok = fits(rid, q.x, q.x + q.y); }
                        need = !ok;
                    }
                }
            }
        }
    }
    const int lane = threadIdx.x & 63;
    bkey = wave_max_u64(bkey);
    if (lane == 0) s_key[threadIdx.x >> 6] = bkey;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long k = s_key[0]; for (int w = 1; w < 4; ++w) k = s_key[w] > k ? s_key[w] : k; bp_block_key[blockIdx.x] = k; }
    const unsigned long long m = __ballot(need);
    if (!m) return;
    int base = 0;
    if (lane == __ffsll((long long)m) - 1) base = atomicAdd(count, (int)__popcll(m));
    base = __shfl(base, __ffsll((long long)m) - 1, 64);
    if (need) list[base + (int)__popcll(m & ((1ull << lane) - 1))] = (uint32_t)r;
}
// the same keys by themselves (breakpoint support asked for without an edge stage over these records in front of it)
__global__ __launch_bounds__(256) void k_bp_keys(RecView R, const uint8_t* cls, unsigned long long* bp_block_key) {
    __shared__ unsigned long long s_key[4];
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long bkey = 0;
    if (r < R.n) bkey = bp_record_key(cls[r], R.flag[r], R.refid[r], R.pos[r], R.mrefid[r], R.mpos[r]);
    bkey = wave_max_u64(bkey);
    if ((threadIdx.x & 63) == 0) s_key[threadIdx.x >> 6] = bkey;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long k = s_key[0]; for (int w = 1; w < 4; ++w) k = s_key[w] > k ? s_key[w] : k; bp_block_key[blockIdx.x] = k; }
}
// front[t] = largest key in front of tile t of the breakpoint kernel (a tile = `per_tile` workgroups of 256 records).  Two small
// launches over the keys, 1024 per workgroup, coalesced: the maximum of every 1024, then per workgroup the maximum of the groups in
// front of it (a few hundred values) and an exclusive max-scan of its own 1024.
__global__ __launch_bounds__(1024) void k_bp_key_reduce(const unsigned long long* key, int nkeys, unsigned long long* part) {
    __shared__ unsigned long long s_w[16];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    unsigned long long v = wave_max_u64(i < nkeys ? key[i] : 0ull);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long m = 0; for (int w = 0; w < 16; ++w) m = s_w[w] > m ? s_w[w] : m; part[blockIdx.x] = m; }
}
__global__ __launch_bounds__(1024) void k_bp_key_scan(const unsigned long long* key, int nkeys, const unsigned long long* part, int per_tile, unsigned long long* front, int ntiles) {
    __shared__ unsigned long long s_w[16];
    __shared__ unsigned long long s_before;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int i = blockIdx.x * 1024 + t;
    if (wave == 0) {  // everything in front of this workgroup
        unsigned long long m = 0;
        for (int b = lane; b < (int)blockIdx.x; b += 64) { const unsigned long long p = part[b]; m = p > m ? p : m; }
        m = wave_max_u64(m);
        if (lane == 0) s_before = m;
    }
    const unsigned long long own = i < nkeys ? key[i] : 0ull;
    unsigned long long x = own;  // inclusive max-scan inside the wave
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long y = __shfl_up(x, d, 64); if (lane >= d && y > x) x = y; }
    if (lane == 63) s_w[wave] = x;
    __syncthreads();
    unsigned long long run = s_before;
    for (int w = 0; w < wave; ++w) run = s_w[w] > run ? s_w[w] : run;
    unsigned long long e = __shfl_up(x, 1, 64);  // exclusive
    if (lane == 0) e = 0;
    e = e > run ? e : run;
    if (i < nkeys && i % per_tile == 0 && i / per_tile < ntiles) front[i / per_tile] = e;
}
// Pass 2: the full rule set for one record of the work list
__device__ __forceinline__ void edges_record(const RecView& R, const NodeView& N, const EdgeParams2& P2, const uint8_t* keep, int64_t r, unsigned long long* hk, uint32_t* hv, uint32_t hmask, int32_t* flags, int32_t* stripes) {
    const EdgeParams P{P2.dp, P2.di};
    // ---- everything the record itself needs is loaded up front, the loads independent of each other (one memory round trip
    // instead of a chain of them) and not waiting for the keep byte: the fixed fields and both block offsets; behind them, again
    // together, the first two own blocks in read-offset order and the index geometry of the record's chromosome
    const uint8_t kp = keep[r];
    const int flag = R.flag[r], rid = R.refid[r], mrid = R.mrefid[r], mp = R.mpos[r];
    const uint32_t bo = R.blk_off[r], bo1 = R.blk_off[r + 1];
    if (!(kp & K_BUILD)) return;
    if (P2.ablate & 4) { if (flag == 0xffff) flags[1] = 1; return; }
    // ---- incoming hint (SegmentGraph.cpp:1607-1609: the node of the previous record's block 0).  Most records never need it: when
    // block 0 lies deep inside its node every incoming hint leads to that node (block0_fit), and that node is where the running
    // index starts.  Only the few that really consult the hint -- block 0 near a node boundary, an unlocatable block that leaves
    // the running index outside the table, more than OWNCAP blocks -- walk back through the participating records (keep[] &
    // K_BUILD) to one whose block 0 pins the hint, then replay forward; their block-0 fits are recomputed on the way (no
    // per-record arrays, no separate pass).
    auto prev_part = [&](int64_t q) { --q; while (q >= 0 && !(keep[q] & K_BUILD)) --q; return q; };
    auto next_part = [&](int64_t q) { ++q; while (q < R.n && !(keep[q] & K_BUILD)) ++q; return q; };
    auto incoming_hint = [&]() -> int {
        int h = 0, a, b, home;
        bool deep;
        int64_t q = prev_part(r), anchor = -1;
        while (q >= 0) {
            block0_fit(R, N, q, a, b, home, deep);
            if (a == b) { anchor = q; h = a; break; }
            q = prev_part(q);
        }
        // replay the un-pinned records between the anchor (or the start of the stream, hint 0) and r
        for (int64_t t = next_part(anchor); t < r; t = next_part(t)) { block0_fit(R, N, t, a, b, home, deep); h = hint_step(h, a, b, home); }
        return h;
    };
    ListRec l;
    l.nown = (int)(bo1 - bo); l.first = flag & 0x40; l.rev = flag & 0x10; l.stub = !(flag & 0x8) && mrid != -1;
    const int nown = l.nown, nt = nown + (l.stub ? 1 : 0);
    int4 q0 = make_int4(0, 0, 0, 0);
    int4 q1 = q0;  // and the second one (spliced reads: nearly half of the records)
    if (nown > 0) q0 = R.b_pack[bo + (l.rev ? (uint32_t)(nown - 1) : 0u)];
    if (nown > 1) q1 = R.b_pack[bo + (l.rev ? (uint32_t)(nown - 2) : 1u)];
    int geo_c = -1, geo0 = 0, geo1 = 0;  // fine_off[geo_c], fine_off[geo_c + 1]
    if (rid >= 0 && rid < N.n_ref) { geo_c = rid; geo0 = N.fine_off[rid]; geo1 = N.fine_off[rid + 1]; }
    auto home_of = [&](int c, int p) { return c == geo_c ? node_home_geo(N, geo0, geo1, p) : node_home(N, c, p); };
    if (nt == 0) return;
    const bool own_first = l.first;            // own blocks form list F (FirstRead) iff first-mate
    // block 0 of the stub-augmented record (rec_block0): the first element of list F if it has one, else of list S
    int c0, p0, e0;
    {
        const bool own0 = own_first ? nown > 0 : !l.stub;  // is block 0 an own block?  (F = own blocks iff first-mate; the other list is [stub])
        if (own0) { c0 = rid; p0 = q0.x; e0 = q0.x + q0.y; } else { c0 = mrid; p0 = mp; e0 = mp + 15; }
    }
    int a0 = 1, b0 = 0, home0 = -1;
    bool deep0 = false;
    int ci = -1;                         // one-entry cache of the node table: most records stay inside one node
    int4 cp = make_int4(0, 0, 0, 0);
    if (c0 >= 0 && c0 < N.n_ref) {
        home0 = home_of(c0, p0);
        cp = N.pack[home0]; ci = home0;
        const int hp = cp.y, he = hp + cp.z;
        if (e0 > hp + 5 && p0 < he - 5 && e0 <= he + 5) { a0 = home0; b0 = home0; deep0 = true; }
        else fit_around(N, c0, p0, e0, home0, cp, a0, b0);
    }
    (void)a0; (void)b0;
    bool hint_known = !deep0;
    int hint = deep0 ? home0 : incoming_hint();
    if (P2.ablate & 2) { if (hint == -12345) flags[1] = 1; return; }
    auto node_pack = [&](int i) -> int4 { if (i != ci) { cp = N.pack[i]; ci = i; } return cp; };
    // LocateRead for one block with running index i (locate_one), the running node's fields through the cache
    auto locate = [&](int c, int p, int end, int& i) -> int {
        if (i < 0 || i >= N.n) i = hint;
        const int4 ni = node_pack(i);
        if (c < 0 || c >= N.n_ref) { if (ni.x < c) i = N.n; else i = -1; return -1; }
        if (ni.x == c && p >= ni.y - 5 && end <= ni.y + ni.z + 5) return i;  // the running node still fits (+-5 test of SegmentGraph.cpp:1213)
        int a, b;
        const int home = home_of(c, p);
        const int4 nh = N.pack[home];
        fit_around(N, c, p, end, home, nh, a, b);
        ci = home; cp = nh;  // the block's home node is nearly always the answer and the next running node
        const bool nonempty = a <= b;
        const bool up = ni.x < c || (ni.x == c && ni.y <= p);
        if (up) {
            if (nonempty && a >= i) { i = a; return a; }
            i = N.chr_start[c + 1];
            return -1;
        }
        if (nonempty && b <= i) { i = b; return b; }
        i = N.chr_start[c] - 1;
        return -1;
    };
    // ---- the stub-augmented, read-offset-sorted record is streamed block by block (own blocks first for a first-mate
    // record, the 15-base mate stub first otherwise); everything LocateRead + the edge rules need is carried in
    // registers: no per-thread arrays, no scratch
    constexpr int OWNCAP = 8;
    int ownnode[OWNCAP];
#pragma unroll
    for (int k = 0; k < OWNCAP; ++k) ownnode[k] = -2;
    int i = hint, node0 = -1, ffi = hint;
    int4 qc = q0, qn = q1;
    // trimmed data of: first own block, last own block, previous own block, stub
    int of_c = 0, of_p = 0, of_rp = 0; bool of_rev = false;
    int ol_p = 0, ol_rp = 0, ol_mr = 0, ol_node = -1; bool ol_rev = false;
    int pv_c = 0, pv_p = 0, pv_rp = 0, pv_node = -1; bool pv_rev = false;
    int st_c = 0, st_p = 0, st_rp = 0, st_mr = 0, st_node = -1; bool st_rev = false;
    bool own_enddisc = false;
    for (int k = 0; k < nt; ++k) {
        const bool is_stub = l.stub && (own_first ? k == nown : k == 0);
        const int ko = own_first ? k : k - (l.stub ? 1 : 0);  // index among own blocks
        int bc, bp, bm, brp, bmr; bool brev;
        if (is_stub) { bc = mrid; bp = mp; bm = 15; brp = 0; bmr = 15; brev = flag & 0x20; }
        else {
            const int4 q = qc;  // own blocks arrive one iteration ahead of their use
            qc = qn;
            if (ko + 2 < nown) qn = R.b_pack[bo + (l.rev ? (uint32_t)(nown - 3 - ko) : (uint32_t)(ko + 2))];
            bc = rid; bp = q.x; bm = q.y; brp = (int)((uint32_t)q.z & 0xffffu); bmr = (int)((uint32_t)q.z >> 16); brev = l.rev;
        }
        if ((i < 0 || i >= N.n) && !hint_known) { hint = incoming_hint(); hint_known = true; }  // (the walk falls back to the incoming hint)
        const int nd = locate(bc, bp, bp + bm, i);
        if (nd >= 0) {  // trim to the node (SegmentGraph.cpp:1229-1248)
            const int4 nn = node_pack(nd);
            int np = nn.y, ne = np + nn.z;
            if (bp < np) { int d = np - bp; if (!brev) brp += d; bm -= d; bmr -= d; bp = np; }
            if (bp + bm > ne) { int d = bp + bm - ne; if (brev) brp += d; bm -= d; bmr -= d; }
        } else {
            // boundary edge of an unlocatable block (SegmentGraph.cpp:1612-1618).  The reference scans up from the
            // refreshed hint while node.end < block.start, then down while node.start > block.start: a block that
            // starts exactly on a node boundary lands in the node BELOW the boundary when the scan arrives from below.
            if (bc < 0 || bc >= N.n_ref) atomicOr(&flags[0], 8);
            else {
                int h = node_home(N, bc, bp);
                if (N.pos[h] == bp && h > N.chr_start[bc] && ffi <= h - 1) --h;
                if (!(P2.ablate & 1)) emit_edge(hk, hv, hmask, h, false, h + 1, true, flags, stripes, N.n);
            }
        }
        if (k == 0) { node0 = nd; ffi = nd != -1 ? nd : hint; }
        if (is_stub) { st_c = bc; st_p = bp; st_rp = brp; st_mr = bmr; st_node = nd; st_rev = brev; }
        else {
            if (ko == 0) { of_c = bc; of_p = bp; of_rp = brp; of_rev = brev; }
            else {
                // consecutive blocks of one mate in different nodes (SegmentGraph.cpp:1631-1653)
                if (pv_node != nd && pv_node != -1 && nd != -1) if (!(P2.ablate & 1)) emit_edge(hk, hv, hmask, pv_node, pv_rev, nd, !brev, flags, stripes, N.n);
                // IsEndDiscordant on the trimmed blocks (ReadRec.cpp:178-209)
                if (pv_c != bc || pv_rev != brev) own_enddisc = true;
                else {
                    bool refup = pv_p < bp, readup = pv_rp < brp;
                    if (!pv_rev && refup != readup) own_enddisc = true;
                    if (pv_rev && refup == readup) own_enddisc = true;
                }
            }
#pragma unroll
            for (int q = 0; q < OWNCAP; ++q) if (q == ko) ownnode[q] = nd;
            pv_c = bc; pv_p = bp; pv_rp = brp; pv_node = nd; pv_rev = brev;
            ol_p = bp; ol_rp = brp; ol_mr = bmr; ol_node = nd; ol_rev = brev;
        }
    }
    (void)node0;
    // ---- pair edge, first-mate records only: F = own blocks, S = [stub] (SegmentGraph.cpp:1655-1685)
    if (own_first && nown > 0 && l.stub && !own_enddisc) {
        const int a = ol_node, b = st_node;
        bool isoverlap = (a == b);  // "i equals the node of a SecondMate block"
        if (nown <= OWNCAP) {
#pragma unroll
            for (int q = 0; q < OWNCAP; ++q) if (q < nown && ownnode[q] == b) isoverlap = true;
        } else {
            // more own blocks than the register file keeps: walk the own list again (same hint chain, same nodes)
            if (!hint_known) { hint = incoming_hint(); hint_known = true; }
            int i2 = hint;
            for (int ko = 0; ko < nown; ++ko) {
                DBlk bb = own_block_sorted(R, r, ko, nown, l.rev);
                if (locate_one(N, bb.refid, bb.refpos, bb.refpos + bb.matchref, i2, hint) == b) isoverlap = true;
            }
        }
        int ad = a - b; if (ad < 0) ad = -ad;
        if (nown > 1 && ad < 3) isoverlap = true;
        if (a != b && a != -1 && b != -1 && !isoverlap) {
            // IsPairDiscordant(false) on the trimmed record; the absent second mate has TotalLen 0 (ReadRec.cpp:211-228)
            const int ftot = (int)R.totlen[r], stot = 0;
            bool pd;
            if (of_c != st_c || of_rev == st_rev) pd = true;
            else if (!of_rev && of_p - of_rp > st_p - (stot - st_rp - st_mr)) pd = true;
            else if (!st_rev && st_p - st_rp > ol_p - (ftot - ol_rp - ol_mr)) pd = true;
            else pd = false;
            if (pd == dev_edge_discordant(N, P, a, ol_rev, b, st_rev)) if (!(P2.ablate & 1)) emit_edge(hk, hv, hmask, a, ol_rev, b, st_rev, flags, stripes, N.n);
        }
    }
}
__global__ void k_edges(RecView R, NodeView N, EdgeParams2 P2, const uint8_t* keep, const uint32_t* list, const int32_t* count, unsigned long long* hk, uint32_t* hv, uint32_t hmask, int32_t* flags, int32_t* stripes) {
    const int n = *count;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
        if (__hip_atomic_load(&flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4) return;  // table too small: the pass is repeated anyway
        edges_record(R, N, P2, keep, (int64_t)list[idx], hk, hv, hmask, flags, stripes);
    }
}


__global__ void k_hash_compact(const unsigned long long* hk, const uint32_t* hv, uint32_t slots, int32_t* counter, unsigned long long* okey, uint32_t* oval) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= slots) return;
    unsigned long long k = hk[i];
    if (k == ~0ull) return;
    int s = atomicAdd(counter, 1);
    okey[s] = k; oval[s] = hv[i];
}

// ------------------------------------------------------------------------------------------------ K10: BP support
// Sorted breakpoint list (chr,pos); the reference walks it with a cursor that advances by at most one entry per
// pass-3 record (SegmentGraph.cpp:3157-3158).  m(r) = number of breakpoints record r is "beyond"; the cursor after r
// is cur(r) = cur(r-1) + [cur(r-1) < m(r)], so cur <= M = prefix-max(m), and once cur == M it stays equal to M until
// the next record where M grows (an "event").  k_bp_count assumes cur == M everywhere; every event gets a wave
// (k_bp_walk) that replays the true recurrence from the event until the cursor has caught up and corrects the
// records it passed.  An event whose start lies inside the walk of an earlier valid one is absorbed by it
// (k_bp_chain decides that in stream order; events are indexed by their M value, which grows with the stream).
// `bucket` is the coarse position index of the node table's geometry (NodeView::bucket_off): the answer for the first
// base of every 16 KiB stretch, so a query is one load and a short walk
struct BPView { int32_t n; const int32_t *chr, *pos; int dp; const int32_t *bucket, *bucket_off; int32_t n_ref; };
__device__ __forceinline__ int bp_lower_bound_search(const BPView& B, int c, int p) {  // first j with (chr,pos) >= (c,p)
    int lo = 0, hi = B.n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (B.chr[mid] < c || (B.chr[mid] == c && B.pos[mid] < p)) lo = mid + 1; else hi = mid; }
    return lo;
}
__device__ __forceinline__ int bp_lower_bound(const BPView& B, int c, int p) {
    const int first = B.bucket_off[c], last = B.bucket_off[c + 1] - 1;
    int b = first + (p <= 0 ? 0 : (p >> NODE_BUCKET_SHIFT));
    if (b > last) b = last;
    int j = B.bucket[b];
    while (j < B.n && B.chr[j] == c && B.pos[j] < p) ++j;
    return j;
}
__global__ void k_bp_buckets(BPView B, int n_ref, int total, int32_t* bucket) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    int lo = 0, hi = n_ref;  // chromosome of this bucket: last c with bucket_off[c] <= g
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (B.bucket_off[mid] <= g) lo = mid; else hi = mid; }
    bucket[g] = bp_lower_bound_search(B, lo, (g - B.bucket_off[lo]) << NODE_BUCKET_SHIFT);
}
__device__ __forceinline__ int bp_start(const RecView& R, int64_t r) {  // SegmentGraph.cpp:3147-3150
    int c = R.refid[r], st = R.pos[r];
    if (!(R.flag[r] & 0x8) && R.mrefid[r] == c) st = R.mpos[r];
    return st;
}
// +-1 into the difference array for breakpoints [max(first covered, cursor), first not covered).  Neighbouring records
// cover the same breakpoints, so equal indices inside the wave are combined before the atomic.  WAVE: called by all
// lanes of a wave (k_bp_count); otherwise by single lanes (k_bp_walk corrections).
__device__ __forceinline__ void wave_add(int32_t* arr, bool valid, int idx, int val) {
    unsigned long long active = __ballot(valid);
    const int lane = threadIdx.x & 63;
    if (!valid) { idx = -1; val = 0; }
    while (active) {
        const int leader = __ffsll((long long)active) - 1;
        const int k = __shfl(idx, leader, 64);
        const unsigned long long same = __ballot(idx == k) & active;
        int v = idx == k ? val : 0;
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        if (lane == leader && v) atomicAdd(&arr[k], v);
        active &= ~same;
    }
}
template <bool WAVE>
__device__ __forceinline__ void bp_contribute(const RecView& R, const BPView& B, int64_t r, bool live, int before, int cur, int sign, int32_t* diff) {
    int lo = 0, hi = 0;
    if (live && before < B.n) {  // (else: the reference has left its loop, SegmentGraph.cpp:3144-3145)
        const int c = R.refid[r];
        lo = bp_lower_bound(B, c, bp_start(R, r)); hi = bp_lower_bound(B, c, R.endpos[r]);
        if (lo < cur) lo = cur;
    }
    const bool on = lo < hi;
    if (WAVE) { wave_add(diff, on, lo, sign); wave_add(diff, on, hi, -sign); }
    else if (on) { atomicAdd(&diff[lo], sign); atomicAdd(&diff[hi], -sign); }
}
// sharded runs: number of pass-3 records, and how many of them come before the first event (they can absorb a cursor
// that the earlier shards have left behind schedule without changing anything here)
__global__ void k_bp_boundary(const uint8_t* cls, int64_t n, const int32_t* ev_by_M, unsigned long long* out) {
    const int64_t first = ev_by_M[0] < 0 ? n : ev_by_M[0];
    unsigned long long all = 0, before = 0;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x)
        if (cls[r] & C_P3) { ++all; if (r < first) ++before; }
    for (int d = 32; d >= 1; d >>= 1) { all += __shfl_xor(all, d, 64); before += __shfl_xor(before, d, 64); }
    if ((threadIdx.x & 63) == 0 && all) { atomicAdd(&out[0], all); if (before) atomicAdd(&out[1], before); }
}
__global__ __launch_bounds__(64) void k_bp_chain(int nb, const int32_t* ev_by_M, const int32_t* end_by_M, int32_t* valid) {
    const int lane = threadIdx.x;
    int reach = -1;  // last record covered by a valid walk
    for (int base = 1; base <= nb; base += 64) {
        const int v = base + lane;
        const int ev = v <= nb ? ev_by_M[v] : -1;
        const int en = ev >= 0 ? end_by_M[v] : -1;
        int ok = 0;
        unsigned long long todo = __ballot(ev >= 0);
        while (todo) {
            const int k = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int e = __builtin_amdgcn_readlane(ev, k);
            if (e > reach) { reach = __builtin_amdgcn_readlane(en, k); if (lane == k) ok = 1; }
        }
        if (v <= nb) valid[v] = ok;
    }
}

#include "sq_pass_kernels.inc"

// ------------------------------------------------------------------------------------------------ K8: components
__global__ void k_cc_init(int n, int32_t* parent) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) parent[i] = i;
}
__device__ __forceinline__ int cc_find(int32_t* parent, int x) {
    while (true) {
        int p = parent[x];
        if (p == x) return x;
        int g = parent[p];
        if (g != p) parent[x] = g;  // path halving
        x = p;
    }
}
__global__ void k_cc_union(int m, const int32_t* ea, const int32_t* eb, int32_t* parent) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m) return;
    int a = ea[e], b = eb[e];
    while (true) {  // hook the larger root under the smaller: the root of a component ends up its smallest node id
        a = cc_find(parent, a);
        b = cc_find(parent, b);
        if (a == b) return;
        if (a > b) { int t = a; a = b; b = t; }
        if (atomicCAS(&parent[b], b, a) == b) return;
    }
}
__global__ void k_cc_flatten(int n, int32_t* parent, int32_t* isroot) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int r = i;
    while (parent[r] != r) r = parent[r];
    parent[i] = r;
    isroot[i] = (r == i) ? 1 : 0;
}
struct FArr { const int32_t* a; __device__ int operator()(int64_t i) const { return a[i]; } };
__global__ void k_cc_label(int n, const int32_t* parent, const int32_t* rootrank, int32_t* label) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) label[i] = rootrank[parent[i]];
}

// ------------------------------------------------------------------------------------------------ K9: small orderings
// One component per workgroup, one orientation mask per thread (n <= 8 => <= 256 masks).  For its mask a thread
// collects the "x must precede y" arcs of the compatible edges as an 8x8 bit matrix in one 64-bit register and runs
// Kahn's algorithm on it with bit operations (smallest index first).  If the arcs are acyclic every compatible edge
// is satisfied and the value is their weight sum `ub`; a cyclic orientation is worth strictly less than its `ub`, so
// only those whose `ub` beats the best acyclic value of the block go through the linear-ordering subset DP (in LDS,
// 32 at a time -- almost never).  The block then picks the canonical optimum: max value, then smallest mask, then
// the lexicographically smallest sequence.  No per-thread arrays: nothing spills to scratch.
constexpr int ORD_NMAX = 8;
struct OrdEdge { int u, v, w; bool hu, hv; };
__device__ __forceinline__ OrdEdge ord_edge(const int32_t* edges5, int e) {
    const int32_t* q = edges5 + 5 * (size_t)e;
    return OrdEdge{q[0], q[1], q[4], q[2] != 0, q[3] != 0};
}
// is the edge compatible with the orientation, and if so which end comes first (EdgeSatisfied, SegmentGraph.cpp:3763-3983)
__device__ __forceinline__ bool ord_arc(const OrdEdge& e, int mask, int& from, int& to) {
    const bool yu = !((mask >> e.u) & 1), yv = !((mask >> e.v) & 1);
    bool compat, ufirst;
    if (!e.hu && e.hv) { compat = yu == yv; ufirst = yu; }
    else if (!e.hu && !e.hv) { compat = yu != yv; ufirst = yu; }
    else if (e.hu && e.hv) { compat = yu != yv; ufirst = yv; }
    else { compat = yu == yv; ufirst = !yu; }
    from = ufirst ? e.u : e.v; to = ufirst ? e.v : e.u;
    return compat;
}
__device__ __forceinline__ long long block_max_ll(long long v, long long* lds) {
    for (int d = 32; d >= 1; d >>= 1) { long long o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    long long r = lds[0];
    for (int i = 1; i < 4; ++i) r = lds[i] > r ? lds[i] : r;
    __syncthreads();
    return r;
}
// subset DP of one orientation in LDS: h[S] = best weight still obtainable once the nodes of S are placed
__device__ void ord_dp(const SmallProblem& pr, const int32_t* edges5, int mask, int* arc, int* h, int& value, int* order) {
    const int n = pr.n, nm = 1 << n;
    for (int i = 0; i < 64; ++i) arc[i] = 0;
    for (int e = 0; e < pr.ecount; ++e) {
        OrdEdge ed = ord_edge(edges5, pr.eoff + e);
        int from, to;
        if (ord_arc(ed, mask, from, to)) arc[from * 8 + to] += ed.w;
    }
    h[nm - 1] = 0;
    for (int S = nm - 2; S >= 0; --S) {
        int best = -1;
        for (int v = 0; v < n; ++v) {
            if ((S >> v) & 1) continue;
            int gain = 0;
            for (int u = 0; u < n; ++u) if ((S >> u) & 1) gain += arc[u * 8 + v];
            int t = gain + h[S | (1 << v)];
            if (t > best) best = t;
        }
        h[S] = best;
    }
    value = h[0];
    if (order) {  // lexicographically smallest optimal sequence
        int S = 0;
        for (int p = 0; p < n; ++p)
            for (int v = 0; v < n; ++v) {
                if ((S >> v) & 1) continue;
                int gain = 0;
                for (int u = 0; u < n; ++u) if ((S >> u) & 1) gain += arc[u * 8 + v];
                if (gain + h[S | (1 << v)] == h[S]) { order[p] = v; S |= 1 << v; break; }
            }
    }
}
// the same DP by one wave: h[S] for all subsets of one size at a time (they only need the next larger size), the gain of
// appending v to S read from two half tables (low / high four nodes of S) instead of a sum over S
struct OrdWaveLds { int arc[64]; int half[2][8][16]; int h[256]; };
// LDS traffic of one wave is processed in issue order; this only keeps the compiler from reordering across the point
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
__device__ int ord_dp_wave(const SmallProblem& pr, const int32_t* edges5, int mask, OrdWaveLds& L) {
    const int n = pr.n, nm = 1 << n, lane = threadIdx.x & 63;
    L.arc[lane] = 0;
    wave_sync();
    for (int e = lane; e < pr.ecount; e += 64) {
        OrdEdge ed = ord_edge(edges5, pr.eoff + e);
        int from, to;
        if (ord_arc(ed, mask, from, to)) atomicAdd(&L.arc[from * 8 + to], ed.w);
    }
    wave_sync();
    for (int t = lane; t < 256; t += 64) {  // half[hi][v][x] = sum of arc[u][v] over the nodes u of the half-subset x
        const int hi = t >> 7, v = (t >> 4) & 7, x = t & 15;
        int g = 0;
        for (int u = 0; u < 4; ++u) if ((x >> u) & 1) g += L.arc[(u + 4 * hi) * 8 + v];
        L.half[hi][v][x] = g;
    }
    if (lane == 0) L.h[nm - 1] = 0;
    wave_sync();
    for (int k = n - 1; k >= 0; --k) {
        for (int S = lane; S < nm; S += 64) {
            if (__popc(S) != k) continue;
            int best = -1;
            for (int v = 0; v < n; ++v) {
                if ((S >> v) & 1) continue;
                const int t = L.half[0][v][S & 15] + L.half[1][v][S >> 4] + L.h[S | (1 << v)];
                best = t > best ? t : best;
            }
            L.h[S] = best;
        }
        wave_sync();
    }
    return L.h[0];
}
__global__ __launch_bounds__(256) void k_order_small(const SmallProblem* probs, const int32_t* edges5, int32_t* out_mask, int32_t* out_order, int32_t* out_value) {
    __shared__ long long s_red[4];
    __shared__ OrdWaveLds s_wave[4];
    __shared__ int s_h[256];     // single-thread DP of the winner (traceback)
    __shared__ int s_arc[64];
    __shared__ int s_ub[256], s_cmask[256], s_cub[256], s_val[256];
    const SmallProblem pr = probs[blockIdx.x];
    const int n = pr.n, nm = 1 << n, mask = threadIdx.x, wave = threadIdx.x >> 6;
    const bool live = mask < nm;
    int ub = 0, value = -1;
    unsigned long long inm = 0;  // bit 8*y + x: x must precede y
    bool acyclic = false;
    unsigned packed = 0;         // Kahn order, 3 bits per position
    if (live) {
        for (int e = 0; e < pr.ecount; ++e) {
            OrdEdge ed = ord_edge(edges5, pr.eoff + e);
            int from, to;
            if (ord_arc(ed, mask, from, to)) { ub += ed.w; inm |= 1ull << (8 * to + from); }
        }
        unsigned remaining = (unsigned)nm - 1;
        acyclic = true;
        for (int p = 0; p < n; ++p) {
            int v = -1;
            for (int cnd = 0; cnd < n; ++cnd)
                if (((remaining >> cnd) & 1) && !((unsigned)(inm >> (8 * cnd)) & remaining & 0xffu)) { v = cnd; break; }
            if (v < 0) { acyclic = false; break; }
            remaining &= ~(1u << v);
            packed |= (unsigned)v << (3 * p);
        }
        if (acyclic) value = ub;
    }
    int best = (int)block_max_ll(value, s_red);  // best acyclic value of the block
    // cyclic orientations that could still win, most promising first, one per wave at a time; the bar rises as they finish
    const bool cand = live && !acyclic && ub > best;
    s_ub[threadIdx.x] = cand ? ub : -1;
    s_val[threadIdx.x] = -1;
    __syncthreads();
    int rank = -1;
    if (cand) {
        rank = 0;
        for (int j = 0; j < 256; ++j) { int o = s_ub[j]; rank += (o > ub || (o == ub && j < (int)threadIdx.x)) ? 1 : 0; }
        s_cmask[rank] = mask; s_cub[rank] = ub;
    }
    const int ncand = (int)block_max_ll(rank, s_red) + 1;  // (also publishes s_cmask / s_cub)
    for (int base = 0; base < ncand; base += 4) {
        if (s_cub[base] <= best) break;  // sorted by bound: nothing behind can win (a cyclic orientation is worth less than its bound)
        const int mine = base + wave;
        if (mine < ncand && s_cub[mine] > best) {
            const int v = ord_dp_wave(pr, edges5, s_cmask[mine], s_wave[wave]);
            if ((threadIdx.x & 63) == 0) s_val[mine] = v;
        }
        __syncthreads();
        for (int w = 0; w < 4; ++w) if (base + w < ncand && s_val[base + w] > best) best = s_val[base + w];
        __syncthreads();
    }
    if (cand) value = s_val[rank];
    // canonical winner: max value, then smallest mask
    const long long key = live ? (((long long)value << 16) | (long long)(0xffff - mask)) : -1;
    const long long win = block_max_ll(key, s_red);
    if (live && key == win) {
        int order[ORD_NMAX];
        if (acyclic) for (int p = 0; p < n; ++p) order[p] = (packed >> (3 * p)) & 7;
        else ord_dp(pr, edges5, mask, s_arc, s_h, value, order);
        out_mask[blockIdx.x] = mask;
        out_value[blockIdx.x] = value;
        for (int p = 0; p < n; ++p) out_order[(size_t)blockIdx.x * ORD_NMAX + p] = order[p];
    }
}

// ------------------------------------------------------------------------------------------------ K9b: mid-size orderings
// Components of 9..19 nodes (MincutRecursion's n < 20 branch, SegmentGraph.cpp:3271-3314), one per workgroup.  The last node
// stays forward (reversing every node and the sequence satisfies the same edges), so there are 2^(n-1) orientation masks.
//   * masks are visited in increasing order, 256 at a time: the high bits `h` select a batch, a thread takes one low-bit
//     pattern.  A batch is skipped when the optimistic bound of `h` (every edge with a free end counts) cannot beat the best
//     value so far STRICTLY (bounds of 256 batches are computed in parallel, the batches then run in order, so the first
//     optimum found has the smallest mask);
//   * a thread whose own bound `ub` (weight of the compatible edges) beats the bar runs Kahn on its precedence rows (LDS, one
//     column per thread): acyclic => every compatible edge is satisfied, value = ub; cyclic => worth less than ub, the mask
//     goes onto the candidate list;
//   * candidates whose ub still beats the best acyclic value get the exact value from a wave: strongly connected components
//     of the arc graph (reachability closure by lane shuffles), arcs between different components are all satisfiable, every
//     non-trivial component goes through the subset DP h[S] = best weight obtainable inside it once S is placed (LDS, one
//     popcount layer at a time);
//   * canonical winner = max value, then smallest mask; its sequence = the lexicographically smallest optimal one (Kahn
//     smallest-first, or the greedy walk over the DP tables, as the host solver does).
// Anything that does not fit (more than OM_EMAX edges, a weight >= 2^20, more than OM_CAND live candidates, a strongly
// connected component of more than OM_SMAX nodes) sets status 1 and the host solver takes that component.
constexpr int OM_NMAX = 19, OM_EMAX = 512, OM_CAND = 2048, OM_SMAX = 12, OM_DPW = (1 << OM_SMAX) + 64;
struct OrdMidLds {
    uint32_t edge[OM_EMAX];            // u | v << 5 | hu << 10 | hv << 11 | w << 12
    uint32_t rows[OM_NMAX][256];       // rows[y][t]: nodes that must precede y under thread t's mask
    uint32_t cand_mask[OM_CAND];
    int cand_ub[OM_CAND];
    int bnd[256];
    unsigned long long best_key;       // (value + 1) << 20 | (0xFFFFF - mask); 0 = nothing yet
    int ncand, status;
    int arc[4][OM_NMAX * OM_NMAX];     // per wave: arc weights of one orientation
    int dp[4][OM_DPW];                 // per wave: h[] of one strongly connected component (the final walk uses all four areas)
};
__device__ __forceinline__ bool om_arc(uint32_t e, uint32_t mask, int& from, int& to, int& w) {
    const int u = e & 31, v = (e >> 5) & 31;
    const bool hu = (e >> 10) & 1, hv = (e >> 11) & 1;
    w = (int)(e >> 12);
    const bool yu = !((mask >> u) & 1), yv = !((mask >> v) & 1);
    bool compat, ufirst;
    if (hu != hv) { compat = yu == yv; ufirst = hv ? yu : !yu; }
    else { compat = yu != yv; ufirst = hu ? yv : yu; }
    from = ufirst ? u : v; to = ufirst ? v : u;
    return compat;
}
// exact value of one (cyclic) orientation by one wave; -1 when a strongly connected component exceeds OM_SMAX.
// With `walk` (wave 0 only, at the end) the DP tables of all components are kept (dp areas of all four waves) and lane 0
// builds the canonical sequence into order[].
__device__ int om_scc_value(OrdMidLds& L, int n, int m, uint32_t mask, int wave, bool walk, int* order) {
    const int lane = threadIdx.x & 63;
    int* a = L.arc[wave];
    for (int i = lane; i < n * n; i += 64) a[i] = 0;
    wave_sync();
    for (int e = lane; e < m; e += 64) {
        int from, to, w;
        if (om_arc(L.edge[e], mask, from, to, w)) atomicAdd(&a[from * n + to], w);
    }
    wave_sync();
    // reachability closure: lane x holds the set reachable from x
    uint32_t r = 0;
    if (lane < n) { r = 1u << lane; for (int y = 0; y < n; ++y) if (a[lane * n + y] > 0) r |= 1u << y; }
    for (int k = 0; k < n; ++k) { const uint32_t rk = (uint32_t)__shfl((int)r, k, 64); if ((r >> k) & 1) r |= rk; }
    uint32_t mutual = 0;
    for (int y = 0; y < n; ++y) { const uint32_t ry = (uint32_t)__shfl((int)r, y, 64); if (lane < n && ((ry >> lane) & 1) && ((r >> y) & 1)) mutual |= 1u << y; }
    const int rep = lane < n ? __ffs((int)mutual) - 1 : -1;
    // arcs between different components: all satisfiable
    int val = 0;
    for (int y = 0; y < n; ++y) { const int ry = __shfl(rep, y, 64); if (lane < n && ry != rep) val += a[lane * n + y]; }
    for (int d = 32; d >= 1; d >>= 1) val += __shfl_xor(val, d, 64);
    int* const dp0 = walk ? L.dp[0] : L.dp[wave];
    const int dpcap = walk ? 4 * OM_DPW : OM_DPW;
    int dpoff = 0;
    for (int x = 0; x < n; ++x) {  // components in order of their smallest member (uniform loop)
        const uint32_t mem = (uint32_t)__shfl((int)mutual, x, 64);
        const int rx = __shfl(rep, x, 64);
        if (rx != x || __popc(mem) < 2) continue;
        const int sz = __popc(mem), full = (1 << sz) - 1;
        if (sz > OM_SMAX || dpoff + full + 1 > dpcap) return -1;
        int* h = dp0 + dpoff;
        // member list: i-th set bit of mem
        auto member = [&](int i) { uint32_t t = mem; for (int q = 0; q < i; ++q) t &= t - 1; return __ffs((int)t) - 1; };
        if (lane == 0) h[full] = 0;
        wave_sync();
        for (int k = sz - 1; k >= 0; --k) {
            for (int S = lane; S <= full; S += 64) {
                if (__popc(S) != k) continue;
                int best = -1;
                for (int vi = 0; vi < sz; ++vi) {
                    if ((S >> vi) & 1) continue;
                    const int v = member(vi);
                    int g = 0;
                    for (int ui = 0; ui < sz; ++ui) if ((S >> ui) & 1) g += a[member(ui) * n + v];
                    const int t = g + h[S | (1 << vi)];
                    best = t > best ? t : best;
                }
                h[S] = best;
            }
            wave_sync();
        }
        val += h[0];
        if (walk) dpoff += full + 1;
    }
    if (walk) {  // (component sets and representatives through LDS: the walk below is lane 0's alone)
        if (lane < n) { L.bnd[lane] = (int)mutual; L.bnd[32 + lane] = rep; }
        wave_sync();
    }
    if (walk && lane == 0) {
        // lexicographically smallest optimal sequence: the smallest node whose predecessors from other components are placed
        // and whose own component can still reach its optimum (sq_order.cpp HostSolver::leaf)
        uint32_t done = 0;
        const uint32_t* mut = (const uint32_t*)L.bnd;
        const int* reps = L.bnd + 32;
        for (int p = 0; p < n; ++p)
            for (int v = 0; v < n; ++v) {
                if ((done >> v) & 1) continue;
                bool ok = true;
                for (int x = 0; x < n && ok; ++x) if (a[x * n + v] > 0 && reps[x] != reps[v] && !((done >> x) & 1)) ok = false;
                if (!ok) continue;
                const uint32_t mem = mut[v];
                if (__popc(mem) > 1) {
                    // offset of this component's table: components are stored in order of their smallest member
                    int off = 0;
                    for (int x = 0; x < reps[v]; ++x) if (reps[x] == x && __popc(mut[x]) > 1) off += 1 << __popc(mut[x]);
                    const int* h = dp0 + off;
                    // placed subset and local index of v inside the component
                    int S = 0, lv = 0, idx = 0;
                    for (uint32_t t = mem; t; t &= t - 1, ++idx) { const int y = __ffs((int)t) - 1; if ((done >> y) & 1) S |= 1 << idx; if (y == v) lv = idx; }
                    int g = 0; idx = 0;
                    for (uint32_t t = mem; t; t &= t - 1, ++idx) { const int y = __ffs((int)t) - 1; if ((S >> idx) & 1) g += a[y * n + v]; }
                    if (g + h[S | (1 << lv)] != h[S]) continue;
                }
                order[p] = v;
                done |= 1u << v;
                break;
            }
    }
    return val;
}
__global__ __launch_bounds__(256) void k_order_mid(const SmallProblem* probs, const int32_t* edges5, int32_t* out_mask, int32_t* out_order, int32_t* out_value, int32_t* out_status) {
    extern __shared__ unsigned long long om_raw[];
    OrdMidLds& L = *(OrdMidLds*)om_raw;
    const SmallProblem pr = probs[blockIdx.x];
    const int n = pr.n, m = pr.ecount, tid = threadIdx.x, wave = tid >> 6;
    if (tid == 0) { L.best_key = 0; L.ncand = 0; L.status = (n < 2 || n > OM_NMAX || m > OM_EMAX) ? 1 : 0; }
    __syncthreads();
    if (L.status == 0)
        for (int e = tid; e < m; e += 256) {
            const int32_t* q = edges5 + 5 * (size_t)(pr.eoff + e);
            if (q[4] >= (1 << 20) || q[4] < 0) L.status = 1;
            L.edge[e] = (uint32_t)q[0] | ((uint32_t)q[1] << 5) | ((uint32_t)(q[2] != 0) << 10) | ((uint32_t)(q[3] != 0) << 11) | ((uint32_t)q[4] << 12);
        }
    __syncthreads();
    if (L.status) { if (tid == 0) { out_status[blockIdx.x] = 1; out_mask[blockIdx.x] = 0; out_value[blockIdx.x] = -1; } return; }
    const int lowb = n - 1 < 8 ? n - 1 : 8, nlow = 1 << lowb, H = 1 << (n - 1 - lowb);
    for (int hbase = 0; hbase < H; hbase += 256) {
        {   // bound of batch h: nodes >= lowb fixed by h, nodes < lowb free (u < v, so a free v means a free u)
            const int h = hbase + tid;
            int b = -1;
            if (h < H) {
                const uint32_t hm = (uint32_t)h << lowb;
                b = 0;
                for (int e = 0; e < m; ++e) {
                    const uint32_t ed = L.edge[e];
                    int from, to, w;
                    if ((int)(ed & 31) < lowb || om_arc(ed, hm, from, to, w)) b += (int)(ed >> 12);
                }
            }
            L.bnd[tid] = b;
        }
        __syncthreads();
        const int nb = H - hbase < 256 ? H - hbase : 256;
        for (int j = 0; j < nb; ++j) {
            const int bar = (int)(L.best_key >> 20) - 1;  // best value so far (-1: none)
            __syncthreads();                               // (every thread has read the same bar before anybody raises it)
            if (L.bnd[j] <= bar) continue;
            if (tid < nlow) {
                const uint32_t mask = ((uint32_t)(hbase + j) << lowb) | (uint32_t)tid;
                int ub = 0;
                for (int e = 0; e < m; ++e) { int from, to, w; if (om_arc(L.edge[e], mask, from, to, w)) ub += w; }
                if (ub > bar) {
                    for (int y = 0; y < n; ++y) L.rows[y][tid] = 0;
                    for (int e = 0; e < m; ++e) { int from, to, w; if (om_arc(L.edge[e], mask, from, to, w)) L.rows[to][tid] |= 1u << from; }
                    uint32_t remaining = (1u << n) - 1;
                    bool acyclic = true;
                    for (int p = 0; p < n && acyclic; ++p) {
                        int v = -1;
                        for (uint32_t t = remaining; t; t &= t - 1) { const int cnd = __ffs((int)t) - 1; if (!(L.rows[cnd][tid] & remaining)) { v = cnd; break; } }
                        if (v < 0) acyclic = false; else remaining &= ~(1u << v);
                    }
                    if (acyclic) atomicMax(&L.best_key, ((unsigned long long)(ub + 1) << 20) | (unsigned long long)(0xFFFFFu - mask));
                    else {
                        const int at = atomicAdd(&L.ncand, 1);
                        if (at < OM_CAND) { L.cand_mask[at] = mask; L.cand_ub[at] = ub; }
                    }
                }
            }
            __syncthreads();
            // drop the candidates the new bar has overtaken (keeps the list short: thread 0, in place)
            const bool trim = L.ncand > OM_CAND / 2;
            __syncthreads();
            if (trim) {
                if (tid == 0) {
                    const int bar2 = (int)(L.best_key >> 20) - 1;
                    const int nc = L.ncand < OM_CAND ? L.ncand : OM_CAND;
                    if (L.ncand > OM_CAND) L.status = 1;
                    int k = 0;
                    for (int i = 0; i < nc; ++i) if (L.cand_ub[i] > bar2) { L.cand_mask[k] = L.cand_mask[i]; L.cand_ub[k] = L.cand_ub[i]; ++k; }
                    L.ncand = k;
                }
                __syncthreads();
            }
        }
        __syncthreads();
    }
    if (tid == 0 && L.ncand > OM_CAND) L.status = 1;
    __syncthreads();
    if (!L.status) {
        // exact values of the cyclic candidates, one per wave at a time; the bar only rises
        const int nc = L.ncand;
        for (int i = wave; i < nc; i += 4) {
            const int bar = (int)(L.best_key >> 20) - 1;
            if (L.cand_ub[i] <= bar) continue;  // (uniform per wave)
            const int v = om_scc_value(L, n, m, L.cand_mask[i], wave, false, nullptr);
            if (v < 0) { if ((tid & 63) == 0) L.status = 1; }
            else if ((tid & 63) == 0) atomicMax(&L.best_key, ((unsigned long long)(v + 1) << 20) | (unsigned long long)(0xFFFFFu - L.cand_mask[i]));
        }
    }
    __syncthreads();
    if (L.status) { if (tid == 0) { out_status[blockIdx.x] = 1; out_mask[blockIdx.x] = 0; out_value[blockIdx.x] = -1; } return; }
    if (wave != 0) return;
    // the winner's sequence (wave 0)
    const unsigned long long key = L.best_key;
    const uint32_t mask = 0xFFFFFu - (uint32_t)(key & 0xFFFFFu);
    const int value = (int)(key >> 20) - 1;
    __shared__ int s_order[OM_NMAX];
    __shared__ int s_acyclic;
    if (tid == 0) {
        for (int y = 0; y < n; ++y) L.rows[y][0] = 0;
        for (int e = 0; e < m; ++e) { int from, to, w; if (om_arc(L.edge[e], mask, from, to, w)) L.rows[to][0] |= 1u << from; }
        uint32_t remaining = (1u << n) - 1;
        bool acyclic = true;
        for (int p = 0; p < n && acyclic; ++p) {
            int v = -1;
            for (uint32_t t = remaining; t; t &= t - 1) { const int cnd = __ffs((int)t) - 1; if (!(L.rows[cnd][0] & remaining)) { v = cnd; break; } }
            if (v < 0) acyclic = false; else { remaining &= ~(1u << v); s_order[p] = v; }
        }
        s_acyclic = acyclic ? 1 : 0;
    }
    wave_sync();
    int status = 0;
    if (!s_acyclic) {
        const int v = om_scc_value(L, n, m, mask, 0, true, s_order);
        if (v != value) status = 1;  // (tables of all components did not fit side by side: the host solver takes it)
    }
    wave_sync();
    if (tid == 0) {
        out_status[blockIdx.x] = status; out_mask[blockIdx.x] = (int32_t)mask; out_value[blockIdx.x] = value;
        for (int p = 0; p < n; ++p) out_order[(size_t)blockIdx.x * OM_NMAX + p] = s_order[p];
    }
}

// ------------------------------------------------------------------------------------------------ K-1: BGZF inflate
// DEFLATE is serial inside a block, so the parallelism is across blocks: every LANE of a wave decodes its own BGZF block (64
// independent streams in lockstep; the vector unit takes 64 symbols per step).  Two kernels: k_inflate_tok2 turns the streams into
// LZ77 tokens, k_lz_resolve2 (two waves per block, the 64 KiB window in LDS) turns the tokens into bytes -- a lane that reads back
// what it has just written would make the whole wave wait for its stores, hence decoding and copying are apart.  (The round-1 /
// early round-2 forms -- one wave per block, the table-driven lane kernel, the one-wave resolve -- are gone; `git log` has them.)
__constant__ uint8_t c_clorder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};  // order of the code-length code lengths in a dynamic block header (RFC 1951)
constexpr int IL_STAGE = 4;
// headers are read and tables are built by every lane for itself, in lockstep with the other lanes that are at a block header: zlib
// closes a block every 16383 symbols, so the lanes of a wave arrive together (a lane that is early waits up to IL_HDR_WAIT steps)
#ifndef SQ_IL_HDR_WAIT
#define SQ_IL_HDR_WAIT 48
#endif
constexpr int IL_HDR_WAIT = SQ_IL_HDR_WAIT;
// The compressed bytes of a lane go through a ring of IL_RING words in LDS (ring[(word % IL_RING) * 64], already offset
// by the lane): the bit buffer refills from LDS, and the ring is topped up from global memory by all lanes in the same
// step, when any of them runs low.  A load issued by one lane in one step would otherwise make the whole wave wait a
// memory round trip in the next (the counters that order memory operations are per wave, not per lane).
// org: byte offset of ring word 0 in the block; rd / ld: words moved into the bit buffer / loaded into the ring.
constexpr int IL_RING = 16;
struct ILane {
    const uint8_t* p; uint32_t* ring; uint32_t n, org; int rd, ld; unsigned long long buf; int cnt;
};
__device__ __forceinline__ uint4 il_load16(const uint8_t* p) { uint4 w; __builtin_memcpy(&w, p, 16); return w; }
__device__ __forceinline__ uint32_t il_pos(const ILane& b) { return b.org + 4u * (uint32_t)b.rd; }  // bytes moved into the bit buffer
__device__ __forceinline__ bool il_low(const ILane& b) { return b.ld - b.rd < 6; }  // (a step takes at most 56 bits, the fixed part of a block header 74)
__device__ __forceinline__ void il_topup(ILane& b) {  // as many 16-byte pieces as fit, loads first, then the LDS stores
    constexpr int PIECES = IL_RING / 4;
    uint4 w[PIECES];
    const int free_pieces = (IL_RING - (b.ld - b.rd)) >> 2;
    const uint8_t* src = b.p + b.org + 4u * (uint32_t)b.ld;
#pragma unroll
    for (int k = 0; k < PIECES; ++k) if (k < free_pieces) w[k] = il_load16(src + 16 * k);
#pragma unroll
    for (int k = 0; k < PIECES; ++k)
        if (k < free_pieces) {
            const int at = b.ld + 4 * k;
            b.ring[((at + 0) & (IL_RING - 1)) * 64] = w[k].x; b.ring[((at + 1) & (IL_RING - 1)) * 64] = w[k].y;
            b.ring[((at + 2) & (IL_RING - 1)) * 64] = w[k].z; b.ring[((at + 3) & (IL_RING - 1)) * 64] = w[k].w;
        }
    b.ld += 4 * free_pieces;
}
__device__ __forceinline__ void il_start(ILane& b, uint32_t at) {
    b.org = at; b.rd = 0; b.ld = 0; b.buf = 0; b.cnt = 0;
    il_topup(b);
}
__device__ __forceinline__ void il_refill(ILane& b) {
    if (b.cnt <= 32) {
        const uint32_t w = b.ring[(b.rd & (IL_RING - 1)) * 64];
        b.buf |= (unsigned long long)w << b.cnt; b.cnt += 32; ++b.rd;
    }
}
__device__ __forceinline__ uint32_t il_take(ILane& b, int k) {
    il_refill(b);
    const uint32_t v = (uint32_t)(b.buf & ((1ull << k) - 1));
    b.buf >>= k; b.cnt -= k;
    return v;
}
// Per-lane decoding tables, all in LDS and interleaved by lane (entry e of lane l at [e * 64 + l]); anything a lane
// fetched from global memory would cost the whole wave a memory round trip per step, and with 64 streams some lane is on
// the rare path nearly every step:
// ---- The token pass: canonical Huffman decoding with the code limits in REGISTERS.
// (Two-level lookup tables per lane as in zlib cost 2.5 KB per lane, 160 KB per wave, ONE wave per CU, and every step was a chain
// of dependent LDS round trips with nobody to hide them: round 1.)  A canonical code needs no table of codes: with the next 15 bits of
// the stream read MSB-first as a number v, the codes of length L occupy [first[L] << (15 - L), (first[L] + count[L]) << (15 - L)),
// ranges that ascend with L.  So the length of the next code is 1 + #{L : v >= limit[L]} -- fourteen compares against values
// that live in 15 registers per code (literal/length and distance) -- and the symbol is sym[(v >> (15 - len)) + K[len]].  What
// stays in LDS per lane: the symbol permutation as bytes (288 + 32; a literal/length symbol >= 256 is told from its rank inside
// its length group, where the literals come first), three 16-entry tables, the input ring and the token stage (the sizes: "Round 4"
// below) -- several waves per CU, and a step is two LDS round trips per code instead of up to nine.
#ifndef SQ_T2_LITS
#define SQ_T2_LITS 3  // literal/length symbols per step (4 until the second half of round 4: with five waves per CU three is 1 % faster, 140.2 against 141.7 ms per C3 step in three A/B pairs)
#endif
constexpr int T2_SYM_LL = 288, T2_SYM_DD = 32, T2_LITS = SQ_T2_LITS;
// Round 4: 496 bytes per lane (round 3: 746).  What went: the 4-bit code lengths of a header (170 B) -- they are only needed between
// reading a header and scattering its symbols, once per ~16 k symbols, and now travel through a per-lane strip of GLOBAL memory, eight
// to a word, written and read back sequentially (T2_LENS_WORDS words per lane; a wave's word k lies side by side: [k * 64 + lane]);
// the two scratch tables of the builder (64 B) -- the counts are taken while the header is read, straight into the table that will
// hold the slot offsets, the scatter uses that table as its cursors and a 15-step pass turns cursors back into offsets; half the token
// stage (16 B).  31 KB per wave: THREE token waves per workgroup (93 KB) beside the 64 KB slot of a resolve workgroup on every CU, a
// third SIMD of every CU decoding -- the token pass is issue-bound at one wave per SIMD, so its throughput is the number of SIMDs it
// gets.
constexpr int T2_LENS_WORDS = 40;  // 320 code lengths of 4 bits
constexpr size_t T2_LDS_BYTES = (size_t)(T2_SYM_LL + T2_SYM_DD) * 64 + (size_t)(3 * 16) * 64 * 2 + (size_t)(IL_STAGE + IL_RING) * 64 * 4;
// token slots of a block: a token is a match (>= 3 bytes) or up to three literals, and a literal run is cut short only in front of a match
// or at the end of a deflate block -- at most one token per two bytes but for the (few) deflate-block ends.  A stream that still needs more
// (thousands of tiny deflate blocks) raises the error flag and the file goes to the host reader.
__host__ __device__ inline uint32_t t2_tok_cap(uint32_t isize) { return ((isize >> 1) + 64u + (IL_STAGE - 1)) & ~(uint32_t)(IL_STAGE - 1); }
struct T2Lds {
    uint8_t *sym_ll, *sym_dd;                // [e * 64 + lane]
    int16_t *k_ll, *k_dd;                    // [len * 64 + lane]: symbol slot of a code = (v >> (15 - len)) + k[len]
    uint16_t* nl_ll;                         // nl_ll[len]: slot where the symbols >= 256 of that length begin (while a header is read: the counts of the distance code)
};
// the code lengths of a header on their way through global memory: eight to a word, strictly sequential
struct LensOut { uint32_t* w; uint32_t acc; int n; };  // w: the lane's strip (word k at w[k * 64])
__device__ __forceinline__ void lens_put(LensOut& o, int v) {
    o.acc |= (uint32_t)v << ((o.n & 7) << 2);
    if ((o.n & 7) == 7) { o.w[(o.n >> 3) * 64] = o.acc; o.acc = 0; }
    ++o.n;
}
__device__ __forceinline__ void lens_flush(LensOut& o) { if (o.n & 7) o.w[(o.n >> 3) * 64] = o.acc; }
// (read one word ahead: the builder spends three LDS round trips on a symbol, and a word asked for when its first symbol is needed costs a
// memory round trip per eight symbols on top; the word behind a strip's last one belongs to the next strip or to the slack behind the last)
struct LensIn { const uint32_t* w; uint32_t acc, nxt; int i; };
__device__ __forceinline__ void lens_seek(LensIn& r, const uint32_t* w, int at) { r.w = w; r.i = at; r.acc = w[(at >> 3) * 64]; r.nxt = w[((at >> 3) + 1) * 64]; }
__device__ __forceinline__ int lens_get(LensIn& r) {
    const int v = (int)((r.acc >> ((r.i & 7) << 2)) & 15u);
    ++r.i;
    if ((r.i & 7) == 0) { r.acc = r.nxt; r.nxt = r.w[((r.i >> 3) + 1) * 64]; }
    return v;
}
// canonical tables of one code.  In: cnt[l * 64 + lane] = number of symbols of length l (l = 1..15; cnt may BE ktab), get(i) = length
// of symbol i, asked for i = 0 .. n - 1 in order.  Out: the code limits in REGISTERS, two 16-bit values per register (limit - 1, so that
// the sign of (limit - 1 - v) in 16 bits says v >= limit): lp[j] holds lengths 2j + 1 and 2j + 2, lp[7] the limit of length 15 alone;
// slot offsets in ktab, symbols in symtab, nl_ll for the literal/length code.  Called by the lanes that are at a header (me), each for
// its own code.  Returns false for an over-subscribed set.
typedef short t2_s16x2 __attribute__((ext_vector_type(2)));
template <bool LL, class Get>
__device__ bool t2_build(const T2Lds& L, int lane, bool me, int n, const uint16_t* cnt, Get get, uint32_t (&lp)[8], uint8_t* symtab, int16_t* ktab) {
    bool ok = true;
    if (me) {
        int left = 1, first = 0, off = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) lp[j] = 0;
#pragma unroll
        for (int l = 1; l <= 15; ++l) {
            const int k = cnt[l * 64 + lane];
            left = (left << 1) - k;
            if (left < 0) ok = false;
            const uint32_t lim = (uint32_t)(first + k) << (15 - l);  // <= 32768
            lp[(l - 1) >> 1] |= ((lim - 1u) & 0xffffu) << (((l - 1) & 1) << 4);
            ktab[l * 64 + lane] = (int16_t)off;  // (the scatter's cursor of this length; turned into the slot offset below)
            if (LL) L.nl_ll[l * 64 + lane] = (uint16_t)off;
            off += k;
            first = (first + k) << 1;
        }
        if (ok) {
            for (int i = 0; i < n; ++i) {
                const int l = get(i);
                if (!l) continue;
                const int at = ktab[l * 64 + lane]++;
                symtab[at * 64 + lane] = (uint8_t)i;
                if (LL && i < 256) ++L.nl_ll[l * 64 + lane];
            }
            off = 0; first = 0;
#pragma unroll
            for (int l = 1; l <= 15; ++l) {  // the cursor of a length has arrived at the end of its group = the start of the next
                const int end = ktab[l * 64 + lane], k = end - off;
                ktab[l * 64 + lane] = (int16_t)(off - first);
                off = end;
                first = (first + k) << 1;
            }
        }
    }
    return ok;
}
// length of the next code and its symbol slot; -1 for a bit pattern no code of the set covers.  Seven packed 16-bit
// subtractions decide fourteen "v >= limit" at once (a lone wave per SIMD pays every dependent vector instruction in full:
// the fewer, the better), one more the length-15 limit; the slot offset of the length comes from LDS.
__device__ __forceinline__ int t2_slot(ILane& b, const uint32_t (&lp)[8], const int16_t* ktab, int lane, int& len) {
    const uint32_t v = __brev((uint32_t)b.buf) >> 17;
    const uint32_t vv = v | (v << 16);
    t2_s16x2 acc = {0, 0};
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        t2_s16x2 d = __builtin_bit_cast(t2_s16x2, lp[j]) - __builtin_bit_cast(t2_s16x2, vv);
        acc += d >> 15;  // -1 where v >= limit
    }
    const int l = 1 - ((int)acc.x + (int)acc.y);
    len = l;
    if ((int)(short)(lp[7] & 0xffffu) - (int)v < 0) return -1;  // v >= limit of length 15
    const int idx = (int)(v >> (15 - l)) + (int)ktab[l * 64 + lane];
    b.buf >>= l; b.cnt -= l;
    return idx;
}
// bit buffer refill with the next ring word already in a register (loaded at the refill before: its LDS round trip is over by
// the time it is needed)
__device__ __forceinline__ void t2_refill(ILane& b, uint32_t& ahead) {
    if (b.cnt <= 32) {
        b.buf |= (unsigned long long)ahead << b.cnt; b.cnt += 32; ++b.rd;
        ahead = b.ring[(b.rd & (IL_RING - 1)) * 64];
    }
}
__device__ __forceinline__ uint32_t t2_take(ILane& b, uint32_t& ahead, int k) {
    t2_refill(b, ahead);
    const uint32_t v = (uint32_t)(b.buf & ((1ull << k) - 1));
    b.buf >>= k; b.cnt -= k;
    return v;
}
template <bool PROF>
__global__ __launch_bounds__(320, 1) void k_inflate_tok2(const uint8_t* file, const InflBlock* blocks, int nblocks, int32_t* flags, uint32_t* tok, int32_t* ntok, uint32_t* lens_strips,
                                                    unsigned long long* prof) {
    // One to five waves per workgroup (SQUID_TOK_WPB), each with its own T2_LDS_BYTES and its own 64 blocks; they never talk to each other.
    // With the resolve that keeps no window in LDS (k_lz_resolve3, the default) the workgroups are single waves and the dispatcher packs five
    // of them onto a CU (158.7 of its 160 KB); with the LDS resolve (k_lz_resolve2) three waves make a workgroup of 93 KB, a CU takes one of
    // them and no second, and the 64 KB that remain are exactly the slot of a resolve workgroup.
    extern __shared__ uint16_t il_lds_all[];  // T2_LDS_BYTES per wave
    uint16_t* il_lds = il_lds_all + (size_t)(threadIdx.x >> 6) * (T2_LDS_BYTES / 2);
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pc = 0;
    auto tick = [&](int k) { if (PROF) { const unsigned long long now = __builtin_amdgcn_s_memtime(); pt[k] += now - pc; pc = now; } };
    if (PROF) pc = __builtin_amdgcn_s_memtime();
    __shared__ uint8_t sh_clo[32];
    T2Lds L;
    L.k_ll = (int16_t*)il_lds;                 L.k_dd = L.k_ll + 16 * 64;
    L.nl_ll = (uint16_t*)(L.k_dd + 16 * 64);
    uint32_t* stage = (uint32_t*)(L.nl_ll + 16 * 64);  // tokens on their way out: stage[(k % IL_STAGE) * 64 + lane]
    uint32_t* ring = stage + IL_STAGE * 64;
    L.sym_ll = (uint8_t*)(ring + IL_RING * 64);  L.sym_dd = L.sym_ll + T2_SYM_LL * 64;
    const int lane = threadIdx.x & 63;
    if (lane < 19) sh_clo[lane] = c_clorder[lane];  // (all waves write the same values)
    wave_sync();
    const int gwave = blockIdx.x * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6);
    const int bi = gwave * 64 + lane;
    const bool have = bi < nblocks;
    InflBlock blk{0, 0, 0, 0, 0};
    if (have) blk = blocks[bi];
    uint32_t* tk = tok + blk.toff;
    const uint32_t tcap = t2_tok_cap(blk.isize);
    uint32_t* lens_w = lens_strips + (size_t)gwave * (64 * T2_LENS_WORDS) + lane;  // this lane's strip of code lengths (word k at [k * 64])
    uint32_t nt = 0;
    bool done = !have || blk.isize == 0, err = false;
    auto emit = [&](uint32_t v) {
        if (nt >= tcap) { err = true; done = true; return; }  // (more tokens than the block's slots hold: see t2_tok_cap)
        stage[(nt % IL_STAGE) * 64 + lane] = v;
        if ((++nt % IL_STAGE) == 0) {  // a full stage: IL_STAGE consecutive tokens in one wide store
            static_assert(IL_STAGE == 4, "one 16-byte store per stage");
            uint4 w{stage[lane], stage[64 + lane], stage[2 * 64 + lane], stage[3 * 64 + lane]};
            __builtin_memcpy(tk + nt - IL_STAGE, &w, 16);
        }
    };
    ILane b;
    b.p = file + blk.coff; b.n = blk.clen; b.ring = ring + lane;
    il_start(b, 0);
    uint32_t lp_ll[8], lp_dd[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { lp_ll[q] = 0; lp_dd[q] = 0; }
    uint32_t ahead = b.ring[0];  // (il_start has filled the ring)
    uint32_t outpos = 0;
    bool in_block = false, last = false, stored = false;
    uint32_t stored_left = 0, stored_at = 0;
    int hdr_wait = 0;
    // what the step has produced -- with up to four literal/length symbols per step at most three tokens (two pending literals + one
    // fill a token, two more wait, a match flushes them and adds its own): stored at the top of the next step, where all lanes
    // are together again (one copy of the store code)
    uint32_t tq0 = 0, tq1 = 0, tq2 = 0;
    int ntq = 0;
    auto push = [&](uint32_t t_) { tq0 = ntq == 0 ? t_ : tq0; tq1 = ntq == 1 ? t_ : tq1; tq2 = ntq == 2 ? t_ : tq2; ++ntq; };  // (selects: the three stay in registers)
    static_assert(T2_LITS <= 4, "the token queue of a step holds three tokens");
    uint32_t lit = 0;         // literals waiting for company (up to three per token)
    int nlit = 0;
    while (__any(!done)) {
        tick(7);
        if (ntq) { emit(tq0); if (ntq > 1) emit(tq1); if (ntq > 2) emit(tq2); ntq = 0; }
        if (PROF) ++pt[6];
        tick(0);
        // ---- block headers, by all the lanes that are at one
        const unsigned long long need = __ballot(!done && !in_block && !stored);
        if (need && (need == __ballot(!done) || ++hdr_wait >= IL_HDR_WAIT)) {
            hdr_wait = 0;
            const bool me = (need >> lane) & 1;
            int kind = -1, nlen = 0, ndist = 0, ncode = 0;  // 0 stored, 1 fixed code, 2 dynamic code, -1 corrupt
            // the code lengths go out to the lane's strip as they are decoded, counted on the way: literal/length counts into k_ll (the
            // table the builder turns into slot offsets), distance counts into nl_ll (free until the literal/length code is built)
            LensOut lo{lens_w, 0, 0};
            int eob_len = 0;
            auto put = [&](int v) {
                if (lo.n == 256) eob_len = v;
                if (v) { if (lo.n < nlen) ++L.k_ll[v * 64 + lane]; else ++L.nl_ll[v * 64 + lane]; }
                lens_put(lo, v);
            };
            if (me) {
                if (il_low(b)) il_topup(b);
                if (il_pos(b) <= b.n + 8) {
                    last = t2_take(b, ahead, 1);
                    const uint32_t type = t2_take(b, ahead, 2);
                    if (type == 0) {
                        b.buf >>= (b.cnt & 7); b.cnt -= (b.cnt & 7);
                        const uint32_t len = t2_take(b, ahead, 16), nl = t2_take(b, ahead, 16);
                        stored_at = il_pos(b) - (uint32_t)(b.cnt >> 3);
                        if ((len ^ 0xffff) == nl && stored_at + len <= b.n && outpos + len <= blk.isize) {
                            kind = 0;
                            stored_left = len;
                            if (len) stored = true; else { il_start(b, stored_at); ahead = b.ring[0]; if (last) done = true; }
                        }
                    } else if (type == 1 || type == 2) {
                        for (int l = 0; l < 16; ++l) { L.k_ll[l * 64 + lane] = 0; L.nl_ll[l * 64 + lane] = 0; }
                        if (type == 1) {  // the fixed code of RFC 1951 3.2.6, through the same door as a dynamic one
                            kind = 1; nlen = 288; ndist = 30;
                            for (int i = 0; i < 144; ++i) put(8);
                            for (int i = 144; i < 256; ++i) put(9);
                            for (int i = 256; i < 280; ++i) put(7);
                            for (int i = 280; i < 288; ++i) put(8);
                            for (int i = 0; i < 30; ++i) put(5);
                            lens_flush(lo);
                        } else {
                            nlen = (int)t2_take(b, ahead, 5) + 257; ndist = (int)t2_take(b, ahead, 5) + 1;
                            ncode = (int)t2_take(b, ahead, 4) + 4;
                            if (nlen <= 286 && ndist <= 30) kind = 2;
                        }
                    }
                }
            }
            if (__any(me && kind == 2)) {
                // the code-length code (19 symbols of 3-bit lengths, in a register) in the distance tables, then the literal/length +
                // distance lengths with it
                const bool dyn = me && kind == 2;
                unsigned long long cl_lens = 0;
                if (dyn) {
                    for (int l = 0; l < 16; ++l) L.k_dd[l * 64 + lane] = 0;
                    for (int i = 0; i < ncode; ++i) { const int v = (int)t2_take(b, ahead, 3); cl_lens |= (unsigned long long)v << (3 * sh_clo[i]); if (v) ++L.k_dd[v * 64 + lane]; }
                }
                bool ok = t2_build<false>(L, lane, dyn, 19, (const uint16_t*)L.k_dd, [&](int i) { return (int)((cl_lens >> (3 * i)) & 7); }, lp_dd, L.sym_dd, L.k_dd);
                int prev = 0;
                bool busy = dyn && ok;
                while (__any(busy)) {
                    if (busy) {
                        if (il_low(b)) il_topup(b);
                        t2_refill(b, ahead);
                        int cl;
                        const int at = t2_slot(b, lp_dd, L.k_dd, lane, cl);
                        const int sym = at < 0 ? -1 : (int)L.sym_dd[(at & 31) * 64 + lane];
                        if (sym < 0 || sym > 18 || il_pos(b) > b.n + 8) { ok = false; busy = false; }
                        else if (sym < 16) { put(sym); prev = sym; }
                        else {
                            int rep, v = 0;
                            if (sym == 16) { v = prev; rep = 3 + (int)t2_take(b, ahead, 2); if (lo.n == 0) ok = false; }
                            else if (sym == 17) rep = 3 + (int)t2_take(b, ahead, 3);
                            else rep = 11 + (int)t2_take(b, ahead, 7);
                            if (!ok || lo.n + rep > nlen + ndist) { ok = false; busy = false; }
                            else { while (rep--) put(v); prev = v; }
                        }
                        if (busy && lo.n >= nlen + ndist) busy = false;
                    }
                }
                if (dyn && ok) lens_flush(lo);
                if (dyn && ok && eob_len == 0) ok = false;  // no end-of-block code
                if (dyn && !ok) kind = -1;
            }
            if (__any(me && kind > 0)) {
                // the distance code first: its counts sit in nl_ll, which the literal/length build then takes for its own use
                const bool bld = me && kind > 0;
                LensIn ri{lens_w, 0, 0, 0};
                if (bld) lens_seek(ri, lens_w, nlen);
                const bool ok_dd = t2_build<false>(L, lane, bld, ndist, L.nl_ll, [&](int) { return lens_get(ri); }, lp_dd, L.sym_dd, L.k_dd);
                if (bld) lens_seek(ri, lens_w, 0);
                const bool ok_ll = t2_build<true>(L, lane, bld, nlen, (const uint16_t*)L.k_ll, [&](int) { return lens_get(ri); }, lp_ll, L.sym_ll, L.k_ll);
                if (bld && !(ok_ll && ok_dd)) kind = -1;
            }
            if (me) { if (kind < 0) { err = true; done = true; } else if (kind > 0) in_block = true; }
        }
        tick(1);
        if (__any(!done && il_low(b))) { if (!done) il_topup(b); }  // every lane, in the same step
        tick(2);
        if (done) continue;
        if (stored) {  // a slice of a stored block
            const uint32_t k = stored_left < 15 ? stored_left : 15;  // five literal tokens of three bytes
            for (uint32_t i = 0; i < k; i += 3) {
                const uint32_t m = k - i < 3 ? k - i : 3;
                uint32_t t_ = m << 24;
                for (uint32_t q = 0; q < m; ++q) t_ |= (uint32_t)b.p[stored_at + i + q] << (8 * q);
                emit(t_);
            }
            outpos += k; stored_at += k; stored_left -= k;
            if (!stored_left) { stored = false; il_start(b, stored_at); ahead = b.ring[0]; if (last) done = true; }
            continue;
        }
        if (!in_block) continue;  // (waiting at a header)
        // ---- up to T2_LITS literal/length symbols per step.  The distance half of a step (below) is executed whenever ANY lane has
        // a match, i.e. nearly every step, for the few lanes that have one; it costs more than a literal/length decode.  So the
        // lanes keep decoding literals -- most symbols of a BAM stream -- until they meet a length symbol or the end of the block,
        // and the wave goes through the distance half once per step for all of them.  Literals travel three to a token (bits
        // 24..25 = how many, the bytes in bits 0..23): the resolve pass takes 64 tokens per round.
        int sym = -1;  // the length / end-of-block symbol the lane stopped at
        bool go = true;
#pragma unroll
        for (int rep = 0; rep < T2_LITS; ++rep) {
            if (!go) continue;
            t2_refill(b, ahead);
            int cl;
            const int at = t2_slot(b, lp_ll, L.k_ll, lane, cl);
            if (at < 0 || at >= T2_SYM_LL || il_pos(b) > b.n + 8) { err = true; done = true; go = false; continue; }
            const int sy = (int)L.sym_ll[at * 64 + lane] + (at >= (int)L.nl_ll[cl * 64 + lane] ? 256 : 0);
            if (sy >= 256) { sym = sy; go = false; continue; }
            if (outpos >= blk.isize) { err = true; done = true; go = false; continue; }
            lit |= (uint32_t)sy << (8 * nlit);
            ++nlit; ++outpos;
            if (nlit == 3) { push(lit | (3u << 24)); lit = 0; nlit = 0; }
        }
        tick(3);
        if (done || sym < 0) continue;
        if (nlit) { push(lit | ((uint32_t)nlit << 24)); lit = 0; nlit = 0; }  // the literals in front of a match / the end of the block
        if (sym == 256) { in_block = false; if (last) done = true; continue; }
        const int ls = sym - 257;
        if (ls >= 29) { err = true; done = true; continue; }
        // base and extra bits of the length / distance symbol by arithmetic (RFC 1951 3.2.5).  One refill covers the rest of the
        // step: 5 + 15 + 13 bits at most.
        t2_refill(b, ahead);
        const uint32_t lx = ls < 8 || ls == 28 ? 0u : (uint32_t)(ls - 4) >> 2;
        const uint32_t lbase = ls < 8 ? 3u + (uint32_t)ls : (ls == 28 ? 258u : 3u + ((4u + ((uint32_t)ls & 3u)) << lx));
        const uint32_t len = lbase + ((uint32_t)b.buf & ((1u << lx) - 1));
        b.buf >>= lx; b.cnt -= (int)lx;
        int dl;
        const int dat = t2_slot(b, lp_dd, L.k_dd, lane, dl);
        const int ds = dat < 0 ? -1 : (int)L.sym_dd[(dat & 31) * 64 + lane];
        if (ds < 0 || ds >= 30) { err = true; done = true; continue; }
        const uint32_t dx = ds < 4 ? 0u : (uint32_t)(ds - 2) >> 1;
        const uint32_t dbase = ds < 4 ? 1u + (uint32_t)ds : 1u + ((2u + ((uint32_t)ds & 1u)) << dx);
        const uint32_t dist = dbase + ((uint32_t)b.buf & ((1u << dx) - 1));
        b.buf >>= dx; b.cnt -= (int)dx;
        if (dist > outpos || outpos + len > blk.isize) { err = true; done = true; continue; }
        push(0x80000000u | (len << 16) | (dist - 1)); outpos += len;
        tick(4);
    }
    if (PROF && lane == 0) { for (int k = 0; k < 8; ++k) prof[(size_t)blockIdx.x * 8 + k] = pt[k]; }
    if (ntq) { emit(tq0); if (ntq > 1) emit(tq1); if (ntq > 2) emit(tq2); }
    if (have && (err || outpos != blk.isize)) atomicOr(&flags[0], 512);
    if (have) {
        for (uint32_t k = nt - nt % IL_STAGE; k < nt; ++k) tk[k] = stage[(k % IL_STAGE) * 64 + lane];
        ntok[bi] = (int32_t)nt;
    }
}

// The token pass of round 6: one wave per BGZF block, the lanes on 64 consecutive stretches of its bit stream (sq_inflate_spec.inc).
template <int CH, int PB>
__global__ __launch_bounds__(64) void k_inflate_spec(const uint8_t* file, const InflBlock* blocks, int nblocks, int32_t* flags, uint32_t* tok, int32_t* ntok) {
    extern __shared__ uint32_t isp_lds[];
    const int bi = (int)blockIdx.x;
    if (bi >= nblocks) return;
    const InflBlock blk = blocks[bi];
    bool err = false;
    uint32_t nt = 0;
    if (blk.isize) nt = isp::inflate_block_spec<CH, PB>((wv::lds_u32*)isp_lds, file + blk.coff, blk.clen, tok + blk.toff, isp::tok_cap_spec(blk.isize, blk.clen), err);
    if (threadIdx.x == 0) { ntok[bi] = (int32_t)nt; if (err) atomicOr(&flags[0], 512); }
}
struct SpecProf {  // cycles per phase of the block, summed over the launch (one atomic per counter and block)
    unsigned long long* g; unsigned long long t0; unsigned long long c[9];
    __device__ SpecProf(unsigned long long* g_) : g(g_), t0(__builtin_amdgcn_s_memtime()) { for (int i = 0; i < 9; ++i) c[i] = 0; }
    __device__ __forceinline__ void tick(int k) { const unsigned long long now = __builtin_amdgcn_s_memtime(); c[k] += now - t0; t0 = now; }
    __device__ __forceinline__ void count(int k, uint32_t n = 1) { c[k] += n; }
    __device__ __forceinline__ void finish() { if ((threadIdx.x & 63u) == 0) for (int i = 0; i < 9; ++i) atomicAdd(&g[i], c[i]); }
};
template <int CH, int PB>
__global__ __launch_bounds__(64) void k_inflate_spec_prof(const uint8_t* file, const InflBlock* blocks, int nblocks, int32_t* flags, uint32_t* tok, int32_t* ntok, unsigned long long* prof) {
    extern __shared__ uint32_t isp_lds[];
    const int bi = (int)blockIdx.x;
    if (bi >= nblocks) return;
    const InflBlock blk = blocks[bi];
    bool err = false;
    uint32_t nt = 0;
    if (blk.isize) nt = isp::inflate_block_spec<CH, PB, SpecProf>((wv::lds_u32*)isp_lds, file + blk.coff, blk.clen, tok + blk.toff, isp::tok_cap_spec(blk.isize, blk.clen), err, SpecProf(prof));
    if (threadIdx.x == 0) { ntok[bi] = (int32_t)nt; if (err) atomicOr(&flags[0], 512); }
}
// (the variants the tuning entry sq_debug_token_bench can time; the reader runs SPEC_CH / SPEC_PB)
#ifndef SQ_SPEC_CH
#define SQ_SPEC_CH 384  // (round 6, last: 384-bit stretches -- 12.6 windows per block instead of 19.2 at 10.1 instead of 9.1 KB per wave: 4.5 against 5.1 ms per GiB alone)
#endif
#ifndef SQ_SPEC_PB
#define SQ_SPEC_PB 10
#endif
constexpr int SPEC_CH = SQ_SPEC_CH, SPEC_PB = SQ_SPEC_PB;
template <int CH, int PB>
static void launch_inflate_spec(hipStream_t s, const uint8_t* file, const InflBlock* blocks, int nb, int32_t* flags, uint32_t* tok, int32_t* ntok) {
    typedef isp::Lay<CH, PB> Y;
    hipLaunchKernelGGL((k_inflate_spec<CH, PB>), dim3(nb), dim3(64), (size_t)Y::BYTES, s, file, blocks, nb, flags, tok, ntok);
}

// Inclusive prefix sum over the wave with DPP row shifts (no LDS traffic).
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return x;
}
// LZ77 resolution: one wave per BGZF block, the block's output (<= 64 KiB) in LDS.  64 tokens per round: a prefix sum of
// their lengths places them, literals are stored at once, and the matches copy in as few sub-rounds as their dependencies
// allow -- a match is ready when its source lies below the output of the first match still pending (everything below
// that is final); a match overlapping its own output only needs the bytes in front of it, its pattern repeats.
__global__ __launch_bounds__(128) void k_lz_resolve2(const uint32_t* tok, const int32_t* ntok, const InflBlock* blocks, int nblocks, unsigned long long out_base, uint8_t* outbuf, int32_t* flags) {
    extern __shared__ uint8_t lz_win[];
    __shared__ uint32_t s_base[2];  // [r & 1]: where the output of round r starts (written by the wave that prepares round r - 1 ... see below)
    __shared__ int s_bad;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const InflBlock blk = blocks[blockIdx.x];
    const uint32_t* t = tok + blk.toff;
    const int n = ntok[blockIdx.x];
    const int nr = (n + 63) / 64;  // rounds; wave w prepares and copies the rounds r with (r & 1) == w
    if (threadIdx.x == 0) { s_base[0] = 0; s_base[1] = 0; s_bad = 0; }
    __syncthreads();
    uint32_t nxt = wave * 64 + lane < n ? t[wave * 64 + lane] : 0;  // tokens of this wave's next round
    // state of the round this wave has prepared and not yet copied
    uint32_t o = 0, len = 0, dist = 1, src = 0, ready_at = 0;
    bool pending = false;
    for (int step = 0; step <= nr; ++step) {
        const bool bad_now = s_bad != 0;
        if (!bad_now && step < nr && (step & 1) == wave) {
            // ---- half 1 of round `step`
            const int r0 = step * 64;
            const uint32_t tk = nxt;
            const int i = r0 + lane;
            if (i + 128 < n) nxt = t[i + 128];
            const bool valid = i < n, is_m = valid && (tk >> 31);
            const uint32_t nl = (tk >> 24) & 3u;  // a literal token carries 1..3 bytes (0 stands for 1: the one-literal tokens of the first token pass)
            len = !valid ? 0u : (is_m ? (tk >> 16) & 0x1ffu : (nl ? nl : 1u));
            const uint32_t inc = wave_scan_incl(len);
            const uint32_t base = s_base[step & 1];
            o = base + inc - len;
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            bool bad = base + total > blk.isize;  // (uniform)
            if (!bad) {
                if (valid && !is_m) { lz_win[o] = (uint8_t)tk; if (len > 1) lz_win[o + 1] = (uint8_t)(tk >> 8); if (len > 2) lz_win[o + 2] = (uint8_t)(tk >> 16); }
                dist = (tk & 0x7fffu) + 1;
                pending = is_m;
                if (pending && dist > o) { bad = true; pending = false; }
                src = o - dist;
                ready_at = src + len < o ? src + len : o;  // the match needs the bytes below this
            } else pending = false;
            if (__any(bad)) { if (lane == 0) s_bad = 1; pending = false; }
            if (lane == 0) s_base[(step + 1) & 1] = base + total;  // (the slot of round step - 1, which has been read)
        } else if (!bad_now && step >= 1 && ((step - 1) & 1) == wave) {
            // ---- half 2 of round `step - 1`: the matches, in as few sub-rounds as their dependencies allow
            unsigned long long pm = __ballot(pending);
            while (pm) {
                const int first = __ffsll((long long)pm) - 1;
                const uint32_t hwm = (uint32_t)__builtin_amdgcn_readlane((int)o, __builtin_amdgcn_readfirstlane(first));
                wave_sync();
                if (pending && ready_at <= hwm) {
                    if (dist >= 8) {  // eight bytes per load and store (the source of a slice lies at least 8 bytes below its target)
                        if (len <= 8) {
                            // most matches of a BAM stream: one load, two overlapping stores (bytes 0..3 and len-4..len-1; a match of
                            // three bytes: 0..1 and 2) instead of a byte loop whose every step waits for its own load
                            unsigned long long w;
                            __builtin_memcpy(&w, lz_win + src, 8);  // (the bytes behind the match are read, not stored)
                            if (len >= 4) {
                                const uint32_t lo4 = (uint32_t)w, hi4 = (uint32_t)(w >> (8 * (len - 4)));
                                __builtin_memcpy(lz_win + o, &lo4, 4); __builtin_memcpy(lz_win + o + len - 4, &hi4, 4);
                            } else {
                                const uint16_t lo2 = (uint16_t)w; const uint8_t b2 = (uint8_t)(w >> 16);
                                __builtin_memcpy(lz_win + o, &lo2, 2); lz_win[o + 2] = b2;
                            }
                        } else {
                            uint32_t k = 0;
                            for (; k + 8 <= len; k += 8) { unsigned long long w; __builtin_memcpy(&w, lz_win + src + k, 8); __builtin_memcpy(lz_win + o + k, &w, 8); }
                            if (k < len) {  // the tail the same way: the last eight bytes of the match, overlapping what is already there
                                unsigned long long w;
                                __builtin_memcpy(&w, lz_win + src + len - 8, 8); __builtin_memcpy(lz_win + o + len - 8, &w, 8);
                            }
                        }
                        pending = false;
                    }
                    // every byte comes from [src, src + min(dist, len)): final, so the reads of a slice go out together
                    uint32_t j = 0;  // k mod dist
                    for (uint32_t k = 0; pending && k < len; k += 8) {
                        uint8_t v[8];
                        uint32_t jj = j;
#pragma unroll
                        for (int q = 0; q < 8; ++q) { v[q] = lz_win[src + jj]; if (++jj == dist) jj = 0; }
                        j = jj;
#pragma unroll
                        for (int q = 0; q < 8; ++q) if (k + q < len) lz_win[o + k + q] = v[q];
                    }
                    pending = false;
                }
                pm = __ballot(pending);
            }
        }
        __syncthreads();
    }
    const uint32_t produced = s_base[nr & 1];
    if (s_bad || produced != blk.isize) { if (threadIdx.x == 0) atomicOr(&flags[0], 512); return; }
    uint8_t* out = outbuf + (blk.uoff - out_base);
    if ((((uintptr_t)out) & 15) == 0) {
        const uint32_t words = blk.isize >> 4;
        for (uint32_t w = threadIdx.x; w < words; w += 128) ((uint4*)out)[w] = ((const uint4*)lz_win)[w];
        for (uint32_t k = (words << 4) + threadIdx.x; k < blk.isize; k += 128) out[k] = lz_win[k];
    } else
        for (uint32_t k = threadIdx.x; k < blk.isize; k += 128) out[k] = lz_win[k];
}

// Record boundaries of the inflated stream, on the device: 8 KiB slices find their first boundary by validating a chain of
// plausible record headers (as the host reader does), walk from there, and a check kernel verifies that every slice ends
// exactly where the next one started (any disagreement sends the file through the host reader instead).
constexpr int REC_SLICE = 8192;
// LZ77 resolution WITHOUT a window in LDS (round 4, SQUID_RESOLVE_GLOBAL): one wave per BGZF block, the block's output written to its
// place in HBM as it is produced and the matches read from there (a block's 64 KiB sit in L2 while it is worked on).  A match costs a
// memory round trip instead of an LDS one, so a block takes ten times longer than in k_lz_resolve2 -- but the kernel needs no LDS and
// few registers: thousands of blocks are in flight instead of one per CU, and the 64 KB the LDS form reserves on every CU go to the token
// pass (five token waves per CU instead of three).  Same rounds of 64 tokens and the same readiness rule as k_lz_resolve2; between the
// stores of a sub-round and the loads of the next a workgroup-scope fence (one wave per workgroup on one CU: its L1 is the only cache in
// between, the fence is the wait for the stores).
// The same resolve with the bytes of a round assembled in LDS and stored as whole words (round 6).  k_lz_resolve3 writes a round's output where the
// tokens put it: three byte stores per literal token, two or three short stores per match -- 164 M write requests per GiB of output, 6.5 bytes each --, and reads
// the sources of its matches from memory even when the same round wrote them.  tools/l2_probe.cpp: a wave that writes 192 bytes per round and reads 8 bytes per lane
// from a few hundred bytes behind takes 1.4-2.6 us per round with byte stores and 0.7-1.2 us with 4- or 16-byte stores; the reads leave the L2 either way.
// Here a round's literals and the copies of its matches go to a per-wave staging area of RS_STAGE bytes (496, 512 with its slack: sixteen resolve waves beside sixteen token
// waves still fit a CU's 160 KB), matches whose source lies inside the round are copied there from the staging area itself, and the round leaves with one
// 4-byte store per lane.  A round of more than RS_STAGE bytes (long matches) takes k_lz_resolve3's way.
#ifndef SQ_RS_STAGE
#define SQ_RS_STAGE 496
#endif
constexpr int RS_STAGE = SQ_RS_STAGE;  // (+ 16 bytes of slack = one 512-byte LDS allocation at 496)
__global__ __launch_bounds__(64) void k_lz_resolve5(const uint32_t* tok, const int32_t* ntok, const InflBlock* blocks, int nblocks, unsigned long long out_base, uint8_t* outbuf, int32_t* flags) {
    __shared__ __attribute__((aligned(16))) uint8_t st_mem[RS_STAGE + 16];
    const InflBlock blk = blocks[blockIdx.x];
    const bool ok = rsv::resolve_block<RS_STAGE>(tok + blk.toff, ntok[blockIdx.x], outbuf + (blk.uoff - out_base), blk.isize, (wv::lds_u8*)st_mem);  // (sq_resolve.inc)
    if (!ok && threadIdx.x == 0) atomicOr(&flags[0], 512);
}
// (experiment: SQUID_RESOLVE_LDS=<bytes> of unused dynamic LDS per resolve wave = a cap on the waves a CU holds -- 160 KB / bytes)
static bool resolve_staged() { return std::getenv("SQUID_RESOLVE_STAGED") == nullptr || std::atoi(std::getenv("SQUID_RESOLVE_STAGED")) != 0; }  // (the default; 0: k_lz_resolve3.  Read per call: tests switch it)
static unsigned resolve_lds_pad() { static const unsigned v = std::getenv("SQUID_RESOLVE_LDS") ? (unsigned)std::atoi(std::getenv("SQUID_RESOLVE_LDS")) : 0u; return v; }
__global__ __launch_bounds__(64) void k_lz_resolve3(const uint32_t* tok, const int32_t* ntok, const InflBlock* blocks, int nblocks, unsigned long long out_base, uint8_t* outbuf, int32_t* flags) {
    const int lane = threadIdx.x;
    const InflBlock blk = blocks[blockIdx.x];
    const uint32_t* t = tok + blk.toff;
    const int n = ntok[blockIdx.x];
    uint8_t* out = outbuf + (blk.uoff - out_base);
    uint32_t base = 0;
    bool bad = false;
    uint32_t nxt = lane < n ? t[lane] : 0;
    for (int r0 = 0; r0 < n; r0 += 64) {
        const uint32_t tk = nxt;
        const int i = r0 + lane;
        if (i + 64 < n) nxt = t[i + 64];
        const bool valid = i < n, is_m = valid && (tk >> 31);
        const uint32_t nl = (tk >> 24) & 3u;
        const uint32_t len = !valid ? 0u : (is_m ? (tk >> 16) & 0x1ffu : (nl ? nl : 1u));
        const uint32_t inc = wave_scan_incl(len);
        const uint32_t o = base + inc - len;
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        if (base + total > blk.isize) { bad = true; break; }  // (uniform)
        if (valid && !is_m) { out[o] = (uint8_t)tk; if (len > 1) out[o + 1] = (uint8_t)(tk >> 8); if (len > 2) out[o + 2] = (uint8_t)(tk >> 16); }
        const uint32_t dist = (tk & 0x7fffu) + 1;
        bool pending = is_m;
        if (__any(pending && dist > o)) { bad = true; break; }
        const uint32_t src = o - dist;
        const uint32_t ready_at = src + len < o ? src + len : o;  // the match needs the bytes below this
        unsigned long long pm = __ballot(pending);
        while (pm) {
            const int first = __ffsll((long long)pm) - 1;
            const uint32_t hwm = (uint32_t)__builtin_amdgcn_readlane((int)o, __builtin_amdgcn_readfirstlane(first));  // everything below the first pending match is final
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // ... and written
            if (pending && ready_at <= hwm) {
                if (dist >= 8) {
                    if (len <= 8) {
                        unsigned long long w;
                        __builtin_memcpy(&w, out + src, 8);  // (the bytes behind the match are read, not stored; the block buffers carry slack behind their last block)
                        if (len >= 4) {
                            const uint32_t lo4 = (uint32_t)w, hi4 = (uint32_t)(w >> (8 * (len - 4)));
                            __builtin_memcpy(out + o, &lo4, 4); __builtin_memcpy(out + o + len - 4, &hi4, 4);
                        } else {
                            const uint16_t lo2 = (uint16_t)w; const uint8_t b2 = (uint8_t)(w >> 16);
                            __builtin_memcpy(out + o, &lo2, 2); out[o + 2] = b2;
                        }
                    } else {
                        uint32_t k = 0;
                        for (; k + 8 <= len; k += 8) { unsigned long long w; __builtin_memcpy(&w, out + src + k, 8); __builtin_memcpy(out + o + k, &w, 8); }
                        if (k < len) { unsigned long long w; __builtin_memcpy(&w, out + src + len - 8, 8); __builtin_memcpy(out + o + len - 8, &w, 8); }
                    }
                } else {
                    // every byte comes from [src, src + min(dist, len)): final
                    uint32_t j = 0;
                    for (uint32_t k = 0; k < len; k += 8) {
                        uint8_t v[8];
                        uint32_t jj = j;
#pragma unroll
                        for (int q = 0; q < 8; ++q) { v[q] = out[src + jj]; if (++jj == dist) jj = 0; }
                        j = jj;
#pragma unroll
                        for (int q = 0; q < 8; ++q) if (k + q < len) out[o + k + q] = v[q];
                    }
                }
                pending = false;
            }
            pm = __ballot(pending);
        }
        base += total;
    }
    if (bad || base != blk.isize) { if (lane == 0) atomicOr(&flags[0], 512); }
}

struct RecScan { const uint8_t* u; unsigned long long begin, limit; int nref; int first_ref, end_ref, with_unplaced; /* first_ref < 0: keep everything */ };
__device__ __forceinline__ int ld32u(const uint8_t* p) { return (int)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24)); }
__device__ __forceinline__ long rec_plausible(const RecScan& S, unsigned long long p) {
    if (S.limit - p < 36) return -1;
    const uint8_t* q = S.u + p;
    const int bs = ld32u(q);
    if (bs < 34 || bs > (1 << 26)) return -1;
    const int refid = ld32u(q + 4), pos = ld32u(q + 8), mrefid = ld32u(q + 24), mpos = ld32u(q + 28), lseq = ld32u(q + 20);
    const int lname = q[12], ncig = q[16] | (q[17] << 8);
    if (refid < -1 || refid >= S.nref || mrefid < -1 || mrefid >= S.nref || pos < -1 || mpos < -1 || lseq < 0 || lname < 1) return -1;
    const unsigned long long need = 32ull + lname + 4ull * ncig + ((unsigned long long)lseq + 1) / 2 + lseq;
    if (need > (unsigned long long)bs) return -1;
    if (p + 4 + 32 + lname <= S.limit && q[4 + 32 + lname - 1] != 0) return -1;
    // the optional fields must tile the rest of the record exactly (a header read one or two bytes early can pass
    // everything above and even continue into true records: its "fields" are 50 KB of other records and never tile)
    if (p + 4 + (unsigned long long)bs <= S.limit) {
        const uint8_t *a = q + 4 + need, *e = q + 4 + bs;
        while (a < e) {
            if (e - a < 3) return -1;
            const uint8_t ty = a[2];
            const bool alpha0 = (a[0] >= 'A' && a[0] <= 'Z') || (a[0] >= 'a' && a[0] <= 'z');
            const bool alnum1 = (a[1] >= 'A' && a[1] <= 'Z') || (a[1] >= 'a' && a[1] <= 'z') || (a[1] >= '0' && a[1] <= '9');
            if (!alpha0 || !alnum1) return -1;
            const uint8_t* v = a + 3;
            unsigned long long sz;
            if (ty == 'A' || ty == 'c' || ty == 'C') sz = 1;
            else if (ty == 's' || ty == 'S') sz = 2;
            else if (ty == 'i' || ty == 'I' || ty == 'f') sz = 4;
            else if (ty == 'Z' || ty == 'H') { const uint8_t* z = v; while (z < e && *z) ++z; if (z >= e) return -1; sz = (unsigned long long)(z - v) + 1; }
            else if (ty == 'B') {
                if (e - v < 5) return -1;
                const uint8_t st = v[0];
                const unsigned long long es = (st == 'c' || st == 'C') ? 1 : ((st == 's' || st == 'S') ? 2 : ((st == 'i' || st == 'I' || st == 'f') ? 4 : 0));
                if (!es) return -1;
                sz = 5 + es * (unsigned long long)(uint32_t)ld32u(v + 1);
            } else return -1;
            if (sz > (unsigned long long)(e - v)) return -1;
            a = v + sz;
        }
    }
    return bs;
}
// One WAVE per slice (round 6; one thread per slice until then: a serial walk over the ~130 byte positions in front of the slice's first
// record, every step a dependent load -- 1 ms per 512 MB batch, as long as a third of the token pass): the 64 lanes test 64 consecutive positions
// at once, each with the same chain test as before, and the lowest position that passes wins -- the same answer, in two or three rounds.
__global__ __launch_bounds__(64) void k_rec_sync(RecScan S, long long nslices, int synced, long long* sync) {
    const long long s = (long long)blockIdx.x;
    const int lane = (int)threadIdx.x;
    if (s >= nslices) return;
    const unsigned long long lo = S.begin + (unsigned long long)s * REC_SLICE, hi = lo + REC_SLICE < S.limit ? lo + REC_SLICE : S.limit;
    if (s == 0 && synced) { if (lane == 0) sync[0] = (long long)S.begin; return; }
    long long found = -1;
    for (unsigned long long base = lo; base < hi; base += 64) {
        const unsigned long long p = base + (unsigned long long)lane;
        bool good = false;
        if (p < hi) {
            unsigned long long q = p;
            int ok = 0;
            for (;;) {
                const long bs = rec_plausible(S, q);
                if (bs < 0) break;
                if (q + 4 + (unsigned long long)bs > S.limit) { if (ok >= 2) good = true; break; }
                q += 4 + (unsigned long long)bs;
                if (++ok == 4 || q == S.limit) { good = true; break; }
            }
        }
        const unsigned long long m = __ballot(good);
        if (m) { found = (long long)(base + (unsigned long long)(__ffsll((long long)m) - 1)); break; }
    }
    if (lane == 0) sync[s] = found;
}
__device__ __forceinline__ bool rec_owned(const RecScan& S, int refid) {
    if (S.first_ref < 0) return true;
    if (refid < 0) return S.with_unplaced != 0;
    return refid >= S.first_ref && refid < S.end_ref;
}
template <bool EMIT>
__global__ void k_rec_walk(RecScan S, long long nslices, const long long* sync, int32_t* count, long long* end_p, const int32_t* base, unsigned long long* offs) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslices) return;
    const unsigned long long lo = S.begin + (unsigned long long)s * REC_SLICE, hi = lo + REC_SLICE < S.limit ? lo + REC_SLICE : S.limit;
    long long p0 = sync[s];
    int n = 0;
    unsigned long long p = p0 < 0 ? hi : (unsigned long long)p0;
    int32_t at = EMIT ? base[s] : 0;
    if (p0 >= 0)
        while (p < hi) {
            if (S.limit - p < 4) break;
            const int bs = ld32u(S.u + p);
            if (bs < 32 || S.limit - p < 4ull + (unsigned long long)bs) break;  // incomplete (or corrupt: the parser will say so)
            if (rec_owned(S, ld32u(S.u + p + 4))) { if (EMIT) offs[at + n] = p; ++n; }
            p += 4ull + (unsigned long long)bs;
        }
    if (!EMIT) { count[s] = n; end_p[s] = p0 < 0 ? -1 : (long long)p; }
}
__global__ void k_rec_check(long long nslices, const long long* sync, const long long* end_p, int32_t* flags, long long* tail) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslices) return;
    // the walk of the last slice that has a boundary stops at the incomplete tail of the range (slices are chained, so that is the largest stop)
    if (sync[s] >= 0 && (s + 1 == nslices || sync[s + 1] < 0)) atomicMax((unsigned long long*)tail, (unsigned long long)end_p[s]);
    if (s == 0 || sync[s] < 0) return;  // (a slice without a boundary lies inside a record longer than a slice, or behind the last complete one)
    // a slice must start where the last slice before it that has a boundary stopped (the slices in between lie inside one
    // long record)
    long long t = s - 1;
    while (t >= 0 && sync[t] < 0) --t;
    if (t >= 0 && end_p[t] != sync[s]) atomicOr(&flags[0], 1024);
}

// ================================================================================================ host wrappers
static inline dim3 grid_for(int64_t n, int threads) { return dim3((unsigned)((n + threads - 1) / threads)); }

// HIP-event bracket on the library stream.  Events come from a pool and are only read back by dev_flush_timers()
// (called at the end of every ABI call), so timing a kernel never stalls the host.
struct EvTimer {
    sq_ctx* c; int slot; bool on; hipStream_t st;
    EvTimer(sq_ctx* c, const char* name, double bytes, hipStream_t on_stream = nullptr) : c(c), on(true), st(on_stream ? on_stream : c->stream) {
        DeviceRecords& D = *c->dev;
        std::lock_guard<std::mutex> lk(D.ev_mu);
        if (D.ev_used == D.ev_pool.size()) {
            hipEvent_t a = nullptr, b = nullptr;
            (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            D.ev_pool.push_back(std::make_pair(a, b));
        }
        slot = (int)D.ev_used++;
        D.ev_pending.push_back(DeviceRecords::Pending{name, bytes, slot});
        (void)hipEventRecord(D.ev_pool[slot].first, st);
    }
    void stop() {
        if (!on) return;
        on = false;
        std::lock_guard<std::mutex> lk(c->dev->ev_mu);
        (void)hipEventRecord(c->dev->ev_pool[slot].second, st);
    }
    ~EvTimer() { stop(); }
};
void dev_flush_timers(sq_ctx* c) {
    if (!c->dev) return;
    DeviceRecords& D = *c->dev;
    // per name: sum of the launch durations, and the time during which at least one launch of the name was running (the launches of
    // the BGZF reader sit on several streams and overlap) -- from the launches' start / end offsets against the first event of the batch
    struct Iv { const char* name; float s, e; };
    std::vector<Iv> iv;
    iv.reserve(D.ev_pending.size());
    hipEvent_t origin = D.ev_pending.empty() ? nullptr : D.ev_pool[D.ev_pending.front().slot].first;
    for (const DeviceRecords::Pending& p : D.ev_pending) {
        float ms = 0, s0 = 0;
        (void)hipEventSynchronize(D.ev_pool[p.slot].second);
        (void)hipEventElapsedTime(&ms, D.ev_pool[p.slot].first, D.ev_pool[p.slot].second);
        c->timer.add(p.name, ms, p.bytes);
        if (hipEventElapsedTime(&s0, origin, D.ev_pool[p.slot].first) != hipSuccess) s0 = 0;  // (may be negative: an event of another stream recorded earlier)
        iv.push_back(Iv{p.name, s0, s0 + ms});
    }
    std::sort(iv.begin(), iv.end(), [](const Iv& a, const Iv& b) { return a.name != b.name ? a.name < b.name : a.s < b.s; });
    for (size_t i = 0; i < iv.size();) {
        size_t j = i;
        double busy = 0, lo = iv[i].s, hi = iv[i].e;
        for (; j < iv.size() && iv[j].name == iv[i].name; ++j) {
            if (iv[j].s > hi) { busy += hi - lo; lo = iv[j].s; hi = iv[j].e; }
            else if (iv[j].e > hi) hi = iv[j].e;
        }
        busy += hi - lo;
        c->timer.add_busy(iv[i].name, busy);
        if (!std::strcmp(iv[i].name, "k_inflate_tok2") && j - i >= 8) {
            // how many token passes the runtime really ran side by side (its hardware queues: each buffer set has a stream of its own and
            // kernels that share a queue run one after the other).  The launches are sorted by start; a sweep over starts and ends.
            std::vector<std::pair<float, int>> evs;
            for (size_t q = i; q < j; ++q) { evs.push_back(std::make_pair(iv[q].s, 1)); evs.push_back(std::make_pair(iv[q].e, -1)); }
            std::sort(evs.begin(), evs.end());
            int cur = 0, mx = 0;
            for (const auto& e : evs) { cur += e.second; mx = std::max(mx, cur); }
            if (c->ingest_dfile) c->counts.token_passes_side_by_side = std::max<int64_t>(c->counts.token_passes_side_by_side, mx);
            const char* env = std::getenv("GPU_MAX_HW_QUEUES");
            static std::atomic<bool> warned{false};
            // (only a STAGED file says something about the queues: a file that is still arriving feeds four token passes at a time by itself)
            if (mx <= 4 && c->ingest_dfile && j - i >= 10 && D.il_depth > 4 && (!env || std::atoi(env) > 4) && !warned.exchange(true))
                std::fprintf(stderr, "squid_hip: at most %d token passes ran side by side although %d buffer sets were in flight: the HIP runtime of this process works with its "
                                     "default of four hardware queues (GPU_MAX_HW_QUEUES=8 has to be in the environment BEFORE the process makes its first HIP call, INTEGRATION.md); "
                                     "the GPU reader is about 1.4x slower than it could be\n", mx, D.il_depth);
        }
        i = j;
    }
    D.ev_pending.clear();
    D.ev_used = 0;
}

// The ingest keeps the token passes of up to seven batches, the resolve of the batch in front of them and the record parse of the one
// before on the GPU side by side.  The HIP runtime maps all streams of one priority onto four hardware queues unless it is told otherwise
// when it initialises, and kernels that share a queue run one after the other: with four queues at most four token passes overlap and
// a C3 step takes 205 ms instead of 147.  The variable has to be set by the HOST PROGRAM before its first HIP call
// (`build/squid` does in main, `squid_amd/__init__.py` and `bench.py` at import; round 4 set it from a static constructor of this library --
// a setenv at dlopen time, unsafe beside other threads of the host and silent about hosts that had initialised HIP already).  What the
// library does instead: dev_flush_timers counts how many token passes really ran side by side (sq_counts.token_passes_side_by_side) and says
// so on stderr when that is four or fewer although more buffer sets were in flight.

int dev_create(sq_ctx* c) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(c, SQ_E_NODEVICE, "no HIP device visible; libsquid_hip has no CPU path");
    if (c->P.device < 0 || c->P.device >= ndev) return fail(c, SQ_E_NODEVICE, "device ordinal out of range");
    HIPCHK(hipSetDevice(c->P.device));
    HIPCHK(hipStreamCreate(&c->stream));
    c->dev = new DeviceRecords();
    HIPCHK(c->dev->flags.reserve(64));
    // The runtime loads this library's code object for the device when the first of its kernels is asked for, and two threads asking while
    // it is still being loaded is a race inside the runtime ("Cannot find Symbol with name ...", abort): seen once in some twenty processes
    // when round 4 let a helper thread do the asking beside the caller, and possible ever since sq_ingest_files decodes the chimeric BAM (and
    // launches its first kernel) on a thread of its own.  So the load happens HERE, on the creating thread, before any other thread exists.
    {
        hipFuncAttributes a;
        if (hipFuncGetAttributes(&a, (const void*)k_lz_resolve3) != hipSuccess) (void)hipGetLastError();
    }
    return SQ_OK;
}

void dev_destroy(sq_ctx* c) {
    if (!c->dev) return;
    DeviceRecords& D = *c->dev;
    D.refid.release(); D.pos.release(); D.mrefid.release(); D.mpos.release(); D.endpos.release(); D.b_refpos.release(); D.b_matchref.release();
    D.b_pack.release(); D.n_pack.release(); D.r_pack.release();
    D.flag.release(); D.totlen.release(); D.b_readpos.release(); D.b_matchread.release(); D.mapq.release(); D.aux.release(); D.blk_off.release();
    D.nm_blob.release(); D.nm_off.release(); D.g_win.release(); D.p2_list.release(); D.p2_words.release(); D.tile_cnt.release(); D.tile_K.release(); D.tile_zcnt2.release(); D.zc_v.release(); D.zc_K.release(); D.zc_refid.release(); D.zc_pos.release(); D.tile_ob.release(); D.zc_ob.release();
    D.tile_rank.release(); D.tile_zbase.release(); D.tile_zcnt.release(); D.z_idx.release(); D.z_chr.release(); D.z_right.release(); D.rc_cluster.release(); D.rc_pos.release(); D.rc_len.release(); D.p1_sc.release();
    D.tile_first.release(); D.tile_max.release(); D.r_break.release(); D.sum_items.release(); D.bp_before.release();
    D.cls.release(); D.keep.release(); D.prev1.release(); D.prev2.release(); D.rank1.release(); D.restoff.release();
    D.scratch_a.release(); D.scratch_b.release(); D.scratch_c.release(); D.spine.release();
    D.part_prev.release(); D.part_next.release(); D.b0_a.release(); D.b0_b.release(); D.b0_home.release(); D.fc_seg.release();
    D.srec.release(); D.rest_refpos.release(); D.rest_matchref.release();
    D.n_chr.release(); D.n_bucket.release();
    D.acc_a.release(); D.acc_b.release(); D.acc_c.release();
    D.h_key.release(); D.h_val.release(); D.flags.release(); D.bam_chunk.release(); D.bam_off.release(); D.chim_hash.release(); D.chim_off.release(); D.chim_len.release(); D.chim_blob.release(); D.chim_dead.release(); D.chim_slot_of.release(); D.chim_in_off.release(); D.chim_in_len.release(); D.parse_nblk.release(); D.parse_rel.release(); D.parse_first2.release();
    D.calib.release(); D.okey.release(); D.oval.release(); D.other64.release(); D.spine64.release(); D.okey64.release(); D.zflag.release();
    D.cl_chr.release(); D.trig.release(); D.cl_bucket.release();
    D.ord_e.release(); D.ord_o.release(); D.ord_v.release(); D.ord_me.release(); D.ord_mo.release(); D.g_i.release(); D.g_x.release(); D.g_d.release(); D.g_b.release();
    D.pin.release(); for (auto& st : D.il_set) { st.in.release(); st.tab.release(); st.tok.release(); st.lens.release(); st.ntok.release(); st.flags.release(); if (st.ready) (void)hipEventDestroy(st.ready); if (st.freed) (void)hipEventDestroy(st.freed); if (st.copied) (void)hipEventDestroy(st.copied); st.ready = st.freed = st.copied = nullptr; }
    for (auto& q : D.il_stream) if (q) { (void)hipStreamDestroy(q); q = nullptr; }
    for (auto& ps : D.il_post) { ps.out.release(); ps.big.release(); ps.rec_sync.release(); ps.rec_end.release(); ps.rec_cnt.release(); ps.rec_base.release(); ps.flags.release(); ps.spine.release(); ps.bam_off.release(); if (ps.carried) { (void)hipEventDestroy(ps.carried); ps.carried = nullptr; } }
    if (D.il_parse_stream) { (void)hipStreamDestroy(D.il_parse_stream); D.il_parse_stream = nullptr; }
    if (D.order_stream) { (void)hipStreamDestroy(D.order_stream); D.order_stream = nullptr; }
    if (D.il_host) { (void)hipHostFree(D.il_host); D.il_host = nullptr; }
    for (int t = 0; t < DeviceRecords::H2D_THREADS; ++t) {
        for (int b = 0; b < 2; ++b) { if (D.h2d_pin[t][b]) (void)hipHostFree(D.h2d_pin[t][b]); D.h2d_pin[t][b] = nullptr; if (D.h2d_ev[t][b]) (void)hipEventDestroy(D.h2d_ev[t][b]); D.h2d_ev[t][b] = nullptr; }
        if (D.h2d_stream[t]) (void)hipStreamDestroy(D.h2d_stream[t]);
        D.h2d_stream[t] = nullptr;
    }
    D.feed_pool.reset();
    for (int t = 0; t < DeviceRecords::FEED_THREADS_MAX; ++t) {
        for (int b = 0; b < 2; ++b) { if (D.feed_pin[t][b]) (void)hipHostFree(D.feed_pin[t][b]); D.feed_pin[t][b] = nullptr; if (D.feed_buf_ev[t][b]) (void)hipEventDestroy(D.feed_buf_ev[t][b]); D.feed_buf_ev[t][b] = nullptr; }
        if (D.feed_stream[t]) (void)hipStreamDestroy(D.feed_stream[t]);
        D.feed_stream[t] = nullptr;
    }
    for (hipEvent_t e : D.feed_piece_ev) (void)hipEventDestroy(e);
    D.feed_piece_ev.clear();
    D.stream_file.release();
    D.bgzf_out.release(); D.bgzf_carry.release(); D.staged.release(); D.rec_sync.release(); D.rec_end.release(); D.rec_cnt.release(); D.rec_base.release(); D.stripes.release(); D.bp_bucket.release(); D.bp_key.release(); D.bp_front.release(); D.depth_tiles.release(); D.tile_part.release(); D.bp_ev.release(); D.bp_end.release(); D.bp_valid.release();
    for (auto& e : D.ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    delete c->dev;
    c->dev = nullptr;
    if (c->stream) (void)hipStreamDestroy(c->stream);
    c->stream = nullptr;
}

int dev_append_records(sq_ctx* c, const sq_aln_batch* b) {
    DeviceRecords& D = *c->dev;
    hipStream_t s = c->stream;
    const size_t n0 = (size_t)D.n, nb0 = (size_t)D.nb, n1 = n0 + (size_t)b->n_rec, nb1 = nb0 + (size_t)b->n_blk;
    if (nb1 >= 0xffffffffull) return fail(c, SQ_E_CAPACITY, "more than 2^32 aligned blocks");
    for (int64_t i = 0; i < b->n_rec; ++i)
        if (b->blk_off[i + 1] - b->blk_off[i] > 256) return fail(c, SQ_E_CAPACITY, "a record with more than 256 aligned blocks");
#define GROW(buf, used, want) HIPCHK(D.buf.grow_keep(used, want, s))
    GROW(refid, n0, n1); GROW(pos, n0, n1); GROW(mrefid, n0, n1); GROW(mpos, n0, n1); GROW(endpos, n0, n1);
    GROW(flag, n0, n1); GROW(totlen, n0, n1); GROW(mapq, n0, n1); GROW(aux, n0, n1);
    GROW(blk_off, n0 ? n0 + 1 : 0, n1 + 1);
    GROW(b_refpos, nb0, nb1); GROW(b_matchref, nb0, nb1); GROW(b_readpos, nb0, nb1); GROW(b_matchread, nb0, nb1);
#undef GROW
#define UP(buf, src, off, cnt) if (cnt) HIPCHK(hipMemcpyAsync(D.buf.p + (off), (src), (cnt) * sizeof(*D.buf.p), hipMemcpyHostToDevice, s))
    size_t nr = (size_t)b->n_rec, nbk = (size_t)b->n_blk;
    UP(refid, b->refid, n0, nr); UP(pos, b->pos, n0, nr); UP(mrefid, b->mate_refid, n0, nr); UP(mpos, b->mate_pos, n0, nr); UP(endpos, b->end_pos, n0, nr);
    UP(flag, b->flag, n0, nr); UP(totlen, b->totlen, n0, nr); UP(mapq, b->mapq, n0, nr); UP(aux, b->aux, n0, nr);
    UP(b_refpos, b->b_refpos, nb0, nbk); UP(b_matchref, b->b_matchref, nb0, nbk); UP(b_readpos, b->b_readpos, nb0, nbk); UP(b_matchread, b->b_matchread, nb0, nbk);
#undef UP
    // block offsets are rebased onto the resident block arrays
    std::vector<uint32_t> off(nr + 1);
    for (size_t i = 0; i <= nr; ++i) off[i] = b->blk_off[i] - b->blk_off[0] + (uint32_t)nb0;
    HIPCHK(hipMemcpyAsync(D.blk_off.p + n0, off.data(), (nr + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    HIPCHK(D.b_pack.grow_keep(nb0, nb1, s));
    if (nb1 > nb0) hipLaunchKernelGGL(k_pack_blocks, dim3((unsigned)((nb1 - nb0 + 255) / 256)), dim3(256), 0, s, (int64_t)nb0, (int64_t)nb1, D.b_refpos.p, D.b_matchref.p, D.b_readpos.p, D.b_matchread.p, D.b_pack.p);
    HIPCHK(hipStreamSynchronize(s));
    D.n = (int64_t)n1;
    D.nb = (int64_t)nb1;
    c->counts.n_concordant = D.n;
    c->counts.n_blocks = D.nb;
    return SQ_OK;
}

void dev_clear_records(sq_ctx* c) {
    if (!c->dev) return;
    c->dev->n = 0; c->dev->nb = 0; c->dev->k1 = 0; c->dev->r_pack_n = 0; c->dev->p2_valid_n = -1;
}
// sq_release_reader_buffers: what the GPU reader keeps between ingests (every buffer is made again by the ingest that next needs it)
int dev_release_reader(sq_ctx* c) {
    if (!c->dev) return SQ_OK;
    DeviceRecords& D = *c->dev;
    HIPCHK(hipSetDevice(c->P.device));
    HIPCHK(hipDeviceSynchronize());
    D.stream_file.release(); D.staged.release();
    for (auto& st : D.il_set) { st.in.release(); st.tab.release(); st.tok.release(); st.lens.release(); st.ntok.release(); st.flags.release(); }
    for (auto& ps : D.il_post) { ps.out.release(); ps.big.release(); ps.rec_sync.release(); ps.rec_end.release(); ps.rec_cnt.release(); ps.rec_base.release(); ps.flags.release(); ps.spine.release(); ps.bam_off.release(); }
    D.bgzf_out.release(); D.bgzf_carry.release();
    for (auto& pr : D.feed_pin) for (auto& b : pr) if (b) { (void)hipHostFree(b); b = nullptr; }
    D.feed_pin_bytes = 0;
    return SQ_OK;
}
// sq_stage_bam: bytes != null copies a file into HBM; bytes == null returns the resident copy
// File bytes (a mapping of the page cache: pageable memory) to the device.  One hipMemcpy of pageable memory goes through the
// runtime's staging buffer on the calling thread, 15-18 GB/s here -- the longest chain of an ingest that does not find the file in
// HBM.  Four threads copy 16 MiB pieces into page-locked buffers of their own and send them off asynchronously (two buffers each, so
// the next piece is being copied while the last one travels).  Returns when all of it has arrived.
static int h2d_parallel(sq_ctx* c, uint8_t* dst, const uint8_t* src, size_t n) {
    DeviceRecords& D = *c->dev;
    constexpr int T = DeviceRecords::H2D_THREADS;
    constexpr size_t CH = (size_t)16 << 20;
    if (n < 8 * CH || std::getenv("SQUID_H2D_SERIAL")) { HIPCHK(hipMemcpy(dst, src, n, hipMemcpyHostToDevice)); return SQ_OK; }
    std::lock_guard<std::mutex> lk(D.h2d_mu);
    for (int t = 0; t < T; ++t) {
        if (!D.h2d_stream[t]) HIPCHK(hipStreamCreateWithFlags(&D.h2d_stream[t], hipStreamNonBlocking));
        for (int b = 0; b < 2; ++b) {
            if (!D.h2d_pin[t][b]) HIPCHK(hipHostMalloc((void**)&D.h2d_pin[t][b], CH, hipHostMallocDefault));
            if (!D.h2d_ev[t][b]) HIPCHK(hipEventCreateWithFlags(&D.h2d_ev[t][b], hipEventDisableTiming));
        }
    }
    std::atomic<size_t> next{0};
    std::atomic<int> bad{0};
    const int device = c->P.device;
    auto work = [&](int t) {
        if (hipSetDevice(device) != hipSuccess) { bad = 1; return; }
        bool used[2] = {false, false};
        int b = 0;
        for (;;) {
            const size_t off = next.fetch_add(CH);
            if (off >= n || bad.load()) break;
            const size_t len = std::min(CH, n - off);
            if (used[b] && hipEventSynchronize(D.h2d_ev[t][b]) != hipSuccess) { bad = 1; break; }
            std::memcpy(D.h2d_pin[t][b], src + off, len);
            if (hipMemcpyAsync(dst + off, D.h2d_pin[t][b], len, hipMemcpyHostToDevice, D.h2d_stream[t]) != hipSuccess || hipEventRecord(D.h2d_ev[t][b], D.h2d_stream[t]) != hipSuccess) { bad = 1; break; }
            used[b] = true;
            b ^= 1;
        }
        if (hipStreamSynchronize(D.h2d_stream[t]) != hipSuccess) bad = 1;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    if (bad.load()) { (void)hipGetLastError(); return fail(c, SQ_E_HIP, "host to device copy of the file bytes failed"); }
    return SQ_OK;
}

// ---- A file in the page cache -> HBM, as a stream that runs ahead of the GPU reader (dev_ingest_bgzf without a staged copy).
// T threads take the 8 MiB pieces of the file range in order: pread() into a page-locked buffer of their own (no mapping: no page
// tables to fill, the kernel copies straight from the page cache), an asynchronous copy to the piece's place in `stream_file`, an
// event per piece.  The thread that holds a piece also walks the BGZF block headers inside it (the bytes are in its buffer and in
// its cache): where the first header of a piece lies is known once the piece before has been walked, a chain of a few
// microseconds per link -- the block index grows as the file is read, nobody touches the file a second time.  The token pass of a
// batch waits for exactly the pieces that hold its blocks (wait_bytes); the copy of the rest of the file goes on meanwhile.
// What the round-3 path did instead: a mapping whose page tables a helper filled (24 GB/s), the header walk over it, and per
// batch one blocking copy by four threads in front of the batch's token pass -- 390 ms per C3 step where the same step from a
// copy already in HBM takes 256.
struct FileFeeder {
    const size_t P = std::getenv("SQUID_FEED_PIECE_MB") ? (size_t)std::max(1, std::atoi(std::getenv("SQUID_FEED_PIECE_MB"))) << 20 : (size_t)8 << 20;  // a piece
    sq_ctx* c;
    DeviceRecords& D;
    int fd = -1, T = 0;
    size_t file_n = 0, lo = 0, hi = 0, npieces = 0;
    bool started = false;
    std::atomic<size_t> next{0};
    std::atomic<bool> abort{false}, failed{false};
    std::unique_ptr<std::atomic<uint8_t>[]> issued;  // per piece: its copy is queued and its event recorded
    struct WalkState { size_t p = 0, total = 0; bool done = false; };
    bool walk = false;
    size_t stop = (size_t)-1;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<BgzfRange> found;             // (mu) blocks walked and not yet passed on
    bool walk_over = false;                   // (mu)
    WalkState end_state;                      // (mu) where the walk ended
    bool walk_bad = false;
    std::string what;
    double t_setup_ms = 0, walk_ms = 0;
    std::chrono::steady_clock::time_point t_start;
    std::unique_ptr<std::atomic<float>[]> t_issued;  // per piece: milliseconds after start() at which its copy was queued (SQUID_INGEST_TIMING)
    std::atomic<long long> us_pread{0}, us_bufwait{0};  // summed over the threads (SQUID_INGEST_TIMING)

    FileFeeder(sq_ctx* c_) : c(c_), D(*c_->dev) {}
    ~FileFeeder() { cancel(); if (fd >= 0) ::close(fd); }
    const uint8_t* dfile() const { return D.stream_file.p - lo; }  // "device address of file offset 0" (only offsets in [lo, hi) exist)

    int start(const char* path, size_t from, size_t upto, bool do_walk, size_t walk_p, size_t walk_total, size_t walk_stop) {
        const auto t0 = std::chrono::steady_clock::now();
        fd = ::open(path, O_RDONLY);
        struct stat st;
        if (fd < 0 || fstat(fd, &st) != 0) return fail(c, SQ_E_IO, std::string("cannot open bamfile ") + path);
        file_n = (size_t)st.st_size;
        lo = from / P * P;
        hi = std::min(file_n, upto);
        if (hi <= lo) return fail(c, SQ_E_ARG, "internal: empty file range");
        npieces = (hi - lo + P - 1) / P;
        static const int env_t = std::getenv("SQUID_FEED_THREADS") ? std::atoi(std::getenv("SQUID_FEED_THREADS")) : 0;
        T = env_t > 0 ? env_t : usable_cpus() / std::max(1, c->P.world_size);  // (the rank's share of the CPUs the process may really use: cgroup quota, not the CPUs it can see)
        // (while the pairing of a large chimeric BAM runs on the host threads -- sq_ingest_files, the chimeric records came through this
        // reader just before -- the file is not what the step waits for and the readers only take CPU time from what it does wait for:
        // a third of the share.  Dense config, load 1.11-1.27 s with the full share, 0.95-1.08 s with five readers, same boxes, interleaved)
        if (env_t <= 0 && c->chim_pairing_running) T = std::max(std::min(T, 4), T / 3);  // (never more than the share)
        T = std::max(2, std::min({T, 16, (int)DeviceRecords::FEED_THREADS_MAX, (int)npieces}));
        if (env_t > 0) T = std::max(1, std::min({env_t, (int)DeviceRecords::FEED_THREADS_MAX, (int)npieces}));
        if (npieces < 2) T = 1;
        HIPCHK(D.stream_file.reserve(hi - lo + 512));
        HIPCHK(hipMemsetAsync(D.stream_file.p + (hi - lo), 0, 512, c->stream));  // (the input rings of the token pass read up to 80 bytes ahead)
        while (D.feed_piece_ev.size() < npieces) { hipEvent_t e; HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); D.feed_piece_ev.push_back(e); }
        issued.reset(new std::atomic<uint8_t>[npieces]);
        t_issued.reset(new std::atomic<float>[npieces]);
        for (size_t j = 0; j < npieces; ++j) { issued[j].store(0, std::memory_order_relaxed); t_issued[j].store(0.f, std::memory_order_relaxed); }
        walk = do_walk; stop = walk_stop;
        end_state = WalkState{walk_p, walk_total, false};
        walk_over = !walk;
        t_setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        t_start = std::chrono::steady_clock::now();
        bufs.assign((size_t)std::min(2 * T, 2 * (int)DeviceRecords::FEED_THREADS_MAX), Buf{});  // page-locked buffers, two per reader: made by the issuer thread as it starts
        if (!D.feed_pool) D.feed_pool.reset(new FeedPool(DeviceRecords::FEED_THREADS_MAX + 2));
        started = true;
        // threads 0 .. T-1 read, thread T queues the copies, thread T + 1 walks the headers
        D.feed_pool->start(T + 1 + (walk ? 1 : 0), [this](int t) { if (t == T) issue(); else if (t == T + 1) walk_headers(); else work(t); });
        return SQ_OK;
    }
    // `failed` / `abort` are tested by the waiters under `mu` (more) and `bmu` (work): they are stored with BOTH mutexes held, so a waiter
    // is either in front of its test -- and sees the flag -- or already blocked -- and gets the notification (a store outside the mutexes
    // could land between a waiter's test and its block: the notification was lost and a reader slept for good)
    void raise(std::atomic<bool>& flag, const char* msg) {
        { std::lock_guard<std::mutex> lb(bmu); std::lock_guard<std::mutex> lk(mu); if (msg && what.empty()) what = msg; flag = true; }
        wake_all();
    }
    void fail_with(const char* msg) { raise(failed, msg); }
    void wake_all() { cv.notify_all(); bcv.notify_all(); }
    // The BGZF block headers, on a thread of its own: one small pread per block -- the 4 bytes in front of a header are the inflated size
    // of the block before, so one read yields both -- 1.8 M blocks a second, ahead of any copy.  (The first form of this walked the
    // headers inside the copy threads' buffers, piece after piece: where a piece's first header lies is known once the piece before has
    // been walked, and that chain put all sixteen threads in lockstep with whichever of them was waiting for a free buffer.)
    void walk_headers() {
        const auto t0 = std::chrono::steady_clock::now();
        WalkState w = end_state;
        std::vector<BgzfRange> batch;
        bool have_prev = false, bad = false;
        BgzfRange prev{};
        auto publish = [&](bool over) {
            std::lock_guard<std::mutex> lk(mu);
            found.insert(found.end(), batch.begin(), batch.end());
            batch.clear();
            end_state = w;
            if (over) { walk_over = true; walk_bad = bad; }
        };
        uint8_t h[64];
        for (;;) {
            if (abort.load() || failed.load()) { publish(true); break; }
            const bool more = !bad && w.p + 18 <= file_n && w.p <= stop;
            const size_t at = have_prev ? w.p - 4 : w.p, need = (more ? 18 : 0) + (have_prev ? 4 : 0);
            ssize_t got = 0;
            if (need) { got = ::pread(fd, h, std::min<size_t>(sizeof h, file_n - at), (off_t)at); if (got < (ssize_t)need) { fail_with("cannot read the bamfile"); publish(true); break; } }
            const uint8_t* d = h + (have_prev ? 4 : 0);
            if (have_prev) { std::memcpy(&prev.isize, h, 4); prev.uoff = w.total; w.total += prev.isize; batch.push_back(prev); have_prev = false; }
            if (!more) { w.done = true; publish(true); break; }
            if (d[0] != 0x1f || d[1] != 0x8b || !(d[3] & 4)) { bad = true; continue; }
            const uint32_t xlen = d[10] | (d[11] << 8);
            int bsize = -1;
            std::vector<uint8_t> extra;
            const uint8_t* x = d + 12;
            if (12 + (size_t)xlen > (size_t)got - (size_t)(d - h)) {  // (an extra field longer than the usual six bytes: fetch it whole)
                extra.resize(xlen);
                if (::pread(fd, extra.data(), xlen, (off_t)(w.p + 12)) != (ssize_t)xlen) { bad = true; continue; }
                x = extra.data();
            }
            for (size_t o = 0; o + 4 <= xlen;) {
                const uint32_t slen = x[o + 2] | (x[o + 3] << 8);
                if (x[o] == 'B' && x[o + 1] == 'C' && slen == 2 && o + 6 <= xlen) bsize = (x[o + 4] | (x[o + 5] << 8)) + 1;
                o += 4 + slen;
            }
            if (bsize < (int)(12 + xlen + 8) || w.p + (size_t)bsize > file_n) { bad = true; continue; }
            prev.coff = w.p + 12 + xlen; prev.clen = (uint32_t)(bsize - 12 - (int)xlen - 8);
            have_prev = true;
            w.p += (size_t)bsize;
            if (batch.size() >= 512) { publish(false); cv.notify_all(); }
        }
        cv.notify_all();
        walk_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    // ---- copy: reader threads only pread() into page-locked buffers; ONE thread talks to the HIP runtime (queues the copies, records the
    // events, polls for finished ones).  With sixteen threads each queueing copies and blocking in hipEventSynchronize for a free buffer,
    // the thread that launches the kernels found the runtime's locks taken most of the time: the first batches were queued 80-120 ms
    // into the ingest although their bytes had long arrived (measured), and the whole pipeline ran a third slower than from HBM.
    struct Buf { uint8_t* p = nullptr; size_t piece = 0; };
    std::vector<Buf> bufs;
    std::mutex bmu;
    std::condition_variable bcv;
    std::vector<int> free_bufs, filled_bufs;   // (bmu) indices into bufs
    size_t pieces_issued = 0;                  // (issuer only)
    void work(int t) {  // a reader
        auto tick = [](std::chrono::steady_clock::time_point& t0) { const auto now = std::chrono::steady_clock::now(); const long long us = std::chrono::duration_cast<std::chrono::microseconds>(now - t0).count(); t0 = now; return us; };
        (void)t;
        for (;;) {
            const size_t j = next.fetch_add(1);
            if (j >= npieces || abort.load() || failed.load()) break;
            static const long fail_at = std::getenv("SQUID_FEED_FAIL_AT") ? std::atol(std::getenv("SQUID_FEED_FAIL_AT")) : -1;  // (tests: a read error at that piece)
            if (fail_at >= 0 && (long)j == fail_at) { fail_with("cannot read the bamfile (injected by SQUID_FEED_FAIL_AT)"); break; }
            const size_t off = lo + j * P, len = std::min(P, hi - off);
            auto tk = std::chrono::steady_clock::now();
            int bi = -1;
            {
                std::unique_lock<std::mutex> lk(bmu);
                bcv.wait(lk, [&]() { return !free_bufs.empty() || abort.load() || failed.load(); });
                if (free_bufs.empty()) break;
                bi = free_bufs.back(); free_bufs.pop_back();
            }
            us_bufwait += tick(tk);
            uint8_t* buf = bufs[(size_t)bi].p;
            size_t got = 0;
            while (got < len) { const ssize_t r = ::pread(fd, buf + got, len - got, (off_t)(off + got)); if (r <= 0) break; got += (size_t)r; }
            us_pread += tick(tk);
            if (got < len) { fail_with("cannot read the bamfile"); bcv.notify_all(); break; }
            { std::lock_guard<std::mutex> lk(bmu); bufs[(size_t)bi].piece = j; filled_bufs.push_back(bi); }
            bcv.notify_all();
        }
    }
    void issue() {  // the one thread of the feeder that makes HIP calls
        if (hipSetDevice(c->P.device) != hipSuccess) { fail_with("hipSetDevice"); bcv.notify_all(); return; }
        constexpr int NS = 4;
        for (int q = 0; q < NS; ++q) if (!D.feed_stream[q] && hipStreamCreateWithFlags(&D.feed_stream[q], hipStreamNonBlocking) != hipSuccess) { fail_with("hipStreamCreate"); bcv.notify_all(); return; }
        // the buffers (kept by the context: the next read finds them); the readers start with the first one that exists -- pinning 256 MiB
        // takes 50-80 ms the first time and would otherwise stand in front of a cold start's first copy
        // (... so they are made one per turn of the loop below, between the copies of the pieces that the first ones already carry)
        size_t made = 0;
        if (D.feed_pin_bytes != P) {  // buffers of an earlier read with another piece size (SQUID_FEED_PIECE_MB changed inside the process): made again
            for (auto& pr : D.feed_pin) for (auto& b : pr) if (b) { (void)hipHostFree(b); b = nullptr; }
            D.feed_pin_bytes = P;
        }
        auto make_buffer = [&]() -> bool {
            const size_t i = made;
            if (!D.feed_pin[i / 2][i % 2] && hipHostMalloc((void**)&D.feed_pin[i / 2][i % 2], P, hipHostMallocDefault) != hipSuccess) { fail_with("hipHostMalloc"); bcv.notify_all(); return false; }
            if (!D.feed_buf_ev[i / 2][i % 2] && hipEventCreateWithFlags(&D.feed_buf_ev[i / 2][i % 2], hipEventDisableTiming) != hipSuccess) { fail_with("hipEventCreate"); bcv.notify_all(); return false; }
            bufs[i].p = D.feed_pin[i / 2][i % 2];
            { std::lock_guard<std::mutex> lk(bmu); free_bufs.push_back((int)i); }
            bcv.notify_all();
            ++made;
            return true;
        };
        for (int i = 0; i < 2 && made < bufs.size(); ++i) if (!make_buffer()) return;
        // copies on one stream finish in order: only the oldest copy of every stream is asked about (a poll of all thirty-two buffers
        // every few microseconds kept the runtime's lock busy for the thread that launches the kernels)
        // few copies queued at a time: the small device -> host read-backs of the batch loop (flags, record counts) travel on the same DMA
        // engines, and every 8 MiB piece queued in front of one of them is 160 us of waiting for the thread that drives the pipeline
        static const size_t max_inflight = std::getenv("SQUID_FEED_INFLIGHT") ? (size_t)std::max(1, std::atoi(std::getenv("SQUID_FEED_INFLIGHT"))) : 8;
        std::deque<int> inflight[NS];
        std::deque<int> waiting;  // filled, not yet queued
        size_t done = 0, rr = 0, n_inflight = 0;
        while (done < npieces && !abort.load() && !failed.load()) {
            if (made < bufs.size() && !make_buffer()) return;
            {
                std::unique_lock<std::mutex> lk(bmu);
                if (made == bufs.size() && filled_bufs.empty() && (waiting.empty() || n_inflight >= max_inflight)) bcv.wait_for(lk, std::chrono::microseconds(n_inflight ? 50 : 500));
                for (int bi : filled_bufs) waiting.push_back(bi);
                filled_bufs.clear();
            }
            std::vector<int> take;
            while (!waiting.empty() && n_inflight + take.size() < max_inflight) { take.push_back(waiting.front()); waiting.pop_front(); }
            for (int bi : take) {
                const size_t j = bufs[(size_t)bi].piece, off = lo + j * P, len = std::min(P, hi - off);
                const int q = (int)(rr++ % NS);
                hipStream_t st = D.feed_stream[q];
                if (hipMemcpyAsync(D.stream_file.p + (off - lo), bufs[(size_t)bi].p, len, hipMemcpyHostToDevice, st) != hipSuccess || hipEventRecord(D.feed_piece_ev[j], st) != hipSuccess) { fail_with("host to device copy of the file bytes failed"); break; }
                t_issued[j].store(std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_start).count(), std::memory_order_relaxed);
                issued[j].store(1, std::memory_order_release);
                inflight[q].push_back(bi);
                ++n_inflight;
            }
            bool freed = false;
            for (int q = 0; q < NS; ++q)
                while (!inflight[q].empty()) {
                    const int bi = inflight[q].front();
                    const hipError_t e = hipEventQuery(D.feed_piece_ev[bufs[(size_t)bi].piece]);
                    if (e == hipErrorNotReady) { (void)hipGetLastError(); break; }
                    if (e != hipSuccess) { fail_with("host to device copy of the file bytes failed"); break; }
                    { std::lock_guard<std::mutex> lk(bmu); free_bufs.push_back(bi); }
                    inflight[q].pop_front();
                    --n_inflight; ++done; freed = true;
                }
            if (freed) bcv.notify_all();
        }
        for (int q = 0; q < NS; ++q) (void)hipStreamSynchronize(D.feed_stream[q]);
        bcv.notify_all();
    }
    // the IndexMore of a streamed read: the blocks walked since the last call; false when the walk is over and everything has been passed on
    bool more(std::vector<BgzfRange>& v) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&]() { return !found.empty() || walk_over || failed.load() || abort.load(); });
        if (found.empty()) return false;
        v.insert(v.end(), found.begin(), found.end());
        found.clear();
        return true;
    }
    // the file bytes [from, to) must have arrived before anything queued on `s` after this call runs
    int wait_bytes(size_t from, size_t to, hipStream_t s) {
        if (to <= from) return SQ_OK;
        from = std::max(from, lo); to = std::min(to, hi);
        for (size_t j = (from - lo) / P; j <= (to - 1 - lo) / P && j < npieces; ++j) {
            while (!issued[j].load(std::memory_order_acquire)) {
                if (failed.load() || abort.load()) return fail(c, SQ_E_IO, error());
                std::this_thread::sleep_for(std::chrono::microseconds(20));
            }
            HIPCHK(hipStreamWaitEvent(s, D.feed_piece_ev[j], 0));
        }
        return SQ_OK;
    }
    std::string error() { std::lock_guard<std::mutex> lk(mu); return what.empty() ? std::string("the read of the bamfile was given up") : what; }
    void finish() {  // (idempotent) every piece copied and the walk over, or the threads told to stop
        if (started) { D.feed_pool->wait(); started = false; }
    }
    void cancel() { raise(abort, nullptr); finish(); }
};

int dev_stage_file(sq_ctx* c, const uint8_t* bytes, size_t n, const uint8_t** dptr) {
    DeviceRecords& D = *c->dev;
    HIPCHK(hipSetDevice(c->P.device));
    if (bytes) {
        HIPCHK(D.staged.reserve(n + 512));
        { const int rc = h2d_parallel(c, D.staged.p, bytes, n); if (rc) return rc; }
        HIPCHK(hipMemset(D.staged.p + n, 0, 512));
    }
    if (!D.staged.p) return fail(c, SQ_E_ARG, "no staged file");
    *dptr = D.staged.p;
    return SQ_OK;
}

// device copy of the chimeric QNAME set (sorted unique names incl. "", ledger B9) as an open-addressing table
int dev_upload_chim_names(sq_ctx* c) {
    DeviceRecords& D = *c->dev;
    const std::vector<std::string>& names = c->chim_names;
    uint32_t slots = 16;
    while (slots < 2 * names.size() + 2) slots <<= 1;
    std::vector<unsigned long long> hash(slots, 0);
    std::vector<uint32_t> off(slots, 0), len(slots, 0);
    std::vector<char> blob(1, 0);
    for (const std::string& nm : names) {
        unsigned long long h = 1469598103934665603ull;
        for (unsigned char ch : nm) { h ^= ch; h *= 1099511628211ull; }
        if (h == 0) h = 1;
        uint32_t s = chim_slot(h, slots - 1);
        while (hash[s]) s = (s + 1) & (slots - 1);
        hash[s] = h; off[s] = (uint32_t)blob.size(); len[s] = (uint32_t)nm.size();
        blob.insert(blob.end(), nm.begin(), nm.end());
    }
    HIPCHK(D.chim_hash.reserve(slots)); HIPCHK(D.chim_off.reserve(slots)); HIPCHK(D.chim_len.reserve(slots)); HIPCHK(D.chim_blob.reserve(blob.size()));
    HIPCHK(hipMemcpyAsync(D.chim_hash.p, hash.data(), slots * 8, hipMemcpyHostToDevice, c->stream)); HIPCHK(hipMemcpyAsync(D.chim_off.p, off.data(), slots * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(D.chim_len.p, len.data(), slots * 4, hipMemcpyHostToDevice, c->stream)); HIPCHK(hipMemcpyAsync(D.chim_blob.p, blob.data(), blob.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    D.chim_mask = names.empty() ? 0 : slots - 1;
    D.chim_provisional = false;
    if (D.chim_dead.p) HIPCHK(hipMemsetAsync(D.chim_dead.p, 0, D.chim_dead.cap, c->stream));
    return SQ_OK;
}

// The QNAME table straight from the decoded chimeric records (called on the helper thread of sq_ingest_files, on a stream of its own):
// `blob` = the name bytes of the batch, (off, len) one entry per usable record (+ one of length 0: the reference's set always holds "").
int dev_chim_begin(sq_ctx* c, const char* blob, size_t blob_bytes, const uint32_t* off, const uint32_t* len, size_t n) {
    HIPCHK(hipSetDevice(c->P.device));
    DeviceRecords& D = *c->dev;
    if (!D.chim_stream) HIPCHK(hipStreamCreateWithFlags(&D.chim_stream, hipStreamNonBlocking));
    hipStream_t s = D.chim_stream;
    uint32_t slots = 16;
    while ((size_t)slots < 2 * n + 2) slots <<= 1;
    HIPCHK(D.chim_hash.reserve(slots)); HIPCHK(D.chim_off.reserve(slots)); HIPCHK(D.chim_len.reserve(slots)); HIPCHK(D.chim_dead.reserve(slots));
    HIPCHK(D.chim_blob.reserve(blob_bytes + 1)); HIPCHK(D.chim_in_off.reserve(n)); HIPCHK(D.chim_in_len.reserve(n));
    HIPCHK(hipMemsetAsync(D.chim_hash.p, 0, (size_t)slots * 8, s));
    HIPCHK(hipMemsetAsync(D.chim_dead.p, 0, D.chim_dead.cap, s));
    if (blob_bytes) HIPCHK(hipMemcpyAsync(D.chim_blob.p, blob, blob_bytes, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(D.chim_in_off.p, off, n * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(D.chim_in_len.p, len, n * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_chim_insert, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, D.chim_blob.p, D.chim_in_off.p, D.chim_in_len.p, (int)n, slots - 1, D.chim_hash.p, D.chim_off.p, D.chim_len.p);
    HIPCHK(hipStreamSynchronize(s));
    D.chim_mask = slots - 1;
    D.chim_provisional = true;
    return SQ_OK;
}
// dev_chim_begin for records that came through the GPU reader: their names and flags are on the device (library stream; the caller has
// nothing else in flight)
int dev_chim_begin_captured(sq_ctx* c) {
    HIPCHK(hipSetDevice(c->P.device));
    DeviceRecords& D = *c->dev;
    hipStream_t s = c->stream;
    const size_t n = (size_t)D.n + 1;
    uint32_t slots = 16;
    while ((size_t)slots < 2 * n + 2) slots <<= 1;
    HIPCHK(D.chim_hash.reserve(slots)); HIPCHK(D.chim_off.reserve(slots)); HIPCHK(D.chim_len.reserve(slots)); HIPCHK(D.chim_dead.reserve(slots));
    HIPCHK(D.chim_blob.reserve(D.nm_bytes + 1)); HIPCHK(D.chim_in_off.reserve(n)); HIPCHK(D.chim_in_len.reserve(n));
    HIPCHK(hipMemsetAsync(D.chim_hash.p, 0, (size_t)slots * 8, s));
    HIPCHK(hipMemsetAsync(D.chim_dead.p, 0, D.chim_dead.cap, s));
    if (D.nm_bytes) HIPCHK(hipMemcpyAsync(D.chim_blob.p, D.nm_blob.p, D.nm_bytes, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(k_chim_entries, grid_for((int64_t)n, 256), dim3(256), 0, s, D.chim_blob.p, D.nm_off.p, (uint32_t)D.nm_bytes, D.flag.p, D.n, D.chim_in_off.p, D.chim_in_len.p);
    hipLaunchKernelGGL(k_chim_insert, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, D.chim_blob.p, D.chim_in_off.p, D.chim_in_len.p, (int)n, slots - 1, D.chim_hash.p, D.chim_off.p, D.chim_len.p);
    HIPCHK(hipStreamSynchronize(s));
    D.chim_mask = slots - 1;
    D.chim_provisional = true;
    return SQ_OK;
}
// after the pairing: the names of the fragments its PCR-duplicate removal dropped leave the set, and the records that matched one of them
// lose their bit (library stream; every record parse is over)
int dev_chim_finalize(sq_ctx* c, const std::vector<std::string>& dead_names) {
    DeviceRecords& D = *c->dev;
    if (!D.chim_provisional) return SQ_OK;
    D.chim_provisional = false;
    if (dead_names.empty()) return SQ_OK;
    hipStream_t s = c->stream;
    std::vector<char> blob;
    std::vector<uint32_t> off, len;
    for (const std::string& nm : dead_names) { off.push_back((uint32_t)blob.size()); len.push_back((uint32_t)nm.size()); blob.insert(blob.end(), nm.begin(), nm.end()); }
    const size_t n = off.size();
    DBuf<char> dblob; DBuf<uint32_t> doff, dlen;
    HIPCHK(dblob.reserve(blob.size() + 1)); HIPCHK(doff.reserve(n)); HIPCHK(dlen.reserve(n));
    HIPCHK(hipMemcpyAsync(dblob.p, blob.data(), blob.size(), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(doff.p, off.data(), n * 4, hipMemcpyHostToDevice, s)); HIPCHK(hipMemcpyAsync(dlen.p, len.data(), n * 4, hipMemcpyHostToDevice, s));
    const ChimSetView C{D.chim_mask, D.chim_hash.p, D.chim_off.p, D.chim_len.p, D.chim_blob.p, nullptr};
    hipLaunchKernelGGL(k_chim_mark_dead, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dblob.p, doff.p, dlen.p, (int)n, C, D.chim_dead.p);
    int32_t hf = 0;
    if (D.n > 0) {
        HIPCHK(D.flags.reserve(64));
        HIPCHK(hipMemsetAsync(D.flags.p, 0, 8 * 4, s));
        hipLaunchKernelGGL(k_chim_fixup, grid_for(D.n, 256), dim3(256), 0, s, D.n, D.chim_slot_of.p, D.chim_dead.p, D.aux.p, D.flags.p);
        D.r_pack_n = 0;  // (aux bytes changed)
        HIPCHK(hipMemcpyAsync(&hf, D.flags.p, 4, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    dblob.release(); doff.release(); dlen.release();
    if (hf & 256) return fail(c, SQ_E_ASSERT, "record without stored bases for an aligned block (reference asserts, ReadRec.cpp:64)");
    return SQ_OK;
}

// QNAMEs of the records of a parsed chunk (the chimeric BAM through K0, sq_ctx::capture_names): lengths, then -- behind a scan -- the bytes,
// without the terminating NUL, one name behind the other as the host decoder leaves them (HostBatch::names / name_off)
__global__ void k_name_len(const uint8_t* bam, const unsigned long long* rec_off, int64_t n, int32_t* len) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int lname = bam[rec_off[r] + 4 + 8];
    len[r] = lname > 0 ? lname - 1 : 0;
}
__global__ void k_name_copy(const uint8_t* bam, const unsigned long long* rec_off, int64_t n, const int32_t* rel, uint32_t base, char* blob, uint32_t* off_out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const uint8_t* src = bam + rec_off[r] + 4 + 32;
    const int lname = bam[rec_off[r] + 4 + 8], L = lname > 0 ? lname - 1 : 0;
    char* dst = blob + base + (uint32_t)rel[r];
    for (int i = 0; i < L; ++i) dst[i] = (char)src[i];
    off_out[r] = base + (uint32_t)rel[r];
}
// K0: parse `n_rec` BAM records of an inflated chunk on the device and append them to the resident SoA
static int parse_device(sq_ctx* c, const uint8_t* d_bam, size_t nbytes, const unsigned long long* d_off, int64_t n_rec, hipStream_t on = nullptr, int32_t* flags16 = nullptr, DBuf<int32_t>* scan_state = nullptr);
int dev_parse_append(sq_ctx* c, const uint8_t* bam, size_t nbytes, const unsigned long long* rec_off, int64_t n_rec) {
    if (n_rec == 0) return SQ_OK;
    HIPCHK(hipSetDevice(c->P.device));  // the file reader calls this from its sink thread (the current device is per thread)
    DeviceRecords& D = *c->dev;
    hipStream_t s = c->stream;
    HIPCHK(D.bam_chunk.reserve(nbytes + 64)); HIPCHK(D.bam_off.reserve((size_t)n_rec));
    HIPCHK(hipMemcpyAsync(D.bam_chunk.p, bam, nbytes, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(D.bam_off.p, rec_off, (size_t)n_rec * 8, hipMemcpyHostToDevice, s));
    return parse_device(c, D.bam_chunk.p, nbytes, D.bam_off.p, n_rec);
}
// the records at d_off[0..n_rec) of the inflated bytes d_bam (both in device memory) -> appended to the resident SoA
// (`on`, `flags16`, `scan_state`: the stream, a 16-int flag block and the scan state to use -- the file ingest parses on its own stream)
static int parse_device(sq_ctx* c, const uint8_t* d_bam, size_t nbytes, const unsigned long long* d_off, int64_t n_rec, hipStream_t on, int32_t* flags16, DBuf<int32_t>* scan_state) {
    if (!c->capture_names) { const int rc = chim_join_names(c); if (rc) return rc; }  // the QNAME table of the chimeric BAM (sq_ingest_files decodes that file meanwhile)
    DeviceRecords& D = *c->dev;
    hipStream_t s = on ? on : c->stream;
    int32_t* const fl = flags16 ? flags16 : D.flags.p;
    DBuf<int32_t>& spine = scan_state ? *scan_state : D.spine;
    const size_t n0 = (size_t)D.n, nb0 = (size_t)D.nb, n1 = n0 + (size_t)n_rec;
    HIPCHK(D.parse_nblk.reserve((size_t)n_rec)); HIPCHK(D.parse_rel.reserve((size_t)n_rec)); HIPCHK(D.parse_first2.reserve(2 * (size_t)n_rec));
    HIPCHK(hipMemsetAsync(fl, 0, 8 * 4, s));
    int32_t* tot = fl + 8;
    // a file ingest tells how much is still to come: size the arrays for all of it at the first growth instead of
    // reallocating (and copying) them chunk after chunk
    const double scale = (c->ingest_total_bytes && c->ingest_seen_bytes) ? 1.02 * (double)c->ingest_total_bytes / (double)c->ingest_seen_bytes : 0.0;
    const size_t rec_want = scale > 0 ? std::max(n1, (size_t)((double)n1 * scale) + 1024) : n1;
#define GROW(buf, used, want) HIPCHK(D.buf.grow_keep(used, want, s))
    if (D.refid.cap < n1) { GROW(refid, n0, rec_want); GROW(pos, n0, rec_want); GROW(mrefid, n0, rec_want); GROW(mpos, n0, rec_want); GROW(endpos, n0, rec_want);
                            GROW(flag, n0, rec_want); GROW(totlen, n0, rec_want); GROW(mapq, n0, rec_want); GROW(aux, n0, rec_want); }
    if (D.chim_slot_of.cap < n1) GROW(chim_slot_of, std::min(n0, D.chim_slot_of.cap), rec_want);
    if (D.blk_off.cap < n1 + 1) GROW(blk_off, n0 ? n0 + 1 : 0, rec_want + 1);
    ChimSetView C{D.chim_mask, D.chim_mask ? D.chim_hash.p : nullptr, D.chim_off.p, D.chim_len.p, D.chim_blob.p, D.chim_dead.p};
    if (c->capture_names) C = ChimSetView{0, nullptr, D.chim_off.p, D.chim_len.p, D.chim_blob.p, D.chim_dead.p};  // (the chimeric BAM itself: no name set to look its records up in)
    ParseParams P{(int)(signed char)(((c->P.phred_type ? 33 : 64) + c->P.min_phred) & 0xff), c->P.max_lowphred_len, c->P.min_mapqual, c->capture_names ? 1 : 0};
    // one pass over the inflated bytes: every field of a record, its block count and its first two blocks (k_parse_records); the blocks
    // go to their places behind the scan of the counts (k_parse_place)
    { EvTimer t(c, "k_parse_records", (double)nbytes + 64.0 * n_rec, s);
      const ParseOut O{D.refid.p + n0, D.pos.p + n0, D.mrefid.p + n0, D.mpos.p + n0, D.endpos.p + n0, D.flag.p + n0, D.totlen.p + n0, D.mapq.p + n0, D.aux.p + n0, D.chim_slot_of.p + n0, D.parse_nblk.p, D.parse_first2.p};
      // (staging size by the mean record length of the chunk -- the bytes in front of its first record count in, which only errs towards the larger size)
      const double need = 64.0 * 1.08 * (double)nbytes / (double)n_rec + 64.0;
      static const int force = std::getenv("SQUID_PARSE_LDS_KB") ? std::atoi(std::getenv("SQUID_PARSE_LDS_KB")) : 0;  // (tests: 18 / 22 / 28 / 40 / 63)
      const int kb = force ? force : need <= PARSE_LDS ? 18 : need <= 22528 ? 22 : need <= 28672 ? 28 : need <= 40960 ? 40 : 63;
      if (kb <= 18) hipLaunchKernelGGL(k_parse_records<PARSE_LDS>, grid_for(n_rec, PARSE_THREADS), dim3(PARSE_THREADS), 0, s, d_bam, nbytes, d_off, n_rec, C, P, O, fl);
      else if (kb <= 22) hipLaunchKernelGGL(k_parse_records<22528>, grid_for(n_rec, PARSE_THREADS), dim3(PARSE_THREADS), 0, s, d_bam, nbytes, d_off, n_rec, C, P, O, fl);
      else if (kb <= 28) hipLaunchKernelGGL(k_parse_records<28672>, grid_for(n_rec, PARSE_THREADS), dim3(PARSE_THREADS), 0, s, d_bam, nbytes, d_off, n_rec, C, P, O, fl);
      else if (kb <= 40) hipLaunchKernelGGL(k_parse_records<40960>, grid_for(n_rec, PARSE_THREADS), dim3(PARSE_THREADS), 0, s, d_bam, nbytes, d_off, n_rec, C, P, O, fl);
      else hipLaunchKernelGGL(k_parse_records<65536 - 64>, grid_for(n_rec, PARSE_THREADS), dim3(PARSE_THREADS), 0, s, d_bam, nbytes, d_off, n_rec, C, P, O, fl);
      HIPCHK((device_scan<OpSum, true>(s, n_rec, FArrN{D.parse_nblk.p}, D.parse_rel.p, spine, tot))); }
    int32_t nblk_total = 0;
    HIPCHK(hipMemcpyAsync(&nblk_total, tot, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const size_t nb1 = nb0 + (size_t)nblk_total;
    if (nb1 >= 0xffffffffull) return fail(c, SQ_E_CAPACITY, "more than 2^32 aligned blocks");
    const size_t blk_want = scale > 0 ? std::max(nb1 + 1, (size_t)((double)(nb1 + 1) * scale) + 1024) : nb1 + 1;
    if (D.b_refpos.cap < nb1 + 1) { GROW(b_refpos, nb0, blk_want); GROW(b_matchref, nb0, blk_want); GROW(b_readpos, nb0, blk_want); GROW(b_matchread, nb0, blk_want); }
    if (D.b_pack.cap < nb1 + 1) GROW(b_pack, nb0, blk_want);
#undef GROW
    { EvTimer t(c, "k_parse_place", 44.0 * n_rec + 28.0 * nblk_total, s);
      hipLaunchKernelGGL(k_parse_place, grid_for(n_rec, 256), dim3(256), 0, s, d_bam, d_off, n_rec, D.parse_nblk.p, D.parse_rel.p, D.parse_first2.p, (uint32_t)nb0, D.totlen.p + n0, D.blk_off.p + n0,
                         D.b_refpos.p, D.b_matchref.p, D.b_readpos.p, D.b_matchread.p, D.b_pack.p); }
    const uint32_t endoff = (uint32_t)nb1;
    HIPCHK(hipMemcpyAsync(D.blk_off.p + n1, &endoff, 4, hipMemcpyHostToDevice, s));
    int32_t hf = 0;
    HIPCHK(hipMemcpyAsync(&hf, fl, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (hf & 128) return fail(c, SQ_E_IO, "corrupt BAM record");
    if (hf & 2048) return fail(c, SQ_E_CAPACITY, "a read longer than 65535 bases or with more than 256 aligned blocks (the record layout keeps 16-bit read offsets)");
    if (hf & 256) return fail(c, SQ_E_ASSERT, "record without stored bases for an aligned block (reference asserts, ReadRec.cpp:64)");
    if (c->capture_names) {
        if (n0 == 0) D.nm_bytes = 0;
        { EvTimer t(c, "k_names", (double)n_rec * 48.0, s);
          hipLaunchKernelGGL(k_name_len, grid_for(n_rec, 256), dim3(256), 0, s, d_bam, d_off, n_rec, D.parse_nblk.p);
          HIPCHK((device_scan<OpSum, true>(s, n_rec, FArrN{D.parse_nblk.p}, D.parse_rel.p, spine, tot))); }
        int32_t nm_total = 0;
        HIPCHK(hipMemcpyAsync(&nm_total, tot, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (D.nm_bytes + (size_t)nm_total >= 0xffffffffull) return fail(c, SQ_E_CAPACITY, "more than 4 GB of read names");
        HIPCHK(D.nm_blob.grow_keep(D.nm_bytes, std::max<size_t>(D.nm_bytes + (size_t)nm_total + 64, rec_want * 16), s));
        HIPCHK(D.nm_off.grow_keep(n0, rec_want + 1, s));
        hipLaunchKernelGGL(k_name_copy, grid_for(n_rec, 256), dim3(256), 0, s, d_bam, d_off, n_rec, D.parse_rel.p, (uint32_t)D.nm_bytes, D.nm_blob.p, D.nm_off.p + n0);
        HIPCHK(hipStreamSynchronize(s));  // (the chunk's bytes belong to the caller again)
        D.nm_bytes += (size_t)nm_total;
    }
    D.n = (int64_t)n1;
    D.nb = (int64_t)nb1;
    c->counts.n_concordant = D.n;
    c->counts.n_blocks = D.nb;
    return SQ_OK;
}

// Tuning entry (tools/tok_bench.py): the token pass and the resolve of the first blocks of a BGZF file, each kernel ALONE on the device and
// timed with HIP events -- what the reader's overlapped launches cannot show.  variant: CH * 100 + PB of k_inflate_spec (51211, 25610, ...),
// 2 = k_inflate_tok2.  check != 0: the resolved bytes of every block against zlib.  out[0] token pass ms, out[1] resolve ms (averages over
// reps), out[2] inflated bytes, out[3] compressed bytes, out[4] blocks, out[5] tokens, out[6] blocks that differ from zlib.
int dev_token_bench(sq_ctx* c, const char* path, int variant, int max_blocks, int reps, int check, double* out) {
    HIPCHK(hipSetDevice(c->P.device));
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(c, SQ_E_IO, std::string("cannot open ") + path);
    std::vector<uint8_t> raw;
    std::vector<InflBlock> tab;
    unsigned long long uoff = 0;
    const bool spec = variant != 2;
    for (int b = 0; b < max_blocks; ++b) {
        uint8_t h[18];
        if (std::fread(h, 1, 18, f) != 18) break;
        const size_t bsize = (size_t)(h[16] | (h[17] << 8)) + 1;
        const size_t at = raw.size();
        raw.resize(at + bsize);
        std::memcpy(raw.data() + at, h, 18);
        if (std::fread(raw.data() + at + 18, 1, bsize - 18, f) != bsize - 18) { raw.resize(at); break; }
        uint32_t isize; std::memcpy(&isize, raw.data() + at + bsize - 4, 4);
        tab.push_back(InflBlock{at + 18, (uint32_t)(bsize - 26), isize, uoff, 0});
        uoff += isize;
    }
    std::fclose(f);
    if (tab.empty()) return fail(c, SQ_E_IO, "no BGZF block read");
    unsigned long long slots = 0;
    for (InflBlock& ib : tab) { ib.toff = slots; slots += spec ? isp::tok_cap_spec(ib.isize, ib.clen) : t2_tok_cap(ib.isize); }
    const int nb = (int)tab.size();
    DBuf<uint8_t> d_in, d_out; DBuf<InflBlock> d_tab; DBuf<uint32_t> d_tok, d_lens; DBuf<int32_t> d_ntok, d_flags;
    struct Rel { DBuf<uint8_t>&a, &b; DBuf<InflBlock>& t; DBuf<uint32_t>&k, &l; DBuf<int32_t>&n, &fl; ~Rel() { a.release(); b.release(); t.release(); k.release(); l.release(); n.release(); fl.release(); } } rel{d_in, d_out, d_tab, d_tok, d_lens, d_ntok, d_flags};
    HIPCHK(d_in.reserve(raw.size() + 512)); HIPCHK(d_out.reserve((size_t)uoff + 512)); HIPCHK(d_tab.reserve((size_t)nb)); HIPCHK(d_tok.reserve((size_t)slots + 64));
    HIPCHK(d_ntok.reserve((size_t)nb)); HIPCHK(d_flags.reserve(16)); HIPCHK(d_lens.reserve(((size_t)nb + 128) * T2_LENS_WORDS));
    HIPCHK(hipMemcpy(d_in.p, raw.data(), raw.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_tab.p, tab.data(), (size_t)nb * sizeof(InflBlock), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(d_flags.p, 0, 64));
    HIPCHK(hipFuncSetAttribute((const void*)k_inflate_tok2<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * (int)T2_LDS_BYTES));
    hipStream_t s = c->stream;
    hipEvent_t e0, e1, e2;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1)); HIPCHK(hipEventCreate(&e2));
    double tok_ms = 0, res_ms = 0;
    for (int r = 0; r < reps + 1; ++r) {  // (the first round warms up)
        HIPCHK(hipEventRecord(e0, s));
        switch (variant) {
            case 2: hipLaunchKernelGGL(k_inflate_tok2<false>, dim3((nb + 63) / 64), dim3(64), T2_LDS_BYTES, s, d_in.p, d_tab.p, nb, d_flags.p, d_tok.p, d_ntok.p, d_lens.p, nullptr); break;
            case 51211: launch_inflate_spec<512, 11>(s, d_in.p, d_tab.p, nb, d_flags.p, d_tok.p, d_ntok.p); break;
            case 51210: launch_inflate_spec<512, 10>(s, d_in.p, d_tab.p, nb, d_flags.p, d_tok.p, d_ntok.p); break;
            case 25610: launch_inflate_spec<256, 10>(s, d_in.p, d_tab.p, nb, d_flags.p, d_tok.p, d_ntok.p); break;
            case 25611: launch_inflate_spec<256, 11>(s, d_in.p, d_tab.p, nb, d_flags.p, d_tok.p, d_ntok.p); break;
            case 38411: launch_inflate_spec<384, 11>(s, d_in.p, d_tab.p, nb, d_flags.p, d_tok.p, d_ntok.p); break;
            case 38410: launch_inflate_spec<384, 10>(s, d_in.p, d_tab.p, nb, d_flags.p, d_tok.p, d_ntok.p); break;
            case 35210: launch_inflate_spec<352, 10>(s, d_in.p, d_tab.p, nb, d_flags.p, d_tok.p, d_ntok.p); break;
            case 41610: launch_inflate_spec<416, 10>(s, d_in.p, d_tab.p, nb, d_flags.p, d_tok.p, d_ntok.p); break;
            case 44810: launch_inflate_spec<448, 10>(s, d_in.p, d_tab.p, nb, d_flags.p, d_tok.p, d_ntok.p); break;
            case 38409: launch_inflate_spec<384, 9>(s, d_in.p, d_tab.p, nb, d_flags.p, d_tok.p, d_ntok.p); break;
            case 32010: launch_inflate_spec<320, 10>(s, d_in.p, d_tab.p, nb, d_flags.p, d_tok.p, d_ntok.p); break;
            case 19210: launch_inflate_spec<192, 10>(s, d_in.p, d_tab.p, nb, d_flags.p, d_tok.p, d_ntok.p); break;
            case 102411: launch_inflate_spec<1024, 11>(s, d_in.p, d_tab.p, nb, d_flags.p, d_tok.p, d_ntok.p); break;
            case 25609: launch_inflate_spec<256, 9>(s, d_in.p, d_tab.p, nb, d_flags.p, d_tok.p, d_ntok.p); break;
            case 12809: launch_inflate_spec<128, 9>(s, d_in.p, d_tab.p, nb, d_flags.p, d_tok.p, d_ntok.p); break;
            case 12810: launch_inflate_spec<128, 10>(s, d_in.p, d_tab.p, nb, d_flags.p, d_tok.p, d_ntok.p); break;
            case 51209: launch_inflate_spec<512, 9>(s, d_in.p, d_tab.p, nb, d_flags.p, d_tok.p, d_ntok.p); break;
            default: return fail(c, SQ_E_ARG, "unknown token pass variant");
        }
        HIPCHK(hipEventRecord(e1, s));
        if (resolve_staged()) hipLaunchKernelGGL(k_lz_resolve5, dim3(nb), dim3(64), 0, s, d_tok.p, d_ntok.p, d_tab.p, nb, 0ull, d_out.p, d_flags.p + 4);
        else hipLaunchKernelGGL(k_lz_resolve3, dim3(nb), dim3(64), resolve_lds_pad(), s, d_tok.p, d_ntok.p, d_tab.p, nb, 0ull, d_out.p, d_flags.p + 4);
        HIPCHK(hipEventRecord(e2, s));
        HIPCHK(hipStreamSynchronize(s));
        float a = 0, b = 0;
        HIPCHK(hipEventElapsedTime(&a, e0, e1)); HIPCHK(hipEventElapsedTime(&b, e1, e2));
        if (r) { tok_ms += a; res_ms += b; }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
    if (variant == 51211 || variant == 25610) {  // where the cycles go: one more launch with the phase clock
        DBuf<unsigned long long> d_prof;
        HIPCHK(d_prof.reserve(16)); HIPCHK(hipMemset(d_prof.p, 0, 16 * 8));
        if (variant == 51211) { typedef isp::Lay<512, 11> Y; hipLaunchKernelGGL((k_inflate_spec_prof<512, 11>), dim3(nb), dim3(64), (size_t)Y::BYTES, s, d_in.p, d_tab.p, nb, d_flags.p, d_tok.p, d_ntok.p, d_prof.p); }
        else { typedef isp::Lay<256, 10> Y; hipLaunchKernelGGL((k_inflate_spec_prof<256, 10>), dim3(nb), dim3(64), (size_t)Y::BYTES, s, d_in.p, d_tab.p, nb, d_flags.p, d_tok.p, d_ntok.p, d_prof.p); }
        unsigned long long pr[9];
        HIPCHK(hipStreamSynchronize(s));
        HIPCHK(hipMemcpy(pr, d_prof.p, sizeof pr, hipMemcpyDeviceToHost));
        d_prof.release();
        double tot = 0; for (int i = 0; i < 6; ++i) tot += (double)pr[i];
        std::fprintf(stderr, "token pass %d, cycles of a block (s_memtime, 100 MHz ticks?): %.0f | headers %.1f %% tables %.1f %% window loads %.1f %% scans %.1f %% token writing %.1f %% rest %.1f %% | per block: %.1f deflate blocks, %.1f windows, %.2f scan rounds per window\n",
                     variant, tot / nb, 100.0 * pr[0] / tot, 100.0 * pr[1] / tot, 100.0 * pr[2] / tot, 100.0 * pr[3] / tot, 100.0 * pr[4] / tot, 100.0 * pr[5] / tot, (double)pr[8] / nb, (double)pr[6] / nb, (double)pr[7] / std::max<double>(1, (double)pr[6]));
    }
    int32_t fl[8];
    HIPCHK(hipMemcpy(fl, d_flags.p, sizeof fl, hipMemcpyDeviceToHost));
    std::vector<int32_t> nt((size_t)nb);
    HIPCHK(hipMemcpy(nt.data(), d_ntok.p, (size_t)nb * 4, hipMemcpyDeviceToHost));
    long long ntok = 0;
    for (int32_t v : nt) ntok += v;
    long bad = (fl[0] | fl[4]) ? -1 : 0;
    if (check && bad == 0) {
        std::vector<uint8_t> got((size_t)uoff), want;
        HIPCHK(hipMemcpy(got.data(), d_out.p, (size_t)uoff, hipMemcpyDeviceToHost));
        for (const InflBlock& b : tab) {
            want.resize(b.isize);
            z_stream zs; std::memset(&zs, 0, sizeof zs);
            inflateInit2(&zs, -15);
            zs.next_in = (Bytef*)(raw.data() + b.coff); zs.avail_in = b.clen; zs.next_out = want.data(); zs.avail_out = b.isize;
            inflate(&zs, Z_FINISH); inflateEnd(&zs);
            if (b.isize && std::memcmp(want.data(), got.data() + b.uoff, b.isize) != 0) ++bad;
        }
    }
    out[0] = tok_ms / std::max(1, reps); out[1] = res_ms / std::max(1, reps); out[2] = (double)uoff; out[3] = (double)raw.size(); out[4] = nb; out[5] = (double)ntok; out[6] = (double)bad;
    return SQ_OK;
}

// K-1 + K0 for a whole file (or a shard's block range): compressed bytes -> HBM, inflate, record boundaries, parse -- the
// host only indexes the BGZF blocks.  The range is streamed in batches of SQUID_TOK_CAP_MB of inflated bytes (fixed-size
// device buffers whatever the file size), up to eight in flight: while batch k is resolved and cut into records on the library stream
// and batch k - 1 is parsed on the parse stream, the batches behind are in the token pass on their own streams, queued by a planner
// thread as buffer sets come free (the compressed bytes are in HBM already or arrive there through the feeder).  The incomplete record at
// the end of a batch is carried in front of the next one.  Returns 2 when the device-side boundary check (or the
// inflate) is not satisfied: the records appended so far are dropped again and the caller takes the host reader.
int dev_ingest_bgzf(sq_ctx* c, const uint8_t* file, std::vector<BgzfRange>& blocks, size_t b0, size_t b1, size_t begin, bool synced, int nref, const IndexMore& index_more_in, size_t file_bytes, GpuFileSrc* src) {
    if (b1 <= b0) return SQ_OK;
    const auto w_entry = std::chrono::steady_clock::now();
    HIPCHK(hipSetDevice(c->P.device));
    DeviceRecords& D = *c->dev;
    hipStream_t s = c->stream;
    {   // the batch buffers take ~25 GB next to the record arrays: on a GPU that is short of memory the host reader runs instead
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < ((size_t)32 << 30)) { (void)hipGetLastError(); return 2; }
    }
    const int64_t n_save = D.n, nb_save = D.nb;
    auto give_up = [&]() { (void)hipDeviceSynchronize(); D.n = n_save; D.nb = nb_save; D.r_pack_n = std::min(D.r_pack_n, n_save); c->counts.n_concordant = D.n; c->counts.n_blocks = D.nb; return 2; };
    const bool report = std::getenv("SQUID_INGEST_TIMING") != nullptr, check = std::getenv("SQUID_INFLATE_CHECK") != nullptr;
    const auto w0 = std::chrono::steady_clock::now();
    auto since_ms = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    // a batch = 256 waves of the token pass (1 GiB inflated); five batches fill the 1280 token slots of the machine (five waves per CU).
    // Measured at C3 (round 4, eight sets): 512 MB 157 ms per step, 640 MB 141, 768 MB 137, 1 GiB 136, 1.25 GB 135
    // (round 6, wave-per-block token pass: a launch is over in a few milliseconds whatever its size, so the batches are as small as the chain
    // behind them -- resolve, boundaries, parse -- allows: 512 MB; from the page cache 156-161 ms per C3 step against 170-198 with 1 GiB)
    const bool tok_spec_early = std::getenv("SQUID_TOK_SPEC") == nullptr || std::atoi(std::getenv("SQUID_TOK_SPEC")) != 0;
    unsigned long long cap = std::getenv("SQUID_TOK_CAP_MB") ? (unsigned long long)std::atoll(std::getenv("SQUID_TOK_CAP_MB")) << 20 : (tok_spec_early ? 128ull : 256ull) * 64 * 65536;  // (made smaller below for a short range)
    if (report) std::fprintf(stderr, "GPU ingest: entry + %.1f ms: device chosen, memory asked about\n", since_ms(w_entry));
    HIPCHK(hipFuncSetAttribute((const void*)k_lz_resolve2, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 16));
    HIPCHK(hipFuncSetAttribute((const void*)k_inflate_tok2<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * (int)T2_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)k_inflate_tok2<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * (int)T2_LDS_BYTES));
    // token waves per workgroup (SQUID_TOK_WPB, measured in DESIGN.md): 1 with the LDS-free resolve -- five single-wave workgroups per CU --;
    // with the LDS resolve 3 (93 KB), which leave the 64 KB slot of a resolve workgroup free on every CU
    static const bool resolve_global = std::getenv("SQUID_RESOLVE_GLOBAL") == nullptr || std::atoi(std::getenv("SQUID_RESOLVE_GLOBAL")) != 0;  // k_lz_resolve3 (no LDS window); 0: k_lz_resolve2
    static const int tok_wpb = std::getenv("SQUID_TOK_WPB") ? std::max(1, std::min(5, std::atoi(std::getenv("SQUID_TOK_WPB")))) : (resolve_global ? 1 : 3);
    // round 6: the token pass as one wave per BGZF block (k_inflate_spec, sq_inflate_spec.inc); SQUID_TOK_SPEC=0 runs the lane-per-block pass
    const bool tok_spec = std::getenv("SQUID_TOK_SPEC") == nullptr || std::atoi(std::getenv("SQUID_TOK_SPEC")) != 0;
    static const bool tok_prof = std::getenv("SQUID_TOK_PROF") != nullptr;
    if (tok_prof) HIPCHK(D.tok_prof.reserve(8 * 4096));
    // (buffer sets in flight: the lane-per-block pass wants eight -- 1280 token waves resident --, the wave-per-block pass fills the machine from one
    // launch: three sets keep the token pass a batch or two ahead of the resolve)
    const int il_depth = std::getenv("SQUID_IL_DEPTH") ? std::max(2, std::min((int)DeviceRecords::IL_DEPTH_MAX, std::atoi(std::getenv("SQUID_IL_DEPTH")))) : (tok_spec ? 3 : (resolve_global ? 8 : 5));
    D.il_depth = il_depth;
    // The resolve of a batch writes at a fixed place of the set's inflated buffer, `room` bytes in: the incomplete record the batch in front
    // ends with (known only when that batch's boundaries are) is copied in front of it afterwards.  A longer tail -- a single record of more
    // than a megabyte -- moves the batch to a buffer of its own (PostSet::big).  SQUID_CARRY_ROOM=<bytes, a multiple of 16> (tests: 64 takes nearly every batch that way).
    const unsigned long long room = std::getenv("SQUID_CARRY_ROOM") ? ((unsigned long long)std::max(0ll, std::atoll(std::getenv("SQUID_CARRY_ROOM"))) + 15) / 16 * 16 : (unsigned long long)1 << 20;
    // the stream and the events of a buffer set are made when its first batch is staged (on the planner thread): a stream of a priority
    // level that has none yet costs the runtime a hardware queue, 7-8 ms each in a process that has just started -- eight of them in
    // front of the first batch were 60 ms of a cold start
    auto ensure_set = [&D](int qi) -> int {
        hipStream_t& q = D.il_stream[qi];
        if (!q) {
            // lowest priority: a token wave holds its CU for tens of milliseconds, and the resolve / boundary / parse kernels
            // of the batch in front (library stream) should get the CUs that come free first
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            static const bool spread = std::getenv("SQUID_IL_SPREAD") != nullptr && std::atoi(std::getenv("SQUID_IL_SPREAD")) != 0;
            static const int tok_prio = std::getenv("SQUID_TOK_PRIO") ? std::atoi(std::getenv("SQUID_TOK_PRIO")) : 0;  // 0: lowest (default), 1: the library stream's, 2: highest
            const int prio = tok_prio == 1 ? 0 : tok_prio == 2 ? hi : (spread && qi >= 4 ? (lo + hi) / 2 : lo);  // (experiment: the runtime keeps a pool of hardware queues per priority level)
            if (hipStreamCreateWithPriority(&q, hipStreamNonBlocking, prio) != hipSuccess) { (void)hipGetLastError(); if (hipStreamCreate(&q) != hipSuccess) return (int)SQ_E_HIP; }
        }
        DeviceRecords::InflSet& st = D.il_set[qi];
        if (!st.ready && hipEventCreateWithFlags(&st.ready, hipEventDisableTiming) != hipSuccess) return (int)SQ_E_HIP;
        if (!st.freed && hipEventCreateWithFlags(&st.freed, hipEventDisableTiming) != hipSuccess) return (int)SQ_E_HIP;
        if (!st.copied && hipEventCreateWithFlags(&st.copied, hipEventDisableTiming) != hipSuccess) return (int)SQ_E_HIP;
        if (!D.il_post[qi].carried && hipEventCreateWithFlags(&D.il_post[qi].carried, hipEventDisableTiming) != hipSuccess) return (int)SQ_E_HIP;
        return SQ_OK;
    };
    if (report) std::fprintf(stderr, "GPU ingest: entry + %.1f ms: kernel attributes\n", since_ms(w_entry));
    // The compressed bytes: in HBM already (sq_stage_bam), or streamed there now by the feeder, which then also walks the block
    // headers (its more() replaces the caller's index walk over a mapping); the per-batch blocking copy of round 3 only remains for
    // files that would not fit beside the batch buffers.
    std::unique_ptr<FileFeeder> feed;
    IndexMore index_more = index_more_in;
    if (!c->ingest_dfile && src && src->path) {
        size_t free_b = 0, total_b = 0;
        const bool walk = (bool)index_more_in;
        const size_t from = blocks[b0].coff;
        const size_t upto = walk ? (src->stop == (size_t)-1 ? (size_t)-1 : src->stop + 2 * 65536 + 64) : (size_t)(blocks[std::min(b1, blocks.size()) - 1].coff + blocks[std::min(b1, blocks.size()) - 1].clen + 8);
        const size_t need = upto == (size_t)-1 ? file_bytes : upto - from;  // bytes of the file that will lie in HBM
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need + ((size_t)30 << 30) < free_b) {
            feed.reset(new FileFeeder(c));
            const int rc = feed->start(src->path, from, upto, walk, src->walk_p, src->walk_total, src->stop);
            if (rc) return rc;
            if (walk) index_more = [&feed](std::vector<BgzfRange>& v) { return feed->more(v); };
            if (report) std::fprintf(stderr, "GPU ingest: entry + %.1f ms: file streamed by %d threads, %zu pieces, set-up %.1f ms\n", since_ms(w_entry), feed->T, feed->npieces, feed->t_setup_ms);
        } else (void)hipGetLastError();
    }
    struct FeedGuard { std::unique_ptr<FileFeeder>& f; ~FeedGuard() { if (f) f->cancel(); } } feed_guard{feed};  // (every way out stops the threads)
    struct Batch { size_t at, end; unsigned long long coff0, cbytes, bbase, bbytes; };
    std::vector<Batch> batches;
    std::mutex bm;  // `batches` and `blocks` grow on the planner thread (below); the batch loop reads them through batch_of()
    auto batch_of = [&](size_t k) { std::lock_guard<std::mutex> lk(bm); return batches[k]; };
    // batches are planned as they are needed: with index_more the block index itself grows batch by batch (b1 = npos)
    bool more_blocks = (bool)index_more;
    size_t plan_at = b0;
    auto plan = [&](size_t k) -> bool {  // makes batches[k] exist; false when the range is used up
        while (batches.size() <= k) {
            const size_t at = plan_at;
            // the first batches are small, so that the GPU has work after a few milliseconds of index walk and copy
            // (no ramp when the file is in HBM already: small batches only leave CUs empty for the length of a token wave)
            static const unsigned long long ramp_env = std::getenv("SQUID_TOK_RAMP_MB") ? (unsigned long long)std::atoll(std::getenv("SQUID_TOK_RAMP_MB")) : 0;
            const unsigned long long ramp0 = ramp_env ? ramp_env : (c->ingest_dfile ? 1024 : 128);
            const unsigned long long ramp = ramp0 << (20 + std::min<size_t>(batches.size(), 8));
            unsigned long long bcap = std::min(cap, ramp);
            // (the end of the range in smaller batches: what stands behind the last byte is one batch's way through token pass, resolve, boundaries and parse)
            // (While the block index is still being walked the rest of the range is an estimate from the file's size; SQUID_TOK_TAPER=0: only with a
            // full index, as until round 6.  Measured and dropped: batches that keep shrinking to 16 MB -- twelve more batches per C3 step, each with its
            // trip through the boundary chain: staged 94 -> 102 ms.)
            static const bool taper = std::getenv("SQUID_TOK_TAPER") == nullptr || std::atoi(std::getenv("SQUID_TOK_TAPER")) != 0;
            if (tok_spec_early && at < blocks.size()) {
                const size_t stop0 = more_blocks || b1 == (size_t)-1 ? blocks.size() : std::min(b1, blocks.size());
                unsigned long long left = ~0ull;
                if (!more_blocks) { if (stop0 > at) left = blocks[stop0 - 1].uoff + blocks[stop0 - 1].isize - blocks[at].uoff; }
                else if (taper && at > b0 && file_bytes) {
                    const BgzfRange &f = blocks[b0], &a = blocks[at];
                    const double ratio = (double)(a.uoff - f.uoff) / (double)std::max<unsigned long long>(1, a.coff - f.coff);
                    const unsigned long long end_c = (src && src->stop != (size_t)-1) ? std::min<unsigned long long>(src->stop, file_bytes) : (unsigned long long)file_bytes;
                    left = end_c > a.coff ? (unsigned long long)(ratio * (double)(end_c - a.coff)) : 0;
                }
                if (left < 2 * cap) bcap = std::min(bcap, std::max<unsigned long long>((unsigned long long)64 << 20, cap / 4));
            }
            while (more_blocks && (blocks.size() <= at || blocks.back().uoff + blocks.back().isize - blocks[at].uoff <= bcap)) {
                std::vector<BgzfRange> got;  // (the wait for the walk happens outside the lock)
                more_blocks = index_more(got);
                std::lock_guard<std::mutex> lk(bm);
                blocks.insert(blocks.end(), got.begin(), got.end());
            }
            const size_t stop = more_blocks || b1 == (size_t)-1 ? blocks.size() : std::min(b1, blocks.size());
            if (at >= stop) return false;
            size_t end = at;
            while (end < stop && (end == at || blocks[end].uoff + blocks[end].isize - blocks[at].uoff <= bcap)) ++end;
            { std::lock_guard<std::mutex> lk(bm); batches.push_back(Batch{at, end, blocks[at].coff, blocks[end - 1].coff + blocks[end - 1].clen - blocks[at].coff, blocks[at].uoff, blocks[end - 1].uoff + blocks[end - 1].isize - blocks[at].uoff}); }
            plan_at = end;
        }
        return true;
    };
    const unsigned long long first_uoff = blocks[b0].uoff;
    // inflated size of the whole range, for sizing the record arrays once: exact with a full index, else from the file size
    auto range_bytes_estimate = [&]() -> unsigned long long {
        std::lock_guard<std::mutex> lk(bm);
        if (!index_more) { const size_t e = std::min(b1, blocks.size()); return blocks[e - 1].uoff + blocks[e - 1].isize - first_uoff; }
        const BgzfRange &f = blocks.front(), &l = blocks.back();  // (a .bai shard's walk starts in the middle of the file: ratio over the walked part only)
        const double ratio = (double)(l.uoff + l.isize - f.uoff) / (double)std::max<unsigned long long>(1, l.coff + l.clen - f.coff);
        return (unsigned long long)(ratio * 1.03 * (double)(file_bytes - std::min<unsigned long long>(file_bytes, f.coff)));
    };
    // A short range -- a chromosome shard of an eight-rank run holds an eighth of the file -- is cut into as many batches as a whole file, so that
    // the stages behind the token pass overlap as they do there: with 512 MB batches such a shard was three batches that went through the pipeline
    // nearly one after the other (tools/shard_project.py: 36-55 ms per rank where 21 would be its share).
    if (tok_spec_early && !std::getenv("SQUID_TOK_CAP_MB")) {
        unsigned long long est = range_bytes_estimate();
        if (index_more && src && src->stop != (size_t)-1) {
            std::lock_guard<std::mutex> lk(bm);
            const BgzfRange &f = blocks.front(), &l = blocks.back();
            const double ratio = (double)(l.uoff + l.isize - f.uoff) / (double)std::max<unsigned long long>(1, l.coff + l.clen - f.coff);
            if (src->stop > f.coff) est = (unsigned long long)(ratio * 1.03 * (double)(std::min<unsigned long long>(src->stop, file_bytes) - f.coff));
        }
        cap = std::max<unsigned long long>((unsigned long long)64 << 20, std::min(cap, est / 24));
    }
    Shard sh_here = c->shard;
    if (c->capture_names) sh_here.on = false;  // (the chimeric BAM is every rank's, whole)
    const Shard& sh = sh_here;
    // stage A of batch k: compressed bytes and block table to the device, token pass
    auto stage_a = [&](size_t k) -> int {  // (SQ_OK also when there is no batch k)
        if (!plan(k)) return SQ_OK;
        const Batch B = batches[k];
        { const int rc = ensure_set((int)(k % (size_t)D.il_depth)); if (rc) return fail(c, rc, "cannot create the stream of a buffer set"); }
        DeviceRecords::InflSet& st = D.il_set[k % (size_t)D.il_depth];
        hipStream_t sa = D.il_stream[k % (size_t)D.il_depth];
        const int nb = (int)(B.end - B.at);
        // largest compressed blocks first: the lanes of a wave get blocks of similar length (a wave takes as long as its
        // longest lane) and the long waves start first
        st.host_tab.resize((size_t)nb);
        for (int i = 0; i < nb; ++i) { const BgzfRange& b = blocks[B.at + (size_t)i]; st.host_tab[(size_t)i] = InflBlock{b.coff - B.coff0, b.clen, b.isize, b.uoff, 0}; }
        std::stable_sort(st.host_tab.begin(), st.host_tab.end(), [](const InflBlock& x, const InflBlock& y) { return x.clen > y.clen; });
        unsigned long long tok_slots = 0;  // token slots of the batch: half a slot per inflated byte (t2_tok_cap), block after block in table order
        for (InflBlock& ib : st.host_tab) { ib.toff = tok_slots; tok_slots += tok_spec ? isp::tok_cap_spec(ib.isize, ib.clen) : t2_tok_cap(ib.isize); }
        // (the set's stream runs token pass and resolve of batch k - depth in front of this batch's: its tokens have been read by then)
        const auto wa0 = std::chrono::steady_clock::now();
        // sized for a full batch at once (the first batches are small): growing a buffer later frees the old one, and freeing
        // device memory waits for the kernels of the other batches
        const unsigned long long full = std::max<unsigned long long>(B.bbytes, std::min<unsigned long long>(cap, range_bytes_estimate()));
        const double cratio = (double)B.cbytes / (double)std::max<unsigned long long>(B.bbytes, 1);
        const uint8_t* dfile = c->ingest_dfile ? c->ingest_dfile : (feed ? feed->dfile() : nullptr);  // the file is (or is arriving) in HBM: the kernels read it in place
        if (!dfile) HIPCHK(st.in.reserve(std::max((size_t)B.cbytes, (size_t)(cratio * 1.1 * (double)full)) + 256));  // (the input rings read up to 80 bytes ahead)
        st.src = dfile ? dfile + B.coff0 : st.in.p;
        HIPCHK(st.tab.reserve(std::max((size_t)nb, (size_t)(full / 60000)))); HIPCHK(st.flags.reserve(4));
        // (a full batch's token slots: half a slot per inflated byte + per block the slack of t2_tok_cap / tok_cap_spec, which also takes an eighth of a slot per compressed byte)
        const size_t tok_full = tok_spec ? (size_t)(full / 2 + (unsigned long long)(cratio * 1.1 * (double)full) / 8 + 260 * (full / 60000 + 1)) : (size_t)(full / 2 + 80 * (full / 60000 + 1));
        HIPCHK(st.tok.reserve(std::max((size_t)tok_slots, tok_full) + 64)); HIPCHK(st.ntok.reserve(std::max((size_t)nb, (size_t)(full / 60000))));
        if (!tok_spec) HIPCHK(st.lens.reserve((std::max((size_t)nb, (size_t)(full / 60000)) + 128) * T2_LENS_WORDS));  // (one strip of code lengths per lane of the token pass)
        const double wa1 = since_ms(wa0);
        if (!dfile) {
            // (the copy no longer travels on the set's stream: the token pass and the resolve that last read this buffer are waited for here)
            if (k >= (size_t)D.il_depth) HIPCHK(hipEventSynchronize(st.ready));  // (of batch k - depth: recorded behind its resolve)
            const int rc = h2d_parallel(c, st.in.p, file + B.coff0, (size_t)B.cbytes);
            if (rc) return rc;
        }
        const double wa2 = since_ms(wa0);
        if (feed) { const int rc = feed->wait_bytes((size_t)B.coff0, (size_t)(B.coff0 + B.cbytes) + 256, sa); if (rc) return rc; }  // the token pass waits for its own pieces only
        const double wa3 = since_ms(wa0);
        struct Rep { bool on; size_t k; double at, a1, a2, a3, gb; std::chrono::steady_clock::time_point t0; ~Rep() { if (on) std::fprintf(stderr, "GPU ingest: batch %zu planned at %.1f ms, buffers %.1f ms, copy of %.2f GB returned after %.1f ms, its pieces queued after %.1f ms, table + token pass queued after %.1f ms\n", k, at, a1, gb, a2 - a1, a3 - a2, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() - a3); } }
            rep{report && k < 3, k, std::chrono::duration<double, std::milli>(wa0 - w0).count(), wa1, wa2, wa3, (double)B.cbytes * 1e-9, wa0};
        HIPCHK(hipMemcpyAsync(st.tab.p, st.host_tab.data(), (size_t)nb * sizeof(InflBlock), hipMemcpyHostToDevice, sa));
        HIPCHK(hipMemsetAsync(st.flags.p, 0, 4 * 4, sa));
        if (tok_spec) {
            EvTimer t1(c, "k_inflate_spec", (double)B.cbytes + (double)B.bbytes * 2, sa);
            launch_inflate_spec<SPEC_CH, SPEC_PB>(sa, st.src, st.tab.p, nb, st.flags.p, st.tok.p, st.ntok.p);
        } else {   // k_inflate_tok2 stays on the set's own stream: the token kernels of up to four batches run side by side
            EvTimer t1(c, "k_inflate_tok2", (double)B.cbytes + (double)B.bbytes * 2, sa);
            if (tok_prof && k == 8) hipLaunchKernelGGL(k_inflate_tok2<true>, dim3((nb + 63) / 64), dim3(64), T2_LDS_BYTES, sa, st.src, st.tab.p, nb, st.flags.p, st.tok.p, st.ntok.p, st.lens.p, D.tok_prof.p);
            else hipLaunchKernelGGL(k_inflate_tok2<false>, dim3(((nb + 63) / 64 + tok_wpb - 1) / tok_wpb), dim3(64 * tok_wpb), tok_wpb * T2_LDS_BYTES, sa, st.src, st.tab.p, nb, st.flags.p, st.tok.p, st.ntok.p, st.lens.p, nullptr);
        }
        // the resolve, right behind: into the set's inflated buffer, which batch k - depth has left -- its parse is over (the planner waits for
        // that iteration of the batch loop) and its tail has been copied in front of the batch behind it (`carried`)
        DeviceRecords::PostSet& P = D.il_post[k % (size_t)D.il_depth];
        HIPCHK(P.out.reserve((size_t)room + (size_t)full + ((size_t)1 << 20) + 64));
        if (k >= (size_t)D.il_depth) HIPCHK(hipStreamWaitEvent(sa, P.carried, 0));
        {
            EvTimer t2(c, resolve_global ? (resolve_staged() ? "k_lz_resolve5" : "k_lz_resolve3") : "k_lz_resolve2", (double)B.bbytes * 3, sa);
            if (resolve_global && resolve_staged()) hipLaunchKernelGGL(k_lz_resolve5, dim3(nb), dim3(64), 0, sa, st.tok.p, st.ntok.p, st.tab.p, nb, B.bbase, P.out.p + room, st.flags.p);
            else if (resolve_global) hipLaunchKernelGGL(k_lz_resolve3, dim3(nb), dim3(64), resolve_lds_pad(), sa, st.tok.p, st.ntok.p, st.tab.p, nb, B.bbase, P.out.p + room, st.flags.p);
            else hipLaunchKernelGGL(k_lz_resolve2, dim3(nb), dim3(128), 65536 + 16, sa, st.tok.p, st.ntok.p, st.tab.p, nb, B.bbase, P.out.p + room, st.flags.p);
        }
        HIPCHK(hipEventRecord(st.ready, sa));
        return SQ_OK;
    };
    // Stage A runs on a planner thread, batch after batch, as far ahead of the batch loop as there are buffer sets: set k % depth is free
    // for batch k once the batch loop is through with batch k - depth (its parse is over, the front of the batch behind it queued).  The thread blocks
    // where stage A blocks -- in the header walk, in the wait for a batch's file pieces to be queued for copy, in the copy of pageable
    // memory -- and the batch loop does not: round 4's first form queued depth - 1 batches before the first resolve, which with eight sets
    // meant waiting for 40 % of a streamed file before anything was resolved.
    std::mutex pm;
    std::condition_variable pcv;
    size_t staged = 0, iters_done = 0;  // (pm) batches through stage A; batches the loop below is through with
    bool planner_over = false, planner_stop = false;
    int planner_rc = SQ_OK;
    std::string planner_err;  // (its error text: moved into c->err by this thread once the planner is over)
    std::thread planner([&]() {
        ErrSink sink(&planner_err);
        int rc = hipSetDevice(c->P.device) == hipSuccess ? SQ_OK : (int)SQ_E_HIP;
        try {
            for (size_t k = 0; rc == SQ_OK; ++k) {
                { std::unique_lock<std::mutex> lk(pm); pcv.wait(lk, [&]() { return planner_stop || k < iters_done + (size_t)D.il_depth; }); if (planner_stop) break; }
                if (!plan(k)) break;
                rc = stage_a(k);
                if (rc == SQ_OK) { std::lock_guard<std::mutex> lk(pm); staged = k + 1; }
                pcv.notify_all();
            }
        } catch (const std::exception& e) {  // (the block index and the batch tables grow on this thread: out of memory must not end the process)
            rc = fail(c, SQ_E_CAPACITY, std::string("GPU reader, planner thread: ") + e.what());
        }
        { std::lock_guard<std::mutex> lk(pm); planner_rc = rc; planner_over = true; }
        pcv.notify_all();
    });
    // (declared behind feed_guard: runs first on every way out -- the feeder is cancelled here too, so that a planner blocked in it returns)
    struct PlannerGuard { std::thread& t; std::mutex& m; std::condition_variable& cv; bool& stop; std::unique_ptr<FileFeeder>& f; bool& over;
        ~PlannerGuard() {
            bool done; { std::lock_guard<std::mutex> lk(m); stop = true; done = over; } cv.notify_all();
            if (!done && f) f->cancel();
            if (t.joinable()) t.join();
            if (!done) (void)hipDeviceSynchronize();  // (given up half way: what the thread queued last reads buffers the caller is about to reuse)
        } } planner_guard{planner, pm, pcv, planner_stop, feed, planner_over};
    auto wait_staged = [&](size_t k) -> int {  // 1: batch k is through stage A; 0: there is no batch k; < 0 / 2: stage A failed
        std::unique_lock<std::mutex> lk(pm);
        pcv.wait(lk, [&]() { return staged > k || planner_over; });
        if (staged > k) return 1;
        if (planner_rc && !planner_err.empty()) c->err = planner_err;  // (the planner is over: nobody else writes it)
        return planner_rc > 0 ? (int)SQ_E_HIP : planner_rc;  // (error codes are negative)
    };
    auto batch_done = [&]() { { std::lock_guard<std::mutex> lk(pm); ++iters_done; } pcv.notify_all(); };
    { const int r = wait_staged(0); if (r <= 0) return r; }
    const double w_first = since_ms(w0);
    // Behind the resolve: the front of batch k + 1 -- the tail of batch k copied in front of its bytes, slice boundaries -- is queued on the
    // library stream before batch k is parsed on the parse stream, so the two overlap.  The front needs the bytes of the incomplete record
    // at the end of the batch before (`carry`, known once that batch's boundaries are): this is the one chain that runs from batch to
    // batch, and since round 6 the resolve is no longer part of it (rounds 1-5 resolved batch k + 1 behind its carry: token passes
    // waited for resolves, resolves for the boundary search of the batch in front and a trip to the host -- a period of 4.0 ms per
    // 512 MB batch of which the token pass, the longest kernel, took 2.8).
    if (!D.il_parse_stream) {
        // (highest priority: the parse of batch k shares the machine with the resolves and token passes of the batches behind -- thousands of
        // one-wave workgroups that take every free wave slot -- and the batch loop waits for the parse)
        static const bool parse_hi = std::getenv("SQUID_PARSE_PRIO") == nullptr || std::atoi(std::getenv("SQUID_PARSE_PRIO")) != 0;
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (!parse_hi || hipStreamCreateWithPriority(&D.il_parse_stream, hipStreamNonBlocking, hi) != hipSuccess) { (void)hipGetLastError(); HIPCHK(hipStreamCreateWithFlags(&D.il_parse_stream, hipStreamNonBlocking)); }
    }
    if (!D.il_host) HIPCHK(hipHostMalloc((void**)&D.il_host, 64 * sizeof(int32_t)));
    hipStream_t sp = D.il_parse_stream;
    struct Front { const uint8_t* base = nullptr; /* the set's inflated buffer, or PostSet::big */ unsigned long long at = 0 /* where the carried bytes start */, carry = 0, limit = 0; long long nsl = 0; RecScan S{}; };
    Front fr[2];
    auto issue_front = [&](size_t k, unsigned long long carry_in, const uint8_t* carry_src) -> int {
        const Batch B = batch_of(k);
        DeviceRecords::InflSet& st = D.il_set[k % (size_t)D.il_depth];
        DeviceRecords::PostSet& P = D.il_post[k % (size_t)D.il_depth];
        Front& F = fr[k & 1];
        int32_t* hk = D.il_host + 32 * (k & 1);  // [0..9] flags of the front, [10..13] of the token pass and the resolve, [16..17] where the walk stopped
        F.carry = carry_in;
        HIPCHK(P.flags.reserve(16));
        HIPCHK(hipMemsetAsync(P.flags.p, 0, 16 * 4, s));
        HIPCHK(hipStreamWaitEvent(s, st.ready, 0));  // token pass and resolve of the batch
        if (carry_in <= room) {
            F.base = P.out.p; F.at = room - carry_in; F.limit = room + B.bbytes;
            if (carry_in) HIPCHK(hipMemcpyAsync(P.out.p + F.at, carry_src, (size_t)carry_in, hipMemcpyDeviceToDevice, s));
        } else {  // (the batch's own bytes start 16-byte aligned there as well)
            const unsigned long long pad = (16 - carry_in % 16) % 16;
            HIPCHK(P.big.reserve((size_t)(pad + carry_in + B.bbytes) + ((size_t)1 << 20) + 64));
            F.base = P.big.p; F.at = pad; F.limit = pad + carry_in + B.bbytes;
            HIPCHK(hipMemcpyAsync(P.big.p + pad, carry_src, (size_t)carry_in, hipMemcpyDeviceToDevice, s));
            HIPCHK(hipMemcpyAsync(P.big.p + pad + carry_in, P.out.p + room, (size_t)B.bbytes, hipMemcpyDeviceToDevice, s));
        }
        if (k > 0) HIPCHK(hipEventRecord(D.il_post[(k - 1) % (size_t)D.il_depth].carried, s));  // the buffer of batch k - 1 is read by its parse only from here on
        HIPCHK(hipMemcpyAsync(hk + 10, st.flags.p, 4 * 4, hipMemcpyDeviceToHost, s));  // (before the set goes back to the planner, whose next batch clears them)
        F.S = RecScan{F.base, F.at + (k == 0 ? (unsigned long long)begin : 0ull), F.limit, nref, sh.on ? sh.first_ref : -1, sh.on ? sh.end_ref : 0, (sh.on && c->P.rank == c->P.world_size - 1) ? 1 : 0};
        F.nsl = F.S.limit > F.S.begin ? (long long)((F.S.limit - F.S.begin + REC_SLICE - 1) / REC_SLICE) : 0;
        hk[16] = 0; hk[17] = 0;
        if (F.nsl > 0) {
            const long long nsl = F.nsl;
            HIPCHK(P.rec_sync.reserve((size_t)nsl)); HIPCHK(P.rec_end.reserve((size_t)nsl + 1)); HIPCHK(P.rec_cnt.reserve((size_t)nsl)); HIPCHK(P.rec_base.reserve((size_t)nsl));
            int32_t* tot = P.flags.p + 8;
            long long* tail_d = P.rec_end.p + nsl;  // where the walk stopped: the start of the incomplete tail
            HIPCHK(hipMemsetAsync(tail_d, 0, 8, s));
            { EvTimer t(c, "k_rec_sync+walk+check", 2.0 * (double)(F.S.limit - F.S.begin));
              // (single-wave workgroups: these kernels run beside the resolves and the token waves, whose one-wave
              // workgroups take wave slots one at a time as they come free -- a workgroup that needs four on one CU at once waited 3 ms)
              hipLaunchKernelGGL(k_rec_sync, dim3((unsigned)nsl), dim3(64), 0, s, F.S, nsl, (k > 0 || synced) ? 1 : 0, P.rec_sync.p);
              hipLaunchKernelGGL(k_rec_walk<false>, grid_for(nsl, 64), dim3(64), 0, s, F.S, nsl, P.rec_sync.p, P.rec_cnt.p, P.rec_end.p, nullptr, nullptr);
              hipLaunchKernelGGL(k_rec_check, grid_for(nsl, 64), dim3(64), 0, s, nsl, P.rec_sync.p, P.rec_end.p, P.flags.p, tail_d);
              HIPCHK((device_scan<OpSum, true>(s, nsl, FArr{P.rec_cnt.p}, P.rec_base.p, P.spine, tot))); }
            HIPCHK(hipMemcpyAsync(hk + 16, tail_d, 8, hipMemcpyDeviceToHost, s));
        }
        HIPCHK(hipMemcpyAsync(hk, P.flags.p, 10 * 4, hipMemcpyDeviceToHost, s));
        return SQ_OK;
    };
    long check_bad = 0;
    { int rc = issue_front(0, 0, nullptr); if (rc) { (void)give_up(); return rc; } }
    for (size_t k = 0;; ++k) {
        const Batch B = batch_of(k);
        DeviceRecords::PostSet& P = D.il_post[k % (size_t)D.il_depth];
        const Front F = fr[k & 1];
        const int32_t* hk = D.il_host + 32 * (k & 1);
        HIPCHK(hipStreamSynchronize(s));  // the front of batch k
        int32_t h[10], ha[4];
        for (int q = 0; q < 10; ++q) h[q] = hk[q];
        for (int q = 0; q < 4; ++q) ha[q] = hk[10 + q];
        long long tail = 0;
        std::memcpy(&tail, hk + 16, 8);
        if (check) {  // debugging: every block against zlib
            const uint8_t* out = F.base + F.at + F.carry;
            std::vector<uint8_t> got((size_t)B.bbytes), want;
            HIPCHK(hipMemcpy(got.data(), out, (size_t)B.bbytes, hipMemcpyDeviceToHost));
            for (size_t i = B.at; i < B.end; ++i) {
                BgzfRange b;
                { std::lock_guard<std::mutex> lk(bm); b = blocks[i]; }
                want.resize(b.isize);
                z_stream zs; std::memset(&zs, 0, sizeof zs);
                inflateInit2(&zs, -15);
                zs.next_in = (Bytef*)(file + b.coff); zs.avail_in = b.clen; zs.next_out = want.data(); zs.avail_out = b.isize;
                inflate(&zs, Z_FINISH); inflateEnd(&zs);
                if (std::memcmp(want.data(), got.data() + (b.uoff - B.bbase), b.isize) != 0) {
                    if (check_bad < 5) { size_t q = 0; while (want[q] == got[(b.uoff - B.bbase) + q]) ++q; std::fprintf(stderr, "[inflate check] block %zu (isize %u clen %u) differs at byte %zu: want %02x got %02x\n", i - b0, b.isize, b.clen, q, want[q], got[(b.uoff - B.bbase) + q]); }
                    ++check_bad;
                }
            }
            std::fprintf(stderr, "[inflate check] blocks %zu..%zu of %zu: %ld differ so far, flags %d|%d, records %d, carry in %llu\n", B.at - b0, B.end - b0, b1 - b0, check_bad, h[0], ha[0], h[8], F.carry);
        }
        if ((h[0] | ha[0]) & (512 | 1024)) return give_up();
        // the bytes behind the last complete record go in front of the next batch, whose front is queued now: it runs while this
        // batch is parsed
        const unsigned long long tail_at = tail > 0 ? (unsigned long long)tail : F.S.begin;
        const unsigned long long carry = F.limit > tail_at ? F.limit - tail_at : 0;
        const int has_next = wait_staged(k + 1);
        if (has_next < 0) { (void)give_up(); return has_next; }
        if (has_next) { int rc = issue_front(k + 1, carry, F.base + tail_at); if (rc) { (void)give_up(); return rc; } }
        const int64_t n_rec = h[8];
        if (n_rec > 0) {
            HIPCHK(P.bam_off.reserve((size_t)n_rec));
            hipLaunchKernelGGL(k_rec_walk<true>, grid_for(F.nsl, 64), dim3(64), 0, sp, F.S, F.nsl, P.rec_sync.p, nullptr, nullptr, P.rec_base.p, P.bam_off.p);
            // (sub-batches: the per-launch temporaries and the 32-bit block counters stay small)
            const int64_t kBatch = (int64_t)1 << 24;
            for (int64_t r0 = 0; r0 < n_rec; r0 += kBatch) {
                c->ingest_total_bytes = (size_t)range_bytes_estimate(); c->ingest_seen_bytes = (size_t)(B.bbase + B.bbytes - first_uoff);  // sizes the arrays for the whole range at once
                int rc = parse_device(c, F.base, (size_t)F.limit, P.bam_off.p + r0, std::min(kBatch, n_rec - r0), sp, P.flags.p, &P.spine);
                c->ingest_total_bytes = 0; c->ingest_seen_bytes = 0;
                if (rc) { (void)give_up(); return rc; }
            }
        }
        if (!has_next) { if (report) std::fprintf(stderr, "GPU ingest: %llu bytes left incomplete at the end\n", carry); break; }
        batch_done();  // (the parse is over: parse_device returns behind its last kernel)
    }
    planner.join();  // (over: the loop above ended on its word)
    HIPCHK(hipStreamSynchronize(sp));
    for (auto& q : D.il_stream) if (q) HIPCHK(hipStreamSynchronize(q));
    if (feed) {
        feed->finish();
        if (feed->failed.load()) return fail(c, SQ_E_IO, feed->error());
        if (report) {
            const size_t np = feed->npieces;
            std::fprintf(stderr, "GPU ingest: copy threads together: pread %.1f ms, waiting for a free buffer %.1f ms; header walk %.1f ms on its own thread\n", feed->us_pread.load() * 1e-3, feed->us_bufwait.load() * 1e-3, feed->walk_ms);
            std::fprintf(stderr, "GPU ingest: file pieces queued for copy (ms after the feeder started): #0 %.1f, #15 %.1f, #40 %.1f, #%zu %.1f, last %.1f\n", feed->t_issued[0].load(), feed->t_issued[std::min<size_t>(15, np - 1)].load(),
                         feed->t_issued[std::min<size_t>(40, np - 1)].load(), np / 2, feed->t_issued[np / 2].load(), feed->t_issued[np - 1].load());
        }
        src->streamed = true;
        if (feed->walk) { std::lock_guard<std::mutex> lk(feed->mu); src->walk_p = feed->end_state.p; src->walk_total = feed->end_state.total; src->bad = feed->walk_bad; }
    }
    if (tok_prof && batches.size() > 8) {  // (batch 8 ran the instrumented kernel)
        const int nw = (int)((batches[8].end - batches[8].at + 63) / 64);
        std::vector<unsigned long long> hp(8 * (size_t)nw);
        HIPCHK(hipMemcpy(hp.data(), D.tok_prof.p, hp.size() * 8, hipMemcpyDeviceToHost));
        double sum[8] = {0};
        for (int w = 0; w < nw; ++w) for (int q = 0; q < 8; ++q) sum[q] += (double)hp[8 * (size_t)w + q];
        const double steps = sum[6] / nw;
        std::fprintf(stderr, "token pass profile (batch 8, %d waves, %.0f steps per wave; s_memtime ticks per step): emit+header %.1f, header->topup-check %.1f, topup %.1f, ll decode %.1f, match path %.1f, loop/literal %.1f\n",
                     nw, steps, sum[0] / sum[6], sum[1] / sum[6], sum[2] / sum[6], sum[3] / sum[6], sum[4] / sum[6], sum[7] / sum[6]);
    }
    if (report) std::fprintf(stderr, "GPU ingest: first two batches queued after %.1f ms, all %zu batches through after %.1f ms (%.1f ms since entry; %llu MB per batch)\n", w_first, batches.size(), since_ms(w0), since_ms(w_entry), cap >> 20);
    return SQ_OK;
}

// debugging / tests: copy the resident SoA back to the host
// the QNAMEs captured beside the resident records (sq_ctx::capture_names) into hb.names / hb.name_off
int dev_download_names(sq_ctx* c, HostBatch& hb) {
    DeviceRecords& D = *c->dev;
    const size_t n = (size_t)D.n;
    hb.names.resize(D.nm_bytes); hb.name_off.resize(n + 1);
    if (n) HIPCHK(hipMemcpy(hb.name_off.data(), D.nm_off.p, n * 4, hipMemcpyDeviceToHost));
    if (D.nm_bytes) HIPCHK(hipMemcpy(hb.names.data(), D.nm_blob.p, D.nm_bytes, hipMemcpyDeviceToHost));
    hb.name_off[n] = (uint32_t)D.nm_bytes;
    return SQ_OK;
}
int dev_download_records(sq_ctx* c, HostBatch& hb) {
    DeviceRecords& D = *c->dev;
    const size_t n = (size_t)D.n, nb = (size_t)D.nb;
    // (no clear(): a batch the caller keeps between calls has its pages already -- resize() to the size it had costs nothing)
    hb.names.clear(); hb.name_off.assign(1, 0);
    // every array on a thread of its own when there is something to copy: the room for an array (resize: zeroes and fresh pages, one
    // thread's work per array) is most of the time for a --bwa batch of tens of millions of records
    const int dev = c->P.device;
    std::vector<std::future<hipError_t>> jobs;
    const bool side_by_side = n > 1000000;
    auto down = [&](auto& vec, const auto* src, size_t cnt) {
        auto job = [&vec, src, cnt, dev]() -> hipError_t {
            vec.resize(cnt);
            if (!cnt) return hipSuccess;
            hipError_t e = hipSetDevice(dev);
            return e != hipSuccess ? e : hipMemcpy(vec.data(), src, cnt * sizeof(*src), hipMemcpyDeviceToHost);
        };
        if (side_by_side) jobs.push_back(std::async(std::launch::async, job));
        else { std::promise<hipError_t> p; p.set_value(job()); jobs.push_back(p.get_future()); }
    };
    down(hb.refid, D.refid.p, n); down(hb.pos, D.pos.p, n); down(hb.mrefid, D.mrefid.p, n); down(hb.mpos, D.mpos.p, n); down(hb.endpos, D.endpos.p, n);
    down(hb.flag, D.flag.p, n); down(hb.totlen, D.totlen.p, n); down(hb.mapq, D.mapq.p, n); down(hb.aux, D.aux.p, n);
    down(hb.blk_off, D.blk_off.p, n ? n + 1 : 0);
    down(hb.b_refpos, D.b_refpos.p, nb); down(hb.b_matchref, D.b_matchref.p, nb); down(hb.b_readpos, D.b_readpos.p, nb); down(hb.b_matchread, D.b_matchread.p, nb);
    hipError_t bad = hipSuccess;
    for (auto& j : jobs) { const hipError_t e = j.get(); if (e != hipSuccess) bad = e; }
    if (!n) hb.blk_off.assign(1, 0);
    HIPCHK(bad);
    return SQ_OK;
}

int dev_upload_nodes(sq_ctx* c, const std::vector<Node>& nodes) {
    DeviceRecords& D = *c->dev;
    hipStream_t s = c->stream;
    const int n = (int)nodes.size(), nref = (int)c->ref_len.size();
    std::vector<int32_t> chr(n), pos(n), len(n), cs(nref + 1, n), bo(nref + 1, 0), fo(nref + 1, 0);
    for (int i = n - 1; i >= 0; --i) { chr[i] = nodes[i].chr; pos[i] = nodes[i].pos; len[i] = nodes[i].len; }
    // chr_start[k] = first node with chr >= k
    int j = 0;
    for (int k = 0; k <= nref; ++k) { while (j < n && nodes[j].chr < k) ++j; cs[k] = j; }
    for (int k = 0; k < nref; ++k) {
        if (cs[k] == cs[k + 1]) return fail(c, SQ_E_ARG, "internal: a reference without nodes (the tiling covers every chromosome)");
        bo[k + 1] = bo[k] + (int32_t)(((int64_t)std::max(c->ref_len[k], 1) + (1 << NODE_BUCKET_SHIFT) - 1) >> NODE_BUCKET_SHIFT);
        const int64_t f = (int64_t)fo[k] + (((int64_t)std::max(c->ref_len[k], 1) + (1 << NODE_FINE_SHIFT) - 1) >> NODE_FINE_SHIFT) + 1;
        if (f > INT32_MAX) return fail(c, SQ_E_CAPACITY, "references too long for the node position index");
        fo[k + 1] = (int32_t)f;
    }
    const int total = fo[nref];
    // one packed upload: chr | pos | len | chr_start | bucket_off | fine_off
    const size_t words = 3 * (size_t)n + 3 * ((size_t)nref + 1);
    HIPCHK(D.n_chr.reserve(words)); HIPCHK(D.n_bucket.reserve(std::max(total, 1)));
    D.pin.reset();
    int32_t* h = D.pin.take_n<int32_t>(words);
    if (!h) return fail(c, SQ_E_HIP, "hipHostMalloc failed");
    std::memcpy(h, chr.data(), (size_t)n * 4); std::memcpy(h + n, pos.data(), (size_t)n * 4); std::memcpy(h + 2 * (size_t)n, len.data(), (size_t)n * 4);
    std::memcpy(h + 3 * (size_t)n, cs.data(), ((size_t)nref + 1) * 4); std::memcpy(h + 3 * (size_t)n + nref + 1, bo.data(), ((size_t)nref + 1) * 4);
    std::memcpy(h + 3 * (size_t)n + 2 * ((size_t)nref + 1), fo.data(), ((size_t)nref + 1) * 4);
    HIPCHK(hipMemcpyAsync(D.n_chr.p, h, words * 4, hipMemcpyHostToDevice, s));
    NodeView& nv = D.nv;
    nv.n = n; nv.n_ref = nref; nv.chr = D.n_chr.p; nv.pos = D.n_chr.p + n; nv.len = D.n_chr.p + 2 * (size_t)n; nv.chr_start = D.n_chr.p + 3 * (size_t)n;
    nv.fine = D.n_bucket.p; nv.bucket_off = D.n_chr.p + 3 * (size_t)n + nref + 1; nv.fine_off = D.n_chr.p + 3 * (size_t)n + 2 * ((size_t)nref + 1);
    HIPCHK(D.n_pack.reserve((size_t)std::max(n, 1)));
    nv.pack = D.n_pack.p;
    if (n) hipLaunchKernelGGL(k_node_pack, dim3((n + 255) / 256), dim3(256), 0, s, n, nv.chr, nv.pos, nv.len, D.n_pack.p);
    if (total) { EvTimer t(c, "k_node_buckets", 4.0 * total); hipLaunchKernelGGL(k_node_buckets, dim3((total + 255) / 256), dim3(256), 0, s, nv, total, D.n_bucket.p); }
    HIPCHK(hipStreamSynchronize(s));  // the host vectors go out of scope
    return SQ_OK;
}

// sharded runs, exchange 1: what the last passing records of this shard look like to ReadRec_t::Equal.  (Unsharded: nothing to do --
// the filters run inside k_pass1.)
// Timing-only switches that cut a kernel short: never silently.  The pass still runs (the timers are what such a run is for), a line goes
// to stderr, and sq_build_graph ends with SQ_E_ARG instead of handing out the graph made from the mutilated pass.
static int ablate_switch(sq_ctx* c, const char* name) {
    const char* v = std::getenv(name);
    const int level = v ? std::atoi(v) : 0;
    if (level) {
        if (!c->ablated) std::fprintf(stderr, "squid_hip: %s=%d cuts a kernel short (timing only): no graph, no SV calls from this run\n", name, level);
        c->ablated = true;
    }
    return level;
}

int dev_classify(sq_ctx* c, int32_t last_info[4]) {
    DeviceRecords& D = *c->dev;
    hipStream_t s = c->stream;
    const int64_t n = D.n;
    RecView R = D.view();
    HIPCHK(D.cls.reserve(n + 4)); HIPCHK(D.keep.reserve(n + 4));
    if (std::getenv("SQUID_CALIB")) {
        const int64_t words = (int64_t)1 << 28;  // 1 GiB: larger than the 256 MiB Infinity Cache
        HIPCHK(D.calib.reserve(words));
        HIPCHK(hipMemsetAsync(D.calib.p, 1, words * 4, s));
        EvTimer t(c, "k_calib_read4", 4.0 * words);
        hipLaunchKernelGGL(k_calib_read4, dim3(2048), dim3(256), 0, s, D.calib.p, words, D.flags.p + 16);
    }
    if (last_info) { last_info[0] = last_info[1] = last_info[2] = last_info[3] = 0; }
    if (n == 0 || !last_info) return SQ_OK;
    hipLaunchKernelGGL(k_last_info_raw, dim3(1), dim3(1), 0, s, R, c->P.min_mapqual, D.flags.p + 24);
    HIPCHK(hipMemcpyAsync(last_info, D.flags.p + 24, 16, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return SQ_OK;
}

// Pass 1 over the resident records (k_pass1): filters, duplicate drop, and everything the segmentation automaton needs from the
// stream -- zero-coverage records, cluster triggers, ConcordRest candidates -- in ONE read of the records.  The lists stay on the
// device (dev_segment_support fetches them); the scalars come back here.  `seed`: running other-pair of earlier shards.
int dev_pass1(sq_ctx* c, const std::vector<int32_t>& cl_chr, const std::vector<int32_t>& cl_start, const std::vector<int32_t>& cl_right, Pass1Result& out) {
    DeviceRecords& D = *c->dev;
    hipStream_t s = c->stream;
    const int64_t n = D.n;
    const int ncl = (int)cl_chr.size();
    out = Pass1Result();
    out.other_max = INT64_MIN;
    D.k1 = 0; D.p1_zc = 0; D.p1_rest = 0; D.p1_ntiles = 0;
    D.h_tile_rank.assign(1, 0);
    c->counts.n_kept_p1 = 0; c->counts.n_kept_p2 = 0;
    if (n >= ((int64_t)1 << 31) - P1_TILE) return fail(c, SQ_E_CAPACITY, "more than 2^31 records on one device");
    HIPCHK(D.cl_chr.reserve(3 * (size_t)std::max(ncl, 1))); HIPCHK(D.trig.reserve(std::max(ncl, 1)));
    if (ncl) {  // one packed upload: chr | start | right
        std::vector<int32_t> pack(3 * (size_t)ncl);
        std::copy(cl_chr.begin(), cl_chr.end(), pack.begin()); std::copy(cl_start.begin(), cl_start.end(), pack.begin() + ncl); std::copy(cl_right.begin(), cl_right.end(), pack.begin() + 2 * (size_t)ncl);
        HIPCHK(hipMemcpy(D.cl_chr.p, pack.data(), pack.size() * 4, hipMemcpyHostToDevice));
    }
    D.cl_n = ncl;
    // position index of the cluster table (16 KiB stretches per reference)
    const int n_ref = (int)c->ref_len.size();
    std::vector<int32_t> bo(n_ref + 1, 0);
    for (int k = 0; k < n_ref; ++k) {
        const int64_t nb = (int64_t)bo[k] + (((int64_t)std::max(c->ref_len[k], 1) + (1 << NODE_BUCKET_SHIFT) - 1) >> NODE_BUCKET_SHIFT);
        if (nb > INT32_MAX) return fail(c, SQ_E_CAPACITY, "references too long for the position index");
        bo[k + 1] = (int32_t)nb;
    }
    const int cl_total = bo[n_ref];
    HIPCHK(D.cl_bucket.reserve((size_t)n_ref + 1 + (size_t)std::max(cl_total, 1)));
    HIPCHK(hipMemcpy(D.cl_bucket.p, bo.data(), ((size_t)n_ref + 1) * 4, hipMemcpyHostToDevice));
    if (n == 0) return SQ_OK;
    for (int32_t len : c->ref_len) (void)len;
    if (c->ref_len.size() >= ((size_t)1 << 30)) return fail(c, SQ_E_CAPACITY, "too many references");
    const int ntiles = (int)((n + P1_TILE - 1) / P1_TILE);
    ClusterView C{ncl, D.cl_chr.p, D.cl_chr.p + ncl, D.cl_chr.p + 2 * (size_t)ncl, n_ref, D.cl_bucket.p, D.cl_bucket.p + n_ref + 1};
    if (ncl && cl_total) hipLaunchKernelGGL(k_cluster_buckets, dim3((cl_total + 255) / 256), dim3(256), 0, s, C, cl_total, D.cl_bucket.p + n_ref + 1);
    if (D.r_pack_n < n) {  // the packed rows of the records that arrived since the last pass (layout work of the ingest, done at first use:
        // every way records get here -- GPU reader, host batches, the record cache, the name-set fix-up -- is covered by this one place)
        HIPCHK(D.r_pack.grow_keep(2 * (size_t)D.r_pack_n, 2 * (size_t)n, s));
        EvTimer t(c, "k_pack_records", 60.0 * (double)(n - D.r_pack_n));
        hipLaunchKernelGGL(k_pack_records, grid_for(n - D.r_pack_n, 256), dim3(256), 0, s, D.r_pack_n, n, D.view(), D.r_pack.p);
        D.r_pack_n = n;
    }
    RecView R = D.view();
    HIPCHK(D.cls.reserve(n + 4)); HIPCHK(D.keep.reserve(n + 4));
    HIPCHK(D.tile_rank.reserve((size_t)ntiles + 1)); HIPCHK(D.tile_first.reserve(ntiles)); HIPCHK(D.tile_max.reserve(ntiles)); HIPCHK(D.tile_zbase.reserve(ntiles)); HIPCHK(D.tile_zcnt.reserve(ntiles));
    HIPCHK(D.tile_cnt.reserve(ntiles)); HIPCHK(D.tile_K.reserve(ntiles)); HIPCHK(D.tile_zcnt2.reserve(ntiles)); HIPCHK(D.tile_ob.reserve(2 * (size_t)ntiles));
    HIPCHK(D.p1_sc.reserve(P1S_WORDS));
    int32_t sc[P1S_WORDS], trig_last = INT32_MAX;
    if (D.zcap < (size_t)ntiles * P1_ZFIX + ((size_t)1 << 16)) D.zcap = (size_t)ntiles * P1_ZFIX + ((size_t)1 << 16);
    for (;;) {  // the two lists start small and grow when they overflow (the counts are exact either way)
        HIPCHK(D.z_idx.reserve(D.zcap)); HIPCHK(D.z_chr.reserve(D.zcap)); HIPCHK(D.z_right.reserve(D.zcap));
        HIPCHK(D.zc_v.reserve(D.zcap)); HIPCHK(D.zc_K.reserve(D.zcap)); HIPCHK(D.zc_refid.reserve(D.zcap)); HIPCHK(D.zc_pos.reserve(D.zcap)); HIPCHK(D.zc_ob.reserve(D.zcap));
        HIPCHK(D.rc_cluster.reserve(D.rc_cap)); HIPCHK(D.rc_pos.reserve(D.rc_cap)); HIPCHK(D.rc_len.reserve(D.rc_cap));
        HIPCHK(hipMemsetAsync(D.p1_sc.p, 0, P1S_WORDS * 4, s));
        if (ncl) HIPCHK(hipMemsetAsync(D.trig.p, 0x7f, (size_t)ncl * 4, s));
        P1Args A;
        A.min_mapq = c->P.min_mapqual; A.prior_mask = c->shard.on ? c->shard.dedup_mask : 0; A.RL = c->read_len;
        A.ablate = ablate_switch(c, "SQUID_P1_ABLATE");
        A.cls = D.cls.p; A.keep = D.keep.p; A.tile_cnt = D.tile_cnt.p; A.tile_K = D.tile_K.p; A.tile_ob = D.tile_ob.p; A.tile_first = D.tile_first.p; A.tile_max = D.tile_max.p;
        A.zc_v = D.zc_v.p; A.zc_K = D.zc_K.p; A.zc_refid = D.zc_refid.p; A.zc_pos = D.zc_pos.p; A.zc_ob = D.zc_ob.p; A.zcap = (int)D.zcap; A.zfix_end = ntiles * P1_ZFIX; A.tile_zbase = D.tile_zbase.p; A.tile_zcnt = D.tile_zcnt.p;
        A.trig = D.trig.p; A.rc_cluster = D.rc_cluster.p; A.rc_pos = D.rc_pos.p; A.rc_len = D.rc_len.p; A.rc_cap = (int)D.rc_cap;
        A.sc = D.p1_sc.p;
        {   // reads: 32 B of fixed fields per record (two 16-byte words) + its first and last block (16 B each); writes: class and keep byte
            EvTimer t(c, SQ_P1W_ITEMS > 0 ? "k_pass1w" : "k_pass1", 32.0 * n + 16.0 * D.nb);
#if SQ_P1W_ITEMS > 0
            static const int p1_waves = std::getenv("SQUID_P1_WAVES") ? std::atoi(std::getenv("SQUID_P1_WAVES")) : (P1_ITEMS >= 4 ? 3 : 4);  // (measured at C3: four records per lane at three waves per SIMD 0.85 ms, two at four waves 0.98)
            if (p1_waves == 4) hipLaunchKernelGGL((k_pass1w<4>), dim3(ntiles), dim3(P1_THREADS), 0, s, R, C, A);
            else if (p1_waves == 3) hipLaunchKernelGGL((k_pass1w<3>), dim3(ntiles), dim3(P1_THREADS), 0, s, R, C, A);
            else if (p1_waves == 5) hipLaunchKernelGGL((k_pass1w<5>), dim3(ntiles), dim3(P1_THREADS), 0, s, R, C, A);
            else if (p1_waves == 6) hipLaunchKernelGGL((k_pass1w<6>), dim3(ntiles), dim3(P1_THREADS), 0, s, R, C, A);
            else hipLaunchKernelGGL((k_pass1w<8>), dim3(ntiles), dim3(P1_THREADS), 0, s, R, C, A);
#else
            static const bool prof = std::getenv("SQUID_P1_PROF") != nullptr;  // s_memtime sums per section of a tile (thread 0 of every workgroup)
            if (prof) {
                HIPCHK(D.tok_prof.reserve(16)); HIPCHK(hipMemsetAsync(D.tok_prof.p, 0, 16 * 8, s));
                hipLaunchKernelGGL((k_pass1<true, 4>), dim3(ntiles), dim3(P1_THREADS), 0, s, R, C, A, D.tok_prof.p);
                unsigned long long hp[8];
                HIPCHK(hipMemcpyAsync(hp, D.tok_prof.p, sizeof hp, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s));
                static const char* nm[7] = {"start", "load+classify+dedup", "scan", "clusters+z+triggers+rest", "run of slots", "emit", "-"};
                for (int q = 0; q < 7; ++q) std::fprintf(stderr, "[k_pass1] %-28s %10.0f ticks per tile\n", nm[q], (double)hp[q] / (double)std::max<unsigned long long>(hp[7], 1));
            } else {
                static const int p1_waves = std::getenv("SQUID_P1_WAVES") ? std::atoi(std::getenv("SQUID_P1_WAVES")) : 5;  // (five waves per SIMD with 44 B of spills: 1.49 ms at C3; four without: 1.65; six: 2.49)
                if (p1_waves == 5) hipLaunchKernelGGL((k_pass1<false, 5>), dim3(ntiles), dim3(P1_THREADS), 0, s, R, C, A, (unsigned long long*)nullptr);
                else if (p1_waves == 6) hipLaunchKernelGGL((k_pass1<false, 6>), dim3(ntiles), dim3(P1_THREADS), 0, s, R, C, A, (unsigned long long*)nullptr);
                else if (p1_waves == 8) hipLaunchKernelGGL((k_pass1<false, 8>), dim3(ntiles), dim3(P1_THREADS), 0, s, R, C, A, (unsigned long long*)nullptr);
                else hipLaunchKernelGGL((k_pass1<false, 4>), dim3(ntiles), dim3(P1_THREADS), 0, s, R, C, A, (unsigned long long*)nullptr);
            }
#endif
        }
        {   // per tile: count 4 + pair 8 + two keys 16 in, rank 4 + pair 8 out
            EvTimer t(c, "k_tile_scan", 40.0 * ntiles);
            const int ngroups = (ntiles + TS_THREADS - 1) / TS_THREADS;
            HIPCHK(D.tile_part.reserve((size_t)ngroups + 1));
            hipLaunchKernelGGL(k_tile_partial, dim3(ngroups), dim3(TS_THREADS), 0, s, ntiles, D.tile_cnt.p, D.tile_ob.p, D.tile_max.p, D.tile_part.p);
            hipLaunchKernelGGL(k_tile_scan, dim3(ngroups), dim3(TS_THREADS), 0, s, ntiles, D.tile_cnt.p, D.tile_ob.p, D.tile_first.p, D.tile_max.p, D.tile_part.p, D.tile_rank.p, D.tile_ob.p + ntiles, D.p1_sc.p, SQ_P1W_ITEMS > 0 ? D.tile_K.p : (const int32_t*)nullptr, D.trig.p);
            if (ncl) hipLaunchKernelGGL(k_trig_rank, dim3((ncl + 255) / 256), dim3(256), 0, s, ncl, ntiles, D.tile_rank.p, D.trig.p);
        }
        HIPCHK(hipMemcpyAsync(sc, D.p1_sc.p, sizeof sc, hipMemcpyDeviceToHost, s));
        if (ncl) HIPCHK(hipMemcpyAsync(&trig_last, D.trig.p + (ncl - 1), 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (sc[P1S_FLAGS] & P1F_UNSORTED) return fail(c, SQ_E_UNSORTED, "concordant BAM is not coordinate sorted (README.md:23 requires it)");
        if (sc[P1S_FLAGS] & P1F_NEGATIVE_END) return fail(c, SQ_E_CAPACITY, "a concordant record ends at a negative reference position");
        bool again = false;
        const size_t zneed = (size_t)ntiles * P1_ZFIX + (size_t)sc[P1S_ZC];
        if (zneed > D.zcap) { D.zcap = zneed + zneed / 4 + 1024; again = true; }
        if ((size_t)sc[P1S_REST] > D.rc_cap) { D.rc_cap = (size_t)sc[P1S_REST] + (size_t)sc[P1S_REST] / 4 + 1024; again = true; }
        if (!again) break;
    }
    D.k1 = sc[P1S_KEPT]; D.p1_zc = ntiles * P1_ZFIX + sc[P1S_ZC]; D.p1_rest = sc[P1S_REST]; D.p1_ntiles = ntiles;
    out.trigger_last = trig_last;  // (first kept record behind the last cluster; >= kept: none)
    out.kept = sc[P1S_KEPT]; out.n_rest = sc[P1S_REST];
    out.first_kept[0] = sc[P1S_FIRST_REFID]; out.first_kept[1] = sc[P1S_FIRST_POS];
    out.other_max = (long long)(((unsigned long long)(uint32_t)sc[P1S_OTHER_HI] << 32) | (uint32_t)sc[P1S_OTHER_LO]);
    c->counts.n_kept_p1 = out.kept;
    c->counts.n_kept_p2 = out.n_rest;  // (re-used slot: ConcordRest candidates)
    return SQ_OK;
}

// The lists of pass 1: zero-coverage records (+ the running pair in front of each; k_zfinal settles the candidates of k_pass1 with
// the running pair of everything in front of their tile, `seed` = that of earlier shards), trigger record of every cluster,
// ConcordRest candidates, kept records in front of every tile.  One synchronisation.
int dev_segment_support(sq_ctx* c, int ncl, long long seed, SegSupport& out) {
    DeviceRecords& D = *c->dev;
    hipStream_t s = c->stream;
    const int64_t k = D.k1;
    out.zidx.clear(); out.z_ochr.clear(); out.z_oright.clear(); out.rest_cluster.clear(); out.rest_pos.clear(); out.rest_len.clear();
    out.trigger.assign(ncl, (int32_t)k);
    const int zc = D.p1_zc, cnt = D.p1_rest, ntiles = D.p1_ntiles;
    D.h_tile_rank.assign((size_t)ntiles + 1, 0);
    if (D.n == 0) return SQ_OK;
    const int n_ref = (int)c->ref_len.size();
    ClusterView C{ncl, D.cl_chr.p, D.cl_chr.p + ncl, D.cl_chr.p + 2 * (size_t)ncl, n_ref, D.cl_bucket.p, D.cl_bucket.p + n_ref + 1};
    {   // per candidate 24 B in, per zero-coverage record 12 B out; per tile its run descriptor
        EvTimer t(c, "k_zfinal", 24.0 * zc + 24.0 * ntiles);
        hipLaunchKernelGGL(k_zfinal, dim3((ntiles + 255) / 256), dim3(256), 0, s, ntiles, C, seed, c->read_len, D.tile_rank.p, D.tile_ob.p + ntiles, D.tile_zbase.p, D.tile_zcnt.p, (int)D.zcap, D.zc_v.p, D.zc_K.p, D.zc_ob.p,
                           D.zc_refid.p, D.zc_pos.p, D.tile_zcnt2.p, D.z_idx.p, D.z_chr.p, D.z_right.p);
    }
    // the survivors in stream order: offsets of the tiles' runs by a scan, one gather kernel, then three dense arrays come back
    int32_t nz_dev = 0;
    HIPCHK(D.scratch_b.reserve((size_t)ntiles + 1));
    int32_t* tot = D.flags.p + 8;
    if (zc) {
        HIPCHK((device_scan<OpSum, true>(s, ntiles, FArr{D.tile_zcnt2.p}, D.scratch_b.p, D.spine, tot)));
        HIPCHK(hipMemcpyAsync(&nz_dev, tot, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (nz_dev < 0 || nz_dev > zc) return fail(c, SQ_E_ARG, "internal: zero-coverage run outside the list");
        HIPCHK(D.scratch_c.reserve(3 * (size_t)std::max(nz_dev, 1)));
        if (nz_dev) hipLaunchKernelGGL(k_zcompact, grid_for(ntiles, 256), dim3(256), 0, s, ntiles, D.tile_zbase.p, D.tile_zcnt2.p, D.scratch_b.p, D.z_idx.p, D.z_chr.p, D.z_right.p, D.scratch_c.p, (int)nz_dev);
    }
    D.pin.reset();
    int32_t *hz = D.pin.take_n<int32_t>(3 * (size_t)nz_dev), *hr = D.pin.take_n<int32_t>(3 * (size_t)cnt), *ht = D.pin.take_n<int32_t>(ncl), *hk = D.pin.take_n<int32_t>((size_t)ntiles + 1);
    if (!hz || !hr || !ht || !hk) return fail(c, SQ_E_HIP, "hipHostMalloc failed");
    if (nz_dev) HIPCHK(hipMemcpyAsync(hz, D.scratch_c.p, 3 * (size_t)nz_dev * 4, hipMemcpyDeviceToHost, s));
    if (cnt) {
        HIPCHK(hipMemcpyAsync(hr, D.rc_cluster.p, (size_t)cnt * 4, hipMemcpyDeviceToHost, s)); HIPCHK(hipMemcpyAsync(hr + cnt, D.rc_pos.p, (size_t)cnt * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(hr + 2 * (size_t)cnt, D.rc_len.p, (size_t)cnt * 4, hipMemcpyDeviceToHost, s));
    }
    if (ncl) HIPCHK(hipMemcpyAsync(ht, D.trig.p, (size_t)ncl * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(hk, D.tile_rank.p, ((size_t)ntiles + 1) * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    out.zidx.assign(hz, hz + nz_dev); out.z_ochr.assign(hz + nz_dev, hz + 2 * (size_t)nz_dev); out.z_oright.assign(hz + 2 * (size_t)nz_dev, hz + 3 * (size_t)nz_dev);
    const int nz = (int)out.zidx.size();
    out.rest_cluster.assign(hr, hr + cnt); out.rest_pos.assign(hr + cnt, hr + 2 * (size_t)cnt); out.rest_len.assign(hr + 2 * (size_t)cnt, hr + 3 * (size_t)cnt);
    // trigger of cluster c = first kept record with more than c clusters behind it: the kernel left, per cluster, the first record
    // with exactly c + 1 behind it
    int32_t run = (int32_t)k;
    for (int q = ncl - 1; q >= 0; --q) { if (ht[q] < run) run = ht[q]; out.trigger[q] = run; }
    D.h_tile_rank.assign(hk, hk + ntiles + 1);
    if (std::getenv("SQUID_PREP_DEBUG")) std::fprintf(stderr, "[prepare] kept %lld zero-coverage records %d ConcordRest candidates %d clusters %d\n", (long long)k, nz, cnt, ncl);
    return SQ_OK;
}

// The window elements of the given kept-index ranges [lo,hi), built from the records of the tiles that hold them (k_summarise_tiles)
// and copied into one page-locked host buffer; range i starts at range_off[i] of `compact`.  `term` (sharded runs): the record with
// kept index D.k1, i.e. the first kept record of the next shard, which is not resident here.
int dev_fetch_stream(sq_ctx* c, const std::vector<std::pair<int64_t, int64_t>>& ranges, const StreamRec*& compact, std::vector<int64_t>& range_off, const StreamRec* term) {
    DeviceRecords& D = *c->dev;
    hipStream_t s = c->stream;
    auto t0 = std::chrono::steady_clock::now();
    const int nr = (int)ranges.size();
    range_off.assign(nr, 0);
    compact = nullptr;
    if (!nr) return SQ_OK;
    static_assert(sizeof(StreamRec) == 24, "StreamRec is six words");
    const std::vector<int32_t>& TR = D.h_tile_rank;
    const int ntiles = D.p1_ntiles;
    const int64_t K = D.k1;
    std::vector<SumItem> items;
    long long total = 0;
    std::vector<std::pair<int, int64_t>> patch;  // (range, offset) of the appended record
    for (int i = 0; i < nr; ++i) {
        range_off[i] = total;
        const int64_t lo = ranges[i].first, hi = ranges[i].second;
        const int64_t hi_own = std::min<int64_t>(hi, K);
        if (lo < hi_own) {
            // first tile with kept records behind index lo, then every tile that starts in front of hi
            int t = (int)(std::upper_bound(TR.begin() + 1, TR.begin() + 1 + ntiles, (int32_t)lo) - (TR.begin() + 1));
            for (; t < ntiles && TR[t] < hi_own; ++t) {
                if (TR[t + 1] == TR[t]) continue;
                items.push_back(SumItem{(int64_t)t * P1_TILE, TR[t], (int32_t)lo, (int32_t)hi_own, (int64_t)total});
            }
        }
        if (hi > K) {
            if (!term || hi != K + 1 || lo > K) return fail(c, SQ_E_ARG, "internal: stream range beyond the kept records");
            patch.push_back(std::make_pair(i, total + (K - lo)));
        }
        total += hi - lo;
    }
    StreamRec* dst = D.pin.take_n<StreamRec>((size_t)total);
    if (!dst) return fail(c, SQ_E_HIP, "hipHostMalloc failed");
    if (!items.empty()) {
        HIPCHK(D.srec.reserve((size_t)total)); HIPCHK(D.sum_items.reserve(items.size()));
        HIPCHK(hipMemcpyAsync(D.sum_items.p, items.data(), items.size() * sizeof(SumItem), hipMemcpyHostToDevice, s));
        {   // per summarised tile: keep 1 + class 1 + the fixed fields and the first block of its kept records in, 24 B out
            EvTimer t(c, "k_summarise_tiles", (double)items.size() * P1_TILE * 34.0 + 24.0 * (double)total);
            hipLaunchKernelGGL(k_summarise_tiles, dim3((unsigned)items.size()), dim3(P1_THREADS), 0, s, D.view(), D.cls.p, D.keep.p, D.sum_items.p, D.srec.p);
        }
        HIPCHK(hipMemcpyAsync(dst, D.srec.p, (size_t)total * sizeof(StreamRec), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));  // (items goes out of scope)
    }
    for (const auto& pt : patch) dst[pt.second] = *term;
    compact = dst;
    c->timer.add("d2h_stream_summary", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), 0);
    return SQ_OK;
}

// *D.r_break = index of the kept record with rank n_break: the records at or behind it are not consumed (SegmentGraph.cpp:338-339)
static int set_r_break(sq_ctx* c, int64_t n_break) {
    DeviceRecords& D = *c->dev;
    hipStream_t s = c->stream;
    HIPCHK(D.r_break.reserve(1));
    const long long all = (long long)D.n;
    HIPCHK(hipMemcpyAsync(D.r_break.p, &all, 8, hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));  // (`all` is a local)
    if (n_break >= D.k1 || D.p1_ntiles == 0) return SQ_OK;
    const std::vector<int32_t>& TR = D.h_tile_rank;
    if ((int)TR.size() != D.p1_ntiles + 1) return fail(c, SQ_E_ARG, "internal: dev_segment_support first");
    const int t = (int)(std::upper_bound(TR.begin() + 1, TR.begin() + 1 + D.p1_ntiles, (int32_t)n_break) - (TR.begin() + 1));  // first tile that ends behind rank n_break
    if (t >= D.p1_ntiles) return SQ_OK;
    hipLaunchKernelGGL(k_find_rank, dim3(1), dim3(P1_THREADS), 0, s, D.n, D.keep.p, (int64_t)t * P1_TILE, TR[t], (int32_t)n_break, D.r_break.p);
    return SQ_OK;
}

// ReadsOther (non-first blocks of the consumed kept records) in stream order, pulled to the host only when it
// contains a block of <= 3 bases (see k_gather_other); runs before the nodes exist so that the host can repeat the
// reference's std::sort in the background
int dev_gather_other(sq_ctx* c, int64_t n_break, bool& has_tiny, std::vector<int32_t>& other_chr, std::vector<int32_t>& other_pos, std::vector<int32_t>& other_len, bool always_fetch) {
    DeviceRecords& D = *c->dev;
    hipStream_t s = c->stream;
    const int64_t n = D.n;
    has_tiny = false;
    other_chr.clear(); other_pos.clear(); other_len.clear();
    if (n == 0) return SQ_OK;
    RecView R = D.view();
    { const int rb = set_r_break(c, n_break); if (rb) return rb; }
    int32_t* tot = D.flags.p + 8;
    HIPCHK(D.b0_b.reserve(n));
    HIPCHK(hipMemsetAsync(D.flags.p, 0, 8 * 4, s));
    { EvTimer t(c, "scan_other_offsets", 9.0 * n); HIPCHK((device_scan<OpSum, true>(s, n, FOtherCount{R, D.keep.p, D.r_break.p}, D.b0_b.p, D.spine, tot))); }
    int32_t cnt = 0;
    HIPCHK(hipMemcpyAsync(&cnt, tot, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (cnt == 0) return SQ_OK;
    HIPCHK(D.scratch_b.reserve(cnt)); HIPCHK(D.scratch_c.reserve(cnt)); HIPCHK(D.b0_a.reserve(cnt));
    { EvTimer t(c, "k_gather_other", 13.0 * n + 8.0 * D.nb + 12.0 * cnt);
      hipLaunchKernelGGL(k_gather_other, grid_for(n, 256), dim3(256), 0, s, R, D.keep.p, D.r_break.p, D.b0_b.p, D.scratch_b.p, D.scratch_c.p, D.b0_a.p, D.flags.p); }
    int32_t hf = 0;
    HIPCHK(hipMemcpyAsync(&hf, D.flags.p, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    has_tiny = hf & 64;
    if (!has_tiny && !always_fetch) return SQ_OK;
    auto t0 = std::chrono::steady_clock::now();
    other_chr.resize(cnt); other_pos.resize(cnt); other_len.resize(cnt);
    HIPCHK(hipMemcpyAsync(other_chr.data(), D.scratch_b.p, (size_t)cnt * 4, hipMemcpyDeviceToHost, s)); HIPCHK(hipMemcpyAsync(other_pos.data(), D.scratch_c.p, (size_t)cnt * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(other_len.data(), D.b0_a.p, (size_t)cnt * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    c->timer.add("d2h_reads_other", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), 0);
    return SQ_OK;
}

// K3: per-node read support and summed block length of the consumed stream prefix
int dev_node_depth(sq_ctx* c, const std::vector<Node>& nodes, int64_t n_break, std::vector<int32_t>& support, std::vector<int64_t>& sumlen, bool& need_exact_other,
                   std::vector<int32_t>& amb_plus, std::vector<int32_t>& amb_minus, std::vector<int32_t>& other_len) {
    DeviceRecords& D = *c->dev;
    hipStream_t s = c->stream;
    const int64_t n = D.n;
    const int nn = (int)nodes.size();
    if (D.nv.n != nn) return fail(c, SQ_E_ARG, "internal: dev_upload_nodes first");
    const NodeView nv = D.nv;
    // one accumulator block: [main cnt | main sum | other cnt | other sum | amb+ | amb-] (the copies of the first four sit
    // behind it), one memset; counters: flags[0..7] and the stripes right behind them, one memset
    const size_t nst = (size_t)nn * NODE_STRIPES;
    HIPCHK(D.acc_a.reserve(6 * (size_t)nn + 4 * nst));
    int32_t* acc = D.acc_a.p;
    int32_t *a_mc = acc + 6 * (size_t)nn, *a_ms = a_mc + nst, *a_oc = a_ms + nst, *a_os = a_oc + nst, *a_ap = acc + 4 * (size_t)nn, *a_am = acc + 5 * (size_t)nn;
    HIPCHK(hipMemsetAsync(acc, 0, (6 * (size_t)nn + 4 * nst) * 4, s));
    HIPCHK(D.flags.reserve(64 + NSTRIPE));
    int32_t* stripes = D.flags.p + 64;
    HIPCHK(hipMemsetAsync(D.flags.p, 0, (64 + NSTRIPE) * 4, s));
    RecView R = D.view();
    if (n > 0) {
        { const int rb = set_r_break(c, n_break); if (rb) return rb; }
        const int ntiles = (int)((n + ST_TILE - 1) / ST_TILE);
        // per tile: its largest early node, the cursor at its first record; the tiles to correct and the cursor in front of them
        HIPCHK(D.depth_tiles.reserve(4 * (size_t)ntiles + 4 + (size_t)(ntiles + 1023) / 1024));
        unsigned int *t_agg = D.depth_tiles.p, *t_first = t_agg + ntiles, *t_front = t_first + ntiles;
        int32_t *t_flagged = (int32_t*)(t_front + ntiles), *n_flagged = D.flags.p + 7;  // (flags were just zeroed; read back with them)
        DepthTiles T{t_agg, t_first, n_flagged, t_flagged, t_front, ablate_switch(c, "SQUID_D2_ABLATE"), nullptr, nullptr};
        static const bool fuse = ST_THREADS == 64 && !(std::getenv("SQUID_NO_FUSE") && std::atoi(std::getenv("SQUID_NO_FUSE")));
        D.p2_valid_n = -1;
        if (fuse && D.r_pack_n >= n && n < 0xffffffffll) {
            // k_pass2w: the depth sums of the tiles that lie inside one node AND the first pass of the edge stage, in one read of the records
            // (record rows 32 + keep 1 per record, first and last block 16 each); k_depth2 then sweeps only the tiles left on the list
            HIPCHK(D.p2_list.reserve((size_t)n)); HIPCHK(D.p2_words.reserve((size_t)ntiles + 4)); HIPCHK(D.bp_key.reserve((size_t)((n + 255) / 256) + 1));
            HIPCHK(hipMemsetAsync(D.p2_words.p, 0, 8, s));
            P2Args A2{D.keep.p, D.r_break.p, T, a_mc, a_ms, a_oc, a_os, D.flags.p, stripes, D.p2_words.p + 2, D.p2_words.p + 1, D.bp_key.p, D.p2_list.p, D.p2_words.p, std::getenv("SQUID_EDGES_ALL") ? 1 : 0};
            { EvTimer t(c, "k_pass2w", 33.0 * n + 32.0 * n);
              static const int p2_waves = std::getenv("SQUID_P2_WAVES") ? std::atoi(std::getenv("SQUID_P2_WAVES")) : 6;  // (C3: 3 waves per SIMD 0.77 ms, 4 0.64, 5 0.56, 6 0.51)
              if (p2_waves == 3) hipLaunchKernelGGL((k_pass2w<3>), dim3(ntiles), dim3(64), 0, s, R, nv, A2);
              else if (p2_waves == 4) hipLaunchKernelGGL((k_pass2w<4>), dim3(ntiles), dim3(64), 0, s, R, nv, A2);
              else if (p2_waves == 7) hipLaunchKernelGGL((k_pass2w<7>), dim3(ntiles), dim3(64), 0, s, R, nv, A2);
              else if (p2_waves == 8) hipLaunchKernelGGL((k_pass2w<8>), dim3(ntiles), dim3(64), 0, s, R, nv, A2);
              else if (p2_waves == 5) hipLaunchKernelGGL((k_pass2w<5>), dim3(ntiles), dim3(64), 0, s, R, nv, A2);
              else if (p2_waves == 6) hipLaunchKernelGGL((k_pass2w<6>), dim3(ntiles), dim3(64), 0, s, R, nv, A2);
              else hipLaunchKernelGGL((k_pass2w<6>), dim3(ntiles), dim3(64), 0, s, R, nv, A2); }
            D.p2_valid_n = n; D.bp_key_n = n;
            T.list = D.p2_words.p + 2; T.n_list = D.p2_words.p + 1;
            { EvTimer t(c, "k_depth2", 0);  // (the tiles k_pass2w left: their bytes are part of its read)
              hipLaunchKernelGGL(k_depth2<false>, dim3((unsigned)std::min(ntiles, 1024)), dim3(ST_THREADS), 0, s, R, nv, D.keep.p, D.r_break.p, T, ntiles, a_mc, a_ms, a_oc, a_os, a_ap, a_am, D.flags.p, stripes); }
        } else {
        // keep 1 + block offset 4 + refid 4 per record; 16 B per block of a consumed record (the cursor never leaves the kernel)
        EvTimer t(c, "k_depth2", 9.0 * n + 16.0 * D.nb * ((double)D.k1 / (double)n));
        hipLaunchKernelGGL(k_depth2<false>, dim3(ntiles), dim3(ST_THREADS), 0, s, R, nv, D.keep.p, D.r_break.p, T, ntiles, a_mc, a_ms, a_oc, a_os, a_ap, a_am, D.flags.p, stripes);
        }
        { EvTimer t(c, "k_depth_check+fix", 8.0 * ntiles);
          unsigned int* t_part = (unsigned int*)(t_flagged + ntiles) + 4;
          hipLaunchKernelGGL(k_depth_partial, dim3((ntiles + 1023) / 1024), dim3(1024), 0, s, t_agg, ntiles, t_part);
          hipLaunchKernelGGL(k_depth_check, dim3((ntiles + 1023) / 1024), dim3(1024), 0, s, t_agg, t_part, t_first, ntiles, n_flagged, t_flagged, t_front, ntiles);
          hipLaunchKernelGGL(k_depth2<true>, dim3(64), dim3(ST_THREADS), 0, s, R, nv, D.keep.p, D.r_break.p, T, ntiles, a_mc, a_ms, a_oc, a_os, a_ap, a_am, D.flags.p, stripes);
          if (nn) hipLaunchKernelGGL(k_fold_stripes, dim3((nn + 255) / 256), dim3(256), 0, s, nn, a_mc, a_ms, a_oc, a_os, acc); }
    }
    D.pin.reset();
    int32_t* h = D.pin.take_n<int32_t>(6 * (size_t)nn + 8 + NSTRIPE);
    if (!h) return fail(c, SQ_E_HIP, "hipHostMalloc failed");
    int32_t *mc = h, *ms = h + nn, *oc = h + 2 * nn, *os = h + 3 * nn, *ap = h + 4 * nn, *am = h + 5 * nn, *hf = h + 6 * nn, *hs = hf + 8;
    if (nn) HIPCHK(hipMemcpyAsync(h, acc, 6 * (size_t)nn * 4, hipMemcpyDeviceToHost, s));  // folded sums + the two bound arrays
    HIPCHK(hipMemcpyAsync(hf, D.flags.p, 8 * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(hs, stripes, NSTRIPE * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (hf[0] & 1) return fail(c, SQ_E_UNSORTED, "concordant stream is not coordinate sorted (depth cursor left its chromosome)");
    c->timer.add("depth_tiles_corrected", 0.0, (double)hf[7], 1);  // (a count, not bytes: tiles whose sweep cursor came from further ahead, k_depth2<true>)
    need_exact_other = hf[0] & 2;
    amb_plus.assign(ap, ap + nn); amb_minus.assign(am, am + nn);
    support.assign(nn * 2, 0);
    sumlen.assign(nn * 2, 0);
    for (int i = 0; i < nn; ++i) { support[i] = mc[i]; sumlen[i] = ms[i]; support[nn + i] = oc[i]; sumlen[nn + i] = os[i]; }
    // sumlen layout: [0,nn) main, [nn,2nn) other; element 2nn = |ReadsOther|
    long long n_other = 0;
    for (int i = 0; i < NSTRIPE; ++i) n_other += hs[i];
    support.push_back((int32_t)std::min<long long>(n_other, INT32_MAX));
    (void)other_len;
    return SQ_OK;
}

// K4+K5: locate blocks, emit raw edges, count equal keys
int dev_concordant_edges(sq_ctx* c, const std::vector<Node>& nodes, std::vector<Edge>& unique_edges) {
    DeviceRecords& D = *c->dev;
    hipStream_t s = c->stream;
    const int64_t n = D.n;
    unique_edges.clear();
    if (n == 0) return SQ_OK;
    if (D.nv.n != (int)nodes.size()) return fail(c, SQ_E_ARG, "internal: dev_upload_nodes first");
    const NodeView nv = D.nv;
    RecView R = D.view();
    EdgeParams2 ep{c->P.concord_dist_pos, c->P.concord_dist_idx, ablate_switch(c, "SQUID_EDGES_ABLATE")};  // (debugging: parts of k_edges switched off, timing only)
    D.pin.reset();
    int32_t* h = D.pin.take_n<int32_t>(8 + NSTRIPE);
    if (!h) return fail(c, SQ_E_HIP, "hipHostMalloc failed");
    HIPCHK(D.stripes.reserve(NSTRIPE));
    if (n >= 0xffffffffll) return fail(c, SQ_E_CAPACITY, "more than 2^32 records");
    HIPCHK(D.scratch_a.reserve((size_t)n));  // the work list of the edge stage
    while (D.h_slots < (1u << 28) && (size_t)D.h_slots < 4 * nodes.size()) D.h_slots <<= 1;  // (unique concordant edges ~ number of segments)
    for (;;) {  // the table starts small and grows when it fills up (unique edges are few)
        const uint32_t slots = D.h_slots;
        HIPCHK(D.h_key.reserve(slots)); HIPCHK(D.h_val.reserve(slots));
        HIPCHK(D.okey.reserve(slots)); HIPCHK(D.oval.reserve(slots));
        HIPCHK(hipMemsetAsync(D.h_key.p, 0xff, (size_t)slots * 8, s));
        HIPCHK(hipMemsetAsync(D.h_val.p, 0, (size_t)slots * 4, s));
        HIPCHK(hipMemsetAsync(D.flags.p, 0, 8 * 4, s));
        HIPCHK(hipMemsetAsync(D.stripes.p, 0, NSTRIPE * 4, s));
        // pass 1 reads keep 1 + flag 2 + refid, mate refid, mate pos, block offset 4 each = 19 B per record and 16 B per block;
        // pass 2 sees the ~1 % of the records that pass 1 could not clear (its bytes are not counted)
        uint32_t* list = (uint32_t*)D.scratch_a.p;
        int32_t* count = D.flags.p + 6;
        HIPCHK(D.bp_key.reserve((size_t)((n + 255) / 256) + 1));
        if (D.p2_valid_n == n) {  // the depth stage of this pass has made the work list and the cursor keys already (k_pass2w)
            list = D.p2_list.p;
            HIPCHK(hipMemcpyAsync(count, D.p2_words.p, 4, hipMemcpyDeviceToDevice, s));
        } else {
            EvTimer t(c, "k_edges_near", 23.0 * n + 16.0 * D.nb);  // (+ pos 4 for the breakpoint-cursor keys)
            hipLaunchKernelGGL(k_edges_near, grid_for(n, 256), dim3(256), 0, s, R, nv, D.keep.p, D.bp_key.p, list, count, std::getenv("SQUID_EDGES_ALL") ? 1 : 0);
        }
        D.bp_key_n = n;
        { EvTimer t(c, "k_edges", 0);
          hipLaunchKernelGGL(k_edges, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, s, R, nv, ep, D.keep.p, list, count, D.h_key.p, D.h_val.p, slots - 1, D.flags.p, D.stripes.p); }
        // compact right away (wasted only if the table turns out to have overflowed): one synchronisation for both
        { EvTimer t(c, "k_hash_compact", 12.0 * slots); hipLaunchKernelGGL(k_hash_compact, grid_for(slots, 256), dim3(256), 0, s, D.h_key.p, D.h_val.p, slots, D.flags.p + 4, D.okey.p, D.oval.p); }
        HIPCHK(hipMemcpyAsync(h, D.flags.p, 8 * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(h + 8, D.stripes.p, NSTRIPE * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (!(h[0] & 4)) break;
        if (D.h_slots >= (1u << 28)) return fail(c, SQ_E_CAPACITY, "edge hash table full");
        D.h_slots <<= 2;
    }
    D.p2_valid_n = -1;  // (the list belongs to this pass)
    if (h[0] & 8) return fail(c, SQ_E_ASSERT, "edge node index out of range (the reference asserts at SegmentGraph.cpp:1617)");
    if (h[0] & 16) return fail(c, SQ_E_CAPACITY, "record with more aligned blocks than the edge kernel handles");
    long long n_raw = 0;
    for (int i = 0; i < NSTRIPE; ++i) n_raw += h[8 + i];
    c->counts.n_raw_edges = n_raw;
    const int cnt = h[4];
    unsigned long long* hk = D.pin.take_n<unsigned long long>((size_t)cnt);
    uint32_t* hv = D.pin.take_n<uint32_t>((size_t)cnt);
    if (!hk || !hv) return fail(c, SQ_E_HIP, "hipHostMalloc failed");
    if (cnt) { HIPCHK(hipMemcpyAsync(hk, D.okey.p, (size_t)cnt * 8, hipMemcpyDeviceToHost, s)); HIPCHK(hipMemcpyAsync(hv, D.oval.p, (size_t)cnt * 4, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s)); }
    unique_edges.resize(cnt);
    for (int i = 0; i < cnt; ++i) {
        Edge e;
        e.a = (int32_t)(hk[i] >> 32); e.b = (int32_t)((hk[i] & 0xffffffffull) >> 2); e.ha = (hk[i] >> 1) & 1; e.hb = hk[i] & 1; e.w = (int32_t)hv[i]; e.gw = 0;
        unique_edges[i] = e;
    }
    c->counts.n_unique_edges = cnt;
    return SQ_OK;
}

// K8: connected components; label = rank of the component's smallest node id
int dev_connected_components(sq_ctx* c, int n_nodes, const std::vector<Edge>& edges, std::vector<int32_t>& label) {
    DeviceRecords& D = *c->dev;
    hipStream_t s = c->stream;
    const int m = (int)edges.size();
    label.assign(n_nodes, 0);
    if (n_nodes == 0) return SQ_OK;
    std::vector<int32_t> eab(2 * (size_t)m);
    for (int i = 0; i < m; ++i) { eab[i] = edges[i].a; eab[(size_t)m + i] = edges[i].b; }
    HIPCHK(D.scratch_a.reserve(n_nodes)); HIPCHK(D.scratch_b.reserve(n_nodes)); HIPCHK(D.scratch_c.reserve(n_nodes)); HIPCHK(D.acc_b.reserve(2 * (size_t)std::max(m, 1)));
    HIPCHK(D.acc_c.reserve(n_nodes));
    if (m) HIPCHK(hipMemcpy(D.acc_b.p, eab.data(), eab.size() * 4, hipMemcpyHostToDevice));
    int32_t *d_ea = D.acc_b.p, *d_eb = D.acc_b.p + m;
    {
        EvTimer t(c, "k_cc", 8.0 * m + 16.0 * n_nodes);
        hipLaunchKernelGGL(k_cc_init, grid_for(n_nodes, 256), dim3(256), 0, s, n_nodes, D.scratch_a.p);
        if (m) hipLaunchKernelGGL(k_cc_union, grid_for(m, 256), dim3(256), 0, s, m, d_ea, d_eb, D.scratch_a.p);
        hipLaunchKernelGGL(k_cc_flatten, grid_for(n_nodes, 256), dim3(256), 0, s, n_nodes, D.scratch_a.p, D.scratch_b.p);
        HIPCHK((device_scan<OpSum, true>(s, n_nodes, FArr{D.scratch_b.p}, D.scratch_c.p, D.spine, nullptr)));
        hipLaunchKernelGGL(k_cc_label, grid_for(n_nodes, 256), dim3(256), 0, s, n_nodes, D.scratch_a.p, D.scratch_c.p, D.acc_c.p);
    }
    HIPCHK(hipMemcpyAsync(label.data(), D.acc_c.p, n_nodes * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return SQ_OK;
}

// K9: batched exact ordering of components with at most ORD_NMAX nodes
int dev_order_small(sq_ctx* c, const std::vector<SmallProblem>& probs, const std::vector<int32_t>& edges5, std::vector<int32_t>& out_mask, std::vector<int32_t>& out_order,
                    int nmax) {
    hipStream_t s = c->stream;
    const int np = (int)probs.size();
    out_mask.assign(np, 0);
    out_order.assign((size_t)np * ORD_NMAX, 0);
    if (!np) return SQ_OK;
    if (nmax > ORD_NMAX) return fail(c, SQ_E_ARG, "dev_order_small: nmax too large");
    DeviceRecords& D = *c->dev;
    // one packed upload (problems | edges) and one packed download (masks | orders) through page-locked staging
    static_assert(sizeof(SmallProblem) == 12, "SmallProblem travels as three ints");
    const size_t in_words = 3 * (size_t)np + edges5.size(), out_words = (size_t)np * (1 + ORD_NMAX);
    DBuf<int32_t>&din = D.ord_e, &dout = D.ord_o, &dval = D.ord_v;
    HIPCHK(din.reserve(in_words + 1)); HIPCHK(dout.reserve(out_words)); HIPCHK(dval.reserve(np));
    D.pin.reset();
    int32_t *hin = D.pin.take_n<int32_t>(in_words), *hout = D.pin.take_n<int32_t>(out_words);
    if (!hin || !hout) return fail(c, SQ_E_HIP, "hipHostMalloc failed");
    std::memcpy(hin, probs.data(), (size_t)np * 12);
    if (edges5.size()) std::memcpy(hin + 3 * (size_t)np, edges5.data(), edges5.size() * 4);
    HIPCHK(hipMemcpyAsync(din.p, hin, in_words * 4, hipMemcpyHostToDevice, s));
    { EvTimer t(c, "k_order_small", 0);
      hipLaunchKernelGGL(k_order_small, dim3(np), dim3(256), 0, s, (const SmallProblem*)din.p, din.p + 3 * (size_t)np, dout.p, dout.p + np, dval.p); }
    HIPCHK(hipMemcpyAsync(hout, dout.p, out_words * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    out_mask.assign(hout, hout + np);
    out_order.assign(hout + np, hout + out_words);
    return SQ_OK;
}

// K9b: components of 9..19 nodes; status[i] != 0 => the kernel's capacities were exceeded, the caller solves problem i on the host
int dev_order_mid(sq_ctx* c, const std::vector<SmallProblem>& probs, const std::vector<int32_t>& edges5, std::vector<int32_t>& out_mask, std::vector<int32_t>& out_order,
                  std::vector<int32_t>& out_value, std::vector<int32_t>& out_status, bool own_stream) {
    // own_stream: called from a helper thread while the caller's thread runs k_order_small on the library stream -- the two batches are
    // independent and a dozen workgroups each: side by side they take the time of the longer one (C3: 1.25 + 0.48 -> 1.25 ms)
    hipStream_t s = c->stream;
    if (own_stream) {
        HIPCHK(hipSetDevice(c->P.device));
        if (!c->dev->order_stream) HIPCHK(hipStreamCreateWithFlags(&c->dev->order_stream, hipStreamNonBlocking));
        s = c->dev->order_stream;
    }
    const int np = (int)probs.size();
    out_mask.assign(np, 0); out_value.assign(np, -1); out_status.assign(np, 1);
    out_order.assign((size_t)np * OM_NMAX, 0);
    if (!np) return SQ_OK;
    DeviceRecords& D = *c->dev;
    const size_t in_words = 3 * (size_t)np + edges5.size(), out_words = (size_t)np * (3 + OM_NMAX);
    DBuf<int32_t>&din = D.ord_me, &dout = D.ord_mo;
    HIPCHK(din.reserve(in_words + 1)); HIPCHK(dout.reserve(out_words));
    std::vector<int32_t> hin(in_words), hout(out_words);
    std::memcpy(hin.data(), probs.data(), (size_t)np * 12);
    if (edges5.size()) std::memcpy(hin.data() + 3 * (size_t)np, edges5.data(), edges5.size() * 4);
    HIPCHK(hipMemcpyAsync(din.p, hin.data(), in_words * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipFuncSetAttribute((const void*)k_order_mid, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(OrdMidLds)));
    { EvTimer t(c, "k_order_mid", 0, s);
      hipLaunchKernelGGL(k_order_mid, dim3(np), dim3(256), sizeof(OrdMidLds), s, (const SmallProblem*)din.p, din.p + 3 * (size_t)np, dout.p, dout.p + 3 * (size_t)np, dout.p + np, dout.p + 2 * (size_t)np); }
    HIPCHK(hipMemcpyAsync(hout.data(), dout.p, out_words * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    out_mask.assign(hout.begin(), hout.begin() + np);
    out_value.assign(hout.begin() + np, hout.begin() + 2 * (size_t)np);
    out_status.assign(hout.begin() + 2 * (size_t)np, hout.begin() + 3 * (size_t)np);
    out_order.assign(hout.begin() + 3 * (size_t)np, hout.end());
    return SQ_OK;
}

// K10: concordant-fragment support of every breakpoint
int dev_breakpoint_support(sq_ctx* c, const std::vector<std::pair<int, int>>& bps, std::vector<int32_t>& coverage, int cur_prev, BpBoundary* bb, bool raw_diff) {
    DeviceRecords& D = *c->dev;
    hipStream_t s = c->stream;
    const int64_t n = D.n;
    const int nb = (int)bps.size();
    coverage.assign(raw_diff ? nb + 1 : nb, 0);
    if (bb) { *bb = BpBoundary(); bb->cur_end = cur_prev; }
    if (!nb || !n) return SQ_OK;
    std::vector<int32_t> bc(nb), bp(nb);
    for (int i = 0; i < nb; ++i) { bc[i] = bps[i].first; bp[i] = bps[i].second; }
    HIPCHK(D.acc_b.reserve(2 * (size_t)nb)); HIPCHK(D.acc_c.reserve(nb + 1));
    bc.insert(bc.end(), bp.begin(), bp.end());  // one packed upload: chr | pos
    HIPCHK(hipMemcpy(D.acc_b.p, bc.data(), 2 * (size_t)nb * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemsetAsync(D.acc_c.p, 0, (nb + 1) * 4, s));
    HIPCHK(hipMemsetAsync(D.flags.p, 0, 8 * 4, s));
    RecView R = D.view();
    const int n_ref = (int)c->ref_len.size();
    if (D.nv.n_ref != n_ref || !D.nv.bucket_off) return fail(c, SQ_E_ARG, "internal: breakpoint support before the node table was uploaded");
    std::vector<int32_t> bo(n_ref + 1, 0);
    for (int k = 0; k < n_ref; ++k) bo[k + 1] = bo[k] + (int32_t)(((int64_t)std::max(c->ref_len[k], 1) + (1 << NODE_BUCKET_SHIFT) - 1) >> NODE_BUCKET_SHIFT);
    const int total = bo[n_ref];
    HIPCHK(D.bp_bucket.reserve(std::max(total, 1)));
    BPView B{nb, D.acc_b.p, D.acc_b.p + nb, c->P.concord_dist_pos, D.bp_bucket.p, D.nv.bucket_off, n_ref};
    for (int i = 0; i < nb; ++i) if (bc[i] < 0 || bc[i] >= n_ref) return fail(c, SQ_E_ARG, "breakpoint on an unknown reference");
    if (total) hipLaunchKernelGGL(k_bp_buckets, dim3((total + 255) / 256), dim3(256), 0, s, B, n_ref, total, D.bp_bucket.p);
    HIPCHK(D.bp_ev.reserve(nb + 1)); HIPCHK(D.bp_before.reserve(nb + 1)); HIPCHK(D.bp_end.reserve(nb + 1)); HIPCHK(D.bp_valid.reserve(nb + 1));
    HIPCHK(hipMemsetAsync(D.bp_ev.p, 0xFF, (nb + 1) * 4, s));

    int32_t* agg = D.flags.p + 28;  // [28] max of m over all records, [29] cursor of a walk that ran into the end of the stream
    const int32_t minus1 = -1;
    HIPCHK(hipMemcpyAsync(agg + 1, &minus1, 4, hipMemcpyHostToDevice, s));
    {   // class 1 + refid, pos, mate refid, mate pos, end 4 each + flag 2 per record; nothing written per record
        const int ntiles = (int)((n + ST_TILE - 1) / ST_TILE), nkeys = (int)((n + 255) / 256);
        HIPCHK(D.bp_key.reserve((size_t)nkeys + 1)); HIPCHK(D.bp_front.reserve((size_t)ntiles + 1 + (size_t)(nkeys + 1023) / 1024 + 1));
        if (D.bp_key_n != n) { hipLaunchKernelGGL(k_bp_keys, dim3(nkeys), dim3(256), 0, s, R, D.cls.p, D.bp_key.p); D.bp_key_n = n; }  // (normally left by the edge stage)
        { EvTimer t(c, "k_bp_key_prefix", 8.0 * nkeys * 2 + 8.0 * ntiles);
          const int ng = (nkeys + 1023) / 1024;
          unsigned long long* part = D.bp_front.p + ntiles + 1;
          hipLaunchKernelGGL(k_bp_key_reduce, dim3(ng), dim3(1024), 0, s, D.bp_key.p, nkeys, part);
          hipLaunchKernelGGL(k_bp_key_scan, dim3(ng), dim3(1024), 0, s, D.bp_key.p, nkeys, part, ST_TILE / 256, D.bp_front.p, ntiles); }
        EvTimer t(c, "k_bp2", 23.0 * n);
        hipLaunchKernelGGL(k_bp2, dim3(ntiles), dim3(ST_THREADS), 0, s, R, B, D.cls.p, cur_prev, D.bp_front.p, ntiles, D.bp_ev.p, D.bp_before.p, D.acc_c.p, agg);
    }
    { EvTimer t(c, "k_bp_walk", 0);
      hipLaunchKernelGGL(k_bp_walk2<false>, dim3(nb), dim3(64), 0, s, R, B, D.cls.p, cur_prev, D.bp_ev.p, D.bp_before.p, D.bp_end.p, D.bp_valid.p, D.acc_c.p, agg + 1);
      hipLaunchKernelGGL(k_bp_chain, dim3(1), dim3(64), 0, s, nb, D.bp_ev.p, D.bp_end.p, D.bp_valid.p);
      hipLaunchKernelGGL(k_bp_walk2<true>, dim3(nb), dim3(64), 0, s, R, B, D.cls.p, cur_prev, D.bp_ev.p, D.bp_before.p, D.bp_end.p, D.bp_valid.p, D.acc_c.p, agg + 1); }
    unsigned long long hcnt[3] = {0, 0, 0};
    if (bb) {
        HIPCHK(D.okey.reserve(4));
        HIPCHK(hipMemsetAsync(D.okey.p, 0, 16, s));
        hipLaunchKernelGGL(k_bp_boundary, dim3(1024), dim3(256), 0, s, D.cls.p, n, D.bp_ev.p, D.okey.p);
        HIPCHK(hipMemcpyAsync(hcnt, D.okey.p, 16, hipMemcpyDeviceToHost, s));
    }
    std::vector<int32_t> diff(nb + 1);
    int32_t hagg[2] = {0, 0}, first_ev = -1;
    HIPCHK(hipMemcpyAsync(&first_ev, D.bp_ev.p, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(diff.data(), D.acc_c.p, (nb + 1) * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(hagg, agg, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (bb) {
        bb->cur_end = hagg[1] >= 0 ? hagg[1] : std::max(hagg[0], cur_prev);
        bb->has_p3 = hagg[0] != INT_MIN;
        bb->has_event = first_ev >= 0;
        bb->n_p3 = (int64_t)hcnt[0]; bb->absorb = (int64_t)hcnt[1];
    }
    if (raw_diff) { coverage = diff; return SQ_OK; }
    int run = 0;
    for (int i = 0; i < nb; ++i) { run += diff[i]; coverage[i] = run; }
    return SQ_OK;
}

// serial restatement of the cursor walk on the host (SQUID_BP_HOST=1): the pass-3 records' (chr, start, end) are
// pulled back in stream order.  Debug cross-check of k_bp_walk only; never taken by default.
struct FP3 { const uint8_t* cls; __device__ int operator()(int64_t i) const { return (cls[i] & C_P3) ? 1 : 0; } };
__global__ void k_p3_gather(RecView R, const uint8_t* cls, const int32_t* rank3, int32_t* o_chr, int32_t* o_st, int32_t* o_en) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R.n || !(cls[r] & C_P3)) return;
    int c = R.refid[r], st = R.pos[r];
    if (!(R.flag[r] & 0x8) && R.mrefid[r] == c) st = R.mpos[r];
    o_chr[rank3[r]] = c; o_st[rank3[r]] = st; o_en[rank3[r]] = R.endpos[r];
}
int dev_breakpoint_support_exact(sq_ctx* c, const std::vector<std::pair<int, int>>& bps, std::vector<int32_t>& coverage) {
    DeviceRecords& D = *c->dev;
    hipStream_t s = c->stream;
    const int64_t n = D.n;
    RecView R = D.view();
    int32_t* tot = D.flags.p + 8;
    HIPCHK(D.scratch_b.reserve(n));
    HIPCHK((device_scan<OpSum, true>(s, n, FP3{D.cls.p}, D.scratch_b.p, D.spine, tot)));
    int32_t n3 = 0;
    HIPCHK(hipMemcpyAsync(&n3, tot, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(D.scratch_a.reserve(std::max(n3, 1))); HIPCHK(D.scratch_c.reserve(std::max(n3, 1))); HIPCHK(D.part_prev.reserve(std::max(n3, 1)));
    hipLaunchKernelGGL(k_p3_gather, grid_for(n, 256), dim3(256), 0, s, R, D.cls.p, D.scratch_b.p, D.scratch_a.p, D.scratch_c.p, D.part_prev.p);
    std::vector<int32_t> ch(n3), st(n3), en(n3);
    HIPCHK(hipMemcpyAsync(ch.data(), D.scratch_a.p, n3 * 4, hipMemcpyDeviceToHost, s)); HIPCHK(hipMemcpyAsync(st.data(), D.scratch_c.p, n3 * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(en.data(), D.part_prev.p, n3 * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    const size_t nb = bps.size();
    coverage.assign(nb, 0);
    size_t ind = 0;
    for (int i = 0; i < n3 && ind != nb; ++i) {
        if (ch[i] > bps[ind].first || (ch[i] == bps[ind].first && st[i] > bps[ind].second + c->P.concord_dist_pos)) ind++;
        for (size_t j = ind; j < nb; ++j) {
            if (ch[i] == bps[j].first && st[i] <= bps[j].second && en[i] > bps[j].second) coverage[j]++;
            else if (ch[i] < bps[j].first || (ch[i] == bps[j].first && en[i] <= bps[j].second)) break;
        }
    }
    c->timer.add("host_bp_cursor_exact", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), 0);
    return SQ_OK;
}

#include "sq_graph_kernels.inc"

}  // namespace sq
