// C-ABI entry points of libsquid_hip.so (include/squid_hip.h) and the stage pipeline behind them.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <future>
#include <memory>

#include <sys/stat.h>
#include <sys/mman.h>
#include <fcntl.h>
#include <unistd.h>

#include "sq_internal.h"
#include <malloc.h>
#include "sq_parsort.h"

namespace sq {

// sq_ingest_files decodes the chimeric file on a second thread while the calling thread may report its own errors into c->err:
// the helper's messages go to c->chim_err (this thread-local names it) and reach c->err when chim_join collects the result
static thread_local std::string* tl_err_sink = nullptr;
// a helper thread of the library sends its error texts to a string of its own (the thread that owns the context moves it into c->err once
// the helper has been joined): the chimeric decode of sq_ingest_files, the planner thread of the GPU reader
ErrSink::ErrSink(std::string* to) { tl_err_sink = to; }
ErrSink::~ErrSink() { tl_err_sink = nullptr; }
int fail(sq_ctx* c, int code, const std::string& msg) {
    if (tl_err_sink) *tl_err_sink = msg;
    else if (c) c->err = msg;
    return code;
}
int chim_join_names(sq_ctx* c) {
    if (!c->chim_names_future.valid()) return SQ_OK;
    const int rc = c->chim_names_future.get();  // (once: the future is invalid afterwards)
    if (rc) { c->err = c->chim_err; return rc; }
    return SQ_OK;
}
int chim_join(sq_ctx* c) {
    if (!c->chim_future.valid()) return SQ_OK;
    const int rc_names = chim_join_names(c);  // (a concordant file without records never reached the parse)
    const int rc = c->chim_future.get();
    if (rc) { c->err = c->chim_err; return rc; }
    if (rc_names) return rc_names;
    return dev_chim_finalize(c, c->chim_dead);
}

int Timer::slot(const char* name) {
    for (size_t i = 0; i < names.size(); ++i) if (names[i] == name || !std::strcmp(names[i], name)) return (int)i;
    names.push_back(name); ms.push_back(0); bytes.push_back(0); busy.push_back(0); launches.push_back(0);
    return (int)names.size() - 1;
}
void Timer::add(const char* name, double ms_, double bytes_, int64_t n) {
    int s = slot(name);
    ms[s] += ms_; bytes[s] += bytes_; launches[s] += n;
}
void Timer::add_busy(const char* name, double ms_) { busy[slot(name)] += ms_; }
void Timer::clear() { names.clear(); ms.clear(); bytes.clear(); busy.clear(); launches.clear(); }

void GraphSnap::take(const std::vector<Node>& N, const std::vector<Edge>& E, const std::vector<int32_t>* lab) {
    const size_t n = N.size(), m = E.size();
    chr.resize(n); pos.resize(n); len.resize(n); support.resize(n); depth.resize(n); label.assign(n, 0);
    for (size_t i = 0; i < n; ++i) { chr[i] = N[i].chr; pos[i] = N[i].pos; len[i] = N[i].len; support[i] = N[i].support; depth[i] = N[i].depth; }
    if (lab) label = *lab;
    ind1.resize(m); ind2.resize(m); weight.resize(m); gweight.resize(m); h1.resize(m); h2.resize(m);
    for (size_t i = 0; i < m; ++i) { ind1[i] = E[i].a; ind2[i] = E[i].b; weight[i] = E[i].w; gweight[i] = E[i].gw; h1[i] = E[i].ha; h2[i] = E[i].hb; }
}
void GraphSnap::view(sq_graph* g) const {
    g->n_nodes = (int32_t)chr.size(); g->n_edges = (int32_t)ind1.size();
    g->chr = chr.data(); g->pos = pos.data(); g->len = len.data(); g->support = support.data(); g->label = label.data(); g->avgdepth = depth.data();
    g->ind1 = ind1.data(); g->ind2 = ind2.data(); g->weight = weight.data(); g->groupweight = gweight.data(); g->head1 = h1.data(); g->head2 = h2.data();
}

int dev_breakpoint_support_exact(sq_ctx* c, const std::vector<std::pair<int, int>>& bps, std::vector<int32_t>& coverage);


// ---- exchange payloads (chromosome-sharded runs): plain little-endian PODs appended to a byte vector
struct Packer {
    std::vector<uint8_t>& b;
    explicit Packer(std::vector<uint8_t>& buf) : b(buf) { b.clear(); }
    template <class T> void put(const T& v) { const uint8_t* p = (const uint8_t*)&v; b.insert(b.end(), p, p + sizeof(T)); }
    template <class T> void put_vec(const std::vector<T>& v) { put<int64_t>((int64_t)v.size()); const uint8_t* p = (const uint8_t*)v.data(); b.insert(b.end(), p, p + v.size() * sizeof(T)); }
};
struct Unpacker {
    const std::vector<uint8_t>& b; size_t at = 0; bool ok = true;
    explicit Unpacker(const std::vector<uint8_t>& buf) : b(buf) {}
    template <class T> T get() { T v{}; if (at + sizeof(T) > b.size()) { ok = false; return v; } std::memcpy(&v, b.data() + at, sizeof(T)); at += sizeof(T); return v; }
    template <class T> void get_vec(std::vector<T>& v) {
        int64_t n = get<int64_t>();
        if (!ok || n < 0 || at + (size_t)n * sizeof(T) > b.size()) { ok = false; v.clear(); return; }
        v.resize((size_t)n);
        if (n) std::memcpy(v.data(), b.data() + at, (size_t)n * sizeof(T));
        at += (size_t)n * sizeof(T);
    }
};
enum : int32_t { X_DEDUP = 0x51a0, X_STREAM, X_SEEDS, X_GRAPH, X_BPSUP, X_OTHER };  // payload tags (a mismatch means the ranks are out of step)

// locals of build_graph that have to survive an exchange
struct GraphBuild {
    int stage = 0;
    std::shared_ptr<SegPlan> plan;
    std::vector<Blk> disc;
    int64_t n_break = 0;
    int64_t trigger_last = 0;
    std::vector<Node> seeds;
    std::vector<Edge> raw, conc, before;  // before: the edges in front of FilterEdges (the filter is repeated when a depth decision was ambiguous)
    std::vector<uint8_t> keep;          // KeepEdge of FilterbyInterleaving
    std::vector<int32_t> sup, amb_plus, amb_minus;
    std::vector<int64_t> sl;
    bool tiny_boundary = false;
    std::vector<int32_t> first_chr;   // sharded: chromosome of every rank's first kept record (-1: none)
    // sharded seed exchange: this shard's seeds under the three possible pasts (see stage 3)
    std::vector<Node> seedsA, seedsB, seedsC;
    std::vector<int32_t> sensB;
    bool hasC = false;
    Node seedC{0, 0, 0, 0, 0.0};
    std::future<double> clusters;     // segment_clusters running next to the record kernels
    ~GraphBuild() { if (clusters.valid()) clusters.wait(); }  // (a pooled task's future does not wait by itself; the task writes into this object)
};

static int need_exchange(sq_ctx* c) {
    c->x_pending = true; c->x_ready = false;
    return SQ_NEED_EXCHANGE;
}
// the gathered payloads of the exchange that has just completed, or an error when the caller skipped it
static int take_exchange(sq_ctx* c, int32_t tag) {
    if (!c->x_ready || (int)c->xgot.size() != c->P.world_size) return fail(c, SQ_E_ARG, "sharded run: sq_exchange_unpack has not been called for the pending exchange");
    c->x_ready = false;
    for (const auto& v : c->xgot) { int32_t t = 0; if (v.size() < 4) return fail(c, SQ_E_ARG, "sharded run: short exchange payload"); std::memcpy(&t, v.data(), 4); if (t != tag) return fail(c, SQ_E_ARG, "sharded run: ranks are out of step (payload tag mismatch)"); }
    return SQ_OK;
}

static void pack_seeds(sq_ctx* c, const GraphBuild& g) {
    auto flat = [](const std::vector<Node>& v) { std::vector<int32_t> f; f.reserve(v.size() * 3); for (const Node& n : v) { f.push_back(n.chr); f.push_back(n.pos); f.push_back(n.len); } return f; };
    Packer pk(c->xbuf);
    pk.put<int32_t>(X_SEEDS);
    pk.put_vec(flat(g.seedsA));
    pk.put<int32_t>(c->shard.prior_kept ? 1 : 0);
    pk.put_vec(flat(g.seedsB));
    pk.put_vec(g.sensB);
    pk.put<int32_t>(g.hasC ? 1 : 0);
    pk.put<int32_t>(g.seedC.chr); pk.put<int32_t>(g.seedC.pos); pk.put<int32_t>(g.seedC.len);
    pk.put_vec(flat(g.seedsC));
}

// CompressNode .. MultiplyDisEdges of the constructor (SegmentGraph.cpp:117-122), shared by the STAR and the --bwa path
static int finish_graph(sq_ctx* c, bool host_filters) {
    int rc;
    {
        HostClock hc(c, host_filters ? "host_compress" : "wall_compress");
        rc = host_filters ? compress_nodes(c) : dev_compress_nodes(c);
        if (rc) return rc;
        if (c->keep_stages) c->snap[5].take(c->nodes, c->edges, nullptr);
        rc = host_filters ? further_compress(c) : dev_further_compress(c);
        if (rc == 2) rc = further_compress(c);  // a node with more discordant edges than the kernel's lists hold
        if (rc) return rc;
    }
    rc = dev_connected_components(c, (int)c->nodes.size(), c->edges, c->label);
    if (rc) return rc;
    multiply_discordant(c, false);
    c->snap[0].take(c->nodes, c->edges, &c->label);
    c->graph_built = true;
    c->gb.reset();
    // ExactBreakpoint only needs the final graph and the trimmed fragments: start it now, sq_call_sv collects it
    c->bp_early = std::make_shared<BPMap>();
    c->bp_future = c->pool->submit([c]() {
        const auto t0 = std::chrono::steady_clock::now();
        const int r2 = exact_breakpoints(c, *c->bp_early);
        c->bp_early_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return r2;
    });
    return SQ_OK;
}

// `squid --bwa`: BuildNode_BWA and RawEdges on the host over the decoded batch (sq_bwa.cpp), then the same edge reduction, filters,
// compression and component labelling as the STAR path (SegmentGraph.cpp:104-124 with UsingSTAR == false).  Node depths are exact
// here (no unstable sort in this mode's depth sweep), so the coverage-ratio test needs no bounds.
static int build_graph_bwa(sq_ctx* c) {
    HostClock wall(c, "wall_build_graph");
    if (c->shard.on) return fail(c, SQ_E_ARG, "--bwa input is not chromosome-sharded");
    c->graph_built = false; c->ordered = false;
    c->depth_bounds = false; c->depth_ambiguous = false;
    std::vector<Edge> raw;
    int rc = bwa_nodes_and_edges(c, raw);
    if (rc) return rc;
    // (counts.n_raw_edges: set by bwa_raw_edges -- the edges as the loop emitted them; `raw` arrives summed per stretch)
    c->edges.clear();
    { HostClock hc(c, "host_edge_reduce"); reduce_edges(raw, c->edges, c->pool ? std::min(c->pool->size() + 1, 32) : 1); }
    c->counts.n_unique_edges = (int64_t)c->edges.size();
    if (c->keep_stages) c->snap[2].take(c->nodes, c->edges, nullptr);
    static const bool host_filters = std::getenv("SQUID_HOST_FILTERS") != nullptr;
    {
        HostClock hc(c, host_filters ? "host_filters" : "wall_filters");
        std::vector<uint8_t> keep;
        if (host_filters) filter_by_weight(c); else if ((rc = dev_filter_by_weight(c))) return rc;
        if (c->keep_stages) c->snap[3].take(c->nodes, c->edges, nullptr);
        if (host_filters) filter_by_interleaving(c, keep); else if ((rc = dev_filter_by_interleaving(c, keep))) return rc;
        if (host_filters) filter_edges(c, keep); else if ((rc = dev_filter_edges(c, keep))) return rc;
        if (c->keep_stages) c->snap[4].take(c->nodes, c->edges, nullptr);
    }
    return finish_graph(c, host_filters);
}

static void drop_early_clusters(sq_ctx* c) { c->plan_early.reset(); c->disc_early.clear(); c->disc_early.shrink_to_fit(); c->clusters_early_ms = -1; }
static int build_graph(sq_ctx* c) {
    HostClock wall(c, "wall_build_graph");
    Shard& sh = c->shard;
    const int W = c->P.world_size, me = c->P.rank;
    if (!c->gb) c->gb = std::make_shared<GraphBuild>();
    GraphBuild& g = *c->gb;
    int rc;
    if (g.stage == 0) {
        c->graph_built = false;
        c->ordered = false;
        // the cluster table only needs the chimeric fragments: build it on a second thread next to the record kernels
        // (the table is a function of the chimeric fragments and ReadLen alone: built beside the ingest by sq_ingest_files, or by the first pass,
        // and KEPT by the context until the fragments change -- a later pass over the same records, sq_reset and a new set of -w/-r/-a, takes
        // it as it is; everything a pass writes into the plan is written again by the next one)
        if (c->clusters_early_ms >= 0 && c->plan_early) {
            g.plan = c->plan_early; g.disc = c->disc_early;
            std::promise<double> done;
            done.set_value(c->clusters_early_ms);
            g.clusters = done.get_future();
            c->clusters_early_ms = 0;  // (a second use costs nothing)
        } else g.clusters = c->pool->submit([c, &g]() { const double ms = segment_clusters(c, g.plan, g.disc); c->plan_early = g.plan; c->disc_early = g.disc; c->clusters_early_ms = 0; return ms; });
        int32_t last[4];
        rc = dev_classify(c, sh.on ? last : nullptr);
        if (rc) { (void)g.clusters.get(); return rc; }
        g.stage = 1;
        if (sh.on) {  // exchange 1: what the last passing records of every shard look like to ReadRec_t::Equal
            Packer pk(c->xbuf);
            pk.put<int32_t>(X_DEDUP);
            for (int i = 0; i < 4; ++i) pk.put<int32_t>(last[i]);
            return need_exchange(c);
        }
    }
    if (g.stage == 1) {
        if (sh.on) {
            rc = take_exchange(c, X_DEDUP);
            if (rc) return rc;
            sh.dedup_mask = 0;
            for (int r = 0; r < me; ++r) {
                Unpacker u(c->xgot[r]);
                u.get<int32_t>();
                for (int p = 0; p < 2; ++p) {
                    int32_t has = u.get<int32_t>(), empty = u.get<int32_t>();
                    if (has) { if (empty) sh.dedup_mask &= ~(1 << p); else sh.dedup_mask |= 1 << p; }
                }
            }
        }
        const double cl_ms = g.clusters.get();  // (k_pass1 needs the cluster table up front: it reads the records once, for everything)
        c->timer.add("host_cluster_table", cl_ms);
        long long other_max = INT64_MIN;
        int32_t first_kept[2] = {0, 0};
        {
            HostClock hc(c, "host_segment_prepare");
            rc = segment_scan(c, *g.plan, sh.on, g.trigger_last, other_max, first_kept);
            if (rc) return rc;
        }
        g.stage = 2;
        if (sh.on) {  // exchange 2: size, first record, running other-pair and last-cluster trigger of every local stream
            Packer pk(c->xbuf);
            pk.put<int32_t>(X_STREAM);
            pk.put<int64_t>(c->counts.n_kept_p1);
            pk.put<int32_t>(first_kept[0]); pk.put<int32_t>(first_kept[1]);
            pk.put<int64_t>(other_max);
            pk.put<int64_t>(g.trigger_last);
            return need_exchange(c);
        }
    }
    if (g.stage == 2) {
        if (sh.on) {
            rc = take_exchange(c, X_STREAM);
            if (rc) return rc;
            std::vector<int64_t> K(W), tl(W), om(W);
            std::vector<int32_t> fr(W), fp(W);
            for (int r = 0; r < W; ++r) {
                Unpacker u(c->xgot[r]);
                u.get<int32_t>();
                K[r] = u.get<int64_t>(); fr[r] = u.get<int32_t>(); fp[r] = u.get<int32_t>(); om[r] = u.get<int64_t>(); tl[r] = u.get<int64_t>();
                if (!u.ok) return fail(c, SQ_E_ARG, "sharded run: malformed stream payload");
            }
            g.first_chr.assign(W, -1);
            for (int r = 0; r < W; ++r) if (K[r] > 0) g.first_chr[r] = fr[r];
            sh.kept_before = 0; sh.kept_total = 0; sh.prior_kept = false; sh.other_seed = INT64_MIN; sh.has_terminal = false;
            for (int r = 0; r < W; ++r) {
                if (r < me) { sh.kept_before += K[r]; if (K[r] > 0) { sh.prior_kept = true; sh.other_seed = std::max<long long>(sh.other_seed, om[r]); } }
                if (r > me && K[r] > 0 && !sh.has_terminal) { sh.has_terminal = true; sh.term_refid = fr[r]; sh.term_pos = fp[r]; }
                sh.kept_total += K[r];
            }
            // B12: the consumed prefix ends one record behind the trigger of the globally last cluster
            if (g.trigger_last < 0) sh.n_break_global = std::min<int64_t>(sh.kept_total, 1);  // no discordant cluster at all
            else {
                int64_t T = sh.kept_total, off = 0;
                for (int r = 0; r < W; ++r) { if (tl[r] >= 0 && tl[r] < K[r]) { T = off + tl[r]; break; } off += K[r]; }
                sh.n_break_global = std::min<int64_t>(sh.kept_total, T + 2);
            }
        }
        {
            HostClock hc(c, "host_segment_prepare");
            rc = segment_prepare(c, *g.plan, g.n_break);
            if (rc) return rc;
        }
        c->counts.n_break = g.n_break;
        {
            HostClock hc(c, "host_segment_replay");
            g.seedsB.clear(); g.sensB.clear(); g.hasC = false;
            std::future<int> hypB;  // a shard that does not start the stream replays under both pasts, side by side
            if (sh.on && sh.prior_kept) hypB = c->pool->submit([&]() { return segment_replay(c, *g.plan, g.seedsB, true, &g.sensB, nullptr); });
            rc = segment_replay(c, *g.plan, g.seeds, false, nullptr, nullptr);
            g.seedsA = g.seeds;
            if (hypB.valid()) { const int rb = hypB.get(); if (!rc) rc = rb; }
        }
        if (rc) return rc;
        g.stage = 3;
        if (sh.on) { pack_seeds(c, g); return need_exchange(c); }
    }
    if (g.stage == 3) {
        if (sh.on) {
            rc = take_exchange(c, X_SEEDS);
            if (rc) return rc;
            // Seed nodes of all shards, resolved in rank order.  What a shard emits depends on the last node emitted before
            // it: none (A); one on an earlier chromosome than the shard's records, then only its existence matters (B);
            // or -- a discordant cluster in front of a chromosome's first kept record is processed by the shard before --
            // one on the shard's own first chromosome, then the shard replays again behind exactly that node (C) and
            // everybody exchanges once more.
            std::vector<Node> all;
            int redo = -1;
            for (int r = 0; r < W && redo < 0; ++r) {
                Unpacker u(c->xgot[r]);
                u.get<int32_t>();
                std::vector<int32_t> a, b2, se, c3;
                u.get_vec(a);
                const int32_t hasB = u.get<int32_t>();
                u.get_vec(b2); u.get_vec(se);
                const int32_t hasC = u.get<int32_t>();
                const int32_t cc = u.get<int32_t>(), cp = u.get<int32_t>(), cl = u.get<int32_t>();
                u.get_vec(c3);
                if (!u.ok) return fail(c, SQ_E_ARG, "sharded run: malformed seed payload");
                auto append = [&](const std::vector<int32_t>& v, size_t from) { for (size_t i = from; i + 2 < v.size(); i += 3) all.push_back(Node{v[i], v[i + 1], v[i + 2], 0, 0.0}); };
                if (g.first_chr[r] < 0) continue;                      // no kept record: nothing replayed, nothing emitted
                if (all.empty() || !hasB) { append(a, 0); continue; }  // nothing has been emitted before this shard
                const Node& last = all.back();
                bool exact_needed = last.chr >= g.first_chr[r];
                for (int32_t v : se) if (v == last.pos + last.len) exact_needed = true;  // the one comparison without a chromosome test (:623)
                if (!exact_needed) { append(b2, 0); continue; }
                if (hasC && cc == last.chr && cp == last.pos && cl == last.len && c3.size() >= 3) {
                    all.back() = Node{c3[0], c3[1], c3[2], 0, 0.0};  // the shard may have extended the node in front of it
                    append(c3, 3);
                    continue;
                }
                redo = r;
                if (r == me) {
                    HostClock hc(c, "host_segment_replay");
                    g.seedC = last; g.hasC = true;
                    rc = segment_replay(c, *g.plan, g.seedsC, false, nullptr, &g.seedC);
                    if (rc) return rc;
                }
            }
            if (redo >= 0) { pack_seeds(c, g); return need_exchange(c); }
            g.seeds.swap(all);
        }
        {
            HostClock hc(c, "host_tile_genome");
            std::vector<Node> seedcopy = g.seeds;
            rc = tile_genome(c, seedcopy, c->nodes);
        }
        if (rc) return rc;
        rc = dev_upload_nodes(c, c->nodes);
        if (rc) return rc;
        // the chimeric edges are host work on the (small) fragment list: do them on a second thread while this one
        // drives the depth and edge kernels over the concordant stream
        c->edges.clear();
        double chim_ms = 0;
        std::future<int> chim = c->pool->submit([&]() {
            const auto t0 = std::chrono::steady_clock::now();
            const int r2 = chimeric_edges(c, g.raw);
            chim_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            return r2;
        });
        std::vector<int32_t> unused;
        rc = dev_node_depth(c, c->nodes, g.n_break, g.sup, g.sl, g.tiny_boundary, g.amb_plus, g.amb_minus, unused);
        int rc_edges = rc ? rc : dev_concordant_edges(c, c->nodes, g.conc);
        const int rc_chim = chim.get();
        c->timer.add("host_chimeric_edges", chim_ms);
        if (rc_chim) return rc_chim;
        if (rc_edges) return rc_edges;
        g.stage = 4;
        if (sh.on) {  // exchange 4 (the data exchange): per-node depth accumulators of the own chromosomes + locally reduced edges
            const std::vector<Node>& N = c->nodes;
            const int nn = (int)N.size();
            int lo = 0, hi = 0;
            while (lo < nn && N[lo].chr < sh.first_ref) ++lo;
            hi = lo;
            while (hi < nn && N[hi].chr < sh.end_ref) ++hi;
            for (int i = 0; i < nn; ++i)
                if ((i < lo || i >= hi) && (g.sup[i] || g.sup[nn + i] || g.sl[i] || g.sl[nn + i] || g.amb_plus[i] || g.amb_minus[i]))
                    return fail(c, SQ_E_ARG, "internal: a record of this shard was counted for a node of another shard");
            Packer pk(c->xbuf);
            pk.put<int32_t>(X_GRAPH);
            pk.put<int32_t>(lo); pk.put<int32_t>(hi);
            pk.put<int64_t>((int64_t)g.sup[2 * nn]);  // |ReadsOther| of this shard
            pk.put<int32_t>(g.tiny_boundary ? 1 : 0);
            std::vector<int32_t> part;
            auto slice = [&](auto&& get) { part.clear(); for (int i = lo; i < hi; ++i) part.push_back((int32_t)get(i)); pk.put_vec(part); };
            slice([&](int i) { return g.sup[i]; }); slice([&](int i) { return g.sl[i]; }); slice([&](int i) { return g.sup[nn + i]; }); slice([&](int i) { return g.sl[nn + i]; });
            slice([&](int i) { return g.amb_plus[i]; }); slice([&](int i) { return g.amb_minus[i]; });
            std::vector<uint64_t> keys; std::vector<int32_t> ws;
            for (const Edge& e : g.conc) { keys.push_back(edge_pack(e)); ws.push_back(e.w); }
            pk.put_vec(keys); pk.put_vec(ws);
            pk.put<int64_t>(c->counts.n_raw_edges);
            return need_exchange(c);
        }
    }
    if (g.stage != 4 && g.stage != 5) return fail(c, SQ_E_ARG, "internal: bad build stage");
    std::vector<Node>& N = c->nodes;
    const int nn = (int)N.size();
    const bool resumed = g.stage == 5;  // back from the exchange of the ReadsOther lists (exact depth sweep of a sharded run)
    if (sh.on && !resumed) {
        rc = take_exchange(c, X_GRAPH);
        if (rc) return rc;
        g.sup.assign(2 * nn + 1, 0); g.sl.assign(2 * nn, 0); g.amb_plus.assign(nn, 0); g.amb_minus.assign(nn, 0);
        g.tiny_boundary = false; g.conc.clear();
        int64_t n_other = 0, n_raw = 0;
        for (int r = 0; r < W; ++r) {
            Unpacker u(c->xgot[r]);
            u.get<int32_t>();
            const int lo = u.get<int32_t>(), hi = u.get<int32_t>();
            n_other += u.get<int64_t>();
            if (u.get<int32_t>()) g.tiny_boundary = true;
            std::vector<int32_t> v[6];
            for (auto& x : v) u.get_vec(x);
            std::vector<uint64_t> keys; std::vector<int32_t> ws;
            u.get_vec(keys); u.get_vec(ws);
            n_raw += u.get<int64_t>();
            if (!u.ok || lo < 0 || hi > nn || lo > hi || keys.size() != ws.size()) return fail(c, SQ_E_ARG, "sharded run: malformed graph payload");
            for (auto& x : v) if ((int)x.size() != hi - lo) return fail(c, SQ_E_ARG, "sharded run: malformed graph payload");
            for (int i = lo; i < hi; ++i) {
                g.sup[i] += v[0][i - lo]; g.sl[i] += v[1][i - lo]; g.sup[nn + i] += v[2][i - lo]; g.sl[nn + i] += v[3][i - lo];
                g.amb_plus[i] += v[4][i - lo]; g.amb_minus[i] += v[5][i - lo];
            }
            for (size_t i = 0; i < keys.size(); ++i) {
                Edge e;
                e.a = (int32_t)(keys[i] >> 32); e.b = (int32_t)((keys[i] & 0xffffffffull) >> 2); e.ha = (keys[i] >> 1) & 1; e.hb = keys[i] & 1; e.w = ws[i]; e.gw = 0;
                g.conc.push_back(e);
            }
        }
        g.sup[2 * nn] = (int32_t)std::min<int64_t>(n_other, INT32_MAX);
        c->counts.n_raw_edges = n_raw;
    }
    // per-node Support / AvgDepth (SegmentGraph.cpp:766-826): discordant blocks on the host, stream blocks from the GPU
    std::vector<int32_t> dis_cnt(nn), dis_sum(nn);
    {
        HostClock hc(c, "host_depth_discordant");
        const std::vector<Blk>& disc = g.disc;
        size_t it = 0;
        for (int i = 0; i < nn; ++i) {
            int cnt = 0, sum = 0;
            for (; it != disc.size() && disc[it].refid == N[i].chr && disc[it].refpos < N[i].pos + N[i].len; ++it)
                if (disc[it].refpos >= N[i].pos && disc[it].refpos + disc[it].matchref <= N[i].pos + N[i].len) { ++cnt; sum += disc[it].matchref; }
            dis_cnt[i] = cnt; dis_sum[i] = sum;
        }
    }
    // ReadsOther (non-first blocks) is sorted with an unstable std::sort in the reference (SegmentGraph.cpp:781) and a
    // block of <= 3 bases right behind a node boundary is counted for whichever node the sweep cursor is on, which
    // depends on that sort's tie order.  Node depths only feed the coverage-ratio test of FilterEdges, so by default
    // the GPU reports canonical depths plus bounds over all tie orders and FilterEdges checks that no decision can
    // change inside the bounds; only then (or when SQUID_EXACT_DEPTH is set, as the stage-parity tests do) is the
    // reference's sort repeated on the host and the sweep walked exactly.  A sharded run cannot repeat that sort (its
    // tie order depends on the whole list) and fails loudly instead.
    const bool exact_mode = std::getenv("SQUID_EXACT_DEPTH") != nullptr && !sh.on;
    struct OtherR { int32_t chr, pos, len; };
    std::vector<OtherR> other_sorted;
    auto sort_other = [&]() { std::sort(other_sorted.begin(), other_sorted.end(), [](const OtherR& a, const OtherR& b) { return a.chr != b.chr ? a.chr < b.chr : a.pos < b.pos; }); };
    auto exact_sort = [&](bool always) -> int {  // always: fetch the list even when no block sits in a corner (the forced retry of the tests)
        std::vector<int32_t> ochr, opos, olen;
        bool has_tiny = false;
        int r2 = dev_gather_other(c, g.n_break, has_tiny, ochr, opos, olen, always);
        if (r2) return r2;
        other_sorted.resize(has_tiny || always ? ochr.size() : 0);
        for (size_t i = 0; i < other_sorted.size(); ++i) other_sorted[i] = OtherR{ochr[i], opos[i], olen[i]};
        sort_other();
        return SQ_OK;
    };
    // combine in the reference's order: discordant, ReadsMain, ReadsOther, then the division (only when ReadsOther is
    // non-empty, ledger B13).  `other` = per-node (count, sum) of ReadsOther, either canonical or from the exact sweep.
    auto set_depths = [&](const std::vector<int32_t>& ocnt, const std::vector<int32_t>& osum, bool bounds) {
        const int64_t n_other = g.sup[2 * nn];
        c->depth_bounds = bounds;
        for (int i = 0; i < nn; ++i) {
            N[i].support = dis_cnt[i] + g.sup[i];
            double d = dis_sum[i];
            d += (int32_t)g.sl[i];
            double lo = d, hi = d;
            if (n_other != 0) {
                N[i].support += ocnt[i];
                d += osum[i];
                lo = d; hi = d;
                if (bounds) { lo = d - g.amb_minus[i]; hi = d + g.amb_plus[i]; }
                d = 1.0 * d / N[i].len; lo = 1.0 * lo / N[i].len; hi = 1.0 * hi / N[i].len;
            }
            N[i].depth = d; N[i].depth_lo = lo; N[i].depth_hi = hi;
        }
    };
    auto exact_sweep = [&](std::vector<int32_t>& ocnt, std::vector<int32_t>& osum) {
        ocnt.assign(nn, 0); osum.assign(nn, 0);
        size_t it = 0;
        for (int i = 0; i < nn; ++i)
            for (; it != other_sorted.size(); ++it) {
                const OtherR& r = other_sorted[it];
                if (r.chr == N[i].chr && r.pos >= N[i].pos - 3 && r.pos + r.len <= N[i].pos + N[i].len + 3) { ocnt[i]++; osum[i] += r.len; }
                else if (r.pos >= N[i].pos + N[i].len || r.chr != N[i].chr) break;
            }
    };
    std::vector<int32_t> ocnt(nn), osum(nn);
    for (int i = 0; i < nn; ++i) { ocnt[i] = g.sup[nn + i]; osum[i] = (int32_t)g.sl[nn + i]; }
    c->depth_ambiguous = false;
    if (exact_mode && g.tiny_boundary) {
        HostClock hc(c, "host_depth_exact_sweep");
        rc = exact_sort(false);
        if (rc) return rc;
        if (!other_sorted.empty()) exact_sweep(ocnt, osum);
        set_depths(ocnt, osum, false);
    } else set_depths(ocnt, osum, g.tiny_boundary);
    static const bool host_filters = std::getenv("SQUID_HOST_FILTERS") != nullptr;
    if (!resumed) {
        if (c->keep_stages) c->snap[1].take(c->nodes, c->edges, nullptr);
        {
            HostClock hc(c, "host_edge_reduce");
            g.raw.insert(g.raw.end(), g.conc.begin(), g.conc.end());
            reduce_edges(g.raw, c->edges, c->pool ? std::min(c->pool->size() + 1, 32) : 1);
        }
        if (c->keep_stages) c->snap[2].take(c->nodes, c->edges, nullptr);
    }
    // K6 / K7 run on the device (sq_graph_kernels.inc); SQUID_HOST_FILTERS=1 takes the host restatements of sq_graph.cpp instead
    // (kept as a cross-check: tests compare the two stage by stage)
    {
        HostClock hc(c, host_filters ? "host_filters" : "wall_filters");
        if (!resumed) {
            if (host_filters) filter_by_weight(c); else if ((rc = dev_filter_by_weight(c))) return rc;
            if (c->keep_stages) c->snap[3].take(c->nodes, c->edges, nullptr);
            if (host_filters) filter_by_interleaving(c, g.keep); else if ((rc = dev_filter_by_interleaving(c, g.keep))) return rc;
            g.before = c->edges;
            if (host_filters) filter_edges(c, g.keep); else if ((rc = dev_filter_edges(c, g.keep))) return rc;
            if (std::getenv("SQUID_FORCE_DEPTH_RETRY")) c->depth_ambiguous = true;  // (tests: take the exact sweep whatever the bounds say)
        }
        if (c->depth_ambiguous || resumed) {
            // some coverage-ratio decision depends on the tie order: repeat the reference's sort and sweep, then redo the
            // filter with the exact depths.  A sharded run needs every shard's ReadsOther for that (the tie order of the unstable
            // sort depends on the whole list): one more exchange -- the lists, in rank order, are the unsharded stream order
            HostClock hc2(c, "host_depth_exact_retry");
            if (sh.on && !resumed) {
                std::vector<int32_t> ochr, opos, olen;
                bool has_tiny = false;
                rc = dev_gather_other(c, g.n_break, has_tiny, ochr, opos, olen, true);
                if (rc) return rc;
                Packer pk(c->xbuf);
                pk.put<int32_t>(X_OTHER);
                pk.put_vec(ochr); pk.put_vec(opos); pk.put_vec(olen);
                g.stage = 5;
                return need_exchange(c);
            }
            if (sh.on) {
                rc = take_exchange(c, X_OTHER);
                if (rc) return rc;
                other_sorted.clear();
                for (int r = 0; r < W; ++r) {
                    Unpacker u(c->xgot[r]);
                    u.get<int32_t>();
                    std::vector<int32_t> a, b, d;
                    u.get_vec(a); u.get_vec(b); u.get_vec(d);
                    if (!u.ok || a.size() != b.size() || a.size() != d.size()) return fail(c, SQ_E_ARG, "sharded run: malformed ReadsOther payload");
                    for (size_t i = 0; i < a.size(); ++i) other_sorted.push_back(OtherR{a[i], b[i], d[i]});
                }
                sort_other();
            } else {
                rc = exact_sort(true);  // (with nothing fetched the sweep below would zero every ReadsOther contribution)
                if (rc) return rc;
            }
            exact_sweep(ocnt, osum);
            set_depths(ocnt, osum, false);
            c->depth_ambiguous = false;
            c->edges = g.before;
            if (host_filters) filter_edges(c, g.keep); else if ((rc = dev_filter_edges(c, g.keep))) return rc;
        }
        if (c->keep_stages) c->snap[4].take(c->nodes, c->edges, nullptr);
    }
    return finish_graph(c, host_filters);
}

struct SvBuild {
    int stage = 0;
    BPMap bpmap;
    // the breakpoint pairs of every edge, once (edge i: entries [eoff[i], eoff[i + 1]); round 4 looked every edge up three times in the map)
    std::vector<int32_t> eoff;
    std::vector<std::pair<std::pair<int, int>, std::pair<int, int>>> ebp;
    std::vector<uint8_t> eexact;
    std::vector<std::pair<int, int>> BPs;
    std::vector<int32_t> diff;  // this shard's difference array
    int cur_prev = 0;
    BpBoundary bb;
};

static int call_sv(sq_ctx* c) {
    HostClock wall(c, "wall_call_sv");
    if (!c->ordered) return fail(c, SQ_E_ARG, "sq_call_sv before sq_order");
    const std::vector<Node>& N = c->nodes;
    std::vector<Edge>& E = c->edges;
    Shard& sh = c->shard;
    if (!c->svb) c->svb = std::make_shared<SvBuild>();
    SvBuild& v = *c->svb;
    BPMap& bpmap = v.bpmap;
    // breakpoint list of every edge (SegmentGraph.cpp:3091-3109): the pairs ExactBreakpoint found, else the node ends the edge touches
    const size_t m = E.size();
    auto par = [&](size_t n, const std::function<void(size_t, size_t)>& f) {  // [lo, hi) pieces on the context's host threads
        const int np = (c->pool && n > 20000) ? 4 * (c->pool->size() + 1) : 1;
        if (np <= 1) { f(0, n); return; }
        c->pool->parallel_for(np, 1 << 20, [&](int k) { f(n * (size_t)k / (size_t)np, n * ((size_t)k + 1) / (size_t)np); });
    };
    auto build_edge_table = [&]() {
        v.eoff.assign(m + 1, 0); v.eexact.assign(m, 0);
        std::vector<const std::vector<std::pair<int, int>>*> hit(m, nullptr);
        par(m, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i) {
                BPMap::const_iterator it = bpmap.find(edge_pack(E[i]));
                if (it != bpmap.end() && !it->second.empty()) { hit[i] = &it->second; v.eexact[i] = 1; v.eoff[i + 1] = (int32_t)it->second.size(); }
                else v.eoff[i + 1] = 1;
            }
        });
        for (size_t i = 0; i < m; ++i) v.eoff[i + 1] += v.eoff[i];
        v.ebp.resize((size_t)v.eoff[m]);
        par(m, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i) {
                const Edge& e = E[i];
                size_t o = (size_t)v.eoff[i];
                if (hit[i]) for (const auto& p : *hit[i]) v.ebp[o++] = {{N[e.a].chr, p.first}, {N[e.b].chr, p.second}};
                else v.ebp[o] = {{N[e.a].chr, e.ha ? N[e.a].pos : N[e.a].pos + N[e.a].len}, {N[e.b].chr, e.hb ? N[e.b].pos : N[e.b].pos + N[e.b].len}};
            }
        });
    };
    std::vector<std::pair<int, int>>& BPs = v.BPs;
    std::vector<int32_t> cov;
    int rc;
    auto pack_bpsup = [&]() {
        Packer pk(c->xbuf);
        pk.put<int32_t>(X_BPSUP);
        pk.put<int32_t>(v.cur_prev); pk.put<int32_t>(v.bb.cur_end); pk.put<int32_t>(v.bb.has_p3 ? 1 : 0); pk.put<int32_t>(v.bb.has_event ? 1 : 0);
        pk.put<int64_t>(v.bb.n_p3); pk.put<int64_t>(v.bb.absorb);
        pk.put_vec(v.diff);
    };
    if (v.stage == 0) {
        if (c->bp_future.valid()) {
            rc = c->bp_future.get();
            c->timer.add("host_exact_breakpoints", c->bp_early_ms);
            if (rc) return rc;
            bpmap.swap(*c->bp_early);
            c->bp_early.reset();
        } else {
            HostClock hc(c, "host_exact_breakpoints");
            rc = exact_breakpoints(c, bpmap);
            if (rc) return rc;
        }
        build_edge_table();
        BPs.resize(2 * v.ebp.size());
        par(v.ebp.size(), [&](size_t lo, size_t hi) { for (size_t k = lo; k < hi; ++k) { BPs[2 * k] = v.ebp[k].first; BPs[2 * k + 1] = v.ebp[k].second; } });
        // (equal elements are indistinguishable pairs: any sort gives the reference's sorted list; the threaded introsort of sq_parsort.h)
        std_sort_parallel(BPs.begin(), BPs.end(), std::less<std::pair<int, int>>(), c->pool ? std::min(c->pool->size() + 1, 32) : 1, true);
        static const bool bp_host = getenv("SQUID_BP_HOST") != nullptr;  // debug cross-check of k_bp_walk
        if (c->bwa) {  // (--bwa: the records and their names are on the host)
            rc = bwa_breakpoint_support(c, BPs, cov);
            if (rc) return rc;
        } else if (!sh.on) {
            rc = bp_host ? dev_breakpoint_support_exact(c, BPs, cov) : dev_breakpoint_support(c, BPs, cov);
            if (rc) return rc;
        } else {
            // the cursor (SegmentGraph.cpp:3157) is carried from shard to shard: start from the usual state -- every
            // breakpoint on an earlier chromosome is behind it -- and let the exchange verify that guess
            v.cur_prev = (int)(std::lower_bound(BPs.begin(), BPs.end(), std::make_pair(sh.first_ref, INT32_MIN)) - BPs.begin());
            rc = dev_breakpoint_support(c, BPs, v.diff, v.cur_prev, &v.bb, true);
            if (rc) return rc;
            pack_bpsup();
            v.stage = 1;
            return need_exchange(c);
        }
    } else {
        rc = take_exchange(c, X_BPSUP);
        if (rc) return rc;
        const int W = c->P.world_size, me = c->P.rank;
        const size_t nb = BPs.size();
        std::vector<int64_t> total(nb + 1, 0);
        int truth = 0, bad = -1;
        for (int r = 0; r < W; ++r) {
            Unpacker u(c->xgot[r]);
            u.get<int32_t>();
            const int assumed = u.get<int32_t>(), end = u.get<int32_t>(), has = u.get<int32_t>(), has_event = u.get<int32_t>();
            const int64_t n_p3 = u.get<int64_t>(), absorb = u.get<int64_t>();
            std::vector<int32_t> d;
            u.get_vec(d);
            if (!u.ok || d.size() != nb + 1) return fail(c, SQ_E_ARG, "sharded run: malformed breakpoint payload");
            if (!has) continue;  // no pass-3 record: the cursor passes through untouched
            // a cursor that arrives `lag` entries behind the assumed position catches up one entry per pass-3 record;
            // every breakpoint it still has to pass lies on an earlier chromosome, so nothing counted here changes as
            // long as it has caught up before the first record that moves it further
            const int64_t lag = (int64_t)assumed - truth;
            if (lag < 0 || lag > absorb) { bad = r; break; }
            truth = has_event ? end : (int)std::min<int64_t>(assumed, (int64_t)truth + n_p3);
            for (size_t i = 0; i <= nb; ++i) total[i] += d[i];
        }
        if (bad >= 0) {
            // rank `bad` started from a wrong cursor (an earlier shard ended while the cursor was still catching up): it
            // recounts from the real one, everybody exchanges again
            if (bad == me) {
                v.cur_prev = truth;
                rc = dev_breakpoint_support(c, BPs, v.diff, v.cur_prev, &v.bb, true);
                if (rc) return rc;
            }
            pack_bpsup();
            return need_exchange(c);
        }
        cov.assign(nb, 0);
        int64_t run = 0;
        for (size_t i = 0; i < nb; ++i) { run += total[i]; cov[i] = (int32_t)run; }
    }
    // per-edge table in key order (parity tests) -- before the weight sort
    const size_t nbp = v.ebp.size();
    std::vector<int32_t> sup1(nbp), sup2(nbp);
    par(nbp, [&](size_t lo, size_t hi) {
        for (size_t k = lo; k < hi; ++k) {
            sup1[k] = cov[(size_t)(std::lower_bound(BPs.begin(), BPs.end(), v.ebp[k].first) - BPs.begin())];
            sup2[k] = cov[(size_t)(std::lower_bound(BPs.begin(), BPs.end(), v.ebp[k].second) - BPs.begin())];
        }
    });
    c->bp_off.assign(v.eoff.begin(), v.eoff.end());
    c->bp1.resize(nbp); c->bp2.resize(nbp);
    c->bsup1 = sup1; c->bsup2 = sup2;
    par(m, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i)
            for (int32_t k = v.eoff[i]; k < v.eoff[i + 1]; ++k) { c->bp1[(size_t)k] = v.eexact[i] ? v.ebp[(size_t)k].first.second : -1; c->bp2[(size_t)k] = v.eexact[i] ? v.ebp[(size_t)k].second.second : -1; }
    });
    multiply_discordant(c, true);
    // WriteBEDPE (src/WriteIO.cpp:45-124): unstable sort by weight on the key-sorted edge list (ledger B8).  What is sorted is the index of
    // every edge with the reference's comparison on the weights: libstdc++'s introsort takes the same decisions, hence the same order
    HostClock hc(c, "host_select_sv");
    std::vector<int32_t> W(m);
    for (size_t i = 0; i < m; ++i) W[i] = (int32_t)i;
    std::sort(W.begin(), W.end(), [&](int32_t a, int32_t b) { return E[(size_t)a].w > E[(size_t)b].w; });
    // rank / sign of every node in its component order
    std::vector<int> comp(N.size(), -1), rank(N.size(), -1), sign(N.size(), 1);
    par(c->ord_off.size() - 1, [&](size_t lo, size_t hi) {
        for (size_t k = lo; k < hi; ++k)
            for (int j = c->ord_off[k]; j < c->ord_off[k + 1]; ++j) {
                int nd = std::abs(c->ord_nodes[j]) - 1;
                comp[nd] = (int)k; rank[nd] = j - c->ord_off[k]; sign[nd] = c->ord_nodes[j] < 0 ? -1 : 1;
            }
    });
    for (auto& col : c->sv_cols) col.clear();
    c->sv_s1.clear(); c->sv_s2.clear();
    for (const int32_t wi : W) {
        const Edge& e = E[(size_t)wi];
        const Node &a = N[e.a], &b = N[e.b];
        bool conc = a.chr == b.chr && e.ha == 0 && e.hb == 1 && (b.pos - a.pos - a.len <= c->P.concord_dist_pos || e.b - e.a <= c->P.concord_dist_idx);
        if (conc) continue;
        bool ok = false;
        if (comp[e.a] == comp[e.b] && rank[e.a] < rank[e.b] && (bool)e.ha == (sign[e.a] < 0) && (bool)e.hb == (sign[e.b] > 0)) ok = true;
        else if (comp[e.a] == comp[e.b] && rank[e.a] > rank[e.b] && (bool)e.hb == (sign[e.b] < 0) && (bool)e.ha == (sign[e.a] > 0)) ok = true;
        if (!ok) continue;
        for (int32_t k = v.eoff[(size_t)wi]; k < v.eoff[(size_t)wi + 1]; ++k) {
            int b1 = v.ebp[(size_t)k].first.second, b2 = v.ebp[(size_t)k].second.second;
            c->sv_cols[0].push_back(a.chr); c->sv_cols[1].push_back(e.ha ? b1 : a.pos); c->sv_cols[2].push_back(e.ha ? a.pos + a.len : b1);
            c->sv_cols[3].push_back(b.chr); c->sv_cols[4].push_back(e.hb ? b2 : b.pos); c->sv_cols[5].push_back(e.hb ? b.pos + b.len : b2);
            c->sv_cols[6].push_back(e.w); c->sv_cols[7].push_back(sup1[(size_t)k]); c->sv_cols[8].push_back(sup2[(size_t)k]);
            c->sv_s1.push_back(e.ha); c->sv_s2.push_back(e.hb);
        }
    }
    c->svb.reset();
    return SQ_OK;
}

}  // namespace sq

using namespace sq;

// No exception leaves the library: the host stages allocate (std::bad_alloc from the vectors, the raw scratch blocks and the bodies of
// HostPool::parallel_for, which hands a helper's exception to its caller) -- at the C boundary that becomes SQ_E_CAPACITY with a message.
template <class F> static int abi_guard(sq_ctx* c, const char* what, F f) {
    try { return f(); }
    catch (const std::bad_alloc&) { return c ? sq::fail(c, SQ_E_CAPACITY, std::string(what) + ": out of host memory") : (int)SQ_E_CAPACITY; }
    catch (const std::exception& e) { return c ? sq::fail(c, SQ_E_CAPACITY, std::string(what) + ": " + e.what()) : (int)SQ_E_CAPACITY; }
}

extern "C" {

void sq_default_params(sq_params* p) {
    std::memset(p, 0, sizeof *p);
    p->abi_version = SQ_ABI_VERSION;
    p->device = 0;
    p->phred_type = 1; p->max_lowphred_len = 10; p->min_phred = 4; p->min_mapqual = 1;
    p->concord_dist_pos = 50000; p->concord_dist_idx = 20; p->min_edge_weight = 5; p->discordant_ratio = 8; p->max_allowed_degree = 5;
    p->rank = 0; p->world_size = 1;
}

const char* sq_strerror(int code) {
    switch (code) {
        case SQ_OK: return "ok";
        case SQ_E_ARG: return "bad argument or call order";
        case SQ_E_NODEVICE: return "no usable HIP device";
        case SQ_E_HIP: return "HIP runtime error";
        case SQ_E_IO: return "cannot read BAM";
        case SQ_E_UNSORTED: return "input not coordinate sorted";
        case SQ_E_ASSERT: return "input trips a reference assert";
        case SQ_E_CAPACITY: return "internal capacity exceeded";
        case SQ_E_EMPTYCHIM: return "chimeric input has no usable record";
        default: return "unknown error";
    }
}
const char* sq_last_error(sq_ctx* c) { return c ? c->err.c_str() : ""; }

int sq_create(const sq_params* p, sq_ctx** out) {
    if (!p || !out || p->abi_version != SQ_ABI_VERSION) return SQ_E_ARG;
    sq_ctx* c = new sq_ctx();
    c->P = *p;
    c->pool.reset(new HostPool(host_workers(p->world_size)));
    int rc = dev_create(c);
    if (rc) { std::fprintf(stderr, "libsquid_hip: %s\n", c->err.c_str()); dev_destroy(c); delete c; return rc; }
    *out = c;
    return SQ_OK;
}
void sq_destroy(sq_ctx* c) {
    if (!c) return;
    if (c->bp_future.valid()) (void)c->bp_future.get();
    exchange_release(c);
    dev_destroy(c);
    delete c;
}
int sq_set_references(sq_ctx* c, int32_t n_ref, const int32_t* ref_len) {
    if (!c || n_ref < 0 || (n_ref && !ref_len)) return SQ_E_ARG;
    c->ref_len.assign(ref_len, ref_len + n_ref);
    drop_early_clusters(c);
    return SQ_OK;
}
// frags0 = frags, side by side (millions of fragments on a dense sample; sq_reset copies the other way)
static void copy_frags(sq_ctx* c, const std::vector<Frag>& src, std::vector<Frag>& dst) {
    dst.resize(src.size());
    HostPool* pool = c->pool.get();
    const int64_t n = (int64_t)src.size();
    const int pieces = (int)std::min<int64_t>(std::max<int64_t>(1, n / 4096), pool ? 4 * (pool->size() + 1) : 1);
    auto piece = [&](int k) { for (int64_t i = n * k / pieces; i < n * (k + 1) / pieces; ++i) dst[(size_t)i] = src[(size_t)i]; };
    if (pieces <= 1 || !pool) { for (int k = 0; k < pieces; ++k) piece(k); }
    else pool->parallel_for(pieces, 1 << 20, piece);
}
static int sq_ingest_chimeric_impl(sq_ctx* c, const sq_aln_batch* b) {
    if (!c || !b) return SQ_E_ARG;
    drop_early_clusters(c);
    int rc = build_fragments(c, b);
    if (rc) return rc;
    copy_frags(c, c->frags, c->frags0);
    return dev_upload_chim_names(c);
}
int sq_ingest_chimeric(sq_ctx* c, const sq_aln_batch* b) { return abi_guard(c, "sq_ingest_chimeric", [&]() { return sq_ingest_chimeric_impl(c, b); }); }
int sq_chim_contains(sq_ctx* c, const char* name, size_t len) {
    if (!c) return SQ_E_ARG;
    return std::binary_search(c->chim_names.begin(), c->chim_names.end(), std::string(name, len)) ? 1 : 0;
}
// does this shard own records of RefID `id`?  (the unplaced records at the end of a sorted BAM go to the last rank)
static inline bool shard_owns(const sq_ctx* c, int32_t id) {
    const Shard& sh = c->shard;
    if (!sh.on) return true;
    if (id < 0) return c->P.rank == c->P.world_size - 1;
    return id >= sh.first_ref && id < sh.end_ref;
}
static int sq_ingest_concordant_impl(sq_ctx* c, const sq_aln_batch* b) {
    if (!c || !b) return SQ_E_ARG;
    if (c->bwa) return fail(c, SQ_E_ARG, "this context holds a --bwa batch (sq_ingest_bwa_file): sq_clear_records before a STAR-mode ingest -- the mode is per context");
    if (!c->shard.on) return dev_append_records(c, b);
    // sharded: keep the runs of records this rank owns (one run per batch on a sorted stream)
    int64_t i = 0;
    while (i < b->n_rec) {
        while (i < b->n_rec && !shard_owns(c, b->refid[i])) ++i;
        int64_t j = i;
        while (j < b->n_rec && shard_owns(c, b->refid[j])) ++j;
        if (j > i) {
            sq_aln_batch v = *b;
            v.n_rec = j - i; v.n_blk = b->blk_off[j] - b->blk_off[i];
            v.refid += i; v.pos += i; v.mate_refid += i; v.mate_pos += i; v.end_pos += i; v.flag += i; v.mapq += i; v.aux += i; v.totlen += i; v.blk_off += i;
            const uint32_t o = b->blk_off[i] - b->blk_off[0];
            v.b_refpos += o; v.b_matchref += o; v.b_readpos += o; v.b_matchread += o;
            int rc = dev_append_records(c, &v);
            if (rc) return rc;
        }
        i = j;
    }
    return SQ_OK;
}
int sq_ingest_concordant(sq_ctx* c, const sq_aln_batch* b) { return abi_guard(c, "sq_ingest_concordant", [&]() { return sq_ingest_concordant_impl(c, b); }); }
static int ingest_raw(sq_ctx* c, const uint8_t* bam, size_t nbytes, const unsigned long long* rec_off, int64_t n_rec) {
    if (!c->shard.on) return dev_parse_append(c, bam, nbytes, rec_off, n_rec);
    // sharded: RefID sits 4 bytes into a record; upload only the byte range of the runs this rank owns
    auto refid_at = [&](int64_t i) { int32_t v; std::memcpy(&v, bam + rec_off[i] + 4, 4); return v; };
    int64_t i = 0;
    std::vector<unsigned long long> off;
    while (i < n_rec) {
        while (i < n_rec && (rec_off[i] + 8 > nbytes || !shard_owns(c, refid_at(i)))) ++i;
        int64_t j = i;
        while (j < n_rec && rec_off[j] + 8 <= nbytes && shard_owns(c, refid_at(j))) ++j;
        if (j > i) {
            const unsigned long long lo = rec_off[i], hi = j < n_rec ? rec_off[j] : (unsigned long long)nbytes;
            off.resize((size_t)(j - i));
            for (int64_t k = i; k < j; ++k) off[(size_t)(k - i)] = rec_off[k] - lo;
            int rc = dev_parse_append(c, bam + lo, (size_t)(hi - lo), off.data(), j - i);
            if (rc) return rc;
        }
        i = j;
    }
    return SQ_OK;
}
int sq_ingest_concordant_bam(sq_ctx* c, const uint8_t* bam, size_t nbytes, const uint64_t* rec_off, int64_t n_rec) {
    if (!c || (n_rec && (!bam || !rec_off))) return SQ_E_ARG;
    if (c->bwa) return fail(c, SQ_E_ARG, "this context holds a --bwa batch (sq_ingest_bwa_file): sq_clear_records before a STAR-mode ingest -- the mode is per context");
    int rc = ingest_raw(c, bam, nbytes, (const unsigned long long*)rec_off, n_rec);
    dev_flush_timers(c);
    return rc;
}
int sq_read_header(const char* path, int32_t* n_ref, int32_t* ref_len, char* names, size_t names_cap) {
    std::vector<std::string> nm;
    std::vector<int32_t> ln;
    std::string err;
    int rc = read_bam_header(path, nm, ln, err);
    if (rc) return rc;
    if (n_ref) {
        int cap = *n_ref;
        *n_ref = (int32_t)nm.size();
        if (ref_len) for (int i = 0; i < (int)nm.size() && i < cap; ++i) ref_len[i] = ln[i];
    }
    if (names && names_cap) {
        size_t o = 0;
        for (const std::string& s : nm) {
            if (o + s.size() + 1 >= names_cap) break;
            std::memcpy(names + o, s.data(), s.size());
            o += s.size();
            names[o++] = '\n';
        }
        names[o < names_cap ? o : names_cap - 1] = 0;
    }
    return SQ_OK;
}
// the whole chimeric BAM as one batch -> fragments (the batch is consumed where the reader hands it over: no copy of it)
// `early` (sq_ingest_files): as soon as the records are decoded, the device gets the table of all their usable QNAMEs and the promise
// the record parse of the concordant BAM waits for is kept -- the pairing goes on meanwhile
static int chimeric_file_to_fragments(sq_ctx* c, const char* path, int nt, std::string& err, bool early = false, const HostBatch* decoded = nullptr) {
    const auto t_chim0 = std::chrono::steady_clock::now();
    static const bool chim_prof = std::getenv("SQUID_CHIM_PROF") != nullptr;
    auto lap = [&](const char* what) { if (chim_prof) std::fprintf(stderr, "chimeric file: %-28s at %8.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_chim0).count()); };
    drop_early_clusters(c);
    ParseOpts o{c->P.phred_type, c->P.min_phred, c->P.max_lowphred_len, true, nullptr};
    bool got = false, promised = false;
    auto promise = [&](int rc) { if (early && !promised) { promised = true; c->chim_names_promise.set_value(rc); } };
    const std::function<int(const HostBatch&)> take = [&](const HostBatch& hb) {
        sq_aln_batch b;
        hb.view(&b, true);
        got = true;
        if (early && decoded) promise(SQ_OK);  // (chimeric_records_through_the_device has made the table from the names on the device: dev_chim_begin_captured)
        else if (early) {
            // one entry per usable record (mapped, not a duplicate: ReadRec.cpp:344), the name with a trailing /1 or /2 cut off
            // (ReadRec.cpp:62-66), plus the empty name the reference's set always holds (SegmentGraph.cpp:196-201, ledger B9)
            std::vector<uint32_t> off, len;
            off.reserve((size_t)b.n_rec + 1); len.reserve((size_t)b.n_rec + 1);
            for (int64_t i = 0; i < b.n_rec; ++i) {
                if ((b.flag[i] & 0x4) || (b.flag[i] & 0x400)) continue;
                const char* nm = b.name_blob + b.name_off[i];
                size_t L = b.name_off[i + 1] - b.name_off[i];
                if (L >= 2 && nm[L - 2] == '/' && (nm[L - 1] == '1' || nm[L - 1] == '2')) L -= 2;
                off.push_back(b.name_off[i]); len.push_back((uint32_t)L);
            }
            if (!off.empty()) { off.push_back(0); len.push_back(0); }
            lap("records decoded");
            const int rt = off.empty() ? SQ_OK : dev_chim_begin(c, b.name_blob, (size_t)b.name_off[b.n_rec], off.data(), len.data(), off.size());
            promise(rt);
            lap("name table handed to the device");
            if (rt) return rt;
        }
        const int rf = build_fragments(c, &b);
        lap("fragments built");
        return rf;
    };
    // `decoded`: the records came through the GPU reader (chimeric_records_through_the_device); otherwise the host decoder reads the file
    int rc = decoded ? (decoded->refid.empty() ? SQ_OK : take(*decoded)) : parse_bam_file(path, o, (size_t)1 << 40, nt, err, take);
    lap("reader returned");
    if (!rc && !got) rc = fail(c, SQ_E_EMPTYCHIM, "chimeric BAM holds no record");
    promise(rc ? rc : SQ_OK);  // (whatever happened: nobody waits for ever)
    if (rc) return rc;
    if (early && !c->ref_len.empty()) {  // this thread has nothing else to do, the concordant file is still being read
        // (the copy sq_reset restores the fragments from is made next to the cluster table: both only read c->frags)
        std::future<void> copied = std::async(std::launch::async, [c]() { copy_frags(c, c->frags, c->frags0); });
        c->plan_early.reset(); c->disc_early.clear();
        c->clusters_early_ms = segment_clusters(c, c->plan_early, c->disc_early);
        lap("cluster table");
        copied.get();
        lap("fragments copied");
        return SQ_OK;
    }
    copy_frags(c, c->frags, c->frags0);
    lap("fragments copied");
    return SQ_OK;
}
// BuildChimericSBamRecord on a context that has no device side (the junction-sequence utility): c->frags0
}  // extern "C"
int sq::chimeric_fragments_host(sq_ctx* c, const char* path, int threads) { return chimeric_file_to_fragments(c, path, std::max(1, threads), c->err); }
extern "C" {
static int sq_ingest_chimeric_file_impl(sq_ctx* c, const char* path) {
    if (!c || !path) return SQ_E_ARG;
    // (inflate and decode on a few threads: a dense sample has millions of chimeric records; the result does not depend on the count)
    const int nt = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency() / 4));
    int rc = chimeric_file_to_fragments(c, path, nt, c->err);
    return rc ? rc : dev_upload_chim_names(c);
}
int sq_ingest_chimeric_file(sq_ctx* c, const char* path) { return abi_guard(c, "sq_ingest_chimeric_file", [&]() { return sq_ingest_chimeric_file_impl(c, path); }); }
// The chimeric BAM through the GPU reader (K-1 + K0 with sq_ctx::capture_names): BGZF inflate, record boundaries and the record parse on
// the device, the decoded records and their QNAMEs copied back as one batch -- the batch the host decoder (parse_bam_file, keep_names) makes
// of the same file, field for field.  On the dense config the host decoder is 0.6 s of sixteen threads = 9 of the 16 CPU-seconds per second
// a box gives the process; the device does it in the time it takes to get the file there.  Needs an empty record
// store (the records pass through it).  Returns 2 when the route is not taken or gave up: the caller runs the host decoder, whose
// error messages are then the ones the user sees.
static int chimeric_records_through_the_device(sq_ctx* c, const char* path, int n_threads, HostBatch& hb, bool name_table = true) {
    if (!c->dev || c->counts.n_concordant != 0) return 2;
    struct Mode { sq_ctx* c; Mode(sq_ctx* c) : c(c) { c->capture_names = true; } ~Mode() { c->capture_names = false; dev_clear_records(c); c->counts.n_concordant = 0; c->counts.n_blocks = 0; c->ingest_total_bytes = 0; c->ingest_seen_bytes = 0; } } mode(c);
    std::string err;
    c->ingest_total_bytes = 0; c->ingest_seen_bytes = 0;
    std::string saved_err = c->err;
    int rc = scan_bam_file(path, n_threads, err, [&](const uint8_t* bam, size_t nbytes, const unsigned long long* off, int64_t n) { c->ingest_seen_bytes += nbytes; return dev_parse_append(c, bam, nbytes, off, n); },
                           [&](size_t total) { c->ingest_total_bytes = total; }, nullptr,
                           [&](const uint8_t* file, std::vector<BgzfRange>& blocks, size_t b0, size_t b1, size_t begin, bool synced, int nref, const IndexMore& more, size_t file_bytes, GpuFileSrc* src) {
                               return dev_ingest_bgzf(c, file, blocks, b0, b1, begin, synced, nref, more, file_bytes, src); }, false, true, true);
    static const bool prof = std::getenv("SQUID_CHIM_PROF") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (prof) std::fprintf(stderr, "chimeric file on the device: %-24s at %8.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); };
    lap("reader returned (since then)");
    if (rc == SQ_OK && name_table) rc = dev_chim_begin_captured(c);
    lap("name table");
    if (rc == SQ_OK) rc = dev_download_records(c, hb);
    if (rc == SQ_OK) rc = dev_download_names(c, hb);
    lap("records and names copied back");
    dev_flush_timers(c);  // (its launches are not the concordant ingest's)
    if (rc != SQ_OK) { c->err = saved_err; hb.clear(); return 2; }
    return SQ_OK;
}
// both input files in one call: the chimeric BAM (1-2 % of the records, decoded on the host: 17 ms at C3) is read on a helper
// thread while the GPU reader starts on the concordant BAM; the record parse -- the first consumer of the chimeric QNAME set --
// waits for it (chim_join).  Same result as sq_ingest_chimeric_file followed by sq_ingest_concordant_file.
static int sq_ingest_files_impl(sq_ctx* c, const char* chim_path, const char* bam_path, int32_t n_threads) {
    if (!c || !chim_path || !bam_path) return SQ_E_ARG;
    if (c->bwa) return fail(c, SQ_E_ARG, "this context holds a --bwa batch (sq_ingest_bwa_file): sq_clear_records before a STAR-mode ingest -- the mode is per context");
    if (std::getenv("SQUID_HOST_PARSE") || std::getenv("SQUID_SERIAL_LOAD")) {  // (the host parser consults the name set record by record)
        const int rc = sq_ingest_chimeric_file(c, chim_path);
        return rc ? rc : sq_ingest_concordant_file(c, bam_path, n_threads);
    }
    const std::string chim = chim_path;
    c->chim_err.clear();
    c->chim_names_promise = std::promise<int>();
    c->chim_names_future = c->chim_names_promise.get_future();
    // a chimeric BAM of some size goes through the GPU reader first (SQUID_CHIM_GPU=1 / =0 forces / forbids it): the helper thread then
    // starts from decoded records
    std::shared_ptr<HostBatch> decoded;
    {
        struct stat st;
        const char* env = std::getenv("SQUID_CHIM_GPU");
        const bool want = env ? std::atoi(env) != 0 : (::stat(chim_path, &st) == 0 && (size_t)st.st_size >= ((size_t)128 << 20));
        if (want) {
            // (the batch is the context's between calls, like the scratch of a whole-file read on the host: 0.5 GB of pages on the dense
            // config that the next read of a chimeric file finds in place; sq_release_reader_buffers gives them back)
            if (!c->chim_decoded || c->chim_decoded.use_count() > 1) c->chim_decoded = std::make_shared<HostBatch>();
            decoded = c->chim_decoded;
            if (chimeric_records_through_the_device(c, chim_path, n_threads, *decoded) != SQ_OK) decoded.reset();
        }
        c->counts.chimeric_through_gpu_reader = decoded ? 1 : 0;
    }
    c->chim_pairing_running = decoded != nullptr;
    c->chim_future = std::async(std::launch::async, [c, chim, n_threads, decoded]() {
        tl_err_sink = &c->chim_err;
        struct Unsink { sq_ctx* c; ~Unsink() { tl_err_sink = nullptr; c->chim_pairing_running = false; } } unsink{c};
        // (decode threads: the rank's share of the CPUs the process may really use -- eight ranks on one host decode the same file side by side)
        return chimeric_file_to_fragments(c, chim.c_str(), std::max(1, std::min({(int)n_threads, 16, usable_cpus() / std::max(1, c->P.world_size)})), c->chim_err, true, decoded.get());  // (the host decoder of the chimeric BAM: 16 threads 96 ms of decode per 9.6 M records, 64 threads 254 ms)
    });
    const int rc_conc = sq_ingest_concordant_file(c, bam_path, n_threads);
    const int rc_chim = chim_join(c);  // (a concordant file without records never reached the parse)
    return rc_chim ? rc_chim : rc_conc;
}
int sq_ingest_files(sq_ctx* c, const char* chim_path, const char* bam_path, int32_t n_threads) { return abi_guard(c, "sq_ingest_files", [&]() { return sq_ingest_files_impl(c, chim_path, bam_path, n_threads); }); }
int sq_set_source(sq_ctx* c, const char* path) {
    if (!c || !path) return SQ_E_ARG;
    struct stat st;
    if (::stat(path, &st) != 0) return fail(c, SQ_E_IO, std::string("cannot open bamfile ") + path);
    c->source_size = (uint64_t)st.st_size;
    c->source_mtime = (uint64_t)st.st_mtim.tv_sec * 1000000000ull + (uint64_t)st.st_mtim.tv_nsec;
    return SQ_OK;
}
static int sq_ingest_concordant_file_impl(sq_ctx* c, const char* path, int32_t n_threads) {
    if (!c || !path) return SQ_E_ARG;
    if (c->bwa) return fail(c, SQ_E_ARG, "this context holds a --bwa batch (sq_ingest_bwa_file): sq_clear_records before a STAR-mode ingest -- the mode is per context");
    { int r0 = sq_set_source(c, path); if (r0) return r0; }
    if (!std::getenv("SQUID_HOST_PARSE")) {
        // default: the host only inflates BGZF and finds record boundaries; K0 parses the records on the GPU
        const auto t_file0 = std::chrono::steady_clock::now();
        c->ingest_total_bytes = 0; c->ingest_seen_bytes = 0;
        const RefRange rr{c->shard.first_ref, c->shard.end_ref, c->P.rank == c->P.world_size - 1};
        const bool staged = !c->staged_path.empty() && c->staged_path == path;  // compressed bytes already in HBM (sq_stage_bam)
        struct DfileGuard { sq_ctx* c; ~DfileGuard() { c->ingest_dfile = nullptr; } } dfile_guard{c};
        if (staged) {
            struct stat st;
            if (::stat(path, &st) != 0 || (size_t)st.st_size != c->staged_bytes || (uint64_t)st.st_ino != c->staged_ino) return fail(c, SQ_E_IO, "the file changed since sq_stage_bam");
            if (((uint64_t)st.st_mtim.tv_sec * 1000000000ull + (uint64_t)st.st_mtim.tv_nsec) != c->staged_mtime) return fail(c, SQ_E_IO, "the file changed since sq_stage_bam");
            int r0 = dev_stage_file(c, nullptr, 0, &c->ingest_dfile);
            if (r0) return r0;
        }
        int rc = scan_bam_file(path, n_threads, c->err, [&](const uint8_t* bam, size_t nbytes, const unsigned long long* off, int64_t n) { c->ingest_seen_bytes += nbytes; return ingest_raw(c, bam, nbytes, off, n); },
                               [&](size_t total) { c->ingest_total_bytes = c->shard.on ? 0 : total; }, c->shard.on ? &rr : nullptr,
                               [&](const uint8_t* file, std::vector<BgzfRange>& blocks, size_t b0, size_t b1, size_t begin, bool synced, int nref, const IndexMore& more, size_t file_bytes, GpuFileSrc* src) {
                                   return dev_ingest_bgzf(c, file, blocks, b0, b1, begin, synced, nref, more, file_bytes, src); }, staged, true, !staged);
        c->ingest_total_bytes = 0;
        const double t_scan = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_file0).count();
        dev_flush_timers(c);
        if (std::getenv("SQUID_INGEST_TIMING")) std::fprintf(stderr, "ingest %s: reader returned after %.1f ms, timers flushed after %.1f ms\n", path, t_scan, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_file0).count());
        return rc;
    }
    if (c->chim_set.size() != c->chim_names.size()) { c->chim_set.clear(); c->chim_set.insert(c->chim_names.begin(), c->chim_names.end()); }
    ParseOpts o{c->P.phred_type, c->P.min_phred, c->P.max_lowphred_len, false, &c->chim_set};
    return parse_bam_file(path, o, (size_t)1 << 21, n_threads, c->err, [&](const HostBatch& hb) {
        sq_aln_batch b;
        hb.view(&b, false);
        return sq_ingest_concordant(c, &b);
    });
}
int sq_ingest_concordant_file(sq_ctx* c, const char* path, int32_t n_threads) { return abi_guard(c, "sq_ingest_concordant_file", [&]() { return sq_ingest_concordant_file_impl(c, path, n_threads); }); }
// `squid --bwa -b <bam>` (src/Config.cpp:98-100; src/main.cpp:33-37 runs without a chimeric file): every record of the one BAM file,
// decoded on host threads with its QNAME, stays on the host; sq_build_graph then takes BuildNode_BWA / RawEdges (sq_bwa.cpp)
static int sq_ingest_bwa_file_impl(sq_ctx* c, const char* path, int32_t n_threads) {
    if (!c || !path) return SQ_E_ARG;
    if (c->shard.on) return fail(c, SQ_E_ARG, "--bwa input is not chromosome-sharded");
    if (c->ref_len.empty()) return fail(c, SQ_E_ARG, "sq_set_references first");
    { int r0 = sq_set_source(c, path); if (r0) return r0; }
    std::shared_ptr<HostBatch> all = c->bwa_spare ? std::move(c->bwa_spare) : std::make_shared<HostBatch>();
    c->bwa_spare.reset();
    // (the arrays keep their storage -- and, on the way through the GPU reader, their sizes: every element is overwritten by the copy back)
    {   // a file of 1 GiB and more through the GPU reader (SQUID_BWA_GPU=1 / =0 forces / forbids it): BGZF inflate, record boundaries and the
        // record parse on the device, the QNAMEs kept next to the records (sq_ctx::capture_names, as for a large chimeric BAM), one copy
        // back -- the batch the host decoder below makes, field for field; the two order-dependent loops of the mode then run on it
        struct stat st;
        const char* env = std::getenv("SQUID_BWA_GPU");
        const bool want = env ? std::atoi(env) != 0 : (::stat(path, &st) == 0 && (size_t)st.st_size >= ((size_t)1 << 30));
        if (want && chimeric_records_through_the_device(c, path, n_threads, *all, false) == SQ_OK) {
            c->bwa = all;
            c->counts.n_concordant = (int64_t)all->size();
            c->counts.n_blocks = (int64_t)all->b_refpos.size();
            c->counts.chimeric_through_gpu_reader = 1;
            return SQ_OK;
        }
        all->clear();
        c->counts.chimeric_through_gpu_reader = 0;
    }
    ParseOpts o{c->P.phred_type, c->P.min_phred, c->P.max_lowphred_len, true, nullptr};
    const int rc = parse_bam_file(path, o, (size_t)1 << 21, std::max(1, (int)n_threads), c->err, [&](const HostBatch& hb) {
        if (all->names.size() + hb.names.size() >= 0xffffffffull || all->b_refpos.size() + hb.b_refpos.size() >= 0xffffffffull) return fail(c, SQ_E_CAPACITY, "--bwa input beyond 4 GB of read names or 2^32 aligned blocks");
        all->append(hb);
        return (int)SQ_OK;
    });
    if (rc) return rc;
    c->bwa = all;
    c->counts.n_concordant = (int64_t)all->size();
    c->counts.n_blocks = (int64_t)all->b_refpos.size();
    return SQ_OK;
}
int sq_ingest_bwa_file(sq_ctx* c, const char* path, int32_t n_threads) { return abi_guard(c, "sq_ingest_bwa_file", [&]() { return sq_ingest_bwa_file_impl(c, path, n_threads); }); }
// utils/JunctionSequence.cpp as a library call: host work only (no context, no device)
int sq_junction_sequences(const char* bedpe_path, const char* chim_bam_path, const char* fasta_path, const char* out_prefix, char* errbuf, size_t errcap) {
    if (!bedpe_path || !chim_bam_path || !fasta_path || !out_prefix) return SQ_E_ARG;
    sq_ctx local;  // (never sees a device: only the host pool and the parse parameters -- the utility's own defaults, :520-524)
    sq_default_params(&local.P);
    local.pool.reset(new HostPool((int)std::min(15u, std::max(1u, std::thread::hardware_concurrency()) - 1)));
    std::vector<std::string> names;
    std::vector<int32_t> lens;
    int rc = read_bam_header(chim_bam_path, names, lens, local.err);
    if (!rc) { local.ref_len = lens; rc = chimeric_fragments_host(&local, chim_bam_path, 8); }
    if (!rc) rc = junction_sequences(&local, names, bedpe_path, fasta_path, out_prefix);
    if (rc && errbuf && errcap) { std::strncpy(errbuf, local.err.c_str(), errcap - 1); errbuf[errcap - 1] = 0; }
    return rc;
}
int sq_stage_bam(sq_ctx* c, const char* path) {
    if (!c || !path) return SQ_E_ARG;
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return fail(c, SQ_E_IO, std::string("cannot open bamfile ") + path);
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size <= 0) { ::close(fd); return fail(c, SQ_E_IO, std::string("cannot read ") + path); }
    const size_t n = (size_t)st.st_size;
    void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);  // (straight from the page cache: no second host copy of the file)
    ::close(fd);
    if (m == MAP_FAILED) return fail(c, SQ_E_IO, std::string("cannot map ") + path);
    c->staged_path.clear(); c->staged_bytes = 0;
    const uint8_t* d = nullptr;
    int rc = dev_stage_file(c, (const uint8_t*)m, n, &d);
    munmap(m, n);
    if (rc) return rc;
    c->staged_path = path; c->staged_bytes = n;
    c->staged_ino = (uint64_t)st.st_ino;
    c->staged_mtime = (uint64_t)st.st_mtim.tv_sec * 1000000000ull + (uint64_t)st.st_mtim.tv_nsec;
    return SQ_OK;
}
int sq_clear_records(sq_ctx* c) {
    if (!c) return SQ_E_ARG;
    int rc = sq_reset(c);
    if (rc) return rc;
    dev_clear_records(c);
    // (a --bwa batch is gigabytes in fourteen arrays: the context keeps its storage for the next --bwa ingest -- unmapping it and faulting
    // fresh pages in again was 0.25 s per sample at C3 --; sq_release_reader_buffers gives it back)
    if (c->bwa && c->bwa.use_count() == 1) c->bwa_spare = std::move(c->bwa);
    c->bwa.reset();
    const int64_t side_by_side = c->counts.token_passes_side_by_side;  // (a property of the process, not of the records)
    c->counts = sq_counts{};
    c->counts.token_passes_side_by_side = side_by_side;
    c->counts.n_chimeric_records = c->n_chim_records; c->counts.n_chim_fragments = (int64_t)c->frags.size(); c->counts.read_len = c->read_len;
    return SQ_OK;
}
// ---- record cache (include/squid_hip.h): header, then the arrays of DeviceRecords / sq_aln_batch, each 64-byte aligned
namespace {
struct CacheHeader {
    char magic[8];  // "SQSOA1\0\0"
    int64_t n_rec, n_blk;
    int32_t phred_type, min_phred, max_lowphred_len, n_ref;
    uint64_t chim_hash;
    uint64_t reserved[3];
};
uint64_t chim_set_hash(const sq_ctx* c) {  // order-independent
    uint64_t sum = 0x9e3779b97f4a7c15ull * (c->chim_names.size() + 1);
    for (const std::string& nm : c->chim_names) {
        uint64_t h = 1469598103934665603ull;
        for (unsigned char ch : nm) { h ^= ch; h *= 1099511628211ull; }
        h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33;
        sum += h;
    }
    return sum;
}
CacheHeader cache_header(const sq_ctx* c, int64_t n_rec, int64_t n_blk) {
    CacheHeader h;
    std::memset(&h, 0, sizeof h);
    std::memcpy(h.magic, "SQSOA1", 6);
    h.n_rec = n_rec; h.n_blk = n_blk;
    h.phred_type = c->P.phred_type; h.min_phred = c->P.min_phred; h.max_lowphred_len = c->P.max_lowphred_len; h.n_ref = (int32_t)c->ref_len.size();
    h.chim_hash = chim_set_hash(c);
    h.reserved[0] = c->source_size; h.reserved[1] = c->source_mtime;  // (0,0: written from host batches, no file to bind to)
    return h;
}
size_t pad64(size_t n) { return (n + 63) & ~(size_t)63; }
}  // namespace
int sq_save_records(sq_ctx* c, const char* path) {
    if (!c || !path) return SQ_E_ARG;
    HostBatch hb;
    int rc = dev_download_records(c, hb);
    if (rc) return rc;
    const size_t n = hb.refid.size(), nb = hb.b_refpos.size();
    if (hb.blk_off.size() != n + 1) hb.blk_off.assign(n + 1, 0);  // (no record: the offsets array is just {0})
    const CacheHeader h = cache_header(c, (int64_t)n, (int64_t)nb);
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(c, SQ_E_IO, std::string("cannot write ") + path);
    bool ok = std::fwrite(&h, sizeof h, 1, f) == 1;
    static const char zeros[64] = {0};
    auto put = [&](const void* p, size_t bytes) {
        if (bytes) ok = ok && std::fwrite(p, 1, bytes, f) == bytes;
        if (pad64(bytes) > bytes) ok = ok && std::fwrite(zeros, 1, pad64(bytes) - bytes, f) == pad64(bytes) - bytes;
    };
    put(hb.refid.data(), n * 4); put(hb.pos.data(), n * 4); put(hb.mrefid.data(), n * 4); put(hb.mpos.data(), n * 4); put(hb.endpos.data(), n * 4);
    put(hb.flag.data(), n * 2); put(hb.totlen.data(), n * 2); put(hb.mapq.data(), n); put(hb.aux.data(), n); put(hb.blk_off.data(), (n + 1) * 4);
    put(hb.b_refpos.data(), nb * 4); put(hb.b_matchref.data(), nb * 4); put(hb.b_readpos.data(), nb * 2); put(hb.b_matchread.data(), nb * 2);
    ok = std::fclose(f) == 0 && ok;
    if (!ok) return fail(c, SQ_E_IO, std::string("short write to ") + path);
    return SQ_OK;
}
static int sq_load_records_impl(sq_ctx* c, const char* path) {
    if (!c || !path) return SQ_E_ARG;
    if (c->ref_len.empty()) return fail(c, SQ_E_ARG, "sq_set_references first");
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(c, SQ_E_IO, std::string("cannot open ") + path);
    CacheHeader h;
    if (std::fread(&h, sizeof h, 1, f) != 1 || std::memcmp(h.magic, "SQSOA1\0", 8) != 0 || h.n_rec < 0 || h.n_blk < 0) { std::fclose(f); return fail(c, SQ_E_IO, std::string(path) + " is not a record cache"); }
    if (c->counts.n_concordant != 0) { std::fclose(f); return fail(c, SQ_E_ARG, "sq_load_records on a context that already holds concordant records"); }
    const CacheHeader want = cache_header(c, h.n_rec, h.n_blk);
    if ((h.reserved[0] || h.reserved[1]) && (want.reserved[0] || want.reserved[1]) && (h.reserved[0] != want.reserved[0] || h.reserved[1] != want.reserved[1])) {
        std::fclose(f);
        return fail(c, SQ_E_ARG, "record cache was written from another concordant BAM (size or modification time differ)");
    }
    if (h.phred_type != want.phred_type || h.min_phred != want.min_phred || h.max_lowphred_len != want.max_lowphred_len || h.n_ref != want.n_ref || h.chim_hash != want.chim_hash) {
        std::fclose(f);
        return fail(c, SQ_E_ARG, "record cache was written with other -pt/-pl/-pm, another reference list or another chimeric BAM");
    }
    // in pieces of 16 M records: bounded host memory whatever the size of the cache
    const size_t n = (size_t)h.n_rec, nb = (size_t)h.n_blk;
    size_t at = sizeof h;
    const size_t o_refid = at; at += pad64(n * 4);
    const size_t o_pos = at; at += pad64(n * 4);
    const size_t o_mrefid = at; at += pad64(n * 4);
    const size_t o_mpos = at; at += pad64(n * 4);
    const size_t o_endpos = at; at += pad64(n * 4);
    const size_t o_flag = at; at += pad64(n * 2);
    const size_t o_totlen = at; at += pad64(n * 2);
    const size_t o_mapq = at; at += pad64(n);
    const size_t o_aux = at; at += pad64(n);
    const size_t o_blkoff = at; at += pad64((n + 1) * 4);
    const size_t o_brefpos = at; at += pad64(nb * 4);
    const size_t o_bmatchref = at; at += pad64(nb * 4);
    const size_t o_breadpos = at; at += pad64(nb * 2);
    const size_t o_bmatchread = at;
    bool ok = true;
    auto get = [&](void* dst, size_t off, size_t bytes) { if (bytes) ok = ok && fseeko(f, (off_t)off, SEEK_SET) == 0 && std::fread(dst, 1, bytes, f) == bytes; };
    const size_t piece = (size_t)1 << 24;
    HostBatch hb;
    int rc = SQ_OK;
    for (size_t r0 = 0; r0 < n && ok && rc == SQ_OK; r0 += piece) {
        const size_t r1 = std::min(n, r0 + piece), k = r1 - r0;
        hb.clear();
        hb.refid.resize(k); hb.pos.resize(k); hb.mrefid.resize(k); hb.mpos.resize(k); hb.endpos.resize(k); hb.flag.resize(k); hb.totlen.resize(k); hb.mapq.resize(k); hb.aux.resize(k); hb.blk_off.resize(k + 1);
        get(hb.refid.data(), o_refid + r0 * 4, k * 4); get(hb.pos.data(), o_pos + r0 * 4, k * 4); get(hb.mrefid.data(), o_mrefid + r0 * 4, k * 4); get(hb.mpos.data(), o_mpos + r0 * 4, k * 4);
        get(hb.endpos.data(), o_endpos + r0 * 4, k * 4); get(hb.flag.data(), o_flag + r0 * 2, k * 2); get(hb.totlen.data(), o_totlen + r0 * 2, k * 2); get(hb.mapq.data(), o_mapq + r0, k);
        get(hb.aux.data(), o_aux + r0, k); get(hb.blk_off.data(), o_blkoff + r0 * 4, (k + 1) * 4);
        if (!ok) break;
        const uint32_t bo0 = hb.blk_off[0], bo1 = hb.blk_off[k];
        if (bo1 < bo0 || bo1 > nb) { ok = false; break; }
        const size_t kb = bo1 - bo0;
        hb.b_refpos.resize(kb); hb.b_matchref.resize(kb); hb.b_readpos.resize(kb); hb.b_matchread.resize(kb);
        get(hb.b_refpos.data(), o_brefpos + (size_t)bo0 * 4, kb * 4); get(hb.b_matchref.data(), o_bmatchref + (size_t)bo0 * 4, kb * 4);
        get(hb.b_readpos.data(), o_breadpos + (size_t)bo0 * 2, kb * 2); get(hb.b_matchread.data(), o_bmatchread + (size_t)bo0 * 2, kb * 2);
        if (!ok) break;
        for (uint32_t& o : hb.blk_off) o -= bo0;
        sq_aln_batch b;
        hb.view(&b, false);
        rc = sq_ingest_concordant(c, &b);
    }
    std::fclose(f);
    if (!ok) return fail(c, SQ_E_IO, std::string("truncated or corrupt record cache ") + path);
    return rc;
}
int sq_load_records(sq_ctx* c, const char* path) { return abi_guard(c, "sq_load_records", [&]() { return sq_load_records_impl(c, path); }); }
static int sq_build_graph_impl(sq_ctx* c) {
    if (!c) return SQ_E_ARG;
    if (c->ref_len.empty()) return fail(c, SQ_E_ARG, "sq_set_references first");
    if (c->read_len <= 0 && !c->bwa) return fail(c, SQ_E_ARG, "sq_ingest_chimeric first (ReadLen comes from the chimeric BAM)");
    if (c->shard.on != (c->P.world_size > 1)) return fail(c, SQ_E_ARG, "world_size > 1 needs sq_set_shard (and the other way round)");
    if (c->bp_future.valid()) (void)c->bp_future.get();
    if (!c->gb && !c->timer_keep) c->timer.clear();
    if (!c->gb) c->ablated = false;
    int rc = c->bwa ? build_graph_bwa(c) : build_graph(c);
    dev_flush_timers(c);
    if (rc <= 0 && c->ablated) {  // (the timers of the run stay readable; its graph does not exist -- whatever the mutilated pass ran into)
        if (c->bp_future.valid()) (void)c->bp_future.get();
        c->graph_built = false;
        rc = fail(c, SQ_E_ARG, "a timing-only switch (SQUID_P1_ABLATE / SQUID_EDGES_ABLATE) is set: the graph of this run is wrong and is not handed out");
    }
    if (rc < 0) { c->gb.reset(); c->x_pending = false; }
    return rc;
}
int sq_build_graph(sq_ctx* c) { return abi_guard(c, "sq_build_graph", [&]() { return sq_build_graph_impl(c); }); }
int sq_graph_view(sq_ctx* c, int32_t stage, sq_graph* g) {
    if (!c || !g || stage < 0 || stage > 5 || !c->graph_built) return SQ_E_ARG;
    if (stage != 0 && !c->keep_stages) return fail(c, SQ_E_ARG, "the intermediate graphs were not kept (sq_keep_stage_graphs(ctx, 0) before sq_build_graph)");
    c->snap[stage].view(g);
    return SQ_OK;
}
static int sq_order_impl(sq_ctx* c, sq_orders* o) {
    if (!c || !c->graph_built) return SQ_E_ARG;
    if (!c->ordered) { int rc = order_components(c); dev_flush_timers(c); if (rc) return rc; }
    if (o) { o->n_components = (int32_t)c->ord_off.size() - 1; o->comp_off = c->ord_off.data(); o->nodes = c->ord_nodes.data(); }
    return SQ_OK;
}
int sq_order(sq_ctx* c, sq_orders* o) { return abi_guard(c, "sq_order", [&]() { return sq_order_impl(c, o); }); }
static int sq_total_order_impl(sq_ctx* c, sq_orders* o) {
    if (!c || !o || !c->graph_built) return SQ_E_ARG;
    if (!c->ordered) { int rc = order_components(c); dev_flush_timers(c); if (rc) return rc; }
    const int rc = total_order(c);
    if (rc) return rc;
    o->n_components = (int32_t)c->tot_off.size() - 1; o->comp_off = c->tot_off.data(); o->nodes = c->tot_nodes.data();
    return SQ_OK;
}
int sq_total_order(sq_ctx* c, sq_orders* o) { return abi_guard(c, "sq_total_order", [&]() { return sq_total_order_impl(c, o); }); }
static int sq_call_sv_impl(sq_ctx* c, sq_sv_table* t) {
    if (!c || !c->graph_built) return SQ_E_ARG;
    int rc = call_sv(c);
    dev_flush_timers(c);
    if (rc < 0) { c->svb.reset(); c->x_pending = false; }
    if (rc) return rc;
    if (t) {
        t->n_rows = (int32_t)c->sv_cols[0].size();
        t->chr1 = c->sv_cols[0].data(); t->start1 = c->sv_cols[1].data(); t->end1 = c->sv_cols[2].data();
        t->chr2 = c->sv_cols[3].data(); t->start2 = c->sv_cols[4].data(); t->end2 = c->sv_cols[5].data();
        t->score = c->sv_cols[6].data(); t->sup1 = c->sv_cols[7].data(); t->sup2 = c->sv_cols[8].data();
        t->strand1_minus = c->sv_s1.data(); t->strand2_minus = c->sv_s2.data();
    }
    return SQ_OK;
}
int sq_call_sv(sq_ctx* c, sq_sv_table* t) { return abi_guard(c, "sq_call_sv", [&]() { return sq_call_sv_impl(c, t); }); }
static int sq_breakpoints_impl(sq_ctx* c, sq_bp_table* t) {
    if (!c || !t || c->bp_off.empty()) return SQ_E_ARG;
    t->n_edges = (int32_t)c->bp_off.size() - 1;
    t->bp_off = c->bp_off.data(); t->bp1 = c->bp1.data(); t->bp2 = c->bp2.data(); t->sup1 = c->bsup1.data(); t->sup2 = c->bsup2.data();
    return SQ_OK;
}
int sq_breakpoints(sq_ctx* c, sq_bp_table* t) { return abi_guard(c, "sq_breakpoints", [&]() { return sq_breakpoints_impl(c, t); }); }
int sq_debug_token_bench(sq_ctx* c, const char* path, int32_t variant, int32_t max_blocks, int32_t reps, int32_t check, double* out7) {
    if (!c || !path || !out7 || max_blocks <= 0 || reps <= 0) return SQ_E_ARG;
    return abi_guard(c, "sq_debug_token_bench", [&]() { return dev_token_bench(c, path, variant, max_blocks, reps, check, out7); });
}
int sq_get_timing(sq_ctx* c, sq_timing* t) {
    if (!c || !t) return SQ_E_ARG;
    t->n = (int32_t)c->timer.names.size();
    t->names = c->timer.names.data(); t->ms = c->timer.ms.data(); t->launches = c->timer.launches.data(); t->bytes = c->timer.bytes.data(); t->busy_ms = c->timer.busy.data();
    return SQ_OK;
}
int sq_drop_file_cache(void) { drop_file_cache(); return SQ_OK; }
int sq_release_reader_buffers(sq_ctx* c) {
    if (!c) return SQ_E_ARG;
    c->staged_path.clear(); c->staged_bytes = 0;
    drop_whole_file_scratch();
    if (!c->chim_future.valid()) c->chim_decoded.reset();  // (a pairing still running reads it)
    c->bwa_spare.reset();
    return dev_release_reader(c);
}
int sq_keep_stage_graphs(sq_ctx* c, int32_t on) {
    if (!c) return SQ_E_ARG;
    c->keep_stages = on != 0;
    return SQ_OK;
}
int sq_keep_host_memory(void) {
    // (mallopt takes ints: the thresholds stop at 2 GiB - 1; thread arenas -- where the context's host threads allocate -- are kept
    // whole by the top pad: an arena heap is only unmapped when more than the pad would stay free in the one before it)
    const int ok = mallopt(M_TRIM_THRESHOLD, INT32_MAX) & mallopt(M_TOP_PAD, INT32_MAX & ~4095) & mallopt(M_MMAP_THRESHOLD, 32 << 20);
    return ok ? SQ_OK : SQ_E_ARG;
}
int sq_timing_accumulate(sq_ctx* c, int32_t keep) {
    if (!c) return SQ_E_ARG;
    c->timer_keep = keep != 0;
    if (keep) c->timer.clear();
    return SQ_OK;
}
int sq_reset(sq_ctx* c) {
    if (!c) return SQ_E_ARG;
    if (c->bp_future.valid()) (void)c->bp_future.get();
    copy_frags(c, c->frags0, c->frags);  // the graph stages trim the chimeric blocks in place, like the reference does
    c->nodes.clear(); c->edges.clear(); c->label.clear();
    c->graph_built = false; c->ordered = false;
    c->bp_off.clear();
    c->gb.reset(); c->svb.reset(); c->x_pending = false; c->x_ready = false;
    return SQ_OK;
}
int sq_get_counts(sq_ctx* c, sq_counts* k) {
    if (!c || !k) return SQ_E_ARG;
    *k = c->counts;
    k->replay_candidates_checked = c->replay_checked.load(); k->replay_count_mismatches = c->replay_mismatch.load();
    return SQ_OK;
}
int sq_debug_download(sq_ctx* c, sq_aln_batch* b) {
    if (!c || !b) return SQ_E_ARG;
    static thread_local HostBatch hb;
    int rc = dev_download_records(c, hb);
    if (rc) return rc;
    hb.view(b, false);
    return SQ_OK;
}
int sq_debug_bp_support(sq_ctx* c, int32_t n_bp, const int32_t* chr, const int32_t* pos, int32_t* coverage, int32_t host_walk) {
    if (!c || n_bp < 0 || (n_bp && (!chr || !pos || !coverage))) return SQ_E_ARG;
    if (!c->graph_built) return fail(c, SQ_E_ARG, "sq_debug_bp_support before sq_build_graph");
    std::vector<std::pair<int, int>> bps(n_bp);
    for (int i = 0; i < n_bp; ++i) bps[i] = {chr[i], pos[i]};
    if (!std::is_sorted(bps.begin(), bps.end())) return fail(c, SQ_E_ARG, "breakpoints must be sorted by (chr, pos)");
    std::vector<int32_t> cov;
    int rc = host_walk ? dev_breakpoint_support_exact(c, bps, cov) : dev_breakpoint_support(c, bps, cov);
    dev_flush_timers(c);
    if (rc) return rc;
    std::copy(cov.begin(), cov.end(), coverage);
    return SQ_OK;
}
int sq_debug_order(sq_ctx* c, int32_t n, int32_t n_edges, const int32_t* edges5, int32_t use_gpu, int32_t* mask, int32_t* order, int64_t* value) {
    if (!c || n_edges < 0 || (n_edges && !edges5) || !mask || !order || !value) return SQ_E_ARG;
    std::vector<int32_t> e(edges5, edges5 + 5 * (size_t)n_edges), o;
    int32_t m = 0; int64_t v = 0;
    int rc = order_problem_debug(c, n, e, use_gpu != 0, m, o, v);
    dev_flush_timers(c);
    if (rc) return rc;
    *mask = m; *value = v;
    std::copy(o.begin(), o.end(), order);
    return SQ_OK;
}
int sq_debug_blocks(int32_t n, const int32_t* f, uint8_t* rel5, int32_t* perm_pos, int32_t* perm_readpos) {
    if (n < 0 || (n && (!f || !rel5 || !perm_pos || !perm_readpos))) return SQ_E_ARG;
    std::vector<Blk> v((size_t)n);
    for (int i = 0; i < n; ++i) v[(size_t)i] = Blk{f[7 * i], f[7 * i + 1], f[7 * i + 2], f[7 * i + 3], f[7 * i + 4], f[7 * i + 5] != 0, f[7 * i + 6] != 0};
    const size_t nn = (size_t)n * (size_t)n;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            const size_t at = (size_t)i * (size_t)n + (size_t)j;
            rel5[at] = blk_less_pos(v[i], v[j]); rel5[nn + at] = blk_greater_pos(v[i], v[j]); rel5[2 * nn + at] = blk_eq_pos(v[i], v[j]);
            rel5[3 * nn + at] = blk_same(v[i], v[j]); rel5[4 * nn + at] = blk_less_readpos(v[i], v[j]);
        }
    // the library sorts the blocks themselves (sq_segment.cpp: bamdiscordant; sq_chimeric.cpp: SortbyReadPos); the input
    // index rides along in a parallel array by sorting (block, index) pairs with the same comparator
    std::vector<std::pair<Blk, int32_t>> s((size_t)n);
    for (int i = 0; i < n; ++i) s[(size_t)i] = {v[(size_t)i], i};
    std::sort(s.begin(), s.end(), [](const std::pair<Blk, int32_t>& x, const std::pair<Blk, int32_t>& y) { return blk_less_pos(x.first, y.first); });
    for (int i = 0; i < n; ++i) perm_pos[i] = s[(size_t)i].second;
    for (int i = 0; i < n; ++i) s[(size_t)i] = {v[(size_t)i], i};
    std::sort(s.begin(), s.end(), [](const std::pair<Blk, int32_t>& x, const std::pair<Blk, int32_t>& y) { return blk_less_readpos(x.first, y.first); });
    for (int i = 0; i < n; ++i) perm_readpos[i] = s[(size_t)i].second;
    return SQ_OK;
}
int sq_set_shard(sq_ctx* c, int32_t first_ref, int32_t end_ref) {
    if (!c) return SQ_E_ARG;
    if (c->P.world_size <= 1) return fail(c, SQ_E_ARG, "sq_set_shard needs sq_params.world_size > 1");
    if (c->P.rank < 0 || c->P.rank >= c->P.world_size) return fail(c, SQ_E_ARG, "sq_params.rank out of range");
    if (c->ref_len.empty()) return fail(c, SQ_E_ARG, "sq_set_references first");
    if (first_ref < 0 || end_ref < first_ref || end_ref > (int32_t)c->ref_len.size()) return fail(c, SQ_E_ARG, "bad RefID range");
    if (c->counts.n_concordant) return fail(c, SQ_E_ARG, "sq_set_shard after concordant records were ingested");
    c->shard = Shard();
    c->shard.on = true; c->shard.first_ref = first_ref; c->shard.end_ref = end_ref;
    return SQ_OK;
}
int sq_exchange_pack(sq_ctx* c, const void** buf, int64_t* nbytes) {
    if (!c || !buf || !nbytes) return SQ_E_ARG;
    if (!c->x_pending) return fail(c, SQ_E_ARG, "no exchange is pending (sq_build_graph / sq_call_sv return SQ_NEED_EXCHANGE first)");
    *buf = c->xbuf.data(); *nbytes = (int64_t)c->xbuf.size();
    return SQ_OK;
}
static int sq_exchange_unpack_impl(sq_ctx* c, const void* gathered, const int64_t* nbytes_per_rank, int32_t world_size) {
    if (!c || !nbytes_per_rank || world_size != c->P.world_size) return SQ_E_ARG;
    if (!c->x_pending) return fail(c, SQ_E_ARG, "no exchange is pending");
    c->xgot.assign(world_size, {});
    const uint8_t* p = (const uint8_t*)gathered;
    for (int r = 0; r < world_size; ++r) {
        if (nbytes_per_rank[r] < 0 || (nbytes_per_rank[r] && !p)) return SQ_E_ARG;
        c->xgot[r].assign(p, p + nbytes_per_rank[r]);
        p += nbytes_per_rank[r];
    }
    if (c->xgot[c->P.rank] != c->xbuf) return fail(c, SQ_E_ARG, "sq_exchange_unpack: this rank's slot does not hold what sq_exchange_pack returned");
    c->x_pending = false; c->x_ready = true;
    return SQ_OK;
}
int sq_exchange_unpack(sq_ctx* c, const void* gathered, const int64_t* nbytes_per_rank, int32_t world_size) { return abi_guard(c, "sq_exchange_unpack", [&]() { return sq_exchange_unpack_impl(c, gathered, nbytes_per_rank, world_size); }); }

}  // extern "C"
