// C-ABI entry points of libsquid_hip.so (include/squid_hip.h) and the stage pipeline behind them.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <future>
#include <memory>

#include "sq_internal.h"

namespace sq {

int fail(sq_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

int Timer::slot(const char* name) {
    for (size_t i = 0; i < names.size(); ++i) if (names[i] == name || !std::strcmp(names[i], name)) return (int)i;
    names.push_back(name); ms.push_back(0); bytes.push_back(0); launches.push_back(0);
    return (int)names.size() - 1;
}
void Timer::add(const char* name, double ms_, double bytes_, int64_t n) {
    int s = slot(name);
    ms[s] += ms_; bytes[s] += bytes_; launches[s] += n;
}
void Timer::clear() { names.clear(); ms.clear(); bytes.clear(); launches.clear(); }

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

struct HostClock {
    sq_ctx* c; const char* name; std::chrono::steady_clock::time_point t0;
    HostClock(sq_ctx* c, const char* name) : c(c), name(name), t0(std::chrono::steady_clock::now()) {}
    ~HostClock() { c->timer.add(name, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); }
};

static int build_graph(sq_ctx* c) {
    c->timer.clear();
    HostClock wall(c, "wall_build_graph");
    c->graph_built = false;
    c->ordered = false;
    std::vector<StreamRec> recs;
    std::vector<int32_t> rest_p, rest_m;
    int rc = dev_classify_and_summarise(c, recs, rest_p, rest_m);
    if (rc) return rc;
    std::vector<Node> seeds;
    std::vector<Blk> disc;
    int64_t n_break = 0;
    std::shared_ptr<SegPlan> plan;
    {
        HostClock hc(c, "host_segment_prepare");
        rc = segment_prepare(c, plan, n_break, disc);
        if (rc) return rc;
    }
    c->counts.n_break = n_break;
    // ReadsOther (non-first blocks) is sorted with an unstable std::sort in the reference (SegmentGraph.cpp:781) and a
    // block of <= 3 bases right behind a node boundary is counted for whichever node the sweep cursor is on, which
    // depends on that sort's tie order.  Node depths only feed the coverage-ratio test of FilterEdges, so by default
    // the GPU reports canonical depths plus bounds over all tie orders and FilterEdges checks that no decision can
    // change inside the bounds; only then (or when SQUID_EXACT_DEPTH is set, as the stage-parity tests do) is the
    // reference's sort repeated on the host and the sweep walked exactly.
    const bool exact_mode = std::getenv("SQUID_EXACT_DEPTH") != nullptr;
    struct OtherWork { std::vector<int32_t> chr, pos, len; struct R { int32_t chr, pos, len; }; std::vector<R> sorted; bool has_tiny = false; };
    std::shared_ptr<OtherWork> ow = std::make_shared<OtherWork>();
    std::future<void> other_sorted;
    auto start_exact_sort = [&]() -> int {
        int r2 = dev_gather_other(c, n_break, ow->has_tiny, ow->chr, ow->pos, ow->len);
        if (r2) return r2;
        if (ow->has_tiny)
            other_sorted = std::async(std::launch::async, [ow]() {
                ow->sorted.resize(ow->chr.size());
                for (size_t i = 0; i < ow->sorted.size(); ++i) ow->sorted[i] = OtherWork::R{ow->chr[i], ow->pos[i], ow->len[i]};
                std::sort(ow->sorted.begin(), ow->sorted.end(), [](const OtherWork::R& a, const OtherWork::R& b) { return a.chr != b.chr ? a.chr < b.chr : a.pos < b.pos; });
            });
        return SQ_OK;
    };
    if (exact_mode) { rc = start_exact_sort(); if (rc) return rc; }
    {
        HostClock hc(c, "host_segment_replay");
        rc = segment_replay(c, *plan, seeds);
    }
    if (rc) { if (other_sorted.valid()) other_sorted.get(); return rc; }
    {
        HostClock hc(c, "host_tile_genome");
        std::vector<Node> seedcopy = seeds;
        rc = tile_genome(c, seedcopy, c->nodes);
    }
    if (rc) { if (other_sorted.valid()) other_sorted.get(); return rc; }
    // per-node Support / AvgDepth (SegmentGraph.cpp:766-826): discordant blocks on the host, stream blocks from the GPU
    std::vector<Node>& N = c->nodes;
    const int nn = (int)N.size();
    std::vector<int32_t> dis_cnt(nn), dis_sum(nn);
    {
        HostClock hc(c, "host_depth_discordant");
        size_t it = 0;
        for (int i = 0; i < nn; ++i) {
            int cnt = 0, sum = 0;
            for (; it != disc.size() && disc[it].refid == N[i].chr && disc[it].refpos < N[i].pos + N[i].len; ++it)
                if (disc[it].refpos >= N[i].pos && disc[it].refpos + disc[it].matchref <= N[i].pos + N[i].len) { ++cnt; sum += disc[it].matchref; }
            dis_cnt[i] = cnt; dis_sum[i] = sum;
        }
    }
    std::vector<int32_t> sup, amb_plus, amb_minus, unused;
    std::vector<int64_t> sl;
    bool tiny_boundary = false;
    rc = dev_node_depth(c, N, n_break, sup, sl, tiny_boundary, amb_plus, amb_minus, unused);
    if (rc) { if (other_sorted.valid()) other_sorted.get(); return rc; }
    // combine in the reference's order: discordant, ReadsMain, ReadsOther, then the division (only when ReadsOther is
    // non-empty, ledger B13).  `other` = per-node (count, sum) of ReadsOther, either canonical or from the exact sweep.
    auto set_depths = [&](const std::vector<int32_t>& ocnt, const std::vector<int32_t>& osum, bool bounds) {
        const int64_t n_other = sup[2 * nn];
        c->depth_bounds = bounds;
        for (int i = 0; i < nn; ++i) {
            N[i].support = dis_cnt[i] + sup[i];
            double d = dis_sum[i];
            d += (int32_t)sl[i];
            double lo = d, hi = d;
            if (n_other != 0) {
                N[i].support += ocnt[i];
                d += osum[i];
                lo = d; hi = d;
                if (bounds) { lo = d - amb_minus[i]; hi = d + amb_plus[i]; }
                d = 1.0 * d / N[i].len; lo = 1.0 * lo / N[i].len; hi = 1.0 * hi / N[i].len;
            }
            N[i].depth = d; N[i].depth_lo = lo; N[i].depth_hi = hi;
        }
    };
    auto exact_sweep = [&](std::vector<int32_t>& ocnt, std::vector<int32_t>& osum) {
        if (other_sorted.valid()) other_sorted.get();
        ocnt.assign(nn, 0); osum.assign(nn, 0);
        size_t it = 0;
        for (int i = 0; i < nn; ++i)
            for (; it != ow->sorted.size(); ++it) {
                const OtherWork::R& r = ow->sorted[it];
                if (r.chr == N[i].chr && r.pos >= N[i].pos - 3 && r.pos + r.len <= N[i].pos + N[i].len + 3) { ocnt[i]++; osum[i] += r.len; }
                else if (r.pos >= N[i].pos + N[i].len || r.chr != N[i].chr) break;
            }
    };
    std::vector<int32_t> ocnt(nn), osum(nn);
    for (int i = 0; i < nn; ++i) { ocnt[i] = sup[nn + i]; osum[i] = (int32_t)sl[nn + i]; }
    c->depth_ambiguous = false;
    if (exact_mode && tiny_boundary) {
        HostClock hc(c, "host_depth_exact_sweep");
        exact_sweep(ocnt, osum);
        set_depths(ocnt, osum, false);
    } else {
        if (other_sorted.valid()) other_sorted.get();
        set_depths(ocnt, osum, tiny_boundary);
    }
    c->edges.clear();
    std::vector<Edge> raw, conc;
    {
        HostClock hc(c, "host_chimeric_edges");
        rc = chimeric_edges(c, raw);
        if (rc) return rc;
    }
    rc = dev_concordant_edges(c, c->nodes, conc);
    if (rc) return rc;
    c->snap[1].take(c->nodes, c->edges, nullptr);
    {
        HostClock hc(c, "host_edge_reduce");
        raw.insert(raw.end(), conc.begin(), conc.end());
        reduce_edges(raw, c->edges);
    }
    c->snap[2].take(c->nodes, c->edges, nullptr);
    {
        HostClock hc(c, "host_filters");
        filter_by_weight(c);
        c->snap[3].take(c->nodes, c->edges, nullptr);
        std::vector<uint8_t> keep;
        filter_by_interleaving(c, keep);
        std::vector<Edge> before = c->edges;
        filter_edges(c, keep);
        if (c->depth_ambiguous) {
            // some coverage-ratio decision depends on the tie order: repeat the reference's sort and sweep, then redo the
            // filter with the exact depths
            HostClock hc2(c, "host_depth_exact_retry");
            rc = start_exact_sort();
            if (rc) return rc;
            exact_sweep(ocnt, osum);
            set_depths(ocnt, osum, false);
            c->depth_ambiguous = false;
            c->edges = before;
            filter_edges(c, keep);
        }
        c->snap[4].take(c->nodes, c->edges, nullptr);
    }
    {
        HostClock hc(c, "host_compress");
        rc = compress_nodes(c);
        if (rc) return rc;
        c->snap[5].take(c->nodes, c->edges, nullptr);
        rc = further_compress(c);
        if (rc) return rc;
    }
    rc = dev_connected_components(c, (int)c->nodes.size(), c->edges, c->label);
    if (rc) return rc;
    multiply_discordant(c, false);
    c->snap[0].take(c->nodes, c->edges, &c->label);
    c->graph_built = true;
    return SQ_OK;
}

static int call_sv(sq_ctx* c) {
    HostClock wall(c, "wall_call_sv");
    if (!c->ordered) return fail(c, SQ_E_ARG, "sq_call_sv before sq_order");
    const std::vector<Node>& N = c->nodes;
    std::vector<Edge>& E = c->edges;
    BPMap bpmap;
    {
        HostClock hc(c, "host_exact_breakpoints");
        int rc = exact_breakpoints(c, bpmap);
        if (rc) return rc;
    }
    // breakpoint list of every edge (SegmentGraph.cpp:3091-3109)
    auto edge_bps = [&](const Edge& e, std::vector<std::pair<std::pair<int, int>, std::pair<int, int>>>& out, bool& exact) {
        out.clear();
        BPMap::const_iterator it = bpmap.find(edge_pack(e));
        exact = it != bpmap.end() && !it->second.empty();
        if (exact) for (const auto& p : it->second) out.push_back({{N[e.a].chr, p.first}, {N[e.b].chr, p.second}});
        else out.push_back({{N[e.a].chr, e.ha ? N[e.a].pos : N[e.a].pos + N[e.a].len}, {N[e.b].chr, e.hb ? N[e.b].pos : N[e.b].pos + N[e.b].len}});
    };
    std::vector<std::pair<int, int>> BPs;
    std::vector<std::pair<std::pair<int, int>, std::pair<int, int>>> tmp;
    bool exact;
    for (const Edge& e : E) { edge_bps(e, tmp, exact); for (auto& p : tmp) { BPs.push_back(p.first); BPs.push_back(p.second); } }
    std::sort(BPs.begin(), BPs.end());
    std::vector<int32_t> cov;
    static const bool bp_host = getenv("SQUID_BP_HOST") != nullptr;  // debug cross-check of k_bp_walk
    int rc = bp_host ? dev_breakpoint_support_exact(c, BPs, cov) : dev_breakpoint_support(c, BPs, cov);
    if (rc) return rc;
    // per-edge table in key order (parity tests) -- before the weight sort
    c->bp_off.assign(1, 0); c->bp1.clear(); c->bp2.clear(); c->bsup1.clear(); c->bsup2.clear();
    std::map<uint64_t, std::vector<std::pair<int, int>>> support;
    for (const Edge& e : E) {
        edge_bps(e, tmp, exact);
        std::vector<std::pair<int, int>>& s = support[edge_pack(e)];
        for (auto& p : tmp) {
            int i1 = (int)(std::lower_bound(BPs.begin(), BPs.end(), p.first) - BPs.begin()), i2 = (int)(std::lower_bound(BPs.begin(), BPs.end(), p.second) - BPs.begin());
            s.push_back({cov[i1], cov[i2]});
            c->bp1.push_back(exact ? p.first.second : -1); c->bp2.push_back(exact ? p.second.second : -1);
            c->bsup1.push_back(cov[i1]); c->bsup2.push_back(cov[i2]);
        }
        c->bp_off.push_back((int32_t)c->bp1.size());
    }
    multiply_discordant(c, true);
    // WriteBEDPE (src/WriteIO.cpp:45-124): unstable sort by weight on the key-sorted edge list (ledger B8)
    HostClock hc(c, "host_select_sv");
    std::vector<Edge> W = E;
    std::sort(W.begin(), W.end(), [](Edge a, Edge b) { return a.w > b.w; });
    // rank / sign of every node in its component order
    std::vector<int> comp(N.size(), -1), rank(N.size(), -1), sign(N.size(), 1);
    for (size_t k = 0; k + 1 < c->ord_off.size(); ++k)
        for (int j = c->ord_off[k]; j < c->ord_off[k + 1]; ++j) {
            int v = std::abs(c->ord_nodes[j]) - 1;
            comp[v] = (int)k; rank[v] = j - c->ord_off[k]; sign[v] = c->ord_nodes[j] < 0 ? -1 : 1;
        }
    for (auto& col : c->sv_cols) col.clear();
    c->sv_s1.clear(); c->sv_s2.clear();
    for (const Edge& e : W) {
        const Node &a = N[e.a], &b = N[e.b];
        bool conc = a.chr == b.chr && e.ha == 0 && e.hb == 1 && (b.pos - a.pos - a.len <= c->P.concord_dist_pos || e.b - e.a <= c->P.concord_dist_idx);
        if (conc) continue;
        bool ok = false;
        if (comp[e.a] == comp[e.b] && rank[e.a] < rank[e.b] && (bool)e.ha == (sign[e.a] < 0) && (bool)e.hb == (sign[e.b] > 0)) ok = true;
        else if (comp[e.a] == comp[e.b] && rank[e.a] > rank[e.b] && (bool)e.hb == (sign[e.b] < 0) && (bool)e.ha == (sign[e.a] > 0)) ok = true;
        if (!ok) continue;
        edge_bps(e, tmp, exact);
        const std::vector<std::pair<int, int>>& s = support[edge_pack(e)];
        for (size_t k = 0; k < tmp.size(); ++k) {
            int b1 = tmp[k].first.second, b2 = tmp[k].second.second;
            c->sv_cols[0].push_back(a.chr); c->sv_cols[1].push_back(e.ha ? b1 : a.pos); c->sv_cols[2].push_back(e.ha ? a.pos + a.len : b1);
            c->sv_cols[3].push_back(b.chr); c->sv_cols[4].push_back(e.hb ? b2 : b.pos); c->sv_cols[5].push_back(e.hb ? b.pos + b.len : b2);
            c->sv_cols[6].push_back(e.w); c->sv_cols[7].push_back(s[k].first); c->sv_cols[8].push_back(s[k].second);
            c->sv_s1.push_back(e.ha); c->sv_s2.push_back(e.hb);
        }
    }
    return SQ_OK;
}

}  // namespace sq

using namespace sq;

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
    int rc = dev_create(c);
    if (rc) { std::fprintf(stderr, "libsquid_hip: %s\n", c->err.c_str()); dev_destroy(c); delete c; return rc; }
    *out = c;
    return SQ_OK;
}
void sq_destroy(sq_ctx* c) {
    if (!c) return;
    dev_destroy(c);
    delete c;
}
int sq_set_references(sq_ctx* c, int32_t n_ref, const int32_t* ref_len) {
    if (!c || n_ref < 0 || (n_ref && !ref_len)) return SQ_E_ARG;
    c->ref_len.assign(ref_len, ref_len + n_ref);
    return SQ_OK;
}
int sq_ingest_chimeric(sq_ctx* c, const sq_aln_batch* b) {
    if (!c || !b) return SQ_E_ARG;
    int rc = build_fragments(c, b);
    if (rc) return rc;
    c->frags0 = c->frags;
    return dev_upload_chim_names(c);
}
int sq_chim_contains(sq_ctx* c, const char* name, size_t len) {
    if (!c) return SQ_E_ARG;
    return c->chim_set.count(std::string(name, len)) ? 1 : 0;
}
int sq_ingest_concordant(sq_ctx* c, const sq_aln_batch* b) {
    if (!c || !b) return SQ_E_ARG;
    return dev_append_records(c, b);
}
int sq_ingest_concordant_bam(sq_ctx* c, const uint8_t* bam, size_t nbytes, const uint64_t* rec_off, int64_t n_rec) {
    if (!c || (n_rec && (!bam || !rec_off))) return SQ_E_ARG;
    int rc = dev_parse_append(c, bam, nbytes, (const unsigned long long*)rec_off, n_rec);
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
int sq_ingest_chimeric_file(sq_ctx* c, const char* path) {
    if (!c || !path) return SQ_E_ARG;
    ParseOpts o{c->P.phred_type, c->P.min_phred, c->P.max_lowphred_len, true, nullptr};
    HostBatch all;
    all.clear();
    bool got = false;
    int rc = parse_bam_file(path, o, (size_t)1 << 40, 1, c->err, [&](const HostBatch& hb) { all = hb; got = true; return 0; });
    if (rc) return rc;
    if (!got) return fail(c, SQ_E_EMPTYCHIM, "chimeric BAM holds no record");
    sq_aln_batch b;
    all.view(&b, true);
    return sq_ingest_chimeric(c, &b);
}
int sq_ingest_concordant_file(sq_ctx* c, const char* path, int32_t n_threads) {
    if (!c || !path) return SQ_E_ARG;
    if (!std::getenv("SQUID_HOST_PARSE")) {
        // default: the host only inflates BGZF and finds record boundaries; K0 parses the records on the GPU
        int rc = scan_bam_file(path, n_threads, c->err, [&](const uint8_t* bam, size_t nbytes, const unsigned long long* off, int64_t n) { return dev_parse_append(c, bam, nbytes, off, n); });
        dev_flush_timers(c);
        return rc;
    }
    ParseOpts o{c->P.phred_type, c->P.min_phred, c->P.max_lowphred_len, false, &c->chim_set};
    return parse_bam_file(path, o, (size_t)1 << 21, n_threads, c->err, [&](const HostBatch& hb) {
        sq_aln_batch b;
        hb.view(&b, false);
        return sq_ingest_concordant(c, &b);
    });
}
int sq_build_graph(sq_ctx* c) {
    if (!c) return SQ_E_ARG;
    if (c->ref_len.empty()) return fail(c, SQ_E_ARG, "sq_set_references first");
    if (c->read_len <= 0) return fail(c, SQ_E_ARG, "sq_ingest_chimeric first (ReadLen comes from the chimeric BAM)");
    int rc = build_graph(c);
    dev_flush_timers(c);
    return rc;
}
int sq_graph_view(sq_ctx* c, int32_t stage, sq_graph* g) {
    if (!c || !g || stage < 0 || stage > 5 || !c->graph_built) return SQ_E_ARG;
    c->snap[stage].view(g);
    return SQ_OK;
}
int sq_order(sq_ctx* c, sq_orders* o) {
    if (!c || !c->graph_built) return SQ_E_ARG;
    if (!c->ordered) { int rc = order_components(c); dev_flush_timers(c); if (rc) return rc; }
    if (o) { o->n_components = (int32_t)c->ord_off.size() - 1; o->comp_off = c->ord_off.data(); o->nodes = c->ord_nodes.data(); }
    return SQ_OK;
}
int sq_call_sv(sq_ctx* c, sq_sv_table* t) {
    if (!c || !c->graph_built) return SQ_E_ARG;
    int rc = call_sv(c);
    dev_flush_timers(c);
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
int sq_breakpoints(sq_ctx* c, sq_bp_table* t) {
    if (!c || !t || c->bp_off.empty()) return SQ_E_ARG;
    t->n_edges = (int32_t)c->bp_off.size() - 1;
    t->bp_off = c->bp_off.data(); t->bp1 = c->bp1.data(); t->bp2 = c->bp2.data(); t->sup1 = c->bsup1.data(); t->sup2 = c->bsup2.data();
    return SQ_OK;
}
int sq_get_timing(sq_ctx* c, sq_timing* t) {
    if (!c || !t) return SQ_E_ARG;
    t->n = (int32_t)c->timer.names.size();
    t->names = c->timer.names.data(); t->ms = c->timer.ms.data(); t->launches = c->timer.launches.data(); t->bytes = c->timer.bytes.data();
    return SQ_OK;
}
int sq_reset(sq_ctx* c) {
    if (!c) return SQ_E_ARG;
    c->frags = c->frags0;  // the graph stages trim the chimeric blocks in place, like the reference does
    c->nodes.clear(); c->edges.clear(); c->label.clear();
    c->graph_built = false; c->ordered = false;
    c->bp_off.clear();
    return SQ_OK;
}
int sq_get_counts(sq_ctx* c, sq_counts* k) {
    if (!c || !k) return SQ_E_ARG;
    *k = c->counts;
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
int sq_exchange_pack(sq_ctx* c, const void** buf, int64_t* nbytes) {
    (void)buf; (void)nbytes;
    return fail(c, SQ_E_ARG, "chromosome-sharded exchange is not implemented in this build");
}
int sq_exchange_unpack(sq_ctx* c, const void* g, const int64_t* n, int32_t w) {
    (void)g; (void)n; (void)w;
    return fail(c, SQ_E_ARG, "chromosome-sharded exchange is not implemented in this build");
}

}  // extern "C"
