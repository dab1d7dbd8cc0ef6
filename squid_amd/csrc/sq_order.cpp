// Per-component segment ordering (src/SegmentGraph.cpp:3236-3451,3763-3983).
//
// The reference builds one ILP per connected component and hands it to GLPK.  Here the model of appendix D is
// solved exactly and deterministically: components (or the bridge-free pieces MincutRecursion cuts them into)
// with at most 8 nodes go to the GPU as one batch (k_order_small, one component per workgroup); larger ones are
// solved on the host by branch-and-bound over orientations with the same canonical optimum:
//   max objective, then smallest orientation mask (bit i <=> local node i reversed), then the lexicographically
//   smallest left-to-right sequence.
// GLPK's pick among equal-valued optima and Boost's pick among equal min-cuts are not reproducible (neither
// library is available, nothing in the reference pins them); DESIGN.md lists this as "parity unpinned".
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <thread>
#include <cstring>

#include "sq_internal.h"

namespace sq {

namespace {

struct LEdge { int u, v; bool hu, hv; int w; };  // u < v (local indices)

struct Piece {  // a leaf of the min-cut recursion (or a whole small component)
    std::vector<int> ids;       // global node ids, ascending (local index = position)
    std::vector<LEdge> edges;   // incl. backbone edges
    std::vector<int> order;     // result: signed 1-based global ids, left to right
};

struct TreeNode {
    int left = -1, right = -1, piece = -1;
    Edge bridge{};
};

// ---- backbone + local edge list (SegmentGraph.cpp:3275-3286): a weight-1 tail->head edge between consecutive
// component nodes that no edge joins yet
void make_piece(const std::vector<int>& ids, const std::vector<Edge>& E, Piece& p) {
    p.ids = ids;
    const int n = (int)ids.size();
    auto local = [&](int g) { return (int)(std::lower_bound(ids.begin(), ids.end(), g) - ids.begin()); };
    std::vector<char> joined(n, 0);
    for (const Edge& e : E) {
        int u = local(e.a), v = local(e.b);
        p.edges.push_back(LEdge{u, v, (bool)e.ha, (bool)e.hb, e.w});
        if (v == u + 1) joined[u] = 1;
    }
    for (int k = 0; k + 1 < n; ++k)
        if (!joined[k]) p.edges.push_back(LEdge{k, k + 1, false, true, 1});
}

// ---- exact host solver for 9..exact_max nodes
struct HostSolver {
    int n;
    const std::vector<LEdge>& E;
    long best = -1;
    unsigned bestmask = 0;
    std::vector<int> bestorder;
    long n_nodes = 0, n_leaves = 0, n_cyclic = 0;  // search statistics (SQUID_ORDER_PROF)
    HostSolver(int n, const std::vector<LEdge>& E) : n(n), E(E) {}

    static bool compat(const LEdge& e, unsigned mask, bool& ufirst) {
        bool yu = !((mask >> e.u) & 1), yv = !((mask >> e.v) & 1);
        if (e.hu != e.hv) {  // tail->head or head->tail: equal orientations
            if (yu != yv) return false;
            ufirst = e.hv ? yu : !yu;
        } else {  // tail-tail or head-head: opposite orientations
            if (yu == yv) return false;
            ufirst = e.hu ? yv : yu;
        }
        return true;
    }
    long bound(unsigned mask, int k) const {  // nodes > k fixed, nodes <= k free
        long ub = 0;
        for (const LEdge& e : E) {
            bool uf;
            if (e.u <= k || compat(e, mask, uf)) ub += e.w;
        }
        return ub;
    }
    // Linear ordering for one orientation.  Arcs between different strongly connected components of the arc
    // graph can all be satisfied, so only the non-trivial components need the subset DP; the canonical
    // (lexicographically smallest optimal) sequence is then built greedily: the smallest node whose cross-component
    // predecessors are placed and whose own component can still reach its optimum.
    struct SccDP {
        std::vector<int> mem;      // members, ascending
        std::vector<long> h;       // h[S] = best weight still obtainable inside the component once S is placed
        std::vector<int> a;        // local arc weights s x s
        unsigned placed = 0;
        // gain of appending v after the set S: read from two half tables (low / high members of S) instead of a sum over S
        std::vector<long> glo, ghi;
        int lo_bits = 0;
        long gain(unsigned S, int v) const {
            const int s = (int)mem.size();
            return glo[(size_t)(S & ((1u << lo_bits) - 1)) * s + v] + ghi[(size_t)(S >> lo_bits) * s + v];
        }
        void solve() {
            const int s = (int)mem.size();
            const unsigned full = (1u << s) - 1;
            lo_bits = s / 2;
            const int hi_bits = s - lo_bits;
            glo.assign(((size_t)1 << lo_bits) * s, 0); ghi.assign(((size_t)1 << hi_bits) * s, 0);
            for (unsigned x = 1; x < (1u << lo_bits); ++x) { const int u = __builtin_ctz(x); for (int v = 0; v < s; ++v) glo[(size_t)x * s + v] = glo[(size_t)(x & (x - 1)) * s + v] + a[u * s + v]; }
            for (unsigned x = 1; x < (1u << hi_bits); ++x) { const int u = __builtin_ctz(x) + lo_bits; for (int v = 0; v < s; ++v) ghi[(size_t)x * s + v] = ghi[(size_t)(x & (x - 1)) * s + v] + a[u * s + v]; }
            h.assign((size_t)full + 1, 0);
            for (unsigned S = full; S-- > 0;) {
                long b = -1;
                const long* gl = &glo[(size_t)(S & ((1u << lo_bits) - 1)) * s];
                const long* gh = &ghi[(size_t)(S >> lo_bits) * s];
                for (unsigned m = ~S & full; m; m &= m - 1) { const int v = __builtin_ctz(m); b = std::max(b, gl[v] + gh[v] + h[S | (1u << v)]); }
                h[S] = b;
            }
        }
    };
    // all compatible edges satisfied <=> the precedence arcs are acyclic: Kahn on bit masks, smallest index first
    // (= the lexicographically smallest optimal sequence), no allocation
    bool leaf_acyclic(unsigned mask, long ub) {
        unsigned in[32];
        for (int x = 0; x < n; ++x) in[x] = 0;
        for (const LEdge& e : E) {
            bool uf;
            if (!compat(e, mask, uf)) continue;
            if (uf) in[e.v] |= 1u << e.u; else in[e.u] |= 1u << e.v;
        }
        unsigned remaining = n == 32 ? ~0u : (1u << n) - 1;
        int ord[32];
        for (int p = 0; p < n; ++p) {
            int v = -1;
            for (unsigned m = remaining; m; m &= m - 1) { int cnd = __builtin_ctz(m); if (!(in[cnd] & remaining)) { v = cnd; break; } }
            if (v < 0) return false;
            remaining &= ~(1u << v);
            ord[p] = v;
        }
        if (ub > best) { best = ub; bestmask = mask; bestorder.assign(ord, ord + n); }
        return true;
    }
    void leaf(unsigned mask) {
        ++n_cyclic;
        std::vector<int> a((size_t)n * n, 0);
        for (const LEdge& e : E) {
            bool uf;
            if (!compat(e, mask, uf)) continue;
            if (uf) a[e.u * n + e.v] += e.w; else a[e.v * n + e.u] += e.w;
        }
        // reachability closure on bitmasks -> strongly connected components
        std::vector<unsigned> reach(n, 0);
        for (int x = 0; x < n; ++x) { reach[x] = 1u << x; for (int y = 0; y < n; ++y) if (a[x * n + y] > 0) reach[x] |= 1u << y; }
        for (int k = 0; k < n; ++k) for (int x = 0; x < n; ++x) if ((reach[x] >> k) & 1) reach[x] |= reach[k];
        std::vector<int> comp(n, -1);
        std::vector<SccDP> sccs;
        long val = 0;
        for (int x = 0; x < n; ++x) {
            if (comp[x] >= 0) continue;
            SccDP d;
            for (int y = x; y < n; ++y) if (((reach[x] >> y) & 1) && ((reach[y] >> x) & 1)) { comp[y] = (int)sccs.size(); d.mem.push_back(y); }
            sccs.push_back(std::move(d));
        }
        for (int x = 0; x < n; ++x) for (int y = 0; y < n; ++y) if (comp[x] != comp[y]) val += a[x * n + y];
        std::vector<int> localidx(n, 0);
        for (SccDP& d : sccs) {
            const int s = (int)d.mem.size();
            for (int i = 0; i < s; ++i) localidx[d.mem[i]] = i;
            if (s == 1) continue;
            d.a.assign((size_t)s * s, 0);
            for (int i = 0; i < s; ++i) for (int j = 0; j < s; ++j) d.a[i * s + j] = a[d.mem[i] * n + d.mem[j]];
            d.solve();
            val += d.h[0];
        }
        if (!(val > best)) return;
        std::vector<int> order;
        unsigned done = 0;
        for (int p = 0; p < n; ++p)
            for (int v = 0; v < n; ++v) {
                if ((done >> v) & 1) continue;
                bool ok = true;
                for (int x = 0; x < n && ok; ++x) if (a[x * n + v] > 0 && comp[x] != comp[v] && !((done >> x) & 1)) ok = false;
                if (!ok) continue;
                SccDP& d = sccs[comp[v]];
                if (d.mem.size() > 1) {
                    int lv = localidx[v];
                    if (d.gain(d.placed, lv) + d.h[d.placed | (1u << lv)] != d.h[d.placed]) continue;
                    d.placed |= 1u << lv;
                }
                order.push_back(v);
                done |= 1u << v;
                break;
            }
        best = val; bestmask = mask; bestorder = order;
    }
    void run() {
        // depth-first over nodes n-1..0, forward before reversed; a branch must be able to beat the incumbent strictly
        struct Fr { unsigned mask; int k; };
        std::vector<Fr> st;
        st.push_back(Fr{0u, n - 1});
        while (!st.empty()) {
            Fr f = st.back();
            st.pop_back();
            ++n_nodes;
            const long ub = bound(f.mask, f.k);
            if (ub <= best) continue;
            if (f.k < 0) { ++n_leaves; if (!leaf_acyclic(f.mask, ub)) leaf(f.mask); continue; }
            // reversing every node and the whole sequence satisfies the same edges, so the optimum with the smallest mask
            // keeps the last node forward: the other half of the tree is never needed
            if (f.k != n - 1) st.push_back(Fr{f.mask | (1u << f.k), f.k - 1});  // explored second
            st.push_back(Fr{f.mask, f.k - 1});                                  // explored first
        }
    }
};

// Replacement for boost::stoer_wagner_min_cut with unit weights (SegmentGraph.cpp:3316-3325): the reference only asks
// "is the min cut 1?" and then uses the bipartition.  A cut of 1 is a bridge of the multigraph.  Rule (shared with
// the oracle): take the most balanced bridge (minimise |n - 2*side|), ties by position in the
// sorted component edge list; no bridge => "cut > 1".  Bridges by an iterative low-link DFS, O(n + m).
bool bridge_split(int n, const std::vector<std::pair<int, int>>& edges, std::vector<char>& side) {
    const int m = (int)edges.size();
    std::vector<int> head(n + 1, 0), adj(2 * (size_t)m), aid(2 * (size_t)m);
    for (auto& e : edges) { head[e.first + 1]++; head[e.second + 1]++; }
    for (int i = 0; i < n; ++i) head[i + 1] += head[i];
    {
        std::vector<int> fill(head.begin(), head.end() - 1);
        for (int i = 0; i < m; ++i) {
            adj[fill[edges[i].first]] = edges[i].second; aid[fill[edges[i].first]++] = i;
            adj[fill[edges[i].second]] = edges[i].first; aid[fill[edges[i].second]++] = i;
        }
    }
    std::vector<int> disc(n, -1), low(n, 0), parent_edge(n, -1), sub(n, 1), it(n, 0), order;
    order.reserve(n);
    std::vector<int> stack(1, 0);
    int timer = 0;
    disc[0] = low[0] = timer++;
    int best_edge = -1, best_bal = -1, best_child = -1;
    std::vector<int> tin(n, 0), tout(n, 0);
    tin[0] = 0;
    while (!stack.empty()) {
        int x = stack.back();
        if (it[x] < head[x + 1] - head[x]) {
            int k = head[x] + it[x]++;
            int y = adj[k], id = aid[k];
            if (id == parent_edge[x]) continue;          // the tree edge itself; a PARALLEL edge has another id and counts as a back edge
            if (disc[y] < 0) { disc[y] = low[y] = timer++; parent_edge[y] = id; stack.push_back(y); }
            else low[x] = std::min(low[x], disc[y]);
        } else {
            stack.pop_back();
            if (!stack.empty()) {
                int p = stack.back();
                low[p] = std::min(low[p], low[x]);
                sub[p] += sub[x];
                if (low[x] > disc[p]) {  // bridge p - x
                    int bal = std::abs(n - 2 * sub[x]);
                    int id = parent_edge[x];
                    if (best_bal < 0 || bal < best_bal || (bal == best_bal && id < best_edge)) { best_bal = bal; best_edge = id; best_child = x; }
                }
            }
        }
    }
    if (best_edge < 0) return false;
    // side = subtree of best_child: nodes discovered in [disc[child], disc[child] + sub[child])
    side.assign(n, 0);
    for (int v = 0; v < n; ++v) if (disc[v] >= disc[best_child] && disc[v] < disc[best_child] + sub[best_child]) side[v] = 1;
    return true;
}

struct Builder {
    std::vector<TreeNode> tree;
    std::vector<Piece> pieces;
    // MincutRecursion (SegmentGraph.cpp:3264-3451): returns tree node index
    int build(const std::vector<int>& ids, const std::vector<Edge>& E) {
        const int n = (int)ids.size();
        int me = (int)tree.size();
        tree.push_back(TreeNode());
        bool whole = n < 20;
        std::vector<char> side;
        if (!whole) {
            auto local = [&](int g) { return (int)(std::lower_bound(ids.begin(), ids.end(), g) - ids.begin()); };
            std::vector<std::pair<int, int>> le;
            for (const Edge& e : E) le.push_back(std::make_pair(local(e.a), local(e.b)));
            if (!bridge_split(n, le, side)) whole = true;
        }
        if (whole || n == 1) {
            Piece p;
            if (n == 1) { p.ids = ids; p.order.assign(1, ids[0] + 1); }
            else make_piece(ids, E, p);
            tree[me].piece = (int)pieces.size();
            pieces.push_back(std::move(p));
            return me;
        }
        std::vector<int> ids1, ids2;
        for (int k = 0; k < n; ++k) (side[k] ? ids1 : ids2).push_back(ids[k]);
        std::vector<Edge> e1, e2;
        Edge mid{};
        for (const Edge& e : E) {
            bool s1 = std::binary_search(ids1.begin(), ids1.end(), e.a), s2 = std::binary_search(ids1.begin(), ids1.end(), e.b);
            if (s1 && s2) e1.push_back(e);
            else if (!s1 && !s2) e2.push_back(e);
            else mid = e;
        }
        int l = build(ids1, e1), r = build(ids2, e2);
        tree[me].left = l; tree[me].right = r; tree[me].bridge = mid;
        return me;
    }
    // join two ordered halves over the bridge edge (SegmentGraph.cpp:3393-3448)
    std::vector<int> combine(int t) {
        const TreeNode& tn = tree[t];
        if (tn.piece >= 0) return pieces[tn.piece].order;
        std::vector<int> A = combine(tn.left), B = combine(tn.right);
        auto scan = [&](const std::vector<int>& X, int& median, bool& positive, bool& head) {
            std::vector<int> ab;
            positive = false; head = false;
            for (int x : X) {
                ab.push_back(std::abs(x));
                if (std::abs(x) == tn.bridge.a + 1) { positive = x > 0; head = tn.bridge.ha; }
                else if (std::abs(x) == tn.bridge.b + 1) { positive = x > 0; head = tn.bridge.hb; }
            }
            std::sort(ab.begin(), ab.end());
            median = ab[(ab.size() - 1) / 2];
        };
        int m1, m2;
        bool p1, h1, p2, h2;
        scan(A, m1, p1, h1);
        scan(B, m2, p2, h2);
        auto flip = [](std::vector<int>& X) { std::reverse(X.begin(), X.end()); for (int& x : X) x = -x; };
        if (m1 < m2) {
            if (p1 == h1) flip(A);
            if (p2 != h2) flip(B);
            A.insert(A.end(), B.begin(), B.end());
            return A;
        }
        if (p2 == h2) flip(B);
        if (p1 != h1) flip(A);
        B.insert(B.end(), A.begin(), A.end());
        return B;
    }
};

}  // namespace

int order_components(sq_ctx* c) {
    struct W { sq_ctx* c; std::chrono::steady_clock::time_point t0; ~W() { c->timer.add("wall_order", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); } } wall{c, std::chrono::steady_clock::now()};
    const int n = (int)c->nodes.size();
    int ncomp = 0;
    for (int l : c->label) ncomp = std::max(ncomp, l + 1);
    // bucket nodes and edges per component (the reference rescans everything per component, :3248-3253)
    std::vector<std::vector<int>> cn(ncomp);
    std::vector<std::vector<Edge>> ce(ncomp);
    for (int i = 0; i < n; ++i) cn[c->label[i]].push_back(i);
    for (const Edge& e : c->edges) if (e.a != e.b) ce[c->label[e.a]].push_back(e);
    Builder B;
    std::vector<int> roots(ncomp);
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < ncomp; ++k) roots[k] = B.build(cn[k], ce[k]);
    c->timer.add("host_mincut_tree", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    // ---- leaves: <= 8 nodes on the GPU in one batch, the rest on the host
    const int GPU_NMAX = 8, EXACT_MAX = 26;
    std::vector<SmallProblem> probs;
    std::vector<int32_t> edges5;
    std::vector<int> gpu_piece;
    for (size_t pi = 0; pi < B.pieces.size(); ++pi) {
        Piece& p = B.pieces[pi];
        const int pn = (int)p.ids.size();
        if (pn == 1) continue;
        if (pn <= GPU_NMAX) {
            SmallProblem sp{pn, (int)(edges5.size() / 5), (int)p.edges.size()};
            for (const LEdge& e : p.edges) { edges5.push_back(e.u); edges5.push_back(e.v); edges5.push_back(e.hu); edges5.push_back(e.hv); edges5.push_back(e.w); }
            probs.push_back(sp);
            gpu_piece.push_back((int)pi);
        }
    }
    std::vector<int32_t> gmask, gorder;
    int rc = dev_order_small(c, probs, edges5, gmask, gorder, GPU_NMAX);
    if (rc) return rc;
    for (size_t q = 0; q < gpu_piece.size(); ++q) {
        Piece& p = B.pieces[gpu_piece[q]];
        const int pn = (int)p.ids.size();
        p.order.resize(pn);
        for (int pos = 0; pos < pn; ++pos) {
            int l = gorder[q * 8 + pos];
            p.order[pos] = ((gmask[q] >> l) & 1) ? -(p.ids[l] + 1) : (p.ids[l] + 1);
        }
    }
    t0 = std::chrono::steady_clock::now();
    std::vector<int> large;
    for (size_t pi = 0; pi < B.pieces.size(); ++pi) {
        Piece& p = B.pieces[pi];
        const int pn = (int)p.ids.size();
        if (pn <= GPU_NMAX) continue;
        p.order.resize(pn);
        if (pn > EXACT_MAX) {
            // what the reference keeps when glp_intopt gives up (:3287-3292,3984): identity order, all forward
            for (int k = 0; k < pn; ++k) p.order[k] = p.ids[k] + 1;
            continue;
        }
        large.push_back((int)pi);
    }
    static const bool order_prof = std::getenv("SQUID_ORDER_PROF") != nullptr;
    auto solve = [&](int pi) {
        Piece& p = B.pieces[pi];
        const int pn = (int)p.ids.size();
        HostSolver hs(pn, p.edges);
        auto ts = std::chrono::steady_clock::now();
        hs.run();
        if (order_prof) std::fprintf(stderr, "[order] piece n=%d m=%zu nodes=%ld leaves=%ld cyclic=%ld best=%ld  %.3f ms\n", pn, p.edges.size(), hs.n_nodes, hs.n_leaves, hs.n_cyclic, hs.best,
                                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts).count());
        for (int pos = 0; pos < pn; ++pos) {
            int l = hs.bestorder[pos];
            p.order[pos] = ((hs.bestmask >> l) & 1) ? -(p.ids[l] + 1) : (p.ids[l] + 1);
        }
    };
    // the pieces are independent: solve them on a few host threads (biggest first)
    std::sort(large.begin(), large.end(), [&](int x, int y) { return B.pieces[x].ids.size() > B.pieces[y].ids.size(); });
    // (a handful of pieces is done before the threads would have started)
    const int nthr = large.size() <= 8 ? 1 : (int)std::min<size_t>(std::min<size_t>(large.size() / 4, 16), std::max(1u, std::thread::hardware_concurrency()));
    if (nthr <= 1) for (int pi : large) solve(pi);
    else {
        std::atomic<size_t> next{0};
        std::vector<std::thread> pool;
        for (int t = 0; t < nthr; ++t) pool.emplace_back([&]() { for (size_t i; (i = next.fetch_add(1)) < large.size();) solve(large[i]); });
        for (auto& th : pool) th.join();
    }
    c->timer.add("host_order_large", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    c->ord_off.assign(1, 0);
    c->ord_nodes.clear();
    for (int k = 0; k < ncomp; ++k) {
        std::vector<int> o = B.combine(roots[k]);
        c->ord_nodes.insert(c->ord_nodes.end(), o.begin(), o.end());
        c->ord_off.push_back((int32_t)c->ord_nodes.size());
    }
    c->ordered = true;
    return SQ_OK;
}

}  // namespace sq
