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
#include <future>
#include <deque>
#include <memory>
#include <ext/pb_ds/assoc_container.hpp>
#include <ext/pb_ds/tree_policy.hpp>
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

// ---- exact host solver: components the GPU kernels do not take (more than 19 nodes, or their capacities exceeded), up to 128
// nodes.  Orientation masks are 128-bit; the search is bounded by a node budget (the reference gives GLPK 300 s, :3964, and
// keeps the identity order when it fails, :3287-3292,3984 -- `failed` reports that case).
typedef unsigned __int128 Mask;
inline int mask_ctz(Mask m) { const uint64_t lo = (uint64_t)m; return lo ? __builtin_ctzll(lo) : 64 + __builtin_ctzll((uint64_t)(m >> 64)); }
inline bool mask_bit(Mask m, int i) { return (bool)((m >> i) & 1); }
constexpr int HOST_NMAX = 128, SCC_MAX = 22;
struct HostSolver {
    int n;
    const std::vector<LEdge>& E;
    long best = -1;
    Mask bestmask = 0;
    std::vector<int> bestorder;
    long n_nodes = 0, n_leaves = 0, n_cyclic = 0;  // search statistics (SQUID_ORDER_PROF)
    long budget;
    bool failed = false;
    HostSolver(int n, const std::vector<LEdge>& E, long budget = 20000000L) : n(n), E(E), budget(budget) {}

    static bool compat(const LEdge& e, Mask mask, bool& ufirst) {
        bool yu = !mask_bit(mask, e.u), yv = !mask_bit(mask, e.v);
        if (e.hu != e.hv) {  // tail->head or head->tail: equal orientations
            if (yu != yv) return false;
            ufirst = e.hv ? yu : !yu;
        } else {  // tail-tail or head-head: opposite orientations
            if (yu == yv) return false;
            ufirst = e.hu ? yv : yu;
        }
        return true;
    }
    // Optimistic value with nodes > k fixed and nodes <= k free: an edge between two free nodes counts, an edge between two fixed
    // nodes counts when it is compatible, and an edge from a free node u to a fixed node is compatible with exactly one of u's
    // two orientations -- u gets the better of its two sums.  (Much tighter than "every edge with a free end counts" once a
    // component has a few dozen nodes: conflicting evidence around a node is priced as soon as its neighbours are fixed.)
    mutable std::vector<long> gain_f, gain_r;
    long bound(Mask mask, int k) const {
        long ub = 0;
        if (k >= 0) { gain_f.assign((size_t)k + 1, 0); gain_r.assign((size_t)k + 1, 0); }
        for (const LEdge& e : E) {
            if (e.v <= k) { ub += e.w; continue; }   // u < v <= k: both free
            if (e.u <= k) {                          // u free, v fixed
                const bool yv = !mask_bit(mask, e.v);
                const bool want_yu = (e.hu != e.hv) ? yv : !yv;  // the orientation of u under which the edge is compatible
                (want_yu ? gain_f : gain_r)[(size_t)e.u] += e.w;
                continue;
            }
            bool uf;
            if (compat(e, mask, uf)) ub += e.w;
        }
        for (int u = 0; u <= k; ++u) ub += std::max(gain_f[(size_t)u], gain_r[(size_t)u]);
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
    bool leaf_acyclic(Mask mask, long ub) {
        Mask in[HOST_NMAX];
        for (int x = 0; x < n; ++x) in[x] = 0;
        for (const LEdge& e : E) {
            bool uf;
            if (!compat(e, mask, uf)) continue;
            if (uf) in[e.v] |= (Mask)1 << e.u; else in[e.u] |= (Mask)1 << e.v;
        }
        Mask remaining = n == HOST_NMAX ? ~(Mask)0 : (((Mask)1 << n) - 1);
        int ord[HOST_NMAX];
        for (int p = 0; p < n; ++p) {
            int v = -1;
            for (Mask m = remaining; m; m &= m - 1) { int cnd = mask_ctz(m); if (!(in[cnd] & remaining)) { v = cnd; break; } }
            if (v < 0) return false;
            remaining &= ~((Mask)1 << v);
            ord[p] = v;
        }
        if (ub > best) { best = ub; bestmask = mask; bestorder.assign(ord, ord + n); }
        return true;
    }
    void leaf(Mask mask) {
        ++n_cyclic;
        std::vector<int> a((size_t)n * n, 0);
        for (const LEdge& e : E) {
            bool uf;
            if (!compat(e, mask, uf)) continue;
            if (uf) a[e.u * n + e.v] += e.w; else a[e.v * n + e.u] += e.w;
        }
        // reachability closure on bitmasks -> strongly connected components
        std::vector<Mask> reach(n, 0);
        for (int x = 0; x < n; ++x) { reach[x] = (Mask)1 << x; for (int y = 0; y < n; ++y) if (a[x * n + y] > 0) reach[x] |= (Mask)1 << y; }
        for (int k = 0; k < n; ++k) for (int x = 0; x < n; ++x) if (mask_bit(reach[x], k)) reach[x] |= reach[k];
        std::vector<int> comp(n, -1);
        std::vector<SccDP> sccs;
        long val = 0;
        for (int x = 0; x < n; ++x) {
            if (comp[x] >= 0) continue;
            SccDP d;
            for (int y = x; y < n; ++y) if (mask_bit(reach[x], y) && mask_bit(reach[y], x)) { comp[y] = (int)sccs.size(); d.mem.push_back(y); }
            if ((int)d.mem.size() > SCC_MAX) {
                // its subset table would not fit.  One arc of the component at least is violated: the orientation is worth at most
                // ub - (lightest arc inside), which settles it when the incumbent is at least that good; otherwise give up
                long ub = 0;
                int lightest = 0;
                for (int i = 0; i < n * n; ++i) ub += a[i];
                for (int i : d.mem) for (int j : d.mem) if (a[i * n + j] > 0 && (lightest == 0 || a[i * n + j] < lightest)) lightest = a[i * n + j];
                if (ub - lightest <= best) return;
                failed = true;
                return;
            }
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
            n_nodes += (long)1 << s;  // (the tables count against the budget too)
            val += d.h[0];
        }
        if (!(val > best)) return;
        std::vector<int> order;
        Mask done = 0;
        for (int p = 0; p < n; ++p)
            for (int v = 0; v < n; ++v) {
                if (mask_bit(done, v)) continue;
                bool ok = true;
                for (int x = 0; x < n && ok; ++x) if (a[x * n + v] > 0 && comp[x] != comp[v] && !mask_bit(done, x)) ok = false;
                if (!ok) continue;
                SccDP& d = sccs[comp[v]];
                if (d.mem.size() > 1) {
                    int lv = localidx[v];
                    if (d.gain(d.placed, lv) + d.h[d.placed | (1u << lv)] != d.h[d.placed]) continue;
                    d.placed |= 1u << lv;
                }
                order.push_back(v);
                done |= (Mask)1 << v;
                break;
            }
        best = val; bestmask = mask; bestorder = order;
    }
    // A feasible value to start from (round 6).  The orientation search below is exact with any incumbent, but how much of its tree it
    // walks hangs on the incumbent it starts with: the 75-node bridge-free component of the --bwa bench sample took more than the budget
    // from nothing and takes 7 000 nodes from its optimum.  These graphs are a backbone of concordant edges plus a few heavy
    // discordant ones, and nearly all of their weight can be satisfied at once; so: take the edges by descending weight and keep an
    // edge when it agrees with the ones kept so far -- orientations (a parity per node against the root of its tree of kept edges:
    // tail->head / head->tail edges want equal orientations, tail-tail / head-head opposite ones) and precedence (the edge's arc must
    // not close a directed cycle among the kept arcs).  A set of edges kept this way IS satisfiable: a two-colouring of every tree
    // gives the orientations, a topological order of the arcs the sequence.  Its weight is a lower bound of the optimum.
    long greedy_lower_bound() const {
        std::vector<int> idx(E.size());
        for (size_t i = 0; i < E.size(); ++i) idx[i] = (int)i;
        std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return E[(size_t)x].w > E[(size_t)y].w; });
        std::vector<int> par((size_t)n), rel((size_t)n, 0), size((size_t)n, 1);
        for (int i = 0; i < n; ++i) par[(size_t)i] = i;
        auto find = [&](int x, int& p) { p = 0; while (par[(size_t)x] != x) { p ^= rel[(size_t)x]; x = par[(size_t)x]; } return x; };
        std::vector<std::vector<int>> adj((size_t)n);  // kept edges at a node
        auto arc = [&](const LEdge& e, int& from, int& to) {  // the kept edge's arc under the parities as they stand (all of one tree: against the same root)
            int pu, pv;
            (void)find(e.u, pu); (void)find(e.v, pv);
            const bool yu = !pu, yv = !pv;
            const bool ufirst = e.hu != e.hv ? (e.hv ? yu : !yu) : (e.hu ? yv : yu);
            from = ufirst ? e.u : e.v; to = ufirst ? e.v : e.u;
        };
        std::vector<int> stack;
        std::vector<char> seen;
        auto reaches = [&](int a, int b) {
            seen.assign((size_t)n, 0); stack.assign(1, a); seen[(size_t)a] = 1;
            while (!stack.empty()) {
                const int x = stack.back(); stack.pop_back();
                if (x == b) return true;
                for (int ei : adj[(size_t)x]) { int f, t; arc(E[(size_t)ei], f, t); if (f == x && !seen[(size_t)t]) { seen[(size_t)t] = 1; stack.push_back(t); } }
            }
            return false;
        };
        long val = 0;
        for (int ei : idx) {
            const LEdge& e = E[(size_t)ei];
            int pu, pv;
            int ru = find(e.u, pu), rv = find(e.v, pv);
            const int want = e.hu != e.hv ? 0 : 1;
            if (ru == rv) {
                if ((pu ^ pv) != want) continue;
                int f, t; arc(e, f, t);
                if (reaches(t, f)) continue;
            } else {
                if (size[(size_t)ru] < size[(size_t)rv]) { std::swap(ru, rv); std::swap(pu, pv); }
                par[(size_t)rv] = ru; rel[(size_t)rv] = pu ^ pv ^ want; size[(size_t)ru] += size[(size_t)rv];
            }
            adj[(size_t)e.u].push_back(ei); adj[(size_t)e.v].push_back(ei);
            val += e.w;
        }
        return val;
    }
    void run() {
        // depth-first over nodes n-1..0, forward before reversed; a branch must be able to beat the incumbent strictly
        // (the incumbent starts one below the greedy set's weight: the first orientation that reaches it is found, not assumed)
        static const bool seed = std::getenv("SQUID_ORDER_NO_SEED") == nullptr;
        if (seed && n > 12) best = greedy_lower_bound() - 1;
        struct Fr { Mask mask; int k; };
        std::vector<Fr> st;
        st.push_back(Fr{0, n - 1});
        while (!st.empty()) {
            Fr f = st.back();
            st.pop_back();
            if (++n_nodes > budget) failed = true;
            if (failed) { best = -1; return; }
            const long ub = bound(f.mask, f.k);
            if (ub <= best) continue;
            if (f.k < 0) { ++n_leaves; if (!leaf_acyclic(f.mask, ub)) leaf(f.mask); continue; }
            // reversing every node and the whole sequence satisfies the same edges, so the optimum with the smallest mask
            // keeps the last node forward: the other half of the tree is never needed
            if (f.k != n - 1) st.push_back(Fr{f.mask | ((Mask)1 << f.k), f.k - 1});  // explored second
            st.push_back(Fr{f.mask, f.k - 1});                                        // explored first
        }
        if (failed || bestorder.empty()) { failed = true; best = -1; }
    }
};

// Replacement for boost::stoer_wagner_min_cut with unit weights (SegmentGraph.cpp:3316-3325): the reference only asks
// "is the min cut 1?" and then uses the bipartition.  A cut of 1 is a bridge of the multigraph.  Rule (shared with the
// oracle): take the most balanced bridge (minimise |n - 2*side|), ties by position in the
// sorted component edge list; no bridge => "cut > 1".
// MincutRecursion (SegmentGraph.cpp:3264-3451) without redoing the bridge search at every level: the bridges of a part
// of the component are exactly the component's bridges inside it (splitting at a bridge breaks no cycle), so the
// component is contracted ONCE into its tree of 2-edge-connected blobs (low-link DFS) and the recursion runs on that
// tree -- per split one pass over the blobs of the part instead of one over its nodes and edges (the dense-graph config
// peels thousands of small trees off a 20 000-node core: 1.9 s -> a few ms).
struct Builder {
    std::vector<TreeNode> tree;
    std::vector<Piece> pieces;

    int leaf_node(Piece&& p) {
        TreeNode t;
        t.piece = (int)pieces.size();
        pieces.push_back(std::move(p));
        tree.push_back(t);
        return (int)tree.size() - 1;
    }
    int build(const std::vector<int>& ids, const std::vector<Edge>& E) {
        const int n = (int)ids.size(), m = (int)E.size();
        if (n == 1) { Piece p; p.ids = ids; p.order.assign(1, ids[0] + 1); return leaf_node(std::move(p)); }
        if (n < 20) { Piece p; make_piece(ids, E, p); return leaf_node(std::move(p)); }
        auto local = [&](int g) { return (int)(std::lower_bound(ids.begin(), ids.end(), g) - ids.begin()); };
        std::vector<int> ea(m), eb(m);
        for (int i = 0; i < m; ++i) { ea[i] = local(E[i].a); eb[i] = local(E[i].b); }
        // ---- bridges (a parallel edge has another id and counts as a back edge) and blobs
        std::vector<int> head(n + 1, 0), adj(2 * (size_t)m), aid(2 * (size_t)m);
        for (int i = 0; i < m; ++i) { head[ea[i] + 1]++; head[eb[i] + 1]++; }
        for (int i = 0; i < n; ++i) head[i + 1] += head[i];
        {
            std::vector<int> fill(head.begin(), head.end() - 1);
            for (int i = 0; i < m; ++i) { adj[fill[ea[i]]] = eb[i]; aid[fill[ea[i]]++] = i; adj[fill[eb[i]]] = ea[i]; aid[fill[eb[i]]++] = i; }
        }
        std::vector<int> disc(n, -1), low(n, 0), parent_edge(n, -1), parent(n, -1), it(n, 0), order;
        std::vector<char> is_bridge(m, 0);
        order.reserve(n);
        std::vector<int> stack(1, 0);
        int timer = 0;
        disc[0] = low[0] = timer++; order.push_back(0);
        while (!stack.empty()) {
            int x = stack.back();
            if (it[x] < head[x + 1] - head[x]) {
                int k = head[x] + it[x]++;
                int y = adj[k], id = aid[k];
                if (id == parent_edge[x]) continue;
                if (disc[y] < 0) { disc[y] = low[y] = timer++; parent_edge[y] = id; parent[y] = x; order.push_back(y); stack.push_back(y); }
                else low[x] = std::min(low[x], disc[y]);
            } else {
                stack.pop_back();
                if (!stack.empty()) {
                    int p = stack.back();
                    low[p] = std::min(low[p], low[x]);
                    if (low[x] > disc[p]) is_bridge[parent_edge[x]] = 1;
                }
            }
        }
        std::vector<int> blob(n, 0);
        int nblob = 1;
        for (int k = 1; k < n; ++k) { int x = order[k]; blob[x] = is_bridge[parent_edge[x]] ? nblob++ : blob[parent[x]]; }
        std::vector<int> weight(nblob, 0);
        for (int x = 0; x < n; ++x) weight[blob[x]]++;
        // ---- blob tree (edges = bridges, keyed by their position in the component's sorted edge list)
        std::vector<int> th(nblob + 1, 0), tadj, tid;
        for (int i = 0; i < m; ++i) if (is_bridge[i]) { th[blob[ea[i]] + 1]++; th[blob[eb[i]] + 1]++; }
        for (int i = 0; i < nblob; ++i) th[i + 1] += th[i];
        tadj.resize(th[nblob]); tid.resize(th[nblob]);
        {
            std::vector<int> fill(th.begin(), th.end() - 1);
            for (int i = 0; i < m; ++i) if (is_bridge[i]) { int u = blob[ea[i]], v = blob[eb[i]]; tadj[fill[u]] = v; tid[fill[u]++] = i; tadj[fill[v]] = u; tid[fill[v]++] = i; }
        }
        // ---- the recursion on the blob tree.  A part = the blobs reachable from `root` over bridges not cut yet; it is split at its
        // most balanced bridge (minimal |W - 2 * weight of the far side|, ties by the bridge's position in the edge list).
        std::vector<char> cut(m, 0);
        std::vector<int> part_of(nblob, -1);           // final: leaf part id of every blob
        const int top = (int)tree.size();
        tree.push_back(TreeNode());
        std::vector<int> leaf_tnode;                   // leaf part id -> tree node
        const bool simple_recursion = std::getenv("SQUID_MINCUT_SIMPLE") != nullptr;  // the one-traversal-per-split form (cross-check)
        if (simple_recursion) {
            struct Job { int root, tnode; };
            std::vector<Job> jobs(1, Job{0, top});
            std::vector<int> nodes_of, par_b, par_e, sub;  // scratch of one traversal
            par_b.assign(nblob, -1); par_e.assign(nblob, -1); sub.assign(nblob, 0);
            while (!jobs.empty()) {
                const Job jb = jobs.back();
                jobs.pop_back();
                nodes_of.clear();
                nodes_of.push_back(jb.root); par_b[jb.root] = -1; par_e[jb.root] = -1;
                for (size_t q = 0; q < nodes_of.size(); ++q) {  // BFS order: parents before children
                    int u = nodes_of[q];
                    for (int k = th[u]; k < th[u + 1]; ++k) if (!cut[tid[k]] && tid[k] != par_e[u]) { int v = tadj[k]; par_b[v] = u; par_e[v] = tid[k]; nodes_of.push_back(v); }
                }
                long W = 0;
                for (int u : nodes_of) { sub[u] = weight[u]; W += weight[u]; }
                int best_e = -1, best_child = -1;
                long best_bal = -1;
                if (W >= 20) {
                    for (size_t q = nodes_of.size(); q-- > 1;) {
                        int u = nodes_of[q];
                        sub[par_b[u]] += sub[u];
                        long bal = std::labs(W - 2L * sub[u]);
                        if (best_bal < 0 || bal < best_bal || (bal == best_bal && par_e[u] < best_e)) { best_bal = bal; best_e = par_e[u]; best_child = u; }
                    }
                }
                if (best_e < 0) {  // fewer than 20 nodes, or no bridge inside ("min cut > 1"): solved whole
                    const int pid = (int)leaf_tnode.size();
                    for (int u : nodes_of) part_of[u] = pid;
                    leaf_tnode.push_back(jb.tnode);
                    continue;
                }
                cut[best_e] = 1;
                const int l = (int)tree.size(); tree.push_back(TreeNode());
                const int r = (int)tree.size(); tree.push_back(TreeNode());
                tree[jb.tnode].left = l; tree[jb.tnode].right = r; tree[jb.tnode].bridge = E[best_e];
                jobs.push_back(Job{best_child, l});
                jobs.push_back(Job{jb.root, r});
            }
        } else {
            // The same splits without a traversal of the whole part per split (a star-shaped blob tree -- thousands of small parts
            // hanging off one core, the dense-graph config -- made that quadratic: 2.3 s at 20 M records).  The tree is rooted
            // once; a part is a root plus what hangs below it, so "far side of a bridge" is always the subtree of its lower
            // end.  Per part: an ordered set of (subtree weight, bridge) over its blobs except the root; the most balanced bridge
            // is a neighbour of W / 2 in it.  A split takes the lower end's subtree out: the weights of its ancestors inside the
            // part drop (re-keyed one by one), and the set is divided by moving the entries of the SMALLER side into a new
            // set (each entry moves O(log n) times).
            std::vector<int> pb(nblob, -1), pe(nblob, -1), bfs;
            std::vector<long> sub(nblob, 0);
            std::vector<int> cnt(nblob, 1), child_of(m, -1);
            bfs.reserve(nblob);
            bfs.push_back(0);
            for (size_t q = 0; q < bfs.size(); ++q) {
                const int u = bfs[q];
                for (int k = th[u]; k < th[u + 1]; ++k) if (tid[k] != pe[u]) { const int v = tadj[k]; pb[v] = u; pe[v] = tid[k]; child_of[tid[k]] = v; bfs.push_back(v); }
            }
            for (int u = 0; u < nblob; ++u) sub[u] = weight[u];
            for (size_t q = bfs.size(); q-- > 1;) { const int u = bfs[q]; sub[pb[u]] += sub[u]; cnt[pb[u]] += cnt[u]; }
            typedef std::set<std::pair<long, int>> KeySet;  // (weight below the bridge, bridge): unique per blob
            struct Job { int root, tnode; std::unique_ptr<KeySet> keys; };
            std::vector<Job> jobs;
            {
                std::unique_ptr<KeySet> all(new KeySet());
                for (int u = 1; u < nblob; ++u) all->insert(std::make_pair(sub[u], pe[u]));
                jobs.push_back(Job{0, top, std::move(all)});
            }
            std::vector<int> walk;
            auto below = [&](int root, bool with_root, int skip_edge) {  // blobs of the part hanging below `root` (not crossing skip_edge)
                walk.clear();
                walk.push_back(root);
                for (size_t q = 0; q < walk.size(); ++q) {
                    const int u = walk[q];
                    for (int k = th[u]; k < th[u + 1]; ++k) if (tid[k] != pe[u] && !cut[tid[k]] && tid[k] != skip_edge) walk.push_back(tadj[k]);
                }
                if (!with_root) walk.erase(walk.begin());
            };
            while (!jobs.empty()) {
                Job jb = std::move(jobs.back());
                jobs.pop_back();
                KeySet& S = *jb.keys;
                const long W = sub[jb.root];
                int best_e = -1;
                if (W >= 20 && !S.empty()) {
                    // candidates: the lightest subtree with 2 * weight >= W and the heaviest below that (its smallest bridge id)
                    KeySet::iterator hi = S.lower_bound(std::make_pair((W + 1) / 2, INT_MIN));
                    long bal_hi = -1, bal_lo = -1;
                    int e_hi = -1, e_lo = -1;
                    if (hi != S.end()) { bal_hi = 2 * hi->first - W; e_hi = hi->second; }
                    if (hi != S.begin()) {
                        KeySet::iterator lo = std::prev(hi);
                        lo = S.lower_bound(std::make_pair(lo->first, INT_MIN));
                        bal_lo = W - 2 * lo->first; e_lo = lo->second;
                    }
                    if (e_hi >= 0 && (e_lo < 0 || bal_hi < bal_lo || (bal_hi == bal_lo && e_hi < e_lo))) best_e = e_hi; else best_e = e_lo;
                }
                if (best_e < 0) {  // fewer than 20 nodes, or no bridge inside ("min cut > 1"): solved whole
                    const int pid = (int)leaf_tnode.size();
                    below(jb.root, true, -1);
                    for (int u : walk) part_of[u] = pid;
                    leaf_tnode.push_back(jb.tnode);
                    continue;
                }
                const int child = child_of[best_e];
                const long d = sub[child];
                const int dc = cnt[child], total = cnt[jb.root];
                S.erase(std::make_pair(sub[child], best_e));
                for (int a = pb[child]; ; a = pb[a]) {  // the ancestors inside the part lose the subtree
                    if (a != jb.root) S.erase(std::make_pair(sub[a], pe[a]));
                    sub[a] -= d; cnt[a] -= dc;
                    if (a == jb.root) break;
                    S.insert(std::make_pair(sub[a], pe[a]));
                }
                std::unique_ptr<KeySet> other(new KeySet());
                const bool move_child_side = dc - 1 <= total - dc - 1;
                if (move_child_side) below(child, false, -1); else below(jb.root, false, best_e);
                for (int u : walk) { const std::pair<long, int> key(sub[u], pe[u]); S.erase(key); other->insert(key); }
                cut[best_e] = 1;
                const int l = (int)tree.size(); tree.push_back(TreeNode());
                const int r = (int)tree.size(); tree.push_back(TreeNode());
                tree[jb.tnode].left = l; tree[jb.tnode].right = r; tree[jb.tnode].bridge = E[best_e];
                std::unique_ptr<KeySet> mine = std::move(jb.keys);
                jobs.push_back(Job{child, l, move_child_side ? std::move(other) : std::move(mine)});
                jobs.push_back(Job{jb.root, r, move_child_side ? std::move(mine) : std::move(other)});
            }
        }
        // ---- leaves: ids ascending, edges in the component's order
        const int nleaf = (int)leaf_tnode.size();
        std::vector<std::vector<int>> lids(nleaf);
        std::vector<std::vector<Edge>> ledges(nleaf);
        for (int x = 0; x < n; ++x) lids[part_of[blob[x]]].push_back(ids[x]);
        for (int i = 0; i < m; ++i) { int pa = part_of[blob[ea[i]]], pb = part_of[blob[eb[i]]]; if (pa == pb) ledges[pa].push_back(E[i]); }
        for (int q = 0; q < nleaf; ++q) {
            Piece p;
            if (lids[q].size() == 1) { p.ids = lids[q]; p.order.assign(1, lids[q][0] + 1); }
            else make_piece(lids[q], ledges[q], p);
            tree[leaf_tnode[q]].piece = (int)pieces.size();
            pieces.push_back(std::move(p));
        }
        return top;
    }
    // join the ordered halves over the bridge edges, bottom-up (SegmentGraph.cpp:3393-3448); children always have larger
    // tree indices than their parent.  The reference copies, sorts (for the median id) and reverses whole halves at every
    // level, which is quadratic when thousands of small parts are peeled off one big one.  Here a half is a deque with
    // a lazy "reversed and negated" flag plus an order-statistics tree of its ids, and the smaller half is always moved
    // into the larger one: O(n log^2 n) in total, same sequence.
    struct Seq {
        std::deque<int> q;
        bool flipped = false;
        __gnu_pbds::tree<int, __gnu_pbds::null_type, std::less<int>, __gnu_pbds::rb_tree_tag, __gnu_pbds::tree_order_statistics_node_update> ids;
        size_t size() const { return q.size(); }
        int median() const { return *ids.find_by_order((ids.size() - 1) / 2); }
    };
    std::vector<int> combine(int root, int end, std::vector<int>& stored_sign /* scratch, indexed by node id */) {
        std::vector<std::unique_ptr<Seq>> res((size_t)(end - root));
        auto R = [&](int t) -> std::unique_ptr<Seq>& { return res[(size_t)(t - root)]; };
        for (int t = end - 1; t >= root; --t) {
            const TreeNode& tn = tree[t];
            if (tn.piece >= 0) {
                std::unique_ptr<Seq> sq(new Seq());
                for (int x : pieces[tn.piece].order) { sq->q.push_back(x); sq->ids.insert(std::abs(x) - 1); stored_sign[std::abs(x) - 1] = x > 0 ? 1 : -1; }
                R(t) = std::move(sq);
                continue;
            }
            std::unique_ptr<Seq> A = std::move(R(tn.left)), B = std::move(R(tn.right));
            auto endpoint = [&](const Seq& X, bool& positive, bool& head) {  // the bridge end that lies in X: its current sign, its head flag
                const bool a_in = X.ids.find(tn.bridge.a) != X.ids.end();
                const int v = a_in ? tn.bridge.a : tn.bridge.b;
                head = a_in ? tn.bridge.ha : tn.bridge.hb;
                positive = (stored_sign[v] > 0) != X.flipped;
            };
            bool p1, h1, p2, h2;
            endpoint(*A, p1, h1);
            endpoint(*B, p2, h2);
            const int m1 = A->median() + 1, m2 = B->median() + 1;
            Seq *first, *second;
            if (m1 < m2) {
                if (p1 == h1) A->flipped = !A->flipped;
                if (p2 != h2) B->flipped = !B->flipped;
                first = A.get(); second = B.get();
            } else {
                if (p2 == h2) B->flipped = !B->flipped;
                if (p1 != h1) A->flipped = !A->flipped;
                first = B.get(); second = A.get();
            }
            // logical element i of a half: flipped ? -q[size-1-i] : q[i]
            if (first->size() >= second->size()) {  // append `second` to `first`
                const size_t k = second->size();
                for (size_t i = 0; i < k; ++i) {
                    const int y = second->flipped ? -second->q[k - 1 - i] : second->q[i];
                    const int st = first->flipped ? -y : y;
                    if (!first->flipped) first->q.push_back(st); else first->q.push_front(st);
                    stored_sign[std::abs(y) - 1] = st > 0 ? 1 : -1;
                    first->ids.insert(std::abs(y) - 1);
                }
                R(t) = first == A.get() ? std::move(A) : std::move(B);
            } else {                                // prepend `first` to `second`
                const size_t k = first->size();
                for (size_t i = k; i-- > 0;) {
                    const int x = first->flipped ? -first->q[k - 1 - i] : first->q[i];
                    const int st = second->flipped ? -x : x;
                    if (!second->flipped) second->q.push_front(st); else second->q.push_back(st);
                    stored_sign[std::abs(x) - 1] = st > 0 ? 1 : -1;
                    second->ids.insert(std::abs(x) - 1);
                }
                R(t) = second == A.get() ? std::move(A) : std::move(B);
            }
        }
        const Seq& top = *R(root);
        std::vector<int> out(top.size());
        const size_t k = top.size();
        for (size_t i = 0; i < k; ++i) out[i] = top.flipped ? -top.q[k - 1 - i] : top.q[i];
        return out;
    }
};

}  // namespace

int order_components(sq_ctx* c) {
    struct W { sq_ctx* c; std::chrono::steady_clock::time_point t0; ~W() { c->timer.add("wall_order", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); } } wall{c, std::chrono::steady_clock::now()};
    const int n = (int)c->nodes.size();
    int ncomp = 0;
    for (int l : c->label) ncomp = std::max(ncomp, l + 1);
    // nodes and edges per component as two CSR tables (the reference rescans everything per component, :3248-3253): ids ascending, edges in
    // the order of the sorted edge list.  Round 5: the components are worked on in GROUPS of consecutive components, every group with its own
    // builder on the context's host threads -- min-cut trees, problem packing, decoding and the joining of the halves are per component; only
    // the two GPU batches see all of them at once.  (Round 4 walked 125 k components one after the other: 100-140 ms around a 1 ms kernel.)
    std::vector<int32_t> noff((size_t)ncomp + 1, 0), eoffc((size_t)ncomp + 1, 0);
    for (int i = 0; i < n; ++i) noff[(size_t)c->label[i] + 1]++;
    for (const Edge& e : c->edges) if (e.a != e.b) eoffc[(size_t)c->label[e.a] + 1]++;
    for (int k = 0; k < ncomp; ++k) { noff[(size_t)k + 1] += noff[k]; eoffc[(size_t)k + 1] += eoffc[k]; }
    std::vector<int32_t> cnode((size_t)n);
    std::vector<Edge> cedge((size_t)eoffc[ncomp]);
    {
        std::vector<int32_t> fn(noff.begin(), noff.end() - 1), fe(eoffc.begin(), eoffc.end() - 1);
        for (int i = 0; i < n; ++i) cnode[(size_t)fn[c->label[i]]++] = i;
        for (const Edge& e : c->edges) if (e.a != e.b) cedge[(size_t)fe[c->label[e.a]]++] = e;
    }
    const int GPU_NMAX = 8;
    static const long order_budget = std::getenv("SQUID_ORDER_BUDGET") ? std::atol(std::getenv("SQUID_ORDER_BUDGET")) : 20000000L;
    static const bool order_host_mid = std::getenv("SQUID_ORDER_HOST_MID") != nullptr;  // debugging: 9..19 nodes on the host as well
    struct Group {
        int k0 = 0, k1 = 0;
        Builder B;
        std::vector<int> roots, ends;
        std::vector<SmallProblem> probs, mprobs;
        std::vector<int32_t> e5, me5;
        std::vector<int> gpu_piece, mid_piece, large;
        size_t p_base = 0, m_base = 0;  // first problem of the group in the two batches
        std::vector<int> ord_nodes, ord_off;
    };
    const int ngroups = std::max(1, std::min(ncomp, (c->pool && ncomp > 256) ? 8 * (c->pool->size() + 1) : 1));
    std::vector<Group> groups((size_t)ngroups);
    auto each_group = [&](const std::function<void(Group&)>& f) {
        if (ngroups == 1) f(groups[0]);
        else c->pool->parallel_for(ngroups, 1 << 20, [&](int g) { f(groups[(size_t)g]); });
    };
    for (int g = 0; g < ngroups; ++g) { groups[(size_t)g].k0 = (int)((int64_t)ncomp * g / ngroups); groups[(size_t)g].k1 = (int)((int64_t)ncomp * (g + 1) / ngroups); }
    auto t0 = std::chrono::steady_clock::now();
    each_group([&](Group& G) {
        std::vector<int> ids;
        std::vector<Edge> es;
        for (int k = G.k0; k < G.k1; ++k) {
            ids.assign(cnode.begin() + noff[k], cnode.begin() + noff[(size_t)k + 1]);
            es.assign(cedge.begin() + eoffc[k], cedge.begin() + eoffc[(size_t)k + 1]);
            G.roots.push_back(G.B.build(ids, es));
            G.ends.push_back((int)G.B.tree.size());
        }
        // ---- leaves: <= 8 nodes and 9..19 nodes on the GPU (one batch each, one component per workgroup); larger bridge-free
        // pieces -- and the few the mid-size kernel hands back -- on host threads
        for (size_t pi = 0; pi < G.B.pieces.size(); ++pi) {
            Piece& p = G.B.pieces[pi];
            const int pn = (int)p.ids.size();
            if (pn == 1) continue;
            const bool small = pn <= GPU_NMAX, mid = !small && pn <= ORDER_MID_NMAX && !order_host_mid;
            if (!small && !mid) { G.large.push_back((int)pi); continue; }
            std::vector<int32_t>& e5 = small ? G.e5 : G.me5;
            SmallProblem sp{pn, (int)(e5.size() / 5), (int)p.edges.size()};
            for (const LEdge& e : p.edges) { e5.push_back(e.u); e5.push_back(e.v); e5.push_back(e.hu); e5.push_back(e.hv); e5.push_back(e.w); }
            (small ? G.probs : G.mprobs).push_back(sp);
            (small ? G.gpu_piece : G.mid_piece).push_back((int)pi);
        }
    });
    c->timer.add("host_mincut_tree", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    // the two batches: the groups' problems one after the other (edge offsets rebased)
    std::vector<SmallProblem> probs, mprobs;
    std::vector<int32_t> edges5, medges5;
    {
        size_t np = 0, nm = 0, ne = 0, nme = 0;
        for (Group& G : groups) { G.p_base = np; G.m_base = nm; np += G.probs.size(); nm += G.mprobs.size(); ne += G.e5.size(); nme += G.me5.size(); }
        probs.resize(np); mprobs.resize(nm); edges5.resize(ne); medges5.resize(nme);
        std::vector<size_t> eb(groups.size()), meb(groups.size());
        { size_t a = 0, b2 = 0; for (size_t g = 0; g < groups.size(); ++g) { eb[g] = a; meb[g] = b2; a += groups[g].e5.size(); b2 += groups[g].me5.size(); } }
        auto place = [&](int g) {
            Group& G = groups[(size_t)g];
            for (size_t q = 0; q < G.probs.size(); ++q) { SmallProblem sp = G.probs[q]; sp.eoff += (int)(eb[(size_t)g] / 5); probs[G.p_base + q] = sp; }
            for (size_t q = 0; q < G.mprobs.size(); ++q) { SmallProblem sp = G.mprobs[q]; sp.eoff += (int)(meb[(size_t)g] / 5); mprobs[G.m_base + q] = sp; }
            std::copy(G.e5.begin(), G.e5.end(), edges5.begin() + (std::ptrdiff_t)eb[(size_t)g]);
            std::copy(G.me5.begin(), G.me5.end(), medges5.begin() + (std::ptrdiff_t)meb[(size_t)g]);
        };
        if (ngroups == 1) place(0); else c->pool->parallel_for(ngroups, 1 << 20, place);
    }
    std::vector<int32_t> gmask, gorder;
    std::vector<int32_t> mmask, morder, mvalue, mstatus;
    std::future<int> mid;  // (the mid-size batch on a helper thread and a stream of its own, beside the small one)
    if (!mprobs.empty() && !probs.empty()) mid = c->pool->submit([&]() { return dev_order_mid(c, mprobs, medges5, mmask, morder, mvalue, mstatus, true); });
    int rc = dev_order_small(c, probs, edges5, gmask, gorder, GPU_NMAX);
    if (mid.valid()) { const int rm = mid.get(); if (!rc) rc = rm; }
    else if (!rc && !mprobs.empty()) rc = dev_order_mid(c, mprobs, medges5, mmask, morder, mvalue, mstatus);
    if (rc) return rc;
    each_group([&](Group& G) {
        for (size_t q = 0; q < G.gpu_piece.size(); ++q) {
            Piece& p = G.B.pieces[(size_t)G.gpu_piece[q]];
            const int pn = (int)p.ids.size();
            const size_t gq = G.p_base + q;
            p.order.resize(pn);
            for (int pos = 0; pos < pn; ++pos) {
                int l = gorder[gq * 8 + pos];
                p.order[pos] = ((gmask[gq] >> l) & 1) ? -(p.ids[l] + 1) : (p.ids[l] + 1);
            }
        }
        for (size_t q = 0; q < G.mid_piece.size(); ++q) {
            Piece& p = G.B.pieces[(size_t)G.mid_piece[q]];
            const size_t gq = G.m_base + q;
            if (mstatus[gq]) { G.large.push_back(G.mid_piece[q]); continue; }  // beyond the kernel's capacities: host solver
            const int pn = (int)p.ids.size();
            p.order.resize(pn);
            for (int pos = 0; pos < pn; ++pos) {
                int l = morder[gq * ORDER_MID_NMAX + pos];
                p.order[pos] = ((mmask[gq] >> l) & 1) ? -(p.ids[l] + 1) : (p.ids[l] + 1);
            }
        }
    });
    t0 = std::chrono::steady_clock::now();
    static const bool order_prof = std::getenv("SQUID_ORDER_PROF") != nullptr;
    std::atomic<long> unsolved{0};
    auto solve = [&](Piece& p) {
        const int pn = (int)p.ids.size();
        p.order.resize(pn);
        bool ok = pn <= HOST_NMAX;
        if (ok) {
            HostSolver hs(pn, p.edges, order_budget);
            auto ts = std::chrono::steady_clock::now();
            hs.run();
            if (order_prof) std::fprintf(stderr, "[order] piece n=%d m=%zu nodes=%ld leaves=%ld cyclic=%ld best=%ld%s  %.3f ms\n", pn, p.edges.size(), hs.n_nodes, hs.n_leaves, hs.n_cyclic, hs.best,
                                         hs.failed ? " GAVE UP" : "", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts).count());
            ok = !hs.failed && hs.best >= 0;
            if (ok)
                for (int pos = 0; pos < pn; ++pos) {
                    int l = hs.bestorder[pos];
                    p.order[pos] = mask_bit(hs.bestmask, l) ? -(p.ids[l] + 1) : (p.ids[l] + 1);
                }
        }
        if (!ok) {
            // what the reference keeps when glp_intopt gives up (:3287-3292,3984): identity order, all forward -- never silently
            for (int k = 0; k < pn; ++k) p.order[k] = p.ids[k] + 1;
            ++unsolved;
        }
    };
    // the pieces are independent: solve them on a few host threads (biggest first)
    std::vector<Piece*> large;
    for (Group& G : groups) for (int pi : G.large) large.push_back(&G.B.pieces[(size_t)pi]);
    std::sort(large.begin(), large.end(), [](const Piece* x, const Piece* y) { return x->ids.size() > y->ids.size(); });
    // (a handful of pieces is done before a helper would have picked one up)
    if (large.size() <= 4) for (Piece* p : large) solve(*p);
    else c->pool->parallel_for((int)large.size(), 1 << 20, [&](int i) { solve(*large[(size_t)i]); });
    c->counts.n_order_unsolved = unsolved.load();
    if (unsolved.load()) std::fprintf(stderr, "libsquid_hip: %ld component(s) beyond the exact ordering solver (more than %d nodes without a bridge, or search budget exhausted): identity order kept, as the reference does when GLPK gives up\n", unsolved.load(), HOST_NMAX);
    c->timer.add("host_order_large", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    std::vector<int> sign_scratch((size_t)n, 1);  // (indexed by node id: the groups' components are disjoint)
    each_group([&](Group& G) {
        G.ord_off.assign(1, 0);
        for (int k = G.k0; k < G.k1; ++k) {
            std::vector<int> o = G.B.combine(G.roots[(size_t)(k - G.k0)], G.ends[(size_t)(k - G.k0)], sign_scratch);
            G.ord_nodes.insert(G.ord_nodes.end(), o.begin(), o.end());
            G.ord_off.push_back((int)G.ord_nodes.size());
        }
    });
    c->ord_off.assign(1, 0);
    c->ord_nodes.clear();
    for (Group& G : groups) {
        const int32_t base = (int32_t)c->ord_nodes.size();
        c->ord_nodes.insert(c->ord_nodes.end(), G.ord_nodes.begin(), G.ord_nodes.end());
        for (size_t q = 1; q < G.ord_off.size(); ++q) c->ord_off.push_back(base + (int32_t)G.ord_off[q]);
    }
    c->ordered = true;
    return SQ_OK;
}

// tests: one ordering problem (local nodes 0..n-1, edges as u,v,hu,hv,w with u < v) through the GPU kernels (use_gpu: n <= 8
// k_order_small, 9..19 k_order_mid) or the host solver (n <= 128).  order[p] = local node at position p, bit-complemented when
// the node is reversed; mask = the orientation mask when n <= 31, else -1.
int order_problem_debug(sq_ctx* c, int n, const std::vector<int32_t>& edges5, bool use_gpu, int32_t& mask, std::vector<int32_t>& order, int64_t& value) {
    if (n < 2 || n > HOST_NMAX || edges5.size() % 5) return fail(c, SQ_E_ARG, "bad ordering problem");
    for (size_t i = 0; i < edges5.size(); i += 5)
        if (edges5[i] < 0 || edges5[i] >= edges5[i + 1] || edges5[i + 1] >= n || edges5[i + 4] <= 0) return fail(c, SQ_E_ARG, "bad ordering problem edge");
    order.assign(n, 0);
    Mask bm = 0;
    if (use_gpu) {
        if (n > ORDER_MID_NMAX) return fail(c, SQ_E_ARG, "the GPU kernels take at most 19 nodes");
        std::vector<SmallProblem> probs(1, SmallProblem{n, 0, (int)(edges5.size() / 5)});
        std::vector<int32_t> gm, go, gv, gs;
        if (n <= 8) {
            int rc = dev_order_small(c, probs, edges5, gm, go, 8);
            if (rc) return rc;
            value = -1;  // (the kernel's value stays on the device; the caller evaluates the result)
        } else {
            int rc = dev_order_mid(c, probs, edges5, gm, go, gv, gs);
            if (rc) return rc;
            if (gs[0]) return fail(c, SQ_E_CAPACITY, "k_order_mid handed the problem back (capacity)");
            value = gv[0];
        }
        bm = (Mask)(uint32_t)gm[0];
        for (int p = 0; p < n; ++p) order[p] = go[p];
    } else {
        std::vector<LEdge> E;
        for (size_t i = 0; i < edges5.size(); i += 5) E.push_back(LEdge{edges5[i], edges5[i + 1], edges5[i + 2] != 0, edges5[i + 3] != 0, edges5[i + 4]});
        static const long order_budget = std::getenv("SQUID_ORDER_BUDGET") ? std::atol(std::getenv("SQUID_ORDER_BUDGET")) : 20000000L;
        HostSolver hs(n, E, order_budget);
        hs.run();
        if (hs.failed || hs.best < 0) return fail(c, SQ_E_CAPACITY, "ordering problem beyond the exact solver's budget");
        bm = hs.bestmask; value = hs.best;
        for (int p = 0; p < n; ++p) order[p] = hs.bestorder[p];
    }
    mask = n <= 31 ? (int32_t)(uint32_t)bm : -1;
    for (int p = 0; p < n; ++p) if (mask_bit(bm, order[p])) order[p] = ~order[p];
    return SQ_OK;
}

}  // namespace sq
