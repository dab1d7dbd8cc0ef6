// ORACLE (test infrastructure only) -- never linked, imported or executed by the product path.
// PARITY UNPINNED (see o_bam.h), and doubly so here: the arithmetic of this stage lives in two un-vendored
// third parties, GLPK 4.62 `glp_intopt` (reference Makefile:3, call site src/SegmentGraph.cpp:3966) and
// Boost.Graph 1.55 `stoer_wagner_min_cut` (Makefile:1, call site :3325; replaced by the bridge rule of BridgeSplit).
// What is restated:
//   src/SegmentGraph.cpp:3236-3262 Ordering, :3264-3451 MincutRecursion, :3763-3983 GenerateILP (the model).
// The ILP is solved EXACTLY by enumeration; the optimum VALUE is solver independent.
// Among equal-value optima GLPK's choice is unknowable here, so the oracle fixes a canonical one:
//   1. maximum objective;  2. then the smallest orientation mask (bit i set <=> local node i reversed,
//   node 0 = least significant bit);  3. then the lexicographically smallest left-to-right node sequence.
// `OrderStats` reports how many components had optima that disagree on the satisfied discordant edges --
// only those can make `_sv.txt` differ from a GLPK run.
#pragma once
#include "o_graph.h"

namespace oracle {

struct LocalEdge { int u, v; bool hu, hv; int w; bool discordant; };  // u < v local indices

struct OrderStats {
    long components = 0, solved = 0, ambiguous = 0, too_large = 0, mincut_splits = 0, ambiguity_checked = 0;
};

// does the signed arrangement satisfy edge e?  (rows of GenerateILP, :3803-3931; fwd[i]=y_i, pos[i]=rank)
inline bool EdgeSatisfied(const LocalEdge& e, const std::vector<int>& fwd, const std::vector<int>& pos) {
    bool ubefore = pos[e.u] < pos[e.v];
    bool yu = fwd[e.u], yv = fwd[e.v];
    if (!e.hu && e.hv) return yu == yv && ubefore == yu;       // tail -> head
    if (!e.hu && !e.hv) return yu != yv && ubefore == yu;      // tail - tail
    if (e.hu && e.hv) return yu != yv && ubefore == yv;        // head - head
    return yu == yv && ubefore == !yu;                          // head -> tail
}

// arc weights a[x][y] ("x wants to precede y") induced by orientation mask (bit set = reversed)
inline long BuildArcs(int n, const std::vector<LocalEdge>& E, unsigned mask, std::vector<std::vector<int>>* a) {
    long ub = 0;
    if (a) a->assign(n, std::vector<int>(n, 0));
    for (const LocalEdge& e : E) {
        bool yu = !((mask >> e.u) & 1), yv = !((mask >> e.v) & 1);
        bool compat, ufirst;
        if (!e.hu && e.hv) { compat = yu == yv; ufirst = yu; }
        else if (!e.hu && !e.hv) { compat = yu != yv; ufirst = yu; }
        else if (e.hu && e.hv) { compat = yu != yv; ufirst = yv; }
        else { compat = yu == yv; ufirst = !yu; }
        if (!compat) continue;
        ub += e.w;
        if (a) { if (ufirst) (*a)[e.u][e.v] += e.w; else (*a)[e.v][e.u] += e.w; }
    }
    return ub;
}

// exhaustive reference solver for tiny components: all n! * 2^n signed permutations, explicit comparator
inline void SolveBrute(int n, const std::vector<LocalEdge>& E, std::vector<int>& order, unsigned& bestmask, long& bestval, bool* ambiguous) {
    std::vector<int> perm(n), pos(n), fwd(n);
    for (int i = 0; i < n; i++) perm[i] = i;
    bestval = -1; bestmask = 0;
    std::vector<std::vector<bool>> optsets;
    do {
        for (int i = 0; i < n; i++) pos[perm[i]] = i;
        for (unsigned mask = 0; mask < (1u << n); mask++) {
            for (int i = 0; i < n; i++) fwd[i] = !((mask >> i) & 1);
            long val = 0;
            for (const LocalEdge& e : E) if (EdgeSatisfied(e, fwd, pos)) val += e.w;
            bool better = val > bestval || (val == bestval && (mask < bestmask || (mask == bestmask && perm < order)));
            if (ambiguous && val >= bestval) {
                if (val > bestval) optsets.clear();
                std::vector<bool> s;
                for (const LocalEdge& e : E) if (e.discordant) s.push_back(EdgeSatisfied(e, fwd, pos));
                if (std::find(optsets.begin(), optsets.end(), s) == optsets.end()) optsets.push_back(s);
            }
            if (better) { bestval = val; bestmask = mask; order = perm; }
        }
    } while (std::next_permutation(perm.begin(), perm.end()));
    if (ambiguous) *ambiguous = optsets.size() > 1;
}

// ---- linear-ordering sub-problem of one orientation: arcs a[x][y] ("x wants to precede y")
// Acyclic arc set: every arc can be satisfied; the lexicographically smallest optimal sequence is the
// smallest-index-first topological order.  Returns false when the arcs contain a cycle.
inline bool TopoSmallestFirst(int n, const std::vector<std::vector<int>>& a, std::vector<int>& order) {
    std::vector<int> indeg(n, 0);
    for (int x = 0; x < n; x++) for (int y = 0; y < n; y++) if (a[x][y] > 0) indeg[y]++;
    std::vector<bool> done(n, false);
    order.clear();
    for (int p = 0; p < n; p++) {
        int v = -1;
        for (int c = 0; c < n; c++) if (!done[c] && indeg[c] == 0) { v = c; break; }
        if (v < 0) return false;
        done[v] = true;
        order.push_back(v);
        for (int y = 0; y < n; y++) if (a[v][y] > 0) indeg[y]--;
    }
    return true;
}

// Cyclic arc set: subset DP.  h[S] = best weight still obtainable once the set S occupies the first |S|
// positions; gain(S,v) = sum_{u in S} a[u][v] is looked up from two half-mask tables.
struct OrderDP {
    int n, lo_bits;
    std::vector<std::vector<long>> GL, GH;
    std::vector<long> h;
    OrderDP(int n, const std::vector<std::vector<int>>& a) : n(n), lo_bits(n / 2) {
        int hi_bits = n - lo_bits;
        GL.assign(n, std::vector<long>((size_t)1 << lo_bits, 0));
        GH.assign(n, std::vector<long>((size_t)1 << hi_bits, 0));
        for (int v = 0; v < n; v++) {
            for (size_t m = 1; m < GL[v].size(); m++) { int b = __builtin_ctzl(m); GL[v][m] = GL[v][m & (m - 1)] + a[b][v]; }
            for (size_t m = 1; m < GH[v].size(); m++) { int b = __builtin_ctzl(m); GH[v][m] = GH[v][m & (m - 1)] + a[lo_bits + b][v]; }
        }
    }
    long gain(size_t S, int v) const { return GL[v][S & (((size_t)1 << lo_bits) - 1)] + GH[v][S >> lo_bits]; }
    long solve() {
        const size_t full = ((size_t)1 << n) - 1;
        h.assign(full + 1, 0);
        for (size_t S = full; S-- > 0;) {
            long best = -1;
            for (int v = 0; v < n; v++) {
                if ((S >> v) & 1) continue;
                long t = gain(S, v) + h[S | ((size_t)1 << v)];
                if (t > best) best = t;
            }
            h[S] = best;
        }
        return h[0];
    }
    void smallest_order(std::vector<int>& order) const {
        order.clear();
        size_t S = 0;
        for (int p = 0; p < n; p++)
            for (int v = 0; v < n; v++) {
                if ((S >> v) & 1) continue;
                if (gain(S, v) + h[S | ((size_t)1 << v)] == h[S]) { order.push_back(v); S |= (size_t)1 << v; break; }
            }
    }
};

// Branch and bound over orientations.  Nodes are fixed from the highest local index down, forward first, and a
// branch is cut when its optimistic bound cannot STRICTLY beat the incumbent -- so the first incumbent of the
// final value is the one with the smallest orientation mask (rule 2 of the canonical choice).
struct BnB {
    int n;
    const std::vector<LocalEdge>& E;
    long bestval = -1;
    unsigned bestmask = 0;
    std::vector<int> bestorder;
    long leaves = 0, dps = 0;
    BnB(int n, const std::vector<LocalEdge>& E) : n(n), E(E) {}
    // optimistic value with nodes > k fixed by `mask`, nodes <= k free
    long bound(unsigned mask, int k) const {
        long ub = 0;
        for (const LocalEdge& e : E) {
            if (e.u <= k) { ub += e.w; continue; }  // u < v, so u free => at least one end free
            bool yu = !((mask >> e.u) & 1), yv = !((mask >> e.v) & 1);
            bool sameori = (e.hu != e.hv);  // tail->head / head->tail need equal orientation
            if (sameori ? (yu == yv) : (yu != yv)) ub += e.w;
        }
        return ub;
    }
    void leaf(unsigned mask) {
        leaves++;
        std::vector<std::vector<int>> a;
        long ub = BuildArcs(n, E, mask, &a);
        std::vector<int> order;
        long val;
        if (TopoSmallestFirst(n, a, order)) val = ub;
        else { OrderDP dp(n, a); dps++; val = dp.solve(); if (val > bestval) dp.smallest_order(order); }
        if (val > bestval) { bestval = val; bestmask = mask; bestorder = order; }
    }
    void rec(unsigned mask, int k) {  // node k is the next to fix
        if (bound(mask, k) <= bestval) return;
        if (k < 0) { leaf(mask); return; }
        rec(mask, k - 1);
        rec(mask | (1u << k), k - 1);
    }
    void run() { rec(0, n - 1); }
};

inline void SolveDP(int n, const std::vector<LocalEdge>& E, std::vector<int>& order, unsigned& bestmask, long& bestval) {
    BnB b(n, E);
    b.run();
    order = b.bestorder; bestmask = b.bestmask; bestval = b.bestval;
}

// ---- components of more than 26 nodes (MincutRecursion hands them to GenerateILP whole when no weight-1 cut exists,
// :3326-3349; GLPK gets 300 s).  Same canonical optimum as above -- max value, then the smallest orientation mask read as a
// number (bit i = node i reversed), then the lexicographically smallest sequence -- found by an orientation search on
// std::vector<bool> and, per orientation, Tarjan's strongly connected components: arcs between components are all
// satisfiable, each non-trivial component gets the subset DP.  `steps` is a work budget; on exhaustion `ok` is false and
// the caller keeps the identity order (what the reference keeps when GLPK fails, :3287-3292,3984).
struct WideSolver {
    int n;
    const std::vector<LocalEdge>& E;
    long bestval = -1;
    std::vector<bool> bestrev;
    std::vector<int> bestorder;
    long steps = 0, budget;
    bool ok = true;
    WideSolver(int n, const std::vector<LocalEdge>& E, long budget) : n(n), E(E), budget(budget) {}
    static bool Compat(const LocalEdge& e, const std::vector<bool>& rev, bool& ufirst) {
        bool yu = !rev[e.u], yv = !rev[e.v];
        bool compat;
        if (!e.hu && e.hv) { compat = yu == yv; ufirst = yu; }
        else if (!e.hu && !e.hv) { compat = yu != yv; ufirst = yu; }
        else if (e.hu && e.hv) { compat = yu != yv; ufirst = yv; }
        else { compat = yu == yv; ufirst = !yu; }
        return compat;
    }
    // nodes above k are fixed.  Upper bound: compatible edges among fixed nodes, all edges among free nodes, and for every free
    // node the heavier of the two edge sets towards fixed nodes that its forward / its reversed orientation would allow
    long Bound(const std::vector<bool>& rev, int k) const {
        long ub = 0;
        std::vector<long> asfwd(k + 1 > 0 ? k + 1 : 0, 0), asrev(k + 1 > 0 ? k + 1 : 0, 0);
        for (const LocalEdge& e : E) {
            if (e.u > k) { bool uf; if (Compat(e, rev, uf)) ub += e.w; }
            else if (e.v <= k) ub += e.w;
            else {
                // (one of the two orientations of u always fits: equal orientations for tail->head / head->tail edges, opposite otherwise)
                const bool v_forward = !rev[e.v], need_equal = e.hu != e.hv;
                if (need_equal == v_forward) asfwd[e.u] += e.w; else asrev[e.u] += e.w;
            }
        }
        for (int u = 0; u <= k; u++) ub += std::max(asfwd[u], asrev[u]);
        return ub;
    }
    void Leaf(const std::vector<bool>& rev) {
        std::vector<std::vector<int>> a(n, std::vector<int>(n, 0));
        long ub = 0;
        for (const LocalEdge& e : E) { bool uf; if (!Compat(e, rev, uf)) continue; ub += e.w; if (uf) a[e.u][e.v] += e.w; else a[e.v][e.u] += e.w; }
        std::vector<int> order;
        long val;
        if (TopoSmallestFirst(n, a, order)) val = ub;
        else {
            order.clear();
            // Tarjan
            std::vector<int> idx(n, -1), low(n, 0), comp(n, -1), stk;
            std::vector<bool> on(n, false);
            int counter = 0, ncomp = 0;
            std::vector<std::pair<int, int>> call;  // (node, next neighbour)
            for (int r = 0; r < n; r++) {
                if (idx[r] >= 0) continue;
                call.push_back({r, 0});
                idx[r] = low[r] = counter++; stk.push_back(r); on[r] = true;
                while (!call.empty()) {
                    int x = call.back().first, &y = call.back().second;
                    if (y < n) {
                        int t = y++;
                        if (a[x][t] <= 0) continue;
                        if (idx[t] < 0) { idx[t] = low[t] = counter++; stk.push_back(t); on[t] = true; call.push_back({t, 0}); }
                        else if (on[t]) low[x] = std::min(low[x], idx[t]);
                    } else {
                        if (low[x] == idx[x]) { int t; do { t = stk.back(); stk.pop_back(); on[t] = false; comp[t] = ncomp; } while (t != x); ncomp++; }
                        call.pop_back();
                        if (!call.empty()) low[call.back().first] = std::min(low[call.back().first], low[x]);
                    }
                }
            }
            std::vector<std::vector<int>> mem(ncomp);
            for (int x = 0; x < n; x++) mem[comp[x]].push_back(x);
            val = 0;
            for (int x = 0; x < n; x++) for (int y = 0; y < n; y++) if (comp[x] != comp[y]) val += a[x][y];
            std::vector<OrderDP*> dps(ncomp, nullptr);
            std::vector<size_t> placed(ncomp, 0);
            for (int c = 0; c < ncomp && ok; c++) {
                int sz = (int)mem[c].size();
                if (sz < 2) continue;
                if (sz > 22) {
                    // no table for that many subsets.  At least one arc of the component is violated, so the orientation is worth at most
                    // ub - (its lightest arc): enough to discard it when the incumbent is at least as good; otherwise give up
                    int lightest = 0;
                    for (int i : mem[c]) for (int j : mem[c]) if (a[i][j] > 0 && (lightest == 0 || a[i][j] < lightest)) lightest = a[i][j];
                    if (ub - lightest <= bestval) { val = -1; break; }
                    ok = false; break;
                }
                std::vector<std::vector<int>> sub(sz, std::vector<int>(sz, 0));
                for (int i = 0; i < sz; i++) for (int j = 0; j < sz; j++) sub[i][j] = a[mem[c][i]][mem[c][j]];
                dps[c] = new OrderDP(sz, sub);
                steps += (long)1 << sz;
                val += dps[c]->solve();
            }
            if (ok && val < 0) { for (OrderDP* d : dps) delete d; return; }  // discarded by the cycle bound
            if (ok && val > bestval) {
                std::vector<bool> done(n, false);
                for (int p = 0; p < n; p++)
                    for (int v = 0; v < n; v++) {
                        if (done[v]) continue;
                        bool free = true;
                        for (int x = 0; x < n && free; x++) if (a[x][v] > 0 && comp[x] != comp[v] && !done[x]) free = false;
                        if (!free) continue;
                        int c = comp[v];
                        if (dps[c]) {
                            int lv = (int)(std::find(mem[c].begin(), mem[c].end(), v) - mem[c].begin());
                            if (dps[c]->gain(placed[c], lv) + dps[c]->h[placed[c] | ((size_t)1 << lv)] != dps[c]->h[placed[c]]) continue;
                            placed[c] |= (size_t)1 << lv;
                        }
                        order.push_back(v); done[v] = true;
                        break;
                    }
            }
            for (OrderDP* d : dps) delete d;
            if (!ok) return;
        }
        if (val > bestval) { bestval = val; bestrev = rev; bestorder = order; }
    }
    void Rec(std::vector<bool>& rev, int k) {
        if (!ok) return;
        if (++steps > budget) { ok = false; return; }
        if (Bound(rev, k) <= bestval) return;
        if (k < 0) { Leaf(rev); return; }
        Rec(rev, k - 1);                       // forward first: masks are met in increasing numeric order
        if (k != n - 1) { rev[k] = true; Rec(rev, k - 1); rev[k] = false; }  // (the mirror image of a solution has the last node reversed and a larger mask)
    }
    void Run() { std::vector<bool> rev(n, false); Rec(rev, n - 1); if (bestorder.empty()) ok = false; if (!ok) bestval = -1; }
};

// ---- the optimum VALUE of a wide problem by another road (round 6): which edges to give up.
// A set of edges can be satisfied together iff (a) the orientations they ask for agree -- a parity per node against the root of its tree of
// kept edges: tail->head and head->tail edges want equal orientations, tail-tail and head-head edges opposite ones -- and (b) their
// precedence arcs, which the orientations determine up to reversing a whole tree, have no directed cycle.  Depth-first over the edges by
// descending weight, keeping an edge first and dropping it second, cut when what is kept plus what is still undecided cannot beat the best
// set so far: exact, and fast on graphs whose edges nearly all fit together (a backbone plus a few heavy discordant edges: the 75-node
// component of the --bwa sample that neither orientation search could finish takes 1 221 steps).  WideSolver then starts from value - 1
// and only has to find the canonical solution of that value.  `steps` is a work budget; when it runs out, the best set found so far
// is still a lower bound.
struct KeepDrop {
    int n;
    const std::vector<LocalEdge>& E;
    std::vector<int> idx, par, rel, size;
    std::vector<long> suffix;
    std::vector<std::vector<int>> adj;
    long best = -1, steps = 0, budget;
    bool complete = true;
    KeepDrop(int n, const std::vector<LocalEdge>& E, long budget) : n(n), E(E), budget(budget) {
        idx.resize(E.size());
        for (size_t i = 0; i < E.size(); i++) idx[i] = (int)i;
        std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return E[x].w > E[y].w; });
        suffix.assign(E.size() + 1, 0);
        for (size_t i = E.size(); i-- > 0;) suffix[i] = suffix[i + 1] + E[idx[i]].w;
        par.resize(n); rel.assign(n, 0); size.assign(n, 1); adj.assign(n, {});
        for (int i = 0; i < n; i++) par[i] = i;
    }
    int Find(int x, int& p) const { p = 0; while (par[x] != x) { p ^= rel[x]; x = par[x]; } return x; }
    void Arc(const LocalEdge& e, int& from, int& to) const {
        int pu, pv;
        Find(e.u, pu); Find(e.v, pv);
        const bool yu = !pu, yv = !pv;
        bool ufirst;
        if (!e.hu && e.hv) ufirst = yu; else if (!e.hu && !e.hv) ufirst = yu; else if (e.hu && e.hv) ufirst = yv; else ufirst = !yu;  // (rows of GenerateILP, as in BuildArcs)
        from = ufirst ? e.u : e.v; to = ufirst ? e.v : e.u;
    }
    bool Reaches(int a, int b) const {
        std::vector<char> seen(n, 0);
        std::vector<int> st(1, a);
        seen[a] = 1;
        while (!st.empty()) {
            int x = st.back(); st.pop_back();
            if (x == b) return true;
            for (int ei : adj[x]) { int f, t; Arc(E[ei], f, t); if (f == x && !seen[t]) { seen[t] = 1; st.push_back(t); } }
        }
        return false;
    }
    void Rec(size_t i, long val) {
        if (++steps > budget) { complete = false; return; }
        if (val + suffix[i] <= best) return;
        if (i == idx.size()) { best = val; return; }
        const LocalEdge& e = E[idx[i]];
        int pu, pv, ru = Find(e.u, pu), rv = Find(e.v, pv);
        const int want = (e.hu != e.hv) ? 0 : 1;
        bool ok = true;
        if (ru == rv) {
            if ((pu ^ pv) != want) ok = false;
            else { int f, t; Arc(e, f, t); if (Reaches(t, f)) ok = false; }
        }
        if (ok) {
            int joined = -1;
            if (ru != rv) {
                if (size[ru] < size[rv]) { std::swap(ru, rv); std::swap(pu, pv); }
                par[rv] = ru; rel[rv] = pu ^ pv ^ want; size[ru] += size[rv]; joined = rv;
            }
            adj[e.u].push_back(idx[i]); adj[e.v].push_back(idx[i]);
            Rec(i + 1, val + e.w);
            adj[e.u].pop_back(); adj[e.v].pop_back();
            if (joined >= 0) { size[par[joined]] -= size[joined]; par[joined] = joined; rel[joined] = 0; }
            if (!complete) return;
        }
        Rec(i + 1, val);
    }
    long Run() { Rec(0, 0); return best; }
};

class Orderer {
public:
    const SegmentGraph_t& G;
    OrderStats stats;
    int brute_max = 7;     // components up to this size are solved by plain enumeration (and checked for ambiguity)
    int exact_max = 19;    // up to here: branch and bound with a whole-component subset DP; 20..128: WideSolver; beyond: identity
    long wide_budget = 20000000L;
    explicit Orderer(const SegmentGraph_t& g) : G(g) {}

    // src/SegmentGraph.cpp:3236-3262
    std::vector<std::vector<int>> Ordering() {
        int componentsize = 0;
        for (int l : G.Label) if (l > componentsize) componentsize = l;
        componentsize++;
        std::vector<std::vector<int>> BestOrders(componentsize);
        // bucket nodes/edges per component once (the reference rescans everything per component; same result)
        std::vector<std::map<int, int>> CompNodes(componentsize);
        std::vector<std::vector<Edge_t>> CompEdges(componentsize);
        std::vector<int> cnt(componentsize, 0);
        for (int j = 0; j < (int)G.Label.size(); j++) CompNodes[G.Label[j]][j] = cnt[G.Label[j]]++;
        for (const Edge_t& e : G.vEdges)
            if (e.Ind1 != e.Ind2) {
                CompEdges[G.Label[e.Ind1]].push_back(e);
                if (G.Label[e.Ind2] != G.Label[e.Ind1]) CompEdges[G.Label[e.Ind2]].push_back(e);  // cannot happen for a CC labelling
            }
        for (int i = 0; i < componentsize; i++) {
            stats.components++;
            if (CompNodes[i].size() == 1) { BestOrders[i].push_back(CompNodes[i].begin()->first + 1); continue; }
            BestOrders[i] = MincutRecursion(CompNodes[i], CompEdges[i]);
        }
        return BestOrders;
    }

    // Uniqueness certificate for 8..19 nodes (too many for the enumeration of SolveBrute): is the set of satisfied DISCORDANT
    // edges the same in every optimal solution?  For a discordant edge e satisfied by the canonical optimum, delete it: if the
    // optimum of the rest is still `val`, some optimal solution does not satisfy e.  For one that is not satisfied, add 1 to
    // its weight: if the optimum becomes val + 1, some optimal solution satisfies it.
    static bool Ambiguous(int n, const std::vector<LocalEdge>& E, const std::vector<int>& order, unsigned mask, long val) {
        std::vector<int> pos(n), fwd(n);
        for (int p = 0; p < n; p++) pos[order[p]] = p;
        for (int i = 0; i < n; i++) fwd[i] = !((mask >> i) & 1);
        for (size_t k = 0; k < E.size(); k++) {
            if (!E[k].discordant) continue;
            std::vector<LocalEdge> F = E;
            std::vector<int> o2; unsigned m2; long v2;
            if (EdgeSatisfied(E[k], fwd, pos)) {
                F.erase(F.begin() + (long)k);
                SolveDP(n, F, o2, m2, v2);
                if (v2 == val) return true;
            } else {
                F[k].w += 1;
                SolveDP(n, F, o2, m2, v2);
                if (v2 == val + 1) return true;
            }
        }
        return false;
    }

    // every (sub-)problem whose optima disagree on the satisfied discordant edges, for the dump (ambiguous.txt): global node ids of
    // the problem and its discordant edges -- what a reader needs to see whether an SV row hangs on the solver's choice among ties
    std::vector<std::string> ambiguous_notes;
private:
    void NoteAmbiguous(const std::map<int, int>& CompNodes, const std::vector<LocalEdge>& E) {
        std::vector<int> ids(CompNodes.size());
        for (auto& kv : CompNodes) ids[kv.second] = kv.first;
        std::string s = "nodes";
        for (int id : ids) s += " " + std::to_string(id);
        s += "\tdiscordant_edges";
        for (const LocalEdge& e : E)
            if (e.discordant) s += " " + std::to_string(ids[e.u]) + (e.hu ? "H" : "T") + "-" + std::to_string(ids[e.v]) + (e.hv ? "H" : "T") + ":" + std::to_string(e.w);
        ambiguous_notes.push_back(s);
    }
    // backbone edges + exact solve + decode (:3271-3314 and :3326-3370)
    std::vector<int> SolveWhole(std::map<int, int>& CompNodes, std::vector<Edge_t> CompEdges) {
        const int n = (int)CompNodes.size();
        std::vector<int> BestOrder(n, 0);
        size_t edgeidx = 0;
        const size_t n_real = CompEdges.size();  // (the backbone edges appended below are not edges of the graph: they can never become an SV row)
        std::map<int, int>::iterator itnodeend = CompNodes.end();
        itnodeend--;
        for (std::map<int, int>::iterator itnode = CompNodes.begin(); itnode != itnodeend; itnode++) {
            bool isfound = false;
            for (; edgeidx < CompEdges.size() && CompEdges[edgeidx].Ind1 <= itnode->first; edgeidx++)
                if (CompNodes[CompEdges[edgeidx].Ind1] == itnode->second && CompNodes[CompEdges[edgeidx].Ind2] == itnode->second + 1) { isfound = true; break; }
            if (!isfound) {
                std::map<int, int>::iterator tmpit = itnode;
                tmpit++;
                CompEdges.push_back(Edge_t(itnode->first, false, tmpit->first, true, 1));  // ledger B17
            }
        }
        std::vector<LocalEdge> E;
        for (const Edge_t& e : CompEdges) E.push_back(LocalEdge{CompNodes[e.Ind1], CompNodes[e.Ind2], e.Head1, e.Head2, e.Weight, E.size() < n_real && G.IsDiscordant(e)});
        std::vector<int> order;
        unsigned mask = 0;
        long val = 0;
        if (n <= brute_max) {
            bool amb = false;
            SolveBrute(n, E, order, mask, val, &amb);
            if (amb) { stats.ambiguous++; NoteAmbiguous(CompNodes, E); }
            stats.solved++;
        } else if (n <= exact_max) {
            SolveDP(n, E, order, mask, val);
            stats.solved++;
            if (n < 20 && Ambiguous(n, E, order, mask, val)) { stats.ambiguous++; NoteAmbiguous(CompNodes, E); }
        } else if (n <= 128) {
            if (const char* df = std::getenv("ORACLE_DUMP_WIDE")) {  // (tests: the instances beyond the subset DP, one per line: n, then u v hu hv w per edge)
                std::ofstream o(df, std::ios::app);
                o << n;
                for (const LocalEdge& e : E) o << ' ' << e.u << ' ' << e.v << ' ' << e.hu << ' ' << e.hv << ' ' << e.w;
                o << '\n';
            }
            WideSolver ws(n, E, wide_budget);
            {   // the value first (KeepDrop), then the canonical solution of that value; ORACLE_NO_KEEPDROP: the orientation search from nothing
                static const bool off = std::getenv("ORACLE_NO_KEEPDROP") != nullptr;
                if (!off) { KeepDrop kd(n, E, 5000000L); const long v = kd.Run(); if (v > 0) ws.bestval = v - 1; }
            }
            ws.Run();
            if (ws.ok && ws.bestval >= 0) {
                stats.solved++;
                std::vector<int> ids(n);
                for (auto& kv : CompNodes) ids[kv.second] = kv.first;
                for (int p = 0; p < n; p++) { int l = ws.bestorder[p]; BestOrder[p] = ws.bestrev[l] ? (-ids[l] - 1) : (ids[l] + 1); }
                return BestOrder;
            }
            std::cerr << "oracle: component of " << n << " nodes exhausted the exact solver's budget; identity order kept\n";
            stats.too_large++;
            order.resize(n);
            for (int i = 0; i < n; i++) order[i] = i;
            mask = 0;
        } else {
            // identity order, all forward == what the reference keeps when glp_intopt fails (:3287-3292,3984)
            std::cerr << "oracle: component of " << n << " nodes exceeds the exact solver; identity order kept\n";
            stats.too_large++;
            order.resize(n);
            for (int i = 0; i < n; i++) order[i] = i;
            mask = 0;
        }
        std::vector<int> ids(n);
        for (auto& kv : CompNodes) ids[kv.second] = kv.first;
        for (int p = 0; p < n; p++) {
            int l = order[p];
            BestOrder[p] = ((mask >> l) & 1) ? (-ids[l] - 1) : (ids[l] + 1);
        }
        return BestOrder;
    }

    // Stand-in for boost::stoer_wagner_min_cut with unit weights (SegmentGraph.cpp:3316-3325).  The reference only
    // distinguishes "min cut > 1" from "min cut == 1" and, in the latter case, uses the returned bipartition.  A unit
    // weight cut of 1 is a bridge of the multigraph; WHICH bridge Boost returns is unknowable here (parity unpinned),
    // so the rule is fixed as: among all bridges take the most balanced one (minimise |n - 2*side|), ties by the
    // edge's position in the sorted component edge list.  Brute force: drop each candidate edge and flood-fill.
    // The brute force is quadratic; components above 64 nodes (dense-graph configs have ones with 10^5) use
    // BridgeSplitChains, a linear-time restatement of the same rule.  ORACLE_BRIDGE_CHECK=1 runs both on every call
    // up to 2000 nodes and aborts on a difference (tests/test_oracle_kat.py does that on a dense parameter set).
    // ORACLE_BRIDGE_RULE = balanced (default) | first | last | least_balanced picks another bridge instead (position in the sorted
    // component edge list / the most lopsided one): tests/test_bridge_rule.py runs the oracle under every rule and asserts that
    // _sv.txt and the satisfied discordant edges do not depend on the choice Boost would have made.
    static int BridgeRule() {
        static const int rule = []() {
            const char* v = std::getenv("ORACLE_BRIDGE_RULE");
            if (!v || !std::strcmp(v, "balanced")) return 0;
            if (!std::strcmp(v, "first")) return 1;
            if (!std::strcmp(v, "last")) return 2;
            if (!std::strcmp(v, "least_balanced")) return 3;
            std::fprintf(stderr, "ORACLE_BRIDGE_RULE: unknown rule '%s'\n", v); std::abort();
        }();
        return rule;
    }
    static int BridgeSplit(int n, const std::vector<std::pair<int, int>>& edges, std::vector<bool>& parity) {
        static const bool check = std::getenv("ORACLE_BRIDGE_CHECK") != nullptr;
        if (BridgeRule() != 0) return BridgeSplitChains(n, edges, parity, BridgeRule());
        if (n > 64 && !(check && n <= 2000)) return BridgeSplitChains(n, edges, parity);
        const int w = BridgeSplitBrute(n, edges, parity);
        if (check) {
            std::vector<bool> p2;
            const int w2 = BridgeSplitChains(n, edges, p2);
            if (w2 != w || (w == 1 && p2 != parity)) { std::fprintf(stderr, "ORACLE_BRIDGE_CHECK: the two bridge searches disagree (n=%d)\n", n); std::abort(); }
        }
        return w;
    }
    // Same rule in O(n + m): bridges by chain decomposition (Schmidt 2013: DFS tree, then every back edge walks up from
    // its lower end until it meets a visited vertex; tree edges never walked are the bridges), side sizes from DFS
    // subtree sizes.  `parity` = the side that contains the bridge's second endpoint, as in the brute force.
    static int BridgeSplitChains(int n, const std::vector<std::pair<int, int>>& edges, std::vector<bool>& parity, int rule = 0) {
        const int m = (int)edges.size();
        std::vector<std::vector<std::pair<int, int>>> adj(n);  // (neighbour, edge id)
        for (int i = 0; i < m; i++) { adj[edges[i].first].push_back({edges[i].second, i}); adj[edges[i].second].push_back({edges[i].first, i}); }
        std::vector<int> par(n, -1), pedge(n, -1), order, tin(n, -1), sub(n, 1);
        std::vector<size_t> it(n, 0);
        std::vector<int> st(1, 0);
        tin[0] = 0; order.push_back(0);
        while (!st.empty()) {  // (the component is connected)
            int x = st.back();
            if (it[x] < adj[x].size()) {
                auto [y, id] = adj[x][it[x]++];
                if (tin[y] < 0) { tin[y] = (int)order.size(); order.push_back(y); par[y] = x; pedge[y] = id; st.push_back(y); }
            } else st.pop_back();
        }
        for (int k = n - 1; k > 0; k--) sub[par[order[k]]] += sub[order[k]];
        std::vector<char> vis(n, 0), walked(m, 0);
        for (int x : order)
            for (auto [y, id] : adj[x]) {
                if (id == pedge[x] || id == pedge[y] || tin[y] < tin[x]) continue;  // back edges, seen from their upper end x
                walked[id] = 1; vis[x] = 1;
                for (int z = y; !vis[z]; z = par[z]) { vis[z] = 1; walked[pedge[z]] = 1; }
            }
        int bestbal = -1, beste = -1;
        for (int i = 0; i < m; i++) {
            if (walked[i] || edges[i].first == edges[i].second) continue;
            const int child = pedge[edges[i].first] == i ? edges[i].first : edges[i].second;
            if (pedge[child] != i) continue;  // (cannot happen: an unwalked non-loop edge is a tree edge)
            const int bal = std::abs(n - 2 * sub[child]);
            if (rule == 1) { if (beste < 0) beste = i; }                                        // first bridge of the edge list
            else if (rule == 2) beste = i;                                                      // last
            else if (rule == 3) { if (bestbal < 0 || bal > bestbal) { bestbal = bal; beste = i; } }  // the most lopsided
            else if (bestbal < 0 || bal < bestbal) { bestbal = bal; beste = i; }
        }
        parity.assign(n, false);
        if (beste < 0) return 2;
        const int child = pedge[edges[beste].first] == beste ? edges[beste].first : edges[beste].second;
        const int lo = tin[child], hi = tin[child] + sub[child];  // the subtree is a contiguous range of DFS entry times
        const bool v_inside = tin[edges[beste].second] >= lo && tin[edges[beste].second] < hi;
        for (int x = 0; x < n; x++) parity[x] = (tin[x] >= lo && tin[x] < hi) == v_inside;
        return 1;
    }
    static int BridgeSplitBrute(int n, const std::vector<std::pair<int, int>>& edges, std::vector<bool>& parity) {
        std::vector<std::vector<int>> adj(n);
        std::map<std::pair<int, int>, int> mult;
        for (auto& e : edges) { adj[e.first].push_back(e.second); adj[e.second].push_back(e.first); mult[std::minmax(e.first, e.second)]++; }
        int bestbal = -1;
        parity.assign(n, false);
        for (size_t ei = 0; ei < edges.size(); ei++) {
            int u = edges[ei].first, v = edges[ei].second;
            if (u == v || mult[std::minmax(u, v)] != 1) continue;
            std::vector<bool> seen(n, false);
            std::vector<int> st(1, v);
            seen[v] = true;
            int cnt = 0;
            while (!st.empty()) {
                int x = st.back(); st.pop_back(); cnt++;
                for (int y : adj[x]) {
                    if ((x == u && y == v) || (x == v && y == u)) continue;  // the candidate edge (its pair has multiplicity 1)
                    if (!seen[y]) { seen[y] = true; st.push_back(y); }
                }
            }
            if (seen[u]) continue;  // still connected: not a bridge
            int bal = std::abs(n - 2 * cnt);
            if (bestbal < 0 || bal < bestbal) { bestbal = bal; parity = seen; }
        }
        return bestbal < 0 ? 2 : 1;  // "2" stands for any cut value above 1
    }

    // src/SegmentGraph.cpp:3264-3451
    std::vector<int> MincutRecursion(std::map<int, int> CompNodes, std::vector<Edge_t> CompEdges) {
        if (CompNodes.size() == 1) return std::vector<int>(1, CompNodes.begin()->first + 1);
        if (CompNodes.size() < 20) return SolveWhole(CompNodes, CompEdges);
        std::vector<std::pair<int, int>> edges;
        for (const Edge_t& e : CompEdges) edges.push_back(std::make_pair(CompNodes[e.Ind1], CompNodes[e.Ind2]));
        std::vector<bool> parities;
        int w = BridgeSplit((int)CompNodes.size(), edges, parities);
        if (w > 1) return SolveWhole(CompNodes, CompEdges);
        stats.mincut_splits++;
        std::map<int, int> VertexParty1, VertexParty2;
        int count1 = 0, count2 = 0;
        for (auto& kv : CompNodes) {
            if (parities[kv.second]) VertexParty1[kv.first] = count1++;
            else VertexParty2[kv.first] = count2++;
        }
        std::vector<Edge_t> EdgesParty1, EdgesParty2;
        Edge_t EdgeMiddle;
        for (const Edge_t& e : CompEdges) {
            bool p1 = parities[CompNodes[e.Ind1]], p2 = parities[CompNodes[e.Ind2]];
            if (p1 && p2) EdgesParty1.push_back(e);
            else if (!p1 && !p2) EdgesParty2.push_back(e);
            else EdgeMiddle = e;
        }
        std::vector<int> BestParty1 = MincutRecursion(VertexParty1, EdgesParty1);
        std::vector<int> BestParty2 = MincutRecursion(VertexParty2, EdgesParty2);
        auto scan = [&](const std::vector<int>& B, int& median, bool& ispositive, bool& ishead) {
            std::vector<int> tmp;
            for (int x : B) {
                tmp.push_back(std::abs(x));
                if (std::abs(x) == EdgeMiddle.Ind1 + 1) { ispositive = (x > 0); ishead = EdgeMiddle.Head1; }
                else if (std::abs(x) == EdgeMiddle.Ind2 + 1) { ispositive = (x > 0); ishead = EdgeMiddle.Head2; }
            }
            std::sort(tmp.begin(), tmp.end());
            median = tmp[((int)tmp.size() - 1) / 2];
        };
        int median1 = 0, median2 = 0;
        bool ispositive1 = false, ishead1 = false, ispositive2 = false, ishead2 = false;
        scan(BestParty1, median1, ispositive1, ishead1);
        scan(BestParty2, median2, ispositive2, ishead2);
        auto flip = [](std::vector<int>& B) { std::reverse(B.begin(), B.end()); for (int& x : B) x = -x; };
        std::vector<int> BestOrder;
        if (median1 < median2) {
            if (ispositive1 == ishead1) flip(BestParty1);   // the bridge end must face right in the left block
            if (ispositive2 != ishead2) flip(BestParty2);   // and face left in the right block
            BestOrder = BestParty1;
            BestOrder.insert(BestOrder.end(), BestParty2.begin(), BestParty2.end());
        } else {
            if (ispositive2 == ishead2) flip(BestParty2);
            if (ispositive1 != ishead1) flip(BestParty1);
            BestOrder = BestParty2;
            BestOrder.insert(BestOrder.end(), BestParty1.begin(), BestParty1.end());
        }
        return BestOrder;
    }
};

}  // namespace oracle
