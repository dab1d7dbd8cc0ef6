"""CPU: the oracle's three exact ordering solvers against each other (test infrastructure checking itself before it checks the
product).  The ordering problem is GenerateILP's model (src/SegmentGraph.cpp:3763-3983); GLPK is absent, so the optimum is
defined canonically -- max value, then smallest orientation mask, then the lexicographically smallest sequence -- and found by
  brute : every signed permutation (n <= 8 here),
  bnb   : orientation branch-and-bound + whole-component subset DP (n <= 26),
  wide  : orientation search + Tarjan components + per-component subset DP (n <= 128, components above 26 nodes)."""
import random

import oracle_util as ou


def test_brute_force_bnb_and_wide_solver_agree_up_to_8_nodes(built):
    rng = random.Random(7)
    for trial in range(40):
        n = rng.randrange(2, 8) if trial < 34 else 8
        pairs = [(u, v) for u in range(n) for v in range(u + 1, n)]
        rng.shuffle(pairs)
        edges = []
        for u, v in pairs[: rng.randrange(1, min(len(pairs), 2 * n) + 1)]:
            for _ in range(rng.choice([1, 1, 2])):
                edges.append((u, v, rng.randrange(2), rng.randrange(2), rng.randrange(1, 9)))
        b = ou.solve_order(built, "brute", n, edges)
        assert ou.solve_order(built, "bnb", n, edges) == b, (n, edges)
        assert ou.solve_order(built, "wide", n, edges) == b, (n, edges)
        assert ou.order_value(n, edges, b[1], b[2]) == b[0]


def test_bnb_and_wide_solver_agree_on_9_to_19_nodes(built):
    rng = random.Random(8)
    for trial in range(30):
        n = rng.randrange(9, 20)
        edges = ou.random_order_problem(rng, n, conflict=0.4)
        b = ou.solve_order(built, "bnb", n, edges)
        assert ou.solve_order(built, "wide", n, edges) == b, (n, edges)
        assert ou.order_value(n, edges, b[1], b[2]) == b[0]


def test_wide_solver_beats_the_identity_order_on_large_components(built):
    """the search is exact within a work budget and gives up otherwise (the reference gives GLPK 300 s and keeps the identity order
    when it fails, SegmentGraph.cpp:3964,3984): whatever it returns must be consistent, and most component-like instances solve"""
    rng = random.Random(9)
    solved = 0
    for n in (27, 40, 64, 100, 128, 33, 48, 80):
        edges = ou.random_order_problem(rng, n, conflict=0.2, extra=5)
        w = ou.solve_order(built, "wide", n, edges)
        if w is None:
            continue
        solved += 1
        assert ou.order_value(n, edges, w[1], w[2]) == w[0]
        assert w[0] >= ou.order_value(n, edges, 0, list(range(n)))
        assert not (w[1] >> (n - 1)) & 1  # canonical: the last node stays forward
    assert solved >= 5


def test_edge_search_gives_the_optimum_value(built):
    """KeepDrop (round 6): the optimum as the heaviest set of edges that can be satisfied together, searched over the EDGES (keep / drop by
    descending weight; orientation parities in a union-find, precedence arcs kept acyclic) -- a road that shares nothing with the orientation
    searches.  Its value equals theirs on problems of every conflict rate, and the orientation search started from that value returns the
    same canonical solution as the one started from nothing."""
    rng = random.Random(10)
    for trial in range(40):
        n = rng.randrange(3, 20)
        edges = ou.random_order_problem(rng, n, conflict=rng.choice([0.05, 0.2, 0.4, 0.7]))
        b = ou.solve_order(built, "bnb", n, edges)
        assert ou.solve_order_value(built, n, edges) == b[0], (n, edges)
        assert ou.solve_order(built, "wide_seeded", n, edges) == b, (n, edges)
    for n in (27, 40, 64, 33):
        edges = ou.random_order_problem(rng, n, conflict=0.2, extra=5)
        w = ou.solve_order(built, "wide", n, edges)
        if w is not None:
            assert ou.solve_order(built, "wide_seeded", n, edges) == w, (n, edges)
            assert ou.solve_order_value(built, n, edges) == w[0]


def test_the_75_node_component_of_the_bwa_sample_is_solved(built):
    """tests/golden/order/bwa_c3_1m_component_75.txt: the bridge-free component of `gen_synth_bam --config C3 --bwa --records 1000000` that both
    orientation searches of round 5 gave up on (identity order kept on both sides, eleven SV rows lost on both sides, "identical").  The edge
    search finds its optimum -- 8985 of 9024: six weight-1 backbone edges and two light concordant ones given up -- and the orientation search
    started there finds the canonical solution"""
    from pathlib import Path

    text = (Path(__file__).resolve().parent / "golden" / "order" / "bwa_c3_1m_component_75.txt").read_text().split()
    n, m = int(text[0]), int(text[1])
    edges = [tuple(int(x) for x in text[2 + 5 * i: 7 + 5 * i]) for i in range(m)]
    assert n == 75 and len(edges) == 99
    assert ou.solve_order_value(built, n, edges) == 8985
    w = ou.solve_order(built, "wide_seeded", n, edges)
    assert w is not None and w[0] == 8985
    assert ou.order_value(n, edges, w[1], w[2]) == 8985
    assert sorted(w[2]) == list(range(n)) and not (w[1] >> (n - 1)) & 1
