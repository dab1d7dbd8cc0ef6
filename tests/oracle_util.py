"""Helpers around the CPU oracle (test infrastructure): run it with --dump and parse the stage files."""
import subprocess
from pathlib import Path


def run_oracle(build_dir, prefix, outdir, *flags, check=True):
    outdir = Path(outdir)
    dump = outdir / "dump"
    dump.mkdir(parents=True, exist_ok=True)
    cmd = [str(Path(build_dir) / "squid_oracle"), "-b", f"{prefix}.bam", "-c", f"{prefix}.chim.bam", "-o", str(outdir / "oracle"), "--dump", str(dump), *flags]
    rc = subprocess.call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL if not check else None)
    if check and rc:
        raise subprocess.CalledProcessError(rc, cmd)
    return outdir / "oracle_sv.txt", dump


def read_nodes(path):
    rows = []
    for line in Path(path).read_text().splitlines():
        if line.startswith("#"):
            continue
        f = line.split("\t")
        rows.append((int(f[0]), int(f[1]), int(f[2]), int(f[3]), float.fromhex(f[4])) + ((int(f[5]),) if len(f) > 5 else ()))
    return rows


def read_edges(path):
    rows = []
    for line in Path(path).read_text().splitlines():
        if line.startswith("#"):
            continue
        rows.append(tuple(int(x) for x in line.split("\t")))
    return rows


def read_orders(path):
    out = []
    for line in Path(path).read_text().splitlines():
        if line.startswith("#"):
            continue
        out.append([int(x) for x in line.split("\t")[1].split(",")])
    return out


def read_breakpoints(path):
    out = []
    for line in Path(path).read_text().splitlines():
        if line.startswith("#"):
            continue
        f = line.split("\t")
        bps = []
        for item in f[4:]:
            bp, sup = item.split(":")
            b1, b2 = bp.split(",")
            s1, s2 = sup.split(",")
            bps.append((-1 if b1 == "-" else int(b1), -1 if b2 == "-" else int(b2), int(s1), int(s2)))
        out.append(bps)
    return out


def solve_order(build_dir, method, n, edges):
    """one ordering problem through the oracle's solvers (squid_oracle --solve-order brute|bnb|wide): edges = [(u, v, head_u,
    head_v, w)], u < v.  Returns (value, orientation mask as a Python int, left-to-right list of local nodes), or None when the
    wide solver gave up."""
    text = f"{n} {len(edges)}\n" + "".join(" ".join(str(int(x)) for x in e) + "\n" for e in edges)
    out = subprocess.run([str(Path(build_dir) / "squid_oracle"), "--solve-order", method], input=text, capture_output=True, text=True, check=True).stdout.split()
    if out[0] == "FAILED":
        return None
    seq = [int(x) for x in out[1:]]
    return int(out[0]), sum(1 << (-x - 1) for x in seq if x < 0), [abs(x) - 1 for x in seq]


def solve_order_value(build_dir, n, edges):
    """the optimum value of one ordering problem by the oracle's edge search (KeepDrop, squid_oracle --solve-order kdvalue)"""
    text = f"{n} {len(edges)}\n" + "".join(" ".join(str(int(x)) for x in e) + "\n" for e in edges)
    out = subprocess.run([str(Path(build_dir) / "squid_oracle"), "--solve-order", "kdvalue"], input=text, capture_output=True, text=True, check=True).stdout.split()
    return None if out[0] == "FAILED" else int(out[0])


def order_value(n, edges, mask, order):
    """weight of the edges satisfied by (orientation mask, left-to-right order): the four head patterns of GenerateILP
    (src/SegmentGraph.cpp:3763-3983) as restated in oracle/o_order.h EdgeSatisfied"""
    pos = {v: i for i, v in enumerate(order)}
    tot = 0
    for u, v, hu, hv, w in edges:
        yu, yv = not (mask >> u) & 1, not (mask >> v) & 1
        if not hu and hv:
            ok, ufirst = yu == yv, yu
        elif not hu and not hv:
            ok, ufirst = yu != yv, yu
        elif hu and hv:
            ok, ufirst = yu != yv, yv
        else:
            ok, ufirst = yu == yv, not yu
        if ok and (pos[u] < pos[v]) == bool(ufirst):
            tot += w
    return tot


def random_order_problem(rng, n, conflict=0.3, extra=None):
    """a component-like ordering problem: a tail->head backbone between neighbours (as MincutRecursion adds, :3275-3286), random
    further edges of all four head patterns, and -- with probability `conflict` per edge -- a parallel edge with another pattern"""
    edges = []
    for k in range(n - 1):
        if rng.random() < 0.85:
            edges.append((k, k + 1, 0, 1, rng.randrange(1, 6)))
    for _ in range(extra if extra is not None else rng.randrange(1, n + 1)):
        u, v = sorted(rng.sample(range(n), 2))
        hu, hv = rng.randrange(2), rng.randrange(2)
        edges.append((u, v, hu, hv, rng.randrange(1, 40)))
        if rng.random() < conflict:
            edges.append((u, v, 1 - hu, hv, rng.randrange(1, 40)))
    return edges


def planted_order_problem(rng, n, events=3, noise=2):
    """a component as SQUID sees one after a few rearrangements: the genome backbone (tail->head between neighbours, light) plus
    heavy discordant edges for the new adjacencies of a planted arrangement (identity with `events` block inversions / moves) and
    a few light noise edges.  All heavy edges can be satisfied together, as in a real rearranged genome."""
    arr = [(k, 1) for k in range(n)]
    for _ in range(events):
        i, j = sorted(rng.sample(range(n + 1), 2))
        if j - i < 1:
            continue
        block = arr[i:j]
        rest = arr[:i] + arr[j:]
        if rng.random() < 0.5:
            block = [(k, -s) for k, s in reversed(block)]
        at = rng.randrange(len(rest) + 1)
        arr = rest[:at] + block + rest[at:]
    edges = [(k, k + 1, 0, 1, rng.randrange(1, 4)) for k in range(n - 1)]
    for (a, sa), (b, sb) in zip(arr, arr[1:]):
        if sa == 1 and sb == 1 and b == a + 1:
            continue  # still the genome adjacency
        ha, hb = int(sa < 0), int(sb > 0)   # right end of a, left end of b (head = segment start)
        edges.append((a, b, ha, hb, 8 * rng.randrange(3, 12)) if a < b else (b, a, hb, ha, 8 * rng.randrange(3, 12)))
    for _ in range(noise):
        u, v = sorted(rng.sample(range(n), 2))
        edges.append((u, v, rng.randrange(2), rng.randrange(2), rng.randrange(1, 4)))
    return edges
