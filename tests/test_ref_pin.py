"""CPU: the aligned-block type of the oracle and of the product against the REAL
src/SingleBamRec.h, compiled from where it lies (oracle/Makefile target `ref` -> oracle/_ref/ref_singlebamrec; the header
includes only std headers, so it builds without BamTools/GLPK/Boost).  Pinned: operator<, operator>, operator==, Same,
CompReadPos (src/SingleBamRec.h:39-58) and the permutations libstdc++'s std::sort produces with operator<
(bamdiscordant, SegmentGraph.cpp:264 -- ties are NOT ordered, ledger B8) and with CompReadPos (ReadRec.cpp:144-145).
On the GPU box the reference tree is absent and the prebuilt binary travels with the snapshot."""
import ctypes as C
import random
import subprocess

import numpy as np
import pytest

import squid_amd

REF = squid_amd.ROOT / "oracle" / "_ref" / "ref_singlebamrec"


def _cases():
    rng = random.Random(20180105)
    out = []
    # many ties on (RefID, RefPos) and on ReadPos: the introsort tie order is part of what is pinned
    for n, nref, npos in [(1, 1, 1), (2, 1, 1), (17, 2, 3), (64, 3, 5), (200, 4, 40), (256, 25, 1000), (256, 1, 2)]:
        out.append([(rng.randrange(nref), rng.randrange(npos), rng.randrange(0, 6) * 10, rng.randrange(1, 4) * 25, rng.randrange(1, 4) * 25, rng.randrange(2), rng.randrange(2))
                    for _ in range(n)])
    # duplicates of whole blocks (Same must see every field except MapQual)
    base = [(1, 100, 0, 50, 50, 0, 1)] * 5 + [(1, 100, 0, 50, 50, 1, 1), (1, 100, 0, 50, 50, 0, 0), (1, 100, 1, 50, 50, 0, 1), (1, 100, 0, 51, 50, 0, 1), (1, 100, 0, 50, 51, 0, 1)]
    out.append(base)
    # negative ids / positions (unplaced records carry -1)
    out.append([(-1, -1, 0, 0, 0, 0, 0), (0, 0, 0, 1, 1, 0, 1), (-1, 5, 3, 2, 2, 1, 0), (2, -7, 9, 3, 3, 1, 1), (0, 0, 0, 1, 1, 0, 1)])
    return out


def _run(binary, blocks):
    text = "".join(" ".join(str(x) for x in b) + "\n" for b in blocks)
    out = subprocess.run([str(binary)], input=text, capture_output=True, text=True, check=True).stdout
    return dict(line.split(" ", 1) if " " in line else (line, "") for line in out.splitlines())


def _product(blocks):
    lib = squid_amd.load_library()
    n = len(blocks)
    f = np.ascontiguousarray(np.array(blocks, dtype=np.int32).reshape(-1))
    rel = np.zeros(5 * n * n, np.uint8)
    pp, pr = np.zeros(n, np.int32), np.zeros(n, np.int32)
    I32P, U8P = C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
    lib.sq_debug_blocks.argtypes = [C.c_int32, I32P, U8P, I32P, I32P]
    assert lib.sq_debug_blocks(n, f.ctypes.data_as(I32P), rel.ctypes.data_as(U8P), pp.ctypes.data_as(I32P), pr.ctypes.data_as(I32P)) == 0
    m = lambda k: "".join("1" if x else "0" for x in rel[k * n * n:(k + 1) * n * n])
    return {"lt": m(0), "gt": m(1), "eq": m(2), "same": m(3), "readpos": m(4), "sort_pos": " ".join(map(str, pp)), "sort_readpos": " ".join(map(str, pr))}


@pytest.mark.skipif(not REF.exists(), reason="oracle/_ref/ref_singlebamrec not built (make -C oracle ref)")
@pytest.mark.parametrize("case", range(len(_cases())))
def test_block_comparators_and_sort_orders_match_reference_header(built, case):
    blocks = _cases()[case]
    want = _run(REF, blocks)
    assert set(want) == {"lt", "gt", "eq", "same", "readpos", "sort_pos", "sort_readpos"}
    assert _run(built / "oracle_singlebamrec", blocks) == want      # the oracle's restatement
    assert _product(blocks) == want                                  # the product's host block type (no GPU needed)
