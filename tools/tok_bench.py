#!/usr/bin/env python3
"""the token pass and the resolve of the first blocks of a BGZF file, each kernel alone on the device (sq_debug_token_bench).
usage: tok_bench.py file.bam [blocks] [reps] [variant ...]      variants: 2 = k_inflate_tok2, CH*100+PB = k_inflate_spec<CH, PB>"""
import ctypes as C, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import squid_amd
path = sys.argv[1]
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
variants = [int(v) for v in sys.argv[4:]] or [51211, 2]
lib = squid_amd.load_library()
lib.sq_debug_token_bench.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
with squid_amd.Context() as ctx:
    for v in variants:
        out = (C.c_double * 7)()
        rc = lib.sq_debug_token_bench(ctx.h, path.encode(), v, blocks, reps, 1, out)
        if rc:
            print(f"variant {v}: rc {rc} {lib.sq_last_error(ctx.h)}")
            continue
        tok, res, ib, cb, nb, nt, bad = list(out)
        print(f"variant {v:6d}: {int(nb)} blocks, {ib / 1e9:.3f} GB inflated, {cb / 1e9:.3f} GB file | token pass {tok:8.3f} ms = {ib / tok / 1e6:7.1f} GB/s of inflated bytes | resolve {res:7.3f} ms = {ib / res / 1e6:7.1f} GB/s | {nt / ib:.3f} tokens/byte | differ from zlib: {int(bad)}", flush=True)
