"""squid_amd -- MI355X-native SQUID hot path (BAM -> segment graph -> ordering -> _sv.txt).

Thin ctypes binding over the C ABI of ``build/libsquid_hip.so`` (``include/squid_hip.h``).  The Python layer
holds no algorithm: it exists so that tests and ``bench.py`` can drive the same entry points the ``squid``
command line uses.  Importing the package does not need a GPU; creating a :class:`Context` does, and fails
loudly without one (there is no CPU path for the GPU stages).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (the library asks for the same when it is loaded; this also covers a torch that makes the first HIP call)
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
BUILD = ROOT / "build"
LIB_PATH = Path(os.environ["SQUID_LIB"]) if os.environ.get("SQUID_LIB") else BUILD / "libsquid_hip.so"  # (SQUID_LIB: another build of the same library, for A/B runs)

EXPORTS = [
    "sq_default_params", "sq_create", "sq_destroy", "sq_strerror", "sq_last_error", "sq_set_references",
    "sq_ingest_chimeric", "sq_chim_contains", "sq_ingest_concordant", "sq_ingest_concordant_bam", "sq_read_header", "sq_ingest_chimeric_file",
    "sq_ingest_concordant_file", "sq_build_graph", "sq_graph_view", "sq_order", "sq_call_sv", "sq_breakpoints",
    "sq_set_shard", "sq_exchange_pack", "sq_exchange_unpack", "sq_get_timing", "sq_timing_accumulate", "sq_reset", "sq_ingest_files", "sq_stage_bam", "sq_clear_records", "sq_set_source", "sq_save_records", "sq_load_records", "sq_get_counts", "sq_debug_download", "sq_debug_bp_support", "sq_debug_order", "sq_debug_blocks", "sq_drop_file_cache",
    "sq_total_order", "sq_set_allgather", "sq_rccl_unique_id", "sq_rccl_init", "sq_rccl_attach", "sq_exchange", "sq_exchange_stats",
    "sq_rccl_available", "sq_rccl_release", "sq_debug_rccl_selftest", "sq_debug_token_bench", "sq_ingest_bwa_file", "sq_junction_sequences", "sq_release_reader_buffers", "sq_keep_host_memory", "sq_keep_stage_graphs",
]


class SqParams(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device", C.c_int32), ("phred_type", C.c_int32), ("max_lowphred_len", C.c_int32),
                ("min_phred", C.c_int32), ("min_mapqual", C.c_int32), ("concord_dist_pos", C.c_int32), ("concord_dist_idx", C.c_int32),
                ("min_edge_weight", C.c_int32), ("discordant_ratio", C.c_double), ("max_allowed_degree", C.c_int32),
                ("rank", C.c_int32), ("world_size", C.c_int32)]


_P32 = C.POINTER(C.c_int32)
_PU8 = C.POINTER(C.c_uint8)


class SqAlnBatch(C.Structure):
    _fields_ = [("n_rec", C.c_int64), ("n_blk", C.c_int64), ("refid", _P32), ("pos", _P32), ("mate_refid", _P32), ("mate_pos", _P32), ("end_pos", _P32),
                ("flag", C.POINTER(C.c_uint16)), ("mapq", _PU8), ("aux", _PU8), ("totlen", C.POINTER(C.c_uint16)), ("blk_off", C.POINTER(C.c_uint32)),
                ("b_refpos", _P32), ("b_matchref", _P32), ("b_readpos", C.POINTER(C.c_uint16)), ("b_matchread", C.POINTER(C.c_uint16)),
                ("name_off", C.POINTER(C.c_uint32)), ("name_blob", C.c_char_p)]


class SqGraph(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("n_edges", C.c_int32), ("chr", _P32), ("pos", _P32), ("len", _P32), ("support", _P32), ("label", _P32),
                ("avgdepth", C.POINTER(C.c_double)), ("ind1", _P32), ("ind2", _P32), ("weight", _P32), ("groupweight", _P32), ("head1", _PU8), ("head2", _PU8)]


class SqOrders(C.Structure):
    _fields_ = [("n_components", C.c_int32), ("comp_off", _P32), ("nodes", _P32)]


class SqSvTable(C.Structure):
    _fields_ = [("n_rows", C.c_int32), ("chr1", _P32), ("start1", _P32), ("end1", _P32), ("chr2", _P32), ("start2", _P32), ("end2", _P32),
                ("score", _P32), ("sup1", _P32), ("sup2", _P32), ("strand1_minus", _PU8), ("strand2_minus", _PU8)]


class SqBpTable(C.Structure):
    _fields_ = [("n_edges", C.c_int32), ("bp_off", _P32), ("bp1", _P32), ("bp2", _P32), ("sup1", _P32), ("sup2", _P32)]


class SqTiming(C.Structure):
    _fields_ = [("n", C.c_int32), ("names", C.POINTER(C.c_char_p)), ("ms", C.POINTER(C.c_double)), ("launches", C.POINTER(C.c_int64)),
                ("bytes", C.POINTER(C.c_double)), ("busy_ms", C.POINTER(C.c_double))]


class SqCounts(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("n_concordant", "n_blocks", "n_chimeric_records", "n_chim_fragments", "read_len", "n_kept_p1", "n_break",
                                          "n_kept_p2", "n_raw_edges", "n_unique_edges", "n_order_unsolved", "token_passes_side_by_side", "replay_candidates_checked", "replay_count_mismatches", "chimeric_through_gpu_reader")]


class SquidError(RuntimeError):
    pass


_lib = None


def build(force: bool = False) -> None:
    """Compile every native artefact (hipcc --offload-arch=gfx950 cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", str(ROOT), "clean"])
    subprocess.check_call(["make", "-C", str(ROOT), "-j4", "all"])


def load_library() -> C.CDLL:
    """dlopen build/libsquid_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise SquidError(f"{LIB_PATH} is missing: run `make` (or __graft_entry__.build()) first; there is no pure-Python path")
        lib = C.CDLL(str(LIB_PATH))
        lib.sq_strerror.restype = C.c_char_p
        lib.sq_last_error.restype = C.c_char_p
        lib.sq_last_error.argtypes = [C.c_void_p]
        lib.sq_create.argtypes = [C.POINTER(SqParams), C.POINTER(C.c_void_p)]
        lib.sq_destroy.argtypes = [C.c_void_p]
        lib.sq_destroy.restype = None
        lib.sq_set_references.argtypes = [C.c_void_p, C.c_int32, _P32]
        lib.sq_ingest_chimeric_file.argtypes = [C.c_void_p, C.c_char_p]
        lib.sq_ingest_concordant_file.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
        lib.sq_ingest_files.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int32]
        lib.sq_stage_bam.argtypes = [C.c_void_p, C.c_char_p]
        lib.sq_keep_stage_graphs.argtypes = [C.c_void_p, C.c_int32]
        lib.sq_clear_records.argtypes = [C.c_void_p]
        lib.sq_release_reader_buffers.argtypes = [C.c_void_p]
        lib.sq_set_source.argtypes = [C.c_void_p, C.c_char_p]
        lib.sq_save_records.argtypes = [C.c_void_p, C.c_char_p]
        lib.sq_load_records.argtypes = [C.c_void_p, C.c_char_p]
        lib.sq_read_header.argtypes = [C.c_char_p, _P32, _P32, C.c_char_p, C.c_size_t]
        lib.sq_build_graph.argtypes = [C.c_void_p]
        lib.sq_graph_view.argtypes = [C.c_void_p, C.c_int32, C.POINTER(SqGraph)]
        lib.sq_order.argtypes = [C.c_void_p, C.POINTER(SqOrders)]
        lib.sq_call_sv.argtypes = [C.c_void_p, C.POINTER(SqSvTable)]
        lib.sq_breakpoints.argtypes = [C.c_void_p, C.POINTER(SqBpTable)]
        lib.sq_get_timing.argtypes = [C.c_void_p, C.POINTER(SqTiming)]
        lib.sq_get_counts.argtypes = [C.c_void_p, C.POINTER(SqCounts)]
        lib.sq_reset.argtypes = [C.c_void_p]
        lib.sq_chim_contains.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        lib.sq_set_shard.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        lib.sq_exchange_pack.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        lib.sq_exchange_unpack.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64), C.c_int32]
        _lib = lib
    return _lib


def rccl_available() -> bool:
    """librccl can be bound in this process (no communicator is made)"""
    return bool(load_library().sq_rccl_available())


def rccl_unique_id() -> bytes:
    """128 bytes that let world_size ranks join one RCCL communicator (rank 0 makes them, everybody passes them to Context.rccl_init)"""
    buf = C.create_string_buffer(128)
    rc = load_library().sq_rccl_unique_id(buf)
    if rc:
        raise SquidError("sq_rccl_unique_id failed")
    return buf.raw


def keep_host_memory():
    """sq_keep_host_memory: the C allocator keeps what the host stages free (a process that runs sample after sample; process-wide)"""
    rc = load_library().sq_keep_host_memory()
    if rc:
        raise SquidError("sq_keep_host_memory: mallopt refused")


def drop_file_cache():
    """forget the process-wide mapping and block index of the last BAM read (sq_drop_file_cache)"""
    load_library().sq_drop_file_cache()


def read_header(bam_path: str):
    lib = load_library()
    n = C.c_int32(1 << 16)
    lens = (C.c_int32 * (1 << 16))()
    names = C.create_string_buffer(1 << 22)
    rc = lib.sq_read_header(str(bam_path).encode(), C.byref(n), lens, names, len(names))
    if rc:
        raise SquidError(f"cannot read BAM header of {bam_path}: {lib.sq_strerror(rc).decode()}")
    return names.value.decode().split("\n")[: n.value], list(lens[: n.value])


class Context:
    """One GPU context (sq_ctx).  Mirrors the reference's fixed pipeline (src/main.cpp:17-76)."""

    def __init__(self, device: int = 0, star_mapq: bool = True, exchange=None, **params):
        """`exchange`: for a chromosome-sharded context (rank=, world_size=) a callable bytes -> list[bytes] that
        all-gathers one payload per rank (squid_amd.dist.TorchExchange, or a squid_amd.dist.VirtualWorld drives it)."""
        self.exchange = exchange
        self.lib = load_library()
        p = SqParams()
        self.lib.sq_default_params(C.byref(p))
        p.device = device
        if star_mapq and "min_mapqual" not in params:
            p.min_mapqual = 255  # Config.cpp:221-222
        for k, v in params.items():
            setattr(p, k, v)
        self.params = p
        self.h = C.c_void_p()
        rc = self.lib.sq_create(C.byref(p), C.byref(self.h))
        if rc:
            raise SquidError(f"sq_create failed: {self.lib.sq_strerror(rc).decode()} -- the HIP path is mandatory, no CPU fallback")
        self.ref_names: list[str] = []

    def _chk(self, rc: int, what: str):
        if rc:
            raise SquidError(f"{what}: {self.lib.sq_strerror(rc).decode()} ({self.lib.sq_last_error(self.h).decode()})")

    def close(self):
        if self.h:
            self.lib.sq_destroy(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def load(self, bam: str, chim_bam: str, threads: int = 8, shard: tuple | None = None):
        """`shard` = (first_ref, end_ref): keep only the concordant records of those RefIDs (sq_set_shard)"""
        names, lens = read_header(bam)
        self.ref_names = names
        arr = (C.c_int32 * len(lens))(*lens)
        self._chk(self.lib.sq_set_references(self.h, len(lens), arr), "sq_set_references")
        if shard is not None:
            self._chk(self.lib.sq_set_shard(self.h, int(shard[0]), int(shard[1])), "sq_set_shard")
        self._chk(self.lib.sq_ingest_files(self.h, str(chim_bam).encode(), str(bam).encode(), threads), "sq_ingest_files")

    def reset(self):
        self._chk(self.lib.sq_reset(self.h), "sq_reset")

    def stage_bam(self, bam: str):
        """copy the compressed bytes of `bam` into HBM; a later load() of the same path reads them there (sq_stage_bam)"""
        self._chk(self.lib.sq_stage_bam(self.h, str(bam).encode()), "sq_stage_bam")

    def clear_records(self):
        """drop the resident concordant records and all results, keep the device buffers (sq_clear_records)"""
        self._chk(self.lib.sq_clear_records(self.h), "sq_clear_records")

    def keep_stage_graphs(self, on: bool = True):
        """sq_keep_stage_graphs: keep (default) or skip the copies of the intermediate graphs that graph(1..5) hands out"""
        self._chk(self.lib.sq_keep_stage_graphs(self.h, 1 if on else 0), "sq_keep_stage_graphs")

    def release_reader_buffers(self):
        """give back the device / page-locked memory the GPU reader keeps between ingests (sq_release_reader_buffers)"""
        self._chk(self.lib.sq_release_reader_buffers(self.h), "sq_release_reader_buffers")

    def save_records(self, path: str):
        """write the resident concordant records to a cache file (sq_save_records)"""
        self._chk(self.lib.sq_save_records(self.h, str(path).encode()), "sq_save_records")

    def load_cached(self, bam: str, chim_bam: str, cache: str, shard: tuple | None = None):
        """as load(), with the concordant records from a cache file written by save_records (the BAM is only opened
        for its header)"""
        names, lens = read_header(bam)
        self.ref_names = names
        arr = (C.c_int32 * len(lens))(*lens)
        self._chk(self.lib.sq_set_references(self.h, len(lens), arr), "sq_set_references")
        if shard is not None:
            self._chk(self.lib.sq_set_shard(self.h, int(shard[0]), int(shard[1])), "sq_set_shard")
        self._chk(self.lib.sq_ingest_chimeric_file(self.h, str(chim_bam).encode()), "sq_ingest_chimeric_file")
        self._chk(self.lib.sq_set_source(self.h, str(bam).encode()), "sq_set_source")
        self._chk(self.lib.sq_load_records(self.h, str(cache).encode()), "sq_load_records")

    # ---- chromosome-sharded runs: the stage functions pause with SQ_NEED_EXCHANGE (include/squid_hip.h)
    NEED_EXCHANGE = 1

    def exchange_pack(self) -> bytes:
        buf, n = C.c_void_p(), C.c_int64()
        self._chk(self.lib.sq_exchange_pack(self.h, C.byref(buf), C.byref(n)), "sq_exchange_pack")
        return C.string_at(buf, n.value)

    def exchange_unpack(self, parts: list):
        sizes = (C.c_int64 * len(parts))(*[len(x) for x in parts])
        self._chk(self.lib.sq_exchange_unpack(self.h, b"".join(parts), sizes, len(parts)), "sq_exchange_unpack")

    def _run_stage(self, step, what: str):
        while True:
            rc = step()
            if rc != self.NEED_EXCHANGE:
                self._chk(rc, what)
                return
            if self.exchange is None:
                raise SquidError(f"{what}: sharded context without an exchange callable")
            if self.exchange == "native":  # the library all-gathers itself (sq_exchange: RCCL, or the installed all-gather)
                self._chk(self.lib.sq_exchange(self.h), "sq_exchange")
            else:
                self.exchange_unpack(self.exchange(self.exchange_pack()))

    # ---- sq_exchange: the transport of a sharded context
    def rccl_init(self, id128: bytes):
        """join the RCCL communicator made from rank 0's id (rccl_unique_id()); the exchanges then run inside the library"""
        self._chk(self.lib.sq_rccl_init(self.h, id128), "sq_rccl_init")
        self.exchange = "native"

    def load_bwa(self, bam: str, threads: int = 8):
        """`squid --bwa`: one BAM file, no chimeric file (sq_ingest_bwa_file); pass min_mapqual (default 1) and star_mapq=False to Context"""
        names, lens = read_header(bam)
        self.ref_names = names
        arr = (C.c_int32 * len(lens))(*lens)
        self._chk(self.lib.sq_set_references(self.h, len(lens), arr), "sq_set_references")
        self._chk(self.lib.sq_ingest_bwa_file(self.h, str(bam).encode(), threads), "sq_ingest_bwa_file")

    def rccl_release(self):
        self._chk(self.lib.sq_rccl_release(self.h), "sq_rccl_release")
        self.exchange = None

    def rccl_selftest(self):
        """the library's RCCL transport on this context's device with a communicator of one rank (see sq_debug_rccl_selftest)"""
        self._chk(self.lib.sq_debug_rccl_selftest(self.h), "sq_debug_rccl_selftest")

    def set_allgather(self, fn):
        """install a fixed-size all-gather of host buffers: fn(send: bytes) -> bytes of world_size pieces in rank order"""
        FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)

        def tramp(_user, send, nbytes, recv):
            try:
                out = fn(C.string_at(send, nbytes))
                if len(out) != nbytes * max(int(self.params.world_size), 1):  # (never write past the C buffer)
                    return -1
                C.memmove(recv, out, len(out))
                return 0
            except Exception:  # noqa: BLE001 -- a Python exception must not unwind through the C caller
                return -3
        self._allgather_cb = FN(tramp)  # (keep the trampoline alive as long as the context)
        self.lib.sq_set_allgather.argtypes = [C.c_void_p, FN, C.c_void_p]
        self._chk(self.lib.sq_set_allgather(self.h, self._allgather_cb, None), "sq_set_allgather")
        self.exchange = "native"

    def exchange_stats(self) -> tuple:
        n, b = C.c_int64(), C.c_int64()
        self._chk(self.lib.sq_exchange_stats(self.h, C.byref(n), C.byref(b)), "sq_exchange_stats")
        return n.value, b.value

    def build_graph_step(self) -> int:
        """one call of sq_build_graph: 0 = done, NEED_EXCHANGE = all-gather exchange_pack() into exchange_unpack()"""
        rc = self.lib.sq_build_graph(self.h)
        if rc < 0:
            self._chk(rc, "sq_build_graph")
        return rc

    def build_graph(self):
        self._run_stage(lambda: self.lib.sq_build_graph(self.h), "sq_build_graph")

    def graph(self, stage: int = 0) -> dict:
        g = SqGraph()
        self._chk(self.lib.sq_graph_view(self.h, stage, C.byref(g)), "sq_graph_view")
        n, m = g.n_nodes, g.n_edges
        return {
            "nodes": [(g.chr[i], g.pos[i], g.len[i], g.support[i], g.avgdepth[i], g.label[i]) for i in range(n)],
            "edges": [(g.ind1[i], g.head1[i], g.ind2[i], g.head2[i], g.weight[i], g.groupweight[i]) for i in range(m)],
        }

    def order(self) -> list[list[int]]:
        o = SqOrders()
        self._chk(self.lib.sq_order(self.h, C.byref(o)), "sq_order")
        return [[o.nodes[j] for j in range(o.comp_off[k], o.comp_off[k + 1])] for k in range(o.n_components)]

    def total_order(self) -> list[list[int]]:
        """the components stitched into whole new chromosomes (sq_total_order: what -TO prints)"""
        o = SqOrders()
        self._chk(self.lib.sq_total_order(self.h, C.byref(o)), "sq_total_order")
        return [[o.nodes[j] for j in range(o.comp_off[k], o.comp_off[k + 1])] for k in range(o.n_components)]

    def order_sizes(self):
        """sq_order, returning only the number of nodes of every component (numpy array)"""
        import numpy as np

        o = SqOrders()
        self._chk(self.lib.sq_order(self.h, C.byref(o)), "sq_order")
        if o.n_components <= 0:
            return np.zeros(0, np.int64)
        off = np.ctypeslib.as_array(o.comp_off, shape=(o.n_components + 1,))
        return np.diff(off.astype(np.int64))

    def call_sv_step(self) -> int:
        self._sv = SqSvTable()
        rc = self.lib.sq_call_sv(self.h, C.byref(self._sv))
        if rc < 0:
            self._chk(rc, "sq_call_sv")
        return rc

    def sv_rows(self) -> list[tuple]:
        t = self._sv
        return [(t.chr1[i], t.start1[i], t.end1[i], t.chr2[i], t.start2[i], t.end2[i], t.score[i], t.strand1_minus[i], t.strand2_minus[i], t.sup1[i], t.sup2[i])
                for i in range(t.n_rows)]

    def call_sv(self) -> list[tuple]:
        t = SqSvTable()
        self._run_stage(lambda: self.lib.sq_call_sv(self.h, C.byref(t)), "sq_call_sv")
        return [(t.chr1[i], t.start1[i], t.end1[i], t.chr2[i], t.start2[i], t.end2[i], t.score[i], t.strand1_minus[i], t.strand2_minus[i], t.sup1[i], t.sup2[i])
                for i in range(t.n_rows)]

    def breakpoints(self) -> list[list[tuple]]:
        t = SqBpTable()
        self._chk(self.lib.sq_breakpoints(self.h, C.byref(t)), "sq_breakpoints")
        return [[(t.bp1[j], t.bp2[j], t.sup1[j], t.sup2[j]) for j in range(t.bp_off[e], t.bp_off[e + 1])] for e in range(t.n_edges)]

    def records(self) -> dict:
        """copy of the HBM-resident record SoA (tests only)"""
        import numpy as np

        b = SqAlnBatch()
        self.lib.sq_debug_download.argtypes = [C.c_void_p, C.POINTER(SqAlnBatch)]
        self._chk(self.lib.sq_debug_download(self.h, C.byref(b)), "sq_debug_download")
        n, nb = b.n_rec, b.n_blk

        def arr(ptr, cnt, dt):
            return np.ctypeslib.as_array(ptr, shape=(cnt,)).astype(dt).copy() if cnt else np.zeros(0, dt)

        return {"refid": arr(b.refid, n, "i4"), "pos": arr(b.pos, n, "i4"), "mate_refid": arr(b.mate_refid, n, "i4"), "mate_pos": arr(b.mate_pos, n, "i4"),
                "end_pos": arr(b.end_pos, n, "i4"), "flag": arr(b.flag, n, "u2"), "mapq": arr(b.mapq, n, "u1"), "aux": arr(b.aux, n, "u1"), "totlen": arr(b.totlen, n, "u2"),
                "blk_off": arr(b.blk_off, n + 1, "u4"), "b_refpos": arr(b.b_refpos, nb, "i4"), "b_matchref": arr(b.b_matchref, nb, "i4"),
                "b_readpos": arr(b.b_readpos, nb, "u2"), "b_matchread": arr(b.b_matchread, nb, "u2")}

    def bp_support(self, chrs, poss, host_walk: bool = False):
        """coverage of an arbitrary sorted breakpoint list by the resident pass-3 records (tests only)"""
        import numpy as np

        ch = np.ascontiguousarray(chrs, dtype=np.int32)
        po = np.ascontiguousarray(poss, dtype=np.int32)
        out = np.zeros(len(ch), np.int32)
        I32P = C.POINTER(C.c_int32)
        self.lib.sq_debug_bp_support.argtypes = [C.c_void_p, C.c_int32, I32P, I32P, I32P, C.c_int32]
        self._chk(self.lib.sq_debug_bp_support(self.h, len(ch), ch.ctypes.data_as(I32P), po.ctypes.data_as(I32P), out.ctypes.data_as(I32P), int(host_walk)), "sq_debug_bp_support")
        return out

    def order_problem(self, n: int, edges: list, use_gpu: bool):
        """canonical optimum (mask, order, value) of one ordering problem; edges = [(u, v, head_u, head_v, w)], u < v (tests only)"""
        flat = (C.c_int32 * (5 * len(edges)))(*[int(x) for e in edges for x in e])
        mask, value, order = C.c_int32(), C.c_int64(), (C.c_int32 * n)()
        self.lib.sq_debug_order.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
        self._chk(self.lib.sq_debug_order(self.h, n, len(edges), flat, int(use_gpu), C.byref(mask), order, C.byref(value)), "sq_debug_order")
        o = list(order)
        m = sum(1 << (~x) for x in o if x < 0)  # reversed nodes come back bit-complemented
        return m, [x if x >= 0 else ~x for x in o], value.value

    def timing_accumulate(self, keep: bool = True):
        """let the timing table accumulate over repeated build_graph/order/call_sv runs (read it once with timing())"""
        self.lib.sq_timing_accumulate.argtypes = [C.c_void_p, C.c_int32]
        self._chk(self.lib.sq_timing_accumulate(self.h, int(keep)), "sq_timing_accumulate")

    def timing(self) -> dict:
        t = SqTiming()
        self._chk(self.lib.sq_get_timing(self.h, C.byref(t)), "sq_get_timing")
        return {t.names[i].decode(): {"ms": t.ms[i], "launches": t.launches[i], "bytes": t.bytes[i], "busy_ms": t.busy_ms[i]} for i in range(t.n)}

    def counts(self) -> dict:
        k = SqCounts()
        self._chk(self.lib.sq_get_counts(self.h, C.byref(k)), "sq_get_counts")
        return {f: getattr(k, f) for f, _ in SqCounts._fields_}

    def sv_text_fast(self) -> str:
        """sv_text() without a Python tuple per row (dense samples have 1e5 rows)"""
        import numpy as np

        t = SqSvTable()
        self._run_stage(lambda: self.lib.sq_call_sv(self.h, C.byref(t)), "sq_call_sv")
        head = "# chrom1\tstart1\tend1\tchrom2\tstart2\tend2\tname\tscore\tstrand1\tstrand2\tnum_concordantfrag_bp1\tnum_concordantfrag_bp2\n"
        n = t.n_rows
        if n <= 0:
            return head
        col = lambda p: np.ctypeslib.as_array(p, shape=(n,)).tolist()
        names = self.ref_names
        c1, s1, e1, c2, s2, e2, sc, m1, m2, u1, u2 = (col(t.chr1), col(t.start1), col(t.end1), col(t.chr2), col(t.start2), col(t.end2), col(t.score),
                                                      col(t.strand1_minus), col(t.strand2_minus), col(t.sup1), col(t.sup2))
        pm = ("+", "-")
        return head + "".join([f"{names[c1[i]]}\t{s1[i]}\t{e1[i]}\t{names[c2[i]]}\t{s2[i]}\t{e2[i]}\t.\t{sc[i]}\t{pm[m1[i]]}\t{pm[m2[i]]}\t{u1[i]}\t{u2[i]}\n" for i in range(n)])

    def sv_text(self) -> str:
        """The `_sv.txt` file content (src/WriteIO.cpp:49-123)."""
        rows = self.call_sv()
        out = ["# chrom1\tstart1\tend1\tchrom2\tstart2\tend2\tname\tscore\tstrand1\tstrand2\tnum_concordantfrag_bp1\tnum_concordantfrag_bp2\n"]
        for r in rows:
            out.append(f"{self.ref_names[r[0]]}\t{r[1]}\t{r[2]}\t{self.ref_names[r[3]]}\t{r[4]}\t{r[5]}\t.\t{r[6]}\t{'-' if r[7] else '+'}\t{'-' if r[8] else '+'}\t{r[9]}\t{r[10]}\n")
        return "".join(out)


def run_pipeline(bam: str, chim_bam: str, device: int = 0, **params) -> dict:
    """BAM -> graph -> ordering -> SV rows in one call (what `squid -b -c -o` does)."""
    with Context(device=device, **params) as ctx:
        ctx.load(bam, chim_bam)
        ctx.build_graph()
        orders = ctx.order()
        sv = ctx.sv_text()
        return {"orders": orders, "sv_text": sv, "graph": ctx.graph(0), "timing": ctx.timing(), "counts": ctx.counts()}
