"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol include/squid_hip.h declares,
and refuses to run its GPU stages without a device (no CPU fallback)."""
import ctypes as C
import re

import pytest

import squid_amd


def _declared(root):
    text = (root / "include" / "squid_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sq_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(built):
    lib = squid_amd.load_library()
    names = _declared(squid_amd.ROOT)
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/squid_hip.h but not exported"
    assert sorted(squid_amd.EXPORTS) == names


def test_no_torch_or_cxx_types_in_signatures():
    text = (squid_amd.ROOT / "include" / "squid_hip.h").read_text()
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)  # comments may mention torch.distributed
    assert "std::" not in code and "torch" not in code and "at::" not in code and "#include <string>" not in code


def test_header_reader_needs_no_gpu(built, synth):
    pre = synth("C1")
    names, lens = squid_amd.read_header(f"{pre}.bam")
    assert names == ["chr1"] and lens == [10000000]


def test_strerror_and_defaults(built):
    lib = squid_amd.load_library()
    p = squid_amd.SqParams()
    lib.sq_default_params(C.byref(p))
    # defaults of src/Config.cpp:19-28
    assert (p.phred_type, p.max_lowphred_len, p.min_phred, p.min_mapqual) == (1, 10, 4, 1)
    assert (p.concord_dist_pos, p.concord_dist_idx, p.min_edge_weight, p.discordant_ratio, p.max_allowed_degree) == (50000, 20, 5, 8.0, 5)
    assert lib.sq_strerror(-2) == b"no usable HIP device"


def test_context_fails_loudly_without_gpu(built):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(squid_amd.SquidError, match="no CPU fallback"):
        squid_amd.Context()


def test_parallel_sort_reproduces_libstdcxx_sort(tmp_path):
    """sq_parsort.h: std::sort's introsort with its independent sub-ranges on several threads gives std::sort's permutation"""
    import subprocess
    from pathlib import Path

    src = Path(__file__).resolve().parent / "parsort_check.cpp"
    exe = tmp_path / "parsort_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", str(exe), str(src), "-lpthread"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout


def test_the_host_program_asks_for_eight_hardware_queues_not_the_library(built):
    """the GPU reader overlaps up to eight kernels; the HIP runtime gives a process four hardware queues per priority unless
    GPU_MAX_HW_QUEUES says otherwise when it initialises.  Asking is the HOST PROGRAM's business:
    loading the library leaves the environment alone (round 4 called setenv from a static constructor), the Python binding sets
    the variable at import unless it is there already -- and so does `build/squid` in main (squid_main.cpp)"""
    import os
    import subprocess
    import sys

    probe = "libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p; print((libc.getenv(b'GPU_MAX_HW_QUEUES') or b'-').decode())"
    lib = str(squid_amd.LIB_PATH)
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    run = lambda code, e: subprocess.run([sys.executable, "-c", code, lib], env=e, capture_output=True, text=True, check=True).stdout.strip()
    assert run("import ctypes, sys; ctypes.CDLL(sys.argv[1]); " + probe, env) == "-"
    root = str(squid_amd.ROOT)
    assert run(f"import ctypes, sys; sys.path.insert(0, {root!r}); import squid_amd; " + probe, env) == "8"
    assert run(f"import ctypes, sys; sys.path.insert(0, {root!r}); import squid_amd; " + probe, dict(env, GPU_MAX_HW_QUEUES="4")) == "4"
    assert 'setenv("GPU_MAX_HW_QUEUES"' in (squid_amd.ROOT / "squid_amd" / "csrc" / "squid_main.cpp").read_text()


def test_host_pool_loops_return_when_started_from_a_task_of_the_pool(tmp_path):
    """HostPool::parallel_for returns when every index is done, not when every helper task has had its turn: a loop started from a task of
    a one-thread pool (eight ranks sharing a 16-CPU box: the cluster table runs as such a task) used to wait for a helper only its own
    thread could have run"""
    import subprocess
    from pathlib import Path

    src = Path(__file__).resolve().parent / "pool_check.cpp"
    exe = tmp_path / "pool_check"
    subprocess.check_call(["hipcc", "-O2", "-std=c++17", "-I", str(squid_amd.ROOT / "include"), "-o", str(exe), str(src), "-lpthread"], stderr=subprocess.DEVNULL)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout
