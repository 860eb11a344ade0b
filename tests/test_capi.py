"""CPU: the C-ABI library builds, loads and exports every symbol include/rii_amd.h declares; without a GPU
the product path fails loudly instead of falling back."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "rii_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rii_[a-z_0-9A-Z]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from rii_amd import core
    so = core.build_library()
    lib = ctypes.CDLL(so)
    names = header_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "librii_amd.so does not export %s" % n
    assert set(core.exported_symbols()) == set(names), "ctypes signature table out of sync with the header"


@pytest.mark.parametrize("M", [16, 32, 64])
def test_matrix_core_scan_lane_assignment_is_complete_and_conflict_free(M):
    """fscan_mx_kernel splits the M table rows of a code over the four lanes 16 g + n of a wave and lets one matrix
    instruction add them up.  Two properties make that both right and fast, whatever the data: (1) the four lanes of a
    code fetch every subspace exactly once; (2) in every step the 16 lanes of each ds_read_b128 service group of the LDS
    (MI355X: {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32) touch 16 different bank slots -- with the rotated
    table layout the slot of a row is its subspace mod 16.  M = 64 has 8-byte rows: ds_read_b64, service groups {0-31} and
    {32-63}, slot = subspace mod 32."""
    from rii_amd import core
    T = M // 4
    sub = [[core.fscan_lane_subspace(M, lane, t) for t in range(T)] for lane in range(64)]
    assert all(0 <= m < M for row in sub for m in row)
    for n in range(16):
        seen = sorted(sub[16 * g + n][t] for g in range(4) for t in range(T))
        assert seen == list(range(M)), (n, seen)
    if M == 64:
        groups, nslot = [list(range(0, 32)), list(range(32, 64))], 32
    else:
        groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
                  list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
        groups += [[l + 32 for l in g] for g in groups]
        nslot = 16
    assert sorted(l for g in groups for l in g) == list(range(64))
    for t in range(T):
        for g in groups:
            slots = sorted(sub[l][t] % nslot for l in g)
            assert slots == list(range(nslot)), (t, g, slots)
            assert len({sub[l][t] // nslot for l in g}) == 1          # one table half per service group and step
    assert core.fscan_lane_subspace(M, 64, 0) == -1 and core.fscan_lane_subspace(M, 0, T) == -1
    assert core.fscan_lane_subspace(8, 0, 0) == -1


def test_no_gpu_means_loud_failure_not_fallback():
    from rii_amd import core, RiiGpu, RiiAmdError
    if core._lib().rii_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RiiAmdError):
        RiiGpu(np.zeros((2, 4, 3), np.float32), False)


def test_comm_init_refuses_bad_ranks_without_touching_a_device():
    """rii_comm_init validates (id, rank, nranks) before any HIP / RCCL call: a wrong world description is an error code, not a
    hang inside ncclCommInitRank (a DUPLICATE rank cannot be seen by one process alone: RCCL's own rendezvous reports it)."""
    from rii_amd import core
    lib = core._lib()
    out = ctypes.c_void_p()
    ident = ctypes.create_string_buffer(core.COMM_ID_BYTES)
    for rank, nranks in ((0, 0), (-1, 2), (2, 2), (5, 1), (0, -3)):
        assert lib.rii_comm_init(ident, rank, nranks, 0, ctypes.byref(out)) == -1, (rank, nranks)      # RII_ERR_INVALID
        assert not out.value
    assert lib.rii_comm_init(None, 0, 1, 0, ctypes.byref(out)) == -1
    assert lib.rii_comm_init(ident, 0, 1, 0, None) == -1
    assert lib.rii_comm_rank(None) == -1 and lib.rii_comm_size(None) == 0
    # the stateless helpers of the sharded protocol answer without a device as well
    assert lib.rii_ivf_shard_replay_scratch_bytes(3, 8192) == 0 and lib.rii_ivf_shard_replay_scratch_bytes(3, 8193) == 3 * 8193 * 16
    assert lib.rii_ivf_shard_max_select_rows(None, 100, 1000, 0) == -1


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under rii_amd/ may import, include, link or load it."""
    pkg = os.path.join(ROOT, "rii_amd")
    offenders = []
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                continue
            for ln, line in enumerate(open(os.path.join(dirpath, f), errors="replace"), 1):
                low = line.lower()
                if "oracle" in low and any(tok in low for tok in ("import", "#include", "cdll", "dlopen", "-l", "librii_oracle")):
                    offenders.append("%s:%d: %s" % (os.path.join(dirpath, f), ln, line.strip()))
    assert not offenders, offenders


def _build_c_client(tmp_path):
    import subprocess
    from rii_amd import core
    so = core.build_library()
    exe = str(tmp_path / "c_abi_smoke")
    subprocess.check_call(["gcc", "-O1", "-std=c99", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi", "c_abi_smoke.c"), "-o", exe,
                           "-L", os.path.dirname(so), "-lrii_amd", "-Wl,-rpath," + os.path.dirname(so),
                           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_plain_c_client_links_and_fails_loudly_without_gpu(tmp_path):
    """The boundary really is C: a gcc-compiled C99 program includes the header and links the library."""
    import subprocess
    from rii_amd import core
    if core._lib().rii_device_count() > 0:
        pytest.skip("a GPU is present (covered by the gpu-marked variant)")
    exe = _build_c_client(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "failed loudly" in out.stdout, (out.stdout, out.stderr)


@pytest.mark.gpu
def test_plain_c_client_on_gpu(tmp_path):
    import subprocess
    exe = _build_c_client(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "C ABI smoke OK" in out.stdout, (out.stdout, out.stderr)


@pytest.mark.gpu
def test_plain_c_client_shards_without_python(tmp_path):
    """tests/c_abi/c_abi_sharded.c: a gcc-compiled C99 program creates an RCCL communicator through the library (world size 1) and
    runs the query-sharded, database-sharded and device-queries -> host-rows entry points; every row equals the single-engine call."""
    import subprocess
    from rii_amd import core
    so = core.build_library()
    exe = str(tmp_path / "c_abi_sharded")
    subprocess.check_call(["gcc", "-O1", "-std=c99", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi", "c_abi_sharded.c"), "-o", exe,
                           "-L", os.path.dirname(so), "-lrii_amd", "-Wl,-rpath," + os.path.dirname(so),
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "C ABI sharded OK" in out.stdout, (out.stdout, out.stderr)


def test_header_documents_every_engine_option():
    """the public header had drifted behind rii_set_option.  Every key the engine accepts (and every read-only key of
    rii_get_option) must be named in include/rii_amd.h, and the header must not name options the engine no longer has."""
    eng = open(os.path.join(ROOT, "rii_amd", "csrc", "engine.hip")).read()
    a, b = eng.index("RII_API int rii_set_option"), eng.index("RII_API int rii_timing_read")
    keys = set(re.findall(r'k == "([a-z_0-9]+)"', eng[a:b]))
    hdr = open(os.path.join(ROOT, "include", "rii_amd.h")).read()
    doc = hdr[hdr.index("/* Options (rii_set_option"):hdr.index("int rii_set_option")]
    named = set(re.findall(r'"([a-z_0-9]+)"', doc))
    assert keys - named == set(), "options missing from the header: %s" % sorted(keys - named)
    assert named - keys == set(), "the header names options the engine does not have: %s" % sorted(named - keys)


def test_main_module_shim_exposes_riicpp():
    """`rii_amd.main.RiiCpp` is what `import main` resolves to after the one-line swap of INTEGRATION.md §2: it must carry
    every member the reference's `rii/rii.py` touches on `main.RiiCpp` (src/main.cpp:12-54)."""
    from rii_amd import main
    for name in ("reconfigure", "add_codes", "query_linear", "query_ivf", "clear", "verbose", "coarse_centers",
                 "flattened_codes", "posting_lists", "N", "nlist", "__getstate__", "__setstate__"):
        assert hasattr(main.RiiCpp, name), name
