"""ctypes binding of librii_amd.so -- the thin shim that takes the place of the reference's pybind11 module
`main` (src/main.cpp:11-61).  `RiiGpu` exposes the same attribute names as `main.RiiCpp` so that the Python
class `Rii` (rii_amd/api.py, the counterpart of the reference's rii/rii.py) is written against an identical surface, plus the batched
entry points the reference does not have.

The product path never falls back to a CPU implementation: if the HIP library is missing it is built with
hipcc; if there is no GPU, constructing an engine raises RiiAmdError.
"""
import ctypes
import importlib.util
import os
import subprocess
import sys
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SIMD = {"sse": 0, "avx": 1, "avx512": 2}
LUT_MODES = {"exact": 0, "mfma": 1}


class RiiAmdError(RuntimeError):
    pass


def host_simd_arch():
    """The fvec_L2sqr variant the reference would compile to on this host under its `-march=native`
    (setup.py:96; src/distance.h:113,172,219).  Override with RII_SIMD_ARCH=sse|avx|avx512."""
    env = os.environ.get("RII_SIMD_ARCH")
    if env:
        if env not in SIMD:
            raise ValueError("RII_SIMD_ARCH must be one of %s" % list(SIMD))
        return env
    flags = set()
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    flags = set(line.split())
                    break
    except OSError:
        pass
    if "avx512f" in flags:
        return "avx512"
    if "avx" in flags:
        return "avx"
    return "sse"


def library_path():
    return os.path.join(_HERE, "librii_amd.so")


def build_library(force=False):
    """Compile rii_amd/csrc/*.hip for gfx950 with hipcc (cross-compiles without a GPU)."""
    so = library_path()
    csrc = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "rii_amd.h"))
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", csrc, "-j4"], stdout=subprocess.DEVNULL)
    return so


def _bind_hip_runtime():
    """A process must hold ONE HIP runtime.  PyTorch-ROCm wheels ship their own libamdhip64 and load it by file name
    (torch/lib, RPATH $ORIGIN); librii_amd.so asks for the soname.  If this library came first, a later `import torch`
    would bring a second runtime into the process, and that one finds no GPU.  So when torch is installed and not loaded
    yet, bind to ITS runtime: the loader then resolves both requests to the same object.  RII_SYSTEM_HIP=1 opts out."""
    if "torch" in sys.modules or os.environ.get("RII_SYSTEM_HIP") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)


def _lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = library_path()
    if not os.path.exists(so):
        build_library()
    _bind_hip_runtime()
    L = ctypes.CDLL(so)
    c_int, c_i64, c_vp = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p
    f32p = ctypes.POINTER(ctypes.c_float)
    u8p = ctypes.POINTER(ctypes.c_uint8)
    i64p = ctypes.POINTER(ctypes.c_int64)
    i32p = ctypes.POINTER(ctypes.c_int32)
    sig = {
        "rii_last_error": (ctypes.c_char_p, []),
        "rii_version": (ctypes.c_char_p, []),
        "rii_device_count": (c_int, []),
        "rii_create": (c_int, [f32p, c_int, c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(c_vp)]),
        "rii_destroy": (None, [c_vp]),
        "rii_add_codes": (c_int, [c_vp, u8p, c_i64, c_int]),
        "rii_reconfigure": (c_int, [c_vp, c_int, c_int]),
        "rii_clear": (c_int, [c_vp]),
        "rii_set_coarse_centers": (c_int, [c_vp, u8p, c_i64]),
        "rii_set_state": (c_int, [c_vp, u8p, c_i64, u8p, c_i64, i64p, i32p]),
        "rii_set_posting_lists": (c_int, [c_vp, u8p, c_i64, i64p, i32p]),
        "rii_get_N": (c_i64, [c_vp]),
        "rii_get_nlist": (c_i64, [c_vp]),
        "rii_get_M": (c_int, [c_vp]),
        "rii_get_Ks": (c_int, [c_vp]),
        "rii_get_Ds": (c_int, [c_vp]),
        "rii_get_verbose": (c_int, [c_vp]),
        "rii_set_verbose": (c_int, [c_vp, c_int]),
        "rii_get_codewords": (c_int, [c_vp, f32p]),
        "rii_get_codes": (c_int, [c_vp, u8p]),
        "rii_get_coarse_centers": (c_int, [c_vp, u8p]),
        "rii_get_posting_lists": (c_int, [c_vp, i64p, i32p]),
        # the host-pointer query calls take plain addresses: ndarray.ctypes.data_as(POINTER(...)) costs 2.4 us per argument,
        # a quarter of a one-query call's latency
        "rii_query_linear": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_i64, c_vp, c_vp]),
        "rii_query_ivf": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp]),
        "rii_query_linear_dev": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_i64, c_vp, c_vp, c_vp]),
        "rii_query_ivf_dev": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp]),
        "rii_query_linear_dev_to_host": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_i64, c_vp, c_vp, c_vp]),
        "rii_query_ivf_dev_to_host": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp]),
        "rii_ivf_list_lengths_dev": (c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp]),
        "rii_query_ivf_shard_dev": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_int, c_int, c_int,
                                            c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
        "rii_ivf_shard_replay_dev": (c_int, [c_vp, c_int, c_i64, c_int, c_int, c_vp, c_vp, c_vp]),
        "rii_ivf_shard_replay_scratch_bytes": (c_i64, [c_i64, c_int]),
        "rii_ivf_shard_replay_ex_dev": (c_int, [c_vp, c_int, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_i64, c_vp]),
        "rii_ivf_shard_max_select_rows": (c_int, [c_vp, c_i64, c_i64, c_i64]),
        "rii_linear_tie_emit_dev": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_i64, c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_vp]),
        "rii_linear_tie_record_bytes": (c_i64, [c_i64, c_int]),
        "rii_linear_tie_replay_dev": (c_int, [c_vp, c_int, c_i64, c_int, c_int, c_vp, c_vp, c_vp]),
        "rii_merge_record_bytes": (c_i64, [c_i64, c_int, c_int]),
        "rii_merge_topk_dev": (c_int, [c_vp, c_int, c_i64, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
        "rii_merge_topk_ex_dev": (c_int, [c_vp, c_int, c_i64, c_int, c_int, c_int, i64p, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp]),
        "rii_merge_hdr_record_bytes": (c_i64, [c_i64, c_int, c_int]),
        "rii_merge_hdr_scratch_bytes": (c_i64, [c_int, c_i64, c_int]),
        "rii_merge_topk_hdr_dev": (c_int, [c_vp, c_int, c_i64, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_i64, c_vp]),
        "rii_ivf_merge_top1_hdr_dev": (c_int, [c_vp, c_int, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
        "rii_comm_unique_id": (c_int, [c_vp]),
        "rii_comm_init": (c_int, [c_vp, c_int, c_int, c_int, ctypes.POINTER(c_vp)]),
        "rii_comm_destroy": (None, [c_vp]),
        "rii_comm_rank": (c_int, [c_vp]),
        "rii_comm_size": (c_int, [c_vp]),
        "rii_query_linear_qsharded_dev": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_i64, c_vp, c_vp, c_vp]),
        "rii_query_ivf_qsharded_dev": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp]),
        "rii_query_linear_dbsharded_dev": (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_int, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_int, c_vp]),
        "rii_query_ivf_dbsharded_dev": (c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_int, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
        "rii_qshard_begin": (c_i64, [c_i64, c_int, c_int]),
        "rii_qshard_record_bytes": (c_i64, [c_i64, c_int, c_int, c_int]),
        "rii_qshard_unpack_dev": (c_int, [c_vp, c_i64, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
        "rii_dtable": (c_int, [c_vp, f32p, c_i64, f32p]),
        "rii_assign": (c_int, [c_vp, u8p, c_i64, i32p]),
        "rii_fscan_lane_subspace": (c_int, [c_int, c_int, c_int]),
        "rii_set_option": (c_int, [c_vp, ctypes.c_char_p, c_i64]),
        "rii_get_option": (c_i64, [c_vp, ctypes.c_char_p]),
        "rii_timing_read": (c_int, [c_vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), i64p]),
        "rii_timing_reset": (c_int, [c_vp]),
        "rii_synchronize": (c_int, [c_vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    L._rii_signatures = sig
    _LIB = L
    return L


def merge_record_bytes(B, k, payload=False):
    """Bytes of one rank's record in the database-sharding all-gather (include/rii_amd.h: rii_merge_topk_dev)."""
    return int(_lib().rii_merge_record_bytes(int(B), int(k), int(bool(payload))))


def merge_topk_dev(d_gathered, G, B, k, d_out_keys, d_out_dists, stream=0, k_out=None, d_out_payload=0):
    """Device pointers in, asynchronous on `stream`: the k_out smallest of the G gathered records per query under
    (dist, key), with their payloads when d_out_payload is given."""
    _check(_lib().rii_merge_topk_dev(d_gathered, int(G), int(B), int(k), int(k if k_out is None else k_out),
                                     int(bool(d_out_payload)), d_out_keys, d_out_dists, d_out_payload or None, stream or None))


def merge_topk_ex_dev(d_gathered, G, B, k, k_out, id_offsets, d_out_keys, d_out_dists, tie_cols=0, d_out_tie=0, d_out_any=0, stream=0,
                      d_out_payload=0):
    """rii_merge_topk_ex_dev: merge with per-rank id offsets (host sequence of G ints or None) and device-side tie flags."""
    off = None
    if id_offsets is not None:
        off = (ctypes.c_int64 * int(G))(*[int(x) for x in id_offsets])
    _check(_lib().rii_merge_topk_ex_dev(d_gathered, int(G), int(B), int(k), int(k_out), int(bool(d_out_payload)), off, d_out_keys,
                                        d_out_dists, d_out_payload or None, int(tie_cols), d_out_tie or None, d_out_any or None,
                                        stream or None))


def merge_hdr_record_bytes(B, k, payload=False):
    return int(_lib().rii_merge_hdr_record_bytes(int(B), int(k), int(bool(payload))))


def merge_hdr_scratch_bytes(G, B, k):
    return int(_lib().rii_merge_hdr_scratch_bytes(int(G), int(B), int(k)))


def merge_topk_hdr_dev(d_gathered, G, B, k, k_out, d_out_keys, d_out_dists, d_out_payload=0, tie_cols=0, d_out_tie=0, d_out_any=0,
                       d_scratch=0, scratch_bytes=0, stream=0):
    """rii_merge_topk_hdr_dev: the merge of records with the {id offset, status} header (any G, any k)."""
    _check(_lib().rii_merge_topk_hdr_dev(d_gathered, int(G), int(B), int(k), int(k_out), int(bool(d_out_payload)), d_out_keys, d_out_dists,
                                         d_out_payload or None, int(tie_cols), d_out_tie or None, d_out_any or None, d_scratch or None,
                                         int(scratch_bytes), stream or None))


def ivf_merge_top1_hdr_dev(d_gathered, G, B, d_counts, d_out_ids, d_out_dists, d_out_counts, d_out_any=0, stream=0):
    """rii_ivf_merge_top1_hdr_dev: merge + finish of the sharded inverted index's top-1 batch (records with payload and header, k = 2)."""
    _check(_lib().rii_ivf_merge_top1_hdr_dev(d_gathered, int(G), int(B), d_counts, d_out_ids, d_out_dists, d_out_counts, d_out_any or None,
                                             stream or None))


def ivf_shard_replay_scratch_bytes(nf, rows):
    """Bytes of device scratch rii_ivf_shard_replay_ex_dev needs (0 while the rebuilt sequences fit LDS: rows <= 8192)."""
    return int(_lib().rii_ivf_shard_replay_scratch_bytes(int(nf), int(rows)))


def ivf_shard_replay_dev(d_gathered, G, nf, rows, topk, d_out_ids, d_out_dists, stream=0, d_scratch=0, scratch_bytes=0):
    """std::partial_sort replayed over the gathered candidate sequences of nf queries (include/rii_amd.h); any `rows` when the caller
    hands ivf_shard_replay_scratch_bytes(nf, rows) bytes of device scratch."""
    _check(_lib().rii_ivf_shard_replay_ex_dev(d_gathered, int(G), int(nf), int(rows), int(topk), d_out_ids, d_out_dists,
                                              d_scratch or None, int(scratch_bytes), stream or None))


def linear_tie_record_bytes(nf, cap):
    return int(_lib().rii_linear_tie_record_bytes(int(nf), int(cap)))


def linear_tie_replay_dev(d_gathered, G, nf, cap, topk, d_out_ids, d_out_dists, stream=0):
    """Replay of std::partial_sort over the all-gathered candidate lists of database-sharded linear search (include/rii_amd.h)."""
    _check(_lib().rii_linear_tie_replay_dev(d_gathered, int(G), int(nf), int(cap), int(topk), d_out_ids, d_out_dists, stream))


COMM_ID_BYTES = 128


def comm_unique_id():
    """RII_COMM_ID_BYTES opaque bytes from rii_comm_unique_id: created by one rank, handed to all (any transport)."""
    buf = ctypes.create_string_buffer(COMM_ID_BYTES)
    _check(_lib().rii_comm_unique_id(buf))
    return buf.raw


class Comm(object):
    """One RCCL communicator behind the C ABI (rii_comm_init: collective over `nranks` processes, one GPU each) and the sharded
    query entry points that use it: engine kernels -> ONE ncclAllGather -> unpack / merge kernel, all enqueued by the library."""

    def __init__(self, comm_id, rank, nranks, device):
        self._h = ctypes.c_void_p()
        _check(_lib().rii_comm_init(ctypes.c_char_p(comm_id), int(rank), int(nranks), int(device), ctypes.byref(self._h)))
        self.rank, self.size = int(rank), int(nranks)

    def close(self):
        """rii_comm_destroy (synchronises the device, destroys the RCCL communicator).  Idempotent."""
        if getattr(self, "_h", None):
            _lib().rii_comm_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def query_linear_qsharded_dev(self, engine, d_queries, B, topk, d_tids, S, d_out_ids, d_out_dists, stream=0):
        _check(_lib().rii_query_linear_qsharded_dev(engine._h, self._h, d_queries, int(B), int(topk), d_tids or None, int(S), d_out_ids,
                                                    d_out_dists, stream or None))

    def query_ivf_qsharded_dev(self, engine, d_queries, B, topk, d_tids, S, L, d_out_ids, d_out_dists, d_out_counts, stream=0):
        _check(_lib().rii_query_ivf_qsharded_dev(engine._h, self._h, d_queries, int(B), int(topk), d_tids or None, int(S), int(L),
                                                 d_out_ids, d_out_dists, d_out_counts, stream or None))

    def query_linear_dbsharded_dev(self, engine, id_offset, d_queries, B, topk, d_tids_local, S_local, S_global, d_out_ids, d_out_dists,
                                   d_out_tie=0, d_out_overflow=0, tie_cap=0, stream=0):
        _check(_lib().rii_query_linear_dbsharded_dev(engine._h, self._h, int(id_offset), d_queries, int(B), int(topk), d_tids_local or None,
                                                     int(S_local), int(S_global), d_out_ids, d_out_dists, d_out_tie or None,
                                                     d_out_overflow or None, int(tie_cap), stream or None))


def _comm_query_ivf_dbsharded_dev(self, engine, id_offset, N_global, d_queries, B, topk, d_tids_local, S_local, S_global, L, d_out_ids,
                                  d_out_dists, d_out_counts, d_out_tie=0, stream=0):
    _check(_lib().rii_query_ivf_dbsharded_dev(engine._h, self._h, int(id_offset), int(N_global), d_queries, int(B), int(topk),
                                              d_tids_local or None, int(S_local), int(S_global), int(L), d_out_ids, d_out_dists,
                                              d_out_counts, d_out_tie or None, stream or None))


Comm.query_ivf_dbsharded_dev = _comm_query_ivf_dbsharded_dev


def fscan_lane_subspace(M, lane, t):
    """Subspace whose table row lane `lane` of a wave fetches as lookup t in fscan_mx_kernel (host-only introspection)."""
    return int(_lib().rii_fscan_lane_subspace(int(M), int(lane), int(t)))


def exported_symbols():
    """Names declared in include/rii_amd.h that the loaded library must export (used by the CPU tests)."""
    return sorted(_lib()._rii_signatures.keys())


def _ptr(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def _addr(a):
    """Address of a contiguous ndarray's first element (None for an empty one) for a c_void_p parameter."""
    return a.ctypes.data if a.size else None


def _check(rc):
    if rc != 0:
        msg = _lib().rii_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(msg)
        raise RiiAmdError("[%d] %s" % (rc, msg))


_EMPTY_I64 = np.zeros(0, np.int64)


def _rebuild_engine(cls, simd_arch):
    """Unpickling helper: an empty engine (RiiCpp(), src/main.cpp:13) that remembers the SIMD order it emulates."""
    e = cls.__new__(cls)
    e._h = ctypes.c_void_p()
    e._simd = simd_arch
    return e


class RiiGpu(object):
    """MI355X engine with the surface of `main.RiiCpp` (src/main.cpp:12-54)."""

    def __init__(self, codewords=None, verbose=False, simd_arch=None, device=None):
        self._h = ctypes.c_void_p()
        if codewords is None:            # RiiCpp() -- "required in pickle" (main.cpp:13)
            return
        self._create(np.asarray(codewords), bool(verbose), simd_arch, device)

    def _create(self, codewords, verbose, simd_arch, device):
        if codewords.dtype != np.float32:
            raise TypeError("codewords must be float32 (pybind11 array_t<float>, src/main.cpp:14)")
        if codewords.ndim != 3:
            raise ValueError("codewords must have ndim=3 (M, Ks, Ds)")
        cw = np.ascontiguousarray(codewords)
        self._simd = simd_arch or host_simd_arch()
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", os.environ.get("RII_DEVICE", "0")))
            if device >= max(_lib().rii_device_count(), 1):
                device = 0
        self._device = int(device)
        M, Ks, Ds = cw.shape
        h = ctypes.c_void_p()
        _check(_lib().rii_create(_ptr(cw, ctypes.c_float), M, Ks, Ds, int(verbose), SIMD[self._simd], self._device,
                                 ctypes.byref(h)))
        self._h = h
        self._D = M * Ds
        self._tl = threading.local()          # per-thread buffers of the one-query calls

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib().rii_destroy(self._h)
                self._h = ctypes.c_void_p()
        except Exception:
            pass

    # ---- main.cpp:15-16,28 ----
    def reconfigure(self, nlist, iter):
        _check(_lib().rii_reconfigure(self._h, int(nlist), int(iter)))

    def add_codes(self, codes, update_flag):
        codes = np.asarray(codes)
        if codes.dtype != np.uint8:
            raise TypeError("codes must be uint8")
        if codes.ndim != 2 or codes.shape[1] != self.M:
            raise ValueError("codes must have shape (N, M=%d)" % self.M)
        c = np.ascontiguousarray(codes)
        _check(_lib().rii_add_codes(self._h, _ptr(c, ctypes.c_uint8), c.shape[0], int(bool(update_flag))))

    def clear(self):
        _check(_lib().rii_clear(self._h))

    def set_coarse_centers(self, centers):
        """Import coarse centres and rebuild the posting lists by coarse assignment (what the reference reaches
        through its pickle set-state, src/main.cpp:39-52, followed by rii.h:150-155)."""
        c = np.ascontiguousarray(centers, dtype=np.uint8)
        assert c.ndim == 2 and c.shape[1] == self.M
        _check(_lib().rii_set_coarse_centers(self._h, _ptr(c, ctypes.c_uint8), c.shape[0]))

    def set_posting_lists(self, centers, pl_off, pl_ids):
        """Centres [nlist, M] and posting lists in CSR form (pl_off int64 [nlist + 1], pl_ids int32, ascending inside a list)
        installed verbatim over the codes already added (rii_set_posting_lists: the lists half of the pickle set-state)."""
        c = np.ascontiguousarray(centers, dtype=np.uint8)
        off = np.ascontiguousarray(pl_off, dtype=np.int64)
        ids = np.ascontiguousarray(pl_ids, dtype=np.int32)
        assert c.ndim == 2 and c.shape[1] == self.M and off.shape == (c.shape[0] + 1,) and ids.shape == (int(off[-1]),)
        _check(_lib().rii_set_posting_lists(self._h, _ptr(c, ctypes.c_uint8), c.shape[0], _ptr(off, ctypes.c_int64), _ptr(ids, ctypes.c_int32)))

    # ---- helpers ----
    @staticmethod
    def _as_query_batch(q, D):
        q = np.asarray(q)
        if q.dtype != np.float32:
            raise TypeError("query must be float32 (py::arg(\"query\").noconvert(), src/main.cpp:18)")
        if q.ndim != 2 or q.shape[1] != D:
            raise ValueError("queries must have shape (B, D=%d)" % D)
        return np.ascontiguousarray(q)

    @staticmethod
    def _as_tids(t):
        if t is None:
            return _EMPTY_I64
        t = np.asarray(t)
        if t.dtype != np.int64:
            raise TypeError("target_ids must be int64 (py::arg(\"target_ids\").noconvert(), src/main.cpp:20)")
        if t.ndim != 1:
            raise ValueError("target_ids must be 1-D")
        return np.ascontiguousarray(t)

    # ---- batched entry points (NEW) ----
    def query_linear_batch(self, queries, topk, target_ids=None):
        Q = self._as_query_batch(queries, self._D)
        t = self._as_tids(target_ids)
        B = Q.shape[0]
        ids = np.empty((B, topk), np.int64)
        d = np.empty((B, topk), np.float32)
        _check(_lib().rii_query_linear(self._h, _addr(Q), B, int(topk), _addr(t), t.size, _addr(ids), _addr(d)))
        return ids, d

    def query_ivf_batch(self, queries, topk, target_ids, L):
        Q = self._as_query_batch(queries, self._D)
        t = self._as_tids(target_ids)
        B = Q.shape[0]
        ids = np.empty((B, topk), np.int64)
        d = np.empty((B, topk), np.float32)
        cnt = np.empty(B, np.int64)
        _check(_lib().rii_query_ivf(self._h, _addr(Q), B, int(topk), _addr(t), t.size, int(L), _addr(ids), _addr(d), _addr(cnt)))
        return ids, d, cnt

    def query_linear_dev(self, d_queries, B, topk, d_tids, S, d_out_ids, d_out_dists, stream=0):
        """Raw device pointers (ints); asynchronous on `stream` (0/None = the engine's own stream)."""
        _check(_lib().rii_query_linear_dev(self._h, d_queries, B, int(topk), d_tids or None, S, d_out_ids,
                                           d_out_dists, stream or None))

    def query_ivf_dev(self, d_queries, B, topk, d_tids, S, L, d_out_ids, d_out_dists, d_out_counts, stream=0):
        _check(_lib().rii_query_ivf_dev(self._h, d_queries, B, int(topk), d_tids or None, S, int(L), d_out_ids,
                                        d_out_dists, d_out_counts, stream or None))

    def query_linear_dev_to_host(self, d_queries, B, topk, d_tids, S, out_ids, out_dists, stream=0):
        """Device-resident queries (raw pointer) -> rows in the HOST arrays out_ids [B, topk] int64 / out_dists [B, topk] float32 when
        the call returns (rii_query_linear_dev_to_host: rows written by the kernels into the engine's pinned block, flag wait)."""
        _check(_lib().rii_query_linear_dev_to_host(self._h, d_queries, int(B), int(topk), d_tids or None, int(S), out_ids.ctypes.data,
                                                   out_dists.ctypes.data, stream or None))

    def query_ivf_dev_to_host(self, d_queries, B, topk, d_tids, S, L, out_ids, out_dists, out_counts, stream=0):
        _check(_lib().rii_query_ivf_dev_to_host(self._h, d_queries, int(B), int(topk), d_tids or None, int(S), int(L), out_ids.ctypes.data,
                                                out_dists.ctypes.data, out_counts.ctypes.data, stream or None))

    # ---- database-sharded inverted index (device pointers; protocol: include/rii_amd.h, rii_amd/dist.py) ----
    def linear_tie_emit_dev(self, d_queries, nf, topk, d_tids, S, d_bound, id_offset, cap, d_out_ids, d_out_dists, d_out_count,
                            stream=0):
        _check(_lib().rii_linear_tie_emit_dev(self._h, d_queries, int(nf), int(topk), d_tids, int(S), d_bound, int(id_offset),
                                              int(cap), d_out_ids, d_out_dists, d_out_count, stream))

    def ivf_list_lengths_dev(self, d_tids, S, S_global, d_out_len, stream=0):
        _check(_lib().rii_ivf_list_lengths_dev(self._h, d_tids or None, int(S), int(S_global), d_out_len, stream or None))

    def ivf_shard_max_select_rows(self, L, N_global, S_global=0):
        """Largest `rows` query_ivf_shard_dev selects per query at this shape; more rows per query: ask for rows = L."""
        return int(_lib().rii_ivf_shard_max_select_rows(self._h, int(L), int(N_global), int(S_global)))

    def query_ivf_shard_dev(self, d_queries, B, topk, d_tids, S, S_global, L, N_global, d_glen, G, rank, d_out_ids,
                            d_out_dists, d_out_pos, d_out_nloc, d_out_counts, stream=0, rows=0):
        _check(_lib().rii_query_ivf_shard_dev(self._h, d_queries, int(B), int(topk), d_tids or None, int(S), int(S_global),
                                              int(L), int(N_global), d_glen, int(G), int(rank), int(rows), d_out_ids,
                                              d_out_dists, d_out_pos, d_out_nloc, d_out_counts, stream or None))

    # ---- main.cpp:17-27: one query per call, python lists out ----
    # The reference is called this way (README.md:84-140), so the wrapper's own cost counts: per-thread staging and output buffers
    # with their addresses cached (an ndarray's .ctypes costs a microsecond per use), one ctypes call.
    _SINGLE_MAX_TOPK = 4096

    def _single(self, topk):
        tl = self._tl
        try:
            q = tl.q
        except AttributeError:
            qa = np.empty((1, self._D), np.float32)
            q = tl.q = (qa, qa.ctypes.data)
            tl.out = {}
        o = tl.out.get(topk)
        if o is None:
            if len(tl.out) >= 32:
                tl.out.clear()
            ids, d, cnt = np.empty((1, topk), np.int64), np.empty((1, topk), np.float32), np.empty(1, np.int64)
            o = tl.out[topk] = (ids, d, cnt, ids.ctypes.data, d.ctypes.data, cnt.ctypes.data)
        return q, o

    def _single_query(self, query):
        q = np.asarray(query)
        if q.ndim != 1:
            raise ValueError("query must be 1-D")
        if q.dtype != np.float32:
            raise TypeError("query must be float32 (py::arg(\"query\").noconvert(), src/main.cpp:18)")
        if q.shape[0] != self._D:
            raise ValueError("queries must have shape (B, D=%d)" % self._D)
        return q

    def query_linear(self, query, topk, target_ids):
        q = self._single_query(query)
        topk = int(topk)
        if not 1 <= topk <= self._SINGLE_MAX_TOPK:
            ids, d = self.query_linear_batch(q.reshape(1, -1), topk, target_ids)
            return ids[0].tolist(), d[0].tolist()
        t = self._as_tids(target_ids)
        (qa, qp), (ids, d, _, pi, pd, _) = self._single(topk)
        qa[0] = q
        _check(_lib().rii_query_linear(self._h, qp, 1, topk, _addr(t), t.size, pi, pd))
        return ids[0].tolist(), d[0].tolist()

    def query_ivf(self, query, topk, target_ids, L):
        q = self._single_query(query)
        topk = int(topk)
        if not 1 <= topk <= self._SINGLE_MAX_TOPK:
            ids, d, cnt = self.query_ivf_batch(q.reshape(1, -1), topk, target_ids, L)
            n = int(cnt[0])
            return ids[0, :n].tolist(), d[0, :n].tolist()
        t = self._as_tids(target_ids)
        (qa, qp), (ids, d, cnt, pi, pd, pc) = self._single(topk)
        qa[0] = q
        _check(_lib().rii_query_ivf(self._h, qp, 1, topk, _addr(t), t.size, int(L), pi, pd, pc))
        n = int(cnt[0])
        return ids[0, :n].tolist(), d[0, :n].tolist()

    def dtable(self, queries):
        Q = self._as_query_batch(np.atleast_2d(queries), self.M * self.Ds)
        out = np.empty((Q.shape[0], self.M, self.Ks), np.float32)
        _check(_lib().rii_dtable(self._h, _ptr(Q, ctypes.c_float), Q.shape[0], _ptr(out, ctypes.c_float)))
        return out

    def assign(self, codes):
        c = np.ascontiguousarray(codes, dtype=np.uint8)
        out = np.empty(c.shape[0], np.int32)
        _check(_lib().rii_assign(self._h, _ptr(c, ctypes.c_uint8), c.shape[0], _ptr(out, ctypes.c_int32)))
        return out

    # ---- options / timing ----
    def set_option(self, key, value):
        if key == "lut_mode" and isinstance(value, str):
            value = LUT_MODES[value]
        _check(_lib().rii_set_option(self._h, key.encode(), int(value)))

    def get_option(self, key):
        return int(_lib().rii_get_option(self._h, key.encode()))

    def timing_read(self, kernel):
        ms = ctypes.c_double()
        n = ctypes.c_int64()
        _check(_lib().rii_timing_read(self._h, kernel.encode(), ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def timing_reset(self):
        _check(_lib().rii_timing_reset(self._h))

    def synchronize(self):
        _check(_lib().rii_synchronize(self._h))

    # ---- properties, main.cpp:29-34 ----
    @property
    def M(self):
        return _lib().rii_get_M(self._h)

    @property
    def Ks(self):
        return _lib().rii_get_Ks(self._h)

    @property
    def Ds(self):
        return _lib().rii_get_Ds(self._h)

    @property
    def N(self):
        return int(_lib().rii_get_N(self._h))

    @property
    def nlist(self):
        return int(_lib().rii_get_nlist(self._h))

    @property
    def verbose(self):
        return bool(_lib().rii_get_verbose(self._h))

    @verbose.setter
    def verbose(self, v):
        _check(_lib().rii_set_verbose(self._h, int(bool(v))))

    @property
    def codewords(self):
        out = np.empty((self.M, self.Ks, self.Ds), np.float32)
        _check(_lib().rii_get_codewords(self._h, _ptr(out, ctypes.c_float)))
        return out

    def codes_array(self):
        out = np.empty((self.N, self.M), np.uint8)
        _check(_lib().rii_get_codes(self._h, _ptr(out, ctypes.c_uint8)))
        return out

    def coarse_centers_array(self):
        out = np.empty((self.nlist, self.M), np.uint8)
        _check(_lib().rii_get_coarse_centers(self._h, _ptr(out, ctypes.c_uint8)))
        return out

    def posting_lists_csr(self):
        off = np.zeros(self.nlist + 1, np.int64)
        ids = np.empty(max(self.N, 1), np.int32)
        _check(_lib().rii_get_posting_lists(self._h, _ptr(off, ctypes.c_int64), _ptr(ids, ctypes.c_int32)))
        return off, ids[:off[-1]]

    @property
    def coarse_centers(self):            # list[list[int]] like def_readonly of vector<vector<uchar>>
        return self.coarse_centers_array().tolist()

    @property
    def flattened_codes(self):           # list[int] of N*M
        return self.codes_array().reshape(-1).tolist()

    @property
    def posting_lists(self):             # list[list[int]]
        off, ids = self.posting_lists_csr()
        return [ids[off[i]:off[i + 1]].tolist() for i in range(len(off) - 1)]

    # ---- pickle: exactly the 5-tuple of py::pickle in src/main.cpp:35-53 -- (codewords as nested lists [M][Ks][Ds],
    # verbose, coarse_centers list[list[int]], flattened_codes list[int] of N*M, posting_lists list[list[int]]) -- so a
    # state produced here loads into the reference's RiiCpp.__setstate__ and vice versa.  The fvec_L2sqr variant the
    # engine emulates (`simd_arch`, a property of the reference *build*, not of its state) travels out of band: as a
    # constructor argument of __reduce__ for our own pickles, and as the host default for states coming from a reference.
    def __getstate__(self):
        return (self.codewords.tolist(), self.verbose, self.coarse_centers, self.flattened_codes, self.posting_lists)

    def __reduce__(self):
        return (_rebuild_engine, (self.__class__, self._simd), self.__getstate__())

    def __setstate__(self, t):
        if len(t) != 5:                    # src/main.cpp:41-43
            raise RuntimeError("Invalid state when reading pickled item")
        cw, verbose, centers, codes, lists = t
        simd = getattr(self, "_simd", None)
        if getattr(self, "_h", None):      # __setstate__ on a live engine replaces it
            _lib().rii_destroy(self._h)
        self._h = ctypes.c_void_p()
        cw = np.asarray(cw, np.float32)
        self._create(cw, bool(verbose), simd, None)
        M = cw.shape[0]
        centers = np.ascontiguousarray(np.asarray(centers, np.uint8).reshape(-1, M))
        codes = np.ascontiguousarray(np.asarray(codes, np.uint8).reshape(-1, M))
        off = np.concatenate([[0], np.cumsum([len(l) for l in lists])]).astype(np.int64)
        ids = np.concatenate([np.asarray(l, np.int32) for l in lists]) if len(lists) else np.zeros(0, np.int32)
        off = np.ascontiguousarray(off, np.int64)
        ids = np.ascontiguousarray(ids, np.int32)
        if ids.size == 0:
            ids = np.zeros(1, np.int32)
        _check(_lib().rii_set_state(self._h, _ptr(centers, ctypes.c_uint8), centers.shape[0],
                                    _ptr(codes, ctypes.c_uint8), codes.shape[0], _ptr(off, ctypes.c_int64),
                                    _ptr(ids, ctypes.c_int32)))
