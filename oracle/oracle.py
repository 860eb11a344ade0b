"""ctypes front-end of the CPU oracle (oracle/rii_oracle.c) and loader of the real reference build.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under rii_amd/ imports this module.

`OracleRii` mirrors the surface of the reference's pybind11 class `RiiCpp` (src/main.cpp:12-54) so that the
same driver code can run against (a) this restatement, (b) the real reference (`load_reference()`), and
(c) the HIP engine (rii_amd.core.RiiGpu).
"""
import ctypes
import importlib.util
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SIMD = {"sse": 0, "avx": 1, "avx512": 2}


def host_simd_arch():
    """Which fvec_L2sqr variant `-march=native` would select on this host (src/distance.h:113,172,219)."""
    env = os.environ.get("RII_SIMD_ARCH")
    if env:
        return env
    flags = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    flags = line
                    break
    except OSError:
        pass
    toks = set(flags.split())
    if "avx512f" in toks:
        return "avx512"
    if "avx" in toks:
        return "avx"
    return "sse"


def build(force=False):
    so = os.path.join(_HERE, "librii_oracle.so")
    src = os.path.join(_HERE, "rii_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        f32p = ctypes.POINTER(ctypes.c_float)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        i64p = ctypes.POINTER(ctypes.c_int64)
        i32p = ctypes.POINTER(ctypes.c_int32)
        c_int, c_i64 = ctypes.c_int, ctypes.c_int64
        L.oracle_fvec_l2sqr.restype = ctypes.c_float
        L.oracle_fvec_l2sqr.argtypes = [f32p, f32p, ctypes.c_size_t, c_int]
        L.oracle_dtable.argtypes = [f32p, c_int, c_int, c_int, f32p, c_int, f32p]
        L.oracle_adist.restype = ctypes.c_float
        L.oracle_adist.argtypes = [f32p, c_int, c_int, u8p]
        L.oracle_query_linear.restype = c_i64
        L.oracle_query_linear.argtypes = [f32p, c_int, c_int, c_int, u8p, c_i64, f32p, c_int, i64p, c_i64,
                                          c_int, i64p, f32p]
        L.oracle_query_ivf.restype = c_i64
        L.oracle_query_ivf.argtypes = [f32p, c_int, c_int, c_int, u8p, c_i64, u8p, c_i64, i64p, i32p, f32p,
                                       c_int, i64p, c_i64, c_i64, c_int, i64p, f32p]
        L.oracle_l2sq_pqkmeans.restype = ctypes.c_float
        L.oracle_l2sq_pqkmeans.argtypes = [f32p, f32p, c_int, c_int]
        L.oracle_symmetric_tables.argtypes = [f32p, c_int, c_int, c_int, c_int, f32p]
        L.oracle_assign.argtypes = [f32p, c_int, c_int, u8p, c_i64, u8p, c_i64, i32p]
        L.oracle_reconfigure_sample.restype = c_i64
        L.oracle_reconfigure_sample.argtypes = [c_i64, c_i64, i64p]
        L.oracle_pqkmeans_fit.argtypes = [f32p, c_int, c_int, u8p, c_i64, c_i64, c_int, u8p, i32p]
        L.oracle_version.restype = ctypes.c_char_p
        _LIB = L
    return _LIB


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def dtable(codewords, q, arch=None):
    arch = arch or host_simd_arch()
    cw = np.ascontiguousarray(codewords, np.float32)
    M, Ks, Ds = cw.shape
    q = np.ascontiguousarray(q, np.float32)
    out = np.empty((M, Ks), np.float32)
    lib().oracle_dtable(_p(cw, ctypes.c_float), M, Ks, Ds, _p(q, ctypes.c_float), SIMD[arch],
                        _p(out, ctypes.c_float))
    return out


def symmetric_tables(codewords, arch=None):
    arch = arch or host_simd_arch()
    cw = np.ascontiguousarray(codewords, np.float32)
    M, Ks, Ds = cw.shape
    D = np.empty((M, Ks, Ks), np.float32)
    lib().oracle_symmetric_tables(_p(cw, ctypes.c_float), M, Ks, Ds, SIMD[arch], _p(D, ctypes.c_float))
    return D


class OracleRii(object):
    """CPU restatement with the RiiCpp surface (src/main.cpp:12-54, src/rii.h:40-83)."""

    def __init__(self, codewords, verbose=False, simd_arch=None):
        self.codewords = np.array(codewords, dtype=np.float32, order="C")   # copy, like rii.h:86-100
        assert self.codewords.ndim == 3
        self.M, self.Ks, self.Ds = self.codewords.shape
        self.verbose = verbose
        self.arch = simd_arch or host_simd_arch()
        self.codes = np.zeros((0, self.M), np.uint8)
        self.centers = np.zeros((0, self.M), np.uint8)
        self._lists = []          # list of python lists of int
        self._D = None

    # ----- properties of main.cpp:29-34 -----
    @property
    def N(self):
        return int(self.codes.shape[0])

    @property
    def nlist(self):
        return int(self.centers.shape[0])

    @property
    def coarse_centers(self):
        return self.centers.tolist()

    @property
    def flattened_codes(self):
        return self.codes.reshape(-1).tolist()

    @property
    def posting_lists(self):
        return [list(l) for l in self._lists]

    def _tables(self):
        if self._D is None:
            self._D = symmetric_tables(self.codewords, self.arch)
        return self._D

    def set_csr(self, centers, off, ids):
        """Centres and posting lists handed over in CSR form (large synthetic indices: no python list per posting)."""
        self.centers = np.ascontiguousarray(centers, np.uint8)
        self._csr_given = (np.ascontiguousarray(off, np.int64), np.ascontiguousarray(ids, np.int32))
        self._lists = None

    def _csr(self):
        if getattr(self, "_csr_given", None) is not None and self._lists is None:
            return self._csr_given
        off = np.zeros(self.nlist + 1, np.int64)
        for i, l in enumerate(self._lists):
            off[i + 1] = off[i] + len(l)
        ids = np.zeros(max(int(off[-1]), 1), np.int32)
        for i, l in enumerate(self._lists):
            ids[off[i]:off[i + 1]] = l
        return off, ids

    # ----- src/rii.h:335-359 -----
    def update_posting_lists(self, start, num):
        if num == 0:
            return
        D = self._tables()
        sub = np.ascontiguousarray(self.codes[start:start + num])
        assign = np.empty(num, np.int32)
        lib().oracle_assign(_p(D, ctypes.c_float), self.M, self.Ks, _p(sub, ctypes.c_uint8), num,
                            _p(self.centers, ctypes.c_uint8), self.nlist, _p(assign, ctypes.c_int32))
        for n in range(num):
            self._lists[assign[n]].append(start + n)

    # ----- src/rii.h:158-193 -----
    def add_codes(self, codes, update_flag):
        if update_flag and self.nlist == 0:
            raise RuntimeError("reconfigure() must be called before add(update_posting_lists=True)")
        codes = np.ascontiguousarray(codes, np.uint8)
        assert codes.ndim == 2 and codes.shape[1] == self.M
        N0 = self.N
        self.codes = np.ascontiguousarray(np.concatenate([self.codes, codes], axis=0))
        if update_flag:
            self.update_posting_lists(N0, codes.shape[0])

    # ----- src/rii.h:108-156 -----
    def reconfigure(self, nlist, iter):
        assert 0 < nlist <= self.N
        sample = np.empty(min(self.N, nlist * 100), np.int64)
        n = lib().oracle_reconfigure_sample(self.N, nlist, _p(sample, ctypes.c_int64))
        data = np.ascontiguousarray(self.codes[sample[:n]])
        D = self._tables()
        centers = np.zeros((nlist, self.M), np.uint8)
        lib().oracle_pqkmeans_fit(_p(D, ctypes.c_float), self.M, self.Ks, _p(data, ctypes.c_uint8), n, nlist,
                                  iter, _p(centers, ctypes.c_uint8), None)
        self.centers = centers
        self._lists = [[] for _ in range(nlist)]
        self.update_posting_lists(0, self.N)

    def set_coarse_centers(self, centers):
        """(not in the reference) import centres, then rebuild lists as rii.h:150-155 does."""
        self.centers = np.ascontiguousarray(centers, np.uint8)
        self._lists = [[] for _ in range(self.nlist)]
        self.update_posting_lists(0, self.N)

    # ----- py::pickle, src/main.cpp:35-53: the reference's 5-tuple -----
    def __getstate__(self):
        return (self.codewords.tolist(), bool(self.verbose), self.coarse_centers, self.flattened_codes, self.posting_lists)

    def __setstate__(self, t):
        if len(t) != 5:
            raise RuntimeError("Invalid state when reading pickled item")
        arch = getattr(self, "arch", None)
        self.__init__(np.asarray(t[0], np.float32), bool(t[1]), arch)
        self.centers = np.ascontiguousarray(np.asarray(t[2], np.uint8).reshape(-1, self.M))
        self.codes = np.ascontiguousarray(np.asarray(t[3], np.uint8).reshape(-1, self.M))
        self._lists = [list(l) for l in t[4]]

    def clear(self):                                    # src/rii.h:328-333
        self.codes = np.zeros((0, self.M), np.uint8)
        self.centers = np.zeros((0, self.M), np.uint8)
        self._lists = []

    # ----- src/rii.h:195-242 -----
    def query_linear(self, query, topk, target_ids):
        q = np.ascontiguousarray(query, np.float32)
        t = np.ascontiguousarray(target_ids, np.int64)
        assert topk <= self.N and (t.size == 0 or topk <= t.size <= self.N)
        ids = np.empty(topk, np.int64)
        d = np.empty(topk, np.float32)
        lib().oracle_query_linear(_p(self.codewords, ctypes.c_float), self.M, self.Ks, self.Ds,
                                  _p(self.codes, ctypes.c_uint8), self.N, _p(q, ctypes.c_float), topk,
                                  _p(t, ctypes.c_int64), t.size, SIMD[self.arch], _p(ids, ctypes.c_int64),
                                  _p(d, ctypes.c_float))
        return ids.tolist(), [float(x) for x in d]

    # ----- src/rii.h:244-326 -----
    def query_ivf(self, query, topk, target_ids, L):
        q = np.ascontiguousarray(query, np.float32)
        t = np.ascontiguousarray(target_ids, np.int64)
        assert topk <= self.N and topk <= L <= self.N
        off, pl = self._csr()
        ids = np.empty(topk, np.int64)
        d = np.empty(topk, np.float32)
        n = lib().oracle_query_ivf(_p(self.codewords, ctypes.c_float), self.M, self.Ks, self.Ds,
                                   _p(self.codes, ctypes.c_uint8), self.N, _p(self.centers, ctypes.c_uint8),
                                   self.nlist, _p(off, ctypes.c_int64), _p(pl, ctypes.c_int32),
                                   _p(q, ctypes.c_float), topk, _p(t, ctypes.c_int64), t.size, L,
                                   SIMD[self.arch], _p(ids, ctypes.c_int64), _p(d, ctypes.c_float))
        return ids[:n].tolist(), [float(x) for x in d[:n]]


# --------------------------------------------------------------------------------------------------
# The real reference, compiled by `make -C oracle ref` into oracle/_ref/{native,v3}/main*.so.
# --------------------------------------------------------------------------------------------------
def _try_import_ref(subdir):
    d = os.path.join(_HERE, "_ref", subdir)
    if not os.path.isdir(d):
        return None
    sos = [f for f in os.listdir(d) if f.startswith("main") and f.endswith(".so")]
    if not sos:
        return None
    path = os.path.join(d, sos[0])
    # probe in a subprocess first: a -march=native object may SIGILL on a different host CPU
    code = ("import importlib.util,sys,numpy as np;"
            "s=importlib.util.spec_from_file_location('main',%r);m=importlib.util.module_from_spec(s);"
            "s.loader.exec_module(m);"
            "e=m.RiiCpp(np.zeros((2,4,20),np.float32),False);e.add_codes(np.zeros((8,2),np.uint8),False);"
            "e.query_linear(np.zeros(40,np.float32),1,np.array([],np.int64))") % path
    try:
        r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           timeout=120)
    except Exception:
        return None
    if r.returncode != 0:
        return None
    spec = importlib.util.spec_from_file_location("main", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_REF = {}


def load_reference(prefer="native"):
    """Returns (module `main`, simd_arch string, flavour) or (None, None, None) when no build is usable."""
    if "r" in _REF:
        return _REF["r"]
    prefer = os.environ.get("RII_REF_FLAVOUR", prefer)
    order = ["native", "v3"] if prefer == "native" else ["v3", "native"]
    for sub in order:
        mod = _try_import_ref(sub)
        if mod is not None:
            # the fvec_L2sqr variant is fixed at *compile* time (recorded by the Makefile)
            with open(os.path.join(_HERE, "_ref", sub, "ARCH")) as f:
                arch = f.read().strip()
            _REF["r"] = (mod, arch, sub)
            return _REF["r"]
    _REF["r"] = (None, None, None)
    return _REF["r"]
