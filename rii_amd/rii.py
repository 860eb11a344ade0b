"""`Rii` -- the user-facing class, a drop-in for `rii.Rii` (rii/rii.py:6-400) running its query hot path on
one MI355X through `RiiGpu` (rii_amd/core.py).  Same method names, argument meaning, defaults, return types
and error behaviour as the reference; the docstrings cite the lines being mirrored.

Additions: `query_batch()` (the reference answers one query per call) and `Rii(..., device=..., simd_arch=...)`.
"""
import copy
import time

import numpy as np

from .core import RiiGpu


def _is_opq(fq):
    return hasattr(fq, "rotate")


class Rii(object):
    """Reconfigurable Inverted Index.  fine_quantizer: a trained PQ/OPQ codec (nanopq.PQ / nanopq.OPQ or the
    stand-ins of rii_amd.codec) -- rii/rii.py:32-38."""

    def __init__(self, fine_quantizer, device=None, simd_arch=None, _impl_factory=None):
        for attr in ("M", "Ks", "codewords", "encode", "decode", "verbose"):
            assert hasattr(fine_quantizer, attr), "fine_quantizer must be a PQ/OPQ codec (missing %s)" % attr
        assert fine_quantizer.codewords is not None, "Please fit the PQ/OPQ instance first"
        assert fine_quantizer.Ks <= 256, "Ks must be less than 256 so that each code must be uint8"
        self.fine_quantizer = copy.deepcopy(fine_quantizer)
        make = _impl_factory or (lambda cw, verbose: RiiGpu(cw, verbose, simd_arch=simd_arch, device=device))
        self.impl_cpp = make(np.ascontiguousarray(fine_quantizer.codewords, dtype=np.float32),
                             bool(fine_quantizer.verbose))
        self.threshold = None

    # ------------------------------------------------------------------ properties, rii/rii.py:40-121
    @property
    def M(self):
        return self.fine_quantizer.M

    @property
    def Ks(self):
        return self.fine_quantizer.Ks

    @property
    def N(self):
        return self.impl_cpp.N

    @property
    def nlist(self):
        return self.impl_cpp.nlist

    @property
    def codewords(self):
        return self.fine_quantizer.codewords

    @property
    def coarse_centers(self):
        if self.nlist == 0:
            return None
        return np.array(self.impl_cpp.coarse_centers, dtype=self.fine_quantizer.code_dtype)

    @property
    def codes(self):
        if self.N == 0:
            return None
        if hasattr(self.impl_cpp, "codes_array"):
            return self.impl_cpp.codes_array().astype(self.fine_quantizer.code_dtype, copy=False)
        return np.array(self.impl_cpp.flattened_codes, dtype=self.fine_quantizer.code_dtype).reshape(self.N, self.M)

    @property
    def posting_lists(self):
        return self.impl_cpp.posting_lists

    @property
    def verbose(self):
        return self.impl_cpp.verbose

    @verbose.setter
    def verbose(self, v):
        self.fine_quantizer.verbose = v
        self.impl_cpp.verbose = v

    @property
    def L0(self):
        if self.nlist == 0:
            return None
        return int(np.round(self.N / self.nlist))

    # ------------------------------------------------------------------ build, rii/rii.py:123-233
    def reconfigure(self, nlist=None, iter=5):
        if nlist is None:
            nlist = int(np.sqrt(self.N))
        assert 0 < nlist
        self.impl_cpp.reconfigure(nlist, iter)
        self.threshold = estimate_best_threshold_function(
            e=self, queries=self.fine_quantizer.decode(self.codes[:min(100, self.N)]))

    def add(self, vecs, update_posting_lists="auto"):
        assert vecs.ndim == 2
        assert vecs.dtype == np.float32
        self.impl_cpp.add_codes(self.fine_quantizer.encode(vecs),
                                self._resolve_update_posting_lists_flag(update_posting_lists))

    def add_configure(self, vecs, nlist=None, iter=5):
        self.add(vecs=vecs, update_posting_lists=False)
        self.reconfigure(nlist=nlist, iter=iter)
        return self

    def merge(self, engine, update_posting_lists="auto"):
        assert isinstance(engine, Rii)
        assert self.fine_quantizer == engine.fine_quantizer, \
            "Two engines to be merged must have the same fine quantizer"
        if engine.N != 0:
            self.impl_cpp.add_codes(engine.codes, self._resolve_update_posting_lists_flag(update_posting_lists))
        if self.verbose:
            print("The number of codes: {}".format(self.N))

    # ------------------------------------------------------------------ query, rii/rii.py:235-320
    def _prepare(self, topk, L, target_ids, sort_target_ids, method):
        assert 0 < self.N
        assert 0 < self.nlist
        assert method in ["auto", "linear", "ivf"]
        if topk is None:
            topk = self.N
        assert 1 <= topk <= self.N
        if L is None:
            L = self._multiple_of_L0_covering_topk(topk=topk)
        assert topk <= L <= self.N, \
            "Parameters are weird. Make sure topk<=L<=N:  topk={}, L={}, N={}".format(topk, L, self.N)
        if target_ids is None:
            tids = np.array([], dtype=np.int64)
            len_target_ids = self.N
        else:
            assert isinstance(target_ids, np.ndarray)
            assert target_ids.dtype == np.int64
            assert target_ids.ndim == 1
            tids = np.sort(target_ids) if sort_target_ids else target_ids
            len_target_ids = len(tids)
        assert topk <= len_target_ids <= self.N, \
            "Parameters are weird. Make sure topk<=len(target_ids)<=N:  " \
            "topk={}, len(target_ids)={}, N={}".format(topk, len_target_ids, self.N)
        if method == "auto":
            method = "linear" if self._use_linear(len_target_ids, L) else "ivf"
        return topk, L, tids, method

    def query(self, q, topk=1, L=None, target_ids=None, sort_target_ids=True, method="auto"):
        """One query -> (ids int64[topk], dists float64[topk]); empty arrays when the inverted index finds
        fewer than L candidates (rii.h:324-325)."""
        topk, L, tids, method = self._prepare(topk, L, target_ids, sort_target_ids, method)
        q_ = self.fine_quantizer.rotate(q) if _is_opq(self.fine_quantizer) else q
        if method == "linear":
            ids, dists = self.impl_cpp.query_linear(q_, topk, tids)
        else:
            ids, dists = self.impl_cpp.query_ivf(q_, topk, tids, L)
        return np.array(ids, np.int64), np.array(dists)

    def query_batch(self, Q, topk=1, L=None, target_ids=None, sort_target_ids=True, method="auto"):
        """NEW: B queries at once (Q: float32 [B, D]); row b equals query(Q[b], ...).
        Returns (ids int64 [B, topk], dists float32 [B, topk], counts int64 [B]); counts[b] is topk, or 0 where
        the inverted index returns nothing for that query."""
        assert Q.ndim == 2 and Q.dtype == np.float32
        topk, L, tids, method = self._prepare(topk, L, target_ids, sort_target_ids, method)
        Q_ = self.fine_quantizer.rotate(Q) if _is_opq(self.fine_quantizer) else Q
        if method == "linear":
            ids, dists = self.impl_cpp.query_linear_batch(Q_, topk, tids)
            return ids, dists, np.full(Q.shape[0], topk, np.int64)
        return self.impl_cpp.query_ivf_batch(Q_, topk, tids, L)

    def clear(self):
        self.threshold = None
        self.impl_cpp.clear()

    def print_params(self):
        """rii/rii.py:330-372."""
        print("verbose:", self.verbose)
        print("M:", self.M)
        print("Ks:", self.Ks)
        print("fine_quantizer:", self.fine_quantizer)
        print("N:", self.N)
        print("nlist:", self.nlist)
        print("L0:", self.L0)
        print("cordwords.shape:", self.codewords.shape)
        cc = self.coarse_centers
        print("coarse_centers.shape:", None if cc is None else cc.shape)
        cs = self.codes
        print("codes.shape:", None if cs is None else cs.shape)
        lens = [len(p) for p in self.posting_lists]
        shown = ", ".join(str(x) for x in lens[:11]) + (", " if lens[:11] else "") + (" ..." if len(lens) > 11 else "")
        print("[len(poslist) for poslist in posting_lists]: [" + shown + "]")
        for topk in (1, 10, 100):
            L = "None" if self.nlist == 0 else self._multiple_of_L0_covering_topk(topk)
            print("_multiple_of_L0_covering_topk(topk={}): {}".format(topk, L))
        print("threshold function thre_{|S|}=f(L):", self.threshold)
        for S in [10 ** (2 + n) for n in range(5)]:
            use_linear = None if self.threshold is None else self._use_linear(S, self.L0)
            print("_use_linear({S}, L={L0}): {use_linear}".format(S=S, L0=self.L0, use_linear=use_linear))

    # ------------------------------------------------------------------ helpers, rii/rii.py:374-400
    def _multiple_of_L0_covering_topk(self, topk):
        avglen = self.L0
        return min((topk // avglen + 1) * avglen, self.N)

    def _use_linear(self, len_target_ids, L):
        return bool(len_target_ids <= self.threshold(L))

    def _resolve_update_posting_lists_flag(self, flag):
        assert flag in ["auto", True, False]
        if flag == "auto":
            return 0 < self.nlist
        return flag


def estimate_best_threshold_function(e, queries):
    """Fit thre_|S| = f(L): the subset size below which the linear scan beats the inverted index, measured
    by timing both on the engine itself (rii/rii.py:403-486: same sweep over L, doubling search over |S|,
    five bisection steps, 1-D line fit).  The engine is called directly, bypassing Rii.query, like the
    reference does (rii.py:411-413)."""
    topk = 1
    impl = e.impl_cpp
    queries = np.ascontiguousarray(queries, dtype=np.float32)

    def run(qs, tids, L, method):
        t0 = time.time()
        for q in qs:
            if method == "linear":
                impl.query_linear(q, topk, tids)
            else:
                impl.query_ivf(q, topk, tids, L)
        return (time.time() - t0) / qs.shape[0]

    def sweep(L):
        if e.N <= 128:
            return e.N
        sids = [128]
        while sids[-1] * 2 < e.N:
            sids.append(sids[-1] * 2)
        sids.append(e.N)
        for s in sids:
            tids = np.arange(s, dtype=np.int64)
            if run(queries[:3], tids, L, "ivf") < run(queries[:3], tids, L, "linear"):
                if s == 128:
                    if e.verbose:
                        print("ivf is faster than linear scan even if |S|<=128. This is a bit weird. "
                              "Anyway let's set threshold as 128")
                    return 128
                s0, s1 = int(s / 2), s
                for _ in range(5):
                    s_mid = int(np.round((s0 + s1) / 2))
                    tids = np.arange(s_mid, dtype=np.int64)
                    if run(queries, tids, L, "ivf") < run(queries, tids, L, "linear"):
                        s1 = s_mid
                    else:
                        s0 = s_mid
                return s0
        return e.N

    if e.verbose:
        print("===== Threshold selection ====")
    xs, ys = [], []
    for L in [k * e._multiple_of_L0_covering_topk(k) for k in [1, 2, 4, 8, 16]]:
        if e.N < L:
            continue
        xs.append(L)
        ys.append(sweep(L))
        if ys[-1] == e.N:
            break
    z = [0, ys[0]] if len(xs) == 1 else np.polyfit(xs, ys, 1)
    p = np.poly1d(z)
    if e.verbose:
        print("L:", xs)
        print("threshold:", ys)
        print("polyfit coeff:", z)
        print("resultant func:", p)
    return p
