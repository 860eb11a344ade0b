"""User-facing index class: same public surface as the reference's `rii.Rii` (rii/rii.py:6-400 there) -- constructor
argument, properties, `add` / `add_configure` / `reconfigure` / `merge` / `query` / `clear` / `print_params`, argument
meaning, defaults, return dtypes and error behaviour (AssertionError on violated preconditions) -- written against the
MI355X engine (`RiiGpu`, rii_amd/core.py), plus `query_batch` which the reference does not have.

Structure: the argument policy of a search lives in `_SearchPlan` (one place for `query` and `query_batch`), the
linear-vs-inverted-index crossover is learnt by `CrossoverModel` (the engine timed like the reference times its CPU path,
rii/rii.py:403-486 -- same idea, same `threshold(L)` contract: an `np.poly1d`).  Two models are kept: `threshold` (one query
per call: what `query` experiences, the reference's attribute) and `threshold_batch` (a batch per call: per-query cost is
~100x lower there and the crossover sits elsewhere), used by `query_batch(method="auto")`; both are fitted by `reconfigure`.
"""
import copy
import time

import numpy as np

from .core import RiiGpu

_METHODS = ("auto", "linear", "ivf")


def _require(cond, msg=""):
    if not cond:
        raise AssertionError(msg)


class _SearchPlan(object):
    """Everything `query` decides before touching the engine (reference: rii/rii.py:276-318)."""

    __slots__ = ("topk", "L", "tids", "n_targets", "method")

    def __init__(self, index, topk, L, target_ids, sort_target_ids, method, batched=False):
        N = index.N
        _require(N > 0, "the index holds no vectors yet")
        _require(index.nlist > 0, "posting lists are missing: call add_configure() or reconfigure() first")
        _require(method in _METHODS, "method must be one of %s" % (_METHODS,))
        self.topk = N if topk is None else topk
        _require(1 <= self.topk <= N, "topk must lie in [1, N]: topk={}, N={}".format(self.topk, N))
        self.L = index._multiple_of_L0_covering_topk(self.topk) if L is None else L
        _require(self.topk <= self.L <= N,
                 "need topk <= L <= N, got topk={}, L={}, N={}".format(self.topk, self.L, N))
        if target_ids is None:
            self.tids = np.empty(0, dtype=np.int64)
            self.n_targets = N
        else:
            _require(isinstance(target_ids, np.ndarray), "target_ids must be a numpy array")
            _require(target_ids.dtype == np.int64, "target_ids must have dtype int64")
            _require(target_ids.ndim == 1, "target_ids must be one-dimensional")
            self.tids = np.sort(target_ids) if sort_target_ids else target_ids
            self.n_targets = int(self.tids.shape[0])
        _require(self.topk <= self.n_targets <= N,
                 "need topk <= len(target_ids) <= N, got topk={}, len(target_ids)={}, N={}".format(
                     self.topk, self.n_targets, N))
        if method == "auto":
            method = "linear" if index._use_linear(self.n_targets, self.L, batched=batched) else "ivf"
        self.method = method


class Rii(object):
    """Reconfigurable inverted index over PQ / OPQ codes, searched on the GPU.

    fine_quantizer: a *trained* product quantiser exposing M, Ks, codewords, code_dtype, verbose, encode(), decode() and,
    for OPQ, rotate() -- `nanopq.PQ` / `nanopq.OPQ`, or the stand-ins in `rii_amd.codec`.
    """

    def __init__(self, fine_quantizer, device=None, simd_arch=None, _impl_factory=None):
        missing = [a for a in ("M", "Ks", "codewords", "encode", "decode", "verbose") if not hasattr(fine_quantizer, a)]
        _require(not missing, "fine_quantizer is not a PQ/OPQ codec (no %s)" % ", ".join(missing))
        _require(fine_quantizer.codewords is not None, "Please fit the PQ/OPQ instance first")
        _require(fine_quantizer.Ks <= 256, "Ks must be at most 256: codes are stored as uint8")
        self.fine_quantizer = copy.deepcopy(fine_quantizer)
        table = np.ascontiguousarray(fine_quantizer.codewords, dtype=np.float32)
        if _impl_factory is None:
            self.impl_cpp = RiiGpu(table, bool(fine_quantizer.verbose), simd_arch=simd_arch, device=device)
        else:                                   # tests inject another engine with the same surface
            self.impl_cpp = _impl_factory(table, bool(fine_quantizer.verbose))
        self.threshold = None
        self.threshold_batch = None

    # ---- read-only views ------------------------------------------------------------------------------------------
    M = property(lambda self: self.fine_quantizer.M)
    Ks = property(lambda self: self.fine_quantizer.Ks)
    N = property(lambda self: self.impl_cpp.N)
    nlist = property(lambda self: self.impl_cpp.nlist)
    codewords = property(lambda self: self.fine_quantizer.codewords)
    posting_lists = property(lambda self: self.impl_cpp.posting_lists)

    @property
    def coarse_centers(self):
        """uint8 array (nlist, M) of PQ-coded coarse centres, or None before the first (re)configuration."""
        if not self.nlist:
            return None
        return np.asarray(self.impl_cpp.coarse_centers, dtype=self.fine_quantizer.code_dtype)

    @property
    def codes(self):
        """The stored PQ codes as an (N, M) array, or None while empty."""
        n = self.N
        if not n:
            return None
        fetch = getattr(self.impl_cpp, "codes_array", None)
        raw = fetch() if fetch is not None else np.asarray(self.impl_cpp.flattened_codes)
        return np.asarray(raw, dtype=self.fine_quantizer.code_dtype).reshape(n, self.M)

    @property
    def verbose(self):
        return self.impl_cpp.verbose

    @verbose.setter
    def verbose(self, flag):
        self.impl_cpp.verbose = flag
        self.fine_quantizer.verbose = flag

    @property
    def L0(self):
        """Average posting-list length, rounded; None without posting lists."""
        lists = self.nlist
        return int(np.round(self.N / lists)) if lists else None

    # ---- building ---------------------------------------------------------------------------------------------------
    def add(self, vecs, update_posting_lists="auto"):
        _require(vecs.ndim == 2, "vecs must be (Nv, D)")
        _require(vecs.dtype == np.float32, "vecs must be float32")
        self.impl_cpp.add_codes(self.fine_quantizer.encode(vecs), self._resolve_update_posting_lists_flag(update_posting_lists))

    def reconfigure(self, nlist=None, iter=5):
        nlist = int(np.sqrt(self.N)) if nlist is None else nlist
        _require(nlist > 0, "nlist must be positive")
        self.impl_cpp.reconfigure(nlist, iter)
        probes = self.fine_quantizer.decode(self.codes[:min(100, self.N)])
        self.threshold = estimate_best_threshold_function(e=self, queries=probes)
        self.threshold_batch = None
        if hasattr(self.impl_cpp, "query_linear_batch"):       # engines with a batch entry point: the batched crossover as well
            self.threshold_batch = CrossoverModel(self, self.fine_quantizer.decode(self.codes[:min(256, self.N)]), batched=True).fit()

    def add_configure(self, vecs, nlist=None, iter=5):
        self.add(vecs=vecs, update_posting_lists=False)
        self.reconfigure(nlist=nlist, iter=iter)
        return self

    def merge(self, engine, update_posting_lists="auto"):
        _require(isinstance(engine, Rii), "can only merge another Rii")
        _require(self.fine_quantizer == engine.fine_quantizer, "Two engines to be merged must have the same fine quantizer")
        if engine.N:
            self.impl_cpp.add_codes(engine.codes, self._resolve_update_posting_lists_flag(update_posting_lists))
        if self.verbose:
            print("The number of codes: {}".format(self.N))

    def clear(self):
        self.impl_cpp.clear()
        self.threshold = None
        self.threshold_batch = None

    # ---- searching --------------------------------------------------------------------------------------------------
    def _rotated(self, x):
        return self.fine_quantizer.rotate(x) if hasattr(self.fine_quantizer, "rotate") else x

    def query(self, q, topk=1, L=None, target_ids=None, sort_target_ids=True, method="auto"):
        """One query vector -> (ids int64 [n], dists float64 [n]); n == topk, or 0 when the inverted index cannot collect
        L candidates (the reference's empty return)."""
        plan = _SearchPlan(self, topk, L, target_ids, sort_target_ids, method)
        qv = self._rotated(q)
        if plan.method == "linear":
            ids, dists = self.impl_cpp.query_linear(qv, plan.topk, plan.tids)
        else:
            ids, dists = self.impl_cpp.query_ivf(qv, plan.topk, plan.tids, plan.L)
        return np.array(ids, np.int64), np.array(dists)

    def query_batch(self, Q, topk=1, L=None, target_ids=None, sort_target_ids=True, method="auto"):
        """B queries in one call (Q float32 [B, D]).  Row b equals `query(Q[b], ..., method=m)` for the ONE method m the whole batch
        resolves to: with method="linear" / "ivf" that is the method asked for; with method="auto" the batch picks it from the BATCHED
        crossover model (`threshold_batch`, fitted by `reconfigure` / `add_configure` -- never here: a query call does not time
        anything or mutate the index; an index unpickled from a state without it falls back to `threshold`), which can differ from what
        a single `query` call would pick for the same arguments -- and since the inverted index is approximate, so can the rows.
        Returns (ids int64 [B, topk], dists float32 [B, topk], counts int64 [B]) with counts[b] in {topk, 0}."""
        _require(Q.ndim == 2 and Q.dtype == np.float32, "Q must be float32 (B, D)")
        plan = _SearchPlan(self, topk, L, target_ids, sort_target_ids, method, batched=True)
        Qv = self._rotated(Q)
        if plan.method == "linear":
            ids, dists = self.impl_cpp.query_linear_batch(Qv, plan.topk, plan.tids)
            return ids, dists, np.full(Q.shape[0], plan.topk, np.int64)
        return self.impl_cpp.query_ivf_batch(Qv, plan.topk, plan.tids, plan.L)

    # ---- diagnostics ------------------------------------------------------------------------------------------------
    def print_params(self):
        cc, cs = self.coarse_centers, self.codes
        lens = [len(p) for p in self.posting_lists]
        rows = [("verbose", self.verbose), ("M", self.M), ("Ks", self.Ks), ("fine_quantizer", self.fine_quantizer),
                ("N", self.N), ("nlist", self.nlist), ("L0", self.L0), ("codewords.shape", self.codewords.shape),
                ("coarse_centers.shape", None if cc is None else cc.shape), ("codes.shape", None if cs is None else cs.shape),
                ("first posting-list lengths", lens[:11] + (["..."] if len(lens) > 11 else []))]
        for name, value in rows:
            print("{}: {}".format(name, value))
        for k in (1, 10, 100):
            print("default L for topk={}: {}".format(k, None if not self.nlist else self._multiple_of_L0_covering_topk(k)))
        print("threshold function thre_{|S|}=f(L):", self.threshold)
        for exp in range(2, 7):
            S = 10 ** exp
            verdict = None if self.threshold is None else self._use_linear(S, self.L0)
            print("linear scan preferred for |S|={} at L=L0: {}".format(S, verdict))

    # ---- policy helpers (names kept: the reference's tests and users reach for them) -------------------------------------
    def _multiple_of_L0_covering_topk(self, topk):
        step = self.L0
        return min(step * (topk // step + 1), self.N)

    def _use_linear(self, len_target_ids, L, batched=False):
        fb = getattr(self, "threshold_batch", None)          # (absent in pickles written before it existed)
        f = fb if (batched and fb is not None) else self.threshold
        return bool(len_target_ids <= f(L))

    def _resolve_update_posting_lists_flag(self, flag):
        _require(flag in ("auto", True, False), "update_posting_lists must be 'auto', True or False")
        return (self.nlist > 0) if flag == "auto" else flag


class _FlooredLine(np.poly1d):
    """thre_|S| = max(a L + b, floor): an np.poly1d (what the reference's estimator returns) that never falls below `floor`."""

    def __init__(self, coeff, floor=0.0):
        super().__init__(coeff)
        self.__dict__["floor"] = floor

    def __call__(self, val):
        return np.maximum(super().__call__(val), self.__dict__.get("floor", 0.0))


class CrossoverModel(object):
    """Learns, per candidate budget L, the subset size |S|* at which the inverted index starts to beat the linear scan,
    by timing both on the engine itself, and fits |S|* = f(L) with a line.  batched=False: one query per call (the decision
    `Rii.query` experiences: exhaustive single-query scan, pinned staging); batched=True: all probes in one call (what
    `query_batch` experiences: tile tables, filter scan, one launch group per batch)."""

    def __init__(self, index, probes, rounds=5, batched=False):
        self.index, self.impl = index, index.impl_cpp
        self.probes = np.ascontiguousarray(probes, dtype=np.float32)
        self.rounds = rounds
        self.batched = batched

    def _seconds(self, method, tids, L, few):
        """Per-query seconds: the MINIMUM over a few repetitions (a timing is only ever inflated by noise, never deflated)."""
        best = float("inf")
        if self.batched:
            for _ in range(2 if few else 4):
                t0 = time.perf_counter()
                if method == "linear":
                    self.impl.query_linear_batch(self.probes, 1, tids)
                else:
                    self.impl.query_ivf_batch(self.probes, 1, tids, L)
                best = min(best, (time.perf_counter() - t0) / len(self.probes))
            return best
        qs = self.probes[:3] if few else self.probes
        for _ in range(2 if few else 3):
            t0 = time.perf_counter()
            for q in qs:
                if method == "linear":
                    self.impl.query_linear(q, 1, tids)
                else:
                    self.impl.query_ivf(q, 1, tids, L)
            best = min(best, (time.perf_counter() - t0) / len(qs))
        return best

    def _ivf_wins(self, s, L, few):
        tids = np.arange(s, dtype=np.int64)
        return self._seconds("ivf", tids, L, few) < self._seconds("linear", tids, L, few)

    def crossover(self, L):
        N = self.index.N
        if N <= 128:
            return N
        s = max(128, min(int(L), N))                  # |S| < L makes no sense for the inverted index (it must collect L candidates)
        first = s
        while True:                                   # doubling search: first, 2 first, ..., N
            if self._ivf_wins(s, L, few=True):
                break
            if s == N:
                return N                              # the linear scan never loses
            s = N if s * 2 >= N else s * 2
        if s == first:
            if self.index.verbose:
                print("the inverted index already wins at |S|=%d; using it as the crossover" % first)
            return first
        lo, hi = s // 2, s
        for _ in range(self.rounds):                  # bisection between the last loss and the first win
            mid = int(np.round((lo + hi) / 2))
            if self._ivf_wins(mid, L, few=False):
                hi = mid
            else:
                lo = mid
        return lo

    def fit(self):
        idx = self.index
        Ls, cuts = [], []
        for k in (1, 2, 4, 8, 16):
            L = k * idx._multiple_of_L0_covering_topk(k)
            if L > idx.N:
                continue
            Ls.append(L)
            cuts.append(self.crossover(L))
            if cuts[-1] == idx.N:
                break
        coeff = [0, cuts[0]] if len(Ls) == 1 else np.polyfit(Ls, cuts, 1)
        # a larger candidate budget can only make the inverted index dearer, so the crossover cannot fall with L: a negative
        # slope is timing noise -- fall back to the mean crossover (constant in L)
        if len(Ls) > 1 and (not np.all(np.isfinite(coeff)) or coeff[0] < 0):
            coeff = [0, float(np.mean(cuts))]
        # the line is a fit through noisy points and may dip below every crossover that was actually measured at the small-L end
        # (even below zero): never answer less than the smallest measured one
        model = _FlooredLine(coeff, floor=float(min(cuts)))
        if idx.verbose:
            print("crossover |S|* per L:", dict(zip(Ls, cuts)), "->", model)
        return model


def estimate_best_threshold_function(e, queries):
    """Module-level entry point kept for parity with the reference (rii/rii.py:403): returns the np.poly1d f with
    `thre_|S| = f(L)`."""
    return CrossoverModel(e, queries).fit()
