"""Minimal product-quantisation codecs standing in for `nanopq.PQ` / `nanopq.OPQ`.

The reference delegates training/encoding/decoding/rotation to the third-party package nanopq (unpinned,
requirements.txt:2; used at rii/rii.py:33-37,150,185,305-308).  nanopq is not installable here (no network),
and codec training is upstream of the query hot path, so this module provides duck-typed equivalents exposing
the attributes `Rii` reads: M, Ks, Ds, codewords, code_dtype, verbose, encode(), decode() and, for OPQ,
rotate().  Parity with nanopq's own k-means is NOT claimed ("parity unpinned" at this boundary):
`Rii` treats codebooks, codes and (rotated) queries as given inputs.  A real nanopq object can be passed to
`Rii` unchanged.
"""
import numpy as np


def _kmeans(x, k, iters, rng, verbose=False):
    """Lloyd's k-means, initial centres = k distinct random points (like scipy's minit='points')."""
    n = x.shape[0]
    if n < k:
        raise ValueError("need at least Ks=%d training vectors, got %d" % (k, n))
    cent = x[rng.choice(n, k, replace=False)].astype(np.float32).copy()
    xx = (x * x).sum(1)
    for it in range(iters):
        d = xx[:, None] - 2.0 * (x @ cent.T) + (cent * cent).sum(1)[None, :]
        lab = d.argmin(1)
        for j in range(k):
            sel = lab == j
            if sel.any():
                cent[j] = x[sel].mean(0)
    return cent


def _assign(x, cent, chunk=65536):
    out = np.empty(x.shape[0], np.int64)
    cc = (cent * cent).sum(1)[None, :]
    for s in range(0, x.shape[0], chunk):
        xs = x[s:s + chunk]
        d = (xs * xs).sum(1)[:, None] - 2.0 * (xs @ cent.T) + cc
        out[s:s + chunk] = d.argmin(1)
    return out


class PQ(object):
    """Product quantiser: D-dim vectors -> M bytes (Ks <= 256)."""

    def __init__(self, M, Ks=256, verbose=True):
        assert 0 < Ks <= 2 ** 32
        self.M, self.Ks, self.verbose = M, Ks, verbose
        self.code_dtype = np.uint8 if Ks <= 2 ** 8 else (np.uint16 if Ks <= 2 ** 16 else np.uint32)
        self.codewords = None
        self.Ds = None

    def __eq__(self, other):
        if not isinstance(other, PQ):
            return False
        return (self.M, self.Ks, self.verbose, self.code_dtype, self.Ds) == \
               (other.M, other.Ks, other.verbose, other.code_dtype, other.Ds) and \
            np.array_equal(self.codewords, other.codewords)

    def fit(self, vecs, iter=20, seed=123):
        assert vecs.dtype == np.float32 and vecs.ndim == 2
        N, D = vecs.shape
        assert self.Ks < N, "the number of training vectors should be more than Ks"
        assert D % self.M == 0, "input dimension must be dividable by M"
        self.Ds = D // self.M
        rng = np.random.default_rng(seed)
        self.codewords = np.zeros((self.M, self.Ks, self.Ds), np.float32)
        for m in range(self.M):
            if self.verbose:
                print("Training the subspace: {} / {}".format(m, self.M))
            sub = np.ascontiguousarray(vecs[:, m * self.Ds:(m + 1) * self.Ds])
            self.codewords[m] = _kmeans(sub, self.Ks, iter, rng)
        return self

    def encode(self, vecs):
        assert vecs.dtype == np.float32 and vecs.ndim == 2
        N, D = vecs.shape
        assert D == self.Ds * self.M, "input dimension must be Ds * M"
        codes = np.empty((N, self.M), self.code_dtype)
        for m in range(self.M):
            sub = np.ascontiguousarray(vecs[:, m * self.Ds:(m + 1) * self.Ds])
            codes[:, m] = _assign(sub, self.codewords[m])
        return codes

    def decode(self, codes):
        assert codes.ndim == 2 and codes.shape[1] == self.M
        vecs = np.empty((codes.shape[0], self.Ds * self.M), np.float32)
        for m in range(self.M):
            vecs[:, m * self.Ds:(m + 1) * self.Ds] = self.codewords[m][codes[:, m], :]
        return vecs


class OPQ(object):
    """Optimised PQ (non-parametric): an orthonormal rotation R learnt by alternating PQ / Procrustes."""

    def __init__(self, M, Ks=256, verbose=True):
        self.pq = PQ(M, Ks, verbose)
        self.R = None

    def __eq__(self, other):
        return isinstance(other, OPQ) and self.pq == other.pq and np.array_equal(self.R, other.R)

    @property
    def M(self):
        return self.pq.M

    @property
    def Ks(self):
        return self.pq.Ks

    @property
    def Ds(self):
        return self.pq.Ds

    @property
    def verbose(self):
        return self.pq.verbose

    @verbose.setter
    def verbose(self, v):
        self.pq.verbose = v

    @property
    def code_dtype(self):
        return self.pq.code_dtype

    @property
    def codewords(self):
        return self.pq.codewords

    def fit(self, vecs, pq_iter=20, rotation_iter=10, seed=123):
        assert vecs.dtype == np.float32 and vecs.ndim == 2
        D = vecs.shape[1]
        R = np.eye(D, dtype=np.float32)
        for i in range(rotation_iter):
            if self.verbose:
                print("OPQ rotation training: {} / {}".format(i, rotation_iter))
            X = vecs @ R
            pq = PQ(self.M, self.Ks, self.verbose).fit(X, iter=(pq_iter if i == rotation_iter - 1 else 1), seed=seed)
            X_ = pq.decode(pq.encode(X))
            U, _, Vt = np.linalg.svd(vecs.T @ X_)
            self.pq = pq
            self.R = R
            if i != rotation_iter - 1:
                R = (U @ Vt).astype(np.float32)
        return self

    def rotate(self, vecs):
        assert vecs.dtype == np.float32 and vecs.ndim in (1, 2)
        if vecs.ndim == 2:
            return vecs @ self.R
        return (vecs.reshape(1, -1) @ self.R).reshape(-1)

    def encode(self, vecs):
        return self.pq.encode(self.rotate(vecs))

    def decode(self, codes):
        return self.pq.decode(codes) @ self.R.T
