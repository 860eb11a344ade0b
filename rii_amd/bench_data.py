"""Measurement inputs for bench.py / the full-size tests: a seeded SIFT1M-shaped synthetic set (the real
SIFT1M cannot be downloaded here), readers for the real .fvecs/.ivecs files when $SIFT1M_DIR provides them
(format as read by the reference's examples/benchmark/util.py:5-32: int32 dimension header per row),
torch-accelerated PQ training/encoding (codec training is upstream of the hot path) and recall@R
(examples/benchmark/util.py:35-58)."""
import os

import numpy as np


def read_fvecs(path, count=None):
    a = np.fromfile(path, dtype=np.int32, count=-1 if count is None else count * 129)
    d = int(a[0])
    return a.reshape(-1, d + 1)[:, 1:].copy().view(np.float32)


def read_ivecs(path):
    a = np.fromfile(path, dtype=np.int32)
    d = int(a[0])
    return a.reshape(-1, d + 1)[:, 1:].copy()


def sift_like(n_base=1_000_000, n_train=100_000, n_query=10_000, D=128, seed=1234, n_clusters=4096):
    """Non-negative integer-valued clustered vectors like SIFT: cluster means U[0,128)^D, N(0,24^2) noise,
    clipped to [0,255] and rounded.  Returns (base, train, query) float32."""
    sift_dir = os.environ.get("SIFT1M_DIR")
    if sift_dir and os.path.exists(os.path.join(sift_dir, "sift_base.fvecs")) and D == 128:
        base = read_fvecs(os.path.join(sift_dir, "sift_base.fvecs"))[:n_base]
        train = read_fvecs(os.path.join(sift_dir, "sift_learn.fvecs"))[:n_train]
        query = read_fvecs(os.path.join(sift_dir, "sift_query.fvecs"))[:n_query]
        return base, train, query
    rng = np.random.default_rng(seed)
    means = (rng.random((n_clusters, D), dtype=np.float32) * 128.0)

    def draw(n):
        out = np.empty((n, D), np.float32)
        step = 131072
        for s in range(0, n, step):
            m = min(step, n - s)
            c = rng.integers(0, n_clusters, m)
            x = means[c] + rng.standard_normal((m, D), dtype=np.float32) * 24.0
            out[s:s + m] = np.rint(np.clip(x, 0.0, 255.0))
        return out
    return draw(n_base), draw(n_train), draw(n_query)


def more_queries(n, D=128, seed=1234, n_clusters=4096, stream=1):
    """n more query vectors from the distribution sift_like(seed=seed) draws from (same cluster means, fresh noise)."""
    means = (np.random.default_rng(seed).random((n_clusters, D), dtype=np.float32) * 128.0)
    rng = np.random.default_rng([seed, stream])
    c = rng.integers(0, n_clusters, n)
    x = means[c] + rng.standard_normal((n, D), dtype=np.float32) * 24.0
    return np.rint(np.clip(x, 0.0, 255.0)).astype(np.float32)


def train_pq(train, M, Ks=256, iters=10, seed=123, device=None):
    """k-means PQ codebooks (M, Ks, Ds) float32.  torch when available (GPU if `device` says so)."""
    import torch
    dev = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
    x = torch.from_numpy(np.ascontiguousarray(train)).to(dev)
    N, D = x.shape
    Ds = D // M
    g = torch.Generator(device="cpu").manual_seed(seed)
    cw = torch.empty((M, Ks, Ds), dtype=torch.float32, device=dev)
    for m in range(M):
        sub = x[:, m * Ds:(m + 1) * Ds].contiguous()
        cent = sub[torch.randperm(N, generator=g)[:Ks].to(dev)].clone()
        for _ in range(iters):
            lab = torch.cdist(sub, cent).argmin(1)
            sums = torch.zeros_like(cent).index_add_(0, lab, sub)
            cnt = torch.zeros(Ks, device=dev).index_add_(0, lab, torch.ones(N, device=dev))
            nz = cnt > 0
            cent[nz] = sums[nz] / cnt[nz, None]
        cw[m] = cent
    return cw.cpu().numpy()


def encode_pq(vecs, codewords, device=None, chunk=262144):
    import torch
    dev = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
    M, Ks, Ds = codewords.shape
    cw = torch.from_numpy(codewords).to(dev)
    codes = np.empty((vecs.shape[0], M), np.uint8)
    for s in range(0, vecs.shape[0], chunk):
        x = torch.from_numpy(np.ascontiguousarray(vecs[s:s + chunk])).to(dev)
        out = torch.empty((x.shape[0], M), dtype=torch.uint8, device=dev)
        for m in range(M):
            out[:, m] = torch.cdist(x[:, m * Ds:(m + 1) * Ds].contiguous(), cw[m]).argmin(1).to(torch.uint8)
        codes[s:s + chunk] = out.cpu().numpy()
    return codes


def exact_nn(base, query, device=None, chunk=131072):
    """Ground truth: exact fp32 brute-force nearest neighbour id per query."""
    import torch
    dev = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
    q = torch.from_numpy(np.ascontiguousarray(query)).to(dev)
    best_d = torch.full((q.shape[0],), float("inf"), device=dev)
    best_i = torch.zeros((q.shape[0],), dtype=torch.int64, device=dev)
    for s in range(0, base.shape[0], chunk):
        b = torch.from_numpy(np.ascontiguousarray(base[s:s + chunk])).to(dev)
        d = torch.cdist(q, b)
        dm, im = d.min(1)
        upd = dm < best_d
        best_d = torch.where(upd, dm, best_d)
        best_i = torch.where(upd, im + s, best_i)
    return best_i.cpu().numpy()


def recall_at_r(I, gt, r=1):
    """examples/benchmark/util.py:35-58: fraction of queries whose true NN is within the first r results."""
    I = np.asarray(I)
    gt = np.asarray(gt).reshape(-1, 1)
    return float((I[:, :r] == gt[:, :1]).any(axis=1).mean())


# ---- Deep1B-shaped structured data, generated and kept ON the device (round 6: bench.py's `deep_structured` leg) ----
def deep_like_torch(n, D=96, seed=77, n_clusters=16384, sigma=0.8, decay=0.7, device=None, stream=0, chunk=1 << 20):
    """Unit-norm clustered vectors shaped like Deep1B's (PCA-compressed, L2-normalised CNN descriptors, D = 96): coordinate k carries
    the weight (1 + k)^-decay of a PCA spectrum, cluster means and N(0, sigma^2) within-cluster noise are drawn in the unweighted
    space, the weighted sum is re-normalised.  Seeded; generated chunk by chunk on `device`; -> device tensor [n, D].
    `stream` != 0 draws fresh vectors around the SAME means (queries / training sets)."""
    import torch
    dev = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
    gm = torch.Generator(device="cpu").manual_seed(seed)
    means = torch.randn((n_clusters, D), generator=gm).to(dev)
    wgt = (1.0 + torch.arange(D, dtype=torch.float32, device=dev)) ** (-decay)
    g = torch.Generator(device="cpu").manual_seed(seed * 1000003 + stream)
    out = torch.empty((n, D), dtype=torch.float32, device=dev)
    for s0 in range(0, n, chunk):
        m = min(chunk, n - s0)
        c = torch.randint(0, n_clusters, (m,), generator=g).to(dev)
        x = (means[c] + torch.randn((m, D), generator=g).to(dev) * sigma) * wgt
        out[s0:s0 + m] = x / x.norm(dim=1, keepdim=True)
    return out


def encode_pq_torch(x, codewords, chunk=1 << 20):
    """encode_pq for a device tensor: -> uint8 device tensor [n, M]."""
    import torch
    M, Ks, Ds = codewords.shape
    cw = torch.from_numpy(codewords).to(x.device)
    out = torch.empty((x.shape[0], M), dtype=torch.uint8, device=x.device)
    for s0 in range(0, x.shape[0], chunk):
        xs = x[s0:s0 + chunk]
        for m in range(M):
            out[s0:s0 + chunk, m] = torch.cdist(xs[:, m * Ds:(m + 1) * Ds].contiguous(), cw[m]).argmin(1).to(torch.uint8)
    return out


def exact_nn_torch(base, query, chunk=1 << 18):
    """exact_nn for device tensors (squared distances through one matrix product per chunk)."""
    import torch
    qn = (query * query).sum(1, keepdim=True)
    best_d = torch.full((query.shape[0],), float("inf"), device=query.device)
    best_i = torch.zeros((query.shape[0],), dtype=torch.int64, device=query.device)
    for s0 in range(0, base.shape[0], chunk):
        b = base[s0:s0 + chunk]
        d = qn + (b * b).sum(1)[None, :] - 2.0 * (query @ b.T)
        dm, im = d.min(1)
        upd = dm < best_d
        best_d = torch.where(upd, dm, best_d)
        best_i = torch.where(upd, im + s0, best_i)
    return best_i.cpu().numpy()
