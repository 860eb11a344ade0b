"""GPU: seeded differential fuzzing of the whole C-ABI surface against the CPU oracle over random shapes
(M, Ks, Ds, N, duplicates, nlist, iterations, topk, L, target sets).  Bit-exact; linear topk>1 under the tie contract."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import make_problem, assert_same_result

pytestmark = pytest.mark.gpu
E = np.array([], np.int64)


@pytest.mark.parametrize("seed", range(48))
def test_fuzz_against_oracle(seed):
    from rii_amd import RiiGpu
    rng = np.random.default_rng(1000 + seed)
    M = int(rng.integers(1, 41))
    Ks = int(rng.choice([2, 3, 16, 20, 64, 100, 255, 256]))
    Ds = int(rng.integers(1, 10))
    N = int(rng.choice([1, 2, 7, 63, 64, 65, 500, 1025, 3000, 6000]))
    arch = str(rng.choice(["avx512", "avx", "sse"]))
    scale = str(rng.choice(["unit", "sift"]))
    dup = int(N * rng.choice([0.0, 0.0, 0.2, 0.9]))
    cw, codes, qs = make_problem(seed, M, Ks, Ds, N, scale, dup=dup)
    g = RiiGpu(cw, False, simd_arch=arch)
    o = O.OracleRii(cw, False, simd_arch=arch)
    n1 = int(rng.integers(0, N + 1))
    for part in (codes[:n1], codes[n1:]):
        if len(part):
            g.add_codes(part, False); o.add_codes(part, False)
    g.set_option("fast_min_batch", int(rng.choice([0, 128])))
    Q = qs[:5]
    all_codes = [codes]

    def true_dist(q):
        c = np.concatenate(all_codes, 0)
        d = O.dtable(cw, q, arch)
        acc = np.zeros(c.shape[0], np.float32)
        for m in range(M):
            acc = (acc + d[m, c[:, m]]).astype(np.float32)
        return acc

    # --- linear ---
    for _ in range(3):
        S = int(rng.integers(0, N + 1)) if rng.random() < 0.6 else 0
        tids = np.sort(rng.choice(N, S, replace=False)).astype(np.int64) if S else E
        pool = S if S else N
        topk = int(rng.integers(1, min(pool, 60) + 1))
        ids, d = g.query_linear_batch(Q, topk, tids)
        for b in range(len(Q)):
            want = o.query_linear(Q[b], topk, tids)
            assert_same_result((ids[b], d[b]), want, "lin k=%d S=%d seed=%d b=%d" % (topk, S, seed, b))
    # --- inverted index ---
    nlist = int(rng.integers(1, min(N, 200) + 1))
    it = int(rng.integers(0, 4))
    g.reconfigure(nlist, it); o.reconfigure(nlist, it)
    assert g.coarse_centers == o.coarse_centers, "centres seed=%d" % seed
    assert g.posting_lists == o.posting_lists, "lists seed=%d" % seed
    if rng.random() < 0.5:                                    # append more codes, with or without list update
        extra = make_problem(seed + 7, M, Ks, Ds, 50, scale)[1]
        upd = bool(rng.random() < 0.5)
        g.add_codes(extra, upd); o.add_codes(extra, upd)
        all_codes.append(extra)
        assert g.posting_lists == o.posting_lists
    Nn = g.N
    for _ in range(4):
        S = int(rng.integers(1, Nn + 1)) if rng.random() < 0.5 else 0
        tids = np.sort(rng.choice(Nn, S, replace=False)).astype(np.int64) if S else E
        pool = S if S else Nn
        topk = int(rng.integers(1, min(pool, 40) + 1))
        L = int(rng.integers(topk, Nn + 1))
        g.set_option("ivf_fused", int(rng.random() < 0.8))
        ids, d, cnt = g.query_ivf_batch(Q, topk, tids, L)
        for b in range(len(Q)):
            want = o.query_ivf(Q[b], topk, tids, L)
            n = int(cnt[b])
            assert_same_result((ids[b, :n], d[b, :n]), want, "ivf seed=%d k=%d L=%d S=%d b=%d" % (seed, topk, L, S, b))


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_large_db_paths_agree(seed):
    """Databases big enough for the scan-order copy (N >= 65536), random shapes: the filter over the scan order, the filter in
    id order, the exhaustive scan over the scan order and the plain exhaustive scan must return identical top-1 results
    (ids and distance bits) and identical top-k distance rows; checked again after an append."""
    from rii_amd import RiiGpu
    rng = np.random.default_rng(5000 + seed)
    M = int(rng.choice([3, 8, 12, 16, 20, 32, 36, 40, 64]))
    Ks = int(rng.choice([256, 256, 256, 64, 100]))
    Ds = int(rng.choice([1, 2, 4, 6]))
    N = int(rng.integers(65536, 140000))
    B = int(rng.choice([5, 40, 130, 300]))
    cw = rng.random((M, Ks, Ds)).astype(np.float32)
    if rng.random() < 0.4:
        cw = np.round(cw * 255)                      # integer-valued codebooks: exact ties are common
    codes = rng.integers(0, Ks, size=(N, M), dtype=np.uint8)
    ndup = int(N * rng.choice([0.0, 0.01, 0.3]))
    if ndup:
        codes[rng.integers(0, N, ndup)] = codes[rng.integers(0, N, ndup)]
    Q = rng.random((B, M * Ds)).astype(np.float32) * (255 if cw.max() > 2 else 1)
    g = RiiGpu(cw, False, simd_arch="avx512")
    n1 = int(rng.integers(N // 2, N))
    g.add_codes(codes[:n1], False)

    def run(mode, order, topk):
        g.set_option("scan_mode", mode)
        g.set_option("scan_order", order)
        g.set_option("fast_min_batch", int(rng.choice([0, 33])))
        return g.query_linear_batch(Q, topk, None)

    for stage in range(2):
        ref = run(0, 0, 1)
        for mode, order in ((1, 1), (1, 0), (0, 1)):
            got = run(mode, order, 1)
            assert np.array_equal(got[0], ref[0]), "ids seed=%d mode=%d order=%d stage=%d" % (seed, mode, order, stage)
            assert np.array_equal(got[1].view(np.uint32), ref[1].view(np.uint32))
        k = int(rng.integers(2, 50))
        refk = run(0, 0, k)
        for mode, order in ((1, 1), (1, 0)):
            gotk = run(mode, order, k)
            assert np.array_equal(gotk[1].view(np.uint32), refk[1].view(np.uint32)), "topk dists seed=%d" % seed
            assert np.array_equal(gotk[0], refk[0]), "topk ids seed=%d" % seed       # both sides use the canonical (dist, id) order
        if stage == 0:
            g.add_codes(codes[n1:], False)


@pytest.mark.parametrize("seed", range(14))
def test_fuzz_matrix_core_filter_against_oracle(seed):
    """The shapes whose filter scan sums its table bytes on the matrix cores (M = 16 / 32 / 64 at Ks = 256, fscan_mx_kernel),
    forced onto that path at any batch size (fast_min_batch = 0), against the CPU oracle: random Ds and SIMD flavour, sizes
    that end inside a 16-code group or a 1024-code trip, heavy duplication (exact ties: the reference's std::partial_sort
    order), appends, subsets, top-1 and top-k; bit-exact ids and distances."""
    from rii_amd import RiiGpu
    rng = np.random.default_rng(8800 + seed)
    M = int(rng.choice([16, 32, 64]))
    Ks = 256
    Ds = int(rng.choice([1, 2, 3, 4, 8]))
    N = int(rng.choice([15, 16, 17, 1023, 1024, 1040, 2500, 4097, 7000]))
    arch = str(rng.choice(["avx512", "avx", "sse"]))
    scale = str(rng.choice(["unit", "sift"]))
    dup = int(N * rng.choice([0.0, 0.2, 0.9]))
    cw, codes, qs = make_problem(300 + seed, M, Ks, Ds, N, scale, dup=dup)
    g = RiiGpu(cw, False, simd_arch=arch)
    o = O.OracleRii(cw, False, simd_arch=arch)
    g.set_option("fast_min_batch", 0)
    assert g.get_option("scan_mx") == 1
    n1 = int(rng.integers(1, N + 1))
    done = 0
    for part in (codes[:n1], codes[n1:]):
        if not len(part):
            continue
        g.add_codes(part, False); o.add_codes(part, False)
        done += len(part)
        Q = qs[:int(rng.choice([1, 2, 3, 5, 6, 16]))]          # (the oracle takes ~60 ms per query: few queries, many shapes)
        for _ in range(3):
            S = int(rng.integers(1, done + 1)) if rng.random() < 0.5 else 0
            tids = np.sort(rng.choice(done, S, replace=False)).astype(np.int64) if S else E
            pool = S if S else done
            topk = 1 if rng.random() < 0.4 else int(rng.integers(1, min(pool, 70) + 1))
            ids, d = g.query_linear_batch(Q, topk, tids)
            for b in range(len(Q)):
                want = o.query_linear(Q[b], topk, tids)
                assert_same_result((ids[b], d[b]), want, "mx M=%d Ds=%d N=%d k=%d S=%d seed=%d b=%d" % (M, Ds, done, topk, S, seed, b))
