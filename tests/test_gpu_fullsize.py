"""GPU, BASELINE.json's full sizes (N = 1,000,000 codes, D=128, M=32, Ks=256, batch = 1024): size-independent
properties of the hot path, plus an oracle spot-check on a handful of queries (the oracle needs ~20 ms per query at
this size).  Everything goes through the C ABI."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import make_problem, assert_same_result

pytestmark = pytest.mark.gpu
N, M, Ks, Ds, B = 1_000_000, 32, 256, 4, 1024


@pytest.fixture(scope="module")
def world():
    from rii_amd import RiiGpu
    cw, codes, qs = make_problem(2024, M, Ks, Ds, N, "sift", dup=5000)
    rng = np.random.default_rng(5)
    Q = np.round(rng.random((B, M * Ds)) * 255).astype(np.float32)
    Q[:16] = qs
    g = RiiGpu(cw, False, simd_arch="avx512")
    g.add_codes(codes[:400_000], False)           # two appends: exercises the device buffer growth path
    g.add_codes(codes[400_000:], False)
    return g, cw, codes, Q


def test_full_filter_rerank_equals_exhaustive_scan(world):
    g, cw, codes, Q = world
    g.set_option("scan_mode", 1)
    i1, d1 = g.query_linear_batch(Q, 1, None)
    again = g.query_linear_batch(Q, 1, None)
    g.set_option("scan_mode", 0)
    i0, d0 = g.query_linear_batch(Q, 1, None)
    g.set_option("scan_mode", 1)
    assert np.array_equal(i1, i0) and np.array_equal(d1.view(np.uint32), d0.view(np.uint32))
    assert np.array_equal(again[0], i1) and np.array_equal(again[1], d1)          # idempotent


SAMPLE = list(range(16)) + [100, 333, 512, 777, 1000, 1023]


def test_full_oracle_spot_check(world):
    """The DEFAULT path of a 1024-query batch (tile tables -> fscan_mx_kernel -> re-rank: B >= fast_min_batch) compared with the
    oracle directly at N = 1M on a sample of rows; a 6-query batch (exact scan_kernel path) as well."""
    g, cw, codes, Q = world
    assert B >= g.get_option("fast_min_batch") and g.get_option("scan_mode") == 1 and g.get_option("scan_mx") == 1
    o = O.OracleRii(cw, False, simd_arch="avx512")
    o.add_codes(codes, False)
    E = np.array([], np.int64)
    idb, db = g.query_linear_batch(Q, 1, None)                  # the filter path (what bench.py times)
    for b in SAMPLE:
        assert_same_result((idb[b], db[b]), o.query_linear(Q[b], 1, E), "N=1M top-1, B=1024 batch, b=%d" % b)
    ids, d = g.query_linear_batch(Q[:6], 1, None)               # B < fast_min_batch: the exhaustive scan
    ids10, d10 = g.query_linear_batch(Q[:6], 10, None)
    for b in range(6):
        assert_same_result((ids[b], d[b]), o.query_linear(Q[b], 1, E), "N=1M top-1 b=%d" % b)
        wi, wd = o.query_linear(Q[b], 10, E)
        assert np.array_equal(np.asarray(wd, np.float32).view(np.uint32), d10[b].view(np.uint32))


def test_full_batch_equals_the_real_reference_on_every_row(world, reference):
    """All 1024 rows of the default-path batch against the compiled reference's query_linear (src/rii.h:195-242) when oracle/_ref
    loads on this host (~1.5 ms per query with OpenMP)."""
    g, cw, codes, Q = world
    ref, arch, flav = reference           # Ds = 4: fvec_L2sqr is the same arithmetic in all three build flavours
    r = ref.RiiCpp(cw, False)
    r.add_codes(codes, False)
    E = np.array([], np.int64)
    idb, db = g.query_linear_batch(Q, 1, None)
    for b in range(B):
        wi, wd = r.query_linear(Q[b], 1, E)
        assert int(idb[b, 0]) == int(wi[0]) and np.float32(wd[0]).view(np.uint32) == db[b, 0].view(np.uint32), "row %d" % b


def test_full_shard_minimum_is_global_minimum(world):
    """min over disjoint target-id shards of the per-shard top-1 == the global top-1 (what database sharding relies on)."""
    g, cw, codes, Q = world
    gi, gd = g.query_linear_batch(Q, 1, None)
    best_d = np.full(B, np.inf, np.float32)
    best_i = np.full(B, -1, np.int64)
    for s in range(4):
        tids = np.arange(s * N // 4, (s + 1) * N // 4, dtype=np.int64)
        si, sd = g.query_linear_batch(Q, 1, tids)
        assert ((si >= tids[0]) & (si <= tids[-1])).all()
        upd = sd[:, 0] < best_d                      # ascending shards + strict '<' == (dist, id) order
        best_d = np.where(upd, sd[:, 0], best_d)
        best_i = np.where(upd, si[:, 0], best_i)
    assert np.array_equal(best_i, gi[:, 0]) and np.array_equal(best_d.view(np.uint32), gd[:, 0].view(np.uint32))


def test_full_topk_vs_oracle_and_consistency(world):
    """Linear top-10 / top-100 at N = 1M: a sample of queries against the oracle (ids AND order, exact ties included -- the
    database holds 5000 duplicated codes), and size-independent properties over the whole batch."""
    g, cw, codes, Q = world
    i1, d1 = g.query_linear_batch(Q, 1, None)
    i10, d10 = g.query_linear_batch(Q, 10, None)
    i100, d100 = g.query_linear_batch(Q[:128], 100, None)
    o = O.OracleRii(cw, False, simd_arch="avx512")
    o.add_codes(codes, False)
    E = np.array([], np.int64)
    for b in list(range(12)) + [77, 500, 1023]:
        assert_same_result((i10[b], d10[b]), o.query_linear(Q[b], 10, E), "N=1M top-10 b=%d" % b)
    for b in (0, 5, 100, 127):
        assert_same_result((i100[b], d100[b]), o.query_linear(Q[b], 100, E), "N=1M top-100 b=%d" % b)
    assert np.array_equal(d10[:, 0].view(np.uint32), d1[:, 0].view(np.uint32))
    assert (np.diff(d10, axis=1) >= 0).all() and (np.diff(d100, axis=1) >= 0).all()
    assert np.array_equal(d100[:, :10], d10[:128])
    untied = (np.diff(d100[:, :11], axis=1) > 0).all(axis=1)        # rows whose 11 smallest distances are all different
    assert untied.sum() > 64
    assert np.array_equal(i100[untied, :10], i10[:128][untied])
    assert np.array_equal(i10[:128][untied, 0], i1[:128][untied, 0])
    assert all(len(set(r)) == 100 for r in i100)


def test_full_ivf_with_L_equal_N_is_the_linear_scan(world):
    """tests/test_rii.py:178-181 at full size: query_ivf(L=N, target=all) == query_linear."""
    g, cw, codes, Q = world
    g.reconfigure(1024, 2)
    assert g.nlist == 1024 and sum(len(p) for p in g.posting_lists) == N
    li, ld = g.query_linear_batch(Q[:32], 1, None)
    ii, idd, cnt = g.query_ivf_batch(Q[:32], 1, np.arange(N, dtype=np.int64), N)
    assert (cnt == 1).all()
    assert np.array_equal(idd.view(np.uint32), ld.view(np.uint32))
    # ids may differ only where the minimum distance is attained by several codes (ivf walks lists, linear walks ids):
    # then both ids must carry the very same code
    for b in np.nonzero(ii[:, 0] != li[:, 0])[0]:
        assert np.array_equal(codes[ii[b, 0]], codes[li[b, 0]])
    # the bench configuration itself: L = L0, top-1, all 1024 queries answered, results inside the visited lists
    L0 = int(np.round(N / 1024))
    bi, bd, bc = g.query_ivf_batch(Q, 1, None, L0)
    assert (bc == 1).all() and (bd[:, 0] >= ld.min() * 0).all()
    lin_i, lin_d = g.query_linear_batch(Q, 1, None)
    assert (bd[:, 0] >= lin_d[:, 0]).all()           # an inverted-index answer can never beat the exhaustive one


@pytest.fixture(scope="module")
def ivf_world(world):
    """BASELINE configs 3 and 4 at their full size: reconfigure(1024, 5) on the GPU and in the oracle (PQk-means fit on
    100k sampled codes + coarse assignment of all 1M codes: src/rii.h:108-156)."""
    g, cw, codes, Q = world
    g.reconfigure(1024, 5)
    o = O.OracleRii(cw, False, simd_arch="avx512")
    o.add_codes(codes, False)
    o.reconfigure(1024, 5)
    return g, o, Q


def test_full_reconfigure_equals_oracle(ivf_world):
    g, o, Q = ivf_world
    assert g.coarse_centers == o.coarse_centers
    assert g.posting_lists == o.posting_lists


def test_full_config3_ivf_known_answers(ivf_world):
    """Config 3 (N=1M, nlist=1024, L=L0=977, batch 1024) against the oracle: strict ids, distances and counts."""
    g, o, Q = ivf_world
    E = np.array([], np.int64)
    L0 = int(np.round(N / 1024))
    for topk in (1, 10):
        ids, d, cnt = g.query_ivf_batch(Q, topk, None, L0)
        for b in SAMPLE:
            n = int(cnt[b])
            assert_same_result((ids[b, :n], d[b, :n]), o.query_ivf(Q[b], topk, E, L0), "config 3 k=%d b=%d" % (topk, b))
        assert (cnt == topk).all()


def test_full_config4_subset_known_answers(ivf_world):
    """Config 4 (sorted random |S| = 100k target ids shared by the batch): linear and inverted-index search (w = 13)
    against the oracle, strict."""
    g, o, Q = ivf_world
    rng = np.random.default_rng(1234)
    S = np.sort(rng.choice(N, 100_000, replace=False)).astype(np.int64)
    L0 = int(np.round(N / 1024))
    for topk in (1, 10):
        li, ld = g.query_linear_batch(Q, topk, S)
        ii, idd, cnt = g.query_ivf_batch(Q, topk, S, L0)
        assert np.isin(li, S).all()
        for b in SAMPLE:
            assert_same_result((li[b], ld[b]), o.query_linear(Q[b], topk, S), "config 4 linear k=%d b=%d" % (topk, b))
            n = int(cnt[b])
            assert_same_result((ii[b, :n], idd[b, :n]), o.query_ivf(Q[b], topk, S, L0), "config 4 ivf k=%d b=%d" % (topk, b))
            assert np.isin(ii[b, :n], S).all()


def test_large_index_beyond_2_pow_24_codes():
    """20M codes (M=16): code indices above 2^24 (fp32-exact integer range) and several chunks per tile; the filter path
    must agree with the exhaustive scan, and top-k with the full-sort path."""
    from rii_amd import RiiGpu
    Nbig, Mb = 20_000_000, 16
    cw, _, qs = make_problem(77, Mb, 256, 6, 8, "unit")
    rng = np.random.default_rng(123)
    codes = rng.integers(0, 256, size=(Nbig, Mb), dtype=np.uint8)
    plant = rng.integers(1 << 24, Nbig, size=8)
    Q = rng.random((160, Mb * 6)).astype(np.float32)
    for j, pos in enumerate(plant):                      # make query j's nearest code sit at a large index
        codes[pos] = np.argmin(((cw - Q[j].reshape(Mb, 1, 6)) ** 2).sum(-1), axis=1)
    g = RiiGpu(cw, False)
    g.add_codes(codes, False)
    g.set_option("scan_mode", 1)
    i1, d1 = g.query_linear_batch(Q, 1, None)
    i5, d5 = g.query_linear_batch(Q[:16], 5, None)
    g.set_option("scan_mode", 0)
    i0, d0 = g.query_linear_batch(Q, 1, None)
    j5, e5 = g.query_linear_batch(Q[:16], 5, None)
    assert np.array_equal(i1, i0) and np.array_equal(d1.view(np.uint32), d0.view(np.uint32))
    assert np.array_equal(i5, j5) and np.array_equal(d5.view(np.uint32), e5.view(np.uint32))
    assert (i1[:8, 0] == plant).all()
    assert (i1 >= (1 << 24)).any()


def test_deep_shard_64m_codes_filter_equals_exhaustive_and_shards_compose():
    """BASELINE configs[4] per-GPU shape at a size that is a true HBM stream (64 M codes x M = 16 = 1 GB, four times the Infinity
    Cache; the 125 M-code shard of Deep1B over 8 GPUs is the same code path, twice as long): size-independent properties of the
    default path -- (1) filter + re-rank (fscan_mx_dual_kernel) == exhaustive fp32 scan on every row, (2) idempotence, (3) the
    minimum over the whole shard == the minimum of the minima of two half-shard target ranges (what the database-sharded merge
    relies on), (4) a few-query call (exact scan path) agrees with the batch's rows.  Random bytes as codes (throughput shape)."""
    from rii_amd import RiiGpu
    n, m, ds = 64_000_000, 16, 6
    rng = np.random.default_rng(77)
    cw = rng.random((m, 256, ds)).astype(np.float32)
    g = RiiGpu(cw, False, simd_arch="avx512")
    step = 16_000_000
    for s in range(0, n, step):
        g.add_codes(rng.integers(0, 256, size=(step, m), dtype=np.uint8), False)
    assert g.N == n
    Q = rng.random((64, m * ds)).astype(np.float32)
    i1, d1 = g.query_linear_batch(Q, 1, None)
    again = g.query_linear_batch(Q, 1, None)
    assert np.array_equal(again[0], i1) and np.array_equal(again[1], d1)
    g.set_option("scan_mode", 0)
    i0, d0 = g.query_linear_batch(Q, 1, None)
    g.set_option("scan_mode", 1)
    assert np.array_equal(i1, i0) and np.array_equal(d1.view(np.uint32), d0.view(np.uint32))
    few = g.query_linear_batch(Q[:3], 1, None)
    assert np.array_equal(few[0], i1[:3]) and np.array_equal(few[1], d1[:3])
    half = 4_000_000                                   # two adjacent target ranges around the winners of the first queries
    lo = int(max(0, min(int(i1[0, 0]), n - 2 * half)))
    ta, tb = np.arange(lo, lo + half, dtype=np.int64), np.arange(lo + half, lo + 2 * half, dtype=np.int64)
    (ia, da), (ib, db) = g.query_linear_batch(Q[:8], 1, ta), g.query_linear_batch(Q[:8], 1, tb)
    both = g.query_linear_batch(Q[:8], 1, np.concatenate([ta, tb]))
    pick = np.where((da < db) | ((da == db) & (ia < ib)), ia, ib)
    assert np.array_equal(pick, both[0]) and np.array_equal(np.minimum(da, db), both[1])
    assert int(both[0][0, 0]) == int(i1[0, 0])         # (the range was chosen around query 0's winner)
