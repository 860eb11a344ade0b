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


def test_full_oracle_spot_check(world):
    g, cw, codes, Q = world
    o = O.OracleRii(cw, False, simd_arch="avx512")
    o.add_codes(codes, False)
    ids, d = g.query_linear_batch(Q[:6], 1, None)
    ids10, d10 = g.query_linear_batch(Q[:6], 10, None)
    E = np.array([], np.int64)
    for b in range(6):
        assert_same_result((ids[b], d[b]), o.query_linear(Q[b], 1, E), "N=1M top-1 b=%d" % b)
        wi, wd = o.query_linear(Q[b], 10, E)
        assert np.array_equal(np.asarray(wd, np.float32).view(np.uint32), d10[b].view(np.uint32))


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


def test_full_topk_consistency(world):
    g, cw, codes, Q = world
    i1, d1 = g.query_linear_batch(Q, 1, None)
    i10, d10 = g.query_linear_batch(Q, 10, None)
    i100, d100 = g.query_linear_batch(Q[:128], 100, None)
    assert np.array_equal(i10[:, 0], i1[:, 0]) and np.array_equal(d10[:, 0].view(np.uint32), d1[:, 0].view(np.uint32))
    assert (np.diff(d10, axis=1) >= 0).all() and (np.diff(d100, axis=1) >= 0).all()
    assert np.array_equal(i100[:, :10], i10[:128]) and np.array_equal(d100[:, :10], d10[:128])
    assert all(len(set(r)) == 100 for r in i100)
    # every returned id carries its true distance: recompute a sample exactly on the host
    o = O.OracleRii(cw, False, simd_arch="avx512")
    for b in (0, 77):
        dt = O.dtable(cw, Q[b], "avx512")
        for j in (0, 5, 9):
            assert np.float32(O.lib().oracle_adist(dt.ctypes.data_as(O.ctypes.POINTER(O.ctypes.c_float)), M, Ks,
                                                   np.ascontiguousarray(codes[i10[b, j]]).ctypes.data_as(
                                                       O.ctypes.POINTER(O.ctypes.c_uint8)))) == d10[b, j]


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
