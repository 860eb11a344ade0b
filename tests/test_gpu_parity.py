"""GPU parity tests proper: the HIP engine, called through the C ABI (rii_amd.core.RiiGpu -> librii_amd.so),
against (1) the golden vectors recorded from the real reference and (2) the CPU oracle on seeded inputs.
Bit-exact on ids and distances, exactly tied distances included (std::partial_sort's heap order is replayed on the GPU).
"""
import numpy as np
import pytest

from oracle import oracle as O
from tests.replay import CASE_NAMES, NEARTIE_DS, replay_case, replay_neartie, replay_stale, replay_state
from tests.util import make_problem, assert_same_result

pytestmark = pytest.mark.gpu
E = np.array([], np.int64)


def gpu_engine(arch):
    from rii_amd import RiiGpu
    return lambda cw: RiiGpu(cw, False, simd_arch=arch)


def true_dist_fn_for(cw, codes, arch):
    def fn(q):
        dt = O.dtable(cw, q, arch)
        M = cw.shape[0]
        acc = np.zeros(codes.shape[0], np.float32)
        for m in range(M):                       # sequential fp32 sum over m, like ADist
            acc = (acc + dt[m, codes[:, m]]).astype(np.float32)
        return acc
    return fn


@pytest.mark.parametrize("arch", ["avx512", "avx"])
@pytest.mark.parametrize("name", CASE_NAMES)
def test_gpu_replays_golden(name, arch):
    n = replay_case(gpu_engine(arch), name, arch)
    assert n > 50


@pytest.mark.parametrize("arch", ["avx512", "avx"])
@pytest.mark.parametrize("Ds", NEARTIE_DS)
def test_gpu_assignment_neartie_golden(Ds, arch):
    replay_neartie(gpu_engine(arch), Ds, arch)


@pytest.mark.parametrize("arch", ["avx512", "avx", "sse"])
@pytest.mark.parametrize("Ds", [1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 15, 16, 17, 24, 31, 32, 33, 40])
def test_gpu_lut_bitexact(Ds, arch):
    from rii_amd import RiiGpu
    cw, _, qs = make_problem(Ds, 3, 256, Ds, 8, "unit")
    e = RiiGpu(cw, False, simd_arch=arch)
    got = e.dtable(qs[:5])
    for b in range(5):
        want = O.dtable(cw, qs[b], arch)
        assert np.array_equal(got[b].view(np.uint32), want.view(np.uint32)), "Ds=%d arch=%s q=%d" % (Ds, arch, b)


SHAPES = [(32, 256, 4, 20000, "sift"), (16, 256, 6, 20000, "unit"), (8, 256, 16, 5000, "unit"),
          (64, 256, 2, 5000, "unit"), (4, 20, 10, 1000, "unit"), (3, 7, 37, 500, "unit"), (20, 256, 2, 1000, "unit")]


@pytest.mark.parametrize("shape", SHAPES)
def test_gpu_vs_oracle_linear_ivf_batch(shape):
    """Seeded problems at sizes the oracle finishes in seconds: batched GPU results == per-query oracle."""
    from rii_amd import RiiGpu
    M, Ks, Ds, N, scale = shape
    arch = "avx512"
    cw, codes, qs = make_problem(42, M, Ks, Ds, N, scale, dup=N // 10)
    g = RiiGpu(cw, False, simd_arch=arch)
    o = O.OracleRii(cw, False, simd_arch=arch)
    g.add_codes(codes, False); o.add_codes(codes, False)
    td = true_dist_fn_for(cw, codes, arch)
    rng = np.random.default_rng(9)
    sub = np.sort(rng.choice(N, N // 7, replace=False)).astype(np.int64)
    Q = qs[:11]                                         # ragged vs the 4-query table tile on purpose
    for topk in (1, 7, 64):
        for tids in (E, sub):
            ids, d = g.query_linear_batch(Q, topk, tids)
            for b in range(Q.shape[0]):
                want = o.query_linear(Q[b], topk, tids)
                assert_same_result((ids[b], d[b]), want, "linear k=%d S=%d b=%d" % (topk, len(tids), b))
    # top-1 again with the filter path forced (11 queries are below fast_min_batch: the leg above took the exhaustive scan)
    g.set_option("fast_min_batch", 0)
    for tids in (E, sub):
        ids, d = g.query_linear_batch(Q, 1, tids)
        for b in range(Q.shape[0]):
            assert_same_result((ids[b], d[b]), o.query_linear(Q[b], 1, tids), "filter top-1 S=%d b=%d" % (len(tids), b))
    g.set_option("fast_min_batch", 33)
    nlist = max(2, int(np.sqrt(N)) // 2)
    g.reconfigure(nlist, 3); o.reconfigure(nlist, 3)
    assert g.coarse_centers == o.coarse_centers
    assert g.posting_lists == o.posting_lists
    L0 = int(np.round(N / nlist))
    for topk, L in ((1, L0), (1, 5 * L0), (5, L0 + 5), (10, N), (3, 3)):
        for tids in (E, sub):
            if len(tids) and (topk > len(tids)):
                continue
            ids, d, cnt = g.query_ivf_batch(Q, topk, tids, L)
            for b in range(Q.shape[0]):
                want = o.query_ivf(Q[b], topk, tids, L)
                n = int(cnt[b])
                assert_same_result((ids[b, :n], d[b, :n]), want, "ivf k=%d L=%d S=%d b=%d" % (topk, L, len(tids), b))


@pytest.mark.parametrize("nlist", [5000, 20000, 3000])
def test_gpu_ivf_many_lists_and_large_L_vs_oracle(nlist):
    """Shapes past the LDS working sets of round 2 (nlist <= 4096, L <= 4096): the reference's default nlist = sqrt(N) is 11 k at
    a 125 M-code shard and 31.6 k at 1e9 codes (rii/rii.py:143), its SIFT1M benchmark runs L = 5000
    (examples/benchmark/ann_methods.py:19).  Fused kernel with the coarse scores in global scratch (w <= 32), the exact kernel
    with sequences in global scratch and heaps in LDS (flagged queries, and every query when w > 32), strict ids / distances /
    counts against the oracle on the same centres and lists; duplicated codes and duplicated centres force exact ties."""
    from rii_amd import RiiGpu
    arch = "avx512"
    N, M, Ks, Ds = 100_000, 8, 256, 4
    cw, codes, qs = make_problem(1234 + nlist, M, Ks, Ds, N, "sift", dup=N // 20)
    rng = np.random.default_rng(nlist)
    centers = np.ascontiguousarray(codes[rng.integers(0, N, nlist)])          # random codes as centres (some twice: tied coarse distances)
    g = RiiGpu(cw, False, simd_arch=arch)
    g.add_codes(codes, False)
    g.set_coarse_centers(centers)
    o = O.OracleRii(cw, False, simd_arch=arch)
    o.add_codes(codes, False)
    o.set_coarse_centers(centers)
    assert g.posting_lists == o.posting_lists
    sub = np.sort(rng.choice(N, N // 9, replace=False)).astype(np.int64)
    Q = qs[:10]
    L0 = max(1, int(np.round(N / nlist)))
    cases = [(1, L0, E), (1, 5000, E), (5, 5000, E), (50, 200, E), (3, 8192, E), (2, 40, sub), (1, 5000, sub), (10, 20000, E), (1, N, E)]
    n_tied = 0
    for topk, L, tids in cases:
        ids, d, cnt = g.query_ivf_batch(Q, topk, tids, L)
        for b in range(Q.shape[0]):
            wi, wd = o.query_ivf(Q[b], topk, tids, L)
            n = int(cnt[b])
            assert_same_result((ids[b, :n], d[b, :n]), (wi, wd), "nlist=%d k=%d L=%d S=%d b=%d" % (nlist, topk, L, len(tids), b))
            n_tied += int(len(set(np.asarray(wd).tolist())) < len(wd))
    assert n_tied > 0
    # the one-lane emulation kernels (option ivf_fused = 0) must agree too: they stay the fallback for w or topk above 1024
    g.set_option("ivf_fused", 0)
    ids, d, cnt = g.query_ivf_batch(Q[:3], 5, E, 300)
    g.set_option("ivf_fused", 1)
    for b in range(3):
        assert_same_result((ids[b, :int(cnt[b])], d[b, :int(cnt[b])]), o.query_ivf(Q[b], 5, E, 300), "emulation kernels b=%d" % b)


@pytest.mark.parametrize("M,Ks,Ds", [(160, 256, 1), (256, 256, 1), (200, 256, 2)])
def test_gpu_wide_tables_vs_oracle(M, Ks, Ds):
    """Shapes whose one-query table (M * Ks * 4 B = 160 .. 256 KiB) does not fit the LDS budget -- legal in the reference (any
    M <= D, Ks <= 256: rii/rii.py:35, src/rii.h:361-373), refused by round 2's engine.  Tables in global memory (widetab.hip):
    DTable, linear search for every topk incl. exact ties and target ids, reconfigure / coarse assignment, the inverted index."""
    from rii_amd import RiiGpu
    arch = "avx512"
    N = 3000
    cw, codes, qs = make_problem(900 + M, M, Ks, Ds, N, "sift", dup=400)
    g = RiiGpu(cw, False, simd_arch=arch)
    o = O.OracleRii(cw, False, simd_arch=arch)
    assert g.get_option("lut_tile") == 0                       # no table tile fits LDS
    g.add_codes(codes, False); o.add_codes(codes, False)
    Q = qs[:9]
    lut = g.dtable(Q[:3])
    for b in range(3):
        assert np.array_equal(lut[b].view(np.uint32), O.dtable(cw, Q[b], arch).view(np.uint32))
    rng = np.random.default_rng(M)
    sub = np.sort(rng.choice(N, N // 5, replace=False)).astype(np.int64)
    n_tied = 0
    for topk in (1, 6, 64, 1500):
        for tids in (E, sub):
            if len(tids) and topk > len(tids):
                continue
            ids, d = g.query_linear_batch(Q, topk, tids)
            for b in range(Q.shape[0]):
                wi, wd = o.query_linear(Q[b], topk, tids)
                assert_same_result((ids[b], d[b]), (wi, wd), "wide linear M=%d k=%d S=%d b=%d" % (M, topk, len(tids), b))
                n_tied += int(len(set(np.asarray(wd).tolist())) < len(wd))
    assert n_tied > 0
    g.reconfigure(30, 3); o.reconfigure(30, 3)
    assert g.coarse_centers == o.coarse_centers and g.posting_lists == o.posting_lists
    g.add_codes(codes[:100], True); o.add_codes(codes[:100], True)
    assert g.posting_lists == o.posting_lists
    for topk, L in ((1, 100), (5, 100), (3, 3), (20, 900), (2, N)):
        for tids in (E, sub):
            ids, d, cnt = g.query_ivf_batch(Q, topk, tids, L)
            for b in range(Q.shape[0]):
                n = int(cnt[b])
                assert_same_result((ids[b, :n], d[b, :n]), o.query_ivf(Q[b], topk, tids, L), "wide ivf M=%d k=%d L=%d S=%d b=%d" % (M, topk, L, len(tids), b))


def test_gpu_ivf_empty_and_tail_vs_oracle():
    from rii_amd import RiiGpu
    arch = "avx512"
    cw, codes, qs = make_problem(11, 8, 64, 4, 4000, "unit")
    g = RiiGpu(cw, False, simd_arch=arch)
    o = O.OracleRii(cw, False, simd_arch=arch)
    g.add_codes(codes, False); o.add_codes(codes, False)
    g.reconfigure(200, 3); o.reconfigure(200, 3)
    assert g.posting_lists == o.posting_lists
    rng = np.random.default_rng(3)
    n_empty = 0
    for trial in range(60):
        S = int(rng.integers(30, 400))
        tids = np.sort(rng.choice(4000, S, replace=False)).astype(np.int64)
        topk = int(rng.integers(1, 25))
        L = int(rng.integers(topk, S + 1))
        ids, d, cnt = g.query_ivf_batch(qs[:6], topk, tids, L)
        for b in range(6):
            want = o.query_ivf(qs[b], topk, tids, L)
            n = int(cnt[b])
            assert_same_result((ids[b, :n], d[b, :n]), want, "trial %d b=%d" % (trial, b))
            n_empty += (len(want[0]) == 0)


@pytest.mark.parametrize("dup", [0, 1500])
def test_gpu_ivf_small_calls_flag_wait_vs_oracle(dup):
    """Small host-pointer inverted-index calls (the README pattern: one query per call): the fused kernel writes rows, counts and
    fallback flags into the engine's pinned block and raises a sequence flag per query (option host_spin); against the oracle and
    against the copy + synchronise form, with duplicated codes (exact ties -> the flagged fallback runs behind the wait), target
    ids, every query forced through the fallback, and the emulation kernels (ivf_fused = 0)."""
    from rii_amd import RiiGpu
    arch = "avx512"
    cw, codes, qs = make_problem(21 + dup, 32, 256, 4, 10000, "sift" if dup else "unit", dup=dup)
    g = RiiGpu(cw, False, simd_arch=arch)
    o = O.OracleRii(cw, False, simd_arch=arch)
    g.add_codes(codes, False); o.add_codes(codes, False)
    g.reconfigure(100, 3); o.reconfigure(100, 3)
    assert g.posting_lists == o.posting_lists
    rng = np.random.default_rng(5)
    sub = np.sort(rng.choice(10000, 700, replace=False)).astype(np.int64)
    for force, fused in ((0, 1), (1, 1), (0, 0)):
        g.set_option("ivf_force_exact", force)
        g.set_option("ivf_fused", fused)
        for topk, L in ((1, 100), (3, 100), (10, 400), (50, 60)):
            for tids in (E, sub):
                if len(tids) and L > len(tids):
                    continue
                for B in (1, 4):
                    res = {}
                    for spin in (1, 0):
                        g.set_option("host_spin", spin)
                        res[spin] = g.query_ivf_batch(qs[:B], topk, tids, L)
                    g.set_option("host_spin", 1)
                    for b in range(B):
                        want = o.query_ivf(qs[b], topk, tids, L)
                        for spin in (1, 0):
                            ids, d, cnt = res[spin]
                            n = int(cnt[b])
                            assert_same_result((ids[b, :n], d[b, :n]), want,
                                               "k=%d L=%d S=%d B=%d b=%d force=%d fused=%d spin=%d" % (topk, L, len(tids), B, b, force, fused, spin))
    g.set_option("ivf_force_exact", 0); g.set_option("ivf_fused", 1)
    ids, d = g.query_ivf(qs[0], 3, E, 100)                       # the one-query entry point
    want = o.query_ivf(qs[0], 3, E, 100)
    assert_same_result((ids, d), want, "query_ivf")


@pytest.mark.parametrize("case", [(32, 256, 4, 200000, "unit", 0), (32, 256, 4, 60000, "sift", 20000), (16, 256, 6, 70000, "unit", 0),
                                  (8, 16, 2, 40000, "round", 0), (64, 256, 2, 30000, "unit", 3000)])
def test_gpu_few_queries_large_index_one_launch_vs_oracle(case):
    """One to eight queries per host call on an index too large for the one-block kernel: slice_topk_kernel (slices of the codes on
    all CUs, the last block merges) against the oracle and against the general path (option slice_topk = 0); exactly tied distances
    (duplicated codes, integer-valued tables) come back as a flag and are redone on the general path -- same answers required."""
    from rii_amd import RiiGpu
    M, Ks, Ds, N, scale, dup = case
    if scale == "round":
        rng = np.random.default_rng(N)
        cw = np.round(rng.random((M, Ks, Ds)) * 3.0).astype(np.float32)
        qs = np.round(rng.random((16, M * Ds)) * 3.0).astype(np.float32)
        codes = rng.integers(0, Ks, size=(N, M), dtype=np.uint8)
    else:
        cw, codes, qs = make_problem(N + M, M, Ks, Ds, N, scale, dup=dup)
    rng = np.random.default_rng(N + 2)
    g = RiiGpu(cw, False, simd_arch="avx512")
    o = O.OracleRii(cw, False, simd_arch="avx512")
    g.add_codes(codes, False); o.add_codes(codes, False)
    assert g.get_option("slice_topk") == 1
    sub = rng.permutation(N)[:N // 2].astype(np.int64)            # unsorted: positions, not ids, break ties
    n_tied = 0
    for topk in (1, 3, 5, 10, 128):                               # (topk <= 5: wave-level selection; above: histogram bound)
        for tids in (E, sub[:900], sub):                          # (900 ids: inputs small enough for the flag wait; N / 2: they are not)
            for B in (1, 3, 8):
                g.set_option("slice_topk", 1)
                ids, d = g.query_linear_batch(qs[:B], topk, tids)
                g.set_option("slice_topk", 0)
                ids0, d0 = g.query_linear_batch(qs[:B], topk, tids)
                for b in range(B):
                    want = o.query_linear(qs[b], topk, tids)
                    what = "few queries k=%d S=%d B=%d b=%d" % (topk, len(tids), B, b)
                    assert_same_result((ids[b], d[b]), want, what)
                    assert_same_result((ids0[b], d0[b]), want, what + " (general path)")
                    n_tied += int(len(np.unique(np.asarray(want[1]))) < topk)
    g.set_option("slice_topk", 1)
    if dup or scale == "round":
        assert n_tied > 0, "the case was meant to produce exactly tied distances inside the top-k"


def test_gpu_flag_wait_soak():
    """Thousands of one-query calls with changing topk / search kind / target ids through the flag wait (host_spin = 1), each
    compared with the answer of the copy + synchronise form: a flag that is seen before its rows (or a stale word that equals
    the awaited sequence number -- which a flag placed behind the rows once did) shows up here."""
    from rii_amd import RiiGpu
    cw, codes, qs = make_problem(77, 32, 256, 4, 10000, "unit")
    rng = np.random.default_rng(77)
    qs = rng.random((64, 128)).astype(np.float32)
    g = RiiGpu(cw, False, simd_arch="avx512")
    g.add_codes(codes, False)
    g.reconfigure(100, 3)
    sub = np.sort(rng.choice(10000, 300, replace=False)).astype(np.int64)
    shapes = [("linear", 1, E), ("linear", 3, E), ("linear", 10, sub), ("linear", 64, E), ("ivf", 1, E), ("ivf", 3, E), ("ivf", 20, sub)]
    g.set_option("host_spin", 0)
    want = {}
    for si, (kind, topk, tids) in enumerate(shapes):
        for qi in range(len(qs)):
            want[si, qi] = g.query_linear(qs[qi], topk, tids) if kind == "linear" else g.query_ivf(qs[qi], topk, tids, 200)
    g.set_option("host_spin", 1)
    for it in range(6000):
        si, qi = int(rng.integers(len(shapes))), int(rng.integers(len(qs)))
        kind, topk, tids = shapes[si]
        got = g.query_linear(qs[qi], topk, tids) if kind == "linear" else g.query_ivf(qs[qi], topk, tids, 200)
        assert got == want[si, qi], "call %d: %s topk=%d S=%d q=%d" % (it, kind, topk, len(tids), qi)
    # small batches as well (one flag word per query)
    for it in range(300):
        B = int(rng.integers(2, 9))
        rows = rng.integers(0, len(qs), B)
        ids, d = g.query_linear_batch(qs[rows], 3, None)
        for j, qi in enumerate(rows):
            assert (ids[j].tolist(), d[j].tolist()) == want[1, int(qi)], "batch call %d row %d" % (it, j)


def test_gpu_ivf_empty_return_with_stale_lists():
    """rii.h:324-325 is reachable when codes were appended with update_flag=False after a reconfigure (lists
    cover fewer than L ids): fewer than topk hits in the first w lists, then the walk over the unsorted tail
    of the coarse order never reaches L => ([], [])."""
    from rii_amd import RiiGpu
    arch = "avx512"
    cw, codes, qs = make_problem(13, 8, 64, 4, 5050, "unit")
    g = RiiGpu(cw, False, simd_arch=arch)
    o = O.OracleRii(cw, False, simd_arch=arch)
    g.add_codes(codes[:50], False); o.add_codes(codes[:50], False)
    g.reconfigure(10, 3); o.reconfigure(10, 3)
    g.add_codes(codes[50:], False); o.add_codes(codes[50:], False)
    n_empty = n_full = 0
    for topk, L in ((20, 100), (20, 40), (3, 30), (12, 50), (1, 51), (6, 49)):
        ids, d, cnt = g.query_ivf_batch(qs[:8], topk, None, L)
        for b in range(8):
            want = o.query_ivf(qs[b], topk, E, L)
            n = int(cnt[b])
            assert_same_result((ids[b, :n], d[b, :n]), want, "k=%d L=%d b=%d" % (topk, L, b))
            n_empty += (len(want[0]) == 0)
            n_full += (len(want[0]) > 0)
    assert n_empty > 0 and n_full > 0


def test_gpu_single_query_api_types():
    """tests/test_rii.py:127-130: query_linear/query_ivf return (list[int], list[float])."""
    from rii_amd import RiiGpu
    cw, codes, qs = make_problem(1, 4, 20, 10, 1000, "unit")
    g = RiiGpu(cw, False)
    g.add_codes(codes, False)
    g.reconfigure(20, 5)
    ids, d = g.query_linear(qs[0], 10, E)
    assert isinstance(ids, list) and isinstance(ids[0], int) and isinstance(d, list) and isinstance(d[0], float)
    assert len(ids) == 10 and np.all(np.diff(d) >= 0)
    ids2, d2 = g.query_linear(qs[0], 10, np.arange(1000, dtype=np.int64))
    assert ids == ids2 and d == d2                                     # test_rii.py:138-140
    ids4, d4 = g.query_ivf(qs[0], 10, np.arange(1000, dtype=np.int64), 1000)
    assert ids4 == ids and d4 == d                                     # test_rii.py:178-181
    with pytest.raises(TypeError):
        g.query_linear(qs[0].astype(np.float64), 10, E)                # noconvert, main.cpp:18
    with pytest.raises(ValueError):
        g.query_linear(qs[0], 2000, E)                                 # assert topk <= N, rii.h:202


def test_gpu_errors_not_aborts():
    from rii_amd import RiiGpu, RiiAmdError
    cw, codes, qs = make_problem(1, 4, 20, 10, 100, "unit")
    g = RiiGpu(cw, False)
    with pytest.raises(RiiAmdError):
        g.add_codes(codes, True)        # reference: bare `throw;` => std::terminate (rii.h:166-170)
    g.add_codes(codes, False)
    with pytest.raises(ValueError):
        g.reconfigure(0, 5)
    with pytest.raises(ValueError):
        g.reconfigure(101, 5)


def test_gpu_pickle_roundtrip():
    import pickle
    from rii_amd import RiiGpu
    cw, codes, qs = make_problem(5, 8, 256, 4, 3000, "unit")
    g = RiiGpu(cw, False, simd_arch="avx")
    g.add_codes(codes, False)
    g.reconfigure(30, 3)
    g2 = pickle.loads(pickle.dumps(g))
    assert g2._simd == "avx"                                 # travels as a constructor argument, not in the state
    assert g2.N == g.N and g2.nlist == g.nlist and g2.posting_lists == g.posting_lists
    a = g.query_ivf_batch(qs[:4], 5, None, 300)
    b = g2.query_ivf_batch(qs[:4], 5, None, 300)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    # the state is exactly the reference's 5-tuple of plain Python values (src/main.cpp:35-38)
    st = g.__getstate__()
    assert isinstance(st, tuple) and len(st) == 5
    assert isinstance(st[0], list) and isinstance(st[0][0][0][0], float) and isinstance(st[1], bool)
    assert isinstance(st[2][0][0], int) and isinstance(st[3][0], int) and len(st[3]) == 3000 * 8
    assert isinstance(st[4], list) and isinstance(st[4][0], list) and sum(len(l) for l in st[4]) == 3000
    with pytest.raises(RuntimeError):
        RiiGpu.__new__(RiiGpu).__setstate__(st + ("extra",))   # "Invalid state when reading pickled item", main.cpp:41-43


@pytest.mark.parametrize("arch", ["avx512", "avx"])
def test_gpu_loads_reference_pickle_state(arch):
    """f2: a state produced by the real reference (recorded by tests/gen_golden.py) loads into the engine through
    __setstate__ and answers like the reference re-created from the same state."""
    from rii_amd import RiiGpu

    def make(state):
        g = RiiGpu.__new__(RiiGpu)
        g._simd = arch
        g.__setstate__(state)
        return g
    replay_state(make, arch)


def test_gpu_state_loads_into_the_real_reference():
    """f2, other direction: an index built on the GPU, handed to the compiled reference (oracle/_ref) through its own
    py::pickle set-state, answers identically there."""
    from rii_amd import RiiGpu
    ref, arch, _ = O.load_reference()
    if ref is None:
        pytest.skip("no runnable oracle/_ref build on this host")
    cw, codes, qs = make_problem(31, 8, 64, 4, 4000, "unit", dup=300)
    g = RiiGpu(cw, False, simd_arch=arch)
    g.add_codes(codes[:3500], False)
    g.reconfigure(40, 4)
    g.add_codes(codes[3500:], True)
    r = ref.RiiCpp.__new__(ref.RiiCpp)
    r.__setstate__(g.__getstate__())
    assert r.N == g.N and r.nlist == g.nlist and r.posting_lists == g.posting_lists
    sub = np.sort(np.random.default_rng(2).choice(4000, 700, replace=False)).astype(np.int64)
    for b in range(8):
        for tids in (E, sub):
            for topk in (1, 6):
                assert_same_result(g.query_linear(qs[b], topk, tids), r.query_linear(qs[b], topk, tids), "state->ref linear")
                assert_same_result(g.query_ivf(qs[b], topk, tids, 150), r.query_ivf(qs[b], topk, tids, 150), "state->ref ivf")


def test_gpu_mfma_lut_within_tolerance():
    """north_star: distances within 1e-4 (relative) when the table is built on the matrix cores -- against the ORACLE's table
    (RiiCpp::DTable, src/rii.h:361-373), not against the engine's own exact mode; the batch is large enough for the default
    filter path (lut_build_mfma_kernel -> lut_quantize_kernel -> fscan_mx_kernel -> re-rank from the matrix-core table)."""
    from rii_amd import RiiGpu
    cw, codes, qs = make_problem(77, 32, 256, 4, 20000, "sift")
    Q = np.concatenate([qs, qs + 1.0, qs * 0.5, qs + 3.0]).astype(np.float32)            # 64 queries: above fast_min_batch
    g = RiiGpu(cw, False)
    g.add_codes(codes, False)
    want = np.stack([O.dtable(cw, Q[b], "avx512") for b in range(16)]).reshape(16, -1)
    ids_e, d_e = g.query_linear_batch(Q, 1, None)
    g.set_option("lut_mode", "mfma")
    approx = g.dtable(Q[:16]).reshape(16, -1)
    ids_m, d_m = g.query_linear_batch(Q, 1, None)
    g.set_option("timing", 1)
    g.timing_reset()
    g.query_linear_batch(Q, 1, None)
    assert g.timing_read("quant")[1] > 0 and g.timing_read("scan")[1] > 0      # the filter path ran on the matrix-core tables
    g.set_option("timing", 0)
    rel = np.abs(approx - want) / np.maximum(np.abs(want), 1.0)
    assert rel.max() < 1e-4, rel.max()
    o = O.OracleRii(cw, False, simd_arch="avx512")
    o.add_codes(codes, False)
    for b in range(0, 64, 7):
        wi, wd = o.query_linear(Q[b], 1, np.array([], np.int64))
        assert abs(float(d_m[b, 0]) - wd[0]) <= 1e-4 * max(abs(wd[0]), 1.0), (b, d_m[b, 0], wd[0])
    assert np.allclose(d_m, d_e, rtol=1e-4)
    assert (ids_m == ids_e).mean() >= 0.9          # tie-tolerant: near-ties may flip under different rounding


@pytest.mark.parametrize("shape", [(32, 256, 4, 50000, "sift", 0), (32, 256, 4, 30000, "unit", 6000),
                                   (16, 256, 6, 40000, "unit", 0), (8, 256, 16, 20000, "sift", 2000),
                                   (4, 20, 10, 5000, "unit", 0), (20, 256, 2, 9000, "unit", 0), (3, 7, 37, 3000, "unit", 500),
                                   (64, 256, 2, 30000, "unit", 3000), (48, 200, 3, 20000, "sift", 0), (72, 256, 1, 8000, "unit", 0),
                                   (100, 256, 1, 4000, "unit", 0)])
def test_gpu_filter_rerank_equals_exact_scan(shape):
    """scan_mode=1 (8-bit filter + exact re-rank, the default) must return exactly what scan_mode=0 (exact scan of
    every code) returns -- ids and distance bits -- incl. with heavy duplication (tied minima) and for subsets."""
    from rii_amd import RiiGpu
    M, Ks, Ds, N, scale, dup = shape
    cw, codes, qs = make_problem(123, M, Ks, Ds, N, scale, dup=dup)
    rng = np.random.default_rng(2)
    Q = np.concatenate([qs, rng.permutation(qs.reshape(-1)).reshape(qs.shape), qs * 0.5, qs[:5] * 0.0 + 1e3], 0)
    g = RiiGpu(cw, False)
    g.add_codes(codes, False)
    g.set_option("fast_min_batch", 0)                  # force the filter path also for this small batch
    sub = np.sort(rng.choice(N, N // 3, replace=False)).astype(np.int64)
    for tids in (None, sub):
        g.set_option("scan_mode", 1)
        i1, d1 = g.query_linear_batch(Q, 1, tids)
        g.set_option("scan_mode", 0)
        i0, d0 = g.query_linear_batch(Q, 1, tids)
        assert np.array_equal(i1, i0)
        assert np.array_equal(d1.view(np.uint32), d0.view(np.uint32))


def test_gpu_filter_rerank_overflow_falls_back_exactly():
    """More tied-at-the-minimum codes than candidate slots: the re-rank block scans all codes itself."""
    from rii_amd import RiiGpu
    cw, codes, qs = make_problem(7, 32, 256, 4, 40000, "unit")
    codes[5000:25000] = codes[4999]                      # 20001 identical codes
    g = RiiGpu(cw, False)
    g.add_codes(codes, False)
    from oracle import oracle as O
    q = O.OracleRii(cw).codewords[:, codes[4999], :][np.arange(32), np.arange(32)].reshape(-1)   # the code's own centroid
    Q = np.stack([q, qs[0], qs[1]]).astype(np.float32)
    g.set_option("cand_cap", 64)
    g.set_option("fast_min_batch", 0)
    assert g.get_option("fused_tables") == 1
    i1, d1 = g.query_linear_batch(Q, 1, None)            # default: table-free re-rank (it builds its table in LDS on overflow)
    g.set_option("fused_tables", 0)
    i2, d2 = g.query_linear_batch(Q, 1, None)            # round 2's path: re-rank from the fp32 table in global memory
    g.set_option("scan_mode", 0)
    i0, d0 = g.query_linear_batch(Q, 1, None)
    assert i0[0, 0] == 4999 and np.array_equal(i1, i0) and np.array_equal(d1.view(np.uint32), d0.view(np.uint32))
    assert np.array_equal(i2, i0) and np.array_equal(d2.view(np.uint32), d0.view(np.uint32))


def test_gpu_ivf_fused_equals_emulation_path():
    """ivf_fused=1 (default: one launch, flag-gated exact fallback) vs ivf_fused=0 (always the emulation kernels)."""
    from rii_amd import RiiGpu
    cw, codes, qs = make_problem(31, 16, 256, 6, 30000, "unit", dup=3000)
    g = RiiGpu(cw, False)
    g.add_codes(codes, False)
    g.reconfigure(150, 3)
    cen = g.coarse_centers_array()
    cen[5] = cen[3]; cen[77] = cen[3]                    # duplicated centres => exactly tied coarse distances
    g.set_coarse_centers(cen)
    rng = np.random.default_rng(4)
    sub = np.sort(rng.choice(30000, 2500, replace=False)).astype(np.int64)
    Q = np.concatenate([qs, qs[::-1] * 0.7], 0)
    for topk, L in ((1, 200), (1, 3000), (4, 200), (10, 17), (1, 1), (50, 3000), (129, 400), (130, 3000), (200, 250), (700, 3900),
                    (7, 30000)):
        for tids in (None, sub):
            if tids is not None and topk > len(tids):
                continue
            g.set_option("ivf_fused", 1)
            a = g.query_ivf_batch(Q, topk, tids, L)
            g.set_option("ivf_fused", 0)
            b = g.query_ivf_batch(Q, topk, tids, L)
            g.set_option("ivf_fused", 1)
            g.set_option("ivf_force_exact", 1)              # every query through the LDS replay (one wave walks the heap)
            c = g.query_ivf_batch(Q, topk, tids, L)
            g.set_option("ivf_force_exact", 0)
            assert np.array_equal(a[2], b[2]) and np.array_equal(a[2], c[2])
            for r in range(Q.shape[0]):
                n = int(a[2][r])
                assert np.array_equal(a[0][r, :n], b[0][r, :n]) and np.array_equal(a[1][r, :n].view(np.uint32), b[1][r, :n].view(np.uint32))
                assert np.array_equal(a[0][r, :n], c[0][r, :n]) and np.array_equal(a[1][r, :n].view(np.uint32), c[1][r, :n].view(np.uint32))


@pytest.mark.parametrize("shape", [(32, 256, 4, 60000, "sift", 0), (32, 256, 4, 30000, "unit", 9000),
                                   (16, 256, 6, 40000, "unit", 0), (4, 20, 10, 5000, "unit", 0),
                                   (3, 7, 37, 3000, "unit", 500), (20, 256, 2, 2000, "unit", 0),
                                   (64, 256, 2, 20000, "unit", 2000)])
def test_gpu_topk_filter_rerank_equals_sort_path(shape):
    """topk > 1: the two-pass filter (segment minima -> k-th bound -> candidates) + exact streaming top-k must equal
    the exhaustive path (all exact keys, full segmented sort) bit for bit, both being in canonical (dist, id) order."""
    from rii_amd import RiiGpu
    M, Ks, Ds, N, scale, dup = shape
    cw, codes, qs = make_problem(321, M, Ks, Ds, N, scale, dup=dup)
    g = RiiGpu(cw, False)
    g.add_codes(codes, False)
    rng = np.random.default_rng(8)
    sub = np.sort(rng.choice(N, N // 4, replace=False)).astype(np.int64)
    Q = np.concatenate([qs, qs[:7] * 0.3], 0)
    for topk in (2, 10, 100, 400):
        for tids in (None, sub):
            if tids is not None and topk > len(tids):
                continue
            g.set_option("scan_mode", 1)
            i1, d1 = g.query_linear_batch(Q, topk, tids)
            g.set_option("scan_mode", 0)
            i0, d0 = g.query_linear_batch(Q, topk, tids)
            assert np.array_equal(d1.view(np.uint32), d0.view(np.uint32)), (shape, topk)
            assert np.array_equal(i1, i0), (shape, topk)


@pytest.mark.parametrize("case", [(8, 16, 2, 5000, 3.0), (4, 8, 3, 1200, 2.0), (32, 256, 4, 70000, 255.0), (16, 256, 6, 66000, 4.0)])
def test_gpu_linear_topk_tie_order_vs_oracle(case):
    """Exactly tied distances (integer-valued codebooks and queries, duplicated codes): the ids must come back in the order
    std::partial_sort leaves them in (src/rii.h:234-235) -- filter + re-rank path, sort path (k > 1023, scan_mode 0),
    subset search, k == N, and the LDS-friendly scan order (N >= 65536) included."""
    from rii_amd import RiiGpu
    M, Ks, Ds, N, vmax = case
    rng = np.random.default_rng(M * 7 + N)
    cw = np.round(rng.random((M, Ks, Ds)) * vmax).astype(np.float32)
    qs = np.round(rng.random((9, M * Ds)) * vmax).astype(np.float32)
    codes = rng.integers(0, Ks, size=(N, M), dtype=np.uint8)
    codes[rng.integers(0, N, N // 4)] = codes[rng.integers(0, N, N // 4)]
    arch = "avx512"
    g = RiiGpu(cw, False, simd_arch=arch)
    o = O.OracleRii(cw, False, simd_arch=arch)
    g.add_codes(codes, False); o.add_codes(codes, False)
    sub = np.sort(rng.choice(N, N // 3, replace=False)).astype(np.int64)
    ks = tuple(k for k in (2, 5, 50, 700, 1500) if k < N) + ((N,) if N <= 5000 else ())
    n_tied = 0
    for topk in ks:
        for tids in (E, sub):
            if len(tids) and topk > len(tids):
                continue
            for mode in ((1, 0) if topk <= 50 else (1,)):
                g.set_option("scan_mode", mode)
                ids, d = g.query_linear_batch(qs, topk, tids)
                for b in range(len(qs)):
                    want = o.query_linear(qs[b], topk, tids)
                    assert_same_result((ids[b], d[b]), want, "tie order k=%d S=%d mode=%d b=%d" % (topk, len(tids), mode, b))
                    n_tied += int(len(np.unique(np.asarray(want[1]))) < topk)
    g.set_option("scan_mode", 1)
    assert n_tied > 0, "the case was meant to produce exactly tied distances inside the top-k"


@pytest.mark.parametrize("case", [(32, 256, 4, 10000, "sift", 400), (32, 256, 4, 12288, "unit", 0), (16, 256, 6, 9000, "unit", 3000),
                                  (8, 16, 2, 4000, "round", 0), (4, 8, 3, 700, "round", 0), (64, 256, 2, 2000, "unit", 100)])
def test_gpu_small_index_topk_one_launch_vs_oracle(case):
    """A small batch on a small index (the reference's README pattern: N = 10 000, topk = 3, one query per call):
    the one-launch path (smalltopk.hip) against the oracle AND against the general path (option small_topk = 0), exactly tied
    distances (duplicated codes, integer-valued tables), subsets, k == n and k just below the path's limit included."""
    from rii_amd import RiiGpu
    M, Ks, Ds, N, scale, dup = case
    if scale == "round":
        rng = np.random.default_rng(N)
        cw = np.round(rng.random((M, Ks, Ds)) * 3.0).astype(np.float32)
        qs = np.round(rng.random((16, M * Ds)) * 3.0).astype(np.float32)
        codes = rng.integers(0, Ks, size=(N, M), dtype=np.uint8)
    else:
        cw, codes, qs = make_problem(N + M, M, Ks, Ds, N, scale, dup=dup)
    rng = np.random.default_rng(N + 1)
    g = RiiGpu(cw, False, simd_arch="avx512")
    o = O.OracleRii(cw, False, simd_arch="avx512")
    g.add_codes(codes, False); o.add_codes(codes, False)
    assert g.get_option("small_topk") == 1 and g.get_option("host_spin") == 1
    sub = rng.permutation(N)[:N // 3].astype(np.int64)            # unsorted subset: positions, not ids, break ties
    n_tied = 0
    for topk in [k for k in (1, 2, 3, 10, 100, 1023) if k <= N // 3] + ([N] if N <= 1023 else []):
        for tids in (E, sub):
            if len(tids) and topk > len(tids):
                continue
            for B in (1, 5):
                g.set_option("small_topk", 1)
                ids, d = g.query_linear_batch(qs[:B], topk, tids)          # rows written to the pinned block by the kernel, flag spin
                g.set_option("host_spin", 0)
                ids1, d1 = g.query_linear_batch(qs[:B], topk, tids)        # D2H copy + stream synchronisation
                g.set_option("host_spin", 1)
                g.set_option("small_topk", 0)
                ids0, d0 = g.query_linear_batch(qs[:B], topk, tids)
                for b in range(B):
                    want = o.query_linear(qs[b], topk, tids)
                    what = "small topk k=%d S=%d B=%d b=%d" % (topk, len(tids), B, b)
                    assert_same_result((ids[b], d[b]), want, what)
                    assert_same_result((ids1[b], d1[b]), want, what + " (host_spin = 0)")
                    assert_same_result((ids0[b], d0[b]), want, what + " (general path)")
                    n_tied += int(len(np.unique(np.asarray(want[1]))) < topk)
    g.set_option("small_topk", 1)
    if dup or scale == "round":
        assert n_tied > 0, "the case was meant to produce exactly tied distances inside the top-k"


def test_gpu_large_batch_is_chunked_transparently():
    """B above the internal pass size (8192 queries) is processed in slices with identical per-row results."""
    from rii_amd import RiiGpu
    cw, codes, qs = make_problem(17, 8, 256, 4, 3000, "unit")
    g = RiiGpu(cw, False)
    g.add_codes(codes, False)
    g.reconfigure(30, 2)
    rng = np.random.default_rng(0)
    Q = rng.random((20000, 32)).astype(np.float32)
    ids, d = g.query_linear_batch(Q, 3, None)
    i2, d2 = g.query_linear_batch(Q[15000:15040], 3, None)
    assert np.array_equal(ids[15000:15040], i2) and np.array_equal(d[15000:15040], d2)
    a = g.query_ivf_batch(Q, 2, None, 200)
    b = g.query_ivf_batch(Q[9000:9050], 2, None, 200)
    assert all(np.array_equal(x[9000:9050], y) for x, y in zip(a, b))


@pytest.mark.parametrize("M,Ks,Ds,N", [(1, 1, 1, 1), (1, 2, 3, 2), (2, 256, 1, 3), (32, 256, 4, 1), (5, 3, 2, 64), (1, 256, 128, 300)])
def test_gpu_degenerate_shapes_vs_oracle(M, Ks, Ds, N):
    """Smallest possible everything: one code, one subspace, one codeword, nlist == 1 and nlist == N, L == topk == 1."""
    from rii_amd import RiiGpu
    cw, codes, qs = make_problem(M * 1000 + N, M, Ks, Ds, N, "unit")
    g = RiiGpu(cw, False, simd_arch="avx512")
    o = O.OracleRii(cw, False, simd_arch="avx512")
    g.add_codes(codes, False); o.add_codes(codes, False)
    assert g.query_linear_batch(np.zeros((0, M * Ds), np.float32), 1, None)[0].shape == (0, 1)     # empty batch
    for topk in sorted({1, N}):
        ids, d = g.query_linear_batch(qs[:3], topk, None)
        for b in range(3):
            assert_same_result((ids[b], d[b]), o.query_linear(qs[b], topk, E), "degenerate linear k=%d b=%d" % (topk, b))
    for nlist in sorted({1, N}):
        g.reconfigure(nlist, 2); o.reconfigure(nlist, 2)
        assert g.coarse_centers == o.coarse_centers and g.posting_lists == o.posting_lists
        for topk, L in {(1, 1), (1, N), (N, N)}:
            ids, d, cnt = g.query_ivf_batch(qs[:3], topk, None, L)
            for b in range(3):
                n = int(cnt[b])
                assert_same_result((ids[b, :n], d[b, :n]), o.query_ivf(qs[b], topk, E, L), "nlist=%d k=%d L=%d" % (nlist, topk, L))
    one = np.array([N - 1], np.int64)
    ids, d = g.query_linear_batch(qs[:2], 1, one)
    assert (ids == N - 1).all()


@pytest.mark.parametrize("arch", ["avx512", "avx"])
def test_gpu_stale_lists_golden(arch):
    replay_stale(gpu_engine(arch), arch)


def test_gpu_input_layouts_and_bad_inputs():
    """Strided / Fortran-ordered arrays are accepted like pybind11's array_t accepts them; NaN queries and unsorted
    target ids must not crash or hang the engine."""
    from rii_amd import RiiGpu
    cw, codes, qs = make_problem(9, 8, 256, 4, 4000, "unit")
    g = RiiGpu(np.asfortranarray(cw), False)
    g.add_codes(np.asfortranarray(codes), False)
    g.reconfigure(40, 2)
    big = np.zeros((32, 64), np.float32)
    big[::2, ::2] = qs
    strided = big[::2, ::2]                                  # non-contiguous view of the 16 queries
    a = g.query_linear_batch(strided, 3, None)
    b = g.query_linear_batch(np.ascontiguousarray(strided), 3, None)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert g.query_linear(strided[5], 1, E) == g.query_linear(qs[5].copy(), 1, E)
    bad = qs[:4].copy()
    bad[1, 3] = np.nan
    bad[2, 0] = np.inf
    ids, d = g.query_linear_batch(bad, 2, None)             # must return; rows 0 and 3 are ordinary queries
    ok = g.query_linear_batch(qs[:4], 2, None)
    assert np.array_equal(ids[[0, 3]], ok[0][[0, 3]]) and np.array_equal(d[[0, 3]], ok[1][[0, 3]])
    g.query_ivf_batch(bad, 2, None, 300)
    o = O.OracleRii(cw, False)
    o.add_codes(codes, False)
    o.set_coarse_centers(np.array(g.coarse_centers, np.uint8))
    rng = np.random.default_rng(3)
    for tids in (np.array([5, 3, 9], np.int64), np.array([3, 3, 9, 9, 9, 700], np.int64),
                 rng.integers(0, 4000, 900).astype(np.int64)):   # unsorted / duplicated ids: scored as given (rii.h:218-228)
        for topk in (1, 2, 3):
            for b in range(4):
                assert_same_result(g.query_linear(qs[b], topk, tids), o.query_linear(qs[b], topk, tids), "odd tids")
    dups = np.sort(rng.integers(0, 4000, 1500)).astype(np.int64)         # sorted with duplicates: binary_search still works
    for b in range(4):
        assert_same_result(g.query_ivf(qs[b], 3, dups, 200), o.query_ivf(qs[b], 3, dups, 200), "ivf dup tids")
    with pytest.raises(ValueError):
        g.query_linear(qs[0], 1, np.array([3, 4000], np.int64))       # out of range
    with pytest.raises(TypeError):
        g.query_linear(qs[0], 1, np.array([3, 9], np.int32))          # noconvert (src/main.cpp:20)
    with pytest.raises(TypeError):
        g.add_codes(codes.astype(np.int32), False)


def test_concurrent_callers_one_engine():
    """Several host threads hammer ONE engine (ctypes drops the GIL during the call): the engine serialises them, every
    call still returns exactly what a lone caller gets."""
    import threading
    from rii_amd import RiiGpu
    cw, codes, _ = make_problem(77, 16, 256, 4, 20000, "unit")
    qs = np.random.default_rng(77).random((64, 64)).astype(np.float32)
    g = RiiGpu(cw, False, simd_arch="avx512")
    g.add_codes(codes, False)
    g.reconfigure(40, 3)
    want_lin = g.query_linear_batch(qs, 3, None)
    want_ivf = g.query_ivf_batch(qs, 3, None, 400)
    errors = []

    def worker(kind):
        try:
            for it in range(20):
                if kind == 0:
                    ids, d = g.query_linear_batch(qs, 3, None)
                    assert np.array_equal(ids, want_lin[0]) and np.array_equal(d, want_lin[1])
                elif kind == 1:
                    ids, d, c = g.query_ivf_batch(qs, 3, None, 400)
                    assert np.array_equal(ids, want_ivf[0]) and np.array_equal(d, want_ivf[1])
                else:
                    i1, d1 = g.query_linear(qs[it % 64], 3, E)
                    assert list(i1) == list(want_lin[0][it % 64])
        except Exception as ex:          # noqa: BLE001 -- reported to the main thread
            errors.append(repr(ex))

    ts = [threading.Thread(target=worker, args=(k % 3,)) for k in range(6)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors[:3]


def test_alternating_streams_share_scratch_safely():
    """Device-pointer calls on two different streams without host synchronisation in between: the engine chains
    them (they share scratch buffers), so both result sets are right."""
    import torch
    from rii_amd import RiiGpu
    cw, codes, _ = make_problem(78, 32, 256, 4, 60000, "unit")
    qs = np.random.default_rng(78).random((256, 128)).astype(np.float32)
    g = RiiGpu(cw, False, simd_arch="avx512")
    g.add_codes(codes, False)
    want_a = g.query_linear_batch(qs[:128], 1, None)
    want_b = g.query_linear_batch(qs[128:], 1, None)
    dev = torch.device("cuda:0")
    qa = torch.from_numpy(qs[:128]).to(dev)
    qb = torch.from_numpy(qs[128:]).to(dev)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    outs = []
    for rep in range(6):
        ia = torch.empty((128, 1), dtype=torch.int64, device=dev); da = torch.empty((128, 1), dtype=torch.float32, device=dev)
        ib = torch.empty((128, 1), dtype=torch.int64, device=dev); db = torch.empty((128, 1), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        g.query_linear_dev(qa.data_ptr(), 128, 1, 0, 0, ia.data_ptr(), da.data_ptr(), s1.cuda_stream)
        g.query_linear_dev(qb.data_ptr(), 128, 1, 0, 0, ib.data_ptr(), db.data_ptr(), s2.cuda_stream)
        outs.append((ia, da, ib, db))
    torch.cuda.synchronize()
    for ia, da, ib, db in outs:
        assert np.array_equal(ia.cpu().numpy(), want_a[0]) and np.array_equal(da.cpu().numpy(), want_a[1])
        assert np.array_equal(ib.cpu().numpy(), want_b[0]) and np.array_equal(db.cpu().numpy(), want_b[1])


@pytest.mark.parametrize("M,Ds", [(32, 4), (16, 4), (64, 2), (32, 2), (16, 2)])
def test_matrix_core_scan_equals_vector_scan(M, Ds):
    """fscan_mx_kernel (option scan_mx = 1, default: table bytes summed by v_smfmac) against fscan_kernel (scan_mx = 0) and the
    exhaustive scan: identical ids and distances for top-1, top-k and subset search; sizes that end inside a group of 16
    codes, inside a 1024-code trip, below one trip and below one group; appends that re-format a partial last group; exact
    duplicates in the database (ties)."""
    from rii_amd import RiiGpu
    rng = np.random.default_rng(4100 + M)
    # M = 64, Ds = 2: the reference's own benchmark shape (D = 128).  M = 16 / 32: the tables come from qlut_fused_kernel (one
    # launch, quarter tables) and top-1 is re-ranked from the codebook; option fused_tables = 0 selects round 2's two-launch path
    cw = rng.random((M, 256, Ds)).astype(np.float32)
    N = 70000 + 13
    codes = rng.integers(0, 256, size=(N, M), dtype=np.uint8)
    codes[rng.integers(0, N, 2000)] = codes[rng.integers(0, N, 2000)]
    qs = rng.random((150, M * Ds)).astype(np.float32)
    g = RiiGpu(cw, False, simd_arch="avx512")
    assert g.get_option("scan_mx") == 1

    def both(topk, tids=None):
        out = []
        # (scan_mx, scan_mode, fused_tables, scan_dual): default; vector-ALU filter; exhaustive scan; round 2's two-launch tables;
        # M = 16 with one tile per block instead of two (fscan_mx_kernel<4> instead of fscan_mx_dual_kernel)
        # last column: quantisation levels of the fused tables (127 = default; 255 = signed bytes, accumulators start at 128 M)
        for mx, mode, ft, dual, lv in ((1, 1, 1, 1, 127), (0, 1, 1, 1, 127), (1, 0, 1, 1, 127), (1, 1, 0, 1, 127), (1, 1, 1, 0, 127),
                                       (1, 1, 0, 0, 127), (1, 1, 1, 1, 63), (1, 1, 1, 0, 63), (1, 1, 1, 1, 255), (1, 1, 1, 0, 255)):
            g.set_option("scan_mx", mx)
            g.set_option("scan_mode", mode)
            g.set_option("fused_tables", ft)
            g.set_option("scan_dual", dual)
            g.set_option("table_levels", lv)
            out.append(g.query_linear_batch(qs, topk, tids))
        g.set_option("scan_mx", 1)
        g.set_option("scan_mode", 1)
        g.set_option("fused_tables", 1)
        g.set_option("scan_dual", 1)
        g.set_option("table_levels", 127)
        g.set_option("scan_pipe", 0)                 # round 4: the in-order judge against the one-group-late judge (default) ...
        out.append(g.query_linear_batch(qs, topk, tids))
        g.set_option("scan_dual", 0)                 # ... also for M = 16 with one tile per block
        out.append(g.query_linear_batch(qs, topk, tids))
        g.set_option("scan_pipe", 2)
        out.append(g.query_linear_batch(qs, topk, tids))
        g.set_option("scan_dual", 1)
        out.append(g.query_linear_batch(qs, topk, tids))
        g.set_option("scan_pipe", 1)
        a = out[0]
        for o_ in out[1:]:
            assert np.array_equal(a[0], o_[0]) and np.array_equal(a[1], o_[1]), (topk, g.N)
        # an odd number of 16-query tiles (the last two-tile block of the M = 16 kernel holds one live tile) and a ragged last tile
        for nq in (40, 33):
            x = g.query_linear_batch(qs[:nq], topk, tids)
            assert np.array_equal(x[0], a[0][:nq]) and np.array_equal(x[1], a[1][:nq]), (topk, g.N, nq)
        return a

    sizes = [7, 16, 100, 1024, 1500, 5000, 33000 + 5, 66000 + 9, N]          # cumulative appends
    done = 0
    for n in sizes:
        g.add_codes(codes[done:n], False)
        done = n
        both(1)
        if n >= 100:
            both(min(10, n))
        if n >= 5000:
            both(100)
            tids = np.sort(rng.choice(n, 3001, replace=False)).astype(np.int64)
            both(1, tids)
            both(7, tids)
    g.clear()
    g.add_codes(codes[:2049], False)
    both(1)
    both(5)


@pytest.mark.parametrize("lanes,nstreams", [(2, 2), (2, 3), (1, 2)])
def test_scratch_lanes_mixed_calls(lanes, nstreams):
    """Two scratch lanes (engine.hip: ScratchSet): calls that alternate between streams overlap on the device, each on its
    own lane; a third stream, host-pointer calls, an append and option changes in between all stay correct.  Linear
    top-1 / top-k / subset and inverted-index calls are mixed so that every scratch buffer is exercised on both lanes."""
    import torch
    from rii_amd import RiiGpu
    cw, codes, _ = make_problem(79, 32, 256, 4, 70000, "unit")
    rng = np.random.default_rng(79)
    codes[rng.integers(0, 70000, 2000)] = codes[rng.integers(0, 70000, 2000)]          # exact ties
    qs = rng.random((3 * 96, 128)).astype(np.float32)
    tids = np.sort(rng.choice(60000, 9000, replace=False)).astype(np.int64)
    g = RiiGpu(cw, False, simd_arch="avx512")
    g.add_codes(codes[:60000], False)
    g.reconfigure(64, 3)
    g.set_option("lanes", lanes)
    assert g.get_option("lanes") == lanes
    dev = torch.device("cuda:0")
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    d_t = torch.from_numpy(tids).to(dev)
    kinds = [("lin", 1, False), ("lin", 10, False), ("lin", 1, True), ("ivf", 5, False), ("ivf", 1, True), ("lin", 3, True)]

    def want(kind, topk, sub, q):
        t = tids if sub else None
        if kind == "lin":
            return g.query_linear_batch(q, topk, t)
        return g.query_ivf_batch(q, topk, t, 300)

    def issue(kind, topk, sub, dq, st):
        B = dq.shape[0]
        with torch.cuda.stream(st):          # the fills are ordered before the engine's writes on the same stream
            ids = torch.full((B, topk), -7, dtype=torch.int64, device=dev)
            d = torch.full((B, topk), -1.0, dtype=torch.float32, device=dev)
            cnt = torch.zeros((B,), dtype=torch.int64, device=dev)
        pt, S = (d_t.data_ptr(), d_t.numel()) if sub else (0, 0)
        if kind == "lin":
            g.query_linear_dev(dq.data_ptr(), B, topk, pt, S, ids.data_ptr(), d.data_ptr(), st.cuda_stream)
        else:
            g.query_ivf_dev(dq.data_ptr(), B, topk, pt, S, 300, ids.data_ptr(), d.data_ptr(), cnt.data_ptr(), st.cuda_stream)
        return ids, d, cnt

    def round_(n_calls):
        qb = [qs[i * 96:(i + 1) * 96] for i in range(3)]
        dq = [torch.from_numpy(q).to(dev) for q in qb]
        wants = {(k, i): want(*k, qb[i]) for k in kinds for i in range(3)}
        torch.cuda.synchronize()
        got = []
        for c in range(n_calls):
            k, i = kinds[c % len(kinds)], c % 3
            got.append((k, i, issue(*k, dq[i], streams[c % nstreams])))
            if c == n_calls // 2:            # a host-pointer call in the middle (engine's own stream, synchronous)
                h = g.query_linear_batch(qb[0], 1, None)
                assert np.array_equal(h[0], wants[(kinds[0], 0)][0])
        torch.cuda.synchronize()
        for k, i, (ids, d, cnt) in got:
            w = wants[(k, i)]
            if k[0] == "lin":
                assert np.array_equal(ids.cpu().numpy(), w[0]) and np.array_equal(d.cpu().numpy(), w[1]), (k, i)
            else:
                n = cnt.cpu().numpy()
                assert np.array_equal(n, w[2]), (k, i)
                for b in range(len(n)):
                    assert np.array_equal(ids[b, :n[b]].cpu().numpy(), w[0][b, :n[b]]), (k, i, b)
                    assert np.array_equal(d[b, :n[b]].cpu().numpy(), w[1][b, :n[b]]), (k, i, b)

    round_(18)
    g.add_codes(codes[60000:], True)          # mutation between rounds: waits for both lanes
    round_(13)


@pytest.mark.parametrize("M,N", [(32, 70000), (16, 66000), (8, 65536 + 1000), (12, 67000), (64, 66500)])
def test_scan_order_does_not_change_results(M, N):
    """The filter stage scans an LDS-friendly permutation of the codes (scanorder.hip); ids, distances and tie-breaks must
    be exactly those of the id-order scan and of the exhaustive scan -- including on a database with exact duplicates,
    across appends (only the windows past the covered prefix are re-ordered) and after clear()."""
    from rii_amd import RiiGpu
    rng = np.random.default_rng(M * 1000 + 5)
    cw = rng.random((M, 256, 4)).astype(np.float32)
    codes = rng.integers(0, 256, size=(N, M), dtype=np.uint8)
    codes[rng.integers(0, N, 3000)] = codes[rng.integers(0, N, 3000)]          # exact ties
    qs = rng.random((200, M * 4)).astype(np.float32)
    g = RiiGpu(cw, False, simd_arch="avx512")

    def all_modes(topk):
        out = []
        for mode, order in ((1, 1), (1, 0), (0, 0), (0, 1)):
            g.set_option("scan_mode", mode)
            g.set_option("scan_order", order)
            out.append(g.query_linear_batch(qs, topk, None))
        g.set_option("scan_mode", 1)
        g.set_option("scan_order", 1)
        return out

    def check(topk):
        a, b, c, d = all_modes(topk)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert np.array_equal(a[1], c[1])
        assert np.array_equal(c[0], d[0]) and np.array_equal(c[1], d[1])      # exhaustive scan, id order vs scan order
        assert np.array_equal(a[0], c[0])                                     # ties included: both replay std::partial_sort

    first = N - 3000
    g.add_codes(codes[:first], False)
    check(1)
    check(10)
    g.add_codes(codes[first:first + 1500], False)          # grows the last, partial window and adds new ones
    check(1)
    g.add_codes(codes[first + 1500:], False)
    check(1)
    check(33)
    want = g.query_linear_batch(qs, 1, None)
    g.clear()
    g.add_codes(codes[::-1].copy(), False)                 # same N, other codes: the stale order must not survive
    got = g.query_linear_batch(qs, 1, None)
    g.set_option("scan_mode", 0)
    ref = g.query_linear_batch(qs, 1, None)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    assert np.array_equal(got[1], want[1])                 # the same multiset of codes: the same best distances


@pytest.mark.parametrize("B", [3, 6, 40, 200])
def test_degenerate_queries_on_a_large_db_do_not_fault(B):
    """Inf / NaN queries over a database big enough for the scan-order paths: every distance is +inf or NaN, nothing is ever
    'better' than the initial +inf, and the tie logic of the permuted exhaustive scan must not chase the 'nothing chosen yet'
    sentinel (it did: out-of-bounds read).  Ordinary rows of the same batch keep their results in every mode."""
    from rii_amd import RiiGpu
    rng = np.random.default_rng(B)
    M, N = 16, 70000
    cw = rng.random((M, 256, 4)).astype(np.float32)
    codes = rng.integers(0, 256, size=(N, M), dtype=np.uint8)
    Q = rng.random((B, M * 4)).astype(np.float32)
    bad = Q.copy()
    bad[0, :] = np.inf
    bad[1, 5] = np.nan
    if B > 4:
        bad[4, :] = -np.inf
    good = [b for b in range(B) if b not in (0, 1, 4)]
    g = RiiGpu(cw, False, simd_arch="avx512")
    g.add_codes(codes, False)
    g.set_option("scan_mode", 0); g.set_option("scan_order", 0)
    want = g.query_linear_batch(Q, 1, None)
    for mode, order, fmb in ((0, 1, 33), (0, 0, 33), (1, 1, 0), (1, 0, 0), (1, 1, 33)):
        g.set_option("scan_mode", mode); g.set_option("scan_order", order); g.set_option("fast_min_batch", fmb)
        for rep in range(3):
            ids, d = g.query_linear_batch(bad, 1, None)
            assert np.array_equal(ids[good], want[0][good]) and np.array_equal(d[good], want[1][good]), (mode, order, fmb)
        ids, d = g.query_linear_batch(bad, 7, None)
        assert ids.shape == (B, 7)
