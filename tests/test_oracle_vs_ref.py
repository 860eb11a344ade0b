"""Pins the CPU oracle (oracle/rii_oracle.c) bit-for-bit against the real reference build (oracle/_ref).

Runs only where oracle/_ref exists (this container, or a box the .so travelled to).  Everything here is
CPU-only.  Reference entry points: src/main.cpp:12-54.
"""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import make_problem, assert_same_result, ref_with_state, near_tie_assignment_problem

E = np.array([], np.int64)

SHAPES = [  # (M, Ks, Ds, N, scale)
    (32, 256, 4, 3000, "sift"),     # README / SIFT1M shape
    (16, 256, 6, 3000, "unit"),     # Deep1B shape
    (4, 20, 10, 1000, "unit"),      # tests/test_rii.py:12-13 shape
    (20, 256, 2, 1000, "unit"),     # tests/test_rii.py:147-148 shape
    (8, 256, 16, 2000, "sift"),     # exercises the 8/16-wide lanes
    (3, 7, 37, 500, "unit"),        # ragged everything
]


def _lut_ref(ref, cw, q):
    M, Ks, Ds = cw.shape
    out = np.empty((M, Ks), np.float32)
    for m in range(M):
        e = ref.RiiCpp(cw[m:m + 1], False)
        e.add_codes(np.arange(Ks, dtype=np.uint8).reshape(-1, 1), False)
        ids, d = e.query_linear(np.ascontiguousarray(q[m * Ds:(m + 1) * Ds]), Ks, E)
        out[m, ids] = np.array(d, np.float32)
    return out


@pytest.mark.parametrize("Ds", list(range(1, 41)) + [64, 128])
def test_lut_bitexact_all_ds(reference, Ds):
    ref, arch, _ = reference
    cw, _, qs = make_problem(Ds, 2, 256, Ds, 8, "sift")
    want = _lut_ref(ref, cw, qs[0])
    got = O.dtable(cw, qs[0], arch)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("dup", [0, 300])
def test_linear_and_ivf_bitexact(reference, shape, dup):
    ref, arch, _ = reference
    M, Ks, Ds, N, scale = shape
    cw, codes, qs = make_problem(7 + dup, M, Ks, Ds, N, scale, dup=dup)
    r = ref.RiiCpp(cw, False)
    o = O.OracleRii(cw, False, simd_arch=arch)
    r.add_codes(codes, False)
    o.add_codes(codes, False)
    rng = np.random.default_rng(5)
    sub = np.sort(rng.choice(N, N // 10, replace=False)).astype(np.int64)
    tiny = np.array([2, 24, 43, 55, 102, 139, 221, 342, 467, 473, 474, 499], np.int64)
    for q in qs[:6]:
        for topk in (1, 10, 100, N):
            assert_same_result(o.query_linear(q, topk, E), r.query_linear(q, topk, E), "linear k=%d" % topk)
        for tids in (sub, tiny, np.arange(N, dtype=np.int64)):
            k = min(10, len(tids))
            assert_same_result(o.query_linear(q, k, tids), r.query_linear(q, k, tids), "linear subset")
    for nlist in (5, 20, 100):
        r.reconfigure(nlist, 5)
        o.reconfigure(nlist, 5)
        assert o.coarse_centers == r.coarse_centers, "coarse centers nlist=%d" % nlist
        assert o.posting_lists == r.posting_lists, "posting lists nlist=%d" % nlist
        L0 = int(np.round(N / nlist))
        for q in qs[:6]:
            for L in (L0, 4 * L0, N):
                for topk in (1, 10):
                    if topk > L:
                        continue
                    for tids in (E, sub, tiny, np.arange(N, dtype=np.int64)):
                        if len(tids) and topk > len(tids):
                            continue
                        assert_same_result(o.query_ivf(q, topk, tids, L), r.query_ivf(q, topk, tids, L),
                                           "ivf nlist=%d L=%d k=%d S=%d" % (nlist, L, topk, len(tids)))


def test_ivf_empty_return_and_unsorted_tail(reference):
    """rii.h:309,324-325: fewer than topk hits in the first w lists => walk the unsorted tail; never reaching
    L => ([], [])."""
    ref, arch, _ = reference
    cw, codes, qs = make_problem(11, 8, 64, 4, 4000, "unit")
    r = ref.RiiCpp(cw, False)
    o = O.OracleRii(cw, False, simd_arch=arch)
    r.add_codes(codes, False); o.add_codes(codes, False)
    r.reconfigure(200, 3); o.reconfigure(200, 3)
    assert o.posting_lists == r.posting_lists
    n_empty = n_tail = 0
    rng = np.random.default_rng(3)
    for trial in range(40):
        S = int(rng.integers(30, 400))
        tids = np.sort(rng.choice(4000, S, replace=False)).astype(np.int64)
        topk = int(rng.integers(1, 25))
        L = int(rng.integers(topk, S + 1))
        q = qs[trial % 16]
        want = r.query_ivf(q, topk, tids, L)
        got = o.query_ivf(q, topk, tids, L)
        assert_same_result(got, want, "trial %d" % trial)
        n_empty += (len(want[0]) == 0)
    # a case that must return empty: L larger than what the subset can ever supply before the lists run out
    tids = np.arange(0, 4000, 40, dtype=np.int64)      # 100 targets
    want = r.query_ivf(qs[0], 5, tids, 100)
    got = o.query_ivf(qs[0], 5, tids, 100)
    assert_same_result(got, want)


def test_add_codes_update_flag(reference):
    ref, arch, _ = reference
    cw, codes, qs = make_problem(21, 16, 256, 6, 2000, "unit")
    r = ref.RiiCpp(cw, False)
    o = O.OracleRii(cw, False, simd_arch=arch)
    r.add_codes(codes[:1500], False); o.add_codes(codes[:1500], False)
    r.reconfigure(30, 5); o.reconfigure(30, 5)
    r.add_codes(codes[1500:], True); o.add_codes(codes[1500:], True)
    assert o.posting_lists == r.posting_lists
    assert o.flattened_codes == r.flattened_codes


@pytest.mark.parametrize("Ds", [1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 15, 16, 17, 20, 24, 31, 32, 33, 40])
def test_assignment_arithmetic_near_ties(reference, Ds):
    ref, arch, _ = reference
    cw, centers, newc = near_tie_assignment_problem(Ds)
    nl = len(centers)
    r = ref_with_state(ref, cw, centers, np.zeros((0, 2), np.uint8), [[] for _ in range(nl)])
    r.add_codes(newc, True)
    o = O.OracleRii(cw, False, simd_arch=arch)
    o.centers = centers
    o._lists = [[] for _ in range(nl)]
    o.add_codes(newc, True)
    assert o.posting_lists == r.posting_lists


@pytest.mark.parametrize("Ds", [1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 15, 16, 17, 24, 31, 32, 40])
def test_assignment_tables_all_ds(reference, Ds):
    """PQk-means symmetric tables depend on how GCC vectorised pqkmeans.cpp:164-173; probe them through
    the observable posting lists on a near-tie-heavy problem (integer-valued codewords)."""
    ref, arch, _ = reference
    cw, codes, _ = make_problem(100 + Ds, 4, 32, Ds, 1500, "sift")
    r = ref.RiiCpp(cw, False)
    o = O.OracleRii(cw, False, simd_arch=arch)
    r.add_codes(codes, False); o.add_codes(codes, False)
    r.reconfigure(50, 4); o.reconfigure(50, 4)
    assert o.coarse_centers == r.coarse_centers
    assert o.posting_lists == r.posting_lists


def test_ivf_empty_return_with_stale_lists(reference):
    """The `vectors not found` return of rii.h:324-325 and the walk over the *unsorted* tail of the coarse
    order (elements past w after std::partial_sort): reachable once codes are appended with update_flag=False
    after a reconfigure, so that the lists cover fewer than L ids."""
    ref, arch, _ = reference
    cw, codes, qs = make_problem(13, 8, 64, 4, 5050, "unit")
    r = ref.RiiCpp(cw, False)
    o = O.OracleRii(cw, False, simd_arch=arch)
    r.add_codes(codes[:50], False); o.add_codes(codes[:50], False)
    r.reconfigure(10, 3); o.reconfigure(10, 3)
    r.add_codes(codes[50:], False); o.add_codes(codes[50:], False)
    n_empty = n_full = 0
    for topk, L in ((20, 100), (20, 40), (3, 30), (12, 50), (1, 51), (6, 49)):
        for q in qs[:8]:
            want = r.query_ivf(q, topk, E, L)
            assert_same_result(o.query_ivf(q, topk, E, L), want, "k=%d L=%d" % (topk, L))
            n_empty += (len(want[0]) == 0)
            n_full += (len(want[0]) > 0)
    assert n_empty > 0 and n_full > 0


def test_pickle_state_crosses_both_ways(reference):
    """f2 (src/main.cpp:35-53): the reference's get-state tuple loads into the oracle, the oracle's into the reference,
    and both sides answer identically afterwards."""
    ref, arch, _ = reference
    cw, codes, qs = make_problem(41, 8, 64, 4, 2500, "unit", dup=200)
    r = ref.RiiCpp(cw, False)
    r.add_codes(codes, False)
    r.reconfigure(25, 4)
    o = O.OracleRii.__new__(O.OracleRii)
    o.arch = arch
    o.__setstate__(r.__getstate__())
    assert o.posting_lists == r.posting_lists and o.coarse_centers == r.coarse_centers
    r2 = ref.RiiCpp.__new__(ref.RiiCpp)
    r2.__setstate__(o.__getstate__())
    for b in range(6):
        for topk, L in ((1, 100), (5, 100), (3, 2500)):
            want = r.query_ivf(qs[b], topk, E, L)
            assert_same_result(o.query_ivf(qs[b], topk, E, L), want, "ref state -> oracle")
            assert_same_result(r2.query_ivf(qs[b], topk, E, L), want, "oracle state -> ref")
        assert_same_result(r2.query_linear(qs[b], 4, E), r.query_linear(qs[b], 4, E), "oracle state -> ref linear")
