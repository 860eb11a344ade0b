"""Round 6 GPU parity tests (through the C ABI): ivf_rot_kernel -- the conflict-free table gather of the one-query inverted-index
block (table [ks][64 columns], lanes skewed in time over rotated 64-row tiles) -- against ivf_fused_kernel and the CPU oracle."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import make_problem, assert_same_result

pytestmark = pytest.mark.gpu
E = np.array([], np.int64)


@pytest.mark.parametrize("M,Ds,scale,nlist", [(64, 2, "sift", 1000), (32, 4, "sift", 1024), (64, 2, "unit", 100), (32, 4, "unit", 64), (64, 4, "sift", 37),
                                              (32, 2, "unit", 1), (64, 2, "sift", 129)])
def test_ivf_rot_kernel_equals_the_direct_gather_and_the_oracle(M, Ds, scale, nlist):
    """ivf_rot_kernel (option ivf_rot = 2: wherever it applies) against ivf_fused_kernel (ivf_rot = 0) and the oracle, bit for bit:
    ragged lists (lengths that are not multiples of the 64-row tile, empty lists when nlist is large), L from a handful (one partial
    tile) over list-boundary cuts to N (every list: the stop rule's "all w lists walked" arm and the tail walk -> flagged), integer-valued
    data with duplicated rows (exactly tied distances: first minimum in traversal order; tied coarse distances -> flagged queries, replayed
    by the block after it rebuilt the plain table, and -- ivf_inline_exact = 0 -- by the flag-gated exact kernels), every query flagged,
    one lane of work (nlist = 1), 1 .. 16 tiles of centres."""
    from rii_amd import RiiGpu
    N = 30011
    cw, codes, qs = make_problem(1200 + M + nlist + Ds, M, 256, Ds, N, scale, dup=2000 if scale == "sift" else 0)
    rng = np.random.default_rng(6)
    Q = np.concatenate([qs, rng.permutation(qs.reshape(-1)).reshape(qs.shape), qs * 0.5, qs[:5] + 1.0]).astype(np.float32)      # 53 queries
    if scale == "sift":
        Q = np.round(Q)
    o = O.OracleRii(cw, False, simd_arch="avx512")
    o.add_codes(codes, False)
    o.reconfigure(nlist, 2)
    g = RiiGpu(cw, False, simd_arch="avx512")
    g.add_codes(codes, False)
    g.set_coarse_centers(np.array(o.coarse_centers, np.uint8))
    assert g.posting_lists == o.posting_lists
    g.set_option("ivf_quad", 0)
    n0 = g.get_option("ivf_rot_launches")
    L0 = max(1, N // nlist)
    for B in (53, 7):
        for L in (L0, max(1, L0 // 3), min(N, 7 * L0 + 13), min(N, 20 * L0), 5, N):
            want = [o.query_ivf(Q[b], 1, E, L) for b in range(B)]
            for rot, inline, force in ((2, 1, 0), (0, 1, 0), (2, 0, 0), (2, 1, 1), (2, 0, 1)):
                g.set_option("ivf_rot", rot)
                g.set_option("ivf_inline_exact", inline)
                g.set_option("ivf_force_exact", force)
                gi, gd, gc = g.query_ivf_batch(Q[:B], 1, E, L)
                for b in range(B):
                    n = int(gc[b])
                    assert_same_result((gi[b, :n], gd[b, :n]), want[b], "ivf rot=%d inline=%d force=%d M=%d Ds=%d nlist=%d B=%d L=%d b=%d"
                                       % (rot, inline, force, M, Ds, nlist, B, L, b))
    # every ivf_rot = 2 call above whose w = min(nlist, round(L nlist / N) + 3) is within the kernel's 32 picks ran ivf_rot_kernel
    ws = [min(nlist, int(np.round(L * nlist / N)) + 3) for L in (L0, max(1, L0 // 3), min(N, 7 * L0 + 13), min(N, 20 * L0), 5, N)]
    # (the seven-query calls may take the host-flag route of small host-pointer calls, which keeps the one-query kernel)
    assert g.get_option("ivf_rot_launches") - n0 >= 4 * sum(1 for w in ws if w <= 32)
    for k, v in (("ivf_rot", 1), ("ivf_inline_exact", 1), ("ivf_force_exact", 0), ("ivf_quad", 1)):
        g.set_option(k, v)


def test_ivf_rot_copies_follow_the_lists():
    """The rotated tile copies are rebuilt when the lists change: add_codes with update_flag, a second reconfigure, set_posting_lists --
    queries in between keep equal to the oracle's (and to the direct gather's)."""
    from rii_amd import RiiGpu
    import bench
    M, Ds, N = 64, 2, 20000
    cw, codes, qs = make_problem(77, M, 256, Ds, N + 3000, "unit")
    qs = np.concatenate([qs, qs * 0.5, qs[::-1] * 0.25 + 0.1]).astype(np.float32)        # 48 queries: past the small-call route, which keeps the one-query kernel
    o = O.OracleRii(cw, False, simd_arch="avx512")
    g = RiiGpu(cw, False, simd_arch="avx512")
    g.set_option("ivf_rot", 2)
    g.set_option("ivf_quad", 0)
    o.add_codes(codes[:N], False)
    g.add_codes(codes[:N], False)

    def check(L, what):
        gi, gd, gc = g.query_ivf_batch(qs, 1, E, L)
        for b in range(len(qs)):
            n = int(gc[b])
            assert_same_result((gi[b, :n], gd[b, :n]), o.query_ivf(qs[b], 1, E, L), what + " b=%d" % b)

    o.reconfigure(50, 2)
    g.set_coarse_centers(np.array(o.coarse_centers, np.uint8))
    check(2500, "after the first configuration")
    o.add_codes(codes[N:], True)
    g.add_codes(codes[N:], True)
    assert g.posting_lists == o.posting_lists
    check(2500, "after add_codes(update_flag)")
    o.reconfigure(200, 2)
    g.set_coarse_centers(np.array(o.coarse_centers, np.uint8))
    check(3000, "after the second configuration")
    cen = np.random.default_rng(4).integers(0, 256, size=(141, M), dtype=np.uint8)
    off, ids = bench.modulo_lists(N + 3000, 141)
    g.set_posting_lists(cen, off, ids)
    o.set_csr(cen, off, ids)
    check(2100, "after set_posting_lists")
    assert g.get_option("ivf_rot_launches") == 4


def _sharded_check(idx, o, Q, topk, L, what, tids=None):
    import torch
    gi, gd, gc = idx.query_ivf_batch(torch.from_numpy(Q).cuda(), topk, tids, L)
    gi, gd, gc = gi.cpu().numpy(), gd.cpu().numpy(), gc.cpu().numpy()
    for b in range(Q.shape[0]):
        n = int(gc[b])
        assert_same_result((gi[b, :n], gd[b, :n]), o.query_ivf(Q[b], topk, E if tids is None else tids, L), "%s k=%d L=%d b=%d" % (what, topk, L, b))


@pytest.mark.parametrize("arch", ["sse", "avx", "avx512"])
@pytest.mark.parametrize("M,Ds,nlist,scale", [(16, 6, 2000, "unit"), (16, 8, 2000, "sift"), (32, 4, 600, "sift"), (32, 2, 600, "unit"), (16, 4, 2500, "sift"),
                                              (32, 6, 500, "unit")])
def test_shard_coarse_prepass_equals_the_walk_kernels_own_coarse_phase_and_the_oracle(M, Ds, nlist, scale, arch):
    """shard_coarse_quad_kernel (round 6: the coarse phase of the database-sharded inverted index as a pre-pass, four queries per
    block, tables interleaved [m][ks][query], picks by the fast selection) + ivf_shard_any_kernel<PRE> against the walk kernel doing its
    own coarse phase (option shard_pre = 0) and against the oracle, ids and distance bits: every even Ds of the pre-pass in the three
    SIMD orders (src/distance.h:117-252: the low bits of the table), batches that end inside a four-query block (53, 7, 1 queries),
    top-1 and top-k (the register path and the selection buffer of the walk kernel), integer-valued data with duplicated rows (exactly
    tied coarse distances among the w + 1 picks -> `ok` = 0 -> the query's own block scores the centres and replays std::partial_sort),
    every query through that route (shard_force_replay), L from a handful to what w = 7 allows."""
    if Ds <= 4 and arch != "avx512":
        pytest.skip("up to four floats the three SIMD orders coincide")
    from rii_amd import RiiGpu
    from rii_amd import dist as rd
    N = 30011
    cw, codes, qs = make_problem(1600 + M + nlist + Ds, M, 256, Ds, N, scale, dup=3000 if scale == "sift" else 0)
    rng = np.random.default_rng(61)
    Q = np.concatenate([qs, rng.permutation(qs.reshape(-1)).reshape(qs.shape), qs * 0.5, qs[:5] + 1.0]).astype(np.float32)      # 53 queries
    cen = np.ascontiguousarray(codes[rng.integers(0, N, nlist)])
    if scale == "sift":
        Q = np.round(Q)
        cen[rng.integers(0, nlist, nlist // 3)] = cen[rng.integers(0, nlist, nlist // 3)]       # duplicated centres: tied coarse distances
    o = O.OracleRii(cw, False, simd_arch=arch)
    o.add_codes(codes, False)
    o.set_coarse_centers(cen)
    g = RiiGpu(cw, False, simd_arch=arch)
    g.add_codes(codes, False)
    g.set_coarse_centers(cen)
    assert g.posting_lists == o.posting_lists
    idx = rd.DbShardedIndex(g, 0, N)
    L0 = max(1, N // nlist)
    try:
        n0 = g.get_option("shard_pre_launches")
        ncall = 0
        for B in (53, 7, 1):
            for topk, L in ((1, L0), (1, 4 * L0), (3, 2 * L0), (1, 5), (10, 3 * L0)):
                for pre, force in ((2, 0), (0, 0), (2, 1)):
                    g.set_option("shard_pre", pre)
                    g.set_option("shard_force_replay", force)
                    _sharded_check(idx, o, Q[:B], topk, L, "pre=%d force=%d M=%d Ds=%d nlist=%d B=%d" % (pre, force, M, Ds, nlist, B))
                    ncall += pre == 2
        assert g.get_option("shard_pre_launches") - n0 >= ncall          # the pre-pass really ran wherever it was asked for
        # subset search behind the pre-pass: filtered lists (no posting-order rows: the walk kernel gathers by id), w from |S|
        sub = np.sort(rng.choice(N, N // 2, replace=False)).astype(np.int64)
        for pre in (2, 0):
            g.set_option("shard_pre", pre)
            for topk, L in ((1, L0), (3, L0 // 2 + 1)):
                _sharded_check(idx, o, Q[:21], topk, L, "subset pre=%d M=%d Ds=%d nlist=%d" % (pre, M, Ds, nlist), sub)
    finally:
        g.set_option("shard_pre", 1)
        g.set_option("shard_force_replay", 0)


def test_shard_coarse_prepass_stale_lists_and_the_tail_walk():
    """Stale lists (codes added after the lists were built: most lists empty): the walk leaves the first w lists, whose order only
    std::partial_sort's replay knows -- the pre-pass's picks are dropped, the query's block scores the centres itself and replays."""
    from rii_amd import RiiGpu
    from rii_amd import dist as rd
    N, nlist, M, Ds = 20011, 2000, 16, 6
    cw, codes, qs = make_problem(1777, M, 256, Ds, N, "unit")
    cen = np.ascontiguousarray(codes[np.random.default_rng(5).integers(0, N, nlist)])
    n9 = N // 9
    g = RiiGpu(cw, False, simd_arch="avx512")
    g.add_codes(codes[:n9], False)
    g.set_coarse_centers(cen)
    g.add_codes(codes[n9:], False)
    o = O.OracleRii(cw, False, simd_arch="avx512")
    o.add_codes(codes[:n9], False)
    o.set_coarse_centers(cen)
    o.add_codes(codes[n9:], False)
    assert g.posting_lists == o.posting_lists
    idx = rd.DbShardedIndex(g, 0, N)
    try:
        for pre in (2, 0):
            g.set_option("shard_pre", pre)
            for topk, L in ((1, 3), (1, 30), (1, 2), (2, 20), (5, 40)):
                _sharded_check(idx, o, qs, topk, L, "stale pre=%d" % pre)
    finally:
        g.set_option("shard_pre", 1)


@pytest.mark.parametrize("M,Ds", [(16, 6), (32, 6), (16, 10), (32, 2)])
def test_generic_quantiser_levels_and_growing_candidate_buffers(M, Ds):
    """Round 6: the byte tables of the shapes the fused table kernel does not serve (any Ds: the Deep1B shape M = 16, Ds = 6) come from
    the generic quantiser, now at 63 / 127 / 255 levels for the matrix-core scans (option generic_table_levels; 255 = signed bytes, the
    scan's accumulators biased by 128 M).  The slack is proven from the residuals of the levels actually stored, so every setting
    must give the reference's rows -- on tight clusters (thousands of codes within the slack of the minimum: long candidate lists),
    with buffers that overflow (cand_cap forced small: the exhaustive fallback) and with the buffers the engine grows by itself
    from the longest list it has seen.  Linear top-1 and top-k against the oracle, ids and distance bits."""
    from rii_amd import RiiGpu
    N = 40000
    rng = np.random.default_rng(900 + M + Ds)
    cw = rng.random((M, 256, Ds)).astype(np.float32)
    # tight clusters: 40 distinct codes, the rest differ from one of them in one or two subspaces
    base = rng.integers(0, 256, size=(40, M), dtype=np.uint8)
    codes = base[rng.integers(0, 40, N)].copy()
    for _ in range(2):
        rows = rng.integers(0, N, N // 2)
        codes[rows, rng.integers(0, M, N // 2)] = rng.integers(0, 256, N // 2, dtype=np.uint8)
    qs = rng.random((40, M * Ds)).astype(np.float32)
    o = O.OracleRii(cw, False, simd_arch="avx512")
    o.add_codes(codes, False)
    want1 = [o.query_linear(qs[b], 1, E) for b in range(len(qs))]
    want5 = [o.query_linear(qs[b], 5, E) for b in range(8)]
    g = RiiGpu(cw, False, simd_arch="avx512")
    g.add_codes(codes, False)
    g.set_option("fast_min_batch", 0)
    for levels in (0, 63, 127, 255):
        g.set_option("generic_table_levels", levels)
        for rep in range(3):                                  # (the second and third call run with the buffers the first one asked for)
            ids, d = g.query_linear_batch(qs, 1, E)
            for b in range(len(qs)):
                assert_same_result((ids[b], d[b]), want1[b], "generic levels=%d rep=%d M=%d Ds=%d b=%d" % (levels, rep, M, Ds, b))
        ids, d = g.query_linear_batch(qs[:8], 5, E)
        for b in range(8):
            assert_same_result((ids[b], d[b]), want5[b], "generic levels=%d top-5 M=%d Ds=%d b=%d" % (levels, M, Ds, b))
    cand = g.get_option("cand_max")
    g.set_option("cand_cap", 64)                              # every list overflows: the exhaustive fallback
    ids, d = g.query_linear_batch(qs, 1, E)
    for b in range(len(qs)):
        assert_same_result((ids[b], d[b]), want1[b], "generic overflow M=%d Ds=%d b=%d (longest list %d)" % (M, Ds, b, cand))


def test_shard_rows_beyond_8193_with_a_small_L_take_the_collect_form():
    """rows >= L is documented as always served (every owned candidate at the slot of its position), but a request like
    L = 100, rows = 10000 was rejected -- past the 8192-key kernel's row count, the collect form never tried.  Now: the first L slots
    equal what rows = L returns, the rest is padding."""
    import torch
    from rii_amd import RiiGpu
    N, nlist, M, Ds, L, B = 20000, 50, 8, 4, 100, 5
    cw, codes, qs = make_problem(4242, M, 256, Ds, N, "unit")
    o = O.OracleRii(cw, False, simd_arch="avx512")
    o.add_codes(codes, False)
    o.reconfigure(nlist, 2)
    g = RiiGpu(cw, False, simd_arch="avx512")
    g.add_codes(codes, False)
    g.set_coarse_centers(np.array(o.coarse_centers, np.uint8))
    dev = torch.device("cuda", 0)
    q = torch.from_numpy(qs[:B]).to(dev)
    glen = torch.tensor([len(l) for l in g.posting_lists], dtype=torch.int32, device=dev)
    got = {}
    for rows in (L, 10000):
        oi = torch.full((B, rows), -7, dtype=torch.int64, device=dev); od = torch.zeros((B, rows), dtype=torch.float32, device=dev)
        op = torch.zeros((B, rows), dtype=torch.int32, device=dev); on = torch.zeros((B,), dtype=torch.int32, device=dev)
        oc = torch.zeros((B,), dtype=torch.int64, device=dev)
        g.query_ivf_shard_dev(q.data_ptr(), B, 1, 0, 0, 0, L, N, glen.data_ptr(), 1, 0, oi.data_ptr(), od.data_ptr(), op.data_ptr(), on.data_ptr(),
                              oc.data_ptr(), 0, rows)
        torch.cuda.synchronize()
        got[rows] = (oi.cpu().numpy(), od.cpu().numpy(), op.cpu().numpy(), on.cpu().numpy())
    a, b = got[L], got[10000]
    assert np.array_equal(a[3], b[3]) and int(a[3].min()) == L                      # one rank owns every candidate
    for bq in range(B):                    # (the 8192-key kernel hands its rows over sorted by (distance, position), the collect form by position)
        oa = np.argsort(a[2][bq])
        assert np.array_equal(a[2][bq][oa], np.arange(L)) and np.array_equal(b[2][bq, :L], np.arange(L))
        assert np.array_equal(a[0][bq][oa], b[0][bq, :L]) and np.array_equal(a[1][bq][oa].view(np.uint32), b[1][bq, :L].view(np.uint32))
    assert bool((b[0][:, L:] == -1).all()) and bool(np.isinf(b[1][:, L:]).all())
    for bq in range(B):                                                                # ... and the L candidates hold the oracle's top-1
        wi, wd = o.query_ivf(qs[bq], 1, E, L)
        j = int(np.lexsort((a[2][bq], a[1][bq]))[0])
        assert int(a[0][bq, j]) == wi[0] and np.float32(a[1][bq, j]).view(np.uint32) == np.float32(wd[0]).view(np.uint32)


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_sharded_ivf_prepass_against_oracle(seed):
    """Differential fuzz of the sharded inverted index with the coarse pre-pass forced on (shard_pre = 2) and off: random shape within
    the pre-pass's reach (M = 16 / 32, even Ds <= 8, a SIMD order, lists by the hundreds or thousands so that the coarse order lives in
    global scratch), random scale (integer-valued data with duplicated rows and centres: ties), random batch size, L (w <= 7 or
    beyond: the pre-pass then steps aside), topk, target ids or none -- ids and distance bits against the oracle."""
    from rii_amd import RiiGpu
    from rii_amd import dist as rd
    rng = np.random.default_rng(7000 + seed)
    M = int(rng.choice([16, 32]))
    Ds = int(rng.choice([2, 4, 6, 8]))
    arch = str(rng.choice(["avx512", "avx", "sse"]))
    scale = str(rng.choice(["unit", "sift"]))
    N = int(rng.integers(8000, 30000))
    nlist = int(rng.integers(2000, 3000)) if M == 16 else int(rng.integers(450, 900))
    cw, codes, qs = make_problem(7100 + seed, M, 256, Ds, N, scale, dup=(N // 5 if scale == "sift" else 0))
    cen = np.ascontiguousarray(codes[rng.integers(0, N, nlist)])
    if scale == "sift":
        cen[rng.integers(0, nlist, nlist // 4)] = cen[rng.integers(0, nlist, nlist // 4)]
    Q = np.concatenate([qs, qs[::-1] * 0.5, rng.permutation(qs.reshape(-1)).reshape(qs.shape)]).astype(np.float32)
    if scale == "sift":
        Q = np.round(Q)
    o = O.OracleRii(cw, False, simd_arch=arch)
    o.add_codes(codes, False)
    o.set_coarse_centers(cen)
    g = RiiGpu(cw, False, simd_arch=arch)
    g.add_codes(codes, False)
    g.set_coarse_centers(cen)
    idx = rd.DbShardedIndex(g, 0, N)
    L0 = max(1, N // nlist)
    try:
        for trial in range(6):
            B = int(rng.choice([1, 3, 4, 5, 17, 48]))
            tids = np.sort(rng.choice(N, int(rng.integers(N // 3, N)), replace=False)).astype(np.int64) if rng.random() < 0.35 else None
            pool = N if tids is None else len(tids)
            L = int(rng.choice([1, L0, 2 * L0 + 1, 4 * L0, 9 * L0, 40 * L0]))
            L = max(1, min(L, pool))
            topk = 1 if rng.random() < 0.5 else int(rng.integers(1, min(L, 12) + 1))
            for pre in (2, 0):
                g.set_option("shard_pre", pre)
                _sharded_check(idx, o, Q[:B], topk, L, "fuzz seed=%d trial=%d pre=%d M=%d Ds=%d %s %s nlist=%d N=%d B=%d" %
                               (seed, trial, pre, M, Ds, arch, scale, nlist, N, B), tids)
        assert g.get_option("shard_pre_launches") > 0                      # (some trial of every seed is within the pre-pass's reach)
    finally:
        g.set_option("shard_pre", 1)


def test_ivf_top1_merge_for_many_fake_ranks():
    """rii_ivf_merge_top1_hdr_dev (round 6: merge + finish of the sharded inverted index's top-1 batch in one thread-per-query launch) on
    records built by hand for G = 1 ... 70 fake ranks: two rows per query and rank (position, global id, distance), padding rows,
    exactly tied distances across ranks (the smaller POSITION wins: first minimum in traversal order), queries the walk did not find
    (count 0 -> -1 / +inf), and a non-zero status in one header poisoning the batch (ids -2, NaN, counts -1, bit 1 of the word)."""
    import torch
    from rii_amd import core
    rng = np.random.default_rng(66)
    for G, B in ((1, 7), (2, 5), (3, 300), (8, 1000), (70, 9)):
        k = 2
        L = 5000
        pos = np.full((G, B, k), np.iinfo(np.int32).max, np.int64)
        gid = np.full((G, B, k), -1, np.int64)
        d = np.full((G, B, k), np.inf, np.float32)
        for b in range(B):
            n_own = rng.integers(0, 3, G)                                         # rows each rank owns for this query (0 .. 2)
            ps = rng.choice(L, int(n_own.sum()), replace=False) if n_own.sum() else np.zeros(0, np.int64)
            dd = rng.integers(0, 6, int(n_own.sum())).astype(np.float32)         # many exact ties
            at = 0
            for g in range(G):
                rows = sorted(zip(dd[at:at + n_own[g]], ps[at:at + n_own[g]]))
                for j, (dv, pv) in enumerate(rows):
                    pos[g, b, j] = pv; d[g, b, j] = dv; gid[g, b, j] = 1000 * pv + g
                at += n_own[g]
        cnt = (rng.random(B) < 0.85).astype(np.int64)
        rec = core.merge_hdr_record_bytes(B, k, 1)
        buf = torch.zeros((G, rec), dtype=torch.uint8)
        n = B * k
        for g in range(G):
            buf[g, 16:16 + n * 8] = torch.from_numpy(np.ascontiguousarray(pos[g])).reshape(-1).view(torch.uint8)
            buf[g, 16 + n * 8:16 + n * 16] = torch.from_numpy(np.ascontiguousarray(gid[g])).reshape(-1).view(torch.uint8)
            buf[g, 16 + n * 16:16 + n * 20] = torch.from_numpy(np.ascontiguousarray(d[g])).reshape(-1).view(torch.uint8)
        dbuf = buf.cuda()
        dcnt = torch.from_numpy(cnt).cuda()
        oi = torch.empty(B, dtype=torch.int64, device="cuda"); od = torch.empty(B, dtype=torch.float32, device="cuda")
        oc = torch.empty(B, dtype=torch.int64, device="cuda"); anyf = torch.zeros(1, dtype=torch.int32, device="cuda")
        core.ivf_merge_top1_hdr_dev(dbuf.data_ptr(), G, B, dcnt.data_ptr(), oi.data_ptr(), od.data_ptr(), oc.data_ptr(), anyf.data_ptr())
        torch.cuda.synchronize()
        for b in range(B):
            rows = sorted((float(d[g, b, j]), int(pos[g, b, j]), int(gid[g, b, j])) for g in range(G) for j in range(k) if np.isfinite(d[g, b, j]))
            if cnt[b] > 0 and rows:
                assert (int(oi[b]), float(od[b]), int(oc[b])) == (rows[0][2], rows[0][0], int(cnt[b])), (G, B, b)
            else:
                assert int(oi[b]) == -1 and np.isinf(float(od[b])) and int(oc[b]) == int(cnt[b]), (G, B, b)
        assert int(anyf.item()) == 0
        buf[G // 2, 8:12] = torch.tensor([3], dtype=torch.int32).view(torch.uint8)
        dbuf = buf.cuda()
        core.ivf_merge_top1_hdr_dev(dbuf.data_ptr(), G, B, dcnt.data_ptr(), oi.data_ptr(), od.data_ptr(), oc.data_ptr(), anyf.data_ptr())
        torch.cuda.synchronize()
        assert bool((oi == -2).all()) and bool(torch.isnan(od).all()) and bool((oc == -1).all()) and int(anyf.item()) & 2
