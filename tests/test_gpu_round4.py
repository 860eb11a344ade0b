"""Round 4, GPU: the top-1 re-rank folded into the scan launch (fs_tail_rerank), the one-query-per-block table kernel of small
batches, and the device-queries -> host-rows entry points (rii_query_*_dev_to_host).  Everything is compared bit for bit with
the path it replaces and with the CPU oracle; the host-delivery paths are hammered with changing batches so that a stale row
or a flag raised early would show."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import make_problem, assert_same_result

pytestmark = pytest.mark.gpu
E = np.array([], np.int64)


def _engine(M, Ds, N, seed, dup=0):
    from rii_amd import RiiGpu
    rng = np.random.default_rng(seed)
    cw = rng.random((M, 256, Ds)).astype(np.float32)
    codes = rng.integers(0, 256, size=(N, M), dtype=np.uint8)
    if dup:
        codes[rng.integers(0, N, dup)] = codes[rng.integers(0, N, dup)]
    g = RiiGpu(cw, False, simd_arch="avx512")
    g.add_codes(codes, False)
    return g, cw, codes, rng


@pytest.mark.parametrize("M,Ds", [(32, 4), (16, 4), (32, 2), (16, 2)])
def test_fused_rerank_equals_the_separate_kernel_and_the_oracle(M, Ds):
    """option fused_rerank = 1 (default): the last chunk-block of every scan tile re-ranks the tile's queries inside the scan
    launch; = 0: rerank_top1_direct_kernel.  Same ids and distances for every batch size (one-query-per-block tables up to 256
    queries, ragged last tiles, an odd number of tiles for the two-tile M = 16 kernel), duplicated codes (exact ties: the
    smallest id wins), target ids, one and many chunks per tile."""
    g, cw, codes, rng = _engine(M, Ds, 90000 + 7, 500 + M + Ds, dup=3000)
    qs = rng.random((600, M * Ds)).astype(np.float32)
    tids = np.sort(rng.choice(g.N, 30000, replace=False)).astype(np.int64)
    o = O.OracleRii(cw, False, simd_arch="avx512")
    o.add_codes(codes, False)
    for B in (33, 100, 128, 257, 600):
        for t in (None, tids):
            for dual in (1, 0):
                for chunks in (0, 1):
                    g.set_option("scan_dual", dual)
                    g.set_option("scan_chunks", chunks)
                    g.set_option("fused_rerank", 1)
                    a = g.query_linear_batch(qs[:B], 1, t)
                    g.set_option("fused_rerank", 0)
                    b = g.query_linear_batch(qs[:B], 1, t)
                    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), (B, dual, chunks)
            for bq in (0, B // 2, B - 1):
                assert_same_result((a[0][bq], a[1][bq]), o.query_linear(qs[bq], 1, E if t is None else t), "fused re-rank B=%d" % B)
    g.set_option("scan_chunks", 0)
    g.set_option("scan_dual", 1)
    g.set_option("fused_rerank", 1)


def test_fused_rerank_overflowed_candidate_buffers():
    """cand_cap forced tiny: the fused tail scans the overflowed queries exhaustively (exact table in LDS), the others from their
    candidate lists -- still exact."""
    g, cw, codes, rng = _engine(32, 4, 40000, 77, dup=0)
    codes2 = codes.copy()
    codes2[5000:9000] = codes2[4999]             # thousands of duplicates of one code: every query near it overflows
    from rii_amd import RiiGpu
    g = RiiGpu(cw, False, simd_arch="avx512")
    g.add_codes(codes2, False)
    qs = rng.random((80, 128)).astype(np.float32)
    g.set_option("scan_mode", 0)
    want = g.query_linear_batch(qs, 1, None)
    g.set_option("scan_mode", 1)
    for cap in (4, 64):
        g.set_option("cand_cap", cap)
        for fr in (1, 0):
            g.set_option("fused_rerank", fr)
            got = g.query_linear_batch(qs, 1, None)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (cap, fr)


def test_small_batch_tables_equal_the_four_query_blocks():
    """B <= 256 builds the byte tables with one query per block (qlut_fused_kernel<.., 1>): the scan over them gives the same
    candidates' answers as over the four-query blocks' tables -- checked through the results of batches on both sides of the
    switch that share their first rows."""
    g, cw, codes, rng = _engine(32, 4, 120000, 31)
    qs = rng.random((300, 128)).astype(np.float32)
    big = g.query_linear_batch(qs, 1, None)                 # 300 queries: four-query blocks
    for B in (40, 129, 256):
        small = g.query_linear_batch(qs[:B], 1, None)       # one query per block
        assert np.array_equal(small[0], big[0][:B]) and np.array_equal(small[1], big[1][:B]), B
    for lv in (63, 255):
        g.set_option("table_levels", lv)
        small = g.query_linear_batch(qs[:100], 1, None)
        assert np.array_equal(small[0], big[0][:100]) and np.array_equal(small[1], big[1][:100]), lv


@pytest.mark.parametrize("M,Ds", [(32, 4), (16, 6), (8, 16)])
def test_dev_to_host_rows_equal_the_device_call(M, Ds):
    """rii_query_linear_dev_to_host / rii_query_ivf_dev_to_host: queries in HBM, rows in the caller's host arrays when the call
    returns.  Top-1 of the fused shapes takes the flag path (rows written by the scan's tail, one flag per tile), everything else
    one synchronisation: all must equal the plain device call.  Batches change from call to call (a stale row would show)."""
    import torch
    g, cw, codes, rng = _engine(M, Ds, 50000, 900 + M)
    g.reconfigure(64, 3)
    dev = torch.device("cuda:0")
    st = torch.cuda.Stream(dev)
    D = M * Ds
    for it in range(12):
        g.set_option("fused_rerank", (it // 2) & 1)         # rows by the scan's tail + one flag per tile / by the re-rank kernel + a synchronisation
        B = int(rng.integers(1, 400))
        topk = [1, 1, 1, 3, 20][it % 5]
        qs = rng.random((B, D)).astype(np.float32)
        q = torch.from_numpy(qs).to(dev)
        want = g.query_linear_batch(qs, topk, None)
        ids = np.full((B, topk), -7, np.int64)
        d = np.full((B, topk), -7, np.float32)
        g.query_linear_dev_to_host(q.data_ptr(), B, topk, 0, 0, ids, d, st.cuda_stream if it % 2 else 0)
        assert np.array_equal(ids, want[0]) and np.array_equal(d.view(np.uint32), want[1].view(np.uint32)), (it, B, topk)
        if it % 3 == 0:
            t = np.sort(rng.choice(g.N, 5000, replace=False)).astype(np.int64)
            td = torch.from_numpy(t).to(dev)
            want = g.query_linear_batch(qs, topk, t)
            g.query_linear_dev_to_host(q.data_ptr(), B, topk, td.data_ptr(), t.size, ids, d, 0)
            assert np.array_equal(ids, want[0]) and np.array_equal(d.view(np.uint32), want[1].view(np.uint32)), (it, B, topk, "tids")
            L = 900
            wi, wd, wc = g.query_ivf_batch(qs, topk, None, L)
            cnt = np.full(B, -7, np.int64)
            g.query_ivf_dev_to_host(q.data_ptr(), B, topk, 0, 0, L, ids, d, cnt, 0)
            assert np.array_equal(cnt, wc)
            for b in range(B):
                n = int(cnt[b])
                assert np.array_equal(ids[b, :n], wi[b, :n]) and np.array_equal(d[b, :n], wd[b, :n])


def test_dev_to_host_flag_path_under_load_every_word():
    """The flag path again, the way the guide asks for hand-offs to be tested: many back-to-back calls with DIFFERENT answers,
    uneven load (batch sizes from one tile to many), every returned word compared.  Also through the host-pointer batch call
    with and without host_zero_copy (queries read by the kernels from the pinned block)."""
    import torch
    g, cw, codes, rng = _engine(32, 4, 200000, 4242, dup=500)
    dev = torch.device("cuda:0")
    pool = rng.random((4096, 128)).astype(np.float32)
    qd = torch.from_numpy(pool).to(dev)
    g.set_option("fused_rerank", 0)
    ref_i, ref_d = g.query_linear_batch(pool, 1, None)
    g.set_option("fused_rerank", 1)
    for it in range(60):
        B = int(rng.choice([33, 48, 64, 130, 512, 1024]))
        s = int(rng.integers(0, 4096 - B))
        ids = np.empty((B, 1), np.int64)
        d = np.empty((B, 1), np.float32)
        g.query_linear_dev_to_host(qd.data_ptr() + s * 128 * 4, B, 1, 0, 0, ids, d, 0)
        assert np.array_equal(ids, ref_i[s:s + B]) and np.array_equal(d, ref_d[s:s + B]), (it, B, s)
    for zc in (0, 2):
        g.set_option("host_zero_copy", zc)
        for it in range(20):
            B = int(rng.choice([40, 100, 700, 1024, 2000]))
            s = int(rng.integers(0, 4096 - B))
            got = g.query_linear_batch(pool[s:s + B], 1, None)
            assert np.array_equal(got[0], ref_i[s:s + B]) and np.array_equal(got[1], ref_d[s:s + B]), (zc, it, B, s)
    g.set_option("host_zero_copy", 1)
    g.set_option("host_spin", 0)                      # the copy + synchronise form of the same calls
    got = g.query_linear_batch(pool[:300], 1, None)
    assert np.array_equal(got[0], ref_i[:300]) and np.array_equal(got[1], ref_d[:300])


def test_db_sharded_device_path_beyond_the_merge_kernel_limits():
    """DbShardedIndex on device tensors: G * (topk + 1) rows above the merge kernel's LDS sort (MERGE_MAX_KEYS = 8192) fall back to the
    torch merge (the device path used to raise there) -- same rows as the single engine, exact ties replayed.  The limit
    is lowered for the test so that ordinary sizes reach the fallback."""
    import torch
    from rii_amd.dist import DbShardedIndex
    g, cw, codes, rng = _engine(8, 4, 12000, 5, dup=500)
    idx = DbShardedIndex(g, 0, g.N)
    qs = rng.random((9, 32)).astype(np.float32)
    Q = torch.from_numpy(qs).cuda()
    for limit in (8192, 16):
        idx.MERGE_MAX_KEYS = limit
        for topk in (1, 5, 50):
            ids, d = idx.query_linear_batch(Q, topk)
            want = g.query_linear_batch(qs, topk, None)
            assert np.array_equal(ids.cpu().numpy(), want[0]) and np.array_equal(d.cpu().numpy(), want[1]), (limit, topk)
            assert idx.last_tie_flags.shape[0] == 9


def test_query_shard_unpack_and_top1_merge_kernels_for_many_ranks():
    """The exchange kernels of comm.hip for G > 1 (the one-GPU box only ever runs G = 1 through RCCL): records of G fake ranks built
    with numpy -> rii_qshard_unpack_dev gives the rows in batch order for even and ragged splits, with and without counts;
    rii_merge_topk_ex_dev with one row per rank (merge_top1_kernel) equals the general sort kernel, ties across ranks broken by the
    GLOBAL id, padding rows ignored."""
    import ctypes
    import torch
    from rii_amd import core
    L = core._lib()
    rng = np.random.default_rng(11)
    dev = torch.device("cuda:0")
    for B, G, k, counts in ((1024, 8, 1, 0), (1000, 8, 3, 1), (5, 8, 2, 1), (129, 4, 10, 0), (7, 1, 4, 1)):
        rec = int(L.rii_qshard_record_bytes(B, G, k, counts))
        nmax = (B + G - 1) // G
        want_i = rng.integers(0, 1 << 40, size=(B, k)).astype(np.int64)
        want_d = rng.random((B, k)).astype(np.float32)
        want_c = rng.integers(0, k + 1, size=B).astype(np.int64)
        buf = np.zeros((G, rec), np.uint8)
        for r in range(G):
            s, e = int(L.rii_qshard_begin(B, G, r)), int(L.rii_qshard_begin(B, G, r + 1))
            n = e - s
            buf[r, :n * k * 8] = want_i[s:e].reshape(-1).view(np.uint8)
            off = nmax * k * 8
            if counts:
                buf[r, off:off + n * 8] = want_c[s:e].view(np.uint8)
                off += nmax * 8
            buf[r, off:off + n * k * 4] = want_d[s:e].reshape(-1).view(np.uint8)
        g = torch.from_numpy(buf).to(dev)
        oi = torch.empty((B, k), dtype=torch.int64, device=dev)
        od = torch.empty((B, k), dtype=torch.float32, device=dev)
        oc = torch.empty((B,), dtype=torch.int64, device=dev)
        core._check(L.rii_qshard_unpack_dev(g.data_ptr(), B, G, k, counts, oi.data_ptr(), od.data_ptr(), oc.data_ptr() if counts else None, None))
        torch.cuda.synchronize()
        assert np.array_equal(oi.cpu().numpy(), want_i) and np.array_equal(od.cpu().numpy(), want_d), (B, G, k)
        if counts:
            assert np.array_equal(oc.cpu().numpy(), want_c)
    for B, G in ((300, 8), (17, 64), (5, 2)):
        rec = core.merge_record_bytes(B, 1)
        buf = np.zeros((G, rec), np.uint8)
        ids = rng.integers(0, 1000, size=(G, B)).astype(np.int64)
        d = rng.integers(0, 4, size=(G, B)).astype(np.float32)            # many exact ties across the ranks
        pad = rng.random((G, B)) < 0.2
        ids[pad] = np.iinfo(np.int64).max // 2
        d[pad] = np.inf
        for r in range(G):
            buf[r, :B * 8] = ids[r].view(np.uint8)
            buf[r, B * 8:B * 12] = d[r].view(np.uint8)
        offs = [1000 * r for r in range(G)]
        g = torch.from_numpy(buf).to(dev)
        a_i = torch.empty((B, 1), dtype=torch.int64, device=dev); a_d = torch.empty((B, 1), dtype=torch.float32, device=dev)
        b_i = torch.empty((B, 1), dtype=torch.int64, device=dev); b_d = torch.empty((B, 1), dtype=torch.float32, device=dev)
        tie = torch.zeros(B, dtype=torch.int32, device=dev)
        core.merge_topk_ex_dev(g.data_ptr(), G, B, 1, 1, offs, a_i.data_ptr(), a_d.data_ptr())                      # merge_top1_kernel
        core.merge_topk_ex_dev(g.data_ptr(), G, B, 1, 1, offs, b_i.data_ptr(), b_d.data_ptr(), tie_cols=1, d_out_tie=tie.data_ptr())   # the sort kernel
        torch.cuda.synchronize()
        assert torch.equal(a_i, b_i) and torch.equal(a_d, b_d), (B, G)
        gid = np.where(pad, ids, ids + np.array(offs)[:, None])
        for b in range(B):
            j = min(range(G), key=lambda r: (d[r, b], gid[r, b]))
            assert int(a_i[b, 0]) == int(gid[j, b]) and float(a_d[b, 0]) == float(d[j, b])


@pytest.mark.parametrize("M,Ds", [(32, 4), (16, 6)])
def test_ivf_flagged_blocks_redo_their_own_query(M, Ds):
    """option ivf_inline_exact = 1 (default): a block of ivf_fused_kernel that flags its query (exactly tied coarse distances, walk
    into the unsorted tail, ties at the cut) replays std::partial_sort itself; = 0: the flag-gated exact kernels of round 3.  Same
    rows either way -- duplicated centres and codes make the flags common, ivf_force_exact flags every query -- and the oracle's.
    One-query host calls (the fused kernel reads the query from the pinned block) give the batch's rows."""
    from rii_amd import RiiGpu
    rng = np.random.default_rng(60 + M)
    cw = np.round(rng.random((M, 256, Ds)) * 15).astype(np.float32)          # integer-valued tables: exact ties everywhere
    N = 30000
    codes = rng.integers(0, 256, size=(N, M), dtype=np.uint8)
    codes[rng.integers(0, N, 4000)] = codes[rng.integers(0, N, 4000)]
    g = RiiGpu(cw, False, simd_arch="avx512")
    g.add_codes(codes, False)
    g.reconfigure(64, 2)
    cen = g.coarse_centers_array().copy()
    cen[1::7] = cen[0::7][:len(cen[1::7])]                                    # duplicated centres: tied coarse distances
    g.set_coarse_centers(cen)
    o = O.OracleRii(cw, False, simd_arch="avx512")
    o.add_codes(codes, False)
    o.set_coarse_centers(cen) if hasattr(o, "set_coarse_centers") else None
    qs = np.round(rng.random((70, M * Ds)) * 15).astype(np.float32)
    tids = np.sort(rng.choice(N, 3000, replace=False)).astype(np.int64)
    nflag_seen = 0
    for topk in (1, 5, 40):
        for L in (200, 3000, N):
            for t in (None, tids):
                for force in (0, 1):
                    g.set_option("ivf_force_exact", force)
                    res = []
                    for inl in (1, 0):
                        g.set_option("ivf_inline_exact", inl)
                        res.append(g.query_ivf_batch(qs, topk, t, min(L, N)))
                    (ai, ad, ac), (bi, bd, bc) = res
                    assert np.array_equal(ac, bc), (topk, L, force)
                    for b in range(len(qs)):
                        n = int(ac[b])
                        assert np.array_equal(ai[b, :n], bi[b, :n]) and np.array_equal(ad[b, :n].view(np.uint32), bd[b, :n].view(np.uint32)), (topk, L, force, b)
    g.set_option("ivf_force_exact", 0)
    g.set_option("ivf_inline_exact", 1)
    if hasattr(o, "set_coarse_centers"):
        for b in range(0, 70, 9):
            for topk, L in ((1, 200), (5, 3000)):
                gi, gd, gc = g.query_ivf_batch(qs[b:b + 1], topk, None, L)
                assert_same_result((gi[0, :int(gc[0])], gd[0, :int(gc[0])]), o.query_ivf(qs[b], topk, E, L), "ivf inline exact")
    want = g.query_ivf_batch(qs, 3, None, 500)
    for b in range(len(qs)):                       # one query per call: the in-place form of the host call
        ids, d = g.query_ivf(qs[b], 3, E, 500)
        n = int(want[2][b])
        assert ids == want[0][b, :n].tolist() and d == want[1][b, :n].tolist(), b


@pytest.mark.parametrize("M,Ds,scale", [(32, 4, "sift"), (16, 6, "unit"), (8, 16, "sift")])
def test_async_few_queries_take_the_slice_kernel_with_device_side_tie_fallback(M, Ds, scale):
    """rii_query_linear_dev with 1 - 8 queries and topk > 1 on an index too large for the one-block kernel: ONE launch of
    slice_topk_kernel + flag-gated tie kernels (no host decision, the call stays asynchronous).  Integer-valued tables and
    duplicated codes make exactly tied distances common; every row must equal the general path's (option slice_topk = 0), which
    the oracle tests pin."""
    import torch
    from rii_amd import RiiGpu
    cw, codes, _ = make_problem(300 + M, M, 256, Ds, 70000, scale, dup=6000)
    rng = np.random.default_rng(M)
    qs = (np.round(rng.random((8, M * Ds)) * 255) if scale == "sift" else rng.random((8, M * Ds))).astype(np.float32)
    g = RiiGpu(cw, False, simd_arch="avx512")
    g.add_codes(codes, False)
    dev = torch.device("cuda:0")
    tids = np.sort(rng.choice(g.N, 40000, replace=False)).astype(np.int64)
    td = torch.from_numpy(tids).to(dev)
    nties = 0
    for B in (1, 3, 8):
        q = torch.from_numpy(qs[:B]).to(dev)
        for topk in (2, 3, 10, 128):
            for t, tdev in ((None, None), (tids, td)):
                oi = torch.full((B, topk), -5, dtype=torch.int64, device=dev)
                od = torch.full((B, topk), -5, dtype=torch.float32, device=dev)
                g.set_option("slice_topk", 1)
                g.query_linear_dev(q.data_ptr(), B, topk, 0 if t is None else tdev.data_ptr(), 0 if t is None else t.size, oi.data_ptr(), od.data_ptr(), 0)
                g.synchronize()
                g.set_option("slice_topk", 0)
                want = g.query_linear_batch(qs[:B], topk, t)
                assert np.array_equal(oi.cpu().numpy(), want[0]) and np.array_equal(od.cpu().numpy().view(np.uint32), want[1].view(np.uint32)), (B, topk, t is None)
                nties += int((want[1][:, 1:] == want[1][:, :-1]).any())
    g.set_option("slice_topk", 1)
    if scale == "sift":
        assert nties > 0                       # the fallback really ran


def test_dev_to_host_beyond_one_internal_pass():
    """rii_query_linear_dev_to_host / _ivf_ with more queries than one internal pass holds (8192): device buffers + copies + one
    synchronisation -- same rows as the host-pointer call."""
    import torch
    g, cw, codes, rng = _engine(16, 4, 20000, 99)
    g.reconfigure(32, 2)
    B = 9000
    qs = rng.random((B, 64)).astype(np.float32)
    q = torch.from_numpy(qs).cuda()
    want = g.query_linear_batch(qs, 2, None)
    ids, d = np.empty((B, 2), np.int64), np.empty((B, 2), np.float32)
    g.query_linear_dev_to_host(q.data_ptr(), B, 2, 0, 0, ids, d, 0)
    assert np.array_equal(ids, want[0]) and np.array_equal(d, want[1])
    wi, wd, wc = g.query_ivf_batch(qs, 1, None, 700)
    ids1, d1, cnt = np.empty((B, 1), np.int64), np.empty((B, 1), np.float32), np.empty(B, np.int64)
    g.query_ivf_dev_to_host(q.data_ptr(), B, 1, 0, 0, 700, ids1, d1, cnt, 0)
    ok = wc > 0
    assert np.array_equal(cnt, wc) and np.array_equal(ids1[ok], wi[ok]) and np.array_equal(d1[ok], wd[ok])


@pytest.mark.parametrize("M,Ds", [(32, 4), (16, 6), (8, 16)])
def test_ivf_posting_order_code_copy_gives_the_rows_of_the_id_gather(M, Ds):
    """option ivf_list_codes = 1 (default): ivf_fused_kernel reads its candidates' rows from a copy of the codes kept in posting
    order (a target_ids batch: the list filter compacts the rows along with the ids); = 0: rows gathered by id (round 3).  Same rows,
    bit for bit, for top-1 / selection in LDS / the streaming buffer, with and without target ids, before
    and after an append (the copy is rebuilt with the lists), after a reconfigure, and equal to the oracle's."""
    from rii_amd import RiiGpu
    rng = np.random.default_rng(90 + M)
    cw = rng.standard_normal((M, 256, Ds)).astype(np.float32)
    N = 40000
    codes = rng.integers(0, 256, size=(N + 5000, M), dtype=np.uint8)
    g = RiiGpu(cw, False, simd_arch="avx512")
    g.add_codes(codes[:N], False)
    g.reconfigure(100, 2)
    o = O.OracleRii(cw, False, simd_arch="avx512")
    o.add_codes(codes[:N], False)
    qs = rng.standard_normal((96, M * Ds)).astype(np.float32)

    def both(topk, L, t=None):
        res = []
        for lc in (1, 0):
            g.set_option("ivf_list_codes", lc)
            assert g.get_option("ivf_list_codes") == lc
            res.append(g.query_ivf_batch(qs, topk, t, L))
        g.set_option("ivf_list_codes", 1)
        (ai, ad, ac), (bi, bd, bc) = res
        assert np.array_equal(ac, bc)
        for b in range(len(qs)):
            n = int(ac[b])
            assert np.array_equal(ai[b, :n], bi[b, :n]) and np.array_equal(ad[b, :n].view(np.uint32), bd[b, :n].view(np.uint32)), (topk, L, b)
        return res[0]

    for stage in range(3):
        for topk, L in ((1, 300), (1, 5000), (3, 800), (10, 4000), (50, 9000), (200, 20000)):
            both(topk, L)
        nn = g.N
        for S in (37, 4000, nn // 2):              # target ids: the filter compacts the rows along with the ids
            t = np.sort(rng.choice(nn, S, replace=False)).astype(np.int64)
            for topk, L in ((1, 20), (1, 900), (5, 30), (20, 2000)):
                both(topk, min(L, S), t)
        if stage == 0:
            g.add_codes(codes[N:], True)               # append: lists and the posting-order copy follow
        elif stage == 1:
            g.reconfigure(300, 2)
    o.add_codes(codes[N:], False)
    o.set_coarse_centers(g.coarse_centers_array()) if hasattr(o, "set_coarse_centers") else None
    if hasattr(o, "set_coarse_centers"):
        for b in range(0, 96, 11):
            for topk, L in ((1, 300), (10, 4000)):
                gi, gd, gc = g.query_ivf_batch(qs[b:b + 1], topk, None, L)
                assert_same_result((gi[0, :int(gc[0])], gd[0, :int(gc[0])]), o.query_ivf(qs[b], topk, E, L), "ivf posting-order codes")


@pytest.mark.parametrize("M,Ds,subset", [(160, 1, False), (160, 1, True), (16, 4, False)])
def test_tie_emission_of_wide_table_shards_replays_in_the_reference_order(M, Ds, subset):
    """Database sharding, exact ties, tables above the LDS budget (M = 160: 160 KiB): rii_linear_tie_emit_dev used to answer
    RII_ERR_UNSUPPORTED there; now scan_wide_kernel's key rows feed the chunk kernels.  Three shards of one database as three engines on
    this GPU: every shard emits its candidates under the bound of the shards in front of it, the records are laid out as an all-gather
    would, rii_linear_tie_replay_dev replays std::partial_sort over them -- the rows must be the single engine's (which replays ties
    with tie_rows_kernel) and the oracle's, for every query, tied or not.  (M = 16: the same harness over the LDS-table form.)"""
    import torch
    from rii_amd import RiiGpu, core
    rng = np.random.default_rng(1000 + M + subset)
    cw = np.round(rng.random((M, 256, Ds)) * 7).astype(np.float32)            # integer-valued tables: exact ties everywhere
    N = 21000
    codes = rng.integers(0, 4, size=(N, M), dtype=np.uint8)                    # few distinct bytes: many equal distances
    codes[rng.integers(0, N, 3000)] = codes[rng.integers(0, N, 3000)]
    full = RiiGpu(cw, False, simd_arch="avx512")
    full.add_codes(codes, False)
    o = O.OracleRii(cw, False, simd_arch="avx512")
    o.add_codes(codes, False)
    cuts = [0, 9000, 9100, N]                                                  # a shard shorter than k among them
    shards = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        g = RiiGpu(cw, False, simd_arch="avx512")
        g.add_codes(codes[a:b], False)
        shards.append((g, a, b))
    qs = np.round(rng.random((6, M * Ds)) * 7).astype(np.float32)
    Q = torch.from_numpy(qs).cuda()
    tids = np.sort(rng.choice(N, 6000, replace=False)).astype(np.int64) if subset else None
    nf, cap = len(qs), 12288
    for topk in (3, 40, 200):
        want = full.query_linear_batch(qs, topk, tids)
        recs = []
        kth = []                                                                # per shard: its k-th best distance per query (+inf: fewer)
        for g, a, b in shards:
            tl = None if tids is None else (tids[(tids >= a) & (tids < b)] - a)
            nloc = (b - a) if tl is None else len(tl)
            kk = min(topk, nloc)
            dk = np.full(nf, np.inf, np.float32)
            if kk == topk:
                dk = g.query_linear_batch(qs, topk, tl)[1][:, topk - 1].copy()
            bound = np.full(nf, np.inf, np.float32)
            for prev in kth:
                bound = np.minimum(bound, prev)
            kth.append(dk)
            e_ids = torch.zeros((nf, cap), dtype=torch.int64, device="cuda")
            e_d = torch.zeros((nf, cap), dtype=torch.float32, device="cuda")
            e_cnt = torch.zeros((nf + (nf & 1),), dtype=torch.int32, device="cuda")
            if nloc > 0:
                t = None if tl is None else torch.from_numpy(np.ascontiguousarray(tl)).cuda()
                bd = torch.from_numpy(bound).cuda()
                g.linear_tie_emit_dev(Q.data_ptr(), nf, topk, t.data_ptr() if t is not None else 0, 0 if t is None else t.numel(),
                                      bd.data_ptr(), a, cap, e_ids.data_ptr(), e_d.data_ptr(), e_cnt.data_ptr(), 0)
            torch.cuda.synchronize()
            assert int(e_cnt[:nf].max()) <= cap
            rec = torch.cat([x.reshape(-1).view(torch.uint8) for x in (e_cnt, e_ids, e_d)])
            pad = (-rec.numel()) % 16
            if pad:
                rec = torch.cat([rec, torch.zeros(pad, dtype=torch.uint8, device="cuda")])
            assert rec.numel() == core.linear_tie_record_bytes(nf, cap)
            recs.append(rec)
        gg = torch.stack(recs).contiguous()
        r_i = torch.empty((nf, topk), dtype=torch.int64, device="cuda")
        r_d = torch.empty((nf, topk), dtype=torch.float32, device="cuda")
        core.linear_tie_replay_dev(gg.data_ptr(), len(shards), nf, cap, topk, r_i.data_ptr(), r_d.data_ptr(), 0)
        torch.cuda.synchronize()
        assert np.array_equal(r_i.cpu().numpy(), want[0]) and np.array_equal(r_d.cpu().numpy().view(np.uint32), want[1].view(np.uint32)), topk
        for b in (0, 3):
            assert_same_result((r_i[b].cpu().numpy(), r_d[b].cpu().numpy()), o.query_linear(qs[b], topk, E if tids is None else tids),
                               "wide tie emission k=%d" % topk)
    if not subset:                                  # and through the sharded class on one rank (merge flags -> emit -> replay behind one call)
        from rii_amd.dist import DbShardedIndex
        idx = DbShardedIndex(full, 0, N)
        for topk in (3, 40):
            ids, d = idx.query_linear_batch(Q, topk)
            want = full.query_linear_batch(qs, topk, None)
            assert np.array_equal(ids.cpu().numpy(), want[0]) and np.array_equal(d.cpu().numpy(), want[1]), topk
            assert bool(idx.last_tie_flags.any()) and not bool(idx.last_tie_overflow.any())


@pytest.mark.parametrize("M,Ds", [(8, 4), (16, 6), (32, 4), (64, 2), (12, 3)])
def test_one_and_two_query_exact_scan_trips_and_tails(M, Ds):
    """scan_kernel<1> / <2> (round 4: U rows per thread and trip behind a scheduling barrier, gathered table reads, static LDS table,
    8-byte rows for two queries): top-1 of one and two queries over index sizes around the trip boundaries (1024 x U codes per block and
    trip, chunks that end inside a trip, a chunk shorter than one trip), with duplicated codes (first minimum) and target ids, against the
    oracle and against the batch path."""
    from rii_amd import RiiGpu
    rng = np.random.default_rng(4000 + M)
    cw = np.round(rng.random((M, 256, Ds)) * 9).astype(np.float32)            # integer-valued: equal distances happen
    Nmax = 70001
    codes = rng.integers(0, 256, size=(Nmax, M), dtype=np.uint8)
    codes[rng.integers(0, Nmax, 9000)] = codes[rng.integers(0, Nmax, 9000)]
    qs = np.round(rng.random((4, M * Ds)) * 9).astype(np.float32)
    for N in (1, 63, 1023, 1025, 4096, 8191, 8193, 32768 + 5, Nmax):
        g = RiiGpu(cw, False, simd_arch="avx512")
        g.add_codes(codes[:N], False)
        g.set_option("slice_topk", 0)                                         # (the few-query slice kernel ...
        g.set_option("small_topk", 0)                                         #  ... and the one-launch small-index kernel would take these calls)
        o = O.OracleRii(cw, False, simd_arch="avx512")
        o.add_codes(codes[:N], False)
        tids = np.sort(rng.choice(N, max(1, N // 3), replace=False)).astype(np.int64)
        for t in (None, tids):
            for B in (1, 2):
                ids, d = g.query_linear_batch(qs[:B], 1, t)
                for b in range(B):
                    assert_same_result((ids[b], d[b]), o.query_linear(qs[b], 1, E if t is None else t), "N=%d B=%d b=%d" % (N, B, b))
            for nchunks in (1, 7, 64):                                        # chunk boundaries inside a trip
                g.set_option("scan_chunks", nchunks)
                ids, d = g.query_linear_batch(qs[:2], 1, t)
                for b in range(2):
                    assert_same_result((ids[b], d[b]), o.query_linear(qs[b], 1, E if t is None else t), "N=%d chunks=%d b=%d" % (N, nchunks, b))
            g.set_option("scan_chunks", 0)
