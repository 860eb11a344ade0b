"""Round 5, GPU: the database-sharded entry points without their round-4 limits -- the inverted index at any L and any topk
(the reference's billion-scale run asks for L = sqrt(N) ~ 31.6 k: examples/benchmark/run_sift1b.py:105-106), the merge of
more than 8192 rows per query, record headers instead of cached shard offsets, rank-local failures that keep the ranks in
step -- through a ONE-rank communicator behind the C ABI (RCCL; the two-rank decomposition of the same kernels is
tests/test_dist_gloo.py::test_world2_sharded_real_engines_match_single_index).  Everything is compared with the CPU oracle
on the whole database, ids and distance bits, exact ties included."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import make_problem, assert_same_result

pytestmark = pytest.mark.gpu
E = np.array([], np.int64)


def _tied_problem(n):
    """integer-valued codebooks and queries + duplicated codes: exactly tied distances everywhere"""
    rng = np.random.default_rng(77)
    cw = np.round(rng.random((8, 16, 4)) * 3).astype(np.float32)
    codes = rng.integers(0, 16, size=(n, 8), dtype=np.uint8)
    codes[rng.integers(0, n, n // 4)] = codes[rng.integers(0, n, n // 4)]
    qs = np.round(rng.random((6, 32)) * 3).astype(np.float32)
    return cw, codes, qs


def _pair(cw, codes, nlist, it=2):
    from rii_amd import RiiGpu
    o = O.OracleRii(cw, False, simd_arch="avx512")
    o.add_codes(codes, False)
    o.reconfigure(nlist, it)
    g = RiiGpu(cw, False, simd_arch="avx512")
    g.add_codes(codes, False)
    g.set_coarse_centers(np.array(o.coarse_centers, np.uint8))
    assert g.posting_lists == o.posting_lists
    return g, o


def _check_ivf(idx, o, qs, topk, L, tids=None):
    import torch
    gi, gd, gc = idx.query_ivf_batch(torch.from_numpy(qs).cuda(), topk, tids, L)
    gi, gd, gc = gi.cpu().numpy(), gd.cpu().numpy(), gc.cpu().numpy()
    tied = 0
    for b in range(qs.shape[0]):
        wi, wd = o.query_ivf(qs[b], topk, E if tids is None else tids, L)
        what = "sharded ivf k=%d L=%d b=%d" % (topk, L, b)
        assert int(gc[b]) == len(wi), what
        n = len(wi)
        assert np.array_equal(gd[b, :n].view(np.uint32), np.asarray(wd, np.float32).view(np.uint32)), what
        assert list(gi[b, :n]) == list(wi), what
        tied += int(len(set(wd)) < n)
    return tied


def test_sharded_ivf_any_L_any_topk_through_the_c_abi():
    """L in {8193, 32 k, N} x topk in {1, 10, 2000}: the selection-buffer kernel (ivf_shard_any_kernel), sequences rebuilt in global
    scratch for the exact-tie replay (shard_replay_any_kernel, heaps of 10 in LDS / of 2000 walked by one lane), target ids, the
    collect-all route (topk + 1 above what a launch selects) and the global-scratch merge (G x (k + 1) > 8192 rows)."""
    from rii_amd import dist as rd
    n = 40001
    cw, codes, qs = _tied_problem(n)
    g, o = _pair(cw, codes, 40)
    idx = rd.DbShardedIndex(g, 0, n)
    sub = np.sort(np.random.default_rng(6).choice(n, 30000, replace=False)).astype(np.int64)
    tied = 0
    for topk, L, t in ((1, 8193, None), (10, 8193, None), (10, 32000, None), (1, 32000, None), (2000, 32000, None), (10, n, None),
                       (2000, n, None), (7, 20000, sub), (8100, 9000, None), (8192, 8192, None), (1, 977, None), (5, 8192, None)):
        tied += _check_ivf(idx, o, qs[:3], topk, L, t)
    assert tied > 0
    assert g.ivf_shard_max_select_rows(32000, n) < 8100 + 1 <= 9000       # (8100, 9000) really took the collect-all route
    # untied data at a shape with real tables (M = 16, Ks = 256) and many lists: nlist above the LDS limit of the coarse order too
    cwu, codesu, qsu = make_problem(8, 16, 256, 6, 30000, "unit")
    gu, ou = _pair(cwu, codesu, 173, it=1)
    iu = rd.DbShardedIndex(gu, 0, 30000)
    for topk, L in ((1500, 30000), (10, 20000), (1, 9000)):
        _check_ivf(iu, ou, qsu[:4], topk, L)          # (fp32 sums of 16 random entries do collide now and then among 1500 rows)
    assert not bool(iu.last_tie_flags.any())          # ... but not among the best two of 9000


def test_shard_kernel_two_fake_ranks_any_L_and_the_replay_by_position():
    """rii_query_ivf_shard_dev for ranks 0 and 1 of 2 on ONE GPU (two engines, the list lengths exchanged by hand), L = 20 000:
    selection rows merged on the host under (distance, position) = the oracle's top-1 / top-10 wherever no tie decides; rows = L =
    every owned candidate at the slot of its traversal position, rebuilt and replayed by rii_ivf_shard_replay_ex_dev = the
    oracle's answer, ties included."""
    import torch
    from rii_amd import RiiGpu, core
    n = 30001
    cw, codes, qs = _tied_problem(n)
    o = O.OracleRii(cw, False, simd_arch="avx512")
    o.add_codes(codes, False)
    o.reconfigure(30, 2)
    cen = np.array(o.coarse_centers, np.uint8)
    cut = n // 2 + 7
    engs = []
    for s, e in ((0, cut), (cut, n)):
        ge = RiiGpu(cw, False, simd_arch="avx512")
        ge.add_codes(codes[s:e], False)
        ge.set_coarse_centers(cen)
        engs.append((ge, s))
    nl = 30
    lens = []
    for ge, _ in engs:
        t = torch.empty(nl, dtype=torch.int32, device="cuda")
        ge.ivf_list_lengths_dev(0, 0, 0, t.data_ptr())
        ge.synchronize()
        lens.append(t)
    glen = torch.stack(lens).contiguous()
    L, B = 20000, 4
    q = torch.from_numpy(qs[:B]).cuda()
    for topk in (1, 10):
        rows = L
        recs = []
        for r, (ge, s) in enumerate(engs):
            ids = torch.empty((B, rows), dtype=torch.int64, device="cuda")
            d = torch.empty((B, rows), dtype=torch.float32, device="cuda")
            pos = torch.empty((B, rows), dtype=torch.int32, device="cuda")
            nloc = torch.empty((B,), dtype=torch.int32, device="cuda")
            cnt = torch.empty((B,), dtype=torch.int64, device="cuda")
            ge.query_ivf_shard_dev(q.data_ptr(), B, topk, 0, 0, 0, L, n, glen.data_ptr(), 2, r, ids.data_ptr(), d.data_ptr(), pos.data_ptr(),
                                   nloc.data_ptr(), cnt.data_ptr(), 0, rows)
            ge.synchronize()
            p, i, dd = pos.cpu().numpy(), ids.cpu().numpy(), d.cpu().numpy()
            own = p != np.iinfo(np.int32).max
            assert (p[own] == np.nonzero(own)[1]).all(), "rows = L above 8192: slot j holds traversal position j"
            assert int(own.sum()) == int(nloc.sum().item())
            assert (cnt.cpu().numpy() == topk).all()
            recs.append((p.astype(np.int64), np.where(i >= 0, i + s, i), dd))
        assert ((recs[0][0] != np.iinfo(np.int32).max).astype(int) + (recs[1][0] != np.iinfo(np.int32).max).astype(int) == 1).all(), \
            "every one of the L positions is owned by exactly one rank"
        nrec = (B * rows * 20 + 15) // 16 * 16
        buf = torch.zeros((2, nrec), dtype=torch.uint8)
        for r in range(2):
            buf[r, :B * rows * 8] = torch.from_numpy(np.ascontiguousarray(recs[r][0])).reshape(-1).view(torch.uint8)
            buf[r, B * rows * 8:B * rows * 16] = torch.from_numpy(np.ascontiguousarray(recs[r][1])).reshape(-1).view(torch.uint8)
            buf[r, B * rows * 16:B * rows * 20] = torch.from_numpy(np.ascontiguousarray(recs[r][2])).reshape(-1).view(torch.uint8)
        dbuf = buf.cuda()
        nsc = core.ivf_shard_replay_scratch_bytes(B, rows)
        assert nsc == B * rows * 16
        scratch = torch.empty(nsc, dtype=torch.uint8, device="cuda")
        ri = torch.empty((B, topk), dtype=torch.int64, device="cuda")
        rdd = torch.empty((B, topk), dtype=torch.float32, device="cuda")
        with pytest.raises(Exception):
            core.ivf_shard_replay_dev(dbuf.data_ptr(), 2, B, rows, topk, ri.data_ptr(), rdd.data_ptr())     # no scratch: refused, not a fault
        core.ivf_shard_replay_dev(dbuf.data_ptr(), 2, B, rows, topk, ri.data_ptr(), rdd.data_ptr(), 0, scratch.data_ptr(), nsc)
        torch.cuda.synchronize()
        for b in range(B):
            wi, wd = o.query_ivf(qs[b], topk, E, L)
            assert list(ri[b].cpu().numpy()) == list(wi) and np.array_equal(rdd[b].cpu().numpy().view(np.uint32), np.asarray(wd, np.float32).view(np.uint32))
        # the selection form (k + 1 rows per rank) merged by hand: the same distances, the same ids wherever the k + 1 best differ
        k1 = topk + 1
        sel = []
        for r, (ge, s) in enumerate(engs):
            ids = torch.empty((B, k1), dtype=torch.int64, device="cuda")
            d = torch.empty((B, k1), dtype=torch.float32, device="cuda")
            pos = torch.empty((B, k1), dtype=torch.int32, device="cuda")
            nloc = torch.empty((B,), dtype=torch.int32, device="cuda")
            cnt = torch.empty((B,), dtype=torch.int64, device="cuda")
            ge.query_ivf_shard_dev(q.data_ptr(), B, topk, 0, 0, 0, L, n, glen.data_ptr(), 2, r, ids.data_ptr(), d.data_ptr(), pos.data_ptr(),
                                   nloc.data_ptr(), cnt.data_ptr())
            ge.synchronize()
            sel.append((d.cpu().numpy(), pos.cpu().numpy().astype(np.int64), np.where(ids.cpu().numpy() >= 0, ids.cpu().numpy() + s, -1)))
            for b in range(B):            # rows ascending by (distance, position), drawn from the every-candidate record
                assert list(zip(sel[-1][0][b], sel[-1][1][b])) == sorted(zip(sel[-1][0][b], sel[-1][1][b]))
                mine = sorted(zip(recs[r][2][b][recs[r][0][b] < L], recs[r][0][b][recs[r][0][b] < L]))[:k1]
                assert [x[1] for x in mine] == list(sel[-1][1][b][:len(mine)])
        for b in range(B):
            allrows = sorted((float(sel[r][0][b][j]), int(sel[r][1][b][j]), int(sel[r][2][b][j])) for r in range(2) for j in range(k1))
            wi, wd = o.query_ivf(qs[b], topk, E, L)
            assert [x[0] for x in allrows[:topk]] == [float(x) for x in wd]
            if len({x[0] for x in allrows[:k1]}) == k1:
                assert [x[2] for x in allrows[:topk]] == list(wi)


def test_linear_dbsharded_large_topk_headers_and_local_failures():
    """rii_query_linear_dbsharded_dev over a one-rank communicator: G x (k + 1) > 8192 rows merged in global scratch; the shard's
    first id travels in the record header (two differently placed indices on ONE communicator, alternating: ADVICE r4); a heap
    deeper than the replay kernels walk keeps the (distance, id) order and says so; a rank-local failure returns its error and
    leaves the communicator usable."""
    import torch
    from rii_amd import RiiGpu, core
    from rii_amd import dist as rd
    cw, codes, qs = make_problem(6, 16, 256, 6, 12000, "unit")
    g = RiiGpu(cw, False, simd_arch="avx512")
    g.add_codes(codes, False)
    o = O.OracleRii(cw, False, simd_arch="avx512")
    o.add_codes(codes, False)
    comm = rd.get_comm()
    Q = torch.from_numpy(qs[:5]).cuda()
    for topk in (9000, 1, 40):
        oi = torch.empty((5, topk), dtype=torch.int64, device="cuda")
        od = torch.empty((5, topk), dtype=torch.float32, device="cuda")
        tie = torch.empty(5, dtype=torch.int32, device="cuda")
        ovf = torch.empty(5, dtype=torch.int32, device="cuda")
        for start in (0, 1_000_000, 0, 77):                       # the offset of THIS call, whatever the previous call used
            comm.query_linear_dbsharded_dev(g, start, Q.data_ptr(), 5, topk, 0, 0, 0, oi.data_ptr(), od.data_ptr(), tie.data_ptr(), ovf.data_ptr())
            torch.cuda.synchronize()
            for b in range(5):
                wi, wd = o.query_linear(qs[b], topk, E)
                gi = oi[b].cpu().numpy() - start
                assert np.array_equal(od[b].cpu().numpy().view(np.uint32), np.asarray(wd, np.float32).view(np.uint32)), (topk, start, b)
                if int(ovf[b].item()):            # (9000 of 12000 fp32 sums: a few collide exactly; a heap that deep is not replayed)
                    assert topk > 1024 and int(tie[b].item()) == 1
                    assert list(gi) == [i for _, i in sorted(zip(od[b].cpu().numpy().tolist(), gi.tolist()))]
                else:
                    assert list(gi) == list(wi), (topk, start, b)
            if topk <= 1024:
                assert int(ovf.sum().item()) == 0
    # exact ties under a heap of 2000: (distance, id) order kept, flagged as overflow -- never an error
    cwt, codest, qst = _tied_problem(9000)
    gt = RiiGpu(cwt, False, simd_arch="avx512")
    gt.add_codes(codest, False)
    ot = O.OracleRii(cwt, False, simd_arch="avx512")
    ot.add_codes(codest, False)
    Qt = torch.from_numpy(qst[:3]).cuda()
    oi = torch.empty((3, 2000), dtype=torch.int64, device="cuda")
    od = torch.empty((3, 2000), dtype=torch.float32, device="cuda")
    tie = torch.empty(3, dtype=torch.int32, device="cuda")
    ovf = torch.empty(3, dtype=torch.int32, device="cuda")
    comm.query_linear_dbsharded_dev(gt, 0, Qt.data_ptr(), 3, 2000, 0, 0, 0, oi.data_ptr(), od.data_ptr(), tie.data_ptr(), ovf.data_ptr())
    torch.cuda.synchronize()
    assert int(tie.sum().item()) == 3 and torch.equal(tie, ovf)
    for b in range(3):
        wi, wd = ot.query_linear(qst[b], 2000, E)
        assert np.array_equal(od[b].cpu().numpy().view(np.uint32), np.asarray(wd, np.float32).view(np.uint32))
        gi = oi[b].cpu().numpy()
        assert list(gi) == [i for _, i in sorted(zip(od[b].cpu().numpy().tolist(), gi.tolist()))]
    # a rank-local failure (more local target ids than local codes): this rank's error, the communicator stays in step and usable
    bad = torch.zeros(12001, dtype=torch.int64, device="cuda")
    oi = torch.empty((5, 1), dtype=torch.int64, device="cuda")
    od = torch.empty((5, 1), dtype=torch.float32, device="cuda")
    with pytest.raises(Exception, match="S_local"):
        comm.query_linear_dbsharded_dev(g, 0, Q.data_ptr(), 5, 1, bad.data_ptr(), 12001, 20000, oi.data_ptr(), od.data_ptr())
    comm.query_linear_dbsharded_dev(g, 0, Q.data_ptr(), 5, 1, 0, 0, 0, oi.data_ptr(), od.data_ptr())
    torch.cuda.synchronize()
    for b in range(5):
        assert list(oi[b].cpu().numpy()) == o.query_linear(qs[b], 1, E)[0]
    # the query-sharded calls write their rows where the header lives: the next database-sharded call must put it back
    qi = torch.empty((5, 3), dtype=torch.int64, device="cuda")
    qd = torch.empty((5, 3), dtype=torch.float32, device="cuda")
    comm.query_linear_qsharded_dev(g, Q.data_ptr(), 5, 3, 0, 0, qi.data_ptr(), qd.data_ptr())
    comm.query_linear_dbsharded_dev(g, 500, Q.data_ptr(), 5, 1, 0, 0, 0, oi.data_ptr(), od.data_ptr())
    torch.cuda.synchronize()
    for b in range(5):
        assert list(oi[b].cpu().numpy() - 500) == o.query_linear(qs[b], 1, E)[0]
        assert list(qi[b].cpu().numpy()) == o.query_linear(qs[b], 3, E)[0]


def test_set_posting_lists_installs_a_given_partition():
    """rii_set_posting_lists: centres + CSR lists over the codes already added, no assignment pass -- the inverted index then walks
    exactly those lists (oracle with the same CSR), and bad ids are refused."""
    from rii_amd import RiiGpu
    import bench
    cw, codes, qs = make_problem(11, 16, 256, 6, 20000, "unit")
    g = RiiGpu(cw, False, simd_arch="avx512")
    g.add_codes(codes, False)
    nlist = 141
    cen = np.random.default_rng(4).integers(0, 256, size=(nlist, 16), dtype=np.uint8)
    off, ids = bench.modulo_lists(20000, nlist)
    g.set_posting_lists(cen, off, ids)
    o = O.OracleRii(cw, False, simd_arch="avx512")
    o.add_codes(codes, False)
    o.set_csr(cen, off, ids)
    assert g.nlist == nlist and g.posting_lists[3][:3] == [3, 3 + nlist, 3 + 2 * nlist]
    for topk, L in ((1, 142), (5, 1000), (3, 20000)):
        gi, gd, gc = g.query_ivf_batch(qs[:6], topk, E, L)
        for b in range(6):
            wi, wd = o.query_ivf(qs[b], topk, E, L)
            assert int(gc[b]) == len(wi) and list(gi[b, :len(wi)]) == list(wi)
            assert np.array_equal(gd[b, :len(wi)].view(np.uint32), np.asarray(wd, np.float32).view(np.uint32))
    ids_bad = ids.copy()
    ids_bad[5] = 20000
    with pytest.raises(Exception, match="out of range"):
        g.set_posting_lists(cen, off, ids_bad)


@pytest.mark.parametrize("M,scale,nlist", [(32, "sift", 1024), (16, "sift", 300), (32, "unit", 64), (16, "unit", 1), (32, "sift", 37)])
def test_ivf_quad_kernel_equals_one_query_blocks_and_the_oracle(M, scale, nlist):
    """ivf_quad_kernel (four queries per block, tables interleaved [m][ks][query]) against ivf_fused_kernel (option ivf_quad = 0) and
    the oracle: ragged batches (the last block holds 1 - 3 padding queries), every w the shape allows (L from a handful to N),
    target ids, integer-valued data (exactly tied coarse distances -> flagged queries, replayed inside the block and -- option
    ivf_inline_exact = 0 -- by the flag-gated exact kernels), every query flagged (ivf_force_exact), rows gathered by id
    (ivf_list_codes = 0)."""
    from rii_amd import RiiGpu
    N = 30011
    cw, codes, qs = make_problem(900 + M + nlist, M, 256, 4, N, scale, dup=2000 if scale == "sift" else 0)
    rng = np.random.default_rng(5)
    Q = np.concatenate([qs, rng.permutation(qs.reshape(-1)).reshape(qs.shape), qs * 0.5, qs[:13] + 1.0]).astype(np.float32)      # 61 queries
    if scale == "sift":
        Q = np.round(Q)
    o = O.OracleRii(cw, False, simd_arch="avx512")
    o.add_codes(codes, False)
    o.reconfigure(nlist, 2)
    g = RiiGpu(cw, False, simd_arch="avx512")
    g.add_codes(codes, False)
    g.set_coarse_centers(np.array(o.coarse_centers, np.uint8))
    assert g.posting_lists == o.posting_lists
    tids = np.sort(rng.choice(N, 9000, replace=False)).astype(np.int64)
    L0 = max(1, N // nlist)
    n_flagged = 0
    for B in (61, 16, 33):
        for L, t in ((L0, None), (max(1, L0 // 3), None), (min(N, 20 * L0), None), (min(9000, 5 * L0), tids), (N, None)):
            want = [o.query_ivf(Q[b], 1, E if t is None else t, L) for b in range(B)]
            res = {}
            for quad, inline, lc, force in ((2, 1, 1, 0), (0, 1, 1, 0), (2, 0, 1, 0), (2, 1, 0, 0), (2, 1, 1, 1), (2, 0, 1, 1)):
                g.set_option("ivf_quad", quad)
                g.set_option("ivf_inline_exact", inline)
                g.set_option("ivf_list_codes", lc)
                g.set_option("ivf_force_exact", force)
                gi, gd, gc = g.query_ivf_batch(Q[:B], 1, t, L)
                res[(quad, inline, lc, force)] = (gi.copy(), gd.copy(), gc.copy())
                for b in range(B):
                    n = int(gc[b])
                    assert_same_result((gi[b, :n], gd[b, :n]), want[b], "ivf quad=%d inline=%d lcodes=%d force=%d M=%d nlist=%d B=%d L=%d b=%d"
                                       % (quad, inline, lc, force, M, nlist, B, L, b))
            g.set_option("timing", 0)
    for k, v in (("ivf_quad", 1), ("ivf_inline_exact", 1), ("ivf_list_codes", 1), ("ivf_force_exact", 0)):
        g.set_option(k, v)


def test_header_merge_for_many_fake_ranks():
    """rii_merge_topk_hdr_dev = what the database-sharded entry points run behind their all-gather, on records built by hand for
    G = 2 ... 70 fake ranks: shard offsets read from the 16-byte record headers (more than the 64 by-value offsets of round 4), the
    one-thread-per-query top-1 form, the LDS sort, the global-scratch sort (G x k > 8192 rows), exact ties across ranks ordered by
    global id, padding rows, tie flags -- and a non-zero status in one header poisoning the whole batch on every rank."""
    import torch
    from rii_amd import core
    rng = np.random.default_rng(12)
    for G, B, k in ((2, 5, 1), (3, 4, 7), (8, 3, 2000), (70, 6, 1), (70, 2, 3), (5, 2, 1700)):
        offs = np.sort(rng.choice(10**9, G, replace=False)).astype(np.int64)
        ids = np.stack([np.sort(rng.choice(50000, (B, k)), axis=1) for _ in range(G)]).astype(np.int64)          # local ids
        d = np.sort(rng.integers(0, 40, size=(G, B, k)).astype(np.float32), axis=2)                             # many exact ties across ranks
        pad = rng.random((G, B, k)) < 0.05                                                                       # padding rows: key 2^62, +inf
        pad[:, :, 0] = False
        d[pad] = np.inf
        ids[pad] = np.iinfo(np.int64).max // 2
        d = np.sort(d, axis=2)
        rec = core.merge_hdr_record_bytes(B, k)
        assert rec == core.merge_record_bytes(B, k) + 16
        buf = torch.zeros((G, rec), dtype=torch.uint8)
        for g in range(G):
            buf[g, :8] = torch.from_numpy(offs[g:g + 1]).view(torch.uint8)
            buf[g, 16:16 + B * k * 8] = torch.from_numpy(np.ascontiguousarray(ids[g])).reshape(-1).view(torch.uint8)
            buf[g, 16 + B * k * 8:16 + B * k * 12] = torch.from_numpy(np.ascontiguousarray(d[g])).reshape(-1).view(torch.uint8)
        dbuf = buf.cuda()
        k_out = k if k == 1 else min(G * k, k + 1)
        oi = torch.empty((B, k_out), dtype=torch.int64, device="cuda")
        od = torch.empty((B, k_out), dtype=torch.float32, device="cuda")
        tie = torch.empty(B, dtype=torch.int32, device="cuda")
        anyf = torch.zeros(1, dtype=torch.int32, device="cuda")
        nsc = core.merge_hdr_scratch_bytes(G, B, k)
        assert (nsc > 0) == (G * k > 8192)
        scratch = torch.empty(max(nsc, 16), dtype=torch.uint8, device="cuda")
        if k == 1:
            core.merge_topk_hdr_dev(dbuf.data_ptr(), G, B, 1, 1, oi.data_ptr(), od.data_ptr())
        else:
            core.merge_topk_hdr_dev(dbuf.data_ptr(), G, B, k, k_out, oi.data_ptr(), od.data_ptr(), tie_cols=k_out, d_out_tie=tie.data_ptr(),
                                    d_out_any=anyf.data_ptr(), d_scratch=scratch.data_ptr(), scratch_bytes=nsc)
        torch.cuda.synchronize()
        for b in range(B):
            rows = sorted((float(d[g, b, j]), int(ids[g, b, j] + (0 if pad[g, b, j] else offs[g]))) for g in range(G) for j in range(k))[:k_out]
            assert [r[0] for r in rows] == od[b].cpu().tolist(), (G, B, k, b)
            assert [r[1] for r in rows] == oi[b].cpu().tolist(), (G, B, k, b)
            if k > 1:
                want_tie = int(any(rows[j][0] == rows[j + 1][0] and np.isfinite(rows[j + 1][0]) for j in range(k_out - 1)))
                assert int(tie[b].item()) == want_tie
        # one rank reports a failure: every row of the batch is poisoned, bit 1 of the word says so
        buf[G // 2, 8:12] = torch.tensor([7], dtype=torch.int32).view(torch.uint8)
        dbuf = buf.cuda()
        anyf.zero_()
        if k == 1:
            core.merge_topk_hdr_dev(dbuf.data_ptr(), G, B, 1, 1, oi.data_ptr(), od.data_ptr())
        else:
            core.merge_topk_hdr_dev(dbuf.data_ptr(), G, B, k, k_out, oi.data_ptr(), od.data_ptr(), tie_cols=k_out, d_out_tie=tie.data_ptr(),
                                    d_out_any=anyf.data_ptr(), d_scratch=scratch.data_ptr(), scratch_bytes=nsc)
        torch.cuda.synchronize()
        assert bool((oi == -2).all()) and bool(torch.isnan(od).all())
        if k > 1:
            assert int(anyf.item()) & 2 and int(tie.sum().item()) == 0


def test_coarse_selection_and_replay_many_lists():
    """ivf_shard_any_kernel's coarse step with thousands of lists: the fast selection (three keys per thread -> DPP minima -> merge)
    must hand over to the exact std::partial_sort replay (src/rii.h:279-280) whenever it cannot prove the library's order --
    nlist = 5000 duplicated centres (exactly tied coarse distances), w = 4 ... 40, top-1 and top-k, target ids, stale lists (the walk
    leaves the first w lists) -- and the same with the fast selection off (option shard_force_replay: every query replays): all
    equal to the oracle on the same lists."""
    from rii_amd import RiiGpu
    from rii_amd import dist as rd
    n, nlist = 30011, 5000
    cw, codes, qs = _tied_problem(n)
    cen = np.ascontiguousarray(codes[np.random.default_rng(17).integers(0, n, nlist)])
    o = O.OracleRii(cw, False, simd_arch="avx512")
    o.add_codes(codes, False)
    o.set_coarse_centers(cen)
    g = RiiGpu(cw, False, simd_arch="avx512")
    g.add_codes(codes, False)
    g.set_coarse_centers(cen)
    assert g.posting_lists == o.posting_lists
    idx = rd.DbShardedIndex(g, 0, n)
    sub = np.sort(np.random.default_rng(6).choice(n, 9000, replace=False)).astype(np.int64)
    # stale lists (only the first ninth of the codes is listed): most of the 5000 lists are empty, the first w hold fewer than topk
    # candidates -> the walk continues into the unsorted tail (the fast selection hands over to the exact replay) or finds nothing
    n9 = n // 9
    gs = RiiGpu(cw, False, simd_arch="avx512")
    gs.add_codes(codes[:n9], False)
    gs.set_coarse_centers(cen)
    gs.add_codes(codes[n9:], False)
    os_ = O.OracleRii(cw, False, simd_arch="avx512")
    os_.add_codes(codes[:n9], False)
    os_.set_coarse_centers(cen)
    os_.add_codes(codes[n9:], False)
    assert gs.posting_lists == os_.posting_lists
    idxs = rd.DbShardedIndex(gs, 0, n)
    try:
        for force in (0, 1):                                # 1: the fast selection off, every query through the replay
            for e_ in (g, gs):
                e_.set_option("shard_force_replay", force)
            for topk, L, t in ((1, 6, None), (1, 60, None), (1, 220, None), (3, 30, None), (1, 9000, None), (2, 40, sub), (1, n, None)):
                _check_ivf(idx, o, qs, topk, L, t)
            n_tail = 0
            for topk, L in ((1, 3), (1, 30), (1, 2), (1, 200)):
                _check_ivf(idxs, os_, qs, topk, L, None)
    finally:
        for e_ in (g, gs):
            e_.set_option("shard_force_replay", 0)


@pytest.mark.parametrize("arch", ["sse", "avx", "avx512"])
@pytest.mark.parametrize("M,Ds", [(16, 6), (16, 8), (32, 2), (16, 3), (8, 20)])
def test_shard_kernel_builds_its_table_in_every_simd_order(M, Ds, arch):
    """ivf_shard_any_kernel computes its query's exact table itself (RiiCpp::DTable, src/rii.h:361-373).  For Ds = 2 / 6 / 8 at Ks = 256
    the codewords of several subspaces are loaded together and fvec_L2sqr's operations (src/distance.h:117-252) run on registers, one
    folded copy per SIMD variant; other Ds take the plain loop.  Non-integer data: a wrong lane order or a wrong FMA contraction
    changes low bits of the distances.  Against the oracle built for the same variant, through the database-sharded entry point."""
    from rii_amd import RiiGpu
    from rii_amd import dist as rd
    import bench
    n, nlist = 6000, 77
    cw, codes, qs = make_problem(500 + M + Ds, M, 256, Ds, n, "unit")
    cen = np.random.default_rng(M * Ds).integers(0, 256, size=(nlist, M), dtype=np.uint8)
    off, ids = bench.modulo_lists(n, nlist)
    g = RiiGpu(cw, False, simd_arch=arch)
    g.add_codes(codes, False)
    g.set_posting_lists(cen, off, ids)
    o = O.OracleRii(cw, False, simd_arch=arch)
    o.add_codes(codes, False)
    o.set_csr(cen, off, ids)
    idx = rd.DbShardedIndex(g, 0, n)
    for topk, L in ((1, 200), (3, 500), (1, n)):
        _check_ivf(idx, o, qs, topk, L, None)
