"""CPU, world_size 2, gloo: the N>1 paths of rii_amd/dist.py (database sharding with top-k merge, query sharding
with all-gather).  The local engine is the CPU oracle here (tests may use it as the checker/stand-in); on the GPU
box the same classes wrap RiiGpu.  Rendezvous on 127.0.0.1."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.util import make_problem


class _OracleBatch(object):
    """Oracle with the batched surface of RiiGpu (per-query loop)."""

    def __init__(self, cw, codes):
        from oracle import oracle as O
        self.o = O.OracleRii(cw, False, simd_arch="avx512")
        self.o.add_codes(codes, False)

    def query_linear_batch(self, Q, topk, tids=None):
        t = np.array([], np.int64) if tids is None else np.asarray(tids, np.int64)
        ids = np.empty((Q.shape[0], topk), np.int64)
        d = np.empty((Q.shape[0], topk), np.float32)
        for b in range(Q.shape[0]):
            i, dd = self.o.query_linear(Q[b], topk, t)
            ids[b], d[b] = i, dd
        return ids, d

    # ---- database-sharded linear search, exact ties: the protocol of rii_linear_tie_emit_dev / rii_linear_tie_replay_dev ----
    def linear_tie_emit(self, Qf, topk, tl, bound, start, cap):
        """Candidate rows of this shard in index order: every code below the bound (any superset of the codes that touch the
        reference's heap is valid; this stand-in applies the external bound only, the engine also bounds chunk by chunk)."""
        from oracle import oracle as O
        o = self.o
        loc = np.arange(o.N, dtype=np.int64) if tl is None else np.asarray(tl, np.int64)
        nf = Qf.shape[0]
        ids = np.zeros((nf, cap), np.int64)
        dd = np.zeros((nf, cap), np.float32)
        cnt = np.zeros(nf, np.int32)
        for f in range(nf):
            dt = O.dtable(o.codewords, Qf[f], o.arch)
            acc = np.zeros(len(loc), np.float32)
            for m in range(o.M):
                acc = (acc + dt[m, o.codes[loc, m]]).astype(np.float32)
            keep = np.nonzero(acc < bound[f])[0] if np.isfinite(bound[f]) else np.arange(len(loc))
            cnt[f] = len(keep)
            n = min(len(keep), cap)
            ids[f, :n] = loc[keep[:n]] + start
            dd[f, :n] = acc[keep[:n]]
        return ids, dd, cnt

    def linear_tie_replay(self, seq_ids, seq_d, topk):
        """std::partial_sort (the oracle's libstdc++ replay) over the gathered sequence."""
        import ctypes
        from oracle import oracle as O
        pair = np.dtype([("id", "<u8"), ("dist", "<f4")], align=True)
        arr = np.zeros(len(seq_d), pair)
        arr["id"] = np.arange(len(seq_d))
        arr["dist"] = seq_d
        O.lib().oracle_partial_sort(arr.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(topk), ctypes.c_size_t(len(seq_d)))
        sel = arr["id"][:topk].astype(np.int64)
        return np.asarray(seq_ids)[sel], np.asarray(seq_d)[sel]

    def reconfigure(self, nlist, it):
        self.o.reconfigure(nlist, it)

    def query_ivf_batch(self, Q, topk, tids, L):
        t = np.array([], np.int64) if tids is None else np.asarray(tids, np.int64)
        ids = np.full((Q.shape[0], topk), -1, np.int64)
        d = np.full((Q.shape[0], topk), np.inf, np.float32)
        cnt = np.zeros(Q.shape[0], np.int64)
        for b in range(Q.shape[0]):
            i, dd = self.o.query_ivf(Q[b], topk, t, L)
            cnt[b] = len(i)
            ids[b, :len(i)], d[b, :len(i)] = i, dd
        return ids, d, cnt


class _OracleShard(_OracleBatch):
    """CPU stand-in for one rank of the database-sharded inverted index: the protocol of include/rii_amd.h
    (rii_ivf_list_lengths_dev / rii_query_ivf_shard_dev) restated in numpy on top of the oracle's pieces."""

    def set_coarse_centers(self, centers, n_listed=None):
        """Lists over the first n_listed local codes only (stale lists: src/rii.h:283-326 tail walk / empty return)."""
        n = self.o.N if n_listed is None else n_listed
        rest = self.o.codes[n:].copy()
        self.o.codes = np.ascontiguousarray(self.o.codes[:n])
        self.o.set_coarse_centers(centers)
        self.o.codes = np.ascontiguousarray(np.concatenate([self.o.codes, rest], 0))

    def _lists(self, tl):
        ls = [np.asarray(l, np.int64) for l in self.o.posting_lists]
        if tl is not None:
            ls = [l[np.isin(l, tl)] for l in ls]
        return ls

    def ivf_list_lengths(self, tl):
        return np.array([len(l) for l in self._lists(tl)], np.int32)

    def query_ivf_shard(self, Q, topk, tl, S_global, L, N_global, glen, rank, rows=None):
        import ctypes
        from oracle import oracle as O
        o = self.o
        lists = self._lists(tl)
        nlist = o.nlist
        k1 = topk + 1 if rows is None else rows
        B = Q.shape[0]
        ids = np.full((B, k1), -1, np.int64)
        dd = np.full((B, k1), np.inf, np.float32)
        pos = np.full((B, k1), np.iinfo(np.int32).max, np.int32)
        nloc = np.zeros(B, np.int32)
        cnt = np.zeros(B, np.int64)
        w = int(np.round(L * nlist / (S_global if S_global else N_global))) + 3
        w = min(w, nlist)
        total = glen.sum(0)
        before = glen[:rank].sum(0)
        pair = np.dtype([("id", "<u8"), ("dist", "<f4")], align=True)
        codes = np.asarray(o.codes)
        for b in range(B):
            dt = O.dtable(o.codewords, Q[b], o.arch)
            coarse = np.zeros(nlist, pair)
            coarse["id"] = np.arange(nlist)
            cacc = np.zeros(nlist, np.float32)               # sequential fp32 adds over m, all centres at once
            cen = np.asarray(o.centers)
            for m in range(o.M):
                cacc = (cacc + dt[m, cen[:, m]]).astype(np.float32)
            coarse["dist"] = cacc
            O.lib().oracle_partial_sort(coarse.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(w), ctypes.c_size_t(nlist))
            c_cnt, finished = 0, False
            m_d, m_p, m_i = [], [], []                       # this rank's candidates: distance, traversal position, local id
            for c in range(nlist):
                no = int(coarse["id"][c])
                ln = int(total[no])
                take = min(ln, L - c_cnt)
                lst = lists[no]
                own = lst[:max(0, min(len(lst), take - int(before[no])))]          # offsets before + li < take
                if len(own):
                    acc = np.zeros(len(own), np.float32)
                    for m in range(o.M):                     # RiiCpp::ADist: sequential fp32 adds over m
                        acc = (acc + dt[m, codes[own, m]]).astype(np.float32)
                    m_d.append(acc)
                    m_p.append(c_cnt + int(before[no]) + np.arange(len(own), dtype=np.int64))
                    m_i.append(own)
                if c_cnt + ln >= L:
                    c_cnt, finished = L, True
                    break
                c_cnt += ln
                if c + 1 == w and c_cnt >= topk:
                    finished = True
                    break
            cnt[b] = topk if finished else 0
            if finished and m_d:
                md_, mp_, mi_ = np.concatenate(m_d), np.concatenate(m_p), np.concatenate(m_i)
                order = np.lexsort((mp_, md_))[:k1]           # ascending by (distance, position)
                n = len(order)
                nloc[b] = n
                ids[b, :n], dd[b, :n], pos[b, :n] = mi_[order], md_[order], mp_[order]
        return ids, dd, pos, nloc, cnt

    def ivf_shard_replay(self, gp, gi, gd, topk):
        """std::partial_sort (the oracle's restatement) over the candidate sequences rebuilt by position."""
        import ctypes
        from oracle import oracle as O
        pair = np.dtype([("id", "<u8"), ("dist", "<f4")], align=True)
        G, nf, rows = gp.shape
        ri = np.empty((nf, topk), np.int64)
        rd = np.empty((nf, topk), np.float32)
        for f in range(nf):
            ok = gp[:, f, :] < rows
            n = int(ok.sum())
            seq = np.zeros(n, pair)
            ids = np.zeros(n, np.int64)
            ps = gp[:, f, :][ok]
            seq["id"][ps] = ps
            seq["dist"][ps] = gd[:, f, :][ok]
            ids[ps] = gi[:, f, :][ok]
            mid = min(topk, n)                                # (a query the reference answers with ({}, {}) has no candidates)
            O.lib().oracle_partial_sort(seq.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(mid), ctypes.c_size_t(n))
            ri[f], rd[f] = -1, np.inf
            ri[f, :mid] = ids[seq["id"][:mid].astype(np.int64)]
            rd[f, :mid] = seq["dist"][:mid]
        return ri, rd


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _GpuBatch(object):
    """The real engine (both ranks share GPU 0 in this test; the collective is gloo on host tensors)."""

    def __init__(self, cw, codes):
        from rii_amd import RiiGpu
        self.g = RiiGpu(cw, False, simd_arch="avx512", device=0)
        self.g.add_codes(codes, False)

    def query_linear_batch(self, Q, topk, tids=None):
        return self.g.query_linear_batch(Q, topk, tids)

    def linear_tie_emit_dev(self, *a):
        return self.g.linear_tie_emit_dev(*a)

    def reconfigure(self, nlist, it):
        self.g.reconfigure(nlist, it)

    def query_ivf_batch(self, Q, topk, tids, L):
        return self.g.query_ivf_batch(Q, topk, tids, L)

    def set_coarse_centers(self, centers, n_listed=None):
        """Lists over the first n_listed local codes only: re-create the engine around that prefix, then append the rest."""
        codes = self.g.codes_array()
        n = len(codes) if n_listed is None else n_listed
        self.g.clear()
        self.g.add_codes(codes[:n], False)
        self.g.set_coarse_centers(centers)
        if n < len(codes):
            self.g.add_codes(codes[n:], False)

    def ivf_list_lengths(self, tl):
        t = None if tl is None else torch.from_numpy(np.ascontiguousarray(tl)).cuda()
        out = torch.empty(self.g.nlist, dtype=torch.int32, device="cuda")
        self.g.ivf_list_lengths_dev(0 if t is None else t.data_ptr(), 0 if t is None else t.numel(),
                                    0 if tl is None else max(len(tl), 1), out.data_ptr())
        self.g.synchronize()
        return out.cpu().numpy()

    def query_ivf_shard(self, Q, topk, tl, S_global, L, N_global, glen, rank, rows=None):
        B, k1 = Q.shape[0], (topk + 1 if rows is None else rows)
        t = None if tl is None else torch.from_numpy(np.ascontiguousarray(tl)).cuda()
        q = torch.from_numpy(np.ascontiguousarray(Q)).cuda()
        gl = torch.from_numpy(np.ascontiguousarray(glen, np.int32)).cuda()
        ids = torch.empty((B, k1), dtype=torch.int64, device="cuda")
        d = torch.empty((B, k1), dtype=torch.float32, device="cuda")
        pos = torch.empty((B, k1), dtype=torch.int32, device="cuda")
        nloc = torch.empty((B,), dtype=torch.int32, device="cuda")
        cnt = torch.empty((B,), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        self.g.query_ivf_shard_dev(q.data_ptr(), B, topk, 0 if t is None else t.data_ptr(), 0 if t is None else t.numel(),
                                   S_global, L, N_global, gl.data_ptr(), gl.shape[0], rank, ids.data_ptr(), d.data_ptr(),
                                   pos.data_ptr(), nloc.data_ptr(), cnt.data_ptr(), 0, 0 if rows is None else rows)
        self.g.synchronize()
        return ids.cpu().numpy(), d.cpu().numpy(), pos.cpu().numpy(), nloc.cpu().numpy(), cnt.cpu().numpy()

    def ivf_shard_replay(self, gp, gi, gd, topk):
        from rii_amd import core
        G, nf, rows = gp.shape
        n = nf * rows
        nrec = (n * 20 + 15) // 16 * 16
        buf = torch.zeros((G, nrec), dtype=torch.uint8)
        for r in range(G):
            buf[r, :n * 8] = torch.from_numpy(np.ascontiguousarray(gp[r])).reshape(-1).view(torch.uint8)
            buf[r, n * 8:n * 16] = torch.from_numpy(np.ascontiguousarray(gi[r])).reshape(-1).view(torch.uint8)
            buf[r, n * 16:n * 20] = torch.from_numpy(np.ascontiguousarray(gd[r])).reshape(-1).view(torch.uint8)
        dbuf = buf.cuda()
        ri = torch.empty((nf, topk), dtype=torch.int64, device="cuda")
        rd = torch.empty((nf, topk), dtype=torch.float32, device="cuda")
        nsc = core.ivf_shard_replay_scratch_bytes(nf, rows)           # rows > 8192: the rebuilt sequences live in global scratch
        scratch = torch.empty(max(nsc, 16), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        core.ivf_shard_replay_dev(dbuf.data_ptr(), G, nf, rows, topk, ri.data_ptr(), rd.data_ptr(), 0, scratch.data_ptr(), nsc)
        torch.cuda.synchronize()
        return ri.cpu().numpy(), rd.cpu().numpy()

    def ivf_shard_max_select_rows(self, L, N_global, S_global=0):
        return self.g.ivf_shard_max_select_rows(L, N_global, S_global)


def _check_sharded_ivf(rd, rank, world, cw, codes, qs, make_local, use_gpu, ties=False, nlist=40, big_cases=None):
    """Database-sharded inverted index against the single-index oracle on the concatenated database, incl. target ids,
    ranks without targets, stale lists (tail walk into the unsorted coarse order, `not found`)."""
    from oracle import oracle as O
    N = codes.shape[0]
    s, e = rd.shard_range(N, rank, world)
    if nlist == 40:
        trainer = O.OracleRii(cw, False, simd_arch="avx512")
        trainer.add_codes(codes, False)
        trainer.reconfigure(40, 3)
        centers = np.array(trainer.coarse_centers, np.uint8)
    else:                # many lists (past the LDS limits of the sharded kernel): random codes as centres, duplicates included
        centers = np.ascontiguousarray(codes[np.random.default_rng(17).integers(0, N, nlist)])
    Q = qs[:6]
    E = np.array([], np.int64)
    n_tied = 0
    for stale in ((False,) if (ties or big_cases is not None) else (False, True)):
        local = make_local(cw, codes[s:e])
        n_listed = (e - s) // 9 if stale else None
        local.set_coarse_centers(centers, n_listed)
        # the single index with the same lists: every shard's lists, shifted to global ids, concatenated in rank order
        full = O.OracleRii(cw, False, simd_arch="avx512")
        full.add_codes(codes, False)
        full.centers = centers
        full._lists = [[] for _ in range(nlist)]
        for r in range(world):
            rs, re_ = rd.shard_range(N, r, world)
            sh = _OracleShard(cw, codes[rs:re_])
            sh.set_coarse_centers(centers, (re_ - rs) // 9 if stale else None)
            for j, l in enumerate(sh.o.posting_lists):
                full._lists[j].extend(int(x) + rs for x in l)
        idx = rd.DbShardedIndex(local, s, e)
        rng = np.random.default_rng(3)
        sub = np.sort(rng.choice(N, 400, replace=False)).astype(np.int64)
        low = np.sort(rng.choice(N // 2 - 10, 9, replace=False)).astype(np.int64)        # targets on rank 0 only
        cases = [(1, 75, None), (1, 400, None), (5, 300, None), (3, 3, None), (10, N, None), (7, 200, sub), (2, 9, low), (1, 40, low)]
        if nlist != 40:
            cases = [(1, 3, None), (4, 60, None), (2, N, None), (3, 30, sub)]
        if big_cases is not None:
            cases = big_cases
        if stale:
            cases += [(20, 100, None), (20, 40, None), (3, 30, None), (12, 50, None), (1, 51, None)]
        n_empty = 0
        for topk, L, tids in cases:
            gi, gd, gc = idx.query_ivf_batch(Q, topk, tids, L)
            gi, gd, gc = gi.cpu().numpy(), gd.cpu().numpy(), gc.cpu().numpy()
            for b in range(Q.shape[0]):
                wi, wd = full.query_ivf(Q[b], topk, E if tids is None else tids, L)
                what = "sharded ivf stale=%s k=%d L=%d S=%s b=%d" % (stale, topk, L, None if tids is None else len(tids), b)
                assert int(gc[b]) == len(wi), what
                n_empty += (len(wi) == 0)
                n = len(wi)
                assert np.array_equal(gd[b, :n].view(np.uint32), np.asarray(wd, np.float32).view(np.uint32)), what
                assert list(gi[b, :n]) == list(wi), what          # exact ties included (heap replay over the gathered sequence)
                n_tied += int(len(set(wd)) < n)
        if stale:
            assert n_empty > 0, "the stale-list cases were meant to reach the `not found` return"
    return n_tied


def _worker(rank, world, port, q, use_gpu=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rii_amd import dist as rd
        M, Ks, Ds, N = 16, 256, 6, 3001                     # odd N: shards of unequal size
        cw, codes, qs = make_problem(5, M, Ks, Ds, N, "unit")
        full = _OracleBatch(cw, codes)
        Q = qs[:8]
        # --- database sharding ---
        s, e = rd.shard_range(N, rank, world)
        local = (_GpuBatch if use_gpu else _OracleBatch)(cw, codes[s:e])
        idx = rd.DbShardedIndex(local, s, e)
        for topk in (1, 5, 50):
            gi, gd = idx.query_linear_batch(Q, topk)
            wi, wd = full.query_linear_batch(Q, topk)
            assert np.array_equal(gd.numpy().view(np.uint32), wd.view(np.uint32)), "db-sharded dists k=%d" % topk
            assert np.array_equal(gi.numpy(), wi), "db-sharded ids k=%d" % topk
        tids = np.sort(np.random.default_rng(1).choice(N, 300, replace=False)).astype(np.int64)
        gi, gd = idx.query_linear_batch(Q, 7, tids)
        wi, wd = full.query_linear_batch(Q, 7, tids)
        assert np.array_equal(gi.numpy(), wi) and np.array_equal(gd.numpy().view(np.uint32), wd.view(np.uint32))
        # a shard that holds fewer targets than topk must pad, not fail
        few = np.array([0, 1, 2, N - 1], np.int64)
        gi, gd = idx.query_linear_batch(Q, 3, few)
        wi, wd = full.query_linear_batch(Q, 3, few)
        assert np.array_equal(gi.numpy(), wi)
        # exactly tied distances ACROSS the shards: integer-valued tables + duplicated codes.  The merged (dist, id) order is
        # not the reference's there; the flagged queries are replayed over the candidate lists of all shards
        rngt = np.random.default_rng(91)
        cwt = np.round(rngt.random((8, 16, 4)) * 3).astype(np.float32)
        codest = rngt.integers(0, 16, size=(2203, 8), dtype=np.uint8)
        codest[rngt.integers(0, 2203, 900)] = codest[rngt.integers(0, 2203, 900)]
        qst = np.round(rngt.random((6, 32)) * 3).astype(np.float32)
        fullt = _OracleBatch(cwt, codest)
        st_, et_ = rd.shard_range(2203, rank, world)
        idxt = rd.DbShardedIndex((_GpuBatch if use_gpu else _OracleBatch)(cwt, codest[st_:et_]), st_, et_)
        tidt = np.sort(rngt.choice(2203, 700, replace=False)).astype(np.int64)
        n_flag = 0
        for topk, t in ((1, None), (4, None), (30, None), (100, None), (6, tidt), (1000, None)):
            gi, gd = idxt.query_linear_batch(qst, topk, t)
            wi, wd = fullt.query_linear_batch(qst, topk, t)
            n_flag += int(idxt.last_tie_flags.sum())
            assert np.array_equal(gd.numpy().view(np.uint32), wd.view(np.uint32)), "tied db-sharded dists k=%d" % topk
            assert np.array_equal(gi.numpy(), wi), "tied db-sharded ids k=%d (the reference's heap order)" % topk
        assert n_flag > 0
        assert idxt.last_tie_flags.dtype == torch.bool and not bool(idxt.last_tie_overflow.any())
        # ties again with every target id on rank 0: rank 1's share of the targets is EMPTY, it must contribute no candidate
        # (not "all of its codes": an empty share is not "no target set")
        tid0 = np.sort(rngt.choice(rd.shard_range(2203, 0, world)[1], 500, replace=False)).astype(np.int64)
        n_flag0 = 0
        for topk in (3, 12, 60):
            gi, gd = idxt.query_linear_batch(qst, topk, tid0)
            wi, wd = fullt.query_linear_batch(qst, topk, tid0)
            n_flag0 += int(idxt.last_tie_flags.sum())
            assert np.array_equal(gd.numpy().view(np.uint32), wd.view(np.uint32)) and np.array_equal(gi.numpy(), wi), \
                "ties with all targets on rank 0, k=%d" % topk
            assert np.isin(gi.numpy(), tid0).all()
        assert n_flag0 > 0
        # a candidate list longer than TIE_CAP cannot be replayed: the query keeps the (dist, id) answer and SAYS so
        import warnings
        idxt.TIE_CAP = 8
        with warnings.catch_warnings(record=True) as wlog:
            warnings.simplefilter("always")
            gi, gd = idxt.query_linear_batch(qst, 30, None)
        idxt.TIE_CAP = rd.DbShardedIndex.TIE_CAP
        wi, wd = fullt.query_linear_batch(qst, 30, None)
        ov = idxt.last_tie_overflow.numpy()
        assert ov.any() and (ov <= idxt.last_tie_flags.numpy()).all() and any("TIE_CAP" in str(w.message) for w in wlog)
        assert np.array_equal(gd.numpy().view(np.uint32), wd.view(np.uint32))            # the distances are still the reference's
        assert np.array_equal(gi.numpy()[~ov], wi[~ov])                                  # ... and every other row is exact
        for b in np.nonzero(ov)[0]:                                                      # overflowed rows: (dist, id) order
            assert list(gi.numpy()[b]) == [i for _, i in sorted(zip(gd.numpy()[b].tolist(), gi.numpy()[b].tolist()))]
        # --- query sharding ---
        rep = _GpuBatch(cw, codes) if use_gpu else full
        qidx = rd.QueryShardedIndex(rep)
        gi, gd = qidx.query_linear_batch(Q, 4)
        wi, wd = full.query_linear_batch(Q, 4)
        assert np.array_equal(gi.numpy(), wi) and np.array_equal(gd.numpy().view(np.uint32), wd.view(np.uint32))
        # ragged batch: 7 queries over 2 ranks (slices of 4 and 3 rows)
        gi, gd = qidx.query_linear_batch(qs[:7], 3)
        wi, wd = full.query_linear_batch(qs[:7], 3)
        assert gi.shape == (7, 3) and np.array_equal(gi.numpy(), wi) and np.array_equal(gd.numpy().view(np.uint32), wd.view(np.uint32))
        # inverted index, query-sharded (replicated index): identical to the single-index answer, counts included
        full.reconfigure(40, 2)
        if rep is not full:
            rep.reconfigure(40, 2)
        for topk, L, t in ((1, 200, None), (5, 600, tids)):
            gi, gd, gc = qidx.query_ivf_batch(Q, topk, t, L)
            wi, wd, wc = full.query_ivf_batch(Q, topk, t, L)
            assert np.array_equal(gc.numpy(), wc)
            for b in range(Q.shape[0]):
                n = int(wc[b])
                assert np.array_equal(gi.numpy()[b, :n], wi[b, :n]), "ivf ids k=%d" % topk
                assert np.array_equal(gd.numpy()[b, :n].view(np.uint32), wd[b, :n].view(np.uint32))
        _check_sharded_ivf(rd, rank, world, cw, codes, qs, _GpuBatch if use_gpu else _OracleShard, use_gpu)
        # integer-valued codebooks and queries + duplicated codes: exactly tied distances inside the top-k, so the merged
        # (dist, position) order is not enough and the heap replay over the gathered candidate sequence decides
        rng = np.random.default_rng(77)
        cw2 = np.round(rng.random((8, 16, 4)) * 3).astype(np.float32)
        codes2 = rng.integers(0, 16, size=(1501, 8), dtype=np.uint8)
        codes2[rng.integers(0, 1501, 400)] = codes2[rng.integers(0, 1501, 400)]
        qs2 = np.round(rng.random((6, 32)) * 3).astype(np.float32)
        n_tied = _check_sharded_ivf(rd, rank, world, cw2, codes2, qs2, _GpuBatch if use_gpu else _OracleShard, use_gpu, ties=True)
        assert n_tied > 0
        # nlist = 5000 (above the LDS limit of the sharded kernel: coarse order in global scratch, heap in LDS) and L up to N
        _check_sharded_ivf(rd, rank, world, cw2, np.concatenate([codes2] * 5)[:7001], qs2, _GpuBatch if use_gpu else _OracleShard, use_gpu,
                           ties=True, nlist=5000)
        # round 5: L past the 8192 keys the shard kernel sorts in LDS -- the reference's billion-scale run asks for L = sqrt(N) ~ 31.6 k
        # (examples/benchmark/run_sift1b.py:105-106) -- with topk up to thousands: selection buffer, sequences rebuilt in global
        # scratch, and the collect-all route (topk + 1 above what a launch selects); integer-valued data: exact ties everywhere
        nbig = 40001
        codes_big = np.concatenate([codes2] * 27)[:nbig].copy()
        codes_big[:, 0] = np.random.default_rng(5).integers(0, 16, nbig)
        sub_big = np.sort(np.random.default_rng(6).choice(nbig, 30000, replace=False)).astype(np.int64)
        big = [(1, 8193, None), (10, 8193, None), (10, 32000, None), (1, 32000, None), (2000, 32000, None), (10, nbig, None),
               (2000, nbig, None), (7, 20000, sub_big), (8100, 9000, None)]
        n_tied_big = _check_sharded_ivf(rd, rank, world, cw2, codes_big, qs2[:3], _GpuBatch if use_gpu else _OracleShard, use_gpu,
                                        ties=True, big_cases=big)
        assert n_tied_big > 0
        # tables above the LDS budget (M = 160, Ks = 256: 160 KiB; widetab.hip): the sharded entry points used to refuse these shapes
        # (round 4: ivf_shard_kernel<GTAB>, key-row tie emission) -- tied distances across the shards, linear and inverted index
        rngw = np.random.default_rng(177)
        cw3 = np.round(rngw.random((160, 256, 1)) * 3).astype(np.float32)
        codes3 = rngw.integers(0, 6, size=(1801, 160), dtype=np.uint8)
        codes3[rngw.integers(0, 1801, 500)] = codes3[rngw.integers(0, 1801, 500)]
        qs3 = np.round(rngw.random((6, 160)) * 3).astype(np.float32)
        full3 = _OracleBatch(cw3, codes3)
        s3, e3 = rd.shard_range(1801, rank, world)
        idx3 = rd.DbShardedIndex((_GpuBatch if use_gpu else _OracleBatch)(cw3, codes3[s3:e3]), s3, e3)
        n_flag3 = 0
        for topk in (1, 4, 40):
            gi, gd = idx3.query_linear_batch(qs3, topk, None)
            wi, wd = full3.query_linear_batch(qs3, topk, None)
            n_flag3 += int(idx3.last_tie_flags.sum())
            assert np.array_equal(gd.numpy().view(np.uint32), wd.view(np.uint32)) and np.array_equal(gi.numpy(), wi), "wide tables, db-sharded k=%d" % topk
        assert n_flag3 > 0 and not bool(idx3.last_tie_overflow.any())
        assert _check_sharded_ivf(rd, rank, world, cw3, codes3, qs3, _GpuBatch if use_gpu else _OracleShard, use_gpu, ties=True) > 0
        q.put((rank, "ok"))
    except Exception as ex:                                   # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _run_world2(use_gpu):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, use_gpu)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in res), res


def test_world2_gloo_db_and_query_sharding():
    _run_world2(use_gpu=False)


@pytest.mark.gpu
def test_world2_sharded_real_engines_match_single_index():
    """The same two-rank decomposition with the HIP engine on each rank (database shards / query shards), checked
    against the oracle's answer on the concatenated database."""
    _run_world2(use_gpu=True)


def _nccl_world1_worker(port, q):
    """world_size 1 over the "nccl" backend (= RCCL): process-group init and the collectives of rii_amd/dist.py run on
    device tensors, engine -> RCCL -> HIP merge kernel, on the one GPU the test box has."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1)
        from rii_amd import dist as rd
        from rii_amd import RiiGpu, core
        M, Ks, Ds, N = 16, 256, 6, 3001
        cw, codes, qs = make_problem(5, M, Ks, Ds, N, "unit", dup=200)
        full = _OracleBatch(cw, codes)
        g = RiiGpu(cw, False, simd_arch="avx512", device=0)
        g.add_codes(codes, False)
        Q = torch.from_numpy(qs[:7]).cuda()
        idx = rd.DbShardedIndex(g, 0, N)
        for topk in (1, 5, 50):
            gi, gd = idx.query_linear_batch(Q, topk)
            assert gi.is_cuda and gd.is_cuda
            wi, wd = full.query_linear_batch(qs[:7], topk)
            assert np.array_equal(gd.cpu().numpy().view(np.uint32), wd.view(np.uint32)), "db-sharded dists k=%d" % topk
            # duplicated codes: bit-equal distances inside the top-k -> flagged, emitted, gathered over RCCL and replayed on the
            # device in the reference's heap order (rii_linear_tie_emit_dev / rii_linear_tie_replay_dev)
            assert np.array_equal(gi.cpu().numpy(), wi), "db-sharded ids k=%d" % topk
            assert idx.last_tie_flags.is_cuda and idx.last_tie_flags.dtype == torch.bool      # flags never leave the device
            n_lin_flag = n_lin_flag + int(idx.last_tie_flags.sum()) if topk > 1 else 0
        assert n_lin_flag > 0 and not bool(idx.last_tie_overflow.any())
        # the device path reads ONE word per batch (none for top-1): untied data -> no replay, flags all clear
        cwu, codesu, qsu = make_problem(6, M, Ks, Ds, 2000, "unit")
        gu = RiiGpu(cwu, False, simd_arch="avx512", device=0)
        gu.add_codes(codesu, False)
        iu = rd.DbShardedIndex(gu, 0, 2000)
        ui, ud = iu.query_linear_batch(torch.from_numpy(qsu[:7]).cuda(), 5)
        wi, wd = _OracleBatch(cwu, codesu).query_linear_batch(qsu[:7], 5)
        assert np.array_equal(ui.cpu().numpy(), wi) and not bool(iu.last_tie_flags.any())
        # merge kernel with per-rank id offsets and tie flags (rii_merge_topk_ex_dev)
        Bm, km, Gm = 2, 3, 2
        idm = torch.tensor([[[0, 4, 2], [1, 0, 3]], [[0, 1, 2], [5, 6, 7]]], dtype=torch.int64)
        ddm = torch.tensor([[[1., 2., 2.], [0., 1., 9.]], [[1., 3., 4.], [2., 3., 4.]]], dtype=torch.float32)
        nrm = core.merge_record_bytes(Bm, km)
        bm = torch.zeros((Gm, nrm), dtype=torch.uint8)
        for r in range(Gm):
            bm[r, :Bm * km * 8] = idm[r].reshape(-1).view(torch.uint8)
            bm[r, Bm * km * 8:Bm * km * 12] = ddm[r].reshape(-1).view(torch.uint8)
        dbm = bm.cuda()
        oi = torch.empty((Bm, 4), dtype=torch.int64, device="cuda")
        od = torch.empty((Bm, 4), dtype=torch.float32, device="cuda")
        tf = torch.empty(Bm, dtype=torch.int32, device="cuda")
        af = torch.zeros(1, dtype=torch.int32, device="cuda")
        core.merge_topk_ex_dev(dbm.data_ptr(), Gm, Bm, km, 4, [0, 100], oi.data_ptr(), od.data_ptr(), tie_cols=4,
                               d_out_tie=tf.data_ptr(), d_out_any=af.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        assert oi.cpu().tolist() == [[0, 100, 2, 4], [1, 0, 105, 106]] and od.cpu().tolist() == [[1., 1., 2., 2.], [0., 1., 2., 3.]]
        assert tf.cpu().tolist() == [1, 0] and int(af.item()) == 1
        # the merge kernel alone: 3 fake shards with exact ties across shards -> (dist, id) order
        B, k, G = 2, 4, 3
        ids = torch.tensor([[[9, 1, 5, 7], [2, 3, 4, 6]], [[8, 0, 10, 11], [12, 13, 14, 15]],
                            [[20, 21, 22, 23], [24, 25, 26, 27]]], dtype=torch.int64)
        dd = torch.tensor([[[1., 1., 2., 9.], [0., 3., 3., 3.]], [[1., 1.5, 2., 2.], [0., 0., 5., 5.]],
                           [[0.5, 1., 7., 8.], [3., 3., 3., 4.]]], dtype=torch.float32)
        nrec = core.merge_record_bytes(B, k)
        buf = torch.zeros((G, nrec), dtype=torch.uint8)
        for r in range(G):
            buf[r, :B * k * 8] = ids[r].reshape(-1).view(torch.uint8)
            buf[r, B * k * 8:B * k * 12] = dd[r].reshape(-1).view(torch.uint8)
        dbuf = buf.cuda()
        oi = torch.empty((B, k), dtype=torch.int64, device="cuda")
        od = torch.empty((B, k), dtype=torch.float32, device="cuda")
        core.merge_topk_dev(dbuf.data_ptr(), G, B, k, oi.data_ptr(), od.data_ptr(), torch.cuda.current_stream().cuda_stream)
        wi, wd = rd.merge_topk(torch.cat(list(ids), 1), torch.cat(list(dd), 1), k)
        assert torch.equal(oi.cpu(), wi) and torch.equal(od.cpu(), wd), (oi, wi)
        # query sharding + inverted index through the device path
        qidx = rd.QueryShardedIndex(g)
        gi, gd = qidx.query_linear_batch(Q, 4)
        wi, wd = full.query_linear_batch(qs[:7], 4)
        assert gi.is_cuda and np.array_equal(gi.cpu().numpy(), wi)
        g.reconfigure(40, 2); full.reconfigure(40, 2)
        gi, gd, gc = qidx.query_ivf_batch(Q, 5, None, 600)
        wi, wd, wc = full.query_ivf_batch(qs[:7], 5, None, 600)
        assert np.array_equal(gc.cpu().numpy(), wc) and np.array_equal(gi.cpu().numpy(), wi)
        # database-sharded inverted index, device path (engine -> RCCL -> merge kernel with payload), one shard = whole database
        gi, gd, gc = idx.query_ivf_batch(Q, 5, None, 600)
        assert gi.is_cuda and np.array_equal(gc.cpu().numpy(), wc) and np.array_equal(gd.cpu().numpy().view(np.uint32), wd.view(np.uint32))
        tids = np.sort(np.random.default_rng(1).choice(N, 300, replace=False)).astype(np.int64)
        gi, gd, gc = idx.query_ivf_batch(Q, 1, tids, 100)
        wi, wd, wc = full.query_ivf_batch(qs[:7], 1, tids, 100)
        assert np.array_equal(gc.cpu().numpy(), wc) and np.array_equal(gi.cpu().numpy(), wi)
        # exact ties (integer-valued codebooks, duplicated codes): merge flags them, the candidate sequences are gathered
        # and std::partial_sort is replayed on the device (rii_ivf_shard_replay_dev)
        from oracle import oracle as O
        rng = np.random.default_rng(77)
        cw2 = np.round(rng.random((8, 16, 4)) * 3).astype(np.float32)
        codes2 = rng.integers(0, 16, size=(1501, 8), dtype=np.uint8)
        codes2[rng.integers(0, 1501, 400)] = codes2[rng.integers(0, 1501, 400)]
        qs2 = np.round(rng.random((6, 32)) * 3).astype(np.float32)
        o2 = O.OracleRii(cw2, False, simd_arch="avx512")
        o2.add_codes(codes2, False)
        o2.reconfigure(40, 3)
        g2 = RiiGpu(cw2, False, simd_arch="avx512", device=0)
        g2.add_codes(codes2, False)
        g2.set_coarse_centers(np.array(o2.coarse_centers, np.uint8))
        idx2 = rd.DbShardedIndex(g2, 0, 1501)
        n_tied = 0
        for topk, L in ((5, 300), (10, 1501), (3, 40)):
            gi, gd, gc = idx2.query_ivf_batch(torch.from_numpy(qs2).cuda(), topk, None, L)
            n_tied += int(idx2.last_tie_flags.sum().item())
            for b in range(6):
                wi, wd = o2.query_ivf(qs2[b], topk, np.array([], np.int64), L)
                assert int(gc[b].item()) == len(wi) and list(gi[b, :len(wi)].cpu().numpy()) == list(wi), ("tie replay", topk, L, b)
        assert n_tied > 0
        q.put((0, "ok"))
    except Exception:                                         # noqa: BLE001
        import traceback
        q.put((0, "FAIL: " + traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.gpu
def test_nccl_world1_device_resident_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_world1_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=150)
    p.join(60)
    assert res[1] == "ok", res


@pytest.mark.gpu
def test_bench_runs_under_torchrun_world1_with_rccl():
    """bench.py launched the way the driver launches N > 1 (env rendezvous, backend nccl), at world size 1: the timed
    all-gather (`with_gather`) and both strong-scaling forms (QueryShardedIndex / DbShardedIndex: engine -> RCCL -> device merge)
    run over RCCL and the line keeps the contract's keys."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                          "--n-base", "100000", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "with_gather", "host_call", "strong", "uninstrumented"):
        assert key in line, key
    assert line["with_gather"]["backend"] == "nccl" and line["with_gather"]["ms_per_step"] > 0
    assert 0 < line["roofline"]["frac"] <= 1.0
    for form in ("query_sharded", "db_sharded"):
        assert line["strong"][form]["ms_per_step"] > 0 and line["strong"][form]["results_match_single_engine"] is True, line["strong"]


def _bench_two_ranks(cmd_prefix, extra_env, world=2, n_base=100000, batch=1024, timeout=300):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RII_BENCH_BACKEND="gloo", RII_BENCH_DEVICE="0", **extra_env)
    env.pop("WORLD_SIZE", None)
    cmd = cmd_prefix(sys.executable, os.path.join(root, "bench.py")) + ["--gpus", str(world), "--steps", "2", "--warmup", "1", "--n-base",
                                                                        str(n_base), "--batch", str(batch), "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["config"]["global_batch"] == batch * world and line["scaling"] == "weak"
    assert line["with_gather"]["ms_per_step"] > 0 and line["value"] > 0
    # strong scaling: ONE global batch over the ranks (ragged slices when it does not divide), both decompositions, answers equal to
    # the single engine's
    st = line["strong"]
    from rii_amd.dist import shard_range
    assert st["query_sharded"]["rows_per_rank"] == [shard_range(batch, r, world)[1] - shard_range(batch, r, world)[0] for r in range(world)]
    assert st["query_sharded"]["results_match_single_engine"] is True, st["query_sharded"]
    assert st["db_sharded"]["codes_per_rank"] == shard_range(n_base, 0, world)[1] and st["db_sharded"]["results_match_single_engine"] is True, st["db_sharded"]
    assert st["query_sharded"]["ms_per_step"] > 0 and st["db_sharded"]["ms_per_step"] > 0
    return line


@pytest.mark.gpu
def test_bench_eight_ranks_preflight_on_one_gpu():
    """The driver's 8-GPU launch line (`torch.distributed.run --nproc-per-node 8 bench.py --gpus 8`) rehearsed on the one GPU of the test
    box (collectives on gloo, a reduced index and a batch that does NOT divide by eight: ragged query slices, shards of unequal
    size): the broadcast of the inputs, strong.query_sharded / strong.db_sharded with results_match_single_engine, one JSON line --
    inside a fraction of the driver's time limit."""
    import time
    t0 = time.time()
    line = _bench_two_ranks(lambda py, bench: [py, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
                                               "127.0.0.1", "--master-port", str(_free_port()), bench], {}, world=8, n_base=60001, batch=1020, timeout=600)
    assert time.time() - t0 < 600
    assert line["strong"]["query_sharded"]["rows_per_rank"] == [128, 128, 128, 128, 127, 127, 127, 127]


@pytest.mark.gpu
def test_bench_two_ranks_launched_like_the_driver():
    """bench.py under `python -m torch.distributed.run --nproc-per-node 2` (the driver's N > 1 launch), both ranks on the one
    GPU of the test box with the collectives on gloo: the N > 1 control flow (broadcast of the inputs, per-rank batches,
    timed all-gather, MAX over ranks, weak AND strong scaling, one JSON line from rank 0)."""
    _bench_two_ranks(lambda py, bench: [py, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                                        "127.0.0.1", "--master-port", str(_free_port()), bench], {})


@pytest.mark.gpu
def test_bench_gpus_2_typed_directly_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment: bench.py becomes the launcher (round 2 asserted)."""
    _bench_two_ranks(lambda py, bench: [py, bench], {})


def test_bench_self_launch_command_line(monkeypatch):
    """CPU: the launcher bench.py builds for `--gpus N` typed directly (no GPU needed to check the command)."""
    import importlib.util
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}
    monkeypatch.setattr(subprocess, "call", lambda cmd: seen.setdefault("cmd", cmd) and 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "7"]


@pytest.mark.gpu
def test_single_engine_without_process_group_and_bench_deep():
    """No process group at all (one engine, one GPU): the sharding classes stay device-resident (engine -> record -> merge kernel),
    and `bench.py --workload deep` (configs[4] shape through DbShardedIndex) runs as typed at N = 1."""
    import json
    import subprocess
    import sys
    from rii_amd import RiiGpu
    from rii_amd import dist as rd
    assert not dist.is_initialized()
    M, Ks, Ds, N = 16, 256, 6, 5003
    cw, codes, qs = make_problem(8, M, Ks, Ds, N, "unit", dup=300)
    g = RiiGpu(cw, False, simd_arch="avx512", device=0)
    g.add_codes(codes, False)
    full = _OracleBatch(cw, codes)
    idx = rd.DbShardedIndex(g, 0, N)
    Q = torch.from_numpy(qs[:9]).cuda()
    for topk in (1, 4, 40):
        gi, gd = idx.query_linear_batch(Q, topk)
        wi, wd = full.query_linear_batch(qs[:9], topk)
        assert gi.is_cuda and np.array_equal(gi.cpu().numpy(), wi) and np.array_equal(gd.cpu().numpy().view(np.uint32), wd.view(np.uint32))
    gi, gd = rd.QueryShardedIndex(g).query_linear_batch(Q, 3)
    assert gi.is_cuda and np.array_equal(gi.cpu().numpy(), full.query_linear_batch(qs[:9], 3)[0])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "deep", "--n-base", "2000000", "--steps", "3",
                          "--warmup", "1"], env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["roofline"]["kernel"] == "fscan_mx_dual_kernel" and 0 < line["roofline"]["frac"] <= 1.0


@pytest.mark.gpu
def test_bench_default_line_carries_every_baseline_config():
    """`python bench.py` (default workload, N = 1; sizes cut down for the test): the line keeps the contract's keys and `others` holds
    every other BASELINE config with its roofline and a reference baseline whose ids match the GPU's."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "OMP_NUM_THREADS")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--n-base", "200000", "--steps", "3", "--warmup", "1",
                          "--deep-shard", "400000", "--preheat", "0.02"], env=env, capture_output=True, text=True, timeout=400)
    assert out.returncode == 0, out.stderr[-3000:]
    printed = [l for l in out.stdout.splitlines() if l.strip()]
    assert printed[-1].startswith("{") and len([l for l in printed if l.startswith("{")]) == 1, printed[-3:]
    assert len(printed[-1]) < 12000, len(printed[-1])          # round 5's 20.6 KB line came back from the driver unparsed
    line = json.loads(printed[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "others", "full_form"):
        assert key in line, key
    assert line["steps"] == 3 and line["warmup"] == 1 and line["cpu_baseline"]["ids_match_gpu"] is True
    assert line["roofline"]["bound"] == "lds-gather" and 0 < line["roofline"]["frac"] <= 1.0
    full = json.load(open(os.path.join(root, line["full_form"])))          # the long form: same numbers, plus the prose
    assert full["value"] == pytest.approx(line["value"], rel=1e-4) and "note" in full["roofline"] and "preheat" in full
    oth = line["others"]
    for name in ("subset", "ivf", "subset_ivf", "deep_shard"):
        o = oth[name]
        assert o["ms_per_step"] > 0 and o["kernel_ms"] > 0 and 0 < o["roofline"]["frac"] <= 1.0, name
        assert o["cpu_baseline"]["ids_match_gpu"] is True and o["cpu_baseline"]["queries_compared"] > 0, name
    assert oth["ivf"]["roofline"]["bound"] == "valu-issue" and oth["deep_shard"]["kernel"] == "fscan_mx_dual_kernel"
    for name in ("linear", "ivf"):
        r = oth["readme_n10k"][name]
        assert r["p50_ms"] > 0 and r["cpu_baseline"]["ids_match_gpu"] is True


def test_merge_topk_canonical_rule():
    from rii_amd.dist import merge_topk
    ids = torch.tensor([[7, 3, 9, 1, 5]], dtype=torch.int64)
    d = torch.tensor([[2.0, 1.0, 1.0, 2.0, 0.5]], dtype=torch.float32)
    gi, gd = merge_topk(ids, d, 4)
    assert gi.tolist() == [[5, 3, 9, 1]] and gd.tolist() == [[0.5, 1.0, 1.0, 2.0]]


def test_shard_range_covers_everything():
    from rii_amd.dist import shard_range
    for N in (0, 1, 7, 1000, 1001):
        for w in (1, 2, 3, 8):
            r = [shard_range(N, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == N
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
