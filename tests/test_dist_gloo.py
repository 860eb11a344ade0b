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

    def reconfigure(self, nlist, it):
        self.g.reconfigure(nlist, it)

    def query_ivf_batch(self, Q, topk, tids, L):
        return self.g.query_ivf_batch(Q, topk, tids, L)


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
        # --- query sharding ---
        rep = _GpuBatch(cw, codes) if use_gpu else full
        qidx = rd.QueryShardedIndex(rep)
        gi, gd = qidx.query_linear_batch(Q, 4)
        wi, wd = full.query_linear_batch(Q, 4)
        assert np.array_equal(gi.numpy(), wi) and np.array_equal(gd.numpy().view(np.uint32), wd.view(np.uint32))
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
