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
            same = gi.cpu().numpy() == wi
            tied = np.zeros_like(same)
            tied[:, 1:] |= wd[:, 1:] == wd[:, :-1]
            tied[:, :-1] |= wd[:, 1:] == wd[:, :-1]
            assert (same | tied).all(), "db-sharded ids k=%d" % topk
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
    all-gather (`with_gather`) runs over RCCL and the line keeps the contract's keys."""
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
                "vs_baseline", "dtype", "data", "config", "roofline", "with_gather", "host_call"):
        assert key in line, key
    assert line["with_gather"]["backend"] == "nccl" and line["with_gather"]["ms_per_step"] > 0
    assert 0 < line["roofline"]["frac"] <= 1.0


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
