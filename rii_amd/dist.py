"""Multi-GPU sharding of the query path: one process per GPU, `torch.distributed` (backend "nccl" == RCCL over
xGMI on ROCm; "gloo" in the CPU tests).  The reference has no multi-device code at all; the parity
target is the single-index result on the concatenated database.

Two decompositions, both embarrassingly parallel up to one small exchange:
  * query sharding   -- index replicated, rank r answers queries [r*B/W, (r+1)*B/W); exchange = all-gather of
                        the per-rank (ids, dists) rows (B/W * k * 12 bytes: latency-bound).
  * database sharding -- rank r holds codes [start_r, stop_r) (Deep1B-shape: 16 GB of codes -> 2 GB per GPU);
                        every rank answers all B queries on its shard, global id = start_r + local id; exchange =
                        all-gather of B*k (dist, id) pairs per rank, then a k-way merge under the canonical
                        (dist asc, id asc) rule, computed identically on every rank.
No ring all-reduce anywhere: payloads are KBs, so the 7 x 153 GB/s xGMI links are irrelevant; what matters is
one collective per batch.
"""
import numpy as np
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(N, rank, world_size):
    """Contiguous id range of `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(N), int(world_size))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def _comm_device():
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def _as_tensor(a, dtype, device):
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dtype)


def merge_topk(ids, dists, topk):
    """k smallest of each row under (dist asc, id asc).  ids int64 [B, C], dists float32 [B, C] -> [B, topk]."""
    order = torch.sort(ids, dim=1, stable=True).indices                    # secondary key first ...
    d1 = torch.gather(dists, 1, order)
    i1 = torch.gather(ids, 1, order)
    order2 = torch.sort(d1, dim=1, stable=True).indices                    # ... then stable sort on the primary
    return torch.gather(i1, 1, order2)[:, :topk].contiguous(), torch.gather(d1, 1, order2)[:, :topk].contiguous()


def allgather_merge_topk(local_ids, local_dists, topk, id_offset=0, group=None):
    """Database sharding: every rank contributes its local top-k (local ids + id_offset = global ids); returns
    the merged global top-k on every rank."""
    dev = _comm_device()
    ids = _as_tensor(local_ids, torch.int64, dev) + int(id_offset)
    d = _as_tensor(local_dists, torch.float32, dev)
    rank, w = world()
    if w == 1:
        return merge_topk(ids, d, topk)
    gi = [torch.empty_like(ids) for _ in range(w)]
    gd = [torch.empty_like(d) for _ in range(w)]
    dist.all_gather(gi, ids, group=group)
    dist.all_gather(gd, d, group=group)
    return merge_topk(torch.cat(gi, dim=1), torch.cat(gd, dim=1), topk)


def allgather_query_shards(local_ids, local_dists, group=None):
    """Query sharding: concatenate the per-rank result rows in rank order (every rank gets all rows)."""
    dev = _comm_device()
    ids = _as_tensor(local_ids, torch.int64, dev)
    d = _as_tensor(local_dists, torch.float32, dev)
    rank, w = world()
    if w == 1:
        return ids, d
    gi = [torch.empty_like(ids) for _ in range(w)]
    gd = [torch.empty_like(d) for _ in range(w)]
    dist.all_gather(gi, ids, group=group)
    dist.all_gather(gd, d, group=group)
    return torch.cat(gi, dim=0), torch.cat(gd, dim=0)


class DbShardedIndex(object):
    """Database-sharded linear search.  `engine` is this rank's local engine (RiiGpu) holding codes
    [start, stop) of the global database; it must offer query_linear_batch(Q, topk, target_ids)."""

    def __init__(self, engine, start, stop, group=None):
        self.engine, self.start, self.stop, self.group = engine, int(start), int(stop), group

    def query_linear_batch(self, Q, topk, target_ids=None):
        n_local = self.stop - self.start
        k_local = min(topk, n_local)
        tl = None
        if target_ids is not None and len(target_ids):
            t = np.asarray(target_ids, np.int64)
            tl = t[(t >= self.start) & (t < self.stop)] - self.start          # this shard's targets, still sorted
            k_local = min(topk, len(tl))
        if k_local > 0:
            ids, d = self.engine.query_linear_batch(Q, k_local, tl)
        else:
            ids = np.zeros((Q.shape[0], 0), np.int64)
            d = np.zeros((Q.shape[0], 0), np.float32)
        if k_local < topk:            # pad so that every rank contributes the same shape
            pad = topk - k_local
            ids = np.concatenate([ids, np.full((Q.shape[0], pad), np.iinfo(np.int64).max // 2, np.int64)], axis=1)
            d = np.concatenate([d, np.full((Q.shape[0], pad), np.inf, np.float32)], axis=1)
        # global ids: add the shard offset to real entries only
        ids = np.where(np.isfinite(d), ids + self.start, ids)
        return allgather_merge_topk(ids, d, topk, 0, self.group)


class QueryShardedIndex(object):
    """Query-sharded search over a replicated index: rank r answers rows [r*B/W, (r+1)*B/W) of Q."""

    def __init__(self, engine, group=None):
        self.engine, self.group = engine, group

    def query_linear_batch(self, Q, topk, target_ids=None):
        rank, w = world()
        assert Q.shape[0] % w == 0, "batch must divide evenly over the ranks (all-gather of equal shapes)"
        s, e = shard_range(Q.shape[0], rank, w)
        ids, d = self.engine.query_linear_batch(np.ascontiguousarray(Q[s:e]), topk, target_ids)
        return allgather_query_shards(ids, d, self.group)

    def query_ivf_batch(self, Q, topk, target_ids, L):
        """Inverted-index search, query-sharded (the reference's "stop at exactly L candidates in list order" rule is a
        per-query sequential rule, so the index is replicated and the queries are split).
        Returns (ids [B,topk], dists [B,topk], counts [B]) on every rank."""
        rank, w = world()
        assert Q.shape[0] % w == 0, "batch must divide evenly over the ranks (all-gather of equal shapes)"
        s, e = shard_range(Q.shape[0], rank, w)
        ids, d, cnt = self.engine.query_ivf_batch(np.ascontiguousarray(Q[s:e]), topk, target_ids, L)
        gi, gd = allgather_query_shards(ids, d, self.group)
        gc, _ = allgather_query_shards(np.asarray(cnt, np.int64).reshape(-1, 1),
                                       np.zeros((len(cnt), 1), np.float32), self.group)
        return gi, gd, gc.reshape(-1)
