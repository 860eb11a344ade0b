"""Multi-GPU sharding of the query path: one process per GPU, `torch.distributed` (backend "nccl" == RCCL over
xGMI on ROCm; "gloo" in the CPU tests).  The reference has no multi-device code at all; the parity
target is the single-index result on the concatenated database.

Two decompositions, both embarrassingly parallel up to one small exchange:
  * query sharding   -- index replicated, rank r answers its slice of the queries; exchange = ONE all-gather of the
                        per-rank result rows (ids, dists[, counts] packed: B/W * (12 k [+ 8]) bytes, latency-bound).
  * database sharding -- rank r holds codes [start_r, stop_r) (Deep1B-shape: 16 GB of codes -> 2 GB per GPU);
                        every rank answers all B queries on its shard, global id = start_r + local id; exchange = ONE
                        all-gather of a packed record of B*k (id, dist) pairs per rank, then a k-way merge under the
                        canonical (dist asc, id asc) rule computed identically on every rank -- by a HIP kernel
                        (rii_merge_topk_dev, csrc/merge.hip) when the records live in HBM.
With the "nccl" backend everything stays on the device: engine -> device tensors -> RCCL -> merge kernel, no host hop.
No ring all-reduce anywhere: payloads are KBs, so the 7 x 153 GB/s xGMI links are irrelevant; what matters is one
collective per batch.

Inverted index over a database-sharded index (DbShardedIndex.query_ivf_batch): the coarse centres are replicated and the
posting lists local; the reference's global "stop at exactly L candidates in list order" rule is replayed identically on
every rank from the all-gathered per-rank list lengths (nlist ints per rank and batch), each rank scores the candidates it
owns, and the per-rank (dist, traversal position, id) rows are merged under (dist, position); queries whose k+1 best
distances tie exactly get their whole candidate sequence gathered and std::partial_sort replayed on it -- csrc/ivfshard.hip.

Ties across shards of the linear search: the merge orders bit-equal distances of different shards by id; whenever two of the
merged k+1 best distances are bit-equal the query is replayed in the reference's std::partial_sort order over the candidates
every shard emits in index order (DbShardedIndex.query_linear_batch; csrc/tieorder.hip).  The tie flags are computed by the
merge kernel on the device; the host reads one 4-byte word per batch (none for top-1).
"""
import numpy as np
import torch
import torch.distributed as dist


def world(group=None):
    """(rank, size) inside `group` (default: the whole world; no process group: one rank)."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_range(N, rank, world_size):
    """Contiguous id range of `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(N), int(world_size))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def _comm_device(like=None):
    """Where the exchanged tensors live: HBM under "nccl"; host memory under "gloo"; with no process group at all (a single
    engine) wherever the caller's tensor already is (default: the current GPU when there is one)."""
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend() == "nccl":
            return torch.device("cuda", torch.cuda.current_device())
        return torch.device("cpu")
    if isinstance(like, torch.Tensor):
        return like.device
    if torch.cuda.is_available():
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


_SIDE = {}


class _engine_stream(object):
    """The engine's *_dev calls take a hipStream_t, and torch reports its default stream as 0 -- which the C ABI reads as
    "the engine's own stream".  Device-resident sharding therefore runs engine call, packing, RCCL collective and merge on
    ONE explicit (non-default) torch stream per device, fenced against the caller's current stream on entry and exit."""

    def __enter__(self):
        dev = torch.cuda.current_device()
        self.outer = torch.cuda.current_stream()
        self.direct = self.outer.cuda_stream != 0      # the caller already works on an explicit stream: use it, no fences
        if self.direct:
            return self.outer.cuda_stream
        if dev not in _SIDE:
            _SIDE[dev] = torch.cuda.Stream(device=dev)
        self.side = _SIDE[dev]
        self.side.wait_stream(self.outer)
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()
        return self.side.cuda_stream

    def __exit__(self, *exc):
        if self.direct:
            return False
        self.ctx.__exit__(*exc)
        self.outer.wait_stream(self.side)
        return False


class _NullCtx(object):
    def __init__(self, value=None):
        self.value = value

    def __enter__(self):
        return self.value

    def __exit__(self, *exc):
        return False


def _handoff(*tensors):
    """Tensors produced on the side stream and handed to the caller's stream (caching-allocator bookkeeping)."""
    cur = torch.cuda.current_stream()
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            t.record_stream(cur)
    return tensors if len(tensors) != 1 else tensors[0]


def _as_tensor(a, dtype, device):
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dtype)


def _all_gather_bytes(rec, group=None):
    """One collective: every rank's byte record, concatenated in rank order -> uint8 [W, len(rec)].  Runs the collective
    whenever a process group exists (world size 1 included, so that a single-GPU box exercises RCCL too)."""
    rank, w = world(group)
    if not (dist.is_available() and dist.is_initialized()):
        return rec.reshape(1, -1)
    out = torch.empty(w * rec.numel(), dtype=torch.uint8, device=rec.device)
    dist.all_gather_into_tensor(out, rec.reshape(-1).contiguous(), group=group)
    return out.reshape(w, -1)


def _pack(parts):
    """Byte record of the given tensors back to back, padded to 16 bytes (every rank's record starts aligned)."""
    rec = torch.cat([p.reshape(-1).view(torch.uint8) for p in parts])
    pad = (-rec.numel()) % 16
    if pad:
        rec = torch.cat([rec, torch.zeros(pad, dtype=torch.uint8, device=rec.device)])
    return rec


def _field(g, r, off, count, dtype):
    """`count` elements of `dtype` at byte offset `off` of rank r's record in the gathered [W, rec] byte tensor."""
    nbytes = count * torch.empty((), dtype=dtype).element_size()
    return g[r, off:off + nbytes].view(dtype)


def merge_topk(ids, dists, topk):
    """k smallest of each row under (dist asc, id asc).  ids int64 [B, C], dists float32 [B, C] -> [B, topk] (host path)."""
    order = torch.sort(ids, dim=1, stable=True).indices                    # secondary key first ...
    d1 = torch.gather(dists, 1, order)
    i1 = torch.gather(ids, 1, order)
    order2 = torch.sort(d1, dim=1, stable=True).indices                    # ... then stable sort on the primary
    return torch.gather(i1, 1, order2)[:, :topk].contiguous(), torch.gather(d1, 1, order2)[:, :topk].contiguous()


def allgather_merge_topk(local_ids, local_dists, topk, id_offset=0, group=None):
    """Database sharding: every rank contributes its local top-k (local ids + id_offset = global ids, padding rows marked
    by dist = +inf keep their id); returns the merged global top-k on every rank.  Device tensors in -> device tensors out
    through rii_merge_topk_dev; host tensors / numpy -> torch sort on the host (gloo tests)."""
    dev = _comm_device(local_ids)
    d = _as_tensor(local_dists, torch.float32, dev)
    ids = _as_tensor(local_ids, torch.int64, dev)
    if id_offset:
        ids = torch.where(torch.isfinite(d), ids + int(id_offset), ids)
    B, k = ids.shape
    rank, w = world(group)
    if dev.type == "cuda" and w * k <= 8192:
        from . import core
        nrec = core.merge_record_bytes(B, k)
        rec = torch.zeros(nrec, dtype=torch.uint8, device=dev)
        rec[:B * k * 8].view(torch.int64).copy_(ids.reshape(-1))
        rec[B * k * 8:B * k * 12].view(torch.float32).copy_(d.reshape(-1))
        gathered = _all_gather_bytes(rec, group)
        out_ids = torch.empty((B, topk), dtype=torch.int64, device=dev)
        out_d = torch.empty((B, topk), dtype=torch.float32, device=dev)
        if topk == k:
            h = torch.cuda.current_stream().cuda_stream
            if h == 0:                 # default stream: run the merge inside the side-stream fence
                with _engine_stream() as sh:
                    core.merge_topk_dev(gathered.data_ptr(), w, B, k, out_ids.data_ptr(), out_d.data_ptr(), sh)
            else:
                core.merge_topk_dev(gathered.data_ptr(), w, B, k, out_ids.data_ptr(), out_d.data_ptr(), h)
            return out_ids, out_d
    if w == 1 and not dist.is_initialized():
        return merge_topk(ids, d, topk)
    g = _all_gather_bytes(_pack([ids, d]), group)
    gi = [_field(g, r, 0, B * k, torch.int64).reshape(B, k) for r in range(g.shape[0])]
    gd = [_field(g, r, B * k * 8, B * k, torch.float32).reshape(B, k) for r in range(g.shape[0])]
    return merge_topk(torch.cat(gi, dim=1), torch.cat(gd, dim=1), topk)


def allgather_query_shards(local_ids, local_dists, group=None, local_counts=None, rows=None):
    """Query sharding: concatenate the per-rank result rows in rank order (every rank gets all rows) with ONE collective
    over a packed record.  `rows`: per-rank row counts when the batch does not divide evenly (records are padded to the
    largest slice and trimmed after the gather).  Returns (ids, dists) or (ids, dists, counts)."""
    dev = _comm_device(local_ids)
    ids = _as_tensor(local_ids, torch.int64, dev)
    d = _as_tensor(local_dists, torch.float32, dev)
    cnt = None if local_counts is None else _as_tensor(local_counts, torch.int64, dev).reshape(-1)
    rank, w = world(group)
    n, k = ids.shape
    nmax = n if rows is None else int(max(rows))
    if nmax > n:                                           # pad the short slice: equal shapes for the all-gather
        ids = torch.cat([ids, torch.zeros((nmax - n, k), dtype=ids.dtype, device=dev)])
        d = torch.cat([d, torch.zeros((nmax - n, k), dtype=d.dtype, device=dev)])
        if cnt is not None:
            cnt = torch.cat([cnt, torch.zeros(nmax - n, dtype=cnt.dtype, device=dev)])
    parts = [ids] + ([cnt] if cnt is not None else []) + [d]        # 8-byte fields first: every field stays aligned
    g = _all_gather_bytes(_pack(parts), group)
    wg = g.shape[0]
    take = [nmax] * wg if rows is None else [int(r) for r in rows]
    o_cnt = nmax * k * 8
    o_d = o_cnt + (nmax * 8 if cnt is not None else 0)
    oi = torch.cat([_field(g, r, 0, nmax * k, torch.int64).reshape(nmax, k)[:take[r]] for r in range(wg)], dim=0)
    od = torch.cat([_field(g, r, o_d, nmax * k, torch.float32).reshape(nmax, k)[:take[r]] for r in range(wg)], dim=0)
    if cnt is None:
        return oi, od
    return oi, od, torch.cat([_field(g, r, o_cnt, nmax, torch.int64)[:take[r]] for r in range(wg)], dim=0)


def _is_device_engine(engine):
    """Device-resident path: a HIP engine whose exchange tensors live in HBM -- under "nccl", or with no process group at all (a
    single engine on its own GPU); under "gloo" the collectives run on host tensors and the engine's host-pointer surface is used."""
    if not hasattr(engine, "query_linear_dev"):
        return False
    if dist.is_available() and dist.is_initialized():
        return dist.get_backend() == "nccl"
    return torch.cuda.is_available()


_COMMS = {}


def _use_c_comm():
    """The exchange runs behind the C ABI (rii_comm_*: RCCL bound by librii_amd.so itself, engine kernels -> ncclAllGather -> unpack /
    merge enqueued by ONE library call) whenever the records live in HBM: under the "nccl" backend, or with no process group at all
    (a single engine: a one-rank communicator).  Under "gloo" (the CPU tests, and two ranks sharing one GPU) the collectives run on
    host tensors through torch.distributed as before."""
    if not torch.cuda.is_available():
        return False
    if dist.is_available() and dist.is_initialized():
        return dist.get_backend() == "nccl"
    return True


def get_comm(group=None):
    """The process's rii_comm for `group` (default: the whole world; none initialised: one rank), created on first use -- rank 0 of
    the group draws the id (rii_comm_unique_id) and torch.distributed carries its 128 bytes to the others.  The cache is keyed on
    the group OBJECT (which the key keeps alive: a destroyed group's id() cannot come back as another group's); close_comms()
    destroys the communicators explicitly -- call it before destroy_process_group() rather than leaving it to interpreter exit."""
    from . import core
    dev = torch.cuda.current_device()
    rank, w = world(group)
    key = (group, dev)
    c = _COMMS.get(key)
    if c is not None and (c.rank != rank or c.size != w):
        c.close()                # (collective, like close_comms(): the stale RCCL communicator is not left to __del__ at an arbitrary time)
        c = None
    if c is None:
        if w > 1 or (dist.is_available() and dist.is_initialized()):
            box = [core.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            cid = box[0]
        else:
            cid = core.comm_unique_id()
        c = _COMMS[key] = core.Comm(cid, rank, w, dev)
    return c


def close_comms():
    """Destroy every cached communicator (collective per communicator: call it on every rank, before destroy_process_group())."""
    for key in list(_COMMS):
        _COMMS.pop(key).close()


class DbShardedIndex(object):
    """Database-sharded linear search.  `engine` is this rank's local engine holding codes [start, stop) of the global
    database; it must offer query_linear_batch(Q, topk, target_ids) (and query_linear_dev for the device-resident path:
    RiiGpu under the "nccl" backend)."""

    def __init__(self, engine, start, stop, group=None):
        self.engine, self.start, self.stop, self.group = engine, int(start), int(stop), group

    def _local_targets(self, target_ids, topk):
        n_local = self.stop - self.start
        if target_ids is None or len(target_ids) == 0:
            return None, min(topk, n_local)
        t = np.asarray(target_ids, np.int64)
        tl = t[(t >= self.start) & (t < self.stop)] - self.start          # this shard's targets, still sorted
        return tl, min(topk, len(tl))

    TIE_CAP = 12288          # rows of a flagged query's candidate list per rank (8192 = one unbounded chunk, + the bounded rest)
    MERGE_MAX_KEYS = 8192    # G * (k + 1) rows per query the device merge sorts in LDS (rii_merge_topk_ex_dev); above: torch merge
    GATHER_BUDGET = 512 << 20    # bytes of gathered every-candidate rows (G x L x 20 per query) per group of queries (inverted index)

    def all_starts(self):
        """First global id of every rank's shard, in rank order (one tiny all-gather, cached): the merge kernel adds them to the
        LOCAL ids the engines wrote, so no rank rewrites its own rows."""
        if getattr(self, "_all_starts", None) is None:
            rank, w = world(self.group)
            if dist.is_available() and dist.is_initialized():
                t = torch.tensor([self.start], dtype=torch.int64, device=_comm_device())
                out = torch.empty(w, dtype=torch.int64, device=t.device)
                dist.all_gather_into_tensor(out, t, group=self.group)
                self._all_starts = [int(x) for x in out.cpu()]
            else:
                self._all_starts = [self.start]
        return self._all_starts

    def query_linear_batch(self, Q, topk, target_ids=None, out=None):
        """Top-k over the whole sharded database, in the reference's order.  (`out`: optional preallocated (ids [B, topk] int64, dists
        [B, topk] float32) device tensors for the device path.)  Every rank contributes its k + 1 best rows (k = 1:
        its best row -- a heap of one keeps the first minimum in index order = the smallest id, so top-1 never needs a replay);
        the merge under (dist, id) is the reference's answer unless two of the merged k + 1 best distances are bit-equal -- then
        the order (and, at the cut, the membership) is what std::partial_sort makes of ALL distances in index order, and those
        queries are replayed exactly: rii_linear_tie_emit_dev / rii_linear_tie_replay_dev (include/rii_amd.h).

        Device engines ("nccl", or a single engine): engine -> record -> all-gather -> rii_merge_topk_ex_dev, all on one stream;
        the tie flags are computed by the merge kernel and the host reads ONE 4-byte word per batch (none for top-1) to learn
        whether any replay is needed.  `last_tie_flags` (bool tensor [B], identical on every rank) holds the flags of the last
        call, `last_tie_overflow` the flagged queries whose candidate list exceeded TIE_CAP rows on some rank: those keep the
        (dist, id) order among exactly tied distances instead of the reference's (a warning is issued)."""
        rows = topk if topk == 1 else topk + 1
        tl, k_local = self._local_targets(target_ids, rows)
        B = Q.shape[0]
        if _is_device_engine(self.engine):
            return self._query_linear_device(Q, B, topk, rows, tl, k_local, 0 if tl is None else len(target_ids), out)
        big = np.iinfo(np.int64).max // 2
        ids = torch.full((B, rows), big, dtype=torch.int64)
        d = torch.full((B, rows), float("inf"), dtype=torch.float32)
        if k_local > 0:
            li, ld = self.engine.query_linear_batch(np.asarray(Q), k_local, tl)
            ids[:, :k_local] = torch.from_numpy(np.asarray(li, np.int64)) + self.start
            d[:, :k_local] = torch.from_numpy(np.asarray(ld, np.float32))
        g = _all_gather_bytes(_pack([ids, d]), self.group)
        G = g.shape[0]
        gi = [_field(g, r, 0, B * rows, torch.int64).reshape(B, rows) for r in range(G)]
        gd = [_field(g, r, B * rows * 8, B * rows, torch.float32).reshape(B, rows) for r in range(G)]
        mi, md = merge_topk(torch.cat(gi, dim=1), torch.cat(gd, dim=1), rows)
        if topk == 1:
            flags = torch.zeros(B, dtype=torch.bool)
        else:
            flags = ((md[:, :topk] == md[:, 1:rows]) & torch.isfinite(md[:, 1:rows])).any(dim=1)
        out_i, out_d = mi[:, :topk].contiguous(), md[:, :topk].contiguous()
        self.last_tie_flags = flags
        self.last_tie_overflow = torch.zeros(B, dtype=torch.bool)
        fidx = np.nonzero(flags.numpy())[0]
        if len(fidx) and hasattr(self.engine, "linear_tie_emit"):
            self._replay_linear_ties_host(Q, fidx, topk, tl, gd, out_i, out_d)
        elif len(fidx) and hasattr(self.engine, "linear_tie_emit_dev") and torch.cuda.is_available():
            # real engines under a host collective (gloo; the tests' two ranks on one GPU): device emit / replay, host gather
            cdev = torch.device("cuda", torch.cuda.current_device())
            self._replay_linear_ties_device(_as_tensor(Q, torch.float32, cdev), torch.from_numpy(fidx).to(cdev), topk, rows,
                                            None if tl is None else torch.from_numpy(np.ascontiguousarray(tl)).to(cdev), tl,
                                            g.to(cdev), out_i, out_d, cdev, None)
        return out_i, out_d

    def _query_linear_device(self, Q, B, topk, rows, tl, k_local, S_global=0, out=None):
        """Device engines: ONE library call -- engine kernels -> record -> ncclAllGather -> merge kernel (+ the exact-tie replay) are
        enqueued by rii_query_linear_dbsharded_dev (round 4).  Shapes beyond the merge kernel's limits (G x (k + 1) > 8192 rows, more
        than 64 ranks) and host collectives ("gloo") keep the torch path below."""
        rank, G = world(self.group)
        if _use_c_comm() and self.MERGE_MAX_KEYS >= 8192:       # (any G, any topk since round 5; the class attribute is the tests' lever)
            dev = torch.device("cuda", torch.cuda.current_device())
            comm = get_comm(self.group)
            with _engine_stream() as sh:
                q = _as_tensor(Q, torch.float32, dev)
                t = None if tl is None else torch.from_numpy(np.ascontiguousarray(tl)).to(dev)
                if out is None:
                    out = (torch.empty((B, topk), dtype=torch.int64, device=dev), torch.empty((B, topk), dtype=torch.float32, device=dev))
                if topk == 1:
                    comm.query_linear_dbsharded_dev(self.engine, self.start, q.data_ptr(), B, topk, 0 if t is None else t.data_ptr(),
                                                    0 if t is None else t.numel(), S_global, out[0].data_ptr(), out[1].data_ptr(), stream=sh)
                    self.last_tie_flags = self._zero_flags(B, dev)
                    self.last_tie_overflow = self._zero_flags(B, dev)
                else:
                    tie = torch.empty(B, dtype=torch.int32, device=dev)
                    ovf = torch.empty(B, dtype=torch.int32, device=dev)
                    comm.query_linear_dbsharded_dev(self.engine, self.start, q.data_ptr(), B, topk, 0 if t is None else t.data_ptr(),
                                                    0 if t is None else t.numel(), S_global, out[0].data_ptr(), out[1].data_ptr(),
                                                    tie.data_ptr(), ovf.data_ptr(), self.TIE_CAP, sh)
                    self.last_tie_flags = tie.bool()
                    self.last_tie_overflow = ovf.bool()
                    nbad = int(ovf.sum().item())
                    if nbad:
                        import warnings
                        warnings.warn("rii_amd.dist: %d tied quer%s produced more than TIE_CAP=%d replay candidates on some shard; exactly "
                                      "tied distances of those rows are ordered by id, not in std::partial_sort's order (see "
                                      "last_tie_overflow)" % (nbad, "y" if nbad == 1 else "ies", self.TIE_CAP))
            _handoff(self.last_tie_flags, self.last_tie_overflow)
            return _handoff(*out)
        return self._query_linear_device_torch(Q, B, topk, rows, tl, k_local)

    def _query_linear_device_torch(self, Q, B, topk, rows, tl, k_local):
        from . import core
        dev = _comm_device()
        if dev.type != "cuda":
            dev = torch.device("cuda", torch.cuda.current_device())
        with _engine_stream() as sh:
            q = _as_tensor(Q, torch.float32, dev)
            t = None if tl is None else torch.from_numpy(np.ascontiguousarray(tl)).to(dev)
            nrec = core.merge_record_bytes(B, rows)
            nid = B * rows * 8
            if k_local == rows:            # the engine writes its LOCAL ids and distances straight into the record
                rec = torch.empty(nrec, dtype=torch.uint8, device=dev)
                self.engine.query_linear_dev(q.data_ptr(), B, rows, t.data_ptr() if t is not None else 0,
                                             0 if t is None else t.numel(), rec.data_ptr(), rec.data_ptr() + nid, sh)
            else:                          # fewer local codes / targets than rows: padding rows (key 2^62, distance +inf)
                ids = torch.full((B, rows), np.iinfo(np.int64).max // 2, dtype=torch.int64, device=dev)
                d = torch.full((B, rows), float("inf"), dtype=torch.float32, device=dev)
                if k_local > 0:
                    li = torch.empty((B, k_local), dtype=torch.int64, device=dev)
                    ld = torch.empty((B, k_local), dtype=torch.float32, device=dev)
                    self.engine.query_linear_dev(q.data_ptr(), B, k_local, t.data_ptr() if t is not None else 0,
                                                 0 if t is None else t.numel(), li.data_ptr(), ld.data_ptr(), sh)
                    ids[:, :k_local] = li
                    d[:, :k_local] = ld
                rec = torch.zeros(nrec, dtype=torch.uint8, device=dev)
                rec[:nid].view(torch.int64).copy_(ids.reshape(-1))
                rec[nid:nid + B * rows * 4].view(torch.float32).copy_(d.reshape(-1))
            g = _all_gather_bytes(rec, self.group)
            if g.device != dev:            # gloo with real engines (tests on a one-GPU box): the collective ran on the host
                g = g.to(dev)
            G = g.shape[0]
            mi = torch.empty((B, rows), dtype=torch.int64, device=dev)
            md = torch.empty((B, rows), dtype=torch.float32, device=dev)
            self.last_tie_overflow = self._zero_flags(B, dev)
            if G * rows > self.MERGE_MAX_KEYS or G > 64:
                # beyond the merge kernel's LDS sort (rii_merge_topk_ex_dev: G * k <= 8192 keys, G <= 64 offsets): the per-rank id
                # offsets are added to the finite rows here and torch merges under (dist, id), as the host path does
                gi, gd = [], []
                for r in range(G):
                    di = _field(g, r, nid, B * rows, torch.float32).reshape(B, rows)
                    ii = _field(g, r, 0, B * rows, torch.int64).reshape(B, rows)
                    gi.append(torch.where(torch.isfinite(di), ii + int(self.all_starts()[r]), ii))
                    gd.append(di)
                mi, md = merge_topk(torch.cat(gi, dim=1), torch.cat(gd, dim=1), rows)
                if topk == 1:
                    self.last_tie_flags = self._zero_flags(B, dev)
                    out = (mi, md)
                else:
                    tie = ((md[:, :topk] == md[:, 1:rows]) & torch.isfinite(md[:, 1:rows])).any(dim=1)
                    out_i, out_d = mi[:, :topk].contiguous(), md[:, :topk].contiguous()
                    self.last_tie_flags = tie
                    if bool(tie.any().item()) and hasattr(self.engine, "linear_tie_emit_dev"):
                        self._replay_linear_ties_device(q, torch.nonzero(tie).flatten(), topk, rows, t, tl, g, out_i, out_d, dev, sh)
                    out = (out_i, out_d)
            elif topk == 1:
                core.merge_topk_ex_dev(g.data_ptr(), G, B, rows, rows, self.all_starts(), mi.data_ptr(), md.data_ptr(), stream=sh)
                self.last_tie_flags = self._zero_flags(B, dev)
                out = (mi, md)
            else:
                tie = torch.empty(B, dtype=torch.int32, device=dev)
                anyf = torch.zeros(1, dtype=torch.int32, device=dev)
                core.merge_topk_ex_dev(g.data_ptr(), G, B, rows, rows, self.all_starts(), mi.data_ptr(), md.data_ptr(),
                                       tie_cols=rows, d_out_tie=tie.data_ptr(), d_out_any=anyf.data_ptr(), stream=sh)
                out_i, out_d = mi[:, :topk].contiguous(), md[:, :topk].contiguous()
                self.last_tie_flags = tie.bool()
                if int(anyf.item()) and hasattr(self.engine, "linear_tie_emit_dev"):     # the batch's one host read
                    self._replay_linear_ties_device(q, torch.nonzero(tie).flatten(), topk, rows, t, tl, g, out_i, out_d, dev, sh)
                out = (out_i, out_d)
        _handoff(self.last_tie_flags, self.last_tie_overflow)       # (allocated on the side stream, read by the caller's)
        return _handoff(*out)

    def _zero_flags(self, B, dev):
        """A fresh all-false flag tensor (callers may keep or mutate `last_tie_flags` / `last_tie_overflow`: ADVICE r3)."""
        return torch.zeros(B, dtype=torch.bool, device=dev)

    def _tie_bound(self, gd, fsel, topk):
        """bound of this rank = the smallest k-th distance any EARLIER shard reported (an earlier shard's codes come first in the
        reference's index order, so the heap top is already at or below it when this shard's first code is visited)."""
        rank, G = world(self.group)
        bound = torch.full((fsel.numel(),), float("inf"), dtype=torch.float32, device=fsel.device)
        for s in range(rank):
            bound = torch.minimum(bound, gd[s][fsel, topk - 1])           # +inf when shard s holds fewer than k codes
        return bound

    def _note_overflow(self, fsel, ok):
        """A list longer than TIE_CAP rows cannot be replayed: the query keeps its (dist, id) answer, and says so."""
        bad = fsel[~ok]
        if bad.numel():
            self.last_tie_overflow = self.last_tie_overflow.clone()
            self.last_tie_overflow[bad] = True
            import warnings
            warnings.warn("rii_amd.dist: %d tied quer%s produced more than TIE_CAP=%d replay candidates on some shard; exactly tied "
                          "distances of those rows are ordered by id, not in std::partial_sort's order (see last_tie_overflow)"
                          % (int(bad.numel()), "y" if bad.numel() == 1 else "ies", self.TIE_CAP))

    def _replay_linear_ties_device(self, q, fsel, topk, rows, t, tl, g, out_i, out_d, dev, sh):
        """The flagged queries `fsel` redone in the reference's order with the engine's emit / replay kernels.  q, fsel, t, g live
        on `dev`; the candidate lists travel through the process group's own device (HBM under "nccl", host under "gloo")."""
        from . import core
        nf, cap, B = int(fsel.numel()), self.TIE_CAP, q.shape[0]
        gd = [_field(g, r, B * rows * 8, B * rows, torch.float32).reshape(B, rows) for r in range(g.shape[0])]
        ctx = _engine_stream() if sh is None else _NullCtx(sh)
        with ctx as st:
            bd = self._tie_bound(gd, fsel, topk).contiguous()
            qf = q[fsel].contiguous()
            e_ids = torch.zeros((nf, cap), dtype=torch.int64, device=dev)
            e_d = torch.zeros((nf, cap), dtype=torch.float32, device=dev)
            e_cnt = torch.zeros((nf + (nf & 1),), dtype=torch.int32, device=dev)      # padded to 8 bytes
            # a rank whose share of the target ids is EMPTY contributes nothing (S = 0 would mean "no target set" to the engine)
            if not (tl is not None and len(tl) == 0):
                self.engine.linear_tie_emit_dev(qf.data_ptr(), nf, topk, t.data_ptr() if t is not None else 0,
                                                0 if t is None else t.numel(), bd.data_ptr(), self.start, cap, e_ids.data_ptr(),
                                                e_d.data_ptr(), e_cnt.data_ptr(), st)
            rec = _pack([e_cnt, e_ids, e_d])
            assert rec.numel() == core.linear_tie_record_bytes(nf, cap)
            gg = _all_gather_bytes(rec.to(_comm_device(rec)), self.group).to(dev)
            cnts = torch.stack([_field(gg, r, 0, nf, torch.int32) for r in range(gg.shape[0])])
            ok = (cnts <= cap).all(dim=0)                                            # a truncated list cannot be replayed
            r_i = torch.empty((nf, topk), dtype=torch.int64, device=dev)
            r_d = torch.empty((nf, topk), dtype=torch.float32, device=dev)
            core.linear_tie_replay_dev(gg.data_ptr(), gg.shape[0], nf, cap, topk, r_i.data_ptr(), r_d.data_ptr(), st)
            sel = fsel[ok].to(out_i.device)
            out_i[sel] = r_i[ok].to(out_i.device)
            out_d[sel] = r_d[ok].to(out_d.device)
        self._note_overflow(fsel.to(self.last_tie_overflow.device), ok.to(self.last_tie_overflow.device))

    def _replay_linear_ties_host(self, Q, fidx, topk, tl, gd, out_i, out_d):
        """host engines (the CPU stand-ins of the gloo tests): same protocol through engine.linear_tie_emit / linear_tie_replay"""
        nf, cap = len(fidx), self.TIE_CAP
        fsel = torch.from_numpy(fidx)
        bound = self._tie_bound(gd, fsel, topk)
        Qf = np.ascontiguousarray(np.asarray(Q.cpu() if isinstance(Q, torch.Tensor) else Q, np.float32)[fidx])
        if tl is not None and len(tl) == 0:
            e_ids, e_d, e_cnt = np.zeros((nf, cap), np.int64), np.zeros((nf, cap), np.float32), np.zeros(nf, np.int32)
        else:
            e_ids, e_d, e_cnt = self.engine.linear_tie_emit(Qf, topk, tl, bound.numpy(), self.start, cap)
        cnt_t = torch.zeros((nf + (nf & 1),), dtype=torch.int32)
        cnt_t[:nf] = torch.from_numpy(np.asarray(e_cnt, np.int32))
        rec = _pack([cnt_t, torch.from_numpy(np.ascontiguousarray(e_ids, np.int64)), torch.from_numpy(np.ascontiguousarray(e_d, np.float32))])
        g = _all_gather_bytes(rec, self.group)
        cb = cnt_t.numel() * 4
        lists = []
        for r in range(g.shape[0]):
            c = _field(g, r, 0, nf, torch.int32).numpy()
            li = _field(g, r, cb, nf * cap, torch.int64).reshape(nf, cap).numpy()
            ld = _field(g, r, cb + nf * cap * 8, nf * cap, torch.float32).reshape(nf, cap).numpy()
            lists.append((c, li, ld))
        ok = torch.ones(nf, dtype=torch.bool)
        for j, f in enumerate(fidx):
            if any(int(c[j]) > cap for c, _, _ in lists):
                ok[j] = False
                continue
            seq_i = np.concatenate([li[j, :int(c[j])] for c, li, _ in lists])
            seq_d = np.concatenate([ld[j, :int(c[j])] for c, _, ld in lists])
            ri, rd = self.engine.linear_tie_replay(seq_i, seq_d, topk)
            out_i[f] = torch.from_numpy(np.asarray(ri, np.int64))
            out_d[f] = torch.from_numpy(np.asarray(rd, np.float32))
        self._note_overflow(fsel, ok)

    # ---- inverted index over the sharded database (protocol: include/rii_amd.h, csrc/ivfshard.hip) ----
    def total_codes(self):
        """N of the whole database: the sum of the shard sizes (one all-reduce, cached)."""
        if getattr(self, "_n_total", None) is None:
            n = torch.tensor([self.stop - self.start], dtype=torch.int64, device=_comm_device())
            if dist.is_available() and dist.is_initialized():
                dist.all_reduce(n, op=dist.ReduceOp.SUM, group=self.group)
            self._n_total = int(n.item())
        return self._n_total

    def query_ivf_batch(self, Q, topk, target_ids, L):
        """RiiCpp::QueryIvf (src/rii.h:244-326) on the concatenated database.  Every rank must hold the SAME coarse centres
        (engine.set_coarse_centers) with posting lists over its own codes.  Returns (ids [B,topk] global, dists [B,topk],
        counts [B]) on every rank; counts[b] == 0 where the reference returns ({}, {}).  Queries whose merged k+1 best
        distances hold an exact tie (`last_tie_flags`) are redone exactly: every rank sends all the candidates it owns
        and std::partial_sort is replayed over the rebuilt candidate sequence (rii_ivf_shard_replay_dev)."""
        rank, w = world(self.group)
        B = Q.shape[0]
        k1 = topk + 1
        tl, _ = self._local_targets(target_ids, topk)
        S_global = 0 if target_ids is None else len(target_ids)
        N_global = self.total_codes()
        dev = _comm_device()
        if _is_device_engine(self.engine) and _use_c_comm() and self.MERGE_MAX_KEYS >= 8192:
            # ONE library call (round 4): list lengths -> all-gather -> the shard's walk -> all-gather -> merge (+ the exact-tie replay);
            # any L, any topk, any number of ranks (round 5)
            dev = torch.device("cuda", torch.cuda.current_device())
            comm = get_comm(self.group)
            with _engine_stream() as sh:
                t = None if tl is None else torch.from_numpy(np.ascontiguousarray(tl)).to(dev)
                q = _as_tensor(Q, torch.float32, dev)
                ids = torch.empty((B, topk), dtype=torch.int64, device=dev)
                d = torch.empty((B, topk), dtype=torch.float32, device=dev)
                cnt = torch.empty((B,), dtype=torch.int64, device=dev)
                tie = torch.empty((B,), dtype=torch.int32, device=dev)
                comm.query_ivf_dbsharded_dev(self.engine, self.start, N_global, q.data_ptr(), B, topk, 0 if t is None else t.data_ptr(),
                                             0 if t is None else t.numel(), S_global, L, ids.data_ptr(), d.data_ptr(), cnt.data_ptr(),
                                             tie.data_ptr(), sh)
                self.last_tie_flags = tie.bool()
            _handoff(self.last_tie_flags)
            return _handoff(ids, d, cnt)
        if _is_device_engine(self.engine):
            # real engines under a host collective ("gloo": the tests' two ranks on one GPU): the same protocol, driven from here
            from . import core
            with _engine_stream() as sh:
                t = None if tl is None else torch.from_numpy(np.ascontiguousarray(tl)).to(dev)
                nlist = self.engine.nlist
                lens = torch.empty(nlist, dtype=torch.int32, device=dev)
                self.engine.ivf_list_lengths_dev(0 if t is None else t.data_ptr(), 0 if t is None else t.numel(), S_global,
                                                 lens.data_ptr(), sh)
                glen = _all_gather_bytes(lens.view(torch.uint8), self.group).view(torch.int32).reshape(-1, nlist).contiguous()
                q = _as_tensor(Q, torch.float32, dev)
                rows = int(L)
                group = max(1, min(B, self.GATHER_BUDGET // max(w * rows * 20, 1)))       # queries whose every-candidate rows stay under the budget

                def every_candidate(qq):
                    """rows = L for the queries qq: every rank's owned candidates gathered, std::partial_sort replayed on the rebuilt
                    sequences (rii_ivf_shard_replay_ex_dev) -> (ids, dists, counts) of those queries."""
                    nf = int(qq.shape[0])
                    ri = torch.empty((nf, topk), dtype=torch.int64, device=dev)
                    rd = torch.empty((nf, topk), dtype=torch.float32, device=dev)
                    rc = torch.empty((nf,), dtype=torch.int64, device=dev)
                    for f0 in range(0, nf, group):
                        nfc = min(group, nf - f0)
                        fi = torch.empty((nfc, rows), dtype=torch.int64, device=dev)
                        fd = torch.empty((nfc, rows), dtype=torch.float32, device=dev)
                        fp = torch.empty((nfc, rows), dtype=torch.int32, device=dev)
                        fn = torch.empty((nfc,), dtype=torch.int32, device=dev)
                        qc = qq[f0:f0 + nfc].contiguous()
                        self.engine.query_ivf_shard_dev(qc.data_ptr(), nfc, topk, 0 if t is None else t.data_ptr(),
                                                        0 if t is None else t.numel(), S_global, L, N_global, glen.data_ptr(),
                                                        glen.shape[0], rank, fi.data_ptr(), fd.data_ptr(), fp.data_ptr(),
                                                        fn.data_ptr(), rc[f0:f0 + nfc].data_ptr(), sh, rows=rows)
                        g = _all_gather_bytes(_pack([fp.to(torch.int64), torch.where(fi >= 0, fi + self.start, fi), fd]), self.group)
                        nsc = core.ivf_shard_replay_scratch_bytes(nfc, rows)
                        scratch = torch.empty(max(nsc, 16), dtype=torch.uint8, device=dev)
                        core.ivf_shard_replay_dev(g.data_ptr(), g.shape[0], nfc, rows, topk, ri[f0:f0 + nfc].data_ptr(), rd[f0:f0 + nfc].data_ptr(), sh,
                                                  scratch.data_ptr(), nsc)
                    return ri, rd, rc

                if k1 > self.engine.ivf_shard_max_select_rows(L, N_global, S_global):
                    # more rows per query than a launch selects: the collect-all route, the replay IS the answer
                    out_i, out_d, cnt = every_candidate(q)
                    self.last_tie_flags = torch.zeros(B, dtype=torch.bool, device=dev)
                    return _handoff(out_i, out_d, cnt)
                ids = torch.empty((B, k1), dtype=torch.int64, device=dev)
                d = torch.empty((B, k1), dtype=torch.float32, device=dev)
                pos = torch.empty((B, k1), dtype=torch.int32, device=dev)
                nloc = torch.empty((B,), dtype=torch.int32, device=dev)
                cnt = torch.empty((B,), dtype=torch.int64, device=dev)
                self.engine.query_ivf_shard_dev(q.data_ptr(), B, topk, 0 if t is None else t.data_ptr(),
                                                0 if t is None else t.numel(), S_global, L, N_global, glen.data_ptr(),
                                                glen.shape[0], rank, ids.data_ptr(), d.data_ptr(), pos.data_ptr(),
                                                nloc.data_ptr(), cnt.data_ptr(), sh)
                out_i, out_d, cnt = self._merge_ivf(ids, d, pos, cnt, topk)
                flagged = torch.nonzero(self.last_tie_flags).flatten()
                if flagged.numel():                           # exact ties among the k+1 best: replay the heap (same on all ranks)
                    ri, rd, _ = every_candidate(q[flagged].contiguous())
                    out_i[flagged] = ri
                    out_d[flagged] = rd
                out = (out_i, out_d, cnt)
            return _handoff(*out)
        lens = np.asarray(self.engine.ivf_list_lengths(tl), np.int32)
        glen = _all_gather_bytes(torch.from_numpy(lens).view(torch.uint8), self.group).view(torch.int32).reshape(-1, len(lens))
        Qh = np.asarray(Q)
        rows = int(L)
        group = max(1, min(B, self.GATHER_BUDGET // max(w * rows * 20, 1)))

        def every_candidate(Qs):
            """host tensors (gloo): rows = L for the queries Qs, gathered, the engine replays std::partial_sort on the rebuilt sequences"""
            nf = Qs.shape[0]
            ri = np.empty((nf, topk), np.int64)
            rdd = np.empty((nf, topk), np.float32)
            rc = np.empty((nf,), np.int64)
            for f0 in range(0, nf, group):
                fi, fd, fp, _, fc = self.engine.query_ivf_shard(Qs[f0:f0 + group], topk, tl, S_global, L, N_global, glen.numpy(), rank, rows=rows)
                nfc = fi.shape[0]
                fi = torch.from_numpy(np.ascontiguousarray(fi))
                g = _all_gather_bytes(_pack([torch.from_numpy(np.ascontiguousarray(fp)).to(torch.int64),
                                             torch.where(fi >= 0, fi + self.start, fi), torch.from_numpy(np.ascontiguousarray(fd))]), self.group)
                n = nfc * rows
                gp = np.stack([_field(g, r, 0, n, torch.int64).reshape(nfc, rows).numpy() for r in range(g.shape[0])])
                gi = np.stack([_field(g, r, n * 8, n, torch.int64).reshape(nfc, rows).numpy() for r in range(g.shape[0])])
                gd = np.stack([_field(g, r, n * 16, n, torch.float32).reshape(nfc, rows).numpy() for r in range(g.shape[0])])
                ri[f0:f0 + nfc], rdd[f0:f0 + nfc] = self.engine.ivf_shard_replay(gp, gi, gd, topk)
                rc[f0:f0 + nfc] = fc
            return ri, rdd, rc

        max_sel = getattr(self.engine, "ivf_shard_max_select_rows", None)
        if max_sel is not None and k1 > max_sel(L, N_global, S_global):
            ri, rdd, rc = every_candidate(Qh)              # collect-all: the replay is the answer
            self.last_tie_flags = torch.zeros(B, dtype=torch.bool)
            return torch.from_numpy(ri), torch.from_numpy(rdd), torch.from_numpy(rc)
        ids, d, pos, nloc, cnt = self.engine.query_ivf_shard(Qh, topk, tl, S_global, L, N_global, glen.numpy(), rank)
        out_i, out_d, cnt = self._merge_ivf(torch.from_numpy(np.ascontiguousarray(ids)), torch.from_numpy(np.ascontiguousarray(d)),
                                            torch.from_numpy(np.ascontiguousarray(pos)), torch.from_numpy(np.ascontiguousarray(cnt)), topk)
        flagged = torch.nonzero(self.last_tie_flags).flatten()
        if flagged.numel():                                   # host tensors (gloo): same protocol, the engine replays
            ri, rdd, _ = every_candidate(Qh[flagged.numpy()])
            out_i[flagged] = torch.from_numpy(np.ascontiguousarray(ri))
            out_d[flagged] = torch.from_numpy(np.ascontiguousarray(rdd))
        return out_i, out_d, cnt

    def _merge_ivf(self, ids, d, pos, cnt, topk):
        """all-gather of the per-rank (position, global id, dist) rows + merge under (dist, position)."""
        rank, w = world(self.group)
        B, k1 = ids.shape
        dev = ids.device
        gid = torch.where(ids >= 0, ids + self.start, ids)                  # local -> global ids, -1 stays
        pos64 = pos.to(torch.int64)
        if dev.type == "cuda" and w * k1 <= 8192:
            from . import core
            nrec = core.merge_record_bytes(B, k1, True)
            rec = torch.zeros(nrec, dtype=torch.uint8, device=dev)
            n = B * k1
            rec[:n * 8].view(torch.int64).copy_(pos64.reshape(-1))
            rec[n * 8:n * 16].view(torch.int64).copy_(gid.reshape(-1))
            rec[n * 16:n * 20].view(torch.float32).copy_(d.reshape(-1))
            g = _all_gather_bytes(rec, self.group)
            mp = torch.empty((B, k1), dtype=torch.int64, device=dev)
            md = torch.empty((B, k1), dtype=torch.float32, device=dev)
            mi = torch.empty((B, k1), dtype=torch.int64, device=dev)
            core.merge_topk_dev(g.data_ptr(), g.shape[0], B, k1, mp.data_ptr(), md.data_ptr(),
                                torch.cuda.current_stream().cuda_stream, k_out=k1, d_out_payload=mi.data_ptr())
        else:
            g = _all_gather_bytes(_pack([pos64, gid, d]), self.group)
            n = B * k1
            gp = torch.cat([_field(g, r, 0, n, torch.int64).reshape(B, k1) for r in range(g.shape[0])], dim=1)
            gi = torch.cat([_field(g, r, n * 8, n, torch.int64).reshape(B, k1) for r in range(g.shape[0])], dim=1)
            gd = torch.cat([_field(g, r, n * 16, n, torch.float32).reshape(B, k1) for r in range(g.shape[0])], dim=1)
            order = torch.sort(gp, dim=1, stable=True).indices               # secondary key (position) first ...
            d1 = torch.gather(gd, 1, order)
            order2 = torch.sort(d1, dim=1, stable=True).indices              # ... then stable sort on the distance
            sel = torch.gather(order, 1, order2)[:, :k1]
            mi, md = torch.gather(gi, 1, sel), torch.gather(gd, 1, sel)
        found = cnt > 0
        # exact ties among the k+1 best decide nothing for top-1; for top-k they mark the rows whose order may differ from
        # the heap order of std::partial_sort on the concatenated candidate sequence
        tie = (md[:, 1:] == md[:, :-1]) & torch.isfinite(md[:, 1:])
        self.last_tie_flags = (tie.any(dim=1) & found) if topk > 1 else torch.zeros_like(found)
        out_i = torch.where(found[:, None], mi[:, :topk], torch.full_like(mi[:, :topk], -1))
        out_d = torch.where(found[:, None], md[:, :topk], torch.full_like(md[:, :topk], float("inf")))
        return out_i.contiguous(), out_d.contiguous(), cnt


class QueryShardedIndex(object):
    """Query-sharded search over a replicated index: rank r answers rows shard_range(B, r, W) of Q (any B: slices may
    differ by one row)."""

    def __init__(self, engine, group=None):
        self.engine, self.group = engine, group

    def _slices(self, B):
        rank, w = world(self.group)
        rows = [shard_range(B, r, w)[1] - shard_range(B, r, w)[0] for r in range(w)]
        return shard_range(B, rank, w), rows

    def query_linear_batch(self, Q, topk, target_ids=None, out=None):
        (s, e), rows = self._slices(Q.shape[0])
        n = e - s
        if _is_device_engine(self.engine) and _use_c_comm():
            # ONE library call (round 4): this rank's slice through the engine -> ncclAllGather of the packed rows -> unpack kernel
            dev = torch.device("cuda", torch.cuda.current_device())
            comm = get_comm(self.group)
            B = Q.shape[0]
            with _engine_stream() as sh:
                q = _as_tensor(Q, torch.float32, dev)
                t = None if target_ids is None or len(target_ids) == 0 else _as_tensor(target_ids, torch.int64, dev)
                if out is None:
                    out = (torch.empty((B, topk), dtype=torch.int64, device=dev), torch.empty((B, topk), dtype=torch.float32, device=dev))
                comm.query_linear_qsharded_dev(self.engine, q.data_ptr(), B, topk, 0 if t is None else t.data_ptr(),
                                               0 if t is None else t.numel(), out[0].data_ptr(), out[1].data_ptr(), sh)
            return _handoff(*out)
        if _is_device_engine(self.engine):
            dev = _comm_device()
            q = _as_tensor(Q, torch.float32, dev)[s:e].contiguous()
            t = None if target_ids is None or len(target_ids) == 0 else _as_tensor(target_ids, torch.int64, dev)
            with _engine_stream() as sh:
                nmax = int(max(rows))
                if min(rows) == nmax:
                    # even split (the usual case): the engine writes ids and distances straight into this rank's record, ONE
                    # all-gather, and the two outputs are one strided copy each out of the gathered buffer
                    nid = nmax * topk * 8
                    rec_bytes = (nmax * topk * 12 + 15) // 16 * 16
                    rec = torch.empty(rec_bytes, dtype=torch.uint8, device=dev)
                    if n:
                        self.engine.query_linear_dev(q.data_ptr(), n, topk, t.data_ptr() if t is not None else 0,
                                                     0 if t is None else t.numel(), rec.data_ptr(), rec.data_ptr() + nid, sh)
                    g = _all_gather_bytes(rec, self.group)
                    wg = g.shape[0]
                    out = (g[:, :nid].view(torch.int64).reshape(wg * nmax, topk),
                           g[:, nid:nid + nmax * topk * 4].view(torch.float32).reshape(wg * nmax, topk))
                else:
                    ids = torch.empty((n, topk), dtype=torch.int64, device=dev)
                    d = torch.empty((n, topk), dtype=torch.float32, device=dev)
                    if n:
                        self.engine.query_linear_dev(q.data_ptr(), n, topk, t.data_ptr() if t is not None else 0,
                                                     0 if t is None else t.numel(), ids.data_ptr(), d.data_ptr(), sh)
                    out = allgather_query_shards(ids, d, self.group, rows=rows)
            return _handoff(*out)
        else:
            Qh = np.asarray(Q)
            ids, d = self.engine.query_linear_batch(np.ascontiguousarray(Qh[s:e]), topk, target_ids) if n else \
                (np.zeros((0, topk), np.int64), np.zeros((0, topk), np.float32))
        return allgather_query_shards(ids, d, self.group, rows=rows)

    def query_ivf_batch(self, Q, topk, target_ids, L):
        """Inverted-index search, query-sharded (the reference's "stop at exactly L candidates in list order" rule is a
        per-query sequential rule, so the index is replicated and the queries are split).
        Returns (ids [B,topk], dists [B,topk], counts [B]) on every rank."""
        (s, e), rows = self._slices(Q.shape[0])
        n = e - s
        if _is_device_engine(self.engine) and _use_c_comm():
            dev = torch.device("cuda", torch.cuda.current_device())
            comm = get_comm(self.group)
            B = Q.shape[0]
            with _engine_stream() as sh:
                q = _as_tensor(Q, torch.float32, dev)
                t = None if target_ids is None or len(target_ids) == 0 else _as_tensor(target_ids, torch.int64, dev)
                ids = torch.empty((B, topk), dtype=torch.int64, device=dev)
                d = torch.empty((B, topk), dtype=torch.float32, device=dev)
                cnt = torch.empty((B,), dtype=torch.int64, device=dev)
                comm.query_ivf_qsharded_dev(self.engine, q.data_ptr(), B, topk, 0 if t is None else t.data_ptr(),
                                            0 if t is None else t.numel(), L, ids.data_ptr(), d.data_ptr(), cnt.data_ptr(), sh)
            return _handoff(ids, d, cnt)
        if _is_device_engine(self.engine):
            dev = _comm_device()
            q = _as_tensor(Q, torch.float32, dev)[s:e].contiguous()
            t = None if target_ids is None or len(target_ids) == 0 else _as_tensor(target_ids, torch.int64, dev)
            with _engine_stream() as sh:
                ids = torch.empty((n, topk), dtype=torch.int64, device=dev)
                d = torch.empty((n, topk), dtype=torch.float32, device=dev)
                cnt = torch.zeros((n,), dtype=torch.int64, device=dev)
                if n:
                    self.engine.query_ivf_dev(q.data_ptr(), n, topk, t.data_ptr() if t is not None else 0,
                                              0 if t is None else t.numel(), L, ids.data_ptr(), d.data_ptr(),
                                              cnt.data_ptr(), sh)
                out = allgather_query_shards(ids, d, self.group, local_counts=cnt, rows=rows)
            return _handoff(*out)
        else:
            Qh = np.asarray(Q)
            if n:
                ids, d, cnt = self.engine.query_ivf_batch(np.ascontiguousarray(Qh[s:e]), topk, target_ids, L)
            else:
                ids, d, cnt = np.zeros((0, topk), np.int64), np.zeros((0, topk), np.float32), np.zeros(0, np.int64)
        return allgather_query_shards(ids, d, self.group, local_counts=np.asarray(cnt, np.int64)
                                      if not isinstance(cnt, torch.Tensor) else cnt, rows=rows)
