"""Multi-GPU retrieval: one process per GPU, the exchange steps on RCCL over xGMI through the C ABI
(include/sgpt_hip.h: sgpt_comm_init / sgpt_allgather_rows / sgpt_exchange_topk, sgpt_amd/csrc/comm.hip).

The path shards twice:
  * encode is embarrassingly parallel -- weights replicated, sentences split in contiguous, TOKEN-balanced shards of the
    length-sorted list (the reference cuts equal sentence counts, SentenceTransformer.py:159-163, which hands rank 0 the
    longest documents), no communication;
  * search needs ONE exchange: the corpus is partitioned by contiguous document range, each rank keeps the embeddings of
    its own range in its own HBM (never gathered); queries are encoded sharded, all-gathered once (fp32 [nq_r, d] per
    rank), scored against the local shard with a global index base, and the per-rank top-k lists are all-gathered and
    merged on the device.
The reference has no corpus sharding (exact_search.py scores on one device); its only collectives on the inference path
are the two all-gathers of util.mismatched_sizes_all_gather (util.py:326-347), replaced here by a single all-gather
because every rank can compute every rank's row count.

torch.distributed is the BOOTSTRAP only: it carries the 128-byte RCCL unique id from rank 0 to the others (one broadcast
per process group).  `TorchComm` (gloo, CPU tensors) exists for the world-size-2 CPU tests of this plumbing, where no HIP
device -- hence no RCCL -- is available; a real `Context` always takes `RcclComm`.
"""
import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch


def shard_sizes(n: int, world_size: int) -> List[int]:
    """Contiguous equal-count shard sizes, the rule of SentenceTransformer.encode (:159-160)."""
    return [n // world_size + (1 if r < n % world_size else 0) for r in range(world_size)]


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of rank `rank` when n items are split like SentenceTransformer.py:159-163."""
    sizes = shard_sizes(n, world)
    lo = sum(sizes[:rank])
    return lo, lo + sizes[rank]


def balanced_cuts(weights: Sequence[int], world: int, min_one: bool = False) -> np.ndarray:
    """Cut points c[0] = 0 <= c[1] <= ... <= c[world] = n of a list into `world` CONTIGUOUS ranges of (nearly) equal total
    weight: rank r owns items [c[r], c[r+1]).  The weights are per-item costs (token rows of the packed layout, or
    character counts as their proxy); contiguity keeps the length-sorted order inside a shard ("balance by
    tokens, not by sentences" -- with equal counts of a longest-first list, rank 0 of 8 carries ~1.7x the mean token load).
    Every cut lies within one item of its ideal position: max load <= mean + the heaviest item.  Pure function of
    (weights, world): every rank computes the same cuts without communication.
    min_one: with n >= world items no range is left empty (one dominant item otherwise pulls several cuts onto the same
    position: 32 queries, the first one 400 characters long, on 8 ranks gave [0 8 17 20 20 21 21 24 32]) -- for callers
    whose per-rank step cannot run on zero items.  Balance degrades by at most one item per rank."""
    w = np.asarray(weights, dtype=np.int64)
    n = int(w.shape[0])
    cuts = np.zeros(world + 1, dtype=np.int64)
    cuts[world] = n
    if n == 0:
        return cuts
    cum = np.concatenate([[0], np.cumsum(w)])                 # cum[i] = weight of items [0, i)
    total = int(cum[-1])
    for r in range(1, world):
        target = total * r / world
        i = int(np.searchsorted(cum, target, side="left"))   # first i with cum[i] >= target
        if i > 0 and target - cum[i - 1] < cum[i] - target:  # nearer cut
            i -= 1
        cuts[r] = min(max(i, int(cuts[r - 1])), n)
    if min_one and n >= world:
        for r in range(1, world):                            # forward: every range starts after its predecessor's first item
            cuts[r] = max(int(cuts[r]), int(cuts[r - 1]) + 1)
        for r in range(world - 1, 0, -1):                    # backward: ... and leaves one item for every successor
            cuts[r] = min(int(cuts[r]), int(cuts[r + 1]) - 1)
    return cuts


# ---------------------------------------------------------------------------------------------------------------------
class RcclComm:
    """The product transport: one RCCL communicator per context, created through the C ABI."""

    def __init__(self, ctx, group=None):
        import torch.distributed as dist
        from . import _lib
        self.ctx, self.group = ctx, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        lib = ctx.lib
        have = int(lib.sgpt_comm_world(ctx.handle))
        owner = ctx.__dict__.get("_comm_group_key", "unset")
        key = id(group) if group is not None else None
        if have:                                   # this context already carries a communicator (one per ctx)
            # world and rank matching is not enough: two different groups of the same size would silently share one
            # communicator (collectives of one group paired with another group's ranks)
            if have != self.world or int(lib.sgpt_comm_rank(ctx.handle)) != self.rank or owner not in ("unset", key):
                raise _lib.SgptHipError("the context's RCCL communicator belongs to a different process group "
                                        "(one communicator per context: use a second Context for a second group)")
            ctx.__dict__["_comm_group_key"] = key
            return
        # bootstrap: rank 0's unique id travels over whatever backend torch.distributed was initialised with
        on_dev = dist.get_backend(group) == "nccl"
        buf = torch.zeros(_lib.SGPT_COMM_ID_BYTES, dtype=torch.uint8)
        if self.rank == 0:
            raw = (C.c_uint8 * _lib.SGPT_COMM_ID_BYTES)()
            _lib.check(ctx.handle, lib.sgpt_comm_unique_id(raw), "sgpt_comm_unique_id")
            buf = torch.frombuffer(bytearray(bytes(raw)), dtype=torch.uint8).clone()
        if on_dev:
            buf = buf.to(ctx.device)
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast(buf, src=src, group=group)
        ident = (C.c_uint8 * _lib.SGPT_COMM_ID_BYTES)(*buf.cpu().tolist())
        _lib.check(ctx.handle, lib.sgpt_comm_init(ctx.handle, ident, self.rank, self.world), "sgpt_comm_init")
        ctx.__dict__["_comm_group_key"] = key

    def all_gather_rows(self, local: torch.Tensor, counts: Sequence[int], padded: bool = False) -> torch.Tensor:
        """Rank r contributes counts[r] rows; everyone gets the concatenation in rank order (one ncclAllGather).
        padded=True: through the ragged branch (pad, gather, compact) whatever the counts are -- tests of that path."""
        from . import _lib
        from .runtime import _p, _stream_ptr
        ctx = self.ctx
        local = local.to(ctx.device).contiguous()
        if local.shape[0] != counts[self.rank]:
            raise ValueError(f"rank {self.rank} holds {local.shape[0]} rows, the shard plan says {counts[self.rank]}")
        tail = tuple(local.shape[1:])
        row_bytes = int(np.prod(tail, dtype=np.int64)) * local.element_size()
        out = torch.empty((int(sum(counts)),) + tail, dtype=local.dtype, device=ctx.device)
        cnt = (C.c_int64 * self.world)(*[int(c) for c in counts])
        fn = ctx.lib.sgpt_allgather_rows_padded if padded else ctx.lib.sgpt_allgather_rows
        _lib.check(ctx.handle, fn(ctx.handle, _p(local), cnt, row_bytes, _p(out), _stream_ptr(ctx.device)), "sgpt_allgather_rows")
        return out

    def exchange_topk(self, val: torch.Tensor, idx: torch.Tensor, k_out: int,
                      exclude_idx: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """Per-rank [nq, k] (score, global index) lists -> the k_out best of all ranks' candidates, identical everywhere."""
        from . import _lib
        from .runtime import _p, _stream_ptr
        ctx = self.ctx
        val = val.to(device=ctx.device, dtype=torch.float32).contiguous()
        idx = idx.to(device=ctx.device, dtype=torch.int64).contiguous()
        nq, k = val.shape
        ov = torch.empty((nq, k_out), dtype=torch.float32, device=ctx.device)
        oi = torch.empty((nq, k_out), dtype=torch.int64, device=ctx.device)
        ex = None if exclude_idx is None else exclude_idx.to(device=ctx.device, dtype=torch.int64).contiguous()
        _lib.check(ctx.handle, ctx.lib.sgpt_exchange_topk(ctx.handle, _p(val), _p(idx), nq, k, k_out, _p(ex), _p(ov), _p(oi),
                                                          _stream_ptr(ctx.device)), "sgpt_exchange_topk")
        return ov, oi


class TorchComm:
    """gloo stand-in with the same two calls, for the CPU tests of the sharding logic (tests/test_dist_gloo.py), where the
    per-rank scorer / merge are numpy stubs as well.  Never selected for a real Context."""

    def __init__(self, ctx, group=None):
        import torch.distributed as dist
        self.ctx, self.group = ctx, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)

    def all_gather_rows(self, local: torch.Tensor, counts: Sequence[int]) -> torch.Tensor:
        import torch.distributed as dist
        mx = max(counts)
        pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
        out = torch.empty((self.world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, pad, group=self.group)
        return torch.cat([out[r * mx: r * mx + counts[r]] for r in range(self.world)], dim=0)

    def exchange_topk(self, val, idx, k_out, exclude_idx=None):
        import torch.distributed as dist
        val, idx = val.contiguous(), idx.contiguous()
        nq, k = val.shape
        gv = torch.empty((self.world * nq, k), dtype=val.dtype, device=val.device)     # rank-major concatenation
        gi = torch.empty((self.world * nq, k), dtype=idx.dtype, device=idx.device)
        dist.all_gather_into_tensor(gv, val, group=self.group)
        dist.all_gather_into_tensor(gi, idx, group=self.group)
        cv = gv.view(self.world, nq, k).permute(1, 0, 2).reshape(nq, self.world * k).contiguous()
        ci = gi.view(self.world, nq, k).permute(1, 0, 2).reshape(nq, self.world * k).contiguous()
        return self.ctx.topk_merge(cv, ci, k_out, exclude_idx=exclude_idx)


def max_reducer(ctx, group=None):
    """flags int32[n] -> elementwise max over the ranks of `group`, as a numpy array (SGPTModel.sync_precision's `reduce`).
    A control-plane exchange of a few hundred bytes on the bootstrap transport (torch.distributed), not a data-path collective."""
    import torch.distributed as dist

    def reduce(flags: np.ndarray) -> np.ndarray:
        t = torch.from_numpy(np.ascontiguousarray(flags, dtype=np.int32))
        if dist.get_backend(group) == "nccl":
            t = t.to(ctx.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        return t.cpu().numpy()
    return reduce


def is_distributed(group=None) -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def get_comm(ctx, group=None):
    """The communicator of `ctx` for `group` (created on first use: collective).  A real Context gets RcclComm."""
    cache = ctx.__dict__.setdefault("_comms", {})
    key = id(group) if group is not None else None
    if key not in cache:
        from .runtime import Context
        cache[key] = RcclComm(ctx, group) if isinstance(ctx, Context) else TorchComm(ctx, group)
    return cache[key]


# ---- the search exchange steps ----------------------------------------------------------------------------------------
def all_gather_queries(ctx, local_q: torch.Tensor, counts: Sequence[int], group=None) -> torch.Tensor:
    """Every rank contributes its contiguous slice of the query embeddings (counts[r] rows); everyone gets Q[sum, d]."""
    return get_comm(ctx, group).all_gather_rows(local_q, counts)


def exchange_topk(ctx, val: torch.Tensor, idx: torch.Tensor, k_out: Optional[int] = None,
                  exclude_idx: Optional[torch.Tensor] = None, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    return get_comm(ctx, group).exchange_topk(val, idx, val.shape[1] if k_out is None else k_out, exclude_idx)


def sharded_score_topk(ctx, q_local: torch.Tensor, nq_total: int, corpus_shard: torch.Tensor, k: int,
                       idx_base: int, exclude_idx: Optional[torch.Tensor] = None, group=None, dtype=None):
    """One sharded search: all-gather queries -> local fused score+top-k (HIP) -> exchange + merge (RCCL + HIP).
    Returns (values[nq_total,k], global doc indices[nq_total,k]) identical on every rank."""
    comm = get_comm(ctx, group)
    q_all = comm.all_gather_rows(q_local, shard_sizes(nq_total, comm.world))
    val, idx, _ = ctx.score_topk(q_all, corpus_shard, k, idx_base=idx_base, dtype=dtype)
    return comm.exchange_topk(val, idx, k, exclude_idx)
