"""Multi-GPU retrieval: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

The path shards twice:
  * encode is embarrassingly parallel -- weights replicated, sentences split in contiguous shards
    (SentenceTransformer.py:159-163), no communication;
  * search needs ONE exchange: the corpus is partitioned by contiguous document range, each rank
    keeps the embeddings of its own range in its own HBM (never gathered); queries are encoded
    sharded, all-gathered once (fp32 [nq/world, d] per rank), scored against the local shard with a
    global index base, and the per-rank top-k lists are all-gathered and merged.
The reference has no corpus sharding (exact_search.py scores on one device); its only collectives
on the inference path are the two all-gathers of util.mismatched_sizes_all_gather (util.py:326-347),
replaced here by a single equal-size all-gather because shard sizes are a pure function of (n, world).
"""
from typing import List, Optional, Tuple

import torch

from .st import all_gather_rows, shard_sizes


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of rank `rank` when n items are split like SentenceTransformer.py:159-163."""
    sizes = shard_sizes(n, world)
    lo = sum(sizes[:rank])
    return lo, lo + sizes[rank]


def all_gather_queries(local_q: torch.Tensor, nq_total: int, group=None) -> torch.Tensor:
    """Every rank contributes its contiguous slice of the query embeddings; everyone gets Q[nq_total, d]."""
    import torch.distributed as dist
    return all_gather_rows(local_q, shard_sizes(nq_total, dist.get_world_size(group)), group)


def exchange_topk(val: torch.Tensor, idx: torch.Tensor, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """All-gather the per-rank [nq, k] candidate lists -> [nq, world*k] (rank-major inside a row)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    val, idx = val.contiguous(), idx.contiguous()
    nq, k = val.shape
    gv = torch.empty((world * nq, k), dtype=val.dtype, device=val.device)     # rank-major concatenation
    gi = torch.empty((world * nq, k), dtype=idx.dtype, device=idx.device)
    dist.all_gather_into_tensor(gv, val, group=group)
    dist.all_gather_into_tensor(gi, idx, group=group)
    return (gv.view(world, nq, k).permute(1, 0, 2).reshape(nq, world * k).contiguous(),
            gi.view(world, nq, k).permute(1, 0, 2).reshape(nq, world * k).contiguous())


def sharded_score_topk(ctx, q_local: torch.Tensor, nq_total: int, corpus_shard: torch.Tensor, k: int,
                       idx_base: int, exclude_idx: Optional[torch.Tensor] = None, group=None, dtype=None):
    """One sharded search: all-gather queries -> local fused score+top-k (HIP) -> exchange -> merge (HIP).
    Returns (values[nq_total,k], global doc indices[nq_total,k]) identical on every rank."""
    q_all = all_gather_queries(q_local, nq_total, group)
    val, idx, _ = ctx.score_topk(q_all, corpus_shard, k, idx_base=idx_base, dtype=dtype)
    cv, ci = exchange_topk(val, idx, group)
    return ctx.topk_merge(cv, ci, k, exclude_idx=exclude_idx)
