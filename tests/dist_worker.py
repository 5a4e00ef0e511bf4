"""Worker of tests/test_gpu_dist.py::test_two_process_rccl_world (one process per GPU under torchrun): the exchange steps of the
corpus-sharded search on a REAL world of N ranks -- ragged query all-gather, sharded score + top-k with global index bases,
top-k exchange + merge -- each against the single-rank answer computed locally from the same seeded data.  Prints one
line `DIST_WORKER_OK rank=R world=N` per rank.  (the first multi-GPU box exercises this by default.)"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    from sgpt_amd import get_context
    from sgpt_amd.dist import balanced_cuts, get_comm, shard_range, sharded_score_topk
    ctx = get_context(dev)
    comm = get_comm(ctx)
    assert comm.world == world and comm.rank == rank
    g = torch.Generator(device="cpu").manual_seed(5)                 # the same data on every rank
    nq, N, d, k = 203, 40_003, 768, 11
    q = torch.nn.functional.normalize(torch.randn(nq, d, generator=g), dim=1).to(dev)
    c = torch.nn.functional.normalize(torch.randn(N, d, generator=g), dim=1)
    c[30_000:30_050] = c[50:100]                                      # equal scores on different ranks: ties -> lowest index
    c = c.to(dev).to(torch.float16)
    # 1. ragged all-gather: unequal contiguous query slices (cuts balanced on a skewed weight), fp32 rows and int64 rows
    w = np.ones(nq, dtype=np.int64)
    w[:7] = 40
    cuts = balanced_cuts(w.tolist(), world, min_one=True)
    counts = np.diff(cuts).tolist()
    assert len(set(counts)) > 1 or world == 1
    q_all = comm.all_gather_rows(q[cuts[rank]: cuts[rank + 1]].contiguous(), counts)
    assert torch.equal(q_all, q), "ragged all-gather of query rows"
    ids = torch.arange(nq * 3, dtype=torch.int64, device=dev).reshape(nq, 3)
    assert torch.equal(comm.all_gather_rows(ids[cuts[rank]: cuts[rank + 1]].contiguous(), counts), ids)
    eq = [nq // world + (1 if r < nq % world else 0) for r in range(world)]
    lo_q = sum(eq[:rank])
    assert torch.equal(comm.all_gather_rows(q[lo_q: lo_q + eq[rank]].contiguous(), eq, padded=True), q)
    # 2. the sharded search: every rank scores its contiguous document range, the lists are exchanged and merged
    lo, hi = shard_range(N, rank, world)
    excl = torch.full((nq,), -1, dtype=torch.int64)
    excl[9] = 77
    fv, fi = sharded_score_topk(ctx, q[lo_q: lo_q + eq[rank]].contiguous(), nq, c[lo:hi].contiguous(), k, idx_base=lo,
                                exclude_idx=excl, dtype=torch.float16)
    wv, wi, _ = ctx.score_topk(q, c, k + 1, dtype=torch.float16)
    mv, mi = ctx.topk_merge(wv, wi, k, exclude_idx=excl)
    keep = [r for r in range(nq) if r != 9]
    assert torch.equal(fi[keep], mi[keep]) and torch.equal(fv[keep], mv[keep]), "sharded search != single-rank search"
    assert (fi[9] != 77).all()
    # every rank holds the same answer
    chk = fi.clone()
    dist.broadcast(chk, src=0)
    assert torch.equal(chk, fi)
    torch.cuda.synchronize()
    print(f"DIST_WORKER_OK rank={rank} world={world}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
