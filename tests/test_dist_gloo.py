"""world_size-2 gloo test (CPU) of the N>1 plumbing: shard ranges, the query all-gather and the
top-k exchange.  The HIP kernels cannot run here, so the per-rank scorer and the final merge are
stood in by numpy IN THIS TEST ONLY; what is verified is that sharding + collectives + merge give
exactly the single-process oracle answer."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import sgpt_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nq, N, d, k, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sgpt_amd.dist import all_gather_queries, exchange_topk, shard_range
        rng = np.random.default_rng(0)                      # same data on both ranks
        q = rng.standard_normal((nq, d)).astype(np.float32)
        c = rng.standard_normal((N, d)).astype(np.float32)
        qlo, qhi = shard_range(nq, rank, world)
        clo, chi = shard_range(N, rank, world)
        # each rank "encodes" only its query slice, then the RCCL(gloo here) all-gather
        q_all = all_gather_queries(torch.from_numpy(q[qlo:qhi]), nq)
        assert torch.equal(q_all, torch.from_numpy(q))
        # local scorer stand-in (numpy): top-k of the local corpus shard with a global index base
        sc = O.cos_sim(q_all.numpy(), c[clo:chi])
        v, i = O.topk_rows(sc, k)
        cv, ci = exchange_topk(torch.from_numpy(v), torch.from_numpy(i + clo))
        assert cv.shape == (nq, world * k)
        # merge stand-in (numpy): k best of the gathered candidates
        order = np.argsort(-cv.numpy(), axis=1, kind="stable")[:, :k]
        fv = np.take_along_axis(cv.numpy(), order, 1)
        fi = np.take_along_axis(ci.numpy(), order, 1)
        wv, wi = O.topk_rows(O.cos_sim(q, c), k)
        ok = np.array_equal(fi, wi) and np.allclose(fv, wv, atol=1e-6)
        out.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_sharded_search_equals_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 37, 501, 32, 7, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = [out.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(res) == [(0, True), (1, True)]


def test_shard_range_partition():
    from sgpt_amd.dist import shard_range
    for n, w in [(10, 4), (7, 8), (1000, 8), (5, 1), (0, 2)]:
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
