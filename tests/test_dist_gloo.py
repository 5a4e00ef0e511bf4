"""world_size-2 gloo tests (CPU) of the N>1 plumbing.  The HIP kernels cannot run here, so the per-rank
scorer / merge / encoder are stood in by numpy stubs IN THESE TESTS ONLY; what is verified is that the
product's own sharding + collectives + merge code (sgpt_amd.dist.sharded_score_topk, exchange_topk,
all_gather_queries; SentenceTransformerSGPT.encode's torch.distributed branch) gives exactly the
single-process answer.  The same functions run on RCCL in tests/test_gpu_dist.py."""
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


class StubCtx:
    """numpy stand-in for sgpt_amd.runtime.Context with the two calls sharded_score_topk makes."""

    def score_topk(self, q, corpus, k, idx_base=0, run=None, dtype=None):
        sc = O.cos_sim(q.numpy(), corpus.numpy())
        v, i = O.topk_rows(sc, k)
        return torch.from_numpy(v), torch.from_numpy(i + idx_base), k

    def topk_merge(self, val, idx, k, exclude_idx=None):
        v, i = val.numpy().copy(), idx.numpy()
        if exclude_idx is not None:
            v[i == exclude_idx.numpy()[:, None]] = -np.inf
        # descending score, ties -> ascending index (the contract of sgpt_topk_merge)
        order = np.lexsort((i, -v), axis=1)[:, :k]
        return torch.from_numpy(np.take_along_axis(v, order, 1)), torch.from_numpy(np.take_along_axis(i, order, 1))


class FakeEncoder:
    """Deterministic stand-in for SGPTModel: the embedding of a sentence depends only on its ids."""
    device = "cpu"

    class cfg:
        hidden_size = 8
        max_position_embeddings = 64

    calls = 0

    def encode_ids(self, seqs, mode="weightedmean", normalize=False, **kw):
        FakeEncoder.calls += 1
        out = np.zeros((len(seqs), 8), dtype=np.float32)
        for r, s in enumerate(seqs):
            a = np.asarray(s, dtype=np.float64)
            out[r] = [np.sin(0.37 * j + a.sum() * 1e-3) + len(s) * 0.01 * j for j in range(8)]
        t = torch.from_numpy(out)
        return torch.nn.functional.normalize(t, dim=1) if normalize else t


def _run(worker, world, *args):
    port = _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, out) + args) for r in range(world)]
    for p in procs:
        p.start()
    res = [out.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(30)
    return sorted(res)


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _search_worker(rank, world, port, out, nq, N, d, k):
    _init(rank, world, port)
    try:
        from sgpt_amd.dist import all_gather_queries, exchange_topk, shard_range, sharded_score_topk
        rng = np.random.default_rng(0)                      # same data on both ranks
        q = rng.standard_normal((nq, d)).astype(np.float32)
        c = rng.standard_normal((N, d)).astype(np.float32)
        c[7] = c[3]                                         # a duplicated document: tie -> the lower index wins
        qlo, qhi = shard_range(nq, rank, world)
        clo, chi = shard_range(N, rank, world)
        q_all = all_gather_queries(torch.from_numpy(q[qlo:qhi]), nq)
        ok = torch.equal(q_all, torch.from_numpy(q))
        # the product's sharded search, end to end, with the self-match rule on two queries
        excl = np.full(nq, -1, dtype=np.int64)
        excl[2], excl[5] = 11, N - 1
        fv, fi = sharded_score_topk(StubCtx(), torch.from_numpy(q[qlo:qhi]), nq, torch.from_numpy(c[clo:chi]), k,
                                    idx_base=clo, exclude_idx=torch.from_numpy(excl))
        sc = O.cos_sim(q, c)
        sc[np.arange(nq), excl] = np.where(excl >= 0, -np.inf, sc[np.arange(nq), excl])
        wv, wi = O.topk_rows(sc, k)
        ok = ok and np.array_equal(fi.numpy(), wi) and np.allclose(fv.numpy(), wv, atol=1e-6)
        # the exchange alone: [nq, world*k], rank-major inside a row
        v, i = O.topk_rows(O.cos_sim(q, c[clo:chi]), k)
        cv, ci = exchange_topk(torch.from_numpy(v), torch.from_numpy(i + clo))
        ok = ok and cv.shape == (nq, world * k) and np.array_equal(ci[:, rank * k:(rank + 1) * k].numpy(), i + clo)
        out.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_sharded_search_equals_single_process():
    assert _run(_search_worker, 2, 37, 501, 32, 7) == [(0, True), (1, True)]


def _encode_worker(rank, world, port, out, n):
    _init(rank, world, port)
    try:
        from sgpt_amd.st import SentenceTransformerSGPT
        from sgpt_amd.tokenization import SyntheticTokenizer
        rng = np.random.default_rng(3)
        words = ["alpha", "beta", "gamma", "delta", "eps", "zeta", "eta", "theta"]
        sents = [" ".join(rng.choice(words, size=int(rng.integers(1, 30)))) for _ in range(n)]
        st = SentenceTransformerSGPT(FakeEncoder(), SyntheticTokenizer(), max_seq_length=64, pooling_mode="weightedmean")
        emb = st.encode(sents, convert_to_tensor=True, normalize_embeddings=True)        # distributed branch (world 2)
        want = FakeEncoder().encode_ids(st.pipe.batch(sents, True), normalize=True)     # single process, input order
        ok = emb.shape == want.shape and torch.allclose(emb, want, atol=0, rtol=0)
        # every rank returns the full matrix, and each rank only encoded its own shard
        out.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("n", [23, 1])
def test_two_rank_encode_shards_and_unsorts(n):
    """SentenceTransformerSGPT.encode under torch.distributed (SentenceTransformer.py:153-175): length-sorted
    contiguous shards, one all-gather, un-sort -- equals the single-process embeddings row for row (n = 1: one rank
    gets an empty shard)."""
    assert _run(_encode_worker, 2, n) == [(0, True), (1, True)]


def test_shard_range_partition():
    from sgpt_amd.dist import shard_range
    for n, w in [(10, 4), (7, 8), (1000, 8), (5, 1), (0, 2)]:
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
