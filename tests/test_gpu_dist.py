"""-m gpu: the multi-GPU code paths on RCCL through the C ABI (sgpt_comm_init / sgpt_allgather_rows / sgpt_exchange_topk,
sgpt_amd/csrc/comm.hip).  One GPU is enough to prove that librccl loads into the process next to torch's, that the
communicator comes up from a unique id bootstrapped over torch.distributed, that the collectives run on device tensors
on the caller's stream, and that the product's sharding code (sgpt_amd.dist, SentenceTransformerSGPT.encode_ids_distributed,
DenseRetrievalExactSearch.search) gives the single-process result: the process group has world_size 1, so every
collective degenerates to a copy through RCCL.  The same functions run with world_size 2 on gloo in
tests/test_dist_gloo.py; bench.py --gpus N is the N-rank run."""
import os
import socket

import numpy as np
import pytest
import torch

from oracle import sgpt_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_group():
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()
    for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        os.environ.pop(k, None)


def test_rccl_sharded_search_equals_local_search(nccl_group):
    from sgpt_amd import get_context
    from sgpt_amd.dist import RcclComm, all_gather_queries, exchange_topk, get_comm, sharded_score_topk
    ctx = get_context("cuda:0")
    comm = get_comm(ctx)
    assert isinstance(comm, RcclComm) and ctx.lib.sgpt_comm_world(ctx.handle) == 1 and ctx.lib.sgpt_comm_rank(ctx.handle) == 0
    g = torch.Generator(device="cpu").manual_seed(3)
    nq, N, d, k = 100, 20_000, 768, 11
    q = torch.nn.functional.normalize(torch.randn(nq, d, generator=g), dim=1).cuda()
    c = torch.nn.functional.normalize(torch.randn(N, d, generator=g), dim=1).cuda().to(torch.float16)
    assert nccl_group.get_backend() == "nccl"
    q_all = all_gather_queries(ctx, q, [nq])                           # ncclAllGather through sgpt_allgather_rows
    assert torch.equal(q_all, q) and q_all.data_ptr() != q.data_ptr()
    i64 = torch.arange(7 * 3, dtype=torch.int64, device="cuda").reshape(7, 3)
    assert torch.equal(comm.all_gather_rows(i64, [7]), i64)           # any row type: bytes
    excl = torch.full((nq,), -1, dtype=torch.int64)
    excl[4] = 777 + 12
    fv, fi = sharded_score_topk(ctx, q, nq, c, k, idx_base=777, exclude_idx=excl, dtype=torch.float16)
    wv, wi, _ = ctx.score_topk(q, c, k, idx_base=777, dtype=torch.float16)
    mv, mi = ctx.topk_merge(wv, wi, k, exclude_idx=excl)
    assert torch.equal(fi, mi) and torch.equal(fv, mv)
    assert (fi[4] != 789).all()
    cv, ci = exchange_topk(ctx, wv, wi)                               # gather + merge, no exclusion: the list itself
    assert torch.equal(cv, wv) and torch.equal(ci, wi)
    cv, ci = exchange_topk(ctx, wv, wi, k_out=5, exclude_idx=excl)    # fewer columns out, exclusion in the merge
    assert torch.equal(cv, mv[:, :5]) and torch.equal(ci, mi[:, :5])
    with pytest.raises(ValueError):
        comm.all_gather_rows(q[:3], [nq])                             # the shard plan and the local rows disagree
    # the RAGGED branch of sgpt_allgather_rows (pad to the largest block, gather into the exchange workspace, compact in
    # rank order) -- unequal shards take it; a world of one reaches it through the padded entry point
    assert torch.equal(comm.all_gather_rows(q, [nq], padded=True), q)
    assert torch.equal(comm.all_gather_rows(i64, [7], padded=True), i64)
    big = torch.randn(5000, 96, device="cuda")                         # workspace growth inside the padded path
    assert torch.equal(comm.all_gather_rows(big, [5000], padded=True), big)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_fold_of_a_simulated_world_equals_the_single_shard_search(world):
    """The device code that only runs with more than one rank -- the [world][nq][k] -> [nq][world * k] regrouping kernel in
    front of the merge (sgpt_amd/csrc/comm.hip) -- driven on one GPU: a corpus cut into `world` contiguous shards, one
    sgpt_score_topk per shard with its global index base, the lists stacked in rank order exactly as ncclAllGather
    delivers them, folded by sgpt_fold_gathered_topk (the rank-local half of sgpt_exchange_topk).  Must equal the
    single-shard search, values and indices, including the self-id exclusion and ties across shards."""
    from sgpt_amd import get_context
    from sgpt_amd.dist import shard_range
    ctx = get_context("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(100 + world)
    nq, N, d, k = 37, 9_000, 256, 11
    q = torch.nn.functional.normalize(torch.randn(nq, d, generator=g), dim=1).cuda().to(torch.float16)
    c = torch.nn.functional.normalize(torch.randn(N, d, generator=g), dim=1)
    c[5_000:5_040] = c[100:140]                                   # equal scores in different shards: ties -> lowest index
    c = c.cuda().to(torch.float16)
    excl = torch.full((nq,), -1, dtype=torch.int64)
    excl[3] = 17
    vals, idxs = [], []
    for r in range(world):
        lo, hi = shard_range(N, r, world)
        v, i, n = ctx.score_topk(q, c[lo:hi].contiguous(), k, idx_base=lo, dtype=torch.float16)
        assert n == k
        vals.append(v); idxs.append(i)
    fv, fi = ctx.fold_gathered_topk(torch.stack(vals), torch.stack(idxs), k, exclude_idx=excl)
    wv, wi, _ = ctx.score_topk(q, c, k + 1, dtype=torch.float16)
    mv, mi = ctx.topk_merge(wv, wi, k, exclude_idx=excl)
    keep = [r for r in range(nq) if r != 3]                        # (row 3 lost a candidate to the exclusion: k of k + 1 vs k of world * k)
    assert torch.equal(fi[keep], mi[keep]) and torch.equal(fv[keep], mv[keep])
    assert (fi[3] != 17).all() and torch.equal(fv[3, : k - 1], mv[3, : k - 1])
    ref_v, ref_i = torch.topk(q.float() @ c.float().T, k, dim=1)
    assert float((fv[keep] - ref_v[keep]).abs().max()) < 1e-3


def test_rccl_distributed_encode_equals_local_encode(nccl_group):
    from helpers import build_model
    from sgpt_amd.st import SentenceTransformerSGPT
    from sgpt_amd.tokenization import SyntheticTokenizer
    kw = dict(vocab_size=211, max_position_embeddings=96, hidden_size=128, num_layers=4, num_heads=2, window_size=8)
    m = build_model(kw, 11, 0.08, "f16")
    st = SentenceTransformerSGPT(m, SyntheticTokenizer(211), max_seq_length=64)
    rng = np.random.default_rng(9)
    words = ["alpha", "beta", "gamma", "delta", "eps", "zeta", "eta", "theta"]
    sents = [" ".join(rng.choice(words, size=int(rng.integers(1, 40)))) for _ in range(57)]
    seqs = st.pipe.batch(sents, True)
    got = st.encode_ids_distributed(seqs, normalize_embeddings=True)   # shard -> encode -> RCCL all-gather -> un-sort
    want = m.encode_ids(seqs, normalize=True)
    assert got.is_cuda and got.shape == want.shape and float((got - want).abs().max()) < 1e-6
    ref = O.encode(O.synth_weights(O.NeoConfig(**kw), seed=11, std=0.08), O.NeoConfig(**kw), seqs, normalize_embeddings=True)
    assert np.abs(got.cpu().numpy() - ref).max() < 5e-3


def test_rccl_text_search_sharded_branch_equals_plain_search(nccl_group):
    """DenseRetrievalExactSearch.search with the sharded branch forced on (a world of one runs the same collectives):
    balanced corpus range, query all-gather, local top-(k+1) with global indices, sgpt_exchange_topk -- same dict as the
    plain single-process search, on the real encoder through the text API."""
    from helpers import build_model
    from sgpt_amd.beir import CustomEmbedder, DenseRetrievalExactSearch
    from sgpt_amd.tokenization import SyntheticTokenizer
    kw = dict(vocab_size=211, max_position_embeddings=96, hidden_size=128, num_layers=4, num_heads=2, window_size=8)
    m = build_model(kw, 11, 0.08, "f16")
    emb = CustomEmbedder(model_name="synthetic/tiny", model=m, tokenizer=SyntheticTokenizer(211), method="weightedmean",
                         specb=True, maxseqlen=64)
    rng = np.random.default_rng(4)
    words = ["alpha", "beta", "gamma", "delta", "eps", "zeta", "eta", "theta"]
    corpus = {f"d{i}": {"title": "t", "text": " ".join(rng.choice(words, size=int(rng.integers(1, 50))))} for i in range(300)}
    queries = {f"q{i}": " ".join(rng.choice(words, size=int(rng.integers(1, 9)))) for i in range(13)}
    queries["d7"] = "alpha beta gamma"
    plain = DenseRetrievalExactSearch(emb, corpus_chunk_size=128).search(corpus, queries, 10, "cos_sim")
    dres = DenseRetrievalExactSearch(emb, corpus_chunk_size=128, distributed=True)
    got = dres.search(corpus, queries, 10, "cos_sim")
    assert dres.last_shard == (0, 1, 0, 300)
    assert set(got) == set(plain) and "d7" not in got["d7"]
    for qid in plain:
        assert got[qid] == plain[qid], qid


def _run_bench(args, env_extra, timeout=600):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_RANK"):
        env.pop(k, None)
    return subprocess.run([sys.executable] + args, cwd=root, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.timeout(900)
def test_bench_distributed_path_on_rccl_world_one():
    """bench.py's N > 1 code path (RCCL init, sharded query encode + all-gather, top-k exchange + merge, the
    sharded-vs-single-rank check, max-over-ranks timing) under torchrun with a world of one."""
    import json
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = _run_bench(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                    "--master-port", str(port), "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--chunk", "1024",
                    "--nq", "64", "--no-cpu-baseline", "--no-1m", "--no-varlen", "--no-modes"], {"SGPT_BENCH_FORCE_DIST": "1"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["shard_check"]["identical_to_single_rank"] is True


@pytest.mark.timeout(900)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two HIP devices (runs by default on the first multi-GPU box)")
def test_two_process_rccl_world():
    """N = 2 on hardware: `torchrun --nproc-per-node 2 tests/dist_worker.py` -- ragged all-gather, sharded search, exchange --
    and the bench line with --gpus 2."""
    import json
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = _run_bench(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                    "--master-port", str(port), "tests/dist_worker.py"], {"HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert r.returncode == 0 and r.stdout.count("DIST_WORKER_OK") == 2, r.stdout[-2000:] + r.stderr[-3000:]
    r = _run_bench(["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--chunk", "1024", "--nq", "64",
                    "--no-cpu-baseline", "--no-1m", "--no-varlen", "--no-modes"], {"HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["shard_check"]["identical_to_single_rank"] is True


def test_dist_worker_on_a_world_of_one():
    """The same worker under torchrun with one rank (this box): every step it takes on N ranks, degenerate collectives."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = _run_bench(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                    "--master-port", str(port), "tests/dist_worker.py"], {"HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert r.returncode == 0 and "DIST_WORKER_OK rank=0 world=1" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_bench_gpus_more_than_visible_fails_loudly():
    """`python bench.py --gpus N` outside torchrun spawns N ranks itself; with fewer devices it must refuse, not run 1."""
    import torch as _t
    n = _t.cuda.device_count() + 1
    r = _run_bench(["bench.py", "--gpus", str(n), "--steps", "1"], {}, timeout=120)
    assert r.returncode != 0 and "HIP device(s) visible" in (r.stdout + r.stderr)
