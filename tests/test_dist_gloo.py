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
    """numpy stand-in for sgpt_amd.runtime.Context with the calls the sharded search makes (no HIP device here).  Not a
    `Context`, so sgpt_amd.dist.get_comm gives it the gloo transport (`TorchComm`); a real Context always gets RCCL."""
    device = torch.device("cpu")

    def l2_normalize(self, x, out_dtype=torch.float32):
        return torch.from_numpy(O.normalize(np.asarray(x, dtype=np.float32)))

    def score_topk(self, q, corpus, k, idx_base=0, run=None, dtype=None):
        sc = O.dot_score(q.numpy(), corpus.numpy())        # rows are normalised by the caller for cos_sim (util.py:41-43)
        sc[np.isnan(sc)] = -1
        v, i = O.topk_rows(sc, k)
        return torch.from_numpy(v), torch.from_numpy(i + idx_base), k

    def topk_merge(self, val, idx, k, exclude_idx=None):
        v, i = val.numpy().copy(), idx.numpy()
        if exclude_idx is not None:
            v[i == exclude_idx.numpy()[:, None]] = -np.inf
        # descending score, ties -> ascending index (the contract of sgpt_topk_merge)
        v[i < 0] = -np.inf                                    # idx < 0: not a candidate
        order = np.lexsort((i, -v), axis=1)[:, :k]
        ov, oi = np.take_along_axis(v, order, 1), np.take_along_axis(i, order, 1).copy()
        oi[np.isneginf(ov)] = -1                              # ignored / excluded candidates leave as (-inf, -1)
        return torch.from_numpy(ov), torch.from_numpy(oi)


class FakeEncoder:
    """Deterministic stand-in for SGPTModel: the embedding of a sentence depends only on its ids."""
    device = "cpu"
    ctx = StubCtx()

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
        from sgpt_amd.dist import all_gather_queries, exchange_topk, shard_range, shard_sizes, sharded_score_topk
        rng = np.random.default_rng(0)                      # same data on both ranks
        q = rng.standard_normal((nq, d)).astype(np.float32)
        c = rng.standard_normal((N, d)).astype(np.float32)
        c[7] = c[3]                                         # a duplicated document: tie -> the lower index wins
        qlo, qhi = shard_range(nq, rank, world)
        clo, chi = shard_range(N, rank, world)
        ctx = StubCtx()
        q_all = all_gather_queries(ctx, torch.from_numpy(q[qlo:qhi]), shard_sizes(nq, world))
        ok = torch.equal(q_all, torch.from_numpy(q))
        qn, cn = O.normalize(q), O.normalize(c)             # sharded_score_topk takes normalised rows (cos_sim = dot of those)
        # the product's sharded search, end to end, with the self-match rule on two queries
        excl = np.full(nq, -1, dtype=np.int64)
        excl[2], excl[5] = 11, N - 1
        fv, fi = sharded_score_topk(ctx, torch.from_numpy(qn[qlo:qhi]), nq, torch.from_numpy(cn[clo:chi]), k,
                                    idx_base=clo, exclude_idx=torch.from_numpy(excl))
        sc = O.cos_sim(q, c)
        sc[np.arange(nq), excl] = np.where(excl >= 0, -np.inf, sc[np.arange(nq), excl])
        wv, wi = O.topk_rows(sc, k)
        ok = ok and np.array_equal(fi.numpy(), wi) and np.allclose(fv.numpy(), wv, atol=1e-6)
        # the exchange alone (all-gather of the per-rank lists + merge): the global top-k without the exclusion
        v, i = O.topk_rows(O.cos_sim(q, c[clo:chi]), k)
        cv, ci = exchange_topk(ctx, torch.from_numpy(v), torch.from_numpy(i + clo), k)
        gv, gi = O.topk_rows(O.cos_sim(q, c), k)
        ok = ok and cv.shape == (nq, k) and np.array_equal(ci.numpy(), gi) and np.allclose(cv.numpy(), gv, atol=1e-6)
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


def test_balanced_cuts_bound_the_token_load():
    """the contiguous cut of the length-sorted list balances TOKENS.  U{16..128} lengths, 8 ranks:
    equal sentence counts (the reference's rule, SentenceTransformer.py:159-163) give rank 0 ~1.7x the mean load; the
    balanced cut keeps max / mean <= 1.05 (here: within one sentence of the ideal)."""
    from sgpt_amd.dist import balanced_cuts, shard_sizes
    rng = np.random.default_rng(0)
    for n, world in [(100_000, 8), (4096, 8), (1000, 2), (4096, 3)]:
        lens = np.sort(rng.integers(16, 129, size=n))[::-1]                 # longest first
        alloc = (lens + 7) // 8 * 8
        cuts = balanced_cuts(alloc, world)
        assert cuts[0] == 0 and cuts[-1] == n and (np.diff(cuts) >= 0).all()
        load = np.array([alloc[cuts[r]: cuts[r + 1]].sum() for r in range(world)], dtype=np.float64)
        assert load.max() / load.mean() <= 1.05, (n, world, load)
        assert load.max() - load.mean() <= alloc.max()                       # within one sentence of the ideal cut
        lim = np.cumsum([0] + shard_sizes(n, world))
        eq = np.array([alloc[lim[r]: lim[r + 1]].sum() for r in range(world)], dtype=np.float64)
        if world == 8:
            assert eq.max() / eq.mean() > 1.5                                 # what the equal-count rule would have done
    # degenerate inputs: fewer items than ranks, empty list, one giant item
    assert balanced_cuts([5], 4).tolist()[-1] == 1 and (np.diff(balanced_cuts([5], 4)) >= 0).all()
    assert balanced_cuts([], 3).tolist() == [0, 0, 0, 0]
    c = balanced_cuts([1000, 1, 1, 1], 2)
    assert c.tolist() == [0, 1, 4]


class TextModel:
    """Generic BEIR-style model for the text API (encode_queries / encode_corpus on (id, text) tuples): embeddings depend
    only on the text, so a sharded and a single-process search must agree exactly."""

    @staticmethod
    def _emb(text):
        a = np.frombuffer(text.encode(), dtype=np.uint8).astype(np.float64)
        return np.array([np.sin(0.11 * j * a.sum() + 0.7 * j) + 0.05 * len(a) * (j % 3) + np.cos(a[: 1 + j % max(1, len(a))].sum())
                         for j in range(16)], dtype=np.float32)

    def encode_queries(self, queries, batch_size=None, **kw):
        return torch.from_numpy(np.stack([self._emb(t) for _, t in queries]))

    def encode_corpus(self, corpus, batch_size=None, **kw):
        return torch.from_numpy(np.stack([self._emb((d.get("title", "") + " " + d["text"]).strip()) for _, d in corpus]))


def _text_data(n_docs, n_queries):
    rng = np.random.default_rng(5)
    words = ["alpha", "beta", "gamma", "delta", "eps", "zeta", "eta", "theta", "iota", "kappa"]
    corpus = {f"d{i}": {"title": " ".join(rng.choice(words, size=int(rng.integers(0, 3)))),
                        "text": " ".join(rng.choice(words, size=int(rng.integers(1, 40))))} for i in range(n_docs)}
    queries = {f"q{i}": " ".join(rng.choice(words, size=int(rng.integers(1, 8)))) for i in range(n_queries)}
    # query ids that collide with corpus ids: the corpus_id != query_id rule (exact_search.py:118) across shards
    queries["d3"] = "alpha beta"
    queries[f"d{n_docs - 1}"] = "kappa"
    return corpus, queries


def _text_search_worker(rank, world, port, out, n_docs, n_queries, chunk, top_k, fn, long_query=False):
    _init(rank, world, port)
    try:
        from sgpt_amd.beir import DenseRetrievalExactSearch
        corpus, queries = _text_data(n_docs, n_queries)
        if long_query:
            queries = dict([("qlong", "alpha " * 400)] + list(queries.items()))
        dres = DenseRetrievalExactSearch(TextModel(), corpus_chunk_size=chunk, ctx=StubCtx())
        got = dres.search(corpus, queries, top_k, fn)                         # distributed branch (world 2)
        out.put((rank, got, dres.last_shard))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
@pytest.mark.parametrize("n_docs,n_queries,chunk,fn", [(137, 21, 50, "cos_sim"), (137, 21, 50, "dot"), (1, 9, 50, "cos_sim"),
                                                       (60, 1, 7, "cos_sim")])      # 3 queries < 4 per rank: every rank encodes them all
def test_two_rank_text_search_equals_single_process(n_docs, n_queries, chunk, fn):
    """DenseRetrievalExactSearch.search under torch.distributed: token-balanced contiguous corpus
    ranges, sharded query encode + one all-gather, local top-(k+1) with global indices, exchange + merge -- the dict every
    rank returns equals the single-process dict (ids and scores), including the corpus_id == query_id rule and a rank with
    an empty range (n_docs = 1)."""
    from sgpt_amd.beir import DenseRetrievalExactSearch
    top_k = 5
    corpus, queries = _text_data(n_docs, n_queries)
    single = DenseRetrievalExactSearch(TextModel(), corpus_chunk_size=chunk, ctx=StubCtx()).search(corpus, queries, top_k, fn)
    res = _run(_text_search_worker, 2, n_docs, n_queries, chunk, top_k, fn)
    ranges = []
    for rank, got, shard in res:
        assert shard[0] == rank and shard[1] == 2
        ranges.append(shard[2:])
        assert set(got) == set(single)
        for qid in single:
            assert set(got[qid]) == set(single[qid]), (rank, qid)
            assert qid not in got[qid]                                          # the self-match never comes back
            for cid, sc in single[qid].items():
                assert abs(got[qid][cid] - sc) <= 1e-6, (rank, qid, cid)
    assert ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][1] == n_docs       # a partition of the corpus
    if n_docs == 1:
        assert ranges[0][1] - ranges[0][0] == 0 or ranges[1][1] - ranges[1][0] == 0          # one rank had nothing to score


@pytest.mark.timeout(240)
def test_three_rank_text_search_with_one_dominant_query():
    """one query longer than total / world pulled several length-balanced cuts onto the same position;
    the rank with the empty query slice crashed in `all_gather_rows(None)` and the others hung in the collective.  The query
    cuts now leave no rank empty (balanced_cuts(min_one=True)); three ranks, the first query 400 words long."""
    from sgpt_amd.beir import DenseRetrievalExactSearch
    from sgpt_amd.dist import balanced_cuts
    n_docs, n_queries, chunk, top_k, fn = 97, 12, 40, 4, "cos_sim"
    corpus, queries = _text_data(n_docs, n_queries)
    queries = dict([("qlong", "alpha " * 400)] + list(queries.items()))
    lens = [len(q) + 1 for q in queries.values()]
    assert (np.diff(balanced_cuts(lens, 3)) == 0).any()                 # the plain cuts DO leave a rank empty here
    assert (np.diff(balanced_cuts(lens, 3, min_one=True)) > 0).all()
    single = DenseRetrievalExactSearch(TextModel(), corpus_chunk_size=chunk, ctx=StubCtx()).search(corpus, queries, top_k, fn)
    res = _run(_text_search_worker, 3, n_docs, n_queries, chunk, top_k, fn, True)
    assert len(res) == 3
    for rank, got, shard in res:
        assert set(got) == set(single)
        for qid in single:
            assert set(got[qid]) == set(single[qid]), (rank, qid)
            for cid, sc in single[qid].items():
                assert abs(got[qid][cid] - sc) <= 1e-6


def test_balanced_cuts_properties():
    """Property test: the cuts are a monotone partition, a pure function of (weights, world), and no rank is heavier than the
    mean by more than the heaviest item."""
    from hypothesis import given, settings, strategies as st
    from sgpt_amd.dist import balanced_cuts

    @settings(max_examples=200, deadline=None)
    @given(st.lists(st.integers(min_value=1, max_value=2048), min_size=0, max_size=300), st.integers(min_value=1, max_value=9))
    def check(weights, world):
        c = balanced_cuts(weights, world)
        assert c.shape == (world + 1,) and c[0] == 0 and c[-1] == len(weights) and (np.diff(c) >= 0).all()
        assert np.array_equal(c, balanced_cuts(list(weights), world))
        if weights:
            w = np.asarray(weights)
            load = np.array([w[c[r]: c[r + 1]].sum() for r in range(world)], dtype=np.float64)
            assert load.sum() == w.sum() and load.max() <= w.sum() / world + w.max()
            m = balanced_cuts(weights, world, min_one=True)
            assert m[0] == 0 and m[-1] == len(weights) and (np.diff(m) >= 0).all()
            if len(weights) >= world:
                assert (np.diff(m) >= 1).all()                      # no rank is left without an item
    check()


class _PlanModel:
    """The host half of SGPTModel's precision='auto' decision with the device calls stubbed (no HIP device here): the REAL
    sync_precision / _install_from_flags / plan builders of sgpt_amd.model.SGPTModel run on it."""

    def __init__(self, hot_rank):
        from sgpt_amd.model import SGPTConfig
        self.cfg = SGPTConfig(num_layers=3, hidden_size=128, num_heads=2, vocab_size=64, max_position_embeddings=64)
        self.precision, self.dtype, self.precise_qk = "auto", "f16", False
        self._plan_pending, self._att_ok, self.precision_report = True, True, None
        self.hot_rank, self.installed, self.released, self.probed = hot_rank, None, 0, 0

    def probe_precision(self, seqs, pad_left=None):
        self.probed += 1
        crest = np.full((3, 4), 6.0, dtype=np.float32)
        if dist.get_rank() == self.hot_rank:
            crest[1, 3] = 55.0                                    # one GELU-output class of block 1, on ONE rank's data only
        return crest

    def set_precision_plan(self, plan, _keep_pending=False):
        self.installed = np.asarray(plan).copy()
        if not _keep_pending:
            self._plan_pending = False

    def release_split_weights(self):
        self.released += 1
        return 123


def _bind_plan_methods():
    from sgpt_amd.model import SGPTModel
    for name in ("sync_precision", "_install_from_flags", "_crest_limits", "_base_plan", "_x3_plan"):
        setattr(_PlanModel, name, getattr(SGPTModel, name))


class _PlanTextModel(TextModel):
    def __init__(self, hot_rank):
        _bind_plan_methods()
        self.model = _PlanModel(hot_rank)

    def tokenize(self, sentences, is_query):
        return [[1 + (len(t) % 60)] * 4 for t in sentences]


def _precision_sync_worker(rank, world, port, out, hot_rank):
    _init(rank, world, port)
    try:
        from sgpt_amd.beir import DenseRetrievalExactSearch
        corpus, queries = _text_data(40, 12)
        tm = _PlanTextModel(hot_rank)
        DenseRetrievalExactSearch(tm, corpus_chunk_size=16, ctx=StubCtx()).search(corpus, queries, 3, "cos_sim")
        m = tm.model
        out.put((rank, m.precision_report["decided"], None if m.installed is None else m.installed.tolist(), m.released, m.probed,
                 m._plan_pending))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
@pytest.mark.parametrize("hot_rank", [1, -1])
def test_two_rank_precision_decision_is_collective(hot_rank):
    """precision='auto' settles from the first sequences a PROCESS encodes, and the sharded search hands
    every rank a different query slice and corpus shard.  The decision is taken once for the group before anything is
    encoded: outlier activations that only rank 1's data show move BOTH ranks to f16x3 (same plan); clean data on both
    ranks leaves both plain and gives the split weight copies back."""
    res = _run(_precision_sync_worker, 2, hot_rank)
    assert len(res) == 2
    decided = {r[1] for r in res}
    assert decided == ({"x3"} if hot_rank >= 0 else {"plain"})
    plans = [r[2] for r in res]
    assert plans[0] == plans[1]
    for rank, dec, plan, released, probed, pending in res:
        assert probed == 1 and not pending
        assert released == (0 if hot_rank >= 0 else 1)
        if hot_rank >= 0:
            assert np.asarray(plan).all()                       # every class of every block split


def _precision_released_worker(rank, world, port, out, hot_rank):
    _init(rank, world, port)
    try:
        from sgpt_amd.beir import DenseRetrievalExactSearch
        corpus, queries = _text_data(40, 12)
        tm = _PlanTextModel(hot_rank)
        m = tm.model
        if rank == 0:
            # rank 0 encoded on its own earlier: its probe settled on plain operands and the split weight copies went back
            m._install_from_flags(np.zeros((3, 4), dtype=bool), None, probed="first call")
            assert m.precision_report["split_weight_bytes_released"] == 123 and not m._plan_pending
        err = None
        try:
            DenseRetrievalExactSearch(tm, corpus_chunk_size=16, ctx=StubCtx()).search(corpus, queries, 3, "cos_sim")
        except RuntimeError as e:
            err = str(e)
        out.put((rank, err, m.precision_report["decided"] if m.precision_report else None))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
@pytest.mark.parametrize("hot_rank", [1, -1])
def test_two_rank_precision_escalation_after_a_release_fails_on_every_rank(hot_rank):
    """rank 0 has already encoded, settled on plain operands and released its split weight copies; rank 1's
    probe then flags a class.  Rank 0 cannot install the union plan any more -- the refusal must reach BOTH ranks (one collective
    decision), not leave rank 1 waiting in the query all-gather.  Clean data on rank 1: the search simply runs."""
    res = _run(_precision_released_worker, 2, hot_rank)
    assert len(res) == 2
    if hot_rank >= 0:
        assert all(r[1] is not None and "released" in r[1] and "every rank" in r[1] for r in res), res
    else:
        assert all(r[1] is None and r[2] == "plain" for r in res), res


def test_st_encode_device_and_num_proc_are_honoured_or_refused():
    """SentenceTransformer.encode(device=..., num_proc=...) (SentenceTransformer.py:110-127,180-203): the model's own device and a
    single process are accepted, anything else is refused loudly instead of silently ignored (VERDICT r04 missing-4)."""
    from sgpt_amd.st import SentenceTransformerSGPT
    from sgpt_amd.tokenization import SyntheticTokenizer
    st = SentenceTransformerSGPT(FakeEncoder(), SyntheticTokenizer(64), max_seq_length=32)
    sents = ["alpha beta", "gamma", "the cell of gene"]
    base = st.encode(sents)
    assert np.array_equal(st.encode(sents, device="cpu", num_proc=1), base)
    with pytest.raises(ValueError, match="resident on"):
        st.encode(sents, device="cuda:1")
    with pytest.raises(ValueError, match="num_proc"):
        st.encode(sents, num_proc=4)
    with pytest.raises(ValueError, match="output_value"):
        st.encode(sents, output_value="nonsense")
