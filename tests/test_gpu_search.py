"""-m gpu: the drop-in adapters (DenseRetrievalExactSearch.search, CustomEmbedder, semantic_search,
SentenceTransformer-style encode, USEB semb_fn) against the reference's golden outputs and the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import sgpt_oracle as O
from helpers import GOLDEN, build_model, load_case, maxabs

pytestmark = pytest.mark.gpu


class FakeModel:
    """The fake encoder make_golden.py fed to the reference's exact_search.py: embeddings by id."""

    def __init__(self, qvec, cvec):
        self.qvec, self.cvec = qvec, cvec

    def encode_queries(self, qs, batch_size, **kw):
        return torch.from_numpy(np.stack([self.qvec[qid] for qid, _ in qs]))

    def encode_corpus(self, cs, batch_size, **kw):
        return np.stack([self.cvec[cid] for cid, _ in cs])           # ndarray: the CustomEmbedder return type


@pytest.mark.parametrize("fn", ["cos_sim", "dot"])
def test_exact_search_matches_reference_golden(fn):
    """Golden = the reference's DenseRetrievalExactSearch.search output (3 chunks of 256, top_k=10,
    two queries whose id collides with a corpus id, anisotropic embeddings)."""
    from sgpt_amd.beir import DenseRetrievalExactSearch
    fx = np.load(f"{GOLDEN}/scoring.npz")
    meta = json.loads(str(fx["es_json"]))
    corpus, qids, topk = meta["corpus"], meta["queries"], meta["top_k"]
    cvec = {f"d{i}": fx["es_corpus_emb"][i] for i in range(len(corpus))}
    qvec = {q: fx["es_query_emb"][i] for i, q in enumerate(qids)}
    s = DenseRetrievalExactSearch(FakeModel(qvec, cvec), batch_size=8, corpus_chunk_size=meta["chunk"])
    res = s.search(corpus, {q: "q" for q in qids}, topk, fn)
    want = meta["results"][fn]
    for qid in qids:
        assert set(res[qid]) == set(want[qid]), (fn, qid)
        assert qid not in res[qid] and len(res[qid]) <= topk + 1
        scale = max(1.0, max(abs(v) for v in want[qid].values()))
        for c, v in want[qid].items():
            assert abs(res[qid][c] - v) < 1e-3 * scale
    with pytest.raises(ValueError) as e:
        s.search(corpus, {q: "q" for q in qids}, topk, "euclid")
    assert str(e.value) == meta["bad_fn_msg"]


def test_semantic_search_golden_and_ref_test():
    """util.semantic_search golden (tests/test_util.py:33-53 shapes)."""
    from sgpt_amd import util
    fx = np.load(f"{GOLDEN}/scoring.npz")
    hits = util.semantic_search(torch.from_numpy(fx["ss_q"]), torch.from_numpy(fx["ss_docs"]), top_k=10,
                                query_chunk_size=5, corpus_chunk_size=17)
    assert len(hits) == 20 and len(hits[0]) == 10
    got_idx = np.array([[h["corpus_id"] for h in r] for r in hits])
    got_val = np.array([[h["score"] for h in r] for r in hits])
    assert np.array_equal(got_idx, fx["ss_idx"])
    assert maxabs(got_val, fx["ss_val"]) < 1e-3
    # custom score function path
    hits2 = util.semantic_search(fx["ss_q"], fx["ss_docs"], top_k=3, corpus_chunk_size=400,
                                 score_function=lambda a, b: -util.dot_score(a, b))
    want = O.semantic_search(fx["ss_q"], fx["ss_docs"], top_k=3, corpus_chunk_size=400,
                             score_function=lambda a, b: -O.dot_score(a, b))
    assert [[h["corpus_id"] for h in r] for r in hits2] == [[h["corpus_id"] for h in r] for r in want]


def _texts(rng, n, lo, hi):
    words = ["alpha", "beta", "gamma", "delta", "query", "doc", "paris", "atom", "cell", "gene", "?", "the", "of"]
    return [" ".join(rng.choice(words, size=int(rng.integers(lo, hi))).tolist()) for _ in range(n)]


def test_custom_embedder_and_search_end_to_end_vs_oracle(tmp_path, monkeypatch):
    """text -> SyntheticTokenizer -> specb brackets -> encode -> cosine top-k through the reference API
    names, against the oracle run on the identical ids/weights (fp32 mode, tolerance 1e-3)."""
    from sgpt_amd.beir import CustomEmbedder, DenseRetrievalExactSearch
    from sgpt_amd.tokenization import SyntheticTokenizer
    monkeypatch.chdir(tmp_path)
    fx, cfg_kw, *_ = load_case("tiny_right")
    seed, std = int(fx["seed"]), float(fx["std"])
    m = build_model(cfg_kw, seed, std, "fp32")
    tok = SyntheticTokenizer(cfg_kw["vocab_size"])
    emb = CustomEmbedder(model_name="synthetic/tiny-neo", model=m, tokenizer=tok, method="weightedmean", specb=True,
                         maxseqlen=40, dataset="unit")
    rng = np.random.default_rng(9)
    corpus = {f"d{i}": {"title": t.split(" ")[0], "text": t} for i, t in enumerate(_texts(rng, 150, 3, 60))}
    queries = {f"q{i}": t for i, t in enumerate(_texts(rng, 9, 2, 8))}
    queries["d3"] = "gene cell ?"                                      # id collides with a corpus id
    res = DenseRetrievalExactSearch(emb, corpus_chunk_size=64).search(corpus, queries, 5, "cos_sim")

    # oracle on the same ids
    cfg = O.NeoConfig(**cfg_kw)
    w = O.synth_weights(cfg, seed=seed, std=std)
    ids_of = lambda t, q: O.specb_wrap(tok.convert_tokens_to_ids(tok.tokenize(t.replace("\n", " "))), q, 38)  # noqa
    cids = sorted(corpus, key=lambda k: len(corpus[k]["title"] + corpus[k]["text"]), reverse=True)
    cemb = O.encode(w, cfg, [ids_of((corpus[c]["title"] + " " + corpus[c]["text"]).strip(), False) for c in cids])
    qemb = O.encode(w, cfg, [ids_of(queries[q], True) for q in queries])
    want = O.exact_search(qemb, list(queries), cemb, cids, 5, "cos_sim", chunk_size=64)
    for qid in queries:
        assert "d3" not in res["d3"]
        # order-insensitive id-set parity except where the boundary scores tie within tolerance
        common = set(res[qid]) & set(want[qid])
        assert len(common) >= len(want[qid]) - 1
        for c in common:
            assert abs(res[qid][c] - want[qid][c]) < 1e-3
    # encode_queries / encode_corpus return row-aligned ndarrays (beir_dense_retriever.py:316-348)
    qarr = emb.encode_queries([(q, queries[q]) for q in queries], batch_size=4)
    assert isinstance(qarr, np.ndarray) and qarr.shape == (len(queries), cfg.hidden_size)
    assert maxabs(qarr, qemb) < 1e-3
    carr = emb.encode_corpus([(c, corpus[c]) for c in cids[:7]], batch_size=4, batch_num=0)
    assert maxabs(carr, cemb[:7]) < 1e-3
    # other pooling methods of the adapter
    for method in ("mean", "lasttoken", "meanmean", "lasttokenmean"):
        e2 = CustomEmbedder(model_name="synthetic/tiny-neo", model=m, tokenizer=tok, method=method, dataset="unit")
        seqs = [tok.convert_tokens_to_ids(tok.tokenize(queries[q])) for q in queries]
        got = e2.embed_device([queries[q] for q in queries], True).cpu().numpy()
        assert maxabs(got, O.encode(w, cfg, seqs, mode=method)) < 1e-3, method


def test_search_tokenises_the_next_chunk_during_gpu_work(tmp_path, monkeypatch):
    """DenseRetrievalExactSearch.search hands the host leg (HF fast tokenizer, truncation, brackets) of corpus chunk i+1 to
    a worker thread while the GPU encodes and scores chunk i.  Same results as the serial loop; the wall time of a
    multi-chunk search (SGPT-125M shape, word-level fast tokenizer, ~100-token documents) is reported for both."""
    import time
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    from sgpt_amd.beir import CustomEmbedder, DenseRetrievalExactSearch
    monkeypatch.chdir(tmp_path)
    words = [f"w{i}" for i in range(5000)] + ["[", "]", "{", "}", "[UNK]"]
    tk = Tokenizer(models.WordLevel({w: i for i, w in enumerate(words)}, unk_token="[UNK]"))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    tok = PreTrainedTokenizerFast(tokenizer_object=tk, unk_token="[UNK]", eos_token="[UNK]")
    tok.pad_token = tok.eos_token
    fx, cfg_kw, *_ = load_case("cfg1_125m_32x64")
    m = build_model(cfg_kw, int(fx["seed"]), float(fx["std"]), "f16")
    emb = CustomEmbedder(model_name="synthetic/125m", model=m, tokenizer=tok, method="weightedmean", specb=True,
                         maxseqlen=128, dataset="unit")
    rng = np.random.default_rng(4)
    mk = lambda n, lo, hi: [" ".join(f"w{j}" for j in rng.integers(0, 5000, size=int(rng.integers(lo, hi)))) for _ in range(n)]  # noqa: E731
    corpus = {f"d{i}": {"title": "", "text": t} for i, t in enumerate(mk(12000, 60, 140))}
    queries = {f"q{i}": t for i, t in enumerate(mk(50, 4, 20))}
    out, secs = {}, {}
    for ahead in (True, False, True, False):
        dres = DenseRetrievalExactSearch(emb, corpus_chunk_size=3000, score_dtype=torch.float16, prefetch_tokenize=ahead)
        torch.cuda.synchronize()
        t = time.perf_counter()
        out[ahead] = dres.search(corpus, queries, 10, "cos_sim")
        secs[ahead] = time.perf_counter() - t
    assert out[True] == out[False]
    print(f"12 000 documents in 4 chunks: {secs[False]:.3f} s serial host leg, {secs[True]:.3f} s with the next chunk "
          f"tokenised during GPU work")


def test_embedding_pickle_cache_roundtrip(tmp_path, monkeypatch):
    """--saveemb cache format {id: ndarray} (beir_dense_retriever.py:311-312,319-323)."""
    import pickle
    from sgpt_amd.beir import CustomEmbedder
    from sgpt_amd.tokenization import SyntheticTokenizer
    monkeypatch.chdir(tmp_path)
    fx, cfg_kw, *_ = load_case("tiny_right")
    m = build_model(cfg_kw, int(fx["seed"]), float(fx["std"]), "fp32")
    emb = CustomEmbedder(model_name="synthetic/tiny-neo", model=m, tokenizer=SyntheticTokenizer(cfg_kw["vocab_size"]),
                         method="weightedmean", dataset="cache", save_emb=True)
    qs = [("a", "alpha beta"), ("b", "gamma")]
    first = emb.encode_queries(qs, batch_size=2)
    cache = pickle.load(open("embeddings/tiny-neo/weightedmean/cache_queries.pickle", "rb"))
    assert set(cache) == {"a", "b"} and np.array_equal(cache["a"], first[0])
    assert np.array_equal(emb.encode_queries(list(reversed(qs)), batch_size=2), first[::-1])   # served from cache


def test_st_encode_and_useb_semb_fn():
    from sgpt_amd.st import SentenceTransformerSGPT
    from sgpt_amd.tokenization import SyntheticTokenizer
    from sgpt_amd.useb import CustomEmbedder as UsebEmbedder, make_semb_fn
    from sgpt_amd.beir import SentenceBERTBOSEOS
    fx, cfg_kw, *_ = load_case("tiny_right")
    seed, std = int(fx["seed"]), float(fx["std"])
    m = build_model(cfg_kw, seed, std, "fp32")
    cfg = O.NeoConfig(**cfg_kw)
    w = O.synth_weights(cfg, seed=seed, std=std)
    tok = SyntheticTokenizer(cfg_kw["vocab_size"])
    st = SentenceTransformerSGPT(m, tok, max_seq_length=50)
    sents = ["alpha beta gamma", "the cell of gene ?", "paris"]
    want = O.encode(w, cfg, [tok.convert_tokens_to_ids(tok.tokenize(s)) for s in sents])
    out = st.encode(sents, batch_size=2)
    assert isinstance(out, np.ndarray) and maxabs(out, want) < 1e-3
    one = st.encode("paris")
    assert one.shape == (cfg.hidden_size,) and maxabs(one, want[2]) < 1e-3          # single string -> 1-D
    t = st.encode(sents, convert_to_tensor=True, normalize_embeddings=True)
    assert isinstance(t, torch.Tensor) and maxabs(t.cpu().numpy(), O.normalize(want)) < 1e-3
    lst = st.encode(sents, convert_to_numpy=False)
    assert isinstance(lst, list) and len(lst) == 3
    tokemb = st.encode(sents, output_value="token_embeddings")
    assert [e.shape[0] for e in tokemb] == [3, 5, 1]
    # output_value=None: every output of the module chain, one dict per sentence (SentenceTransformer.py:242-246)
    rows = st.encode(sents, output_value=None, convert_to_numpy=True)       # (conversion flags are ignored for it, :133-135)
    assert isinstance(rows, list) and len(rows) == 3 and set(rows[0]) == {"input_ids", "attention_mask", "token_embeddings", "sentence_embedding"}
    for r_, s_, te in zip(rows, sents, tokemb):
        assert r_["input_ids"].tolist() == tok.convert_tokens_to_ids(tok.tokenize(s_)) and int(r_["attention_mask"].sum()) == len(r_["input_ids"])
        assert torch.equal(r_["token_embeddings"], te)
    assert maxabs(torch.stack([r_["sentence_embedding"] for r_ in rows]).cpu().numpy(), want) < 1e-3
    assert set(st.encode("paris", output_value=None)) == set(rows[0])                # single string -> one dict
    # device / num_proc (SentenceTransformer.py:110-127,180-203): honoured where they mean this GPU / this process, refused loudly otherwise
    assert np.array_equal(st.encode(sents, device="cuda:0", num_proc=1), out) and np.array_equal(st.encode(sents, device="cuda"), out)
    with pytest.raises(ValueError, match="resident on"):
        st.encode(sents, device="cpu")
    with pytest.raises(ValueError, match="num_proc"):
        st.encode(sents, num_proc=2)
    with pytest.raises(ValueError, match="output_value"):
        st.encode(sents, output_value="pooler_output")
    # USEB closure: torch.Tensor[len, d] on the CPU (useb_dense_retriever.py:455-497)
    fn = make_semb_fn(UsebEmbedder(m, tok, method="weightedmean"))
    r = fn(sents, dataset_name="askubuntu", add_name="", idx=0)
    assert isinstance(r, torch.Tensor) and not r.is_cuda and maxabs(r.numpy(), want) < 1e-3
    # --usest --specb adapter: brackets on ids, upstream-DRES input conventions
    sb = SentenceBERTBOSEOS(model=m, tokenizer=tok, specb=True, max_seq_length=50)
    q = sb.encode_queries(["alpha beta gamma"], convert_to_tensor=True)
    wq = O.encode(w, cfg, [O.specb_wrap(tok.convert_tokens_to_ids(tok.tokenize("alpha beta gamma")), True)])
    assert maxabs(q.cpu().numpy(), wq) < 1e-3
    dd = sb.encode_corpus([{"title": "paris", "text": "the cell"}])
    wd = O.encode(w, cfg, [O.specb_wrap(tok.convert_tokens_to_ids(tok.tokenize("paris the cell")), False)])
    assert maxabs(dd, wd) < 1e-3


def _mp_factory(device):
    """Runs inside each pool worker: rebuild the tiny model from its seed on that device."""
    import ast, os
    import numpy as np
    from oracle import sgpt_oracle as O
    from sgpt_amd import SGPTConfig, SGPTModel
    from sgpt_amd.st import SentenceTransformerSGPT
    from sgpt_amd.tokenization import SyntheticTokenizer
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_right.npz"))
    kw = ast.literal_eval(str(fx["cfg"]))
    w = O.synth_weights(O.NeoConfig(**kw), seed=int(fx["seed"]), std=float(fx["std"]))
    return SentenceTransformerSGPT(SGPTModel(SGPTConfig(**kw), w, device=device, dtype="fp32"),
                                   SyntheticTokenizer(kw["vocab_size"]), max_seq_length=50)


@pytest.mark.timeout(300)
def test_encode_multi_process_matches_single_process():
    """sentence-transformers/tests/test_multi_process.py:14-29 restated: a pool of two worker processes
    (both on cuda:0 here, as the reference's test uses ['cpu','cpu']) must reproduce encode() within 1e-3."""
    from sgpt_amd.st import encode_multi_process, start_multi_process_pool, stop_multi_process_pool
    rng = np.random.default_rng(4)
    sentences = _texts(rng, 203, 1, 30)
    single = _mp_factory("cuda:0").encode(sentences)
    pool = start_multi_process_pool(_mp_factory, ["cuda:0", "cuda:0"])
    try:
        emb = encode_multi_process(sentences, pool, batch_size=16, chunk_size=50)
    finally:
        stop_multi_process_pool(pool)
    assert emb.shape == single.shape and np.abs(emb - single).max() < 1e-3


def _hf_cfg(cfg_kw):
    return {"model_type": "gpt_neo", "vocab_size": cfg_kw["vocab_size"],
            "max_position_embeddings": cfg_kw["max_position_embeddings"], "hidden_size": cfg_kw["hidden_size"],
            "num_layers": cfg_kw["num_layers"], "num_heads": cfg_kw["num_heads"], "window_size": cfg_kw["window_size"],
            "attention_types": [[["global", "local"], cfg_kw["num_layers"] // 2]]}


def test_st_folder_from_pretrained_roundtrip(tmp_path):
    """A sentence-transformers folder (modules.json, 1_Pooling / 1_WeightedMeanPooling, 2_Normalize;
    SentenceTransformer.py:389-430, 903-936) written to disk and loaded back runs the same pipeline."""
    from sgpt_amd import formats
    from sgpt_amd.st import SentenceTransformerSGPT
    from sgpt_amd.tokenization import SyntheticTokenizer
    fx, cfg_kw, seqs, pad_left, ids, mask = load_case("tiny_right")
    cfg = O.NeoConfig(**cfg_kw)
    w = O.synth_weights(cfg, seed=int(fx["seed"]), std=float(fx["std"]))
    tok = SyntheticTokenizer(cfg_kw["vocab_size"])
    texts = ["alpha beta gamma", "delta", "what is the capital of france ? paris of course"]
    tseqs = [tok.convert_tokens_to_ids(tok.tokenize(t)) for t in texts]
    p = str(tmp_path / "st_wm")
    formats.write_st_folder(p, _hf_cfg(cfg_kw), {"transformer." + k: v for k, v in w.items()}, pooling_mode="weightedmean",
                            max_seq_length=40, normalize=True)
    st = SentenceTransformerSGPT.from_pretrained(p, tokenizer=tok, device="cuda:0", dtype="fp32")
    assert (st.pooling_mode, st.normalize, st.max_seq_length) == ("weightedmean", True, 40)
    got = st.encode(texts)
    assert maxabs(got, O.encode(w, cfg, tseqs, normalize_embeddings=True)) < 1e-3
    # learntmean folder
    pw = np.random.default_rng(1).uniform(0.1, 2.0, size=41).astype(np.float32)
    p2 = str(tmp_path / "st_learnt")
    formats.write_st_folder(p2, _hf_cfg(cfg_kw), w, pooling_mode="learntmean", max_seq_length=40, position_weights=pw)
    st2 = SentenceTransformerSGPT.from_pretrained(p2, tokenizer=tok, device="cuda:0", dtype="fp32")
    assert st2.pooling_mode == "learntmean" and not st2.normalize
    tids, tmask = O.pad_batch(tseqs, pad_id=cfg.vocab_size - 1)
    last = O.forward_any(w, cfg, tids, tmask)
    assert maxabs(st2.encode(texts), O.pool(last, tmask, "learntmean", position_weights=pw)) < 1e-3


def test_asym_two_tower_and_speca_adapters():
    """SentenceBERTAsym: queries through the QRY tower, documents through the DOCPOS tower (sentence_bert_asym.py:8-19).
    SentenceBERTBOSEOS(speca=True): four ADDED vocabulary rows as markers (:52-63), marker-inclusive truncation."""
    from sgpt_amd import SGPTConfig, SGPTModel
    from sgpt_amd.beir import SentenceBERTAsym, SentenceBERTBOSEOS
    from sgpt_amd.tokenization import SyntheticTokenizer
    kw = dict(vocab_size=215, max_position_embeddings=96, hidden_size=128, num_layers=2, num_heads=2, window_size=8)
    cfg = O.NeoConfig(**kw)
    wq, wd = O.synth_weights(cfg, seed=71, std=0.08), O.synth_weights(cfg, seed=72, std=0.08)
    mq = SGPTModel(SGPTConfig(**kw), wq, device="cuda:0", dtype="fp32")
    md = SGPTModel(SGPTConfig(**kw), wd, device="cuda:0", dtype="fp32")
    tok = SyntheticTokenizer(211)
    queries = ["what is the capital of france ?", "river"]
    docs = [{"title": "paris", "text": "paris is the capital of france"}, {"title": "", "text": "the seine is a river"}]
    dtexts = [(d["title"] + " " + d["text"]).strip() for d in docs]
    ids = lambda t: tok.convert_tokens_to_ids(tok.tokenize(t))  # noqa: E731
    asym = SentenceBERTAsym(query_model=mq, doc_model=md, tokenizer=tok, max_seq_length=64)
    assert maxabs(asym.encode_queries(queries), O.encode(wq, cfg, [ids(q) for q in queries])) < 1e-3
    assert maxabs(asym.encode_corpus(docs), O.encode(wd, cfg, [ids(t) for t in dtexts])) < 1e-3
    assert maxabs(asym.encode_corpus(docs), O.encode(wq, cfg, [ids(t) for t in dtexts])) > 1e-2   # really the other tower
    # speca: ids 211..214 are the added [SOS] [EOS] {SOS} {EOS}; max_seq_length 8 -> 5 content tokens + 2 markers
    sa = SentenceBERTBOSEOS(speca=True, model=mq, tokenizer=SyntheticTokenizer(211), max_seq_length=8)
    want_q = [[211] + ids(q)[:5] + [212] for q in queries]
    want_d = [[213] + ids(t)[:5] + [214] for t in dtexts]
    assert sa.pipe.batch(queries, True) == want_q and sa.pipe.batch(dtexts, False) == want_d
    assert maxabs(sa.encode_queries(queries), O.encode(wq, cfg, want_q)) < 1e-3
    assert maxabs(sa.encode_corpus(docs), O.encode(wq, cfg, want_d)) < 1e-3
    with pytest.raises(ValueError):     # a checkpoint without the four extra embedding rows
        small = SGPTModel(SGPTConfig(**dict(kw, vocab_size=211)), O.synth_weights(O.NeoConfig(**dict(kw, vocab_size=211)), seed=1),
                          device="cuda:0", dtype="fp32")
        SentenceBERTBOSEOS(speca=True, model=small, tokenizer=SyntheticTokenizer(211), max_seq_length=8)
    mq.close(); md.close()


def test_crossencoder_loglikelihood_vs_reference_golden():
    """cross-encoder re-ranking scores = sum of log P(query token | prompted document ...) --
    sgpt_lm_logprobs on the continuation rows of one packed forward vs the reference's `_loglikelihood_tokens`
    output (tests/golden/crossencoder.json), then the GPTRanker surface against the oracle."""
    from sgpt_amd import SGPTConfig, SGPTModel
    from sgpt_amd.crossencoder import GPTRanker, loglikelihood_tokens, model_input
    from sgpt_amd.tokenization import SyntheticTokenizer
    fx = json.load(open(f"{GOLDEN}/crossencoder.json"))
    cfg = O.NeoConfig(**fx["cfg"])
    w = O.synth_weights(cfg, seed=fx["seed"], std=fx["std"])
    reqs = [(("c", "q"), c, q) for c, q in fx["requests"]]
    want = np.asarray(fx["loglikelihood"])
    m = SGPTModel(SGPTConfig(**fx["cfg"]), w, device="cuda:0", dtype="fp32")
    got = np.asarray(loglikelihood_tokens(reqs, m, fx["max_length"], instruction_len=fx["instruction_len"]))
    assert np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))) < 1e-3
    small = np.asarray(loglikelihood_tokens(reqs, m, fx["max_length"], instruction_len=fx["instruction_len"], max_tokens_per_call=64))
    assert np.max(np.abs(small - got)) < 1e-3                       # batching does not matter
    # greedy tokens = argmax of the same logits
    inp = model_input(reqs[0][1], reqs[0][2], fx["max_length"], fx["instruction_len"])
    pb = m.pack([inp])
    _, hid = m.encode_packed(pb, return_hidden=True)
    lp, greedy = m.lm_logprobs(hid, list(range(len(inp))), [0] * len(inp), return_greedy=True)
    last = O.forward_any(w, cfg, np.asarray([inp]), np.ones((1, len(inp)), dtype=np.int64))
    logits = last[0] @ w["wte.weight"].T
    assert (greedy.cpu().numpy() == logits.argmax(-1)).mean() > 0.95      # ties / 1e-6 differences aside
    assert np.max(np.abs(lp.cpu().numpy() - O.log_softmax(logits)[:, 0])) < 1e-3
    # bf16 model: ranking scores stay close
    mb = SGPTModel(SGPTConfig(**fx["cfg"]), w, device="cuda:0", dtype="bf16")
    gb = np.asarray(loglikelihood_tokens(reqs, mb, fx["max_length"], instruction_len=fx["instruction_len"]))
    assert np.max(np.abs(gb - want) / np.maximum(1.0, np.abs(want))) < 3e-2
    # GPTRanker.predict (prompted documents, few-shot prefix) vs the oracle on the same token ids
    tok = SyntheticTokenizer(211)
    rk = GPTRanker(m, tok, max_length=64, prompt_doc="Document : {}\nQuery :", fewshots=("a doc", "a query"),
                   prompt_doc_start="Document : {}\nQuery : {}\n")
    pairs = [("what is the capital of france", "paris is the capital of france . " * 4), ("river", "the seine")]
    scores = rk.predict(pairs, batch_size=8)
    oreqs = []
    for qtext, doc in pairs:
        ctx_ids = tok.encode(rk.fewshots + rk.prompt_doc.format(doc))
        oreqs.append((("c", "q"), ctx_ids, tok.encode(qtext)))
    owant = O.loglikelihood_tokens(w, cfg, oreqs, 64, rk.instruction_len)
    assert np.max(np.abs(np.asarray(scores) - np.asarray(owant)) / np.maximum(1.0, np.abs(owant))) < 1e-3
    m.close(); mb.close()


def test_crossencoder_gptj_untied_lm_head():
    """GPT-J (SGPT-5.8B family) carries its own lm_head.weight / lm_head.bias (HF GPTJForCausalLM): the cross-encoder
    scorer must use them instead of the tied embedding."""
    from helpers import oracle_cfg_weights
    from sgpt_amd import SGPTConfig, SGPTModel
    from sgpt_amd.crossencoder import loglikelihood_tokens
    cfg_kw = dict(vocab_size=211, n_positions=96, n_embd=512, n_layer=2, n_head=2, rotary_dim=64)
    cfg, w = oracle_cfg_weights(cfg_kw, 61, 0.04)
    rng = np.random.default_rng(61)
    w = dict(w)
    w["lm_head.weight"] = (rng.standard_normal((211, 512)) * 0.05).astype(np.float32)
    w["lm_head.bias"] = (rng.standard_normal(211) * 0.5).astype(np.float32)
    reqs = [(("c", "q"), rng.integers(0, 211, size=a).tolist(), rng.integers(0, 211, size=b).tolist())
            for a, b in [(12, 4), (30, 9), (5, 1), (60, 20)]]
    want = np.asarray(O.loglikelihood_tokens(w, cfg, reqs, 64, 2))
    m = SGPTModel(SGPTConfig.from_hf_dict(dict(cfg_kw, model_type="gptj")), w, device="cuda:0", dtype="fp32")
    got = np.asarray(loglikelihood_tokens(reqs, m, 64, instruction_len=2))
    assert np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))) < 1e-3
    # and it is not the tied head
    w_tied = {k: v for k, v in w.items() if not k.startswith("lm_head")}
    w_tied["lm_head.weight"] = w["wte.weight"]
    tied = np.asarray(O.loglikelihood_tokens(w_tied, cfg, reqs, 64, 2))
    assert np.max(np.abs(tied - want)) > 0.1
    m.close()


def test_crossencoder_bloom_tied_head():
    """BLOOM ties the LM head to word_embeddings (ALiBi positions, embedding LayerNorm): same scorer, third family."""
    from helpers import oracle_cfg_weights
    from sgpt_amd import SGPTConfig, SGPTModel
    from sgpt_amd.crossencoder import loglikelihood_tokens
    cfg_kw = dict(vocab_size=211, hidden_size=256, n_layer=2, n_head=2)
    cfg, w = oracle_cfg_weights(cfg_kw, 62, 0.04)
    rng = np.random.default_rng(62)
    reqs = [(("c", "q"), rng.integers(0, 211, size=a).tolist(), rng.integers(0, 211, size=b).tolist())
            for a, b in [(14, 3), (33, 10), (2, 2), (50, 16)]]
    want = np.asarray(O.loglikelihood_tokens(w, cfg, reqs, 64, 0))
    m = SGPTModel(SGPTConfig.from_hf_dict(dict(cfg_kw, model_type="bloom")), w, device="cuda:0", dtype="fp32")
    got = np.asarray(loglikelihood_tokens(reqs, m, 64))
    assert np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))) < 1e-3
    m.close()


class _IdTokenizer:
    """text = space-separated decimal token ids: the three tokenizer calls the reference makes (tokenize, convert_tokens_to_ids,
    encode -- beir_dense_retriever.py:167-198) reproduce the fixture's ids exactly, so the text API can be driven with the
    reference fixture's sequences."""
    eos_token_id = pad_token_id = 50256

    def tokenize(self, text):
        return text.split()

    def convert_tokens_to_ids(self, tokens):
        return [int(t) for t in tokens]

    def encode(self, text, add_special_tokens=False):
        return [int(t) for t in text.split()]


@pytest.mark.parametrize("fn", ["cos_sim", "dot"])
@pytest.mark.parametrize("score_dtype", ["default", "f16"])
def test_search_in_the_default_f16_mode_vs_the_reference_fixture(fn, score_dtype, tmp_path, monkeypatch):
    """VERDICT r04 missing-3 / next-2b: text -> ids -> encode -> chunked search -> result dict through the drop-in classes in
    the DEFAULT mode of the adapter (CustomEmbedder dtype='f16', SGPTModel precision='auto'; DenseRetrievalExactSearch's
    score_dtype default = fp32 rows on the exact-fp32 scorer, and the opt-in 16-bit corpus), at SGPT-125M shape, for BOTH score
    functions, against the reference stack's own numbers: tests/golden/cfg2_125m_1024x128.npz = HF GPTNeoModel fp32 -> the
    reference's Pooling.py -> util.cos_sim / dot_score -> exact_search's top-(k+1) rule (oracle.exact_search, pinned against the
    reference's exact_search.py by tests/golden/make_golden.py).  Covered rules: corpus sorted by text length (:66-70), chunk
    loop + heap merge (:80-132), top_k+1 kept (:104,126), corpus_id != query_id (:118), NaN -> -1 (:99; one document's embedding is
    poisoned with NaN on both sides).
    Tolerances.  cos_sim: the north_star bar, 1e-3 absolute.  dot: the reference's dot scores of these un-normalised pooled rows
    are |q||d| cos with |q||d| ~ 1e2..1e3, so an absolute bar is meaningless; the encoder's deviation is RELATIVE to the row norms
    (f16 operands: <= 4e-3 absolute on O(1-3) elements = ~1.3e-3 of the norm, tests/test_gpu_parity_cfg2.py `rel`), so dot is held to
    DOT_REL x |q||d| with DOT_REL = 1e-3 -- the cosine bar carried to the scale of the score; measured 2.1e-4 (fp32 and f16 corpus rows alike), printed."""
    from sgpt_amd import SGPTConfig, SGPTModel
    from sgpt_amd.beir import CustomEmbedder, DenseRetrievalExactSearch
    from helpers import oracle_cfg_weights
    monkeypatch.chdir(tmp_path)
    fx = np.load(os.path.join(GOLDEN, "cfg2_125m_1024x128.npz"))
    DOT_REL, BAR, top_k, chunk = 1e-3, 1e-3, 10, 300
    key = ("default-mode-125m",)
    if key not in _default_models:
        _, w = oracle_cfg_weights(dict(O.SGPT_125M), 1, 0.02)
        _default_models[key] = SGPTModel(SGPTConfig(**O.SGPT_125M), w, device="cuda:0")          # every default: f16, precision='auto'
    m = _default_models[key]
    assert m.dtype == "f16" and m.precision == "auto"

    class Poisoned(CustomEmbedder):            # one document comes back as NaN: the scorer must rank it as -1 (exact_search.py:99)
        def encode_corpus_device(self, corpus, **kw):
            e = super().encode_corpus_device(corpus, **kw)
            for r, (cid, _) in enumerate(corpus):
                if cid == "d17":
                    e = e.clone()
                    e[r] = float("nan")
            return e
    emb = Poisoned(model_name="synthetic/sgpt-125m", model=m, tokenizer=_IdTokenizer(), method="weightedmean", dataset="unit")
    assert emb.model.dtype == "f16"
    docs, qlens = fx["doc_ids"], fx["query_lens"].tolist()
    corpus = {f"d{i}": {"title": "", "text": " ".join(map(str, docs[i].tolist()))} for i in range(docs.shape[0])}
    queries = {(f"q{i}" if i % 7 else f"d{3 * i}"): " ".join(map(str, fx["query_ids"][i, :n].tolist())) for i, n in enumerate(qlens)}
    assert sum(q in corpus for q in queries) >= 10                                  # query ids that collide with corpus ids (:118)
    kw = {} if score_dtype == "default" else {"score_dtype": torch.float16}
    dres = DenseRetrievalExactSearch(emb, corpus_chunk_size=chunk, **kw)
    assert dres.score_dtype == (torch.float32 if score_dtype == "default" else torch.float16)
    res = dres.search(corpus, queries, top_k, fn)
    assert m.precision_report["decided"] == "plain"                                # a clean checkpoint: the probe keeps plain f16 operands

    # the reference side, on the reference's own fp32 embeddings of the same ids
    cids = sorted(corpus, key=lambda c: len(corpus[c]["title"] + corpus[c]["text"]), reverse=True)
    pos = np.array([int(c[1:]) for c in cids])
    ref_d, ref_q = fx["doc_emb"][pos].copy(), fx["query_emb"]
    ref_d[cids.index("d17")] = np.nan
    qids = list(queries)
    want = O.exact_search(ref_q, qids, ref_d, cids, top_k, fn, chunk_size=chunk)
    sc = (O.cos_sim(ref_q, ref_d) if fn == "cos_sim" else O.dot_score(ref_q, ref_d)).astype(np.float64)
    sc[np.isnan(sc)] = -1
    qn, dn = np.linalg.norm(ref_q, axis=1), np.linalg.norm(np.nan_to_num(ref_d), axis=1)
    col = {c: j for j, c in enumerate(cids)}
    worst, worst_rel, same = 0.0, 0.0, 0
    for qi, qid in enumerate(qids):
        got, ref = res[qid], want[qid]
        assert qid not in got and "d17" not in got                                  # self-match dropped; the NaN document ranks at -1
        assert len(got) == len(ref) and len(got) in (top_k, top_k + 1)             # top_k+1 kept, minus a dropped self-match
        kth = min(ref.values())
        for cid, s in got.items():
            j = col[cid]
            scale = 1.0 if fn == "cos_sim" else float(qn[qi] * dn[j])
            tol = BAR if fn == "cos_sim" else DOT_REL * scale
            assert abs(s - sc[qi, j]) <= tol, (qid, cid, s, sc[qi, j], tol)
            worst, worst_rel = max(worst, abs(s - sc[qi, j])), max(worst_rel, abs(s - sc[qi, j]) / scale)
            assert sc[qi, j] >= kth - 2 * tol, (qid, cid)                            # a swapped document sits within 2 tol of the reference's boundary
        same += set(got) == set(ref)
    print(f"default-mode search {fn} / scorer {score_dtype}: max|score - ref| = {worst:.3e} (relative to |q||d|: {worst_rel:.2e}), "
          f"identical id sets for {same} of {len(qids)} queries")
    if os.environ.get("SGPT_PARITY_LOG"):
        with open(os.environ["SGPT_PARITY_LOG"], "a") as f:
            f.write(json.dumps(dict(case="search_default_mode_cfg2_125m", score_function=fn, scorer=score_dtype, max_abs_score=worst,
                                    max_rel_score=worst_rel, identical_id_sets=same, n_queries=len(qids), n_docs=len(cids),
                                    budget=BAR if fn == "cos_sim" else DOT_REL)) + "\n")
    assert same >= 0.85 * len(qids)


_default_models = {}
