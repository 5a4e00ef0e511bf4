"""The oracle against (1) the committed golden vectors produced by the real reference
(tests/golden/make_golden.py: HF GPTNeoModel + reference Pooling.py / util.py /
exact_search.py) and (2) the reference's own offline tests, restated:
sentence-transformers/tests/test_util.py:9-18, 21-30, 33-53, 69-76."""
import ast
import json
import os

import numpy as np
import pytest

from oracle import sgpt_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(golden_dir, tag):
    fx = np.load(os.path.join(golden_dir, f"{tag}.npz"))
    kw = ast.literal_eval(str(fx["cfg"]))
    cfg = O.GPTJConfig(**kw) if "n_embd" in kw else (O.BloomConfig(**kw) if "n_layer" in kw else O.NeoConfig(**kw))
    lens = fx["seq_lens"].tolist()
    ids, mask = fx["ids"].astype(np.int64), fx["mask"].astype(np.int64)
    side = str(fx["pad_side"])
    seqs = [ids[i, :n].tolist() if side == "right" else ids[i, ids.shape[1] - n:].tolist() for i, n in enumerate(lens)]
    return fx, cfg, seqs, ids, mask, side


@pytest.mark.parametrize("tag", ["tiny_right", "tiny_left", "tiny_dh128", "tiny_gptj_right", "tiny_gptj_left",
                                 "tiny_bloom_left", "tiny_bloom_right"])
def test_oracle_encoder_matches_hf_golden(golden_dir, tag):
    fx, cfg, seqs, ids, mask, side = load_case(golden_dir, tag)
    if isinstance(cfg, O.BloomConfig):
        w = O.synth_weights_bloom(cfg, seed=int(fx["seed"]), std=float(fx["std"]))
    elif isinstance(cfg, O.GPTJConfig):
        w = O.synth_weights_gptj(cfg, seed=int(fx["seed"]), std=float(fx["std"]))
    else:
        w = O.synth_weights(cfg, seed=int(fx["seed"]), std=float(fx["std"]))
    last, hs = O.forward_any(w, cfg, ids, mask, output_hidden_states=True)
    real = mask.astype(bool)
    assert np.abs(last[real] - fx["last_hidden"][real]).max() < 2e-4
    assert np.abs(hs[1][real] - fx["hidden_1"][real]).max() < 2e-4
    for mode in ("weightedmean", "mean", "lasttoken"):
        assert np.abs(O.pool(last, mask, mode) - fx[f"emb_{mode}"]).max() < 2e-4
        # pooling alone, fed the reference's own hidden states
        assert np.abs(O.pool(fx["last_hidden"], mask, mode) - fx[f"emb_{mode}"]).max() < 5e-6
    assert np.abs(O.pool(hs[-2], mask, "weightedmean") - fx["emb_weightedmean_layer_m2"]).max() < 2e-4
    enc = O.encode(w, cfg, seqs, batch_size=len(seqs), pad_side=side)
    assert np.abs(enc - fx["emb_weightedmean"]).max() < 2e-4


def test_oracle_cfg3_specb_window(golden_dir):
    """125M shape, specb brackets, S=300 (GPT-Neo local window 256 active)."""
    fx, cfg, seqs, ids, mask, side = load_case(golden_dir, "cfg3_125m_specb_s300")
    assert seqs[0][0] == O.SPECB_DOC_BOS and seqs[0][-1] == O.SPECB_DOC_EOS and len(seqs[0]) == 300
    assert seqs[-1][0] == O.SPECB_QUE_BOS and seqs[-1][-1] == O.SPECB_QUE_EOS
    w = O.synth_weights(cfg, seed=int(fx["seed"]), std=float(fx["std"]))
    enc = O.encode(w, cfg, seqs[:2] + seqs[-1:], batch_size=3)
    want = np.concatenate([fx["emb_weightedmean"][:2], fx["emb_weightedmean"][-1:]])
    # right padding => embeddings do not depend on the batch's max length
    assert np.abs(enc - want).max() < 2e-4


def test_right_padding_is_batch_invariant_left_is_not(golden_dir):
    fx, cfg, seqs, ids, mask, side = load_case(golden_dir, "tiny_right")
    w = O.synth_weights(cfg, seed=int(fx["seed"]), std=float(fx["std"]))
    one = O.encode(w, cfg, [seqs[3]], batch_size=1)
    assert np.abs(one[0] - fx["emb_weightedmean"][3]).max() < 2e-4
    both_l = O.encode(w, cfg, seqs, batch_size=len(seqs), pad_side="left")
    one_l = O.encode(w, cfg, [seqs[3]], batch_size=1, pad_side="left")
    assert np.abs(one_l[0] - both_l[3]).max() > 1e-3


def test_scoring_golden(golden_dir):
    fx = np.load(os.path.join(golden_dir, "scoring.npz"))
    assert np.abs(O.cos_sim(fx["a"], fx["b"]) - fx["cos"]).max() < 1e-6
    assert np.abs(O.dot_score(fx["a"], fx["b"]) - fx["dot"]).max() < 1e-5
    assert np.abs(O.normalize(fx["a"]) - fx["nrm"]).max() < 1e-6
    assert np.abs(O.pairwise_cos_sim(fx["a"][:37], fx["b"]) - fx["pcs"]).max() < 1e-6
    # 1-D inputs promoted (util.py:35-39)
    assert O.cos_sim(fx["a"][0], fx["b"]).shape == (1, 37)
    hits = O.semantic_search(fx["ss_q"], fx["ss_docs"], top_k=10, query_chunk_size=5, corpus_chunk_size=17)
    assert (np.array([[h["corpus_id"] for h in r] for r in hits]) == fx["ss_idx"]).all()
    assert np.abs(np.array([[h["score"] for h in r] for r in hits]) - fx["ss_val"]).max() < 1e-6


@pytest.mark.parametrize("fn", ["cos_sim", "dot"])
def test_exact_search_golden(golden_dir, fn):
    fx = np.load(os.path.join(golden_dir, "scoring.npz"))
    meta = json.loads(str(fx["es_json"]))
    corpus, qids, topk = meta["corpus"], meta["queries"], meta["top_k"]
    cvec = {f"d{i}": fx["es_corpus_emb"][i] for i in range(len(corpus))}
    cids = sorted(corpus, key=lambda k: len(corpus[k].get("title", "") + corpus[k].get("text", "")), reverse=True)
    res = O.exact_search(fx["es_query_emb"], qids, np.stack([cvec[c] for c in cids]), cids, topk, fn,
                         chunk_size=meta["chunk"])
    want = meta["results"][fn]
    for qid in qids:
        assert set(res[qid]) == set(want[qid])
        assert len(res[qid]) <= topk + 1 and qid not in res[qid]
        for c, v in want[qid].items():
            assert abs(res[qid][c] - v) < 1e-4
    with pytest.raises(ValueError) as e:
        O.exact_search(fx["es_query_emb"], qids, fx["es_corpus_emb"], cids, topk, "euclid")
    assert str(e.value) == meta["bad_fn_msg"]


# ---- the reference's own offline tests, restated (sentence-transformers/tests/test_util.py) ----
def test_ref_normalize_embeddings():                       # test_util.py:9-18
    a = np.random.default_rng(0).standard_normal((50, 100))
    for e in O.normalize(a):
        assert len(e) == 100 and abs(np.linalg.norm(e) - 1) < 1e-4


def test_ref_cos_sim_vs_sklearn():                         # test_util.py:21-30
    from sklearn.metrics.pairwise import cosine_similarity
    rng = np.random.default_rng(1)
    a, b = rng.standard_normal((50, 100)), rng.standard_normal((50, 100))
    assert np.abs(cosine_similarity(a, b) - O.cos_sim(a, b)).max() < 1e-3


def test_ref_semantic_search():                            # test_util.py:33-53
    rng = np.random.default_rng(2)
    doc, q = rng.standard_normal((1000, 100)), rng.standard_normal((20, 100))
    hits = O.semantic_search(q, doc, top_k=10, query_chunk_size=5, corpus_chunk_size=17)
    assert len(hits) == 20 and len(hits[0]) == 10
    vals, idx = O.topk_rows(O.cos_sim(q, doc), 10)
    for qi in range(20):
        for h in range(10):
            assert hits[qi][h]["corpus_id"] == idx[qi][h]
            assert abs(hits[qi][h]["score"] - vals[qi][h]) < 1e-3


def test_ref_pairwise_scores():                            # test_util.py:69-76
    from sklearn.metrics.pairwise import paired_cosine_distances
    rng = np.random.default_rng(3)
    a, b = rng.standard_normal((50, 100)), rng.standard_normal((50, 100))
    assert np.allclose(1 - paired_cosine_distances(a, b), O.pairwise_cos_sim(a, b), atol=1e-6)


def test_bf16_round_matches_torch():
    import torch
    a = np.random.default_rng(4).standard_normal(10000).astype(np.float32) * 3
    want = torch.from_numpy(a).to(torch.bfloat16).to(torch.float32).numpy()
    assert (O.bf16_round(a) == want).all()


def test_extras_golden_learntmean_and_fp8():
    """learntmean pooling vs the reference's WeightedMeanPooling.py output; fp8 e4m3fn codes / power-of-two
    scales vs the fixture that make_golden.py checked bit for bit against torch.float8_e4m3fn."""
    fx = np.load(os.path.join(GOLDEN, "extras.npz"))
    got = O.pool(fx["lm_hidden"], fx["lm_mask"], "learntmean", position_weights=fx["lm_pw"])
    assert np.max(np.abs(got - fx["lm_ref"])) < 2e-6
    codes, scale = O.fp8_quantize_rows(fx["fp8_w"])
    assert np.array_equal(codes, fx["fp8_codes"]) and np.array_equal(scale, fx["fp8_scale"])
    assert np.array_equal(O.fp8_dequantize_rows(codes, scale), fx["fp8_deq"])
    # properties: scales are powers of two, |w/scale| <= 448, relative error of normal codes <= 2^-4
    assert np.all(np.log2(scale) == np.round(np.log2(scale)))
    x = fx["fp8_w"] / scale[:, None]
    assert np.abs(x).max() <= 448.0
    deq = fx["fp8_deq"] / scale[:, None]
    normal = np.abs(x) >= 2.0 ** -6
    assert np.all(np.abs(deq - x)[normal] <= np.abs(x)[normal] * 2.0 ** -4)
    assert np.all(np.abs(deq - x)[~normal] <= 2.0 ** -10)
    # torch cross-check when float8 is available in this torch build
    import torch
    if hasattr(torch, "float8_e4m3fn"):
        t8 = torch.from_numpy(x.astype(np.float32)).to(torch.float8_e4m3fn)
        assert np.array_equal(t8.view(torch.uint8).numpy(), codes)


def test_crossencoder_loglikelihood_golden():
    """Cross-encoder scores: golden = the reference's own `_loglikelihood_tokens` (crossencoder/beir/sgptce.py) on HF
    GPTNeoForCausalLM; includes requests longer than max_length (left truncation behind the instruction)."""
    fx = json.load(open(os.path.join(GOLDEN, "crossencoder.json")))
    cfg = O.NeoConfig(**fx["cfg"])
    w = O.synth_weights(cfg, seed=fx["seed"], std=fx["std"])
    reqs = [(("c", "q"), c, q) for c, q in fx["requests"]]
    got = O.loglikelihood_tokens(w, cfg, reqs, fx["max_length"], fx["instruction_len"])
    assert np.max(np.abs(np.asarray(got) - np.asarray(fx["loglikelihood"]))) < 2e-4
    # truncation rule: instruction kept, rest cut from the left, last token dropped
    inp = O.ce_model_input(list(range(100, 170)), list(range(12)), 48, 3)
    assert inp[:3] == [100, 101, 102] and len(inp) == 48 and inp[-1] == 10 and inp[3] == 100 + 70 - (46 - 12)


def test_exact_search_torch_backend_equals_numpy_lines():
    """oracle.exact_search(backend="torch") -- bench.py's CPU search baseline -- returns the hit sets and scores of the
    numpy lines (which the golden fixtures pin against the reference's exact_search.py)."""
    rng = np.random.default_rng(3)
    q = rng.standard_normal((9, 32)).astype(np.float32)
    c = rng.standard_normal((257, 32)).astype(np.float32)
    c[5] = 0.0                                             # zero row: cosine 0 through the eps clamp, not NaN
    qids, cids = [f"q{i}" for i in range(9)], [f"d{i}" for i in range(257)]
    qids[2] = "d7"                                         # the corpus_id != query_id rule
    for fn in ("cos_sim", "dot"):
        a = O.exact_search(q, qids, c, cids, 5, fn, chunk_size=100)
        b = O.exact_search(q, qids, c, cids, 5, fn, chunk_size=100, backend="torch")
        for qid in qids:
            assert set(a[qid]) == set(b[qid]) and "d7" not in b["d7"]
            assert max(abs(a[qid][k] - b[qid][k]) for k in a[qid]) < 1e-5
