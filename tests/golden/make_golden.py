#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REAL reference.

Runs only in the build container (needs /root/reference and HF transformers):
  * HF ``GPTNeoModel`` (eager attention, fp32, eval) -- the un-vendored dependency
    the reference calls at beir_dense_retriever.py:204-205 / Transformer.py:72,
  * the reference's own ``Pooling.py`` (loaded from its file),
  * the pure-torch head of the vendored ``util.py`` (cos_sim, dot_score,
    normalize_embeddings, semantic_search, pairwise_cos_sim),
  * the reference's ``exact_search.py`` with a stub ``beir`` module.

Weights are NOT stored: they are regenerated from a seed by
``oracle.sgpt_oracle.synth_weights`` (numpy Generator streams are stable across
platforms), so fixtures stay small.  While generating, every oracle function is
checked against the reference output (this is what pins the oracle).

    python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import sgpt_oracle as O  # noqa: E402

REF = "/root/reference"
ST = f"{REF}/biencoder/nli_msmarco/sentence-transformers/sentence_transformers"


def load_file_module(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_ref_util():
    """exec lines 1-449 of the vendored util.py (everything below needs
    huggingface_hub symbols that no longer exist)."""
    src = open(f"{ST}/util.py").read().split("\n")
    head = "\n".join(src[:449])
    head = head.replace("import requests\n", "").replace("from tqdm.autonotebook import tqdm", "tqdm = None")
    mod = types.ModuleType("ref_util")
    exec(compile(head, f"{ST}/util.py", "exec"), mod.__dict__)
    return mod


def load_ref_exact_search(ref_util):
    beir = types.ModuleType("beir")
    beir_util = types.ModuleType("beir.util")
    import logging

    class LoggingHandler(logging.Handler):
        def emit(self, record):
            pass

    beir.LoggingHandler = LoggingHandler
    beir_util.cos_sim = ref_util.cos_sim
    beir_util.dot_score = ref_util.dot_score
    beir.util = beir_util
    sys.modules["beir"] = beir
    sys.modules["beir.util"] = beir_util
    return load_file_module("ref_exact_search", f"{REF}/biencoder/beir/custommodels/exact_search.py")


def hf_model(cfg: O.NeoConfig, w):
    from transformers import GPTNeoConfig, GPTNeoModel
    types_ = []
    # HF wants [[pattern, repeat]]: alternate global/local
    assert cfg.attention_layers == ["global" if i % 2 == 0 else "local" for i in range(cfg.num_layers)]
    hc = GPTNeoConfig(vocab_size=cfg.vocab_size, max_position_embeddings=cfg.max_position_embeddings,
                      hidden_size=cfg.hidden_size, num_layers=cfg.num_layers, num_heads=cfg.num_heads,
                      intermediate_size=cfg.intermediate_size, window_size=cfg.window_size,
                      attention_types=[[["global", "local"], cfg.num_layers // 2]],
                      layer_norm_epsilon=cfg.layer_norm_epsilon,
                      attention_dropout=0.0, resid_dropout=0.0, embed_dropout=0.0)
    hc._attn_implementation = "eager"
    m = GPTNeoModel(hc).eval()
    sd = {k: torch.from_numpy(v.copy()) for k, v in w.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("bias" in k and "attn.attention.bias" in k or "masked_bias" in k for k in missing), missing
    return m


def hf_model_gptj(cfg, w):
    from transformers import GPTJConfig, GPTJModel
    hc = GPTJConfig(vocab_size=cfg.vocab_size, n_positions=cfg.max_position_embeddings, n_embd=cfg.hidden_size,
                    n_layer=cfg.num_layers, n_head=cfg.num_heads, rotary_dim=cfg.rotary_dim,
                    n_inner=cfg.intermediate_size, layer_norm_epsilon=cfg.layer_norm_epsilon,
                    resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0, activation_function="gelu_new")
    hc._attn_implementation = "eager"
    m = GPTJModel(hc).eval()
    sd = {k: torch.from_numpy(v.copy()) for k, v in w.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("attn.bias" in k or "masked_bias" in k or "embed_positions" in k for k in missing), missing
    return m


def hf_model_bloom(cfg, w):
    from transformers import BloomConfig, BloomModel
    hc = BloomConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, n_layer=cfg.num_layers, n_head=cfg.num_heads,
                     layer_norm_epsilon=cfg.layer_norm_epsilon, hidden_dropout=0.0, attention_dropout=0.0,
                     apply_residual_connection_post_layernorm=False, pretraining_tp=1, slow_but_exact=False)
    hc._attn_implementation = "eager"
    m = BloomModel(hc).eval()
    sd = {k: torch.from_numpy(v.copy()) for k, v in w.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    return m


def rand_seqs(rng, n, lo, hi, vocab):
    return [rng.integers(0, vocab, size=int(rng.integers(lo, hi + 1))).tolist() for _ in range(n)]


def check(name, got, want, tol):
    err = float(np.max(np.abs(np.asarray(got, dtype=np.float64) - np.asarray(want, dtype=np.float64))))
    print(f"  oracle-vs-reference {name}: max|diff| = {err:.3e} (tol {tol:g})")
    assert err <= tol, (name, err)
    return err


def encoder_case(tag, cfg_kw, seed, seqs, pad_side="right", std=0.02, store_hidden=False, Pooling=None, arch="gpt_neo"):
    if arch == "bloom":
        cfg = O.BloomConfig(**cfg_kw)
        w = O.synth_weights_bloom(cfg, seed=seed, std=std)
        model = hf_model_bloom(cfg, w)
    elif arch == "gptj":
        cfg = O.GPTJConfig(**cfg_kw)
        w = O.synth_weights_gptj(cfg, seed=seed, std=std)
        model = hf_model_gptj(cfg, w)
    else:
        cfg = O.NeoConfig(**cfg_kw)
        w = O.synth_weights(cfg, seed=seed, std=std)
        model = hf_model(cfg, w)
    ids, mask = O.pad_batch(seqs, pad_id=min(O.GPT2_PAD, cfg.vocab_size - 1), side=pad_side)
    with torch.no_grad():
        out = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask),
                    output_hidden_states=True)
    last = out.last_hidden_state.numpy()
    hs = [h.numpy() for h in out.hidden_states]
    # reference pooling (Pooling.py) on HF hidden states
    d = cfg.hidden_size
    emb = {}
    for mode, kw in (("weightedmean", dict(pooling_mode_weightedmean_tokens=True, pooling_mode_mean_tokens=False)),
                     ("mean", dict(pooling_mode_mean_tokens=True))):
        pm = Pooling.Pooling(d, **kw)
        feats = {"token_embeddings": torch.from_numpy(last.copy()), "attention_mask": torch.from_numpy(mask)}
        emb[mode] = pm.forward(feats)["sentence_embedding"].numpy()
    # raw-HF lasttoken (beir_dense_retriever.py:271-282): gather at len-1 of the real tokens
    gi = np.array([np.nonzero(r)[0][-1] for r in mask])
    emb["lasttoken"] = last[np.arange(len(seqs)), gi]
    # raw-HF weightedmean cross-check (beir_dense_retriever.py:258-270)
    ime = torch.from_numpy(mask).unsqueeze(-1).expand(last.shape).float()
    wts = torch.arange(1, last.shape[1] + 1).unsqueeze(0).unsqueeze(-1).expand(last.shape).float()
    raw = (torch.sum(torch.from_numpy(last) * ime * wts, dim=1) / torch.sum(ime * wts, dim=1)).numpy()
    check(f"{tag} Pooling.py vs raw path weightedmean", emb["weightedmean"], raw, 1e-6)
    # layeridx variants (hidden_states[i], beir_dense_retriever.py:233) -- store pooled layer -2 (pre-ln_f input of last block)
    pm = Pooling.Pooling(d, pooling_mode_weightedmean_tokens=True, pooling_mode_mean_tokens=False)
    emb_l2 = pm.forward({"token_embeddings": torch.from_numpy(hs[-2].copy()),
                         "attention_mask": torch.from_numpy(mask)})["sentence_embedding"].numpy()

    # ---- pin the oracle ----
    o_last, o_hs = O.forward_any(w, cfg, ids, mask, output_hidden_states=True)
    real = mask.astype(bool)
    check(f"{tag} last_hidden (real tokens)", o_last[real], last[real], 2e-4)
    for li in (0, 1, len(hs) - 2):
        check(f"{tag} hidden_states[{li}] (real tokens)", o_hs[li][real], hs[li][real], 2e-4)
    for mode in ("weightedmean", "mean", "lasttoken"):
        check(f"{tag} pool {mode}", O.pool(o_last, mask, mode), emb[mode], 2e-4)
        check(f"{tag} pool-only {mode}", O.pool(last, mask, mode), emb[mode], 5e-6)
    enc = O.encode(w, cfg, seqs, mode="weightedmean", batch_size=len(seqs), pad_side=pad_side)
    check(f"{tag} encode()", enc, emb["weightedmean"], 2e-4)

    fx = dict(cfg=np.array(repr(cfg_kw)), arch=np.array(arch), seed=seed, std=std, pad_side=np.array(pad_side),
              seq_lens=np.array([len(s) for s in seqs]), ids=ids.astype(np.int32), mask=mask.astype(np.int8),
              emb_weightedmean=emb["weightedmean"], emb_mean=emb["mean"], emb_lasttoken=emb["lasttoken"],
              emb_weightedmean_layer_m2=emb_l2)
    if store_hidden:
        fx["last_hidden"] = last
        fx["hidden_1"] = hs[1]
    np.savez_compressed(os.path.join(HERE, f"{tag}.npz"), **fx)
    print(f"wrote {tag}.npz")


def main():
    only_j = bool(os.environ.get("GOLDEN_ONLY_GPTJ"))
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    Pooling = load_file_module("ref_pooling", f"{ST}/models/Pooling.py")
    U = load_ref_util()
    ES = load_ref_exact_search(U)

    # ---------------- encoder + pooling ----------------
    if os.environ.get("GOLDEN_ONLY_CFG2"):
        _cfg2_case(Pooling, U, ES)
        return
    if os.environ.get("GOLDEN_ONLY_EXTRAS"):
        _extras_cases()
        _tokenize_cases()
        _crossencoder_case()
        return
    if only_j:
        _gptj_cases(Pooling)
        return
    tiny = dict(vocab_size=211, max_position_embeddings=96, hidden_size=128, num_layers=4,
                num_heads=2, window_size=8)
    rng = np.random.default_rng(100)
    encoder_case("tiny_right", tiny, seed=11, seqs=rand_seqs(rng, 9, 1, 40, 211), std=0.08,
                 store_hidden=True, Pooling=Pooling)
    encoder_case("tiny_left", tiny, seed=11, seqs=rand_seqs(rng, 7, 2, 33, 211), pad_side="left",
                 std=0.08, store_hidden=True, Pooling=Pooling)
    tiny128 = dict(vocab_size=211, max_position_embeddings=96, hidden_size=256, num_layers=2,
                   num_heads=2, window_size=16)          # head_dim 128 (the 1.3B/2.7B head size)
    encoder_case("tiny_dh128", tiny128, seed=12, seqs=rand_seqs(rng, 6, 3, 50, 211), std=0.06,
                 store_hidden=True, Pooling=Pooling)
    # BASELINE config 1: SGPT-125M shape, 32 sentences, seq_len<=64
    rng = np.random.default_rng(0)
    seqs = rand_seqs(rng, 32, 8, 64, 50256)
    seqs[0] = rng.integers(0, 50256, size=64).tolist()
    encoder_case("cfg1_125m_32x64", O.SGPT_125M, seed=0, seqs=seqs, Pooling=Pooling)
    # config-3 style: specb brackets, docs up to 300 tokens -> exercises the 256 local window
    rng = np.random.default_rng(2)
    docs = [O.specb_wrap(rng.integers(0, 50256, size=n).tolist(), is_query=False) for n in (298, 280, 150, 17)]
    qs = [O.specb_wrap(rng.integers(0, 50256, size=n).tolist(), is_query=True) for n in (30, 7)]
    encoder_case("cfg3_125m_specb_s300", O.SGPT_125M, seed=2, seqs=docs + qs, Pooling=Pooling)

    _gptj_cases(Pooling)
    _extras_cases()
    _tokenize_cases()
    _crossencoder_case()

    # ---------------- scoring / top-k (reference util.py + exact_search.py) ----------------
    _scoring_cases(U, ES)
    _cfg2_case(Pooling, U, ES)


def _cfg2_case(Pooling, U, ES):
    """BASELINE configs[1] at a size where the throughput kernels (256x256 GEMM tiles) are the ones that run:
    SGPT-125M shape, 1024 documents x 128 tokens + 100 queries of 4..32 tokens, through the reference stack end to end:
    HF GPTNeoModel fp32 eager -> the reference's Pooling.py (weightedmean) -> the reference's util.cos_sim ->
    the reference's DenseRetrievalExactSearch.search (top-10, cos_sim).  The weights are seed 1 of synth_weights =
    sgpt_amd.synthetic_weights(seed=1), the weights bench.py runs.  Stored: embeddings, cosine scores, ranked ids."""
    import json
    cfg = O.NeoConfig(**O.SGPT_125M)
    w = O.synth_weights(cfg, seed=1)
    model = hf_model(cfg, w)
    torch.set_num_threads(os.cpu_count() or 1)
    rng = np.random.default_rng(21)
    nd, nq, S, topk = 1024, 100, 128, 10
    docs = rng.integers(0, 50256, size=(nd, S), dtype=np.int64)
    qs = [rng.integers(0, 50256, size=int(rng.integers(4, 33))).tolist() for _ in range(nq)]
    pm = Pooling.Pooling(cfg.hidden_size, pooling_mode_weightedmean_tokens=True, pooling_mode_mean_tokens=False)

    def ref_encode(ids, mask):
        out = []
        for s0 in range(0, len(ids), 32):
            i, m = torch.from_numpy(ids[s0:s0 + 32]), torch.from_numpy(mask[s0:s0 + 32])
            h = model(input_ids=i, attention_mask=m).last_hidden_state
            out.append(pm.forward({"token_embeddings": h, "attention_mask": m})["sentence_embedding"].numpy())
        return np.concatenate(out)
    d_emb = ref_encode(docs, np.ones_like(docs))
    q_ids, q_mask = O.pad_batch(qs, pad_id=O.GPT2_PAD, side="right")
    q_emb = ref_encode(q_ids, q_mask)
    cos = U.cos_sim(torch.from_numpy(q_emb), torch.from_numpy(d_emb)).numpy()
    # pin the oracle on a slice (the whole set costs minutes of numpy)
    sel = np.arange(0, nd, 64)
    check("cfg2 oracle encode (16 docs)", O.encode(w, cfg, docs[sel].tolist(), batch_size=16), d_emb[sel], 2e-4)
    check("cfg2 oracle encode (8 queries)", O.encode(w, cfg, qs[:8], batch_size=8), q_emb[:8], 2e-4)
    check("cfg2 oracle cos_sim", O.cos_sim(q_emb, d_emb), cos, 1e-6)

    corpus = {f"d{i}": {"title": "", "text": "x" * (1 + i % 7)} for i in range(nd)}
    queries = {f"q{i}": "q" for i in range(nq)}
    cvec = {f"d{i}": d_emb[i] for i in range(nd)}
    qvec = {f"q{i}": q_emb[i] for i in range(nq)}

    class FakeModel:
        def encode_queries(self, qq, batch_size, **kw):
            return torch.from_numpy(np.stack([qvec[qid] for qid, _ in qq]))

        def encode_corpus(self, cs, batch_size, **kw):
            return torch.from_numpy(np.stack([cvec[cid] for cid, _ in cs]))
    res = ES.DenseRetrievalExactSearch(FakeModel(), batch_size=8, corpus_chunk_size=400).search(corpus, queries, topk, "cos_sim")
    # ranked ids (the reference returns an unordered dict of top_k+1 hits): sort by (-score, doc index)
    ranked = np.array([[int(c[1:]) for c in sorted(res[f"q{i}"], key=lambda c: (-res[f"q{i}"][c], int(c[1:])))[:topk]]
                       for i in range(nq)], dtype=np.int64)
    brute = np.argsort(-cos, axis=1, kind="stable")[:, :topk]
    assert np.array_equal(ranked, brute), "reference exact_search top-10 != brute force on the reference cosine matrix"
    np.savez_compressed(os.path.join(HERE, "cfg2_125m_1024x128.npz"), seed=1, doc_ids=docs.astype(np.int32),
                        query_lens=np.array([len(q) for q in qs]), query_ids=q_ids.astype(np.int32),
                        doc_emb=d_emb, query_emb=q_emb, cos=cos.astype(np.float32), top10=ranked,
                        meta=np.array(json.dumps(dict(nd=nd, nq=nq, S=S, topk=topk, rng_seed=21))))
    srt = -np.sort(-cos, axis=1)
    print(f"wrote cfg2_125m_1024x128.npz; cos range [{cos.min():.3f}, {cos.max():.3f}], "
          f"smallest gap between rank 10 and rank 11: {float((srt[:, 9] - srt[:, 10]).min()):.2e}")


def _gptj_cases(Pooling):
    # GPT-J family (SGPT-5.8B, BASELINE config 4): head_dim 256, rotary_dim 64, parallel block
    tinyj = dict(vocab_size=211, n_positions=96, n_embd=512, n_layer=2, n_head=2, rotary_dim=64)
    rng = np.random.default_rng(300)
    encoder_case("tiny_gptj_right", tinyj, seed=31, seqs=rand_seqs(rng, 7, 1, 70, 211), std=0.04,
                 store_hidden=True, Pooling=Pooling, arch="gptj")
    encoder_case("tiny_gptj_left", tinyj, seed=31, seqs=rand_seqs(rng, 5, 2, 40, 211), pad_side="left", std=0.04,
                 store_hidden=True, Pooling=Pooling, arch="gptj")
    # BLOOM family (sgpt-bloom-7b1, BASELINE config 5): ALiBi, embedding LayerNorm, fused interleaved QKV + biases.
    # 6 heads exercises the non-power-of-two slope branch; BLOOM tokenizers pad LEFT (pooling weights depend on it).
    tinyb = dict(vocab_size=211, hidden_size=384, n_layer=2, n_head=6)
    rng = np.random.default_rng(400)
    encoder_case("tiny_bloom_left", tinyb, seed=41, seqs=rand_seqs(rng, 7, 1, 70, 211), pad_side="left", std=0.04,
                 store_hidden=True, Pooling=Pooling, arch="bloom")
    tinyb2 = dict(vocab_size=211, hidden_size=256, n_layer=2, n_head=2)    # head_dim 128 (the 7b1 head size)
    encoder_case("tiny_bloom_right", tinyb2, seed=42, seqs=rand_seqs(rng, 5, 2, 40, 211), std=0.04,
                 store_hidden=True, Pooling=Pooling, arch="bloom")


def _extras_cases():
    """learntmean pooling vs the reference's WeightedMeanPooling.py + the raw USEB formula;
    fp8 e4m3fn weight codes vs torch.float8_e4m3fn."""
    WMP = load_file_module("ref_wmp", f"{ST}/models/WeightedMeanPooling.py")
    rng = np.random.default_rng(900)
    B, S, d = 6, 37, 64
    h = rng.standard_normal((B, S, d)).astype(np.float32)
    lens = [37, 1, 20, 5, 36, 11]
    mask = np.zeros((B, S), dtype=np.int64)
    for b, n in enumerate(lens):                          # rows 0-2 right padded, rows 3-5 left padded
        if b < 3:
            mask[b, :n] = 1
        else:
            mask[b, S - n:] = 1
    pw = rng.uniform(0.05, 3.0, size=48).astype(np.float32)
    mod = WMP.WeightedMeanPooling(d, num_positions=47, position_weights=torch.nn.Parameter(torch.from_numpy(pw.copy())))
    ref = mod({"token_embeddings": torch.from_numpy(h), "attention_mask": torch.from_numpy(mask)})["sentence_embedding"].numpy()
    # raw path, useb_dense_retriever.py:253-270 (no clamp)
    ime = torch.from_numpy(mask).unsqueeze(-1).expand(B, S, d).float()
    wts = torch.from_numpy(pw)[:S].unsqueeze(0).unsqueeze(-1).expand(B, S, d)
    raw = (torch.sum(torch.from_numpy(h) * ime * wts, dim=1) / torch.sum(ime * wts, dim=1)).numpy()
    got = O.pool(h, mask, "learntmean", position_weights=pw)
    check("learntmean WeightedMeanPooling.py", got, ref, 2e-6)
    check("learntmean raw USEB path", got, raw, 2e-6)

    # fp8: rows with magnitudes over 11 orders, an all-zero row, a row whose amax sits on a scale boundary
    w = (rng.standard_normal((48, 128)) * np.exp(rng.uniform(-12, 4, size=(48, 1)))).astype(np.float32)
    w[5] = 0.0
    w[6, 0] = np.float32(448 * 2.0 ** -4); w[6, 1:] *= 1e-3
    w[7, 0] = np.float32(449 * 2.0 ** -4)
    w[8, ::3] *= 1e-4                                         # subnormal codes
    codes, scale = O.fp8_quantize_rows(w)
    x = torch.from_numpy(w / scale[:, None])
    assert float(x.abs().max()) <= 448.0
    t8 = x.to(torch.float8_e4m3fn)
    assert np.array_equal(t8.view(torch.uint8).numpy(), codes), "oracle e4m3fn encode != torch.float8_e4m3fn"
    deq = O.fp8_dequantize_rows(codes, scale)
    assert np.array_equal(deq, t8.float().numpy() * scale[:, None]), "oracle e4m3fn decode != torch"
    assert np.array_equal(torch.from_numpy(deq).to(torch.bfloat16).float().numpy(), deq), "dequantised weight not exact in bf16"
    assert np.all(np.log2(scale) == np.round(np.log2(scale)))
    print(f"  oracle-vs-torch fp8 e4m3fn: {codes.size} codes bit-identical; scales 2^{int(np.log2(scale.min()))}..2^{int(np.log2(scale.max()))}")
    np.savez_compressed(os.path.join(HERE, "extras.npz"), lm_hidden=h, lm_mask=mask.astype(np.int32), lm_pw=pw, lm_ref=ref,
                        fp8_w=w, fp8_codes=codes, fp8_scale=scale, fp8_deq=deq)
    print("wrote extras.npz")


TOK_WORDS = ["[UNK]", "[", "]", "{", "}", "what", "is", "the", "capital", "of", "france", "paris", "?", "and", "more",
             "words", "here", "a", "b", "c", "d", "e", "city", "river", "seine", "."]


def make_word_tokenizer(words=TOK_WORDS):
    """A real HF fast tokenizer built offline (no vocabulary files exist in this image): word-level over whitespace /
    punctuation.  Shared with tests/test_host_logic.py through the fixture's stored word list."""
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    tk = Tokenizer(models.WordLevel({w: i for i, w in enumerate(words)}, unk_token="[UNK]"))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    tok = PreTrainedTokenizerFast(tokenizer_object=tk, unk_token="[UNK]", eos_token="[UNK]")
    tok.pad_token = tok.eos_token
    return tok


def _tokenize_cases():
    """Token ids of the reference's two tokenisation legs on the same texts:
    (1) sentence-transformers path: Transformer.tokenize -> tokenize_bos_eos (Transformer.py:90-153) configured as
        SentenceBERTBOSEOS.__init__ does for specb and speca (sentence_bert_asym.py:33-63), fed the "[SOS]" / "{SOS}"
        prefixed texts of its encode_queries / encode_corpus (:67-79);
    (2) raw-HF path: the per-text loop of CustomEmbedder.embed (beir_dense_retriever.py:167-191), restated call for call
        with the same tokenizer methods."""
    import json
    TR = load_file_module("ref_transformer", f"{ST}/models/Transformer.py")
    queries = ["what is the capital of france ?", "paris", "a b c d e a b c d e a b c d e", "what is\nthe river"]
    docs = [{"title": "paris", "text": "paris is the capital of france . the river seine is here ."},
            {"title": "", "text": "a b c"}, {"title": "more words", "text": "and more words here and more words here and more"}]
    doc_texts = [(d["title"] + " " + d["text"]).strip() for d in docs]
    out = {"words": TOK_WORDS, "queries": queries, "docs": docs, "cases": []}
    for mode in ("specb", "speca"):
        for max_seq_length in (8, 12, 300):
            tok = make_word_tokenizer()
            me = types.SimpleNamespace(tokenizer=tok, max_seq_length=max_seq_length, do_lower_case=False, replace_bos=False,
                                       bos_spec_token_q=None, eos_spec_token_q=None, bos_spec_token_d=None,
                                       eos_spec_token_d=None, bos_spec_token_q_rep=None, bos_spec_token_d_rep=None)
            enc1 = lambda t: tok.encode(t, add_special_tokens=False)[0]  # noqa: E731
            if mode == "specb":                                   # sentence_bert_asym.py:33-50
                tok.add_tokens(["[SOS]", "{SOS}"], special_tokens=True)
                me.bos_spec_token_q, me.bos_spec_token_d = enc1("[SOS]"), enc1("{SOS}")
                me.bos_spec_token_q_rep, me.eos_spec_token_q = enc1("["), enc1("]")
                me.bos_spec_token_d_rep, me.eos_spec_token_d = enc1("{"), enc1("}")
                me.replace_bos = True
            else:                                                 # :52-63
                tok.add_tokens(["[SOS]", "[EOS]", "{SOS}", "{EOS}"], special_tokens=True)
                me.bos_spec_token_q, me.eos_spec_token_q = enc1("[SOS]"), enc1("[EOS]")
                me.bos_spec_token_d, me.eos_spec_token_d = enc1("{SOS}"), enc1("{EOS}")
            me.tokenize_bos_eos = types.MethodType(TR.Transformer.tokenize_bos_eos, me)

            def run(texts):
                feats = TR.Transformer.tokenize(me, texts)
                ids, mask = feats["input_ids"].tolist(), feats["attention_mask"].tolist()
                return [[t for t, m in zip(r, mr) if m] for r, mr in zip(ids, mask)]
            q_ids = run(["[SOS]" + q for q in queries])
            d_ids = run([("{SOS}" + d["title"] + " " + d["text"]).strip() if "title" in d else "{SOS}" + d["text"].strip()
                         for d in docs])
            out["cases"].append({"path": "st", "mode": mode, "max_seq_length": max_seq_length, "vocab_len": len(tok),
                                 "query_ids": q_ids, "doc_ids": d_ids})
    # raw-HF path, specb on/off
    for specb in (False, True):
        for maxseqlen in (8, 300):
            tok = make_word_tokenizer()
            max_token_len = maxseqlen - 2 if specb else maxseqlen              # :134-136
            bos_q, eos_q = tok.encode("["), tok.encode("]")                      # :146-150 (add_special_tokens default)
            bos_d, eos_d = tok.encode("{"), tok.encode("}")

            def raw(texts, is_query):
                res = []
                for txt in texts:
                    txt = txt.replace("\n", " ")
                    tokens = tok.convert_tokens_to_ids(tok.tokenize(txt))
                    # :182-184 prepare_for_model(add_special_tokens=True): GPT tokenizers add none (Transformer.py:119-120),
                    # and transformers 5.x dropped the method from fast tokenizers -> identity
                    ids = list(tokens[:max_token_len])
                    if specb:
                        ids = (bos_q + ids + eos_q) if is_query else (bos_d + ids + eos_d)
                    res.append(list(ids))
                return res
            out["cases"].append({"path": "raw", "mode": "specb" if specb else "none", "max_seq_length": maxseqlen,
                                 "vocab_len": len(tok), "query_ids": raw(queries, True), "doc_ids": raw(doc_texts, False)})
    with open(os.path.join(HERE, "tokenize.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(f"wrote tokenize.json ({len(out['cases'])} cases from the reference's Transformer.tokenize / embed loop)")


def _crossencoder_case():
    """Cross-encoder log-probabilities from the reference's own `_loglikelihood_tokens` (crossencoder/beir/sgptce.py:93-262,
    exec'd from its source: the module itself imports beir and runs a CLI at import) driving HF GPTNeoForCausalLM."""
    import json
    import collections
    import torch.nn.functional as F
    from transformers import GPTNeoConfig, GPTNeoForCausalLM
    src = open(f"{REF}/crossencoder/beir/sgptce.py").read().split("\n")
    ns = {"torch": torch, "F": F, "collections": collections, "tqdm": (lambda it, disable=False: it)}
    exec(compile("\n".join(src[92:262]), "sgptce.py[93:262]", "exec"), ns)          # group, Reorderer, chunks, _model_call, _loglikelihood_tokens
    cfg_kw = dict(vocab_size=211, max_position_embeddings=96, hidden_size=128, num_layers=4, num_heads=2, window_size=8)
    cfg = O.NeoConfig(**cfg_kw)
    w = O.synth_weights(cfg, seed=51, std=0.08)
    hc = GPTNeoConfig(vocab_size=cfg.vocab_size, max_position_embeddings=cfg.max_position_embeddings, hidden_size=cfg.hidden_size,
                      num_layers=cfg.num_layers, num_heads=cfg.num_heads, intermediate_size=cfg.intermediate_size,
                      window_size=cfg.window_size, attention_types=[[["global", "local"], cfg.num_layers // 2]],
                      layer_norm_epsilon=cfg.layer_norm_epsilon, attention_dropout=0.0, resid_dropout=0.0, embed_dropout=0.0,
                      tie_word_embeddings=True)
    hc._attn_implementation = "eager"
    lm = GPTNeoForCausalLM(hc).eval()
    missing, unexpected = lm.load_state_dict({"transformer." + k: torch.from_numpy(v.copy()) for k, v in w.items()}, strict=False)
    assert not unexpected, unexpected
    lm.tie_weights()
    assert torch.equal(lm.lm_head.weight, lm.transformer.wte.weight)
    rng = np.random.default_rng(510)
    max_length, instruction_len = 48, 3
    reqs = []
    for n_ctx, n_cont in [(20, 5), (3, 1), (40, 9), (70, 12), (10, 30), (47, 2), (5, 5), (60, 48)]:   # several exceed max_length
        reqs.append((("c", "q"), rng.integers(0, 211, size=n_ctx).tolist(), rng.integers(0, 211, size=n_cont).tolist()))
    ref = ns["_loglikelihood_tokens"](reqs, lm, max_length, torch.device("cpu"), disable_tqdm=True, batch_size=1,
                                      instruction_len=instruction_len)
    ref_b3 = ns["_loglikelihood_tokens"](reqs, lm, max_length, torch.device("cpu"), disable_tqdm=True, batch_size=3,
                                         instruction_len=instruction_len)
    got = O.loglikelihood_tokens(w, cfg, reqs, max_length, instruction_len)
    check("cross-encoder log-likelihoods (batch 1)", got, ref, 2e-4)
    check("cross-encoder log-likelihoods (batch 3, right-padded)", got, ref_b3, 2e-4)
    with open(os.path.join(HERE, "crossencoder.json"), "w") as f:
        json.dump({"cfg": cfg_kw, "seed": 51, "std": 0.08, "max_length": max_length, "instruction_len": instruction_len,
                   "requests": [[r[1], r[2]] for r in reqs], "loglikelihood": ref}, f)
    print("wrote crossencoder.json", [round(x, 3) for x in ref])


def _scoring_cases(U, ES):
    rng = np.random.default_rng(7)
    a = rng.standard_normal((50, 100)).astype(np.float32)
    b = rng.standard_normal((37, 100)).astype(np.float32)
    b[3] = 0.0                                     # zero-norm row (eps path of F.normalize)
    cos = U.cos_sim(torch.from_numpy(a), torch.from_numpy(b)).numpy()
    dot = U.dot_score(torch.from_numpy(a), torch.from_numpy(b)).numpy()
    nrm = U.normalize_embeddings(torch.from_numpy(a)).numpy()
    pcs = U.pairwise_cos_sim(torch.from_numpy(a[:37]), torch.from_numpy(b)).numpy()
    check("cos_sim", O.cos_sim(a, b), cos, 1e-6)
    check("dot_score", O.dot_score(a, b), dot, 1e-5)
    check("normalize", O.normalize(a), nrm, 1e-6)
    check("pairwise_cos_sim", O.pairwise_cos_sim(a[:37], b), pcs, 1e-6)
    # semantic_search, the shapes of tests/test_util.py:33-53
    docs_e = rng.standard_normal((1000, 100)).astype(np.float32)
    q_e = rng.standard_normal((20, 100)).astype(np.float32)
    hits = U.semantic_search(torch.from_numpy(q_e), torch.from_numpy(docs_e), top_k=10,
                             query_chunk_size=5, corpus_chunk_size=17)
    ss_idx = np.array([[h["corpus_id"] for h in row] for row in hits], dtype=np.int64)
    ss_val = np.array([[h["score"] for h in row] for row in hits], dtype=np.float32)
    ohits = O.semantic_search(q_e, docs_e, top_k=10, query_chunk_size=5, corpus_chunk_size=17)
    assert (np.array([[h["corpus_id"] for h in row] for row in ohits]) == ss_idx).all()
    check("semantic_search scores", np.array([[h["score"] for h in row] for row in ohits]), ss_val, 1e-6)

    # exact_search.search with a fake encoder (embeddings looked up by id), 3 chunks, anisotropic embeddings
    nd, nq, dd, topk = 700, 13, 96, 10
    ce = rng.standard_normal((nd, dd)).astype(np.float32)
    qe = rng.standard_normal((nq, dd)).astype(np.float32)
    ce[:, :3] *= 30.0
    qe[:, :3] *= 30.0                              # SGPT-like anisotropy
    corpus = {f"d{i}": {"title": "t" * int(rng.integers(0, 5)), "text": "x" * int(rng.integers(1, 50))} for i in range(nd)}
    queries = {f"q{i}": "q" for i in range(nq)}
    # make query ids collide with corpus ids for two queries (the self-match rule :118)
    queries = {("d5" if i == 2 else "d9" if i == 4 else k): v for i, (k, v) in enumerate(queries.items())}
    cvec = {f"d{i}": ce[i] for i in range(nd)}
    qvec = {k: qe[i] for i, k in enumerate(queries)}

    class FakeModel:
        def encode_queries(self, qs, batch_size, **kw):
            return torch.from_numpy(np.stack([qvec[qid] for qid, _ in qs]))

        def encode_corpus(self, cs, batch_size, **kw):
            return torch.from_numpy(np.stack([cvec[cid] for cid, _ in cs]))

    es_out = {}
    for fn in ("cos_sim", "dot"):
        s = ES.DenseRetrievalExactSearch(FakeModel(), batch_size=8, corpus_chunk_size=256)
        res = s.search(corpus, queries, topk, fn)
        # oracle restatement on the same sorted corpus order (:66-71)
        cids = sorted(corpus, key=lambda k: len(corpus[k].get("title", "") + corpus[k].get("text", "")), reverse=True)
        cemb = np.stack([cvec[c] for c in cids])
        ores = O.exact_search(np.stack([qvec[q] for q in queries]), list(queries), cemb, cids, topk, fn, chunk_size=256)
        for qid in queries:
            assert set(res[qid]) == set(ores[qid]), (fn, qid)
            check(f"exact_search[{fn}] {qid}", [ores[qid][c] for c in res[qid]], [res[qid][c] for c in res[qid]], 1e-4)
        es_out[fn] = res
    try:
        ES.DenseRetrievalExactSearch(FakeModel()).search(corpus, queries, topk, "euclid")
        raise AssertionError("expected ValueError")
    except ValueError as e:
        bad_fn_msg = str(e)

    import json
    np.savez_compressed(
        os.path.join(HERE, "scoring.npz"), a=a, b=b, cos=cos, dot=dot, nrm=nrm, pcs=pcs,
        ss_docs=docs_e, ss_q=q_e, ss_idx=ss_idx, ss_val=ss_val,
        es_corpus_emb=ce, es_query_emb=qe,
        es_json=np.array(json.dumps(dict(corpus=corpus, queries=list(queries.keys()), top_k=topk, chunk=256,
                                         results=es_out, bad_fn_msg=bad_fn_msg))))
    print("wrote scoring.npz")


if __name__ == "__main__":
    main()
