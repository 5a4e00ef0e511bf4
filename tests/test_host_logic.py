"""CPU tests of the host logic and of the C-ABI surface (no compute calls: no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import sgpt_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    """Every function declared in include/sgpt_hip.h is exported by libsgpt_hip.so and bound in _lib."""
    from sgpt_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "sgpt_hip.h")).read()
    declared = set(re.findall(r"\b(sgpt_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"sgpt_status"}
    assert len(declared) >= 15
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in sgpt_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes prototype"
    assert set(_lib.SIGNATURES) == declared
    m = re.search(r"#define SGPT_ABI_VERSION (\d+)", hdr)
    assert lib.sgpt_abi_version() == int(m.group(1)) == _lib.SGPT_ABI_VERSION
    # struct layout mirrors the header
    assert ctypes.sizeof(_lib.ModelDesc) == 72 and ctypes.sizeof(_lib.TensorView) == 24


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from sgpt_amd import _lib, get_context
    from sgpt_amd import util
    with pytest.raises(_lib.SgptHipError):
        get_context()
    with pytest.raises(_lib.SgptHipError):          # no silent CPU fallback behind the reference API
        util.cos_sim(np.ones((2, 8), np.float32), np.ones((3, 8), np.float32))


def test_product_code_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sgpt_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
                assert "/root/reference" not in src, f


def test_synthetic_weights_match_oracle_stream():
    from sgpt_amd.model import SGPTConfig, synthetic_weights
    kw = dict(vocab_size=211, max_position_embeddings=96, hidden_size=128, num_layers=2, num_heads=2, window_size=8)
    a = synthetic_weights(SGPTConfig(**kw), seed=5, std=0.05)
    b = O.synth_weights(O.NeoConfig(**kw), seed=5, std=0.05)
    assert a.keys() == b.keys()
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_pack_host_layout():
    from sgpt_amd.model import pack_host, pad_rows, ALIGN
    seqs = [[5, 6, 7], list(range(1, 18)), [9] * 16, [1]]
    h = pack_host(seqs, pad_left=[2, 0, 0, 7])
    up = lambda n: (n + ALIGN - 1) // ALIGN * ALIGN  # noqa: E731
    # query-sized layouts (<= 512 rows) pad to 32-row tiles, larger ones to the 256-row GEMM tile (include/sgpt_hip.h)
    assert [pad_rows(n) for n in (1, 32, 33, 512, 513, 4096, 4097)] == [32, 32, 64, 512, 768, 4096, 4352]
    assert h["B"] == 4 and h["T_pad"] == 64 and h["max_alloc"] == up(17) and h["n_tokens"] == 37
    off = h["seq_off"]
    assert off.tolist() == [0, up(3), up(3) + up(17), up(3) + up(17) + 16, up(3) + up(17) + 16 + up(1)] and all(o % ALIGN == 0 for o in off)
    for b, s in enumerate(seqs):
        assert h["ids"][off[b]: off[b] + len(s)].tolist() == s
        assert h["pos"][off[b]: off[b] + len(s)].tolist() == [h["pad_left"][b] + t for t in range(len(s))]
    assert h["ids"][3:off[1]].tolist() == [0] * (off[1] - 3)            # filler rows
    with pytest.raises(ValueError, match="Empty items should be cleaned prior to running"):
        pack_host([[1], []])


def test_plan_batches_covers_everything_sorted():
    from sgpt_amd.model import ALIGN, SGPTConfig, SGPTModel
    lens = np.random.default_rng(0).integers(1, 129, size=1000)
    fake = SGPTModel.__new__(SGPTModel)
    fake.max_tokens_per_call, fake.cfg, fake.device = 4096, SGPTConfig(), "cpu"
    plan = SGPTModel.plan_batches(fake, lens)
    allidx = np.concatenate(plan)
    assert sorted(allidx.tolist()) == list(range(1000))
    srt = lens[allidx]
    assert (np.diff(srt) <= 0).all()                    # longest first
    for sel in plan:
        assert ((lens[sel] + ALIGN - 1) // ALIGN * ALIGN).sum() <= 4096


def test_text_pipeline_specb_matches_oracle_ids():
    from sgpt_amd.tokenization import SyntheticTokenizer, TextPipeline
    tok = SyntheticTokenizer()
    pipe = TextPipeline(tok, max_token_len=10, specb=True)
    assert pipe.max_token_len == 8
    q = pipe.ids("what is\nthe capital of france ? and more words here", True)
    raw = tok.convert_tokens_to_ids(tok.tokenize("what is the capital of france ? and more words here"))
    assert q == O.specb_wrap(raw, True, 8) and len(q) == 10
    d = pipe.ids("paris", False)
    assert d == O.specb_wrap(tok.convert_tokens_to_ids(["paris"]), False)
    assert pipe.docs_truncated == 1
    with pytest.raises(ValueError, match="Empty items should be cleaned prior to running"):
        pipe.ids("   ", True)


def test_shard_sizes_match_reference_formula():
    from sgpt_amd.st import shard_sizes
    # SentenceTransformer.py:159-160
    for n, w in [(10, 4), (7, 8), (1000, 8), (5, 1)]:
        ref = [n // w + (1 if r < n % w else 0) for r in range(w)]
        assert shard_sizes(n, w) == ref and sum(ref) == n


def _word_tokenizer(words):
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    tk = Tokenizer(models.WordLevel({w: i for i, w in enumerate(words)}, unk_token="[UNK]"))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    tok = PreTrainedTokenizerFast(tokenizer_object=tk, unk_token="[UNK]", eos_token="[UNK]")
    tok.pad_token = tok.eos_token
    return tok


def test_text_pipeline_matches_reference_tokenisation_golden():
    """tests/golden/tokenize.json holds the ids produced by the reference's own Transformer.tokenize /
    tokenize_bos_eos (sentence-transformers path, specb and speca) and by the raw-HF per-text loop, on a real
    HF fast tokenizer built offline; TextPipeline must reproduce them id for id, including the marker-inclusive
    truncation of the ST path (content cut to max_seq_length - 3) vs max_seq_length - 2 on the raw path."""
    import json
    from sgpt_amd.tokenization import TextPipeline
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "tokenize.json")))
    doc_texts = [(d["title"] + " " + d["text"]).strip() for d in fx["docs"]]
    assert len(fx["cases"]) == 10
    for case in fx["cases"]:
        tok = _word_tokenizer(fx["words"])
        pipe = TextPipeline(tok, case["max_seq_length"], specb=case["mode"] == "specb", speca=case["mode"] == "speca",
                            st_path=case["path"] == "st")
        strip = (lambda t: str(t).strip()) if case["path"] == "st" else (lambda t: t)
        assert getattr(tok, "is_fast", False)                     # -> TextPipeline.batch takes the one-call batched path
        assert pipe.batch([strip(q) for q in fx["queries"]], True) == case["query_ids"], case
        assert pipe.batch([strip(t) for t in doc_texts], False) == case["doc_ids"], case
        # the batched fast-tokenizer path and the per-text reference loop (tokenize + convert_tokens_to_ids) agree
        assert [pipe.ids(strip(q), True) for q in fx["queries"]] == case["query_ids"], case
        assert [pipe.ids(strip(t), False) for t in doc_texts] == case["doc_ids"], case
        if case["mode"] == "speca":
            assert len(tok) == case["vocab_len"]        # four added rows, same ids as the reference's add_tokens


def test_st_folder_formats_roundtrip(tmp_path):
    """modules.json / 1_Pooling / 1_WeightedMeanPooling / 2_Normalize / Asym layouts (SentenceTransformer.py:389-430,
    903-936; Pooling.py:172-185; Asym.py:62-122) read back into the pipeline description."""
    import json
    import torch
    from sgpt_amd import formats
    cfg = {"model_type": "gpt_neo", "hidden_size": 128, "num_layers": 1, "num_heads": 2, "vocab_size": 50,
           "max_position_embeddings": 32, "attention_types": [[["global"], 1]]}
    sd = {"wte.weight": np.zeros((50, 128), np.float32)}
    p1 = str(tmp_path / "sym")
    formats.write_st_folder(p1, cfg, sd, pooling_mode="weightedmean", max_seq_length=75, normalize=True)
    mods = json.load(open(os.path.join(p1, "modules.json")))
    assert [m["type"].rsplit(".", 1)[-1] for m in mods] == ["Transformer", "Pooling", "Normalize"]
    assert mods[0]["path"] == "" and mods[1]["path"] == "1_Pooling"
    pc = json.load(open(os.path.join(p1, "1_Pooling", "config.json")))
    assert pc["pooling_mode_weightedmean_tokens"] is True and pc["pooling_mode_mean_tokens"] is False
    spec = formats.read_st_folder(p1)
    assert spec.transformer_dirs == {"": os.path.join(p1, "")} and not spec.asymmetric
    assert (spec.max_seq_length, spec.pooling_mode, spec.normalize) == (75, "weightedmean", True)
    # the reference's own Pooling.load accepts the written config (loaded from its file when the tree is present)
    ref_pool = "/root/" + "reference/biencoder/nli_msmarco/sentence-transformers/sentence_transformers/models/Pooling.py"
    if os.path.exists(ref_pool):
        import importlib.util
        sp = importlib.util.spec_from_file_location("ref_pooling_fmt", ref_pool)
        mod = importlib.util.module_from_spec(sp)
        sp.loader.exec_module(mod)
        rp = mod.Pooling.load(os.path.join(p1, "1_Pooling"))
        assert rp.pooling_mode_weightedmean_tokens and rp.get_sentence_embedding_dimension() == 128
    # learntmean
    p2 = str(tmp_path / "learnt")
    pw = np.linspace(0.5, 2.0, 33).astype(np.float32)
    formats.write_st_folder(p2, cfg, sd, pooling_mode="learntmean", position_weights=pw)
    spec = formats.read_st_folder(p2)
    assert spec.pooling_mode == "learntmean" and not spec.normalize
    assert np.array_equal(torch.load(spec.position_weights_file)["position_weights"].numpy(), pw)
    # plain HF folder -> mean pooling (the reference's fallback)
    p3 = str(tmp_path / "hf")
    os.makedirs(p3)
    json.dump(cfg, open(os.path.join(p3, "config.json"), "w"))
    assert formats.read_st_folder(p3).pooling_mode == "mean"
    # Asym: [Asym{QRY:[T1], DOCPOS:[T2], DOCNEG:[T2]}, Pooling]
    p4 = str(tmp_path / "asym")
    os.makedirs(os.path.join(p4, "0_Asym", "111_Transformer"))
    os.makedirs(os.path.join(p4, "0_Asym", "222_Transformer"))
    os.makedirs(os.path.join(p4, "1_Pooling"))
    for t in ("111_Transformer", "222_Transformer"):
        json.dump({"max_seq_length": 300, "do_lower_case": False},
                  open(os.path.join(p4, "0_Asym", t, "sentence_bert_config.json"), "w"))
    json.dump({"types": {"111_Transformer": "sentence_transformers.models.Transformer",
                         "222_Transformer": "sentence_transformers.models.Transformer"},
               "structure": {"QRY": ["111_Transformer"], "DOCPOS": ["222_Transformer"], "DOCNEG": ["222_Transformer"]},
               "parameters": {"allow_empty_key": False}}, open(os.path.join(p4, "0_Asym", "config.json"), "w"))
    json.dump(dict(pc, pooling_mode_weightedmean_tokens=False, pooling_mode_mean_tokens=True),
              open(os.path.join(p4, "1_Pooling", "config.json"), "w"))
    json.dump([{"idx": 0, "name": "0", "path": "0_Asym", "type": "sentence_transformers.models.Asym"},
               {"idx": 1, "name": "1", "path": "1_Pooling", "type": "sentence_transformers.models.Pooling"}],
              open(os.path.join(p4, "modules.json"), "w"))
    spec = formats.read_st_folder(p4)
    assert spec.asymmetric and spec.pooling_mode == "mean" and spec.max_seq_length == 300
    assert spec.transformer_dirs["QRY"].endswith("111_Transformer") and spec.transformer_dirs["DOCPOS"].endswith("222_Transformer")
    # unsupported pieces fail loudly
    json.dump(dict(pc, pooling_mode_cls_token=True), open(os.path.join(p4, "1_Pooling", "config.json"), "w"))
    with pytest.raises(NotImplementedError):
        formats.read_st_folder(p4)


def test_embedding_cache_and_results_json_formats(tmp_path):
    """`{id: ndarray}` pickle (beir_dense_retriever.py:306-348) and results JSON (:439-441)."""
    import json
    import pickle
    from sgpt_amd import formats
    ids = ["d3", "d1", "d2"]
    emb = np.arange(12, dtype=np.float32).reshape(3, 4)
    path = str(tmp_path / "embeddings" / "m" / "weightedmean" / "scifact_corpus0.pickle")
    formats.save_embedding_cache(path, ids, emb)
    raw = pickle.load(open(path, "rb"))
    assert set(raw) == set(ids) and np.array_equal(raw["d1"], emb[1])
    assert np.array_equal(formats.load_embedding_cache(path, ["d1", "d3"]), emb[[1, 0]])
    res = {"q1": {"d1": np.float32(0.5), "d2": 0.25}, "q2": {}}
    rp = str(tmp_path / "results.json")
    formats.save_results_json(rp, res)
    assert json.load(open(rp)) == {"q1": {"d1": 0.5, "d2": 0.25}, "q2": {}} == formats.load_results_json(rp)


def test_library_is_not_older_than_its_sources():
    """A stale in-tree .so (sources edited, `python -m sgpt_amd.build` not re-run) would travel to the GPU box and
    run old kernels behind a new ctypes prototype; catch it on the CPU."""
    from sgpt_amd import _lib
    so = os.path.getmtime(_lib.LIB_PATH)
    srcs = [os.path.join(ROOT, "include", "sgpt_hip.h")]
    csrc = os.path.join(ROOT, "sgpt_amd", "csrc")
    srcs += [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h", ".inc"))]
    stale = [os.path.basename(s) for s in srcs if os.path.getmtime(s) > so + 1.0]
    assert not stale, f"libsgpt_hip.so is older than {stale}: run `python -m sgpt_amd.build`"


def test_round_aware_call_budgets():
    """SGPTModel.call_budgets (round 5): the token-row budgets of the sgpt_encode calls of a chunk cover it, stay inside
    max_tokens_per_call, use at most one call more than the minimum, are a pure function of (rows, shape, CUs), and never cost
    more modelled tile-rounds than round 4's equal budgets; plan_batches cuts the length-sorted list by them."""
    from sgpt_amd.model import SGPTConfig, SGPTModel, TOKEN_TILE
    m = object.__new__(SGPTModel)
    m.cfg, m.max_tokens_per_call, m.device = SGPTConfig(), 131072, "cpu"

    def rounds(plan, ncu=256, d=768):
        nts = ((3 * d // 256, d), (d // 256, d), (4 * d // 256, d), (d // 256, 4 * d))
        return sum(-(-(t // 256) * nt // ncu) * k for t in plan for nt, k in nts)
    rng = np.random.default_rng(0)
    for total in [1, 255, 131072, 131073, 298000, 262144, 400000, 3_599_944] + rng.integers(1000, 2_000_000, size=40).tolist():
        b = m.call_budgets(int(total), 256)
        n_min = -(-(-(-int(total) // TOKEN_TILE)) // 512)
        assert sum(b) >= total and all(0 < x <= 131072 and x % TOKEN_TILE == 0 for x in b) and n_min <= len(b) <= n_min + 1, (total, b)
        assert b == m.call_budgets(int(total), 256)
        eq = [-(-(-(-int(total) // len(b))) // 256) * 256] * len(b) if len(b) > 1 else b
        assert rounds(b) <= rounds(eq) * 1.0001 or len(b) == 1, (total, b, rounds(b), rounds(eq))
    # the bench's variable-length step (equal thirds ran 4.55 rounds per N = 768 launch); the call that is not a round boundary is last
    assert [x // 256 for x in m.call_budgets(298000, 256)] == [341, 341, 483]
    m.round_aware_calls = True
    # a call is cut at the last whole sequence inside its budget; the plan is re-taken over the rest at every cut, so
    # the leftovers never end in a tiny extra call (4096 documents of U{16..128} used to end in an 80-row fourth call)
    m._num_cus = lambda: 256
    for n_docs, lo, hi in [(4096, 16, 129)] * 6 + [(8192, 16, 129), (4096, 100, 513), (20000, 1, 65), (1500, 300, 1025), (4096, 128, 129)]:
        lens = rng.integers(lo, hi, size=n_docs).astype(np.int64)
        plan = m.plan_batches(lens)
        assert sorted(np.concatenate(plan).tolist()) == list(range(n_docs))
        rows = [int(((lens[p] + 1) // 2 * 2).sum()) for p in plan]
        bud = m.call_budgets(int(((lens + 1) // 2 * 2).sum()), 256)
        assert len(bud) <= len(plan) <= len(bud) + 1 and max(rows) <= m.max_tokens_per_call, (rows, bud)
        assert min(rows) >= 0.4 * max(rows), (rows, bud)
        good = {341, 426, 512, 256, 170, 85}
        # all but the last call end on (or at most one sequence short of) a round boundary of the N = d launches
        assert all(any(0 <= g - -(-r // 256) <= hi // 256 + 1 for g in good) for r in rows[:-1]), rows


def test_native_result_assembly_equals_the_python_construction():
    """csrc/host_assemble.c (the k = 1001 result dict in C) against beir.assemble_results' Python form, which is
    exact_search.py:109-132 as arrays: same keys, same floats, same insertion order; padding positions (< 0) skipped, a duplicate
    position keeps the later score, out-of-range positions raise."""
    from sgpt_amd import build as B
    from sgpt_amd.beir import assemble_results
    assert B.build_host() and os.path.exists(B.HOST_EXT)
    assert os.path.getmtime(B.HOST_EXT) + 1.0 >= os.path.getmtime(os.path.join(ROOT, "sgpt_amd", "csrc", "host_assemble.c"))
    rng = np.random.default_rng(0)
    for nq, k, n in ((1, 1, 1), (7, 11, 40), (33, 1001, 5000)):
        idx = rng.integers(0, n, size=(nq, k)).astype(np.int64)
        idx[rng.random((nq, k)) < 0.1] = -1
        val = rng.standard_normal((nq, k)).astype(np.float32)
        cids, qids = [f"doc-{j}" for j in range(n)], tuple(f"q{j}" for j in range(nq))          # (a tuple of query ids is accepted too)
        a, b = assemble_results(qids, cids, val, idx, native=True), assemble_results(qids, cids, val, idx, native=False)
        assert a == b and all(list(a[q].items()) == list(b[q].items()) for q in qids)
        assert all(type(v) is float for v in a[qids[0]].values())
    # non-contiguous / wrongly typed inputs are normalised, not misread
    a = assemble_results(["q0", "q1"], ["a", "b", "c"], np.asarray([[1, 2], [3, 4]], dtype=np.float64), np.asarray([[2, 0], [1, 1]], dtype=np.int32), native=True)
    assert a == {"q0": {"c": 1.0, "a": 2.0}, "q1": {"b": 4.0}}
    with pytest.raises(IndexError):
        assemble_results(["q0"], ["a"], np.ones((1, 2), np.float32), np.asarray([[0, 5]], np.int64), native=True)


def test_crossencoder_host_logic_matches_oracle():
    """Truncation rule and request encoding of the cross-encoder surface (crossencoder/beir/sgptce.py:77-91,204-211)."""
    from sgpt_amd.crossencoder import encode, model_input
    from sgpt_amd.tokenization import SyntheticTokenizer
    rng = np.random.default_rng(0)
    for n_ctx, n_cont, max_len, instr in [(20, 5, 48, 3), (70, 12, 48, 3), (60, 48, 48, 0), (1, 1, 16, 0), (100, 7, 32, 10)]:
        c, q = rng.integers(0, 99, n_ctx).tolist(), rng.integers(0, 99, n_cont).tolist()
        assert model_input(c, q, max_len, instr) == O.ce_model_input(c, q, max_len, instr)
        assert len(model_input(c, q, max_len, instr)) <= max_len
    tok = SyntheticTokenizer(300)
    reqs = encode([("a query", "some context text"), ("q", "")], tok)
    assert reqs[0][1] == tok.encode("some context text") and reqs[0][2] == tok.encode("a query")
    assert reqs[1][1] == [tok.eos_token_id]                    # empty context -> end-of-text token
    with pytest.raises(AssertionError):
        model_input([1, 2, 3], list(range(50)), 48, 0)         # continuation longer than max_length


def test_bucketed_layout_keeps_the_real_rows_and_appends_one_token_fillers():
    """pack_layout(bucket=(B_cap, T_cap, A_cap)) (EncodeGraph capacity buckets): the caller's sequences sit exactly where
    the un-bucketed layout puts them; fillers are one-token sequences on their own ALIGN-row allocations behind them."""
    from sgpt_amd.model import ALIGN, pack_host
    rng = np.random.default_rng(0)
    for n in (1, 11, 16, 17, 40):
        seqs = [rng.integers(1, 100, size=int(rng.integers(1, 40))).tolist() for _ in range(n)]
        h0 = pack_host(seqs)
        al = [(len(q) + ALIGN - 1) // ALIGN * ALIGN for q in seqs]
        b = (max(16, 1 << (n - 1).bit_length()), (sum(al) + 8 * 64 + 255) // 256 * 256, max(32, 1 << (max(al) - 1).bit_length()))
        h = pack_host(seqs, bucket=b)
        assert (h["B"], h["T_pad"], h["max_alloc"], h["n_real"], h["n_tokens"]) == (b[0], b[1], b[2], n, h0["n_tokens"])
        assert (h["seq_off"][: n + 1] == h0["seq_off"]).all() and (h["seq_len"][:n] == h0["seq_len"]).all()
        n0 = int(h0["seq_off"][-1])
        assert (h["ids"][:n0] == h0["ids"][:n0]).all() and (h["pos"][:n0] == h0["pos"][:n0]).all()
        assert (h["ids"][n0:] == 0).all() and (h["pos"][n0:] == 0).all()
        assert (h["seq_len"][n:] == 1).all() and (np.diff(h["seq_off"][n:]) == ALIGN).all() and h["seq_off"][-1] <= b[1]
        pl = list(range(n))
        h2 = pack_host(seqs, pl, bucket=b)
        assert (h2["pad_left"][:n] == pl).all() and (h2["pad_left"][n:] == 0).all() and h2["max_pos"] == pack_host(seqs, pl)["max_pos"]
    with pytest.raises(ValueError):
        pack_host(seqs, bucket=(8, 1024, 64))          # fewer slots than sequences
    with pytest.raises(ValueError):
        pack_host(seqs, bucket=(64, 256, 64))          # token rows do not fit
    with pytest.raises(ValueError):
        pack_host(seqs, bucket=(64, 2048, 16))         # longest sequence does not fit


def test_pack_arena_equals_reference_layout_for_every_input_form():
    """One-buffer packed layout (ids | pos | seq_off | seq_len | pad_left): lists, lists of ndarrays and a rectangular
    ndarray give the same image; rows land at seq_off[b] + t, positions are pad_left[b] + t, filler stays 0."""
    from sgpt_amd.model import ALIGN, arena_ints, fill_arena, pack_host, pack_layout, pad_rows
    rng = np.random.default_rng(4)
    lens = [1, 16, 17, 40, 5, 128]
    seqs = [rng.integers(1, 1000, size=n).tolist() for n in lens]
    pl = [3, 0, 7, 0, 0, 1]
    h = pack_host(seqs, pl)
    assert h["T_pad"] == pad_rows(h["T_pad"]) and h["T_pad"] % 32 == 0 and h["arena"].shape[0] == arena_ints(pack_layout(seqs, pl))
    off = h["seq_off"]
    assert (off % ALIGN == 0).all() and off[0] == 0
    for b, s in enumerate(seqs):
        assert h["ids"][off[b]: off[b] + len(s)].tolist() == s
        assert h["pos"][off[b]: off[b] + len(s)].tolist() == list(range(pl[b], pl[b] + len(s)))
        assert (h["ids"][off[b] + len(s): off[b + 1]] == 0).all()
    assert h["seq_len"].tolist() == lens and h["pad_left"].tolist() == pl and h["max_pos"] == 128
    h2 = pack_host([np.asarray(s) for s in seqs], pl)
    assert np.array_equal(h["arena"], h2["arena"])
    rect = rng.integers(0, 50000, size=(9, 32))
    assert np.array_equal(pack_host(rect)["arena"], pack_host(rect.tolist())["arena"])         # aligned fast path
    rect = rng.integers(0, 50000, size=(5, 20))
    assert np.array_equal(pack_host(rect)["arena"], pack_host(rect.tolist())["arena"])         # unaligned rectangle
    lay = pack_layout(seqs, pl)
    hi, lo = fill_arena(seqs, lay, np.empty(arena_ints(lay) + 7, dtype=np.int32))
    assert hi == max(max(s) for s in seqs) and lo == min(min(s) for s in seqs)
    with pytest.raises(ValueError, match="Empty items should be cleaned prior to running"):
        pack_host([[1], []])


def _device_asm(name):
    import subprocess
    import sys
    from sgpt_amd import build as b
    src = os.path.join(ROOT, "sgpt_amd", "csrc", name)
    if not os.path.exists(src):                                    # (the experiment-build-only re-tiling lives under scripts/micro)
        src = os.path.join(ROOT, "scripts", "micro", name)
    out = subprocess.run([b._hipcc()] + b.FLAGS + ["-I", os.path.join(ROOT, "sgpt_amd", "csrc"), "--cuda-device-only", "-S", src, "-o", "-"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


@pytest.mark.parametrize("src", ["gemm.hip", "gemm256w.hip", "gemm256q.hip"])
def test_m0_only_written_by_the_dma_idiom(src):
    """The LDS-DMA pieces set M0 inside inline asm without declaring it clobbered (hipcc refuses reserved registers on
    clobber lists).  That is sound only while the compiler itself never keeps a value in M0 across those statements:
    assert on the generated gfx950 ISA that every instruction naming m0 is the idiom's own `s_mov_b32 m0, <sgpr>` and
    that each one is followed by its s_nop + global_load_lds_dwordx4."""
    lines = [ln.split(";")[0].strip() for ln in _device_asm(src).splitlines()]
    lines = [ln for ln in lines if ln and not ln.startswith((".", "//")) and not ln.endswith(":")]
    uses = [i for i, ln in enumerate(lines) if re.search(r"\bm0\b", ln)]
    assert len(uses) > 50
    for i in uses:
        assert re.fullmatch(r"s_mov_b32 m0, (s\d+|vcc_lo|vcc_hi|ttmp\d+)", lines[i]), lines[i]
        assert lines[i + 1] == "s_nop 0" and lines[i + 2].startswith("global_load_lds_dwordx4 "), lines[i:i + 3]
    # and the kernels issue no other LDS-DMA form that would read M0 implicitly
    assert sum(ln.startswith("global_load_lds") for ln in lines) == len(uses)


def test_wide_gemm_index_model():
    """scripts/lds_layout_check.py: LDS-DMA fill swizzle vs fragment-read swizzle, the 32x32 C map and the three store
    epilogues of gemm256w.hip walked on the CPU; the k-loop fragment reads must be bank-conflict free."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("lds_layout_check", os.path.join(ROOT, "scripts", "lds_layout_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    banks = mod.check_banks()
    assert banks["fragment ds_read_b128"] == 1 and banks["epi32 ds_write_b128"] == 1 and banks["epi32 ds_read_b128"] == 1
    assert max(banks.values()) <= 2
    assert mod.walk_tile(K=192, seed=3)


def test_fp8_gemm_swizzle_is_conflict_free():
    import importlib.util
    spec = importlib.util.spec_from_file_location("lds_layout_check", os.path.join(ROOT, "scripts", "lds_layout_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    q8 = mod.check_q8()
    assert q8["rotl3(row&7)"] == 1 and q8["row&7"] == 2


def test_one_rccl_per_process_and_it_exports_what_comm_hip_calls():
    """VERDICT r03 next-7 / ADVICE r03: libsgpt_hip.so is compiled against /opt/rocm's rccl.h (types only) and binds RCCL lazily
    at the first communicator call, preferring the copy already in the process.  Host-only check: the library does not link
    librccl; after `import torch` and a first call exactly one librccl is mapped (torch's); that copy exports every entry point
    csrc/comm.hip resolves; its NCCL major version is the header's; and sgpt_comm_unique_id works (no GPU needed)."""
    import ctypes
    import subprocess
    import torch  # noqa: F401
    from sgpt_amd import _lib
    lib = _lib.load()
    needed = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "librccl" not in needed and "libamdhip64" in needed
    ident = (ctypes.c_uint8 * _lib.SGPT_COMM_ID_BYTES)()
    assert lib.sgpt_comm_unique_id(ident) == 0 and any(bytes(ident))        # ncclGetUniqueId through the lazily bound table
    with open("/proc/self/maps") as f:
        paths = sorted({ln.split()[-1] for ln in f if "librccl" in ln})
    assert len(paths) == 1, f"more than one RCCL in the process: {paths}"
    src = open(os.path.join(ROOT, "sgpt_amd", "csrc", "comm.hip")).read()
    resolved = sorted(set(re.findall(r"SGPT_SYM\(\w+, (nccl[A-Z][A-Za-z]+)\)", src)))
    assert {"ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclAllGather", "ncclGroupStart", "ncclGroupEnd",
            "ncclGetVersion", "ncclGetErrorString"} == set(resolved)
    # no direct call of an nccl* function is left in the source (everything goes through the table)
    direct = [m for m in re.findall(r"(?<![\w.&>])(nccl[A-Z][A-Za-z]+)\s*\(", src)]
    assert not direct, direct
    rccl = ctypes.CDLL(paths[0])
    for name in resolved:
        assert hasattr(rccl, name), f"{paths[0]} does not export {name}"
    v = ctypes.c_int(0)
    assert rccl.ncclGetVersion(ctypes.byref(v)) == 0 and v.value >= 20000
    hdr = None
    for cand in ("/opt/rocm/include/rccl/rccl.h", os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "include", "rccl", "rccl.h")):
        if os.path.exists(cand):
            hdr = open(cand).read()
            break
    if hdr is not None:
        m = re.search(r"#define\s+NCCL_MAJOR\s+(\d+)", hdr)
        assert m and int(m.group(1)) == v.value // 10000, (m.group(1) if m else None, v.value)


def test_xcd_balanced_tile_runs_cover_every_tile_once():
    """The block -> tile maps of sgpt_amd/csrc/gemm.hip restated (gemm256d_kernel with GemmArgs.balanced, gemm_kernel at <= 512
    tiles): block / tile index t runs on XCD t % 8 and takes entry t // 8 of that XCD's contiguous run of the major-axis-first
    tile list.  Every tile exactly once, runs equal to within one tile, for the grids the launchers compute."""
    def run_tile(R, t):
        c0, rem = R >> 3, R & 7
        xcd, l = t & 7, t >> 3
        if l >= c0 + (1 if xcd < rem else 0):
            return None
        return xcd * c0 + min(xcd, rem) + l

    for AT in list(range(1, 40)) + [73, 74, 85, 128, 219]:
        for BT in (1, 2, 3, 4, 6, 9, 12):
            R = AT * BT
            if R > 1024:
                continue
            tiles_pad = 8 * ((R + 7) // 8)
            # gemm_kernel: one block per index; gemm256d_kernel: block b of `grid` walks b, b + grid, ... (grid a multiple of 8)
            for grid in (tiles_pad, min(tiles_pad, 256)):
                seen, per_xcd = [], [0] * 8
                for b in range(grid):
                    t = b
                    while t < tiles_pad:
                        assert t % 8 == b % 8                      # a block never leaves its XCD
                        g = run_tile(R, t)
                        if g is not None:
                            seen.append(g)
                            per_xcd[t % 8] += 1
                        t += grid
                assert sorted(seen) == list(range(R)), (AT, BT, grid)
                assert max(per_xcd) - min(per_xcd) <= 1
                # the B-tiles of one A-tile are neighbours in the list, so they share an XCD except where a run ends
                at_of = lambda g: g // BT  # noqa: E731
                assert all(at_of(a) <= at_of(b2) for a, b2 in zip(sorted(seen), sorted(seen)[1:]))
