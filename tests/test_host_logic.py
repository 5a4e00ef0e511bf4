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
    assert lib.sgpt_abi_version() == 1
    # struct layout mirrors the header
    assert ctypes.sizeof(_lib.ModelDesc) == 64 and ctypes.sizeof(_lib.TensorView) == 24


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
    from sgpt_amd.model import pack_host, ALIGN
    seqs = [[5, 6, 7], list(range(1, 18)), [9] * 16, [1]]
    h = pack_host(seqs, pad_left=[2, 0, 0, 7])
    assert h["B"] == 4 and h["T_pad"] % 256 == 0 and h["max_alloc"] == 32 and h["n_tokens"] == 37
    off = h["seq_off"]
    assert off.tolist() == [0, 16, 48, 64, 80] and all(o % ALIGN == 0 for o in off)
    for b, s in enumerate(seqs):
        assert h["ids"][off[b]: off[b] + len(s)].tolist() == s
        assert h["pos"][off[b]: off[b] + len(s)].tolist() == [h["pad_left"][b] + t for t in range(len(s))]
    assert h["ids"][3:16].tolist() == [0] * 13          # filler rows
    with pytest.raises(ValueError, match="Empty items should be cleaned prior to running"):
        pack_host([[1], []])


def test_plan_batches_covers_everything_sorted():
    from sgpt_amd.model import SGPTModel
    lens = np.random.default_rng(0).integers(1, 129, size=1000)
    fake = SGPTModel.__new__(SGPTModel)
    fake.max_tokens_per_call = 4096
    plan = SGPTModel.plan_batches(fake, lens)
    allidx = np.concatenate(plan)
    assert sorted(allidx.tolist()) == list(range(1000))
    srt = lens[allidx]
    assert (np.diff(srt) <= 0).all()                    # longest first
    for sel in plan:
        assert ((lens[sel] + 15) // 16 * 16).sum() <= 4096


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
