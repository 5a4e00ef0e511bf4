"""Shared helpers for the GPU parity tests (test infrastructure: may import the oracle)."""
import ast
import os

import numpy as np

from oracle import sgpt_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(tag):
    fx = np.load(os.path.join(GOLDEN, f"{tag}.npz"))
    cfg_kw = ast.literal_eval(str(fx["cfg"]))
    lens = fx["seq_lens"].tolist()
    ids, mask = fx["ids"].astype(np.int64), fx["mask"].astype(np.int64)
    side = str(fx["pad_side"])
    S = ids.shape[1]
    seqs = [ids[i, :n].tolist() if side == "right" else ids[i, S - n:].tolist() for i, n in enumerate(lens)]
    pad_left = [0] * len(lens) if side == "right" else [S - n for n in lens]
    return fx, cfg_kw, seqs, pad_left, ids, mask


_models = {}


def oracle_cfg_weights(cfg_kw, seed, std, bf16_linear=False):
    if "n_embd" in cfg_kw:      # GPT-J field names
        cfg = O.GPTJConfig(**cfg_kw)
        return cfg, O.synth_weights_gptj(cfg, seed=seed, std=std, bf16_linear=bf16_linear)
    if "n_layer" in cfg_kw:     # BLOOM field names
        cfg = O.BloomConfig(**cfg_kw)
        return cfg, O.synth_weights_bloom(cfg, seed=seed, std=std, bf16_linear=bf16_linear)
    cfg = O.NeoConfig(**cfg_kw)
    return cfg, O.synth_weights(cfg, seed=seed, std=std, bf16_linear=bf16_linear)


def build_model(cfg_kw, seed, std, dtype, precise_qk=False, precision="plain"):
    """SGPTModel on cuda:0 with the oracle's seeded synthetic weights (cached per test session).  precision='plain' unless a
    test asks otherwise: the kernel-level tests pin the 16-bit path itself, not what the 'auto' probe would choose."""
    from sgpt_amd import SGPTConfig, SGPTModel
    key = (repr(sorted(cfg_kw.items())), seed, std, dtype, precise_qk, precision)
    if key not in _models:
        _, w = oracle_cfg_weights(cfg_kw, seed, std)
        if "n_embd" in cfg_kw:
            scfg = SGPTConfig.from_hf_dict(dict(cfg_kw, model_type="gptj"))
        elif "n_layer" in cfg_kw:
            scfg = SGPTConfig.from_hf_dict(dict(cfg_kw, model_type="bloom"))
        else:
            scfg = SGPTConfig(**cfg_kw)
        kw = dict(precision=precision) if dtype in ("f16", "bf16") else {}
        _models[key] = SGPTModel(scfg, w, device="cuda:0", dtype=dtype, precise_qk=precise_qk, **kw)
    return _models[key]


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


def row_cos(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
