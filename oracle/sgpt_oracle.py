"""CPU oracle for the SGPT bi-encoder retrieval hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``sgpt_amd/`` may import this file: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
use it, and there only as the checker / the reported CPU baseline.

It is a plain numpy (fp32) restatement of the reference algorithm, function by
function.  Citations are relative to /root/reference; ``HF:`` points into the
un-vendored dependency that holds the transformer arithmetic
(huggingface ``transformers``; reference pin ``transformers>=4.6,<5``, setup.py:21
of the vendored sentence-transformers; 5.15.0 installed in this image), whose
eager GPT-Neo path is restated here from its published algorithm.

Parity status: PINNED.  ``tests/golden/make_golden.py`` (run in the build
container, where /root/reference and HF transformers exist) checks every function
below against the real reference code (HF ``GPTNeoModel`` eager fp32, the
reference's own ``Pooling.py`` / ``util.py`` / ``exact_search.py`` loaded from
their files) and commits the input/output vectors under ``tests/golden``;
``tests/test_oracle.py`` re-checks the oracle against those vectors and against
the reference's own offline tests (``tests/test_util.py:9-53,69-76``).
"""
from __future__ import annotations

import heapq
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

F32 = np.float32
FINFO_MIN = np.finfo(np.float32).min


# ----------------------------------------------------------------------------
# Model description + seeded synthetic weights (shared with the GPU tests so the
# CPU and GPU sides read identical bytes)
# ----------------------------------------------------------------------------
class NeoConfig:
    """Subset of HF GPTNeoConfig that the forward pass reads
    (HF:gpt_neo/configuration_gpt_neo.py)."""

    def __init__(self, vocab_size=50257, max_position_embeddings=2048, hidden_size=768,
                 num_layers=12, num_heads=12, intermediate_size=None, window_size=256,
                 attention_layers=None, layer_norm_epsilon=1e-5):
        self.vocab_size = vocab_size
        self.max_position_embeddings = max_position_embeddings
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.num_heads = num_heads
        self.intermediate_size = intermediate_size or 4 * hidden_size
        self.window_size = window_size
        # GPT-Neo default: alternating global/local (HF: attention_types=[[["global","local"], L/2]])
        self.attention_layers = attention_layers or [
            "global" if i % 2 == 0 else "local" for i in range(num_layers)]
        self.layer_norm_epsilon = layer_norm_epsilon

    @property
    def head_dim(self):
        return self.hidden_size // self.num_heads


SGPT_125M = dict(vocab_size=50257, max_position_embeddings=2048, hidden_size=768,
                 num_layers=12, num_heads=12, window_size=256)
SGPT_1_3B = dict(vocab_size=50257, max_position_embeddings=2048, hidden_size=2048,
                 num_layers=24, num_heads=16, window_size=256)
SGPT_2_7B = dict(vocab_size=50257, max_position_embeddings=2048, hidden_size=2560,
                 num_layers=32, num_heads=20, window_size=256)


def bf16_round(a: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> bf16 -> fp32 (the de-quantised weights the
    oracle uses when the GPU runs bf16 MFMA operands)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    rounded = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return rounded.astype(np.uint32).view(np.float32).reshape(a.shape)


def synth_weights(cfg: NeoConfig, seed: int = 0, std: float = 0.02,
                  bf16_linear: bool = False) -> Dict[str, np.ndarray]:
    """Seeded random-init weights under HF GPT-Neo state-dict names (no
    checkpoints exist offline).  Biases/LN get non-trivial values
    so a dropped bias cannot hide.  ``bf16_linear`` rounds the six matmul weights
    of every block to bf16 (what the bf16 MFMA path holds)."""
    rng = np.random.default_rng(seed)
    d, ffn = cfg.hidden_size, cfg.intermediate_size

    def nrm(*shape, s=std):
        return (rng.standard_normal(shape, dtype=np.float32) * F32(s)).astype(F32)

    w = {"wte.weight": nrm(cfg.vocab_size, d), "wpe.weight": nrm(cfg.max_position_embeddings, d, s=std / 2)}
    q = bf16_round if bf16_linear else (lambda x: x)
    for i in range(cfg.num_layers):
        p = f"h.{i}."
        w[p + "ln_1.weight"] = (1.0 + nrm(d, s=0.1)).astype(F32)
        w[p + "ln_1.bias"] = nrm(d, s=0.05)
        w[p + "attn.attention.q_proj.weight"] = q(nrm(d, d))
        w[p + "attn.attention.k_proj.weight"] = q(nrm(d, d))
        w[p + "attn.attention.v_proj.weight"] = q(nrm(d, d))
        w[p + "attn.attention.out_proj.weight"] = q(nrm(d, d))
        w[p + "attn.attention.out_proj.bias"] = nrm(d, s=0.02)
        w[p + "ln_2.weight"] = (1.0 + nrm(d, s=0.1)).astype(F32)
        w[p + "ln_2.bias"] = nrm(d, s=0.05)
        w[p + "mlp.c_fc.weight"] = q(nrm(ffn, d))
        w[p + "mlp.c_fc.bias"] = nrm(ffn, s=0.02)
        w[p + "mlp.c_proj.weight"] = q(nrm(d, ffn))
        w[p + "mlp.c_proj.bias"] = nrm(d, s=0.02)
    w["ln_f.weight"] = (1.0 + nrm(d, s=0.1)).astype(F32)
    w["ln_f.bias"] = nrm(d, s=0.05)
    return w


# ----------------------------------------------------------------------------
# a2: GPT-Neo forward  (HF:gpt_neo/modeling_gpt_neo.py)
# ----------------------------------------------------------------------------
def layer_norm(x, g, b, eps):
    """nn.LayerNorm(eps) over the last dim (HF:gpt_neo:317-319,385)."""
    x = x.astype(F32)
    mean = x.mean(axis=-1, keepdims=True, dtype=F32)
    xc = x - mean
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=F32)
    return (xc / np.sqrt(var + F32(eps))) * g + b


def gelu_new(u):
    """transformers/activations.py NewGELUActivation:
    0.5u(1+tanh(sqrt(2/pi)(u+0.044715u^3)))."""
    u = u.astype(F32)
    c = F32(np.sqrt(2.0 / np.pi))
    return F32(0.5) * u * (F32(1.0) + np.tanh(c * (u + F32(0.044715) * u * u * u)))


def _softmax_lastdim(a):
    m = a.max(axis=-1, keepdims=True)
    e = np.exp(a - m)
    return e / e.sum(axis=-1, keepdims=True, dtype=F32)


def gptneo_forward(w: Dict[str, np.ndarray], cfg: NeoConfig, input_ids, attention_mask=None,
                   output_hidden_states: bool = False, position_offset=None):
    """GPTNeoModel.forward (HF:gpt_neo:398-505), eager attention (105-130), fp32.

    input_ids int[B,S]; attention_mask {0,1}[B,S] (None = all ones).
    Returns last_hidden_state fp32[B,S,d] (post ln_f) and, if asked, the tuple of
    L+1 hidden states (entry i = input to block i; last = post-ln_f, HF:gpt_neo:478,492-497).
    Positions are the absolute padded index arange(S) (HF:gpt_neo:451) -- NOT mask-aware.
    """
    ids = np.asarray(input_ids)
    B, S = ids.shape
    d, H, dh = cfg.hidden_size, cfg.num_heads, cfg.head_dim
    if attention_mask is None:
        attention_mask = np.ones((B, S), dtype=np.int64)
    am = np.asarray(attention_mask)
    pos = np.arange(S)
    x = w["wte.weight"][ids] + w["wpe.weight"][pos][None, :, :]          # :444,462-463
    x = x.astype(F32)

    # causal (+ sliding window) "bias" buffers (HF:gpt_neo:56-66)
    ii = np.arange(S)[:, None]
    jj = np.arange(S)[None, :]
    causal = jj <= ii
    local = causal & (jj > ii - cfg.window_size)                       # tril xor tril(-window)
    # 4-D additive mask: 0 where key may be attended, finfo.min where padded
    # (create_causal_mask(...) for the eager path combines causal & padding; the
    # causal part is re-applied by torch.where below, :113-119)
    pad_add = np.where(am[:, None, None, :] != 0, F32(0), F32(FINFO_MIN)).astype(F32)

    hs = []
    for i in range(cfg.num_layers):
        if output_hidden_states:
            hs.append(x)
        p = f"h.{i}."
        a = layer_norm(x, w[p + "ln_1.weight"], w[p + "ln_1.bias"], cfg.layer_norm_epsilon)
        q = a @ w[p + "attn.attention.q_proj.weight"].T                 # no bias :84-86
        k = a @ w[p + "attn.attention.k_proj.weight"].T
        v = a @ w[p + "attn.attention.v_proj.weight"].T
        q = q.reshape(B, S, H, dh).transpose(0, 2, 1, 3)
        k = k.reshape(B, S, H, dh).transpose(0, 2, 1, 3)
        v = v.reshape(B, S, H, dh).transpose(0, 2, 1, 3)
        sc = np.matmul(q, k.transpose(0, 1, 3, 2)).astype(F32)          # UNSCALED :110
        bias = local if cfg.attention_layers[i] == "local" else causal
        sc = np.where(bias[None, None], sc, F32(FINFO_MIN))             # :113-117
        with np.errstate(over="ignore"):
            sc = sc + pad_add                                           # :119-120
        pr = _softmax_lastdim(sc)                                       # :122
        ctx = np.matmul(pr, v)                                          # :126
        ctx = ctx.transpose(0, 2, 1, 3).reshape(B, S, d)
        ao = ctx @ w[p + "attn.attention.out_proj.weight"].T + w[p + "attn.attention.out_proj.bias"]
        x = ao + x                                                      # :342
        a2 = layer_norm(x, w[p + "ln_2.weight"], w[p + "ln_2.bias"], cfg.layer_norm_epsilon)
        h = gelu_new(a2 @ w[p + "mlp.c_fc.weight"].T + w[p + "mlp.c_fc.bias"])
        m = h @ w[p + "mlp.c_proj.weight"].T + w[p + "mlp.c_proj.bias"]
        x = (x + m).astype(F32)                                         # :348
    x = layer_norm(x, w["ln_f.weight"], w["ln_f.bias"], cfg.layer_norm_epsilon)  # :492
    if output_hidden_states:
        hs.append(x)
        return x, tuple(hs)
    return x


# ----------------------------------------------------------------------------
# a2 (second family): GPT-J forward  (HF:gptj/modeling_gptj.py) -- SGPT-5.8B
# ----------------------------------------------------------------------------
class GPTJConfig:
    """Subset of HF GPTJConfig the forward reads (HF:gptj/configuration_gptj.py)."""
    model_type = "gptj"

    def __init__(self, vocab_size=50400, n_positions=2048, n_embd=4096, n_layer=28, n_head=16, rotary_dim=64,
                 n_inner=None, layer_norm_epsilon=1e-5):
        self.vocab_size = vocab_size
        self.max_position_embeddings = n_positions
        self.hidden_size = n_embd
        self.num_layers = n_layer
        self.num_heads = n_head
        self.rotary_dim = rotary_dim
        self.intermediate_size = n_inner or 4 * n_embd
        self.layer_norm_epsilon = layer_norm_epsilon

    @property
    def head_dim(self):
        return self.hidden_size // self.num_heads


SGPT_5_8B = dict(vocab_size=50400, n_positions=2048, n_embd=4096, n_layer=28, n_head=16, rotary_dim=64)


def synth_weights_gptj(cfg: GPTJConfig, seed: int = 0, std: float = 0.02, bf16_linear: bool = False):
    """Seeded random-init weights under HF GPT-J state-dict names."""
    rng = np.random.default_rng(seed)
    d, ffn = cfg.hidden_size, cfg.intermediate_size

    def nrm(*shape, s=std):
        return (rng.standard_normal(shape, dtype=np.float32) * F32(s)).astype(F32)

    q = bf16_round if bf16_linear else (lambda x: x)
    w = {"wte.weight": nrm(cfg.vocab_size, d, s=std * 2)}
    for i in range(cfg.num_layers):
        p = f"h.{i}."
        w[p + "ln_1.weight"] = (1.0 + nrm(d, s=0.1)).astype(F32)
        w[p + "ln_1.bias"] = nrm(d, s=0.05)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            w[p + f"attn.{n}.weight"] = q(nrm(d, d))
        w[p + "mlp.fc_in.weight"] = q(nrm(ffn, d))
        w[p + "mlp.fc_in.bias"] = nrm(ffn, s=0.02)
        w[p + "mlp.fc_out.weight"] = q(nrm(d, ffn))
        w[p + "mlp.fc_out.bias"] = nrm(d, s=0.02)
    w["ln_f.weight"] = (1.0 + nrm(d, s=0.1)).astype(F32)
    w["ln_f.bias"] = nrm(d, s=0.05)
    return w


def rotary_tables(max_pos: int, dim: int):
    """create_sinusoidal_positions (HF:gptj:47-50): sin/cos [max_pos, dim/2], float32 arithmetic throughout."""
    inv_freq = (F32(1.0) / (F32(10000.0) ** (np.arange(0, dim, 2).astype(F32) / F32(dim)))).astype(F32)
    ang = (np.arange(max_pos).astype(F32)[:, None] * inv_freq[None, :]).astype(F32)
    return np.sin(ang).astype(F32), np.cos(ang).astype(F32)


def _rotate_every_two(x):
    """HF:gptj:57-61: (x0,x1,x2,x3,..) -> (-x1,x0,-x3,x2,..)."""
    out = np.empty_like(x)
    out[..., 0::2] = -x[..., 1::2]
    out[..., 1::2] = x[..., 0::2]
    return out


def gptj_forward(w, cfg: GPTJConfig, input_ids, attention_mask=None, output_hidden_states=False):
    """GPTJModel.forward (HF:gptj:433-560), eager attention (136-159), fp32.  Parallel block
    x = attn(ln_1(x)) + mlp(ln_1(x)) + x (:400-411); rotary on the first rotary_dim dims of every head with
    interleaved pairs (:64-67,190-210); scores / sqrt(head_dim) (:148); no q/k/v/out biases (:98-101);
    positions = arange(S) (:500-503)."""
    ids = np.asarray(input_ids)
    B, S = ids.shape
    d, H, dh, rd = cfg.hidden_size, cfg.num_heads, cfg.head_dim, cfg.rotary_dim
    am = np.ones((B, S), dtype=np.int64) if attention_mask is None else np.asarray(attention_mask)
    x = w["wte.weight"][ids].astype(F32)                                    # :484 (no position embedding)
    sin, cos = rotary_tables(cfg.max_position_embeddings, rd)
    sin = np.repeat(sin[:S], 2, axis=1)[None, :, None, :]                   # repeat_interleave(2) :65-66
    cos = np.repeat(cos[:S], 2, axis=1)[None, :, None, :]
    ii, jj = np.arange(S)[:, None], np.arange(S)[None, :]
    causal = np.where(jj <= ii, F32(0), F32(FINFO_MIN)).astype(F32)[None, None]
    pad_add = np.where(am[:, None, None, :] != 0, F32(0), F32(FINFO_MIN)).astype(F32)
    with np.errstate(over="ignore"):
        mask = np.maximum(causal + pad_add, F32(FINFO_MIN))               # 4-D additive mask (0 / finfo.min)
    hs = []
    for i in range(cfg.num_layers):
        if output_hidden_states:
            hs.append(x)
        p = f"h.{i}."
        a = layer_norm(x, w[p + "ln_1.weight"], w[p + "ln_1.bias"], cfg.layer_norm_epsilon)
        q = (a @ w[p + "attn.q_proj.weight"].T).reshape(B, S, H, dh)
        k = (a @ w[p + "attn.k_proj.weight"].T).reshape(B, S, H, dh)
        v = (a @ w[p + "attn.v_proj.weight"].T).reshape(B, S, H, dh)
        for t in (q, k):                                                     # :197-210
            rot = t[..., :rd]
            t[..., :rd] = rot * cos + _rotate_every_two(rot) * sin
        q, k, v = (t.transpose(0, 2, 1, 3) for t in (q, k, v))
        sc = (np.matmul(q, k.transpose(0, 1, 3, 2)) / F32(np.sqrt(dh))).astype(F32) + mask   # :147-151
        ctx = np.matmul(_softmax_lastdim(sc), v).transpose(0, 2, 1, 3).reshape(B, S, d)
        attn = ctx @ w[p + "attn.out_proj.weight"].T
        mlp = gelu_new(a @ w[p + "mlp.fc_in.weight"].T + w[p + "mlp.fc_in.bias"]) @ w[p + "mlp.fc_out.weight"].T \
            + w[p + "mlp.fc_out.bias"]
        x = (attn + mlp + x).astype(F32)                                     # :411
    x = layer_norm(x, w["ln_f.weight"], w["ln_f.bias"], cfg.layer_norm_epsilon)
    if output_hidden_states:
        hs.append(x)
        return x, tuple(hs)
    return x


# ----------------------------------------------------------------------------
# a2 (third family): BLOOM forward  (HF:bloom/modeling_bloom.py) -- sgpt-bloom-7b1-msmarco
# ----------------------------------------------------------------------------
class BloomConfig:
    """Subset of HF BloomConfig the forward reads (HF:bloom/configuration_bloom.py)."""
    model_type = "bloom"

    def __init__(self, vocab_size=250880, hidden_size=4096, n_layer=30, n_head=32, layer_norm_epsilon=1e-5):
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size
        self.num_layers = n_layer
        self.num_heads = n_head
        self.intermediate_size = 4 * hidden_size
        self.layer_norm_epsilon = layer_norm_epsilon
        self.max_position_embeddings = 2048          # ALiBi: no learned positions; packing limit only

    @property
    def head_dim(self):
        return self.hidden_size // self.num_heads


SGPT_BLOOM_7B1 = dict(vocab_size=250880, hidden_size=4096, n_layer=30, n_head=32)


def synth_weights_bloom(cfg: BloomConfig, seed: int = 0, std: float = 0.02, bf16_linear: bool = False):
    """Seeded random-init weights under HF BLOOM state-dict names (fused, head-interleaved QKV)."""
    rng = np.random.default_rng(seed)
    d, ffn = cfg.hidden_size, cfg.intermediate_size

    def nrm(*shape, s=std):
        return (rng.standard_normal(shape, dtype=np.float32) * F32(s)).astype(F32)

    q = bf16_round if bf16_linear else (lambda x: x)
    w = {"word_embeddings.weight": nrm(cfg.vocab_size, d, s=std * 2),
         "word_embeddings_layernorm.weight": (1.0 + nrm(d, s=0.1)).astype(F32),
         "word_embeddings_layernorm.bias": nrm(d, s=0.05)}
    for i in range(cfg.num_layers):
        p = f"h.{i}."
        for ln in ("input_layernorm", "post_attention_layernorm"):
            w[p + ln + ".weight"] = (1.0 + nrm(d, s=0.1)).astype(F32)
            w[p + ln + ".bias"] = nrm(d, s=0.05)
        w[p + "self_attention.query_key_value.weight"] = q(nrm(3 * d, d))
        w[p + "self_attention.query_key_value.bias"] = nrm(3 * d, s=0.02)
        w[p + "self_attention.dense.weight"] = q(nrm(d, d))
        w[p + "self_attention.dense.bias"] = nrm(d, s=0.02)
        w[p + "mlp.dense_h_to_4h.weight"] = q(nrm(ffn, d))
        w[p + "mlp.dense_h_to_4h.bias"] = nrm(ffn, s=0.02)
        w[p + "mlp.dense_4h_to_h.weight"] = q(nrm(d, ffn))
        w[p + "mlp.dense_4h_to_h.bias"] = nrm(d, s=0.02)
    w["ln_f.weight"] = (1.0 + nrm(d, s=0.1)).astype(F32)
    w["ln_f.bias"] = nrm(d, s=0.05)
    return w


def synth_weights_streams(cfg, seed: int = 0, std: float = 0.02, threads: Optional[int] = None) -> Dict[str, np.ndarray]:
    """Seeded random-init weights for the FULL-SIZE shapes (SGPT-1.3B / 5.8B / bloom-7b1: 1.3-6 G parameters), any of the
    three families, under the same HF state-dict names and scales as synth_weights / synth_weights_gptj /
    synth_weights_bloom -- but every tensor draws from its OWN generator stream ``default_rng([seed, tensor_index])``
    (numpy SeedSequence: stable across platforms), so the tensors are independent of generation order and can be
    produced by a thread pool (Generator.standard_normal releases the GIL): 6 G parameters take ~20 s on 8 cores
    instead of minutes on one stream.  Used by tests/golden/make_golden_large.py and tests/test_gpu_parity_large.py,
    which must read identical bytes."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    d, ffn, L = cfg.hidden_size, cfg.intermediate_size, cfg.num_layers
    spec: List[Tuple[str, tuple, float, float]] = []          # (name, shape, scale, offset)

    def add(name, shape, s, off=0.0):
        spec.append((name, tuple(shape), float(s), float(off)))

    def add_ln(base):
        add(base + ".weight", (d,), 0.1, 1.0)
        add(base + ".bias", (d,), 0.05)
    if isinstance(cfg, BloomConfig):
        add("word_embeddings.weight", (cfg.vocab_size, d), std * 2)
        add_ln("word_embeddings_layernorm")
        for i in range(L):
            p = f"h.{i}."
            add_ln(p + "input_layernorm")
            add_ln(p + "post_attention_layernorm")
            add(p + "self_attention.query_key_value.weight", (3 * d, d), std)
            add(p + "self_attention.query_key_value.bias", (3 * d,), 0.02)
            add(p + "self_attention.dense.weight", (d, d), std)
            add(p + "self_attention.dense.bias", (d,), 0.02)
            add(p + "mlp.dense_h_to_4h.weight", (ffn, d), std)
            add(p + "mlp.dense_h_to_4h.bias", (ffn,), 0.02)
            add(p + "mlp.dense_4h_to_h.weight", (d, ffn), std)
            add(p + "mlp.dense_4h_to_h.bias", (d,), 0.02)
    elif isinstance(cfg, GPTJConfig):
        add("wte.weight", (cfg.vocab_size, d), std * 2)
        for i in range(L):
            p = f"h.{i}."
            add_ln(p + "ln_1")
            for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                add(p + f"attn.{n}.weight", (d, d), std)
            add(p + "mlp.fc_in.weight", (ffn, d), std)
            add(p + "mlp.fc_in.bias", (ffn,), 0.02)
            add(p + "mlp.fc_out.weight", (d, ffn), std)
            add(p + "mlp.fc_out.bias", (d,), 0.02)
    else:
        add("wte.weight", (cfg.vocab_size, d), std)
        add("wpe.weight", (cfg.max_position_embeddings, d), std / 2)
        for i in range(L):
            p = f"h.{i}."
            add_ln(p + "ln_1")
            for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                add(p + f"attn.attention.{n}.weight", (d, d), std)
            add(p + "attn.attention.out_proj.bias", (d,), 0.02)
            add_ln(p + "ln_2")
            add(p + "mlp.c_fc.weight", (ffn, d), std)
            add(p + "mlp.c_fc.bias", (ffn,), 0.02)
            add(p + "mlp.c_proj.weight", (d, ffn), std)
            add(p + "mlp.c_proj.bias", (d,), 0.02)
    add_ln("ln_f")

    def gen(item):
        i, (name, shape, s, off) = item
        a = np.random.default_rng([seed, i]).standard_normal(shape, dtype=np.float32)
        a *= F32(s)
        if off:
            a += F32(off)
        return name, a
    n_thr = threads or min(32, os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=n_thr) as ex:
        return dict(ex.map(gen, enumerate(spec)))


def engineer_outliers(w: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """In place: outlier structure of the kind real GPT-Neo / GPT-2 checkpoints carry, on seeded GPT-Neo weights
    (12+ layers, d >= 768) -- the 16-bit paths must survive it inside the parity bar:
      * two embedding channels x 100 (outlier feature dimensions from the first layer on);
      * block 2: two output channels of mlp.c_proj x 300 (weights and bias): "massive activations" of a few hundred in
        the residual stream against a typical magnitude of ~1, carried through all later LayerNorms;
      * blocks 1 and 6: two hidden units of mlp.c_fc x 100 (GELU outputs ~100 against ~0.3);
      * block 5: ln_1 gamma x 10 on the massive channels;
      * block 3: two hidden units of mlp.c_fc x 1e5 whose GELU output (~1e5) is beyond the IEEE-half range (65504), read
        by mlp.c_proj through columns x 0.01 -- the case the f16 range shifts exist for.
    Deterministic; tests/golden/make_golden_large.py applies it before the HF run, the GPU test before loading."""
    w["wte.weight"][:, [7, 300]] *= F32(100.0)
    w["h.2.mlp.c_proj.weight"][[138, 447], :] *= F32(300.0)
    w["h.2.mlp.c_proj.bias"][[138, 447]] *= F32(300.0)
    for blk in (1, 6):
        w[f"h.{blk}.mlp.c_fc.weight"][[11, 1234], :] *= F32(100.0)
        w[f"h.{blk}.mlp.c_fc.bias"][[11, 1234]] *= F32(100.0)
    w["h.5.ln_1.weight"][[138, 447]] *= F32(10.0)
    w["h.3.mlp.c_fc.weight"][[5, 77], :] *= F32(1e5)
    w["h.3.mlp.c_proj.weight"][:, [5, 77]] *= F32(0.01)
    return w


def alibi_slopes(n_head: int) -> np.ndarray:
    """build_alibi_tensor slopes (HF:bloom:62-79), float32."""
    import math
    cp2 = 2 ** math.floor(math.log2(n_head))
    base = F32(2 ** (-(2 ** -(math.log2(cp2) - 3))))
    slopes = np.power(base, np.arange(1, 1 + cp2, dtype=np.int32).astype(F32)).astype(F32)
    if cp2 != n_head:
        extra_base = F32(2 ** (-(2 ** -(math.log2(2 * cp2) - 3))))
        nrem = min(cp2, n_head - cp2)
        extra = np.power(extra_base, np.arange(1, 1 + 2 * nrem, 2, dtype=np.int32).astype(F32)).astype(F32)
        slopes = np.concatenate([slopes, extra])
    return slopes.astype(F32)


def bloom_gelu(x):
    """bloom_gelu_forward (HF:bloom:112-120) == gelu_new's tanh form."""
    x = x.astype(F32)
    return x * F32(0.5) * (F32(1.0) + np.tanh(F32(0.79788456) * x * (F32(1.0) + F32(0.044715) * x * x)))


def bloom_forward(w, cfg: BloomConfig, input_ids, attention_mask=None, output_hidden_states=False):
    """BloomModel.forward (HF:bloom:441-556): embedding LayerNorm (:499), ALiBi bias
    slope_h * ((cumsum(mask)-1)*mask) (:45-89) added to q.k^T/sqrt(dh) (baddbmm :270-275), fused QKV viewed
    [.., n_head, 3, head_dim] (:214), softmax fp32 (:283), sequential block with
    apply_residual_connection_post_layernorm=False (:356-392)."""
    ids = np.asarray(input_ids)
    B, S = ids.shape
    d, H, dh = cfg.hidden_size, cfg.num_heads, cfg.head_dim
    am = np.ones((B, S), dtype=np.int64) if attention_mask is None else np.asarray(attention_mask)
    x = layer_norm(w["word_embeddings.weight"][ids].astype(F32), w["word_embeddings_layernorm.weight"],
                   w["word_embeddings_layernorm.bias"], cfg.layer_norm_epsilon)
    ar = ((np.cumsum(am, axis=-1) - 1) * am).astype(F32)                       # [B,S]
    alibi = (alibi_slopes(H)[None, :, None] * ar[:, None, :])[:, :, None, :]   # [B,H,1,S]
    ii, jj = np.arange(S)[:, None], np.arange(S)[None, :]
    causal = np.where(jj <= ii, F32(0), F32(FINFO_MIN)).astype(F32)[None, None]
    pad_add = np.where(am[:, None, None, :] != 0, F32(0), F32(FINFO_MIN)).astype(F32)
    with np.errstate(over="ignore"):
        mask = np.maximum(causal + pad_add, F32(FINFO_MIN))
    hs = []
    for i in range(cfg.num_layers):
        if output_hidden_states:
            hs.append(x)
        p = f"h.{i}."
        ln = layer_norm(x, w[p + "input_layernorm.weight"], w[p + "input_layernorm.bias"], cfg.layer_norm_epsilon)
        fused = (ln @ w[p + "self_attention.query_key_value.weight"].T + w[p + "self_attention.query_key_value.bias"])
        fused = fused.reshape(B, S, H, 3, dh)
        q, k, v = (fused[..., j, :].transpose(0, 2, 1, 3) for j in range(3))
        sc = (alibi + np.matmul(q, k.transpose(0, 1, 3, 2)) * F32(1.0 / np.sqrt(dh))).astype(F32) + mask
        ctx = np.matmul(_softmax_lastdim(sc), v).transpose(0, 2, 1, 3).reshape(B, S, d)
        x = (ctx @ w[p + "self_attention.dense.weight"].T + w[p + "self_attention.dense.bias"] + x).astype(F32)
        ln2 = layer_norm(x, w[p + "post_attention_layernorm.weight"], w[p + "post_attention_layernorm.bias"],
                         cfg.layer_norm_epsilon)
        h = bloom_gelu(ln2 @ w[p + "mlp.dense_h_to_4h.weight"].T + w[p + "mlp.dense_h_to_4h.bias"])
        x = (h @ w[p + "mlp.dense_4h_to_h.weight"].T + w[p + "mlp.dense_4h_to_h.bias"] + x).astype(F32)
    x = layer_norm(x, w["ln_f.weight"], w["ln_f.bias"], cfg.layer_norm_epsilon)
    if output_hidden_states:
        hs.append(x)
        return x, tuple(hs)
    return x


def forward_any(w, cfg, ids, mask, output_hidden_states=False):
    if getattr(cfg, "model_type", "gpt_neo") == "gptj":
        return gptj_forward(w, cfg, ids, mask, output_hidden_states=output_hidden_states)
    if getattr(cfg, "model_type", "gpt_neo") == "bloom":
        return bloom_forward(w, cfg, ids, mask, output_hidden_states=output_hidden_states)
    return gptneo_forward(w, cfg, ids, mask, output_hidden_states=output_hidden_states)


# ----------------------------------------------------------------------------
# a4/a5: pooling
# ----------------------------------------------------------------------------
def pool(hidden, attention_mask, mode: str = "weightedmean", clamp: bool = True, position_weights=None):
    """Pooling.forward (sentence_transformers/models/Pooling.py:99-125 weightedmean/mean,
    129-164 lasttoken) and the raw-HF variants (beir_dense_retriever.py:238-242 mean,
    258-270 weightedmean, 271-282 lasttoken).

    weights w_t = t+1 over the PADDED index (Pooling.py:104-112).  ``clamp`` applies the
    1e-9 floor on the weight sum that only Pooling.py:122 has.
    lasttoken follows the raw path's ``len-1`` semantics = index of the last mask==1
    token (beir_dense_retriever.py:198,271-282), not ST's argmin bug (appendix A.8).
    learntmean: w_t = position_weights[t] over the padded index, weight sum clamped at 1e-9
    (models/WeightedMeanPooling.py:21-39; raw path useb_dense_retriever.py:253-270).
    """
    h = np.asarray(hidden, dtype=F32)
    m = np.asarray(attention_mask).astype(F32)
    B, S, d = h.shape
    if mode in ("weightedmean", "mean", "learntmean"):
        wts = m.copy()
        if mode == "weightedmean":
            wts = wts * np.arange(1, S + 1, dtype=F32)[None, :]
        if mode == "learntmean":
            wts = wts * np.asarray(position_weights, dtype=F32)[None, :S]
        num = (h * wts[:, :, None]).sum(axis=1, dtype=F32)
        den = wts.sum(axis=1, dtype=F32)[:, None]
        if clamp:
            den = np.maximum(den, F32(1e-9))
        return (num / den).astype(F32)
    if mode == "lasttoken":
        idx = np.array([int(np.nonzero(r)[0][-1]) if r.any() else 0 for r in m])
        return h[np.arange(B), idx].astype(F32)
    raise ValueError(f"unknown pooling mode {mode}")


# ----------------------------------------------------------------------------
# cross-encoder scoring (crossencoder/beir/sgptce.py:150-262): log P(continuation | context)
# ----------------------------------------------------------------------------
def lm_head(w, cfg):
    """LM head weight [V, d] (+ bias): GPT-Neo / BLOOM tie it to the input embedding (HF tie_word_embeddings),
    GPT-J carries lm_head.weight / lm_head.bias."""
    if "lm_head.weight" in w:
        return w["lm_head.weight"], w.get("lm_head.bias")
    return (w["word_embeddings.weight"] if "word_embeddings.weight" in w else w["wte.weight"]), None


def log_softmax(x):
    x = np.asarray(x, dtype=F32)
    m = x.max(axis=-1, keepdims=True)
    return (x - m - np.log(np.exp(x - m).sum(axis=-1, keepdims=True, dtype=F32))).astype(F32)


def ce_model_input(context_enc, continuation_enc, max_length, instruction_len=0):
    """sgptce.py:204-211: instruction kept, the rest truncated from the left, final token dropped."""
    rest = (list(context_enc[instruction_len:]) + list(continuation_enc))[-(max_length + 1 - instruction_len):]
    return (list(context_enc[:instruction_len]) + rest)[:-1]


def loglikelihood_tokens(w, cfg, requests, max_length, instruction_len=0):
    """`_loglikelihood_tokens` restated: per request the sum over the continuation tokens of
    log_softmax(logits)[position - 1][token] (sgptce.py:233-259), batch size 1 as the reference's default."""
    hw, hb = lm_head(w, cfg)
    out = []
    for _, ctx_enc, cont_enc in requests:
        inp = ce_model_input(ctx_enc, cont_enc, max_length, instruction_len)
        ids = np.asarray([inp], dtype=np.int64)
        last = forward_any(w, cfg, ids, np.ones_like(ids))
        logits = last[0] @ np.asarray(hw, dtype=F32).T
        if hb is not None:
            logits = logits + np.asarray(hb, dtype=F32)
        lsm = log_softmax(logits)
        n, c = len(inp), len(cont_enc)
        rows = lsm[n - c:n]
        out.append(float(rows[np.arange(c), np.asarray(cont_enc)].astype(np.float64).sum()))
    return out


# ----------------------------------------------------------------------------
# fp8 (OCP e4m3fn) weight storage with a power-of-two scale per output channel
# ("fp8-e4m3fn weights, per-output-channel fp32 scales; the oracle uses the
# de-quantised weights in fp32").  Pinned against torch.float8_e4m3fn (tests/golden/make_golden.py).
# ----------------------------------------------------------------------------
def fp8_scales(w):
    """Per row: the smallest power of two s with max|w_row| / s <= 448 (e4m3fn max); 1 for an all-zero row."""
    amax = np.abs(np.asarray(w, dtype=F32)).max(axis=1)
    m, e = np.frexp(amax.astype(np.float64))          # amax = m * 2^e, 0.5 <= m < 1
    # amax / 2^k <= 448 = 0.875 * 2^9  ->  k = e - 9 if m <= 0.875 else e - 8
    k = np.where(m <= 0.875, e - 9, e - 8)
    k = np.clip(k, -126, 127)
    return np.where(amax > 0, np.ldexp(1.0, k), 1.0).astype(F32)


def fp8_e4m3fn_encode(x):
    """fp32 -> e4m3fn code (uint8), round-to-nearest-even, saturating at +-448."""
    x = np.asarray(x, dtype=F32)
    sign = (x.view(np.uint32) >> 24).astype(np.uint32) & 0x80
    a = np.abs(x)
    u = a.view(np.uint32).astype(np.uint64)
    u = u + 0x7FFFF + ((u >> 20) & 1)
    normal = np.minimum((u >> 20).astype(np.int64) - (120 << 3), 0x7E)
    sub = np.rint(a.astype(np.float64) * 512.0).astype(np.int64)
    code = np.where(a < F32(0.015625), sub, normal)
    code = np.where(a >= F32(464.0), 0x7E, code)
    code = np.where(np.isnan(a), 0x7F, code)
    return (sign | code.astype(np.uint32)).astype(np.uint8)


def fp8_e4m3fn_decode(code):
    c = np.asarray(code, dtype=np.uint8).astype(np.int64)
    e, m = (c >> 3) & 15, c & 7
    a = np.where(e > 0, np.ldexp((8 + m).astype(np.float64), e - 10), m * 2.0 ** -9)
    return np.where(c & 0x80, -a, a).astype(F32)


def fp8_quantize_rows(w):
    """[rows, cols] fp32 -> (codes uint8, scale fp32[rows]); w / scale is exact (power of two)."""
    w = np.asarray(w, dtype=F32)
    s = fp8_scales(w)
    return fp8_e4m3fn_encode(w / s[:, None]), s


def fp8_dequantize_rows(codes, scale):
    return (fp8_e4m3fn_decode(codes) * np.asarray(scale, dtype=F32)[:, None]).astype(F32)


def fp8_roundtrip_weights(w: dict) -> dict:
    """The weights a dtype='fp8' model computes with: every 2-D matmul weight of the blocks goes through
    quantise -> de-quantise (rows = output channels); embeddings, LayerNorms and biases stay fp32."""
    out = {}
    for k, v in w.items():
        is_block_matmul = k.startswith("h.") and k.endswith(".weight") and np.asarray(v).ndim == 2
        out[k] = fp8_dequantize_rows(*fp8_quantize_rows(v)) if is_block_matmul else v
    return out


def pool_layers(all_hidden: Sequence[np.ndarray], attention_mask, mode: str):
    """meanmean / lasttokenmean over ALL L+1 hidden states
    (beir_dense_retriever.py:243-257, 284-301)."""
    m = np.asarray(attention_mask).astype(F32)
    hs = np.stack([np.asarray(h, dtype=F32) for h in all_hidden])       # [L+1,B,S,d]
    if mode == "meanmean":
        num = (hs * m[None, :, :, None]).sum(axis=2).sum(axis=0)
        den = (m.sum(axis=1) * hs.shape[0])[:, None]
        return (num / den).astype(F32)
    if mode == "lasttokenmean":
        B = m.shape[0]
        idx = np.array([int(np.nonzero(r)[0][-1]) for r in m])
        return hs[:, np.arange(B), idx].mean(axis=0).astype(F32)
    raise ValueError(mode)


# ----------------------------------------------------------------------------
# a7: scoring
# ----------------------------------------------------------------------------
def _as2d(a):
    a = np.asarray(a, dtype=F32)
    return a[None, :] if a.ndim == 1 else a


def normalize(a, eps: float = 1e-12):
    """torch.nn.functional.normalize(p=2, dim=1): x / max(||x||, eps) (util.py:41-42,
    util.normalize_embeddings :66-70)."""
    a = _as2d(a)
    n = np.sqrt((a * a).sum(axis=1, keepdims=True, dtype=F32))
    return (a / np.maximum(n, F32(eps))).astype(F32)


def cos_sim(a, b):
    """util.cos_sim (sentence_transformers/util.py:24-43) == beir.util.cos_sim."""
    return (normalize(a) @ normalize(b).T).astype(F32)


def dot_score(a, b):
    """util.dot_score (util.py:46-63)."""
    return (_as2d(a) @ _as2d(b).T).astype(F32)


def pairwise_cos_sim(a, b):
    """util.pairwise_cos_sim (util.py:86-97): row-wise cosine."""
    return (normalize(a) * normalize(b)).sum(axis=1, dtype=F32)


# ----------------------------------------------------------------------------
# a8: top-k + chunk merge
# ----------------------------------------------------------------------------
def topk_rows(scores, k):
    """torch.topk(scores, k, dim=1, largest=True, sorted=False) as a set per row:
    returns (values[nq,k], idx[nq,k]) sorted descending for determinism (the
    reference's order is unspecified, exact_search.py:102-108)."""
    s = np.asarray(scores, dtype=F32)
    k = min(k, s.shape[1])
    idx = np.argsort(-s, axis=1, kind="stable")[:, :k]
    return np.take_along_axis(s, idx, axis=1), idx


def exact_search(query_emb, query_ids: List[str], corpus_emb_chunks, corpus_ids: List[str],
                 top_k: int, score_function: str = "cos_sim",
                 chunk_size: int = 50000, backend: str = "numpy") -> Dict[str, Dict[str, float]]:
    """DenseRetrievalExactSearch.search after the encode calls
    (biencoder/beir/custommodels/exact_search.py:80-132): per corpus chunk score
    (:96-98), NaN -> -1 (:99), topk(min(k+1,n)) (:102-108), drop corpus_id == query_id
    (:118), after the first chunk keep heapq.nlargest(min(k+1,len)) (:121-132).

    ``corpus_emb_chunks`` is a callable chunk_index -> fp32[n_chunk,d] or a full array.
    backend "torch": the score / NaN / top-k lines run on the torch CPU primitives the reference itself calls
    (F.normalize + torch.mm, util.py:41-43; torch.topk, exact_search.py:102-108) -- the timing stand-in for the
    reference's CPU search leg in bench.py (scripts/cpu_search_ref_vs_port.py: within a few % of the reference's own
    file on the same inputs; the numpy lines are ~6x slower, an argsort instead of a partial selection).
    """
    if score_function not in ("cos_sim", "dot"):
        raise ValueError(
            "score function: {} must be either (cos_sim) for cosine similarity or (dot) for dot product".format(
                score_function))
    fn = cos_sim if score_function == "cos_sim" else dot_score
    results: Dict[str, Dict[str, float]] = {qid: {} for qid in query_ids}
    n = len(corpus_ids)
    if backend == "torch":
        import torch
        tq = torch.from_numpy(np.ascontiguousarray(query_emb, dtype=np.float32))
    for batch_num, start in enumerate(range(0, n, chunk_size)):
        end = min(start + chunk_size, n)
        sub = corpus_emb_chunks(batch_num) if callable(corpus_emb_chunks) else corpus_emb_chunks[start:end]
        if backend == "torch":
            ts = torch.from_numpy(np.ascontiguousarray(sub, dtype=np.float32))
            if score_function == "cos_sim":
                tsc = torch.mm(torch.nn.functional.normalize(tq, p=2, dim=1), torch.nn.functional.normalize(ts, p=2, dim=1).transpose(0, 1))
            else:
                tsc = torch.mm(tq, ts.transpose(0, 1))
            tsc[torch.isnan(tsc)] = -1
            tv, ti = torch.topk(tsc, min(top_k + 1, tsc.shape[1]), dim=1, largest=True, sorted=False)
            vals, idx = tv.numpy(), ti.numpy()
        else:
            sc = fn(query_emb, sub)
            sc[np.isnan(sc)] = -1
            vals, idx = topk_rows(sc, min(top_k + 1, sc.shape[1]))
        for qi, qid in enumerate(query_ids):
            for sub_id, score in zip(idx[qi].tolist(), vals[qi].tolist()):
                cid = corpus_ids[start + sub_id]
                if cid != qid:
                    results[qid][cid] = score
            if batch_num > 0:
                keep = heapq.nlargest(min(top_k + 1, len(results[qid])), results[qid], key=results[qid].get)
                results[qid] = {k_: results[qid][k_] for k_ in keep}
    return results


def semantic_search(query_emb, corpus_emb, query_chunk_size=100, corpus_chunk_size=500000,
                    top_k=10, score_function=cos_sim):
    """util.semantic_search (util.py:197-258): chunked top-k with final sort."""
    q = _as2d(query_emb)
    c = _as2d(corpus_emb)
    out = [[] for _ in range(len(q))]
    for qs in range(0, len(q), query_chunk_size):
        for cs in range(0, len(c), corpus_chunk_size):
            sc = score_function(q[qs:qs + query_chunk_size], c[cs:cs + corpus_chunk_size])
            vals, idx = topk_rows(sc, min(top_k, sc.shape[1]))
            for r in range(sc.shape[0]):
                for j, v in zip(idx[r].tolist(), vals[r].tolist()):
                    out[qs + r].append({"corpus_id": cs + j, "score": v})
    for i in range(len(out)):
        out[i] = sorted(out[i], key=lambda x: x["score"], reverse=True)[:top_k]
    return out


# ----------------------------------------------------------------------------
# a1: the integer part of tokenisation that does not need a vocabulary
# ----------------------------------------------------------------------------
SPECB_QUE_BOS, SPECB_QUE_EOS, SPECB_DOC_BOS, SPECB_DOC_EOS = 58, 60, 90, 92   # GPT-2 BPE ids of [ ] { }
GPT2_PAD = 50256                                                               # pad = eos


def specb_wrap(ids: Sequence[int], is_query: bool, max_token_len: Optional[int] = None) -> List[int]:
    """beir_dense_retriever.py:183-191 / Transformer.py:131-153 restated on ids:
    truncate to max_token_len (already reduced by 2, :134-136), then bracket."""
    ids = list(ids)
    if max_token_len is not None:
        ids = ids[:max_token_len]
    if is_query:
        return [SPECB_QUE_BOS] + ids + [SPECB_QUE_EOS]
    return [SPECB_DOC_BOS] + ids + [SPECB_DOC_EOS]


def pad_batch(seqs: Sequence[Sequence[int]], pad_id: int = GPT2_PAD, side: str = "right"):
    """tokenizer.pad(padding=True) (beir_dense_retriever.py:201): pad to batch max."""
    S = max(len(s) for s in seqs)
    ids = np.full((len(seqs), S), pad_id, dtype=np.int64)
    mask = np.zeros((len(seqs), S), dtype=np.int64)
    for i, s in enumerate(seqs):
        if side == "right":
            ids[i, :len(s)] = s
            mask[i, :len(s)] = 1
        else:
            ids[i, S - len(s):] = s
            mask[i, S - len(s):] = 1
    return ids, mask


def encode(w, cfg, seqs: Sequence[Sequence[int]], mode="weightedmean", batch_size=32,
           normalize_embeddings=False, layer_idx=-1, pad_side="right"):
    """embed_batcher loop (beir_dense_retriever.py:225-314) on pre-tokenised ids:
    batch -> pad -> forward -> select layer -> pool."""
    out = []
    for i in range(0, len(seqs), batch_size):
        ids, mask = pad_batch(seqs[i:i + batch_size], pad_id=min(GPT2_PAD, cfg.vocab_size - 1), side=pad_side)
        last, hs = forward_any(w, cfg, ids, mask, output_hidden_states=True)
        h = hs[layer_idx]
        if mode in ("meanmean", "lasttokenmean"):
            e = pool_layers(hs, mask, mode)
        else:
            e = pool(h, mask, mode, clamp=True)
        out.append(normalize(e) if normalize_embeddings else e)
    return np.concatenate(out, axis=0)
