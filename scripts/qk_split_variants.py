#!/usr/bin/env python3
"""CPU emulation (the conventions of scripts/numerics_study.py, self-contained) of cheaper variants of the split-precision Q / K projection at SGPT-1.3B shape:
plain f16, the full hi+lo split that ships (three passes), activation-only / weight-only splits and the split on one of the two
projections (two passes).  Not a product or test path."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from types import SimpleNamespace
torch.set_num_threads(os.cpu_count())
# GPT-Neo 1.3B shape (HF:gpt_neo config: alternating global / local-256 layers), random-init weights of std 0.02 under HF names
cfg = SimpleNamespace(hidden_size=2048, num_layers=24, num_heads=16, window_size=256, layer_norm_epsilon=1e-5,
                      attention_layers=["global", "local"] * 12)
g = torch.Generator().manual_seed(1)
def _n(*shape): return torch.randn(*shape, generator=g) * 0.02
w = {"wte.weight": _n(50257, 2048), "wpe.weight": _n(2048, 2048), "ln_f.weight": torch.ones(2048), "ln_f.bias": torch.zeros(2048)}
for i in range(cfg.num_layers):
    p_ = f"h.{i}."
    for nm in ("q", "k", "v", "out"): w[p_ + f"attn.attention.{nm}_proj.weight"] = _n(2048, 2048)
    w[p_ + "attn.attention.out_proj.bias"] = torch.zeros(2048)
    w[p_ + "mlp.c_fc.weight"], w[p_ + "mlp.c_fc.bias"] = _n(8192, 2048), torch.zeros(8192)
    w[p_ + "mlp.c_proj.weight"], w[p_ + "mlp.c_proj.bias"] = _n(2048, 8192), torch.zeros(2048)
    for ln in ("ln_1", "ln_2"): w[p_ + ln + ".weight"], w[p_ + ln + ".bias"] = torch.ones(2048), torch.zeros(2048)
rng = np.random.default_rng(5)
docs = torch.from_numpy(rng.integers(0, 50256, size=(24, 128)))
qs = torch.from_numpy(rng.integers(0, 50256, size=(8, 24)))
def rnd(x, fmt):
    return x if fmt == "f32" else x.to(torch.float16).float()
def split(x):
    hi = rnd(x, "f16"); lo = rnd(x - hi, "f16"); return hi, lo
def forward(ids, variant):
    B, S = ids.shape; d, H = cfg.hidden_size, cfg.num_heads; dh = d // H
    x = w["wte.weight"][ids] + w["wpe.weight"][torch.arange(S)][None]
    causal = torch.tril(torch.ones(S, S, dtype=torch.bool))
    f = "f16" if variant != "fp32" else "f32"
    for i in range(cfg.num_layers):
        p = f"h.{i}."
        a32 = torch.nn.functional.layer_norm(x, (d,), w[p + "ln_1.weight"], w[p + "ln_1.bias"], cfg.layer_norm_epsilon)
        a = rnd(a32, f)
        def proj(name, mode):
            W = w[p + f"attn.attention.{name}_proj.weight"]
            if variant == "fp32": return a32 @ W.T
            Wh, Wl = split(W); ah, al = split(a32)
            if mode == "plain": return ah @ Wh.T
            if mode == "full": return ah @ Wh.T + al @ Wh.T + ah @ Wl.T
            if mode == "asplit": return ah @ Wh.T + al @ Wh.T
            if mode == "wsplit": return ah @ Wh.T + ah @ Wl.T
        qm, km = {"plain": ("plain","plain"), "full": ("full","full"), "asplit": ("asplit","asplit"), "wsplit": ("wsplit","wsplit"),
                  "konly": ("plain","full"), "qonly": ("full","plain"), "fp32": ("plain","plain")}[variant]
        q = rnd(proj("q", qm), f).view(B, S, H, dh).transpose(1, 2)
        k = rnd(proj("k", km), f).view(B, S, H, dh).transpose(1, 2)
        v = rnd(a @ rnd(w[p + "attn.attention.v_proj.weight"], f).T, f).view(B, S, H, dh).transpose(1, 2)
        s = q @ k.transpose(-1, -2)
        mask = causal
        if cfg.attention_layers[i] == "local":
            mask = causal & ~torch.tril(torch.ones(S, S, dtype=torch.bool), -cfg.window_size)
        s = s.masked_fill(~mask, torch.finfo(torch.float32).min)
        m = s.max(-1, keepdim=True).values; pe = torch.exp(s - m); l = pe.sum(-1, keepdim=True)
        ctx = (rnd(pe, f) @ v) / l
        ctx = rnd(ctx.transpose(1, 2).reshape(B, S, d), f)
        x = x + ctx @ rnd(w[p + "attn.attention.out_proj.weight"], f).T + w[p + "attn.attention.out_proj.bias"]
        a2 = rnd(torch.nn.functional.layer_norm(x, (d,), w[p + "ln_2.weight"], w[p + "ln_2.bias"], cfg.layer_norm_epsilon), f)
        u = a2 @ rnd(w[p + "mlp.c_fc.weight"], f).T + w[p + "mlp.c_fc.bias"]
        h = rnd(0.5 * u * (1.0 + torch.tanh(0.7978845608028654 * (u + 0.044715 * u ** 3))), f)
        x = x + h @ rnd(w[p + "mlp.c_proj.weight"], f).T + w[p + "mlp.c_proj.bias"]
    x = torch.nn.functional.layer_norm(x, (d,), w["ln_f.weight"], w["ln_f.bias"], cfg.layer_norm_epsilon)
    wt = torch.arange(1, S + 1, dtype=torch.float32)[None, :, None]
    e = (x * wt).sum(1) / wt.sum(1)
    return torch.nn.functional.normalize(e, dim=1)
with torch.no_grad():
    t=time.time(); d0, q0 = forward(docs, "fp32"), forward(qs, "fp32"); cos0 = q0 @ d0.T; print("fp32", time.time()-t)
    for v in ("plain", "full", "asplit", "wsplit", "konly", "qonly"):
        dd, qq = rnd(forward(docs, v), "f16"), rnd(forward(qs, v), "f16"); cos = qq @ dd.T
        print(f"{v:8s} max|d_emb| {float((dd-d0).abs().max()):.2e}  max|d_cos| {float((cos-cos0).abs().max()):.2e} rms emb {float((dd-d0).pow(2).mean().sqrt()):.2e}", flush=True)
