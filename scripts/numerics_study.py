#!/usr/bin/env python3
"""CPU emulation of the 16-bit MFMA operand roundings of the HIP encoder (GPT-Neo, SGPT-125M shape):
where does the deviation from the fp32 reference enter, and which operand format keeps cosine scores
under the 1e-3 north-star bar?    Pure torch-CPU; not a product or test path.

Every GEMM operand the HIP path stores in 16 bits is rounded here at the same point (weights, LN output, q/k/v,
softmax probabilities, attention context, GELU output, normalised embeddings); accumulation stays fp32."""
import argparse
import sys
import os
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sgpt_oracle as O  # noqa: E402


def rnd(x, fmt):
    if fmt == "f32":
        return x
    return x.to(torch.bfloat16 if fmt == "bf16" else torch.float16).float()


def forward(w, cfg, ids, fm):
    """ids [B,S] long (no padding) -> fp32 [B,d] weighted-mean pooled, normalised.  fm: dict operand -> format."""
    B, S = ids.shape
    d, H = cfg.hidden_size, cfg.num_heads
    dh = d // H
    x = w["wte.weight"][ids] + w["wpe.weight"][torch.arange(S)][None]
    causal = torch.tril(torch.ones(S, S, dtype=torch.bool))
    for i in range(cfg.num_layers):
        p = f"h.{i}."
        a = rnd(torch.nn.functional.layer_norm(x, (d,), w[p + "ln_1.weight"], w[p + "ln_1.bias"], cfg.layer_norm_epsilon), fm["a"])
        q = rnd(a @ rnd(w[p + "attn.attention.q_proj.weight"], fm["w"]).T, fm["qk"]).view(B, S, H, dh).transpose(1, 2)
        k = rnd(a @ rnd(w[p + "attn.attention.k_proj.weight"], fm["w"]).T, fm["qk"]).view(B, S, H, dh).transpose(1, 2)
        v = rnd(a @ rnd(w[p + "attn.attention.v_proj.weight"], fm["w"]).T, fm["v"]).view(B, S, H, dh).transpose(1, 2)
        s = q @ k.transpose(-1, -2)
        mask = causal
        if cfg.attention_layers[i] == "local":
            mask = causal & ~torch.tril(torch.ones(S, S, dtype=torch.bool), -cfg.window_size)
        s = s.masked_fill(~mask, torch.finfo(torch.float32).min)
        # the kernel: online softmax, un-normalised p = exp(s - m) rounded to 16 bit, row sum from fp32 p
        m = s.max(-1, keepdim=True).values
        pe = torch.exp(s - m)
        l = pe.sum(-1, keepdim=True)
        ctx = (rnd(pe, fm["p"]) @ v) / l
        ctx = rnd(ctx.transpose(1, 2).reshape(B, S, d), fm["ctx"])
        x = x + ctx @ rnd(w[p + "attn.attention.out_proj.weight"], fm["w"]).T + w[p + "attn.attention.out_proj.bias"]
        a = rnd(torch.nn.functional.layer_norm(x, (d,), w[p + "ln_2.weight"], w[p + "ln_2.bias"], cfg.layer_norm_epsilon), fm["a"])
        u = a @ rnd(w[p + "mlp.c_fc.weight"], fm["w"]).T + w[p + "mlp.c_fc.bias"]
        h = rnd(0.5 * u * (1.0 + torch.tanh(0.7978845608028654 * (u + 0.044715 * u ** 3))), fm["h"])
        x = x + h @ rnd(w[p + "mlp.c_proj.weight"], fm["w"]).T + w[p + "mlp.c_proj.bias"]
    x = torch.nn.functional.layer_norm(x, (d,), w["ln_f.weight"], w["ln_f.bias"], cfg.layer_norm_epsilon)
    wt = torch.arange(1, S + 1, dtype=torch.float32)[None, :, None]
    e = (x * wt).sum(1) / wt.sum(1)
    return torch.nn.functional.normalize(e, dim=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=48)
    ap.add_argument("--nq", type=int, default=16)
    ap.add_argument("--seq", type=int, default=128)
    ap.add_argument("--std", type=float, default=0.02)
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    cfg = O.NeoConfig(**O.SGPT_125M)
    w = {k: torch.from_numpy(v) for k, v in O.synth_weights(cfg, seed=1, std=args.std).items()}
    rng = np.random.default_rng(5)
    docs = torch.from_numpy(rng.integers(0, 50256, size=(args.docs, args.seq)))
    qs = torch.from_numpy(rng.integers(0, 50256, size=(args.nq, 24)))
    ops = ["w", "a", "qk", "v", "p", "ctx", "h"]
    base = {o: "f32" for o in ops}

    def run(fm, efmt):
        with torch.no_grad():
            return rnd(forward(w, cfg, docs, fm), efmt), rnd(forward(w, cfg, qs, fm), efmt)

    t = time.time()
    d0, q0 = run(base, "f32")
    cos0 = q0 @ d0.T
    print(f"fp32 reference: {time.time() - t:.1f}s; cos range [{cos0.min():.3f}, {cos0.max():.3f}]")

    def report(name, fm, efmt):
        dd, qq = run(fm, efmt)
        cos = qq @ dd.T
        print(f"{name:44s} max|d_emb| {float((dd - d0).abs().max()):.2e}  max|d_cos| {float((cos - cos0).abs().max()):.2e}  "
              f"rms d_cos {float((cos - cos0).pow(2).mean().sqrt()):.2e}")

    report("all bf16 (round-1 mode)", {o: "bf16" for o in ops}, "bf16")
    report("all f16", {o: "f16" for o in ops}, "f16")
    for o in ops:
        fm = dict(base)
        fm[o] = "bf16"
        report(f"only {o} bf16", fm, "f32")
    report("only embeddings bf16", base, "bf16")
    report("only embeddings f16", base, "f16")
    mixed = {o: "f16" for o in ops}
    for keep in (["qk"], ["qk", "v"], ["h"], ["qk", "v", "h"], ["qk", "v", "h", "ctx"]):
        fm = dict(mixed)
        for o in keep:
            fm[o] = "bf16"
        report("f16 except bf16 for " + ",".join(keep), fm, "f16")
    fm = {o: "bf16" for o in ops}
    fm["w"] = "f16"
    report("bf16 activations, f16 weights", fm, "bf16")
    fm = {o: "f16" for o in ops}
    fm["w"] = "bf16"
    report("f16 activations, bf16 weights", fm, "f16")


if __name__ == "__main__":
    main()
