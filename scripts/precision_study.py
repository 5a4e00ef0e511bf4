#!/usr/bin/env python3
"""CPU emulation (torch, fp32 accumulate) of the f16 encoder with PER-(block, class) split-precision operands on a golden
fixture: which operand classes of which blocks have to enter their GEMM as hi + lo pairs for an ill-conditioned checkpoint
(tests/golden/outlier_125m.npz) to land inside the 1e-3 bar, and what statistic of the operand predicts it?
  Not a product or test path; self-contained apart from the oracle's weight generator.

Classes (the producer -> consumer pairs of sgpt_amd/csrc/api.hip): ln1 (LayerNorm-1 output -> Q/K/V projection),
ctx (attention context -> out-projection), ln2 (LayerNorm-2 output -> fc1), h (GELU output -> fc2).  A split class enters
its GEMM as a_hi.W_hi + a_lo.W_hi + a_hi.W_lo (hi = round16(v), lo = round16(v - hi)), exactly what the K' = 3K launch does.
`qk3`: q / k themselves stored as hi + lo pairs and the logits as three MFMA passes (attention kernel variant)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sgpt_oracle as O  # noqa: E402

H = torch.float16


def r16(x):
    return x.to(H).float()


def split(x):
    hi = r16(x)
    return hi, r16(x - hi)


def mm(a, W, mode):
    """a [.., K] fp32 (un-rounded), W [N, K] fp32 -> a . W^T under the operand mode of the class."""
    if mode == "f32":
        return a @ W.T
    if mode == "f16":
        return r16(a) @ r16(W).T
    ah, al = split(a)
    Wh, Wl = split(W)
    if mode == "x3":
        return (ah + al) @ Wh.T + ah @ Wl.T
    if mode == "wsplit":                     # weight-only split: a_hi . (W_hi + W_lo)
        return ah @ Wh.T + ah @ Wl.T
    if mode == "asplit":
        return (ah + al) @ Wh.T
    raise ValueError(mode)


def crest(x):
    """max over rows of max|v| / rms(v) -- and the global figure"""
    rms = x.pow(2).mean(-1).sqrt().clamp_min(1e-30)
    return float((x.abs().amax(-1) / rms).max()), float(x.abs().max() / x.pow(2).mean().sqrt())


def forward(w, cfg, ids_list, plan, stats=None, qk3=False):
    """ids_list: list of 1-D LongTensors (variable length; one sequence at a time keeps the emulation simple and exact).
    plan[(block, cls)] -> 'f16' | 'x3' | 'f32' | 'wsplit' | 'asplit' (default f16)."""
    d, Hh = cfg.hidden_size, cfg.num_heads
    dh = d // Hh
    out = []
    g = lambda i, c: plan.get((i, c), plan.get(("*", c), "f16"))
    # batch sequences of equal length together
    by_len = {}
    for n, s in enumerate(ids_list):
        by_len.setdefault(len(s), []).append(n)
    res = [None] * len(ids_list)
    for S, idxs in by_len.items():
        ids = torch.stack([ids_list[n] for n in idxs])
        B = ids.shape[0]
        x = w["wte.weight"][ids] + w["wpe.weight"][torch.arange(S)][None]
        causal = torch.tril(torch.ones(S, S, dtype=torch.bool))
        for i in range(cfg.num_layers):
            p = f"h.{i}."
            a = torch.nn.functional.layer_norm(x, (d,), w[p + "ln_1.weight"], w[p + "ln_1.bias"], cfg.layer_norm_epsilon)
            if stats is not None:
                stats.setdefault((i, "ln1"), []).append(crest(a))
            m1 = g(i, "ln1")
            mq = g(i, "ln1qk") if (i, "ln1qk") in plan or ("*", "ln1qk") in plan else m1
            q = mm(a, w[p + "attn.attention.q_proj.weight"], mq)
            k = mm(a, w[p + "attn.attention.k_proj.weight"], mq)
            v = mm(a, w[p + "attn.attention.v_proj.weight"], m1)
            mpv = g(i, "pv")                      # P.V product: 'f16' (16-bit p and v), 'x3' (hi + lo pairs), 'f32'
            if qk3 or g(i, "qk") == "x3":
                qh, ql = split(q); kh, kl = split(k)
                f = lambda t: t.view(B, S, Hh, dh).transpose(1, 2)
                s = f(qh + ql) @ f(kh).transpose(-1, -2) + f(qh) @ f(kl).transpose(-1, -2)
            elif g(i, "qk") == "f32":
                s = q.view(B, S, Hh, dh).transpose(1, 2) @ k.view(B, S, Hh, dh).transpose(1, 2).transpose(-1, -2)
            else:
                s = r16(q).view(B, S, Hh, dh).transpose(1, 2) @ r16(k).view(B, S, Hh, dh).transpose(1, 2).transpose(-1, -2)
            fv = lambda t: t.view(B, S, Hh, dh).transpose(1, 2)
            mask = causal
            if cfg.attention_layers[i] == "local":
                mask = causal & ~torch.tril(torch.ones(S, S, dtype=torch.bool), -cfg.window_size)
            s = s.masked_fill(~mask, torch.finfo(torch.float32).min)
            mx = s.max(-1, keepdim=True).values
            pe = torch.exp(s - mx)
            l = pe.sum(-1, keepdim=True)
            if mpv == "f32":
                pvp = pe @ fv(v)
            elif mpv == "x3":
                ph, pl = split(pe); vh, vl = split(v)
                pvp = (ph + pl) @ fv(vh) + ph @ fv(vl)
            else:
                pvp = r16(pe) @ fv(r16(v))
            ctx = (pvp / l).transpose(1, 2).reshape(B, S, d)
            if stats is not None:
                stats.setdefault((i, "ctx"), []).append(crest(ctx))
            x = x + mm(ctx, w[p + "attn.attention.out_proj.weight"], g(i, "ctx")) + w[p + "attn.attention.out_proj.bias"]
            a = torch.nn.functional.layer_norm(x, (d,), w[p + "ln_2.weight"], w[p + "ln_2.bias"], cfg.layer_norm_epsilon)
            if stats is not None:
                stats.setdefault((i, "ln2"), []).append(crest(a))
            u = mm(a, w[p + "mlp.c_fc.weight"], g(i, "ln2")) + w[p + "mlp.c_fc.bias"]
            h = 0.5 * u * (1.0 + torch.tanh(0.7978845608028654 * (u + 0.044715 * u ** 3)))
            if stats is not None:
                stats.setdefault((i, "h"), []).append(crest(h))
            # f16 range shift of the class: a power of two that brings max|h| under 16384 (exact)
            hm = float(h.abs().max())
            sh = 2.0 ** max(0, int(np.ceil(np.log2(hm / 16384.0)))) if hm >= 32768 else 1.0
            x = x + mm(h / sh, w[p + "mlp.c_proj.weight"], g(i, "h")) * sh + w[p + "mlp.c_proj.bias"]
        x = torch.nn.functional.layer_norm(x, (d,), w["ln_f.weight"], w["ln_f.bias"], cfg.layer_norm_epsilon)
        wt = torch.arange(1, S + 1, dtype=torch.float32)[None, :, None]
        e = (x * wt).sum(1) / wt.sum(1)
        for j, n in enumerate(idxs):
            res[n] = e[j]
    return torch.stack(res)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fixture", default="outlier_125m")
    ap.add_argument("--docs", type=int, default=48)
    ap.add_argument("--plans", default="")
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    torch.set_grad_enabled(False)
    fx = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", f"{args.fixture}.npz"))
    meta = json.loads(str(fx["meta"]))
    cfg = O.NeoConfig(**meta["cfg"])
    wn = O.synth_weights_streams(cfg, seed=meta["seed"], std=meta["std"])
    if meta.get("outliers"):
        O.engineer_outliers(wn)
    w = {k: torch.from_numpy(v) for k, v in wn.items()}
    lens = fx["lens"].astype(np.int64)
    off = np.concatenate([[0], np.cumsum(lens)])
    isq = fx["is_query"].astype(bool)
    di = np.nonzero(~isq)[0][: args.docs]
    qi = np.nonzero(isq)[0]
    pick = np.concatenate([di, qi])
    seqs = [torch.from_numpy(fx["ids"][off[i]: off[i + 1]].astype(np.int64)) for i in pick]
    ref = torch.from_numpy(fx["emb"][pick])
    refn = torch.nn.functional.normalize(ref, dim=1)
    nd = len(di)
    cos_ref = refn[nd:] @ refn[:nd].T

    def report(name, plan, **kw):
        t = time.time()
        emb32 = kw.pop("emb32", False)
        e = forward(w, cfg, seqs, plan, **kw)
        kw["emb32"] = emb32
        en = torch.nn.functional.normalize(e, dim=1)
        en16 = en if kw.pop("emb32", False) else r16(en)
        cos = en16[nd:] @ en16[:nd].T
        print(f"{name:58s} max|d emb_n| {float((en - refn).abs().max()):.2e}  max|d cos| {float((cos - cos_ref).abs().max()):.2e}  "
              f"rms {float((cos - cos_ref).pow(2).mean().sqrt()):.2e}   ({time.time() - t:.0f} s)", flush=True)

    stats = {}
    ALL = ("ln1", "ctx", "ln2", "h")
    report("fp32 everywhere (emulation vs fixture)", {("*", c): "f32" for c in ALL + ("qk", "pv")}, stats=stats, emb32=True)
    report("fp32 everywhere, f16 corpus / query rows", {("*", c): "f32" for c in ALL + ("qk", "pv")})
    report("fp32 GEMMs + logits, 16-bit P.V", {("*", c): "f32" for c in ALL + ("qk",)})
    report("every operand hi+lo (GEMMs, logits, P.V)", {("*", c): "x3" for c in ALL + ("qk", "pv")})
    report("hi+lo GEMMs + P.V, 16-bit logits", {("*", c): "x3" for c in ALL + ("pv",)})
    L = cfg.num_layers
    print("crest factors (max over rows of max|v|/rms(v); global max|v|/rms):")
    for c in ("ln1", "ctx", "ln2", "h"):
        print("  " + c + ": " + " ".join(f"{max(s[0] for s in stats[(i, c)]):.0f}/{max(s[1] for s in stats[(i, c)]):.0f}" for i in range(L)))
    report("all f16 (today's default)", {})
    for c in ("ln1", "ctx", "ln2", "h"):
        report(f"f16, class {c} split in every block", {("*", c): "x3"})
    report("f16, every class split (f16x3 GEMMs, 16-bit q/k/v/p)", {("*", c): "x3" for c in ("ln1", "ctx", "ln2", "h")})
    report("f16x3 GEMMs + q/k hi+lo logits", {("*", c): "x3" for c in ("ln1", "ctx", "ln2", "h")}, qk3=True)
    for spec in filter(None, args.plans.split(";")):
        plan = {}
        for item in spec.split(","):
            blk, c, mode = item.split(":")
            plan[("*" if blk == "*" else int(blk), c)] = mode
        report("plan " + spec, plan)


if __name__ == "__main__":
    main()
