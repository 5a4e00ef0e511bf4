#!/usr/bin/env python3
"""Encode-only loop for a kernel trace: 4096-document steps of lengths U{16..128} (LENS=var, default) or fixed 128 (LENS=fixed) or U{16..128} rounded up to multiples of 8 (LENS=var8),
SGPT-125M shape f16, the product's own call planner.  Run under `rocprofv3 --kernel-trace --stats` and compare per-kernel time
per token between the two (where the variable-length step loses against the fixed one)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgpt_amd import SGPTConfig, SGPTModel, synthetic_weights  # noqa: E402

cfg = SGPTConfig()
m = SGPTModel(cfg, synthetic_weights(cfg, seed=1), device="cuda:0", dtype="f16", precision="plain")
m.round_aware_calls = os.environ.get("EQUAL", "0") != "1"
rng = np.random.default_rng(2000)
steps = int(os.environ.get("STEPS", 6))
fixed = os.environ.get("LENS", "var") == "fixed"
packed, tokens = [], 0
for _ in range(steps):
    lens = np.full(4096, 128) if fixed else rng.integers(16, 129, size=4096)
    if os.environ.get("LENS") == "var8":            # same distribution, every length a multiple of 8: V^T tile loads 16-B aligned
        lens = (lens + 7) // 8 * 8
    docs = [rng.integers(0, 50256, size=int(n)) for n in lens]
    plan = m.plan_batches(lens.astype(np.int64))
    packed.append([m.pack([docs[i] for i in sel]) for sel in plan])
    tokens += int(lens.sum())
out = torch.empty((4096, cfg.hidden_size), device="cuda:0")
for pb in packed[0]:
    m.encode_packed(pb, mode="weightedmean", normalize=True, out=out[: pb.B])
torch.cuda.synchronize()
t = time.perf_counter()
for row in packed:
    o = 0
    for pb in row:
        m.encode_packed(pb, mode="weightedmean", normalize=True, out=out[o: o + pb.B])
        o += pb.B
torch.cuda.synchronize()
dt = time.perf_counter() - t
print(f"LENS={os.environ.get('LENS', 'var')} calls/step {[int(pb.T_pad) for pb in packed[0]]}: {steps * 4096 / dt:,.0f} sentences/s, {tokens / dt / 1e6:.3f} M tokens/s, "
      f"{dt / tokens * 1e9:.2f} ns per token")
