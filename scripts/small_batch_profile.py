#!/usr/bin/env python3
"""A 16-query encode (337 tokens, T_pad 512) repeated: the per-kernel durations say whether small batches are bound by
launch latency or by the kernels' own serial latency (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sgpt_amd import SGPTConfig, SGPTModel, synthetic_weights
dev = torch.device("cuda", 0)
cfg = SGPTConfig()
model = SGPTModel(cfg, synthetic_weights(cfg, seed=1), device=dev, dtype="f16")
model.ctx.set_low_latency(os.environ.get("LL", "0") == "1")     # LL=1: the opt-in k-group mode (per ctx)
rng = np.random.default_rng(7)
nq = int(os.environ.get("NQ", "16"))
qs = [rng.integers(0, 50256, size=int(rng.integers(4, 33))).tolist() for _ in range(nq)]
pb = model.pack(qs)
out = torch.empty((nq, 768), device=dev)
for _ in range(5): model.encode_packed(pb, normalize=True, out=out)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(50): model.encode_packed(pb, normalize=True, out=out)
torch.cuda.synchronize()
print(f"LL={os.environ.get('LL', '0')} nq={nq} T_pad={pb.T_pad}: {(time.perf_counter() - t) / 50 * 1e3:.3f} ms per encode")
