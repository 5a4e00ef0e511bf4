#!/usr/bin/env python3
"""A 1000-query encode (18 k tokens, T_pad 21 504) repeated: per-kernel durations of a mid-sized call (run under
rocprofv3 --kernel-trace --stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sgpt_amd import SGPTConfig, SGPTModel, synthetic_weights
dev = torch.device("cuda", 0)
cfg = SGPTConfig()
model = SGPTModel(cfg, synthetic_weights(cfg, seed=1), device=dev, dtype="f16", max_tokens_per_call=131072)
rng = np.random.default_rng(7)
nq = int(os.environ.get("NQ", "1000"))
lo, hi = int(os.environ.get("LMIN", "4")), int(os.environ.get("LMAX", "32"))
qs = [rng.integers(0, 50256, size=int(rng.integers(lo, hi + 1))).tolist() for _ in range(nq)]
pb = model.pack(qs)
out = torch.empty((nq, 768), device=dev)
for _ in range(3): model.encode_packed(pb, normalize=True, out=out)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): model.encode_packed(pb, normalize=True, out=out)
torch.cuda.synchronize()
print(f"nq={nq} T_pad={pb.T_pad}: {(time.perf_counter() - t) / 20 * 1e3:.3f} ms per encode")
