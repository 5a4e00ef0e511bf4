#!/usr/bin/env python3
"""Slab-residency probe: does a call whose activations (~17 KB per token at SGPT-125M shape) fit the 256 MiB
Infinity Cache trade fabric bytes for the mid-size penalty?  Encodes the same 8192 sentences x 128 tokens in calls of CALL
sentences; run plain for the rate, and under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (own passes, --kernel-trace)
for the HBM bytes, summed by scripts/pmc_totals.py and divided by the sentences encoded."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sgpt_amd import SGPTConfig, SGPTModel, get_context, synthetic_weights
CALL, TOTAL = int(os.environ.get("CALL", 1024)), int(os.environ.get("TOTAL", 8192))
dev = torch.device("cuda", 0)
cfg = SGPTConfig()
model = SGPTModel(cfg, synthetic_weights(cfg, seed=1), device=dev, dtype="f16", precision="plain", max_tokens_per_call=CALL * 128)
rng = np.random.default_rng(0)
packed = [model.pack(rng.integers(0, 50256, size=(CALL, 128), dtype=np.int64)) for _ in range(TOTAL // CALL)]
out = torch.empty((CALL, cfg.hidden_size), dtype=torch.float32, device=dev)
for pb in packed[: max(1, 1024 // CALL)]:
    model.encode_packed(pb, normalize=True, out=out)
torch.cuda.synchronize()
t = time.perf_counter()
for pb in packed:
    model.encode_packed(pb, normalize=True, out=out)
torch.cuda.synchronize()
dt = time.perf_counter() - t
print(f"CALL={CALL}: {TOTAL / dt:,.0f} sentences/s ({dt / len(packed) * 1e3:.3f} ms per call); sentences_encoded_incl_warmup={TOTAL + max(1, 1024 // CALL) * CALL}")
