#!/usr/bin/env python3
"""The per-rank shard pass of bench.py's projected_8gpu block on bench.py's own data (SGPT-125M-shape encoder outputs,
perturbed copies), stand-alone: `rocprofv3 --kernel-trace --stats -- python scripts/shard_profile.py` for the kernel table."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sgpt_amd import SGPTConfig, SGPTModel, get_context, synthetic_weights
dev = torch.device("cuda", 0)
ctx = get_context(dev)
cfg = SGPTConfig()
model = SGPTModel(cfg, synthetic_weights(cfg, seed=1), device=dev, dtype="f16", precision="plain")
rng = np.random.default_rng(1000)
NB = int(os.environ.get("BLOCK", 98304))
N = int(os.environ.get("N", 125000))
nq, k1 = int(os.environ.get("NQ", 1000)), 11
rows = []
for _ in range(NB // 1024):
    rows.append(model.encode_ids(rng.integers(0, 50256, size=(1024, 128), dtype=np.int64), normalize=True))
blk = torch.cat(rows).to(torch.float16).float()
qrng = np.random.default_rng(7)
queries = [qrng.integers(0, 50256, size=int(qrng.integers(4, 33))).tolist() for _ in range(nq)]
q = ctx._operand(model.encode_ids(queries, normalize=True), torch.float16)
gen = torch.Generator(device=dev).manual_seed(99)
big = torch.empty((N, cfg.hidden_size), dtype=torch.float16, device=dev)
for s0 in range(0, N, NB):
    e0 = min(N, s0 + NB)
    big[s0:e0] = torch.nn.functional.normalize(blk[: e0 - s0] + 0.02 * torch.randn((e0 - s0, cfg.hidden_size), generator=gen, device=dev), dim=1).to(torch.float16)
sc = ctx.scores(q[:64].contiguous(), big[:4096].contiguous(), dtype=torch.float16)
print(f"score range of the data: min {float(sc.min()):.4f} max {float(sc.max()):.4f} std {float(sc.std()):.5f}; query-row crest {ctx.row_crest(q):.1f}")
for _ in range(2):
    ctx.score_topk(q, big, k1, idx_base=3 * N, dtype=torch.float16)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    v, i, _ = ctx.score_topk(q, big, k1, idx_base=3 * N, dtype=torch.float16)
torch.cuda.synchronize()
print(f"nq={nq} N={N}: {(time.perf_counter() - t) / 5 * 1e3:.3f} ms per pass")
ref = torch.topk(q.float() @ big.float().T, k1, dim=1)
print("top-k ids equal to a plain product:", bool((i - 3 * N == ref.indices).float().mean() > 0.99))
