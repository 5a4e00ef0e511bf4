#!/usr/bin/env python3
"""Scoring-only micro-benchmark: nq queries x N-document 16-bit shard (DT=f16 | bf16), cosine top-(k+1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgpt_amd import get_context
ctx = get_context("cuda:0")
nq, N, d, k = int(os.environ.get("NQ", 1000)), int(os.environ.get("N", 1000000)), 768, int(os.environ.get("K", 11))
DT = torch.bfloat16 if os.environ.get("DT", "f16") == "bf16" else torch.float16
g = torch.Generator(device="cuda").manual_seed(0)
base = torch.randn(1, d, device="cuda", generator=g) * 3          # anisotropic: shared dominant direction
cf = base + torch.randn(N, d, device="cuda", generator=g)
if os.environ.get("DRIFT"):      # score distribution jumps at this fraction of the corpus (documents behind it score higher)
    cf[int(float(os.environ["DRIFT"]) * N):] += 2.0 * base
if os.environ.get("DUP"):        # bench.py's 1 M shard: perturbed copies of a block of DUP documents
    D = int(os.environ["DUP"])
    blk = torch.nn.functional.normalize(cf[:D], dim=1).to(DT).float()
    for s0 in range(0, N, D):
        e0 = min(N, s0 + D)
        cf[s0:e0] = blk[: e0 - s0] + 0.02 * torch.randn((e0 - s0, d), device="cuda", generator=g)
c = torch.nn.functional.normalize(cf, dim=1).to(DT)
del cf
q = torch.nn.functional.normalize(base + torch.randn(nq, d, device="cuda", generator=g), dim=1).to(DT)
for _ in range(2):
    ctx.score_topk(q, c, k, dtype=DT)
torch.cuda.synchronize()
t = time.perf_counter()
reps = int(os.environ.get("REPS", 20))
for _ in range(reps):
    ctx.score_topk(q, c, k, dtype=DT)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / reps
print(f"nq={nq} N={N} k={k}: {dt*1e3:.2f} ms per pass -> {nq/dt:,.0f} queries/s; corpus stream {N*d*2/dt/1e12:.2f} TB/s; {2*nq*N*d/dt/1e12:.0f} TFLOP/s")
