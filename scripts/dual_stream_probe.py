#!/usr/bin/env python3
"""Probe ("18 % of a step is un-fused, HBM-bound, serial work"): can the HBM-bound kernels of one
encode call (LayerNorm, attention, embed, pool) hide under the MFMA-bound GEMMs of ANOTHER call?  Two contexts (each with
its own activation workspace) and two copies of the SGPT-125M weights on one GPU, two HIP streams; the same 1024 x 128
token calls issued (a) all on one stream, (b) alternating between the two streams.  The persistent 256x256 GEMM occupies
every CU's LDS, so two GEMMs never co-reside; LayerNorm needs no LDS and may.
Round 4: CAP=n caps the persistent GEMM grid at n workgroups per launch (Context.set_gemm_cu_cap), so
that 256 - n CUs stay free for the other pipeline's HBM-bound kernels; CALL=512 halves the call so the two pipelines are
half a call out of phase."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgpt_amd import SGPTConfig, SGPTModel, synthetic_weights  # noqa: E402
from sgpt_amd.runtime import Context  # noqa: E402

cfg = SGPTConfig(vocab_size=50257, max_position_embeddings=2048, hidden_size=768, num_layers=12, num_heads=12, window_size=256)
w = synthetic_weights(cfg, seed=1)
dev = torch.device("cuda", 0)
ctxs = [Context(0), Context(0)]
models = [SGPTModel(cfg, w, device=dev, dtype=os.environ.get("DT", "f16"), ctx=c) for c in ctxs]
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
CAP = int(os.environ.get("CAP", 0))
for c in ctxs:
    c.set_gemm_cu_cap(CAP)
CALL = int(os.environ.get("CALL", 1024))
rng = np.random.default_rng(0)
ncalls = int(os.environ.get("CALLS", 16))
packed = [[m.pack(rng.integers(0, 50256, size=(CALL, 128), dtype=np.int64)) for _ in range(ncalls)] for m in models]
outs = [torch.empty((CALL, 768), dtype=torch.float32, device=dev) for _ in range(2)]
torch.cuda.synchronize()


def run(two_streams):
    for i in range(ncalls):
        k = i % 2 if two_streams else 0
        with torch.cuda.stream(streams[k]):
            models[k].encode_packed(packed[k][i], normalize=True, out=outs[k])
    torch.cuda.synchronize()


for mode in (False, True):
    run(mode)
for rnd in range(3):
    for mode in (False, True):
        t = time.perf_counter()
        run(mode)
        dt = time.perf_counter() - t
        print(f"cap {CAP or 256}: round {rnd}: {'two streams (alternating calls)' if mode else 'one stream                     '}: "
              f"{dt / ncalls * 1e3:.3f} ms per {CALL}-sentence call -> {ncalls * CALL / dt:,.0f} sentences/s", flush=True)
