#!/usr/bin/env python3
"""Does a hipGraph of one shard pass (the ~12 dependent launches of sgpt_score_topk) run shorter than the eager enqueue?
nq = 1000 (and 128) against a 125 000-document f16 shard, k = 11: eager loop vs replay of a captured pass."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgpt_amd import get_context  # noqa: E402

ctx = get_context("cuda:0")
d, k = 768, 11
g = torch.Generator(device="cuda").manual_seed(0)
for N in (125000, 1000000):
    base = torch.randn(1, d, device="cuda", generator=g) * 3
    c = torch.nn.functional.normalize(base + torch.randn(N, d, device="cuda", generator=g), dim=1).half()
    for nq in (1000, 128):
        q = torch.nn.functional.normalize(base + torch.randn(nq, d, device="cuda", generator=g), dim=1).half()
        val = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        idx = torch.empty((nq, k), dtype=torch.int64, device="cuda")

        def one():
            ctx.score_topk(q, c, k, idx_base=0, run=(val, idx, 0), dtype=torch.float16)
        for _ in range(3):
            one()
        torch.cuda.synchronize()
        reps = 30
        t = time.perf_counter()
        for _ in range(reps):
            one()
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t) / reps
        want_v, want_i = val.clone(), idx.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            one()
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            one()
        graph.replay()
        torch.cuda.synchronize()
        ok = torch.equal(val, want_v) and torch.equal(idx, want_i)
        t = time.perf_counter()
        for _ in range(reps):
            graph.replay()
        torch.cuda.synchronize()
        rep = (time.perf_counter() - t) / reps
        print(f"N={N} nq={nq}: eager {eager * 1e3:.3f} ms, graph replay {rep * 1e3:.3f} ms per pass (same result: {ok})")
