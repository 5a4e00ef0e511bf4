#!/usr/bin/env python3
"""Where does the short-row select spend its time?  topk_merge (the arg-max path of topk_select_kernel) over [nq, m] candidate lists
for a few (m, k), hipEvent-timed back to back."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgpt_amd import get_context  # noqa: E402

ctx = get_context("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
for nq in (1000, 128):
    for m in (64, 251, 1024, 1891):
        val = torch.randn((nq, m), device="cuda", generator=g)
        idx = torch.stack([torch.randperm(100000, device="cuda", generator=g)[:m] for _ in range(8)]).repeat((nq + 7) // 8, 1)[:nq].contiguous()
        for k in (1, 11, 43):
            for _ in range(3):
                ctx.topk_merge(val, idx, k)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                ctx.topk_merge(val, idx, k)
            e1.record()
            torch.cuda.synchronize()
            print(f"nq={nq} m={m} k={k}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per merge (incl. ~2 output allocations of the wrapper)")
