#!/usr/bin/env python3
"""Query-sized forward (csrc/qgemm.hip) against the bulk kernels on the same sequences: identical bits?  and the latency of
both paths at nq = 1 / 16 / 128 (sgpt_ctx_set_tile_policy(1) keeps the bulk kernels)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sgpt_amd import SGPTConfig, SGPTModel, synthetic_weights
dev = torch.device("cuda", 0)
cfg = SGPTConfig()
dtype = os.environ.get("DTYPE", "f16")
model = SGPTModel(cfg, synthetic_weights(cfg, seed=1), device=dev, dtype=dtype)
rng = np.random.default_rng(7)
for nq in [int(v) for v in os.environ.get("NQS", "1,2,3,4,16,21,32,128").split(",")]:
    qs = [rng.integers(0, 50256, size=int(rng.integers(4, 33))).tolist() for _ in range(nq)]
    pb = model.pack(qs)
    res = {}
    for pol in (0, 2, 1):
        if pol == 1 and pb.T_pad > 1024:
            res[1] = res[2]
            continue
        model.ctx.set_tile_policy(pol)
        out = torch.empty((nq, 768), device=dev)
        for _ in range(5): model.encode_packed(pb, normalize=True, out=out)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(50): model.encode_packed(pb, normalize=True, out=out)
        torch.cuda.synchronize()
        res[pol] = ((time.perf_counter() - t) / 50 * 1e3, out.clone())
        print(f"   policy {pol}: finite {bool(torch.isfinite(out).all())}, range flags {model.range_flags(reset=True)}")
    model.ctx.set_tile_policy(0)
    # the same sequences inside a bulk batch (own rows of a 4096-row call): the bits a corpus-sized call gives them
    big = qs + [rng.integers(0, 50256, size=128).tolist() for _ in range(40)]
    try:
        ref = model.encode_ids(big, normalize=True)[:nq]
    except Exception as e:
        print("   bulk reference failed:", str(e)[:80]); ref = res[True][1]
    a, b = res[0][1], res[1][1]
    print(f"nq={nq} T_pad={pb.T_pad}: query path {res[0][0]:.3f} ms, small-tile kernels of the bulk path {res[2][0]:.3f} ms, 256x256 kernels "
          f"{res[1][0]:.3f} ms; identical bits: {bool(torch.equal(a, b) and torch.equal(a, res[2][1]))} (max diff {float((a - b).abs().max()):.2e}); vs rows of a 5k-row call: "
          f"{bool(torch.equal(a, ref))} (max diff {float((a - ref).abs().max()):.2e}); finite {bool(torch.isfinite(a).all())}", flush=True)
