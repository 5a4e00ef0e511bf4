#!/usr/bin/env python3
"""Query-sized encode latency of the larger SGPT shapes (bench.py's MODELS, random-init weights on the device): the query- / mid-sized
kernels (tile policy 0) against the bulk path's small tiles (policy 2), 1 / 16 queries of 4..32 tokens.  MODELS="1.3b 5.8b" DT=bf16"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from sgpt_amd import SGPTConfig, SGPTModel
dev = torch.device("cuda", 0)
dt = os.environ.get("DT", "f16")
for name in os.environ.get("MODELS", "1.3b 5.8b").split():
    cfg = SGPTConfig.from_hf_dict(bench.MODELS[name]) if "model_type" in bench.MODELS[name] else SGPTConfig(**bench.MODELS[name])
    w = bench.device_random_weights(cfg, dev, seed=1)
    model = SGPTModel(cfg, w, device=dev, dtype=dt, precision="plain", precise_qk=False)
    del w
    rng = np.random.default_rng(7)
    for nq in (1, 16):
        qs = [rng.integers(0, 50000, size=int(rng.integers(4, 33))).tolist() for _ in range(nq)]
        pb = model.pack(qs)
        res = {}
        for pol in (0, 2):
            model.ctx.set_tile_policy(pol)
            out = torch.empty((nq, cfg.hidden_size), device=dev)
            for _ in range(3): model.encode_packed(pb, normalize=True, out=out)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(20): model.encode_packed(pb, normalize=True, out=out)
            torch.cuda.synchronize()
            res[pol] = ((time.perf_counter() - t) / 20 * 1e3, out.clone())
        model.ctx.set_tile_policy(0)
        wbytes = sum(p.numel() for p in []) if False else None
        print(f"{name} {dt} nq={nq} T_pad={pb.T_pad}: query kernels {res[0][0]:.3f} ms, bulk path small tiles {res[2][0]:.3f} ms; identical bits {bool(torch.equal(res[0][1], res[2][1]))}", flush=True)
    model.close(); del model; torch.cuda.empty_cache()
