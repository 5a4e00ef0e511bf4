#!/usr/bin/env python3
"""Where the query side's time goes (nq queries of 4..32 tokens against a resident 1 M-document shard)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sgpt_amd import SGPTConfig, SGPTModel, get_context, synthetic_weights
dev = torch.device("cuda", 0)
ctx = get_context(dev)
cfg = SGPTConfig()
model = SGPTModel(cfg, synthetic_weights(cfg, seed=1), device=dev, dtype="f16", max_tokens_per_call=131072)
rng = np.random.default_rng(7)
c = torch.nn.functional.normalize(torch.randn(1_000_000, 768, device=dev), dim=1).to(torch.float16)
def tm(f, reps=5):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
for nq in (16, 128, 1000):
    qs = [rng.integers(0, 50256, size=int(rng.integers(4, 33))).tolist() for _ in range(nq)]
    t_pack = tm(lambda: model.pack(qs))
    pb = model.pack(qs)
    out = torch.empty((nq, 768), device=dev)
    t_enc = tm(lambda: model.encode_packed(pb, normalize=True, out=out))
    t_ids = tm(lambda: model.encode_ids(qs, normalize=True))
    q16 = ctx._operand(out, torch.float16)
    t_cvt = tm(lambda: ctx._operand(out, torch.float16))
    t_search = tm(lambda: ctx.score_topk(q16, c, 11, dtype=torch.float16))
    print(f"nq={nq}: tokens {pb.n_tokens} (T_pad {pb.T_pad}); pack+H2D {t_pack:.3f} ms, encode_packed {t_enc:.3f} ms, encode_ids (sort+pack+encode+unsort) "
          f"{t_ids:.3f} ms, to f16 {t_cvt:.3f} ms, search {t_search:.3f} ms -> {nq / (t_ids + t_cvt + t_search) * 1e3:,.0f} q/s")
