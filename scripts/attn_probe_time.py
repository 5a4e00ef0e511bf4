import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from sgpt_amd import SGPTConfig, SGPTModel, synthetic_weights
cfg = SGPTConfig()
m = SGPTModel(cfg, synthetic_weights(cfg, seed=1), device="cuda:0", dtype="f16", max_tokens_per_call=1 << 18)
S = int(os.environ.get("S", 512)); B = 131072 // S
ids = np.random.default_rng(0).integers(0, 50256, size=(B, S))
pb = m.pack(ids)
out = torch.empty((B, 768), device="cuda")
for _ in range(2): m.encode_packed(pb, normalize=True, out=out)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): m.encode_packed(pb, normalize=True, out=out)
torch.cuda.synchronize()
print(f"S={S}: {(time.perf_counter() - t) / 5 * 1e3:.2f} ms per 131072-token encode call")
