#!/usr/bin/env python3
"""Which GEMM kernel / tile should a query-sized launch take?  For token rows M = 512 ... 16 384 and the four projection shapes of
SGPT-125M, the time of one launch under each policy (experiment build: SGPT_HIP_LIB=.../libsgpt_hip_exp.so):
  default          the shipped rule (256x256 LDS-DMA kernel from 128 tiles on, else 128x128 / 64x64 register-staged by tile count)
  t128             always the 128x128 register-staged kernel        (SGPT_GEMM128=1 SGPT_T128_MIN=0)
  t64              always the 64x64 one                              (SGPT_GEMM128=1 SGPT_T128_MIN=1000000000)
  t256             the 256x256 kernel wherever the shape allows      (tile policy 1)
Each policy runs in its own process (the switches are read once)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [("qk", 1536, 768, 0, 3), ("out", 768, 768, 2, 0), ("fc1", 3072, 768, 1, 3), ("fc2", 768, 3072, 2, 0)]
MS = [512, 1024, 1536, 2048, 3072, 4096, 6144, 8192, 12288, 16384]
if len(sys.argv) > 1 and sys.argv[1] == "worker":
    sys.path.insert(0, ROOT)
    from sgpt_amd import get_context
    ctx = get_context("cuda:0")
    if os.environ.get("FORCE256") == "1":
        ctx.set_tile_policy(True)
    for name, n, k, epi, odt in SHAPES:
        for m in MS:
            ms = C.c_float(0)
            ctx._chk(ctx.lib.sgpt_bench_gemm(ctx.handle, 3, epi, odt, m, n, k, 20, C.byref(ms)), "bench")
            print(f"{name} {m} {ms.value * 1e3:.2f}", flush=True)
    sys.exit(0)
POL = {"default": {}, "t128": {"SGPT_GEMM128": "1", "SGPT_T128_MIN": "0"}, "t64": {"SGPT_GEMM128": "1", "SGPT_T128_MIN": "1000000000"},
       "t256": {"FORCE256": "1"}}
res = {}
for pol, env in POL.items():
    out = subprocess.run([sys.executable, __file__, "worker"], env=dict(os.environ, **env), capture_output=True, text=True).stdout
    for ln in out.splitlines():
        p = ln.split()
        if len(p) == 3:
            res[(pol, p[0], int(p[1]))] = float(p[2])
print("shape      M   " + "  ".join(f"{p:>8s}" for p in POL) + "   best")
for name, *_ in SHAPES:
    for m in MS:
        row = [res.get((p, name, m), float("nan")) for p in POL]
        best = min(range(len(row)), key=lambda i: row[i])
        print(f"{name:5s} {m:6d}   " + "  ".join(f"{v:8.1f}" for v in row) + f"   {list(POL)[best]}" + ("" if best == 0 or row[0] <= 1.03 * row[best] else f"  (default +{(row[0] / row[best] - 1) * 100:.0f} %)"))
