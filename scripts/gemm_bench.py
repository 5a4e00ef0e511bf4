#!/usr/bin/env python3
"""GEMM micro-benchmark over the encoder's five launch shapes (SGPT-125M, 1024 x 128 tokens): the two 256x256 kernels
(variant 0 = 16x16x32 MFMA, 1 = 32x32x16 MFMA) interleaved in ONE process, several rounds, median and best
(guide rule 24: perf deltas come from within-probe interleaved rounds).  Operands are pseudo-random (never zeros)."""
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from sgpt_amd import get_context  # noqa: E402

ctx = get_context("cuda:0")
M = int(os.environ.get("M", 131072))
ROUNDS = int(os.environ.get("ROUNDS", 5))
DT = {"bf16": 1, "f16": 3}
dts = [d for d in os.environ.get("DTYPES", "f16,bf16").split(",") if d]
variants = [int(v) for v in os.environ.get("VARIANTS", "0").split(",")]
skews = [int(v) for v in os.environ.get("SKEWS", "0").split(",")]
shapes = [("qk    store", 0, True, M, 1536, 768), ("v     vt   ", 4, True, M, 768, 768), ("oproj resid", 2, False, M, 768, 768),
          ("fc1   gelu ", 1, True, M, 3072, 768), ("fc2   resid", 2, False, M, 768, 3072), ("kloop none ", 5, True, M, 3072, 768)]
if os.environ.get("SGPT_GEMM128") or (M // 256) * 3 * 2 <= 256:
    shapes = [sh for sh in shapes if sh[1] != 5]      # the bare k-loop probe exists only in the 256x256 kernels
if "--big" in sys.argv:
    shapes += [("1.3b fc1   ", 1, True, 65536, 8192, 2048), ("1.3b fc2   ", 2, False, 65536, 2048, 8192)]
res = {}
for rnd in range(ROUNDS):
    for dt in dts:
        for v in variants:
          for sk in skews:
            if hasattr(ctx.lib, "sgpt_exp_set_gemm_w"):        # experiment build (SGPT_EXPERIMENTS=1): the A/B knobs exist
                ctx.lib.sgpt_exp_set_gemm_w(v & 1)
                ctx.lib.sgpt_exp_set_gemm_skew(sk)
            elif v or sk:
                raise SystemExit("VARIANTS / SKEWS other than 0 need the experiment build (SGPT_EXPERIMENTS=1 python -m sgpt_amd.build)")
            for name, epi, o16, m, n, k in shapes:
                ms = C.c_float(0)
                ctx._chk(ctx.lib.sgpt_bench_gemm(ctx.handle, DT[dt], epi, DT[dt] if o16 else 0, m, n, k, 10, C.byref(ms)), "bench_gemm")
                res.setdefault((dt, v, sk, name), []).append(ms.value)
print(f"M = {M}, {ROUNDS} interleaved rounds x 10 launches; us per launch median (best) -> TFLOP/s at the median")
for dt in dts:
    for v in variants:
      for sk in skews:
        tot_ms = tot_fl = 0.0
        for name, epi, o16, m, n, k in shapes:
            t = res[(dt, v, sk, name)]
            med, best = statistics.median(t), min(t)
            fl = 2.0 * m * n * k
            if epi != 5 and not name.startswith("1.3b"):
                tot_ms += med; tot_fl += fl
            print(f"{dt} mfma{'32' if v & 1 else '16'} skew {sk:6d} {name} N={n} K={k}: {med * 1e3:8.1f} ({best * 1e3:8.1f}) us  {fl / med / 1e9:7.1f} TFLOP/s")
        print(f"{dt} mfma{'32' if v & 1 else '16'} skew {sk:6d} block total {tot_ms * 1e3:.1f} us -> {tot_fl / tot_ms / 1e9:.1f} TFLOP/s")
