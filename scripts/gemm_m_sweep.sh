#!/bin/bash
# Per-shape GEMM time over token counts between a query batch and the bulk corpus call: default dispatch (variant 0),
# 256x256 tiles forced (variant 2), register-staged 128x128 / 64x64 kernel forced (SGPT_GEMM128=1).
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for m in ${MS:-512 3328 8192 25088 49152}; do
  echo "=== M=$m default / forced 256"; M=$m VARIANTS=0,2 DTYPES=f16 ROUNDS=3 python scripts/gemm_bench.py 2>&1 | grep -E "TFLOP" | grep -v kloop
  echo "=== M=$m forced register-staged"; SGPT_GEMM128=1 M=$m VARIANTS=0 DTYPES=f16 ROUNDS=3 python scripts/gemm_bench.py 2>&1 | grep -E "TFLOP" | grep -v kloop
done
