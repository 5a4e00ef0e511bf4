#!/bin/bash
# Same-box A/B of two builds of the library (same ABI): alternate processes, two rounds each.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for rnd in 1 2; do
  for lib in ${LIBS:-libsgpt_hip_nofold.so libsgpt_hip.so}; do
    echo "=== $lib (round $rnd)"
    SGPT_HIP_LIB=$R/sgpt_amd/lib/$lib VARIANTS=0 DTYPES=${DTYPES:-f16} ROUNDS=${ROUNDS:-3} python scripts/gemm_bench.py 2>&1 | grep -E "TFLOP" | sed 's/skew      0 //'
  done
done
