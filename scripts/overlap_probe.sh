#!/bin/bash
# Experiment: the no-store k-loop variant with one store epilogue's worth of HBM writes (128 KiB per tile) issued from inside
# the k-loop (library built with -DSGPT_PROBE_OVERLAP=1) against the plain no-store k-loop and the real store-epilogue launches.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for rnd in 1 2; do
  for lib in libsgpt_hip.so libsgpt_probe.so; do
    echo "=== $lib (round $rnd)"
    SGPT_HIP_LIB=$R/sgpt_amd/lib/$lib VARIANTS=0 DTYPES=${DTYPES:-f16,bf16} ROUNDS=3 python scripts/gemm_bench.py 2>&1 | grep -E "qk|fc1|kloop"
  done
done
