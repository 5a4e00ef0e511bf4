#!/bin/bash
# Same-box A/B of the small-tile GEMM's prefetch depth (libraries built with -DSGPT_SMALL_PF=n): encode latency of query batches.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for rnd in 1 2; do
  for lib in ${LIBS:-libsgpt_pf1.so libsgpt_pf2.so libsgpt_hip.so libsgpt_pf4.so libsgpt_pf6.so}; do
    for nq in 16 128 1000; do
      echo -n "$lib round $rnd: "; SGPT_HIP_LIB=$R/sgpt_amd/lib/$lib NQ=$nq python scripts/small_batch_profile.py 2>&1 | grep "per encode"
    done
  done
done
