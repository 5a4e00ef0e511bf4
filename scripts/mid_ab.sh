#!/bin/bash
# A/B of library builds on mid-sized encode calls (LIBS="a.so b.so"): 1000 / 128 queries of 4..32 tokens, 1000 / 300 sentences of 8..64
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
OUT=gpurun_out/mid_ab.txt; : > $OUT
for rnd in 1 2; do
 for lib in ${LIBS:-libsgpt_hip.so}; do
  for cfg in "1000 4 32" "128 4 32" "3000 4 32" "1000 8 64" "300 8 64" "32 8 64"; do
    set -- $cfg
    echo -n "$lib r$rnd len $2..$3 " >> $OUT
    SGPT_HIP_LIB=$R/sgpt_amd/lib/$lib NQ=$1 LMIN=$2 LMAX=$3 python scripts/mid_batch_profile.py 2>&1 | tail -1 >> $OUT
  done
 done
done
cat $OUT
