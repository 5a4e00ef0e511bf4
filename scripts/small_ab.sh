#!/bin/bash
# A/B of library builds on query-sized encodes (LIBS="a.so b.so", NQS="1 16 32")
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
OUT=gpurun_out/small_ab.txt; : > $OUT
for rnd in 1 2; do
 for lib in ${LIBS:-libsgpt_hip.so}; do
  for nq in ${NQS:-1 16 32}; do
    echo -n "$lib r$rnd " >> $OUT
    SGPT_HIP_LIB=$R/sgpt_amd/lib/$lib LL=${LL:-0} NQ=$nq python scripts/small_batch_profile.py 2>&1 | tail -1 >> $OUT
  done
 done
done
cat $OUT
