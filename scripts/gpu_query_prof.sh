#!/bin/bash
# Kernel trace of the repeated query-sized encode at NQ = 1 / 16 / 128: per-kernel table (who owns the latency of a small forward)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
for nq in ${NQS:-1 16 128}; do
  rm -rf gpurun_out/qprof; mkdir -p gpurun_out/qprof
  ( cd /tmp && NQ=$nq LL=0 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/qprof -o q -- python $R/scripts/small_batch_profile.py ) > gpurun_out/qprof_$nq.log 2>&1
  grep "per encode" gpurun_out/qprof_$nq.log
  DB=$(find gpurun_out/qprof -name '*.db' | head -1)
  python scripts/prof_summary.py $DB 16 | cut -c1-220 | tee gpurun_out/query_kernel_stats_nq$nq.csv
  python scripts/prof_gaps.py $DB 1600 | tee -a gpurun_out/query_kernel_stats_nq$nq.csv
  rm -rf gpurun_out/qprof
done
