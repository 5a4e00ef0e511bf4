#!/bin/bash
# Kernel trace of the repeated 1000-query encode (NQ, LMIN, LMAX env): per-kernel table + busy / idle share of the timeline
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/mid -o mid -- python $R/scripts/mid_batch_profile.py ) > gpurun_out/mid_prof.log 2>&1
DB=$(find gpurun_out/mid -name '*.db' | head -1)
python scripts/prof_summary.py $DB 14 | cut -c1-200 | tee gpurun_out/mid_batch_kernel_stats.csv
python scripts/prof_gaps.py $DB 1600 | tee -a gpurun_out/mid_batch_kernel_stats.csv
rm -rf gpurun_out/mid
grep "per encode" gpurun_out/mid_prof.log
