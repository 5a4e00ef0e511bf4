#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/mid -o mid -- python $R/scripts/mid_batch_profile.py ) > gpurun_out/mid_prof.log 2>&1
python scripts/prof_summary.py gpurun_out/mid/mid_results.db 14 | cut -c1-200 | tee gpurun_out/mid_batch_kernel_stats.csv; rm -rf gpurun_out/mid
grep "per encode" gpurun_out/mid_prof.log
