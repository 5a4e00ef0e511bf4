#!/bin/bash
# per-kernel durations of a 16-query encode (rocprofv3 kernel trace) -> gpurun_out/small_batch_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/small -o small -- python $R/scripts/small_batch_profile.py ) > gpurun_out/small_prof.log 2>&1
python scripts/prof_summary.py gpurun_out/small/small_results.db 16 | tee gpurun_out/small_batch_kernel_stats.csv; rm -rf gpurun_out/small
grep "per encode" gpurun_out/small_prof.log
