#!/bin/bash
# kernel durations of the 16-query search against a 1 M-document shard
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
( cd /tmp && NQ=${NQ:-16} rocprofv3 --kernel-trace --stats -d $R/gpurun_out/s16 -o s16 -- python $R/scripts/score_bench.py ) > gpurun_out/score16_prof.log 2>&1
grep "per pass" gpurun_out/score16_prof.log
python scripts/prof_summary.py gpurun_out/s16/s16_results.db 12 | cut -c1-220 | tee gpurun_out/score16_kernel_stats.csv; rm -rf gpurun_out/s16
