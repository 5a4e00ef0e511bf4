#!/bin/bash
# rocprofv3 kernel trace of a short bench run; per-kernel summary -> gpurun_out/prof_summary.csv
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
rm -rf $R/gpurun_out/prof; mkdir -p $R/gpurun_out/prof
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o trace -- python $R/bench.py --steps ${BENCH_STEPS:-3} --warmup 1 --no-cpu-baseline --no-modes ${BENCH_ARGS} > $R/gpurun_out/prof_bench.log 2>&1
echo "rocprof rc=$?"
cd $R
python scripts/prof_summary.py gpurun_out/prof/trace_results.db 30 | tee gpurun_out/prof_summary.csv
grep '^{' gpurun_out/prof_bench.log | cut -c1-400
rm -f gpurun_out/prof/trace_results.db
