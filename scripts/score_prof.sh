#!/bin/bash
# rocprofv3 kernel trace of the scoring-only benchmark (nq queries x N bf16 documents)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
rm -rf $R/gpurun_out/sprof; mkdir -p $R/gpurun_out/sprof
cd /tmp
for nq in 1000 128; do NQ=$nq timeout 120 python $R/scripts/score_bench.py 2>/dev/null; done
NQ=${NQ:-1000} timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/sprof -o trace -- python $R/scripts/score_bench.py > $R/gpurun_out/sprof.log 2>&1
cd $R
python scripts/prof_summary.py gpurun_out/sprof/trace_results.db 12 | tee gpurun_out/score_prof_summary.csv
rm -f gpurun_out/sprof/trace_results.db
