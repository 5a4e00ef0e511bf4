#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
python scripts/shard_profile.py 2>&1 | grep -v amdgpu.ids
cd /tmp
rm -rf /tmp/sp; mkdir -p /tmp/sp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/sp -o trace -- python $R/scripts/shard_profile.py > /tmp/sp.log 2>&1
cd $R; python scripts/prof_summary.py /tmp/sp/trace_results.db 16 | cut -c1-230 | tee gpurun_out/r4_shard_profile_bench_data.csv
cd /tmp; rm -rf /tmp/sp2; mkdir -p /tmp/sp2
N=125000 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/sp2 -o trace -- python $R/scripts/score_bench.py > /tmp/sp2.log 2>&1
cd $R; python scripts/prof_summary.py /tmp/sp2/trace_results.db 12 | cut -c1-230 | tee gpurun_out/r4_shard_profile_score_bench.csv
