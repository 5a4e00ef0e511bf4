#!/bin/bash
# Round 4, fourth GPU call: LDS-staged candidate appends + arg-max merge: exactness tests, shard timings, kernel table at 125 k.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_search.py -q --timeout=600 ) > gpurun_out/r4d_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r4d_tests.log | cut -c1-300
( for n in 125000 250000 500000 1000000; do N=$n python scripts/score_bench.py; done; for nq in 128 64 16; do NQ=$nq python scripts/score_bench.py; done; NQ=16 N=125000 python scripts/score_bench.py; NQ=128 N=125000 python scripts/score_bench.py; for dr in 0.1 0.5 0.9; do echo -n "DRIFT=$dr "; DRIFT=$dr python scripts/score_bench.py; done; echo -n "K=101 "; K=101 python scripts/score_bench.py;  echo -n "K=1001 "; K=1001 python scripts/score_bench.py ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4d_score_bench.txt
cd /tmp; rm -rf /tmp/sp2; mkdir -p /tmp/sp2
N=125000 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/sp2 -o trace -- python $R/scripts/score_bench.py > /tmp/sp2.log 2>&1
cd $R; python scripts/prof_summary.py /tmp/sp2/trace_results.db 12 | cut -c1-230 | tee gpurun_out/r4d_shard_profile.csv
