#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_search.py tests/test_gpu_dist.py -q --timeout=600 ) > gpurun_out/r4f_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4f_tests.log | cut -c1-300
( for n in 125000 250000 1000000; do N=$n python scripts/score_bench.py; done; NQ=64 python scripts/score_bench.py; NQ=16 python scripts/score_bench.py; echo -n "DRIFT=0.9 "; DRIFT=0.9 python scripts/score_bench.py; echo -n "K=1001 "; K=1001 python scripts/score_bench.py ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4f_score_bench.txt
cd /tmp; rm -rf /tmp/sp2; mkdir -p /tmp/sp2
N=125000 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/sp2 -o trace -- python $R/scripts/score_bench.py > /tmp/sp2.log 2>&1
cd $R; python scripts/prof_summary.py /tmp/sp2/trace_results.db 10 | cut -c1-200 | tee gpurun_out/r4f_shard_profile.csv
( timeout 900 python bench.py --steps 10 --no-cpu-baseline --no-varlen ) 2>/dev/null | grep '^{' > gpurun_out/r4f_bench.json; python -c "
import json; d=json.load(open('gpurun_out/r4f_bench.json')); print(d['value'], d['queries_per_sec_at_1M_corpus'], json.dumps(d['projected_8gpu']), json.dumps(d['precision_modes']))"
