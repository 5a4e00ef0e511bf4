#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
rm -f gpurun_out/parity.jsonl
( SGPT_PARITY_LOG=$R/gpurun_out/parity.jsonl timeout 1800 python -m pytest tests -m gpu -q --timeout=900 -rA ) > gpurun_out/pytest_full.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/pytest_full.log | tail -1; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_full.log | cut -c1-200
grep -E "^(cfg|outlier|f16 range)" gpurun_out/pytest_full.log | cut -c1-700 > gpurun_out/parity_numbers.txt; grep -E "passed|failed" gpurun_out/pytest_full.log | tail -1 >> gpurun_out/parity_numbers.txt
( for n in 1000000 500000 250000 125000; do N=$n python scripts/score_bench.py; done; for nq in 128 64 16; do NQ=$nq python scripts/score_bench.py; done; for dr in 0.1 0.3 0.6 0.9; do echo -n "DRIFT=$dr "; DRIFT=$dr python scripts/score_bench.py; done; echo -n "K=101 "; K=101 python scripts/score_bench.py; echo -n "K=1001 "; K=1001 python scripts/score_bench.py; echo -n "K=1001 N=125000 "; K=1001 N=125000 python scripts/score_bench.py ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/score_bench.txt
cd /tmp; rm -rf /tmp/sp2; mkdir -p /tmp/sp2
N=125000 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/sp2 -o trace -- python $R/scripts/score_bench.py > /tmp/sp2.log 2>&1
cd $R; python scripts/prof_summary.py /tmp/sp2/trace_results.db 12 | cut -c1-200 | tee gpurun_out/r4g_shard_profile.csv
( timeout 900 python bench.py ) > gpurun_out/bench_full.log 2>&1; echo "bench rc=$?"; grep '^{' gpurun_out/bench_full.log > gpurun_out/bench_n1.json; python -c "
import json; d=json.load(open('gpurun_out/bench_n1.json')); print(d['value'], d['roofline']['frac'], d['queries_per_sec_at_1M_corpus'], json.dumps(d['projected_8gpu']), json.dumps(d['precision_modes']), json.dumps(d['queries_per_sec_at_1M_corpus_k1001']), json.dumps(d['varlen']))"
