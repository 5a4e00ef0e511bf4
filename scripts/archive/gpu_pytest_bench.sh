#!/bin/bash
# The round-end gate in one call: the whole GPU suite (parity log kept), smoke, the default bench line.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out; rm -f gpurun_out/parity.jsonl
( SGPT_PARITY_LOG=$R/gpurun_out/parity.jsonl timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -rA ) > gpurun_out/pytest_full.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/pytest_full.log | tail -1; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_full.log | cut -c1-220
grep -E "^(cfg|outlier|f16 range)" gpurun_out/pytest_full.log | cut -c1-700 > gpurun_out/parity_numbers.txt; grep -E "passed|failed" gpurun_out/pytest_full.log | tail -1 >> gpurun_out/parity_numbers.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -5
t0=$(date +%s); ( timeout 600 python bench.py ) > gpurun_out/bench_full.log 2>&1; echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s"; grep '^{' gpurun_out/bench_full.log > gpurun_out/bench_n1.json; python -c "
import json; d=json.load(open('gpurun_out/bench_n1.json')); print(d['value'], d['roofline']['frac'], d['queries_per_sec_at_1M_corpus'], d['queries_per_sec_at_1M_corpus_incl_query_encode_by_nq'], json.dumps(d['projected_8gpu']), json.dumps(d['varlen']))"
python scripts/query_side_breakdown.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/query_side.txt
