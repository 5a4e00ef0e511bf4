#!/bin/bash
# Round 4, closing call: the whole GPU suite with the parity log, smoke, the default bench line (wall time noted), kernel stats.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
rm -f gpurun_out/parity.jsonl
( SGPT_PARITY_LOG=$R/gpurun_out/parity.jsonl timeout 1800 python -m pytest tests -m gpu -q --timeout=900 -rA ) > gpurun_out/pytest_full.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/pytest_full.log | tail -1; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_full.log | cut -c1-200
grep -E "^(cfg|outlier|f16 range)" gpurun_out/pytest_full.log | cut -c1-700 > gpurun_out/parity_numbers.txt; grep -E "passed|failed" gpurun_out/pytest_full.log | tail -1 >> gpurun_out/parity_numbers.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -6
t0=$(date +%s); ( timeout 900 python bench.py ) > gpurun_out/bench_full.log 2>&1; echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s"; grep '^{' gpurun_out/bench_full.log > gpurun_out/bench_n1.json; python -c "
import json; d=json.load(open('gpurun_out/bench_n1.json')); print(d['value'], d['roofline']['frac'], d['queries_per_sec_at_1M_corpus'], d['queries_per_sec_at_1M_corpus_incl_query_encode_by_nq'], json.dumps(d['projected_8gpu']), json.dumps(d['precision_modes'])[:200], json.dumps(d['queries_per_sec_at_1M_corpus_k1001']), json.dumps(d['varlen']))"
BENCH_STEPS=3 bash scripts/gpu_prof.sh > gpurun_out/prof.log 2>&1; head -12 gpurun_out/prof_summary.csv | cut -c1-200
( for n in 1000000 500000 250000 125000; do N=$n python scripts/score_bench.py; done; for nq in 128 64 16; do NQ=$nq python scripts/score_bench.py; done; for dr in 0.1 0.3 0.6 0.9; do echo -n "DRIFT=$dr "; DRIFT=$dr python scripts/score_bench.py; done; echo -n "K=101 "; K=101 python scripts/score_bench.py; echo -n "K=1001 "; K=1001 python scripts/score_bench.py; echo -n "K=1001 N=125000 "; K=1001 N=125000 python scripts/score_bench.py ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/score_bench.txt
bash scripts/score_prof.sh > gpurun_out/score_prof.log 2>&1; head -8 gpurun_out/score_prof_summary.csv | cut -c1-200
( for nq in 1 16 128; do LL=0 NQ=$nq python scripts/small_batch_profile.py 2>&1 | grep "per encode"; done; python scripts/query_side_breakdown.py ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/query_side.txt
# (calls of at most 131 072 allocated rows: 436 x 300 at S = 300 since sequences are packed on even rows)
( for cfg in "300 1744 436" "512 1024 256"; do set -- $cfg; timeout 300 python bench.py --seq $1 --chunk $2 --call $3 --steps 4 --warmup 1 --no-cpu-baseline --no-1m --no-varlen --no-modes 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seq', $1, d['value'], 'sent/s', d['roofline']['end_to_end_frac_of_mfma_roofline'])"; done ) | tee gpurun_out/seq_lengths.txt
( timeout 600 python bench.py --model 1.3b --no-cpu-baseline --no-1m --no-varlen --no-modes 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1.3b', d['value'], 'sent/s', d['roofline']['end_to_end_frac_of_mfma_roofline'], d['config'].get('precise_qk'))" ) | tee gpurun_out/model_13b.txt
