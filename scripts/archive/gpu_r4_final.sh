#!/bin/bash
# Round-4 artefacts in one gpurun call: GPU suite (+ parity log), default bench line, rocprofv3 kernel summary, PMC traffic,
# GEMM shapes, scorer passes (shards, drift, k), query-sized latency, every model size with its parity column.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
rm -f gpurun_out/parity.jsonl
( SGPT_PARITY_LOG=$R/gpurun_out/parity.jsonl timeout 1800 python -m pytest tests -m gpu -q --timeout=900 -rA ) > gpurun_out/pytest_full.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_full.log
grep -E "^(cfg|outlier|f16 range)" gpurun_out/pytest_full.log | cut -c1-700 > gpurun_out/parity_numbers.txt; grep -E "passed|failed" gpurun_out/pytest_full.log | tail -1 >> gpurun_out/parity_numbers.txt
grep -E "^(FAILED|ERROR)" gpurun_out/pytest_full.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -6
( timeout 900 python bench.py ) > gpurun_out/bench_full.log 2>&1; echo "bench rc=$?"; grep '^{' gpurun_out/bench_full.log > gpurun_out/bench_n1.json; cut -c1-700 gpurun_out/bench_n1.json
BENCH_STEPS=3 bash scripts/gpu_prof.sh > gpurun_out/prof.log 2>&1; head -14 gpurun_out/prof_summary.csv
bash scripts/gpu_pmc.sh > gpurun_out/pmc.log 2>&1; tail -3 gpurun_out/pmc.log; head -c 500 gpurun_out/pmc_traffic.json; echo
( VARIANTS=0 DTYPES=f16,bf16 ROUNDS=3 python scripts/gemm_bench.py ) 2>&1 | grep -v amdgpu.ids > gpurun_out/gemm_shapes.txt; grep "block total" gpurun_out/gemm_shapes.txt
( for n in 1000000 500000 250000 125000; do N=$n python scripts/score_bench.py; done; for nq in 128 64 16; do NQ=$nq python scripts/score_bench.py; done; for dr in 0.1 0.3 0.6 0.9; do echo -n "DRIFT=$dr "; DRIFT=$dr python scripts/score_bench.py; done; echo -n "K=101 "; K=101 python scripts/score_bench.py; echo -n "K=1001 "; K=1001 python scripts/score_bench.py; echo -n "K=1001 N=125000 "; K=1001 N=125000 python scripts/score_bench.py ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/score_bench.txt
bash scripts/score_prof.sh > gpurun_out/score_prof.log 2>&1; head -8 gpurun_out/score_prof_summary.csv
( for nq in 1 16 128; do LL=0 NQ=$nq python scripts/small_batch_profile.py 2>&1 | grep "per encode"; done; python scripts/query_side_breakdown.py ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/query_side.txt
python scripts/models_table.py gpurun_out/parity.jsonl gpurun_out/models.jsonl > gpurun_out/models.log 2>&1; cut -c1-300 gpurun_out/models.jsonl
( for s in 300 512; do timeout 300 python bench.py --seq $s --call $((131072 / s)) --chunk $((4 * (131072 / s))) --steps 4 --warmup 1 --no-cpu-baseline --no-1m --no-varlen 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seq', $s, d['value'], 'sent/s', d['roofline']['end_to_end_frac_of_mfma_roofline'])"; done ) | tee gpurun_out/seq_lengths.txt
