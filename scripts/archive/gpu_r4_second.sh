#!/bin/bash
# Round 4, second GPU call: fixed precision tests, the sampled-threshold scorer schedule (exactness tests + shard timings),
# the precise_qk variants on the 1.3B / 2.7B fixtures with their cost.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_precision.py tests/test_gpu_kernels.py tests/test_gpu_search.py tests/test_gpu_dist.py -q --timeout=600 -k "not x3_mode and not every_class" ) > gpurun_out/r4b_tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r4b_tests.log | cut -c1-300
( timeout 600 python -m pytest tests/test_gpu_encode.py -q --timeout=600 -k "precise_qk" ) 2>&1 | tail -3
( for n in 125000 250000 500000 1000000; do N=$n python scripts/score_bench.py; done; for nq in 128 64 16; do NQ=$nq python scripts/score_bench.py; done; NQ=16 N=125000 python scripts/score_bench.py; NQ=128 N=125000 python scripts/score_bench.py; for dr in 0.1 0.5 0.9; do echo -n "DRIFT=$dr "; DRIFT=$dr python scripts/score_bench.py; done; echo -n "K=101 "; K=101 python scripts/score_bench.py;  echo -n "K=1001 "; K=1001 python scripts/score_bench.py; echo -n "K=1001 N=125000 "; K=1001 N=125000 python scripts/score_bench.py ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_score_bench.txt
rm -f gpurun_out/parity.jsonl
( SGPT_PARITY_LOG=$R/gpurun_out/parity.jsonl timeout 1500 python -m pytest tests/test_gpu_parity_large.py -q --timeout=900 -rA -k "neo13b_specb or neo27b" ) > gpurun_out/r4b_parity_large.log 2>&1; echo "parity rc=$?"
grep -E "^(cfg|outlier)" gpurun_out/r4b_parity_large.log | cut -c1-330 | tee gpurun_out/r4b_parity_numbers.txt; grep -E "passed|failed" gpurun_out/r4b_parity_large.log | tail -1
B="python bench.py --no-cpu-baseline --no-1m --no-varlen"
( for v in logits full+logits qkv+logits attn; do
  ( timeout 600 $B --model 1.3b --steps 4 --warmup 1 --precise-qk $v ) 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1.3b precise_qk=$v', d['value'], 'sent/s', d['ms_per_step'], 'ms/step')"
done
for v in off logits full full+logits qkv+logits attn; do
  ( timeout 600 $B --model 2.7b --steps 3 --warmup 1 --precise-qk $v ) 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('2.7b precise_qk=$v', d['value'], 'sent/s', d['ms_per_step'], 'ms/step')"
done ) 2>&1 | tee gpurun_out/r4b_precise_qk_cost.txt
