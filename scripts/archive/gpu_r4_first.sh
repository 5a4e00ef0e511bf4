#!/bin/bash
# Round 4, first GPU call: the split-precision machinery (new tests), the full-shape parity cases with their new modes, the
# whole suite, and what the precise_qk variants / f16x3 / fp32 modes cost.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_precision.py -q -x --timeout=600 -rA ) > gpurun_out/r4_precision_tests.log 2>&1; echo "precision tests rc=$?"; grep -E "f16x3|split scorer|class .* level|crest|passed|failed|Error" gpurun_out/r4_precision_tests.log | cut -c1-300 | tail -60
rm -f gpurun_out/parity.jsonl
( SGPT_PARITY_LOG=$R/gpurun_out/parity.jsonl timeout 1500 python -m pytest tests/test_gpu_parity_large.py -q --timeout=900 -rA ) > gpurun_out/r4_parity_large.log 2>&1; echo "parity large rc=$?"
grep -E "^(cfg|outlier)" gpurun_out/r4_parity_large.log | cut -c1-700 > gpurun_out/parity_numbers.txt; grep -E "passed|failed" gpurun_out/r4_parity_large.log | tail -1 >> gpurun_out/parity_numbers.txt; cat gpurun_out/parity_numbers.txt
grep -E "^(FAILED|ERROR)" gpurun_out/r4_parity_large.log | cut -c1-300
( timeout 1200 python -m pytest tests -m gpu -q --timeout=900 --deselect tests/test_gpu_parity_large.py --deselect tests/test_gpu_precision.py ) > gpurun_out/r4_pytest_rest.log 2>&1; echo "rest rc=$?"; tail -5 gpurun_out/r4_pytest_rest.log
B="python bench.py --no-cpu-baseline --no-1m --no-varlen"
for v in off logits act+logits full; do
  ( timeout 600 $B --model 1.3b --steps 4 --warmup 1 --precise-qk $v ) 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1.3b precise_qk=$v', d['value'], 'sent/s', d['ms_per_step'], 'ms/step', d['roofline']['achieved'] if d.get('roofline') else None)"
done 2>&1 | tee gpurun_out/r4_precise_qk_cost.txt
( timeout 600 $B --steps 10 ) 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('125m default', d['value'], 'sent/s', d['config'].get('precision_probe'), d['roofline'])" | tee gpurun_out/r4_modes.txt
( timeout 600 $B --steps 4 --precision x3 ) 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('125m f16x3', d['value'], 'sent/s', d['ms_per_step'])" | tee -a gpurun_out/r4_modes.txt
( timeout 600 $B --steps 2 --warmup 1 --dtype fp32 ) 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('125m fp32-MFMA', d['value'], 'sent/s', d['ms_per_step'])" | tee -a gpurun_out/r4_modes.txt
( timeout 600 $B --model 1.3b --steps 3 --warmup 1 --precision x3 ) 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1.3b f16x3', d['value'], 'sent/s', d['ms_per_step'])" | tee -a gpurun_out/r4_modes.txt
