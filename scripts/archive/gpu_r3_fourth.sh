#!/bin/bash
# Round-3: the split-precision Q / K projection (SGPTModel(precise_qk=True)): tests, parity at SGPT-125M and SGPT-1.3B shape, cost.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
rm -f gpurun_out/parity_qk.jsonl
( SGPT_PARITY_LOG=$R/gpurun_out/parity_qk.jsonl timeout 1500 python -m pytest tests/test_gpu_encode.py tests/test_gpu_parity_cfg2.py "tests/test_gpu_parity_large.py" -m gpu -q -s --timeout=900 -k "precise or cfg2 or cfg3_neo13b or graph or range" ) > gpurun_out/pytest_qk.log 2>&1; echo "pytest rc=$?"; grep -E "precise_qk|^cfg[23]|passed|failed|Error" gpurun_out/pytest_qk.log | cut -c1-330
for spec in "125m " "125m --precise-qk" "1.3b " "1.3b --precise-qk" "2.7b --precise-qk"; do set -- $spec; ch=4096; [ "$1" != "125m" ] && ch=1024
  echo -n "$spec: "; timeout 600 python bench.py --model $1 $2 --steps 6 --warmup 1 --chunk $ch --no-cpu-baseline --no-1m --no-varlen 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['achieved'], d['roofline']['end_to_end_frac_of_mfma_roofline'])"; done 2>&1 | tee gpurun_out/precise_qk_cost.txt
