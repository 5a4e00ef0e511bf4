#!/bin/bash
# the whole GPU suite with the parity log (the first step of scripts/gpu_r3_final.sh on its own)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
rm -f gpurun_out/parity.jsonl
( SGPT_PARITY_LOG=$R/gpurun_out/parity.jsonl timeout 1800 python -m pytest tests -m gpu -q --timeout=900 -rA ) > gpurun_out/pytest_full.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_full.log
grep -E "^(cfg|outlier|f16 range)" gpurun_out/pytest_full.log | cut -c1-400 > gpurun_out/parity_numbers.txt; grep -E "passed|failed" gpurun_out/pytest_full.log | tail -1 >> gpurun_out/parity_numbers.txt
