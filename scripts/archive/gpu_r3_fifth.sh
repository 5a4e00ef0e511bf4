#!/bin/bash
# Round-3: what a 16-bit copy of the new residual costs the residual epilogues (probe build -DSGPT_PROBE_X16), the full GPU suite
# with precise_qk as the GPT-Neo >= 2048 default.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
( LIBS="libsgpt_hip.so libsgpt_hip_x16probe.so" ROUNDS=3 bash scripts/ab_libs.sh ) > gpurun_out/ab_x16.txt 2>&1; grep -E "===|oproj|fc2|block total" gpurun_out/ab_x16.txt
rm -f gpurun_out/parity.jsonl
( SGPT_PARITY_LOG=$R/gpurun_out/parity.jsonl timeout 1800 python -m pytest tests -m gpu -q --timeout=900 -rA ) > gpurun_out/pytest_full.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_full.log | cut -c1-200
grep -E "^(cfg|outlier|f16 range|tiny)" gpurun_out/pytest_full.log | cut -c1-400 > gpurun_out/parity_numbers.txt; grep -E "passed|failed" gpurun_out/pytest_full.log | tail -1 >> gpurun_out/parity_numbers.txt
