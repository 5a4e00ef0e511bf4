#!/bin/bash
# Round-3, first GPU call: the whole GPU suite (new: full-shape parity at configs[2]-[4] + the engineered-outlier case, RCCL
# through the C ABI, the sharded search, range shifts), the default bench line, a rocprofv3 kernel summary.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
rm -f gpurun_out/parity_large.jsonl
( SGPT_PARITY_LOG=$R/gpurun_out/parity_large.jsonl timeout 1500 python -m pytest tests/test_gpu_parity_large.py -m gpu -q -s --timeout=900 ) > gpurun_out/pytest_large.log 2>&1; echo "pytest large rc=$?"; grep -E "^(outlier|cfg[345])|passed|failed" gpurun_out/pytest_large.log | cut -c1-330
( timeout 1200 python -m pytest tests -m gpu -q --timeout=600 --deselect tests/test_gpu_parity_large.py -x ) > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/pytest.log | cut -c1-300
( timeout 900 python bench.py ) > gpurun_out/bench_full.log 2>&1; echo "bench rc=$?"; grep '^{' gpurun_out/bench_full.log > gpurun_out/bench_n1.json; cut -c1-1500 gpurun_out/bench_n1.json; tail -5 gpurun_out/bench_full.log | grep -v '^{' | cut -c1-300
BENCH_STEPS=3 bash scripts/gpu_prof.sh > gpurun_out/prof.log 2>&1; head -24 gpurun_out/prof_summary.csv
