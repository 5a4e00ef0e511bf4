#!/bin/bash
# Round artefacts in one gpurun call: tests, bench (N=1 default), kernel-trace stats, PMC traffic.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
( timeout 1200 python -m pytest tests -m gpu -q --timeout=600 ) > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 gpurun_out/pytest.log
( timeout 900 python bench.py ) > gpurun_out/bench_full.log 2>&1; echo "bench rc=$?"; grep '^{' gpurun_out/bench_full.log > gpurun_out/bench_full.json; cut -c1-600 gpurun_out/bench_full.json
bash scripts/gpu_prof.sh > gpurun_out/prof.log 2>&1; head -14 gpurun_out/prof_summary.csv
bash scripts/gpu_pmc.sh > gpurun_out/pmc.log 2>&1; grep -E "gemm256|attn|layernorm" gpurun_out/pmc_summary.csv | head -30
