#!/bin/bash
# Round 2, first GPU call: correctness of everything new under both GEMM variants, A/B micro-bench, two bench lines.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -q --timeout=600 -x ) > gpurun_out/pytest_v0.log 2>&1; echo "pytest(variant 0) rc=$?"; tail -3 gpurun_out/pytest_v0.log
( SGPT_GEMM_W=1 timeout 900 python -m pytest tests -m gpu -q --timeout=600 ) > gpurun_out/pytest_v1.log 2>&1; echo "pytest(variant 1) rc=$?"; tail -3 gpurun_out/pytest_v1.log
( timeout 600 python -m pytest tests/test_gpu_parity_cfg2.py tests/test_gpu_encode.py -m gpu -q -s -k "cfg2 or f16_vs_golden or bf16_vs_golden" ) > gpurun_out/parity_numbers.log 2>&1; grep -E "cfg2|f16:|bf16:" gpurun_out/parity_numbers.log | head -60
( timeout 600 python scripts/gemm_bench.py ) > gpurun_out/gemm_ab.log 2>&1; echo "gemm_bench rc=$?"; cat gpurun_out/gemm_ab.log
( timeout 600 python bench.py --steps 12 ) > gpurun_out/bench_v0.log 2>&1; echo "bench v0 rc=$?"; grep '^{' gpurun_out/bench_v0.log > gpurun_out/bench_v0.json; cut -c1-1500 gpurun_out/bench_v0.json; tail -5 gpurun_out/bench_v0.log | grep -v '^{'
( SGPT_GEMM_W=1 timeout 600 python bench.py --steps 12 --no-cpu-baseline ) > gpurun_out/bench_v1.log 2>&1; echo "bench v1 rc=$?"; grep '^{' gpurun_out/bench_v1.log > gpurun_out/bench_v1.json; cut -c1-900 gpurun_out/bench_v1.json; tail -5 gpurun_out/bench_v1.log | grep -v '^{'
python __graft_entry__.py --smoke 2>&1 | tail -5
