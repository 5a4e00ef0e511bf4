#!/bin/bash
# One gpurun call: smoke + GPU parity tests + a short bench; everything logged under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python __graft_entry__.py --smoke ) > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/smoke.log
( timeout 1200 python -m pytest tests -m gpu -q -rA --timeout=600 ${PYTEST_ARGS} ) > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest.log
( timeout 600 python bench.py --steps ${BENCH_STEPS:-6} --warmup 1 ${BENCH_ARGS} ) > gpurun_out/bench.log 2>&1; echo "bench rc=$?" | tee -a gpurun_out/bench.log
tail -5 gpurun_out/smoke.log; grep -E "passed|failed|error" gpurun_out/pytest.log | tail -5; tail -3 gpurun_out/bench.log
