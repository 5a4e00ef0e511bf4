#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_search.py -q --timeout=600 ) > gpurun_out/r4h_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4h_tests.log | cut -c1-300
( for n in 1000000 500000 250000 125000; do N=$n python scripts/score_bench.py; done; for nq in 128 64 16; do NQ=$nq python scripts/score_bench.py; done; for dr in 0.1 0.3 0.6 0.9; do echo -n "DRIFT=$dr "; DRIFT=$dr python scripts/score_bench.py; done; echo -n "K=101 "; K=101 python scripts/score_bench.py; echo -n "K=1001 "; K=1001 python scripts/score_bench.py; echo -n "K=1001 N=125000 "; K=1001 N=125000 python scripts/score_bench.py; echo -n "N=4M "; N=4000000 python scripts/score_bench.py ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/score_bench.txt
