#!/bin/bash
# Round-3, later call: scorer after the schedule went back to doubling (single-piece fallback kept), tokens per sgpt_encode call
# sweep on the bench step, the new graph / range-shift test.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_encode.py tests/test_gpu_kernels.py tests/test_gpu_search.py -m gpu -q --timeout=600 -x ) > gpurun_out/pytest_part.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_part.log | cut -c1-300
( for nq in 1000 128 64 16; do NQ=$nq python scripts/score_bench.py; done; for dr in 0.1 0.3 0.6 0.9; do echo -n "DRIFT=$dr "; DRIFT=$dr python scripts/score_bench.py; done; for n in 500000 250000 125000; do N=$n python scripts/score_bench.py; done ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/score_bench.txt
bash scripts/score_prof.sh > gpurun_out/score_prof.log 2>&1; head -8 gpurun_out/score_prof_summary.csv
for call in 512 1024 2048 4096; do echo -n "docs per sgpt_encode call $call: "; timeout 600 python bench.py --call $call --steps 12 --no-cpu-baseline --no-1m --no-varlen 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['achieved'], d['roofline']['end_to_end_frac_of_mfma_roofline'])"; done 2>&1 | tee gpurun_out/call_sweep.txt
