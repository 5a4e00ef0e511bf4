#!/bin/bash
# Round-3, second GPU call: folded residual (acc = resid before the k-loop) A/B against the epilogue-add build, the scorer's
# fast-reject epilogue, the dual-stream overlap probe; the whole suite on the new kernels first.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -x ) > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/pytest.log | cut -c1-300
( ROUNDS=3 bash scripts/ab_libs.sh ) > gpurun_out/ab_fold.txt 2>&1; grep -E "===|oproj|fc2|block total" gpurun_out/ab_fold.txt
for rnd in 1 2; do for lib in libsgpt_hip_nofold.so libsgpt_hip.so; do
  echo -n "$lib round $rnd: "; SGPT_HIP_LIB=$R/sgpt_amd/lib/$lib timeout 600 python bench.py --steps 12 --no-cpu-baseline --no-1m --no-varlen 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['end_to_end_frac_of_mfma_roofline'])"
done; done 2>&1 | tee gpurun_out/ab_fold_bench.txt
( python scripts/dual_stream_probe.py ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/dual_stream.txt
( for nq in 1000 128 16; do NQ=$nq python scripts/score_bench.py; done ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/score_bench.txt
