#!/bin/bash
# Same-box A/B: list capacity of the sampled scorer schedule for nq <= 64 (512 / 1024 / 2048 entries)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for rnd in 1 2; do
  for lib in libsgpt_hip.so libsgpt_hip_cap1024.so libsgpt_hip_cap2048.so; do
    for nq in 16 64; do
      echo -n "$lib round $rnd: "; SGPT_HIP_LIB=$R/sgpt_amd/lib/$lib NQ=$nq python scripts/score_bench.py 2>&1 | grep -v amdgpu
    done
  done
done | tee gpurun_out/r4_cap64_ab.txt
