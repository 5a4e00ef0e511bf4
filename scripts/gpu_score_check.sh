#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for lib in libsgpt_hip_nodefer.so libsgpt_hip.so; do
export SGPT_HIP_LIB=$R/sgpt_amd/lib/$lib
for nq in 64 16 1; do echo -n "$lib "; NQ=$nq python scripts/score_bench.py 2>&1 | grep "per pass"; done
for nq in 64 16; do echo -n "$lib drift 0.1: "; DRIFT=0.1 NQ=$nq python scripts/score_bench.py 2>&1 | grep "per pass"; done
done | tee gpurun_out/score_defer_ab.txt
