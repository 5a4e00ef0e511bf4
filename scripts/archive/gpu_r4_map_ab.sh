#!/bin/bash
# XCD-balanced tile runs (libsgpt_hip.so) against the supertile order (libsgpt_hip_premap.so): the whole GPU suite under the
# new library, then alternating encode latencies at query / mid sizes and the GEMM shapes of a mid-size call.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out; : > gpurun_out/map_ab.txt
( timeout 900 python -m pytest tests -m gpu -q --timeout=600 -x ) > gpurun_out/pytest_map.log 2>&1; echo "pytest rc=$? $(grep -E 'passed|failed' gpurun_out/pytest_map.log | tail -1)" >> gpurun_out/map_ab.txt; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_map.log | cut -c1-200 >> gpurun_out/map_ab.txt
for rnd in 1 2; do for lib in libsgpt_hip_premap.so libsgpt_hip.so; do
  for nq in 16 128; do echo "$lib r$rnd $(SGPT_HIP_LIB=$R/sgpt_amd/lib/$lib LL=0 NQ=$nq python scripts/small_batch_profile.py 2>/dev/null | grep 'per encode')" >> gpurun_out/map_ab.txt; done
  for cfg in "300 4 32" "1000 4 32" "3000 4 32" "1000 8 64"; do set -- $cfg
    echo "$lib r$rnd len $2..$3 $(SGPT_HIP_LIB=$R/sgpt_amd/lib/$lib NQ=$1 LMIN=$2 LMAX=$3 python scripts/mid_batch_profile.py 2>/dev/null | grep 'per encode')" >> gpurun_out/map_ab.txt
  done
done; done
cat gpurun_out/map_ab.txt
