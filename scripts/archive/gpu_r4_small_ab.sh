#!/bin/bash
# Same-box A/B of library builds on query-sized encodes: tests per build, then alternating latency runs (1 / 16 / 128 / 1000 queries).
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
LIBS=${LIBS:-"libsgpt_hip_prepre.so libsgpt_hip.so"}
: > gpurun_out/small_ab.txt
for lib in $LIBS; do
  echo "=== $lib" >> gpurun_out/small_ab.txt
  SGPT_HIP_LIB=$R/sgpt_amd/lib/$lib timeout 900 python -m pytest tests/test_gpu_linear.py tests/test_gpu_encode.py -q -x -m gpu 2>&1 | tail -1 >> gpurun_out/small_ab.txt
done
for rnd in 1 2 3; do for lib in $LIBS; do for nq in 1 16 128; do
  echo "$lib r$rnd $(SGPT_HIP_LIB=$R/sgpt_amd/lib/$lib LL=0 NQ=$nq python scripts/small_batch_profile.py 2>/dev/null | grep 'per encode')" >> gpurun_out/small_ab.txt
done; done; done
for lib in $LIBS; do echo "$lib $(SGPT_HIP_LIB=$R/sgpt_amd/lib/$lib NQ=1000 python scripts/mid_batch_profile.py 2>/dev/null | grep 'per encode')" >> gpurun_out/small_ab.txt; done
cat gpurun_out/small_ab.txt
