#!/bin/bash
# Same-box A/B of library builds on the scorer: correctness (scorer + search tests) per build, then alternating timings.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
LIBS=${LIBS:-"libsgpt_hip_prethr.so libsgpt_hip.so"}
: > gpurun_out/score_ab.txt
for lib in $LIBS; do
  echo "=== $lib" >> gpurun_out/score_ab.txt
  SGPT_HIP_LIB=$R/sgpt_amd/lib/$lib timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_search.py -q -x -m gpu 2>&1 | tail -1 >> gpurun_out/score_ab.txt
done
for rnd in 1 2 3; do for lib in $LIBS; do
  for cfg in "1000 1000000" "1000 125000" "128 1000000" "16 1000000"; do set -- $cfg
    echo "$lib r$rnd $(SGPT_HIP_LIB=$R/sgpt_amd/lib/$lib NQ=$1 N=$2 python scripts/score_bench.py 2>/dev/null | grep 'ms per pass')" >> gpurun_out/score_ab.txt
  done
done; done
cat gpurun_out/score_ab.txt
