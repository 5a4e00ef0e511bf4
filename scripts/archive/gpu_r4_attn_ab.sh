#!/bin/bash
# Round-4 attention A/B: LIBS = builds of the same ABI (SGPT_LIB_TAG / SGPT_EXTRA_FLAGS, see attn.hip's switches).
# Per lib: the long-sequence tests, one encode-step timing per sequence length, and the attention kernel's average from a kernel trace.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
LIBS=${LIBS:-"libsgpt_hip.so libsgpt_hip_zz.so"}
SEQS=${SEQS:-"300 512"}
: > gpurun_out/attn_ab.txt
for lib in $LIBS; do
  echo "=== $lib" >> gpurun_out/attn_ab.txt
  SGPT_HIP_LIB=$R/sgpt_amd/lib/$lib timeout 900 python -m pytest tests/test_gpu_encode.py -q -x -m gpu -k "long or window or identical or cfg3" 2>&1 | tail -1 >> gpurun_out/attn_ab.txt
  for S in $SEQS; do
    ( cd /tmp && SGPT_HIP_LIB=$R/sgpt_amd/lib/$lib S=$S rocprofv3 --kernel-trace -d $R/gpurun_out/attnprof -o a -- python $R/scripts/attn_probe_time.py ) > gpurun_out/attn_prof.log 2>&1
    echo "$lib $(grep 'ms per' gpurun_out/attn_prof.log) (under the tracer)" >> gpurun_out/attn_ab.txt
    python scripts/prof_summary.py $(find gpurun_out/attnprof -name '*.db' | head -1) 12 | grep attn | cut -c1-220 >> gpurun_out/attn_ab.txt
    rm -rf gpurun_out/attnprof
  done
done
for rnd in 1 2; do for lib in $LIBS; do for S in $SEQS; do
  echo "$lib r$rnd $(SGPT_HIP_LIB=$R/sgpt_amd/lib/$lib S=$S python scripts/attn_probe_time.py 2>/dev/null | grep 'ms per')" >> gpurun_out/attn_ab.txt
done; done; done
cat gpurun_out/attn_ab.txt
