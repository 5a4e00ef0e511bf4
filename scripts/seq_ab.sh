#!/bin/bash
# A/B of library builds on the encode step at several sequence lengths (LIBS="a.so b.so"); ~131k token rows per call
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
OUT=gpurun_out/seq_ab.txt; : > $OUT
for rnd in 1 2; do
 for lib in ${LIBS:-libsgpt_hip.so}; do
  for cfg in "128 4096 1024" "300 1720 430" "512 1024 256"; do
    set -- $cfg
    echo -n "$lib r$rnd seq $1: " >> $OUT
    SGPT_HIP_LIB=$R/sgpt_amd/lib/$lib python bench.py --steps 6 --warmup 2 --seq $1 --chunk $2 --call $3 --no-cpu-baseline --no-1m --no-varlen ${BENCH_ARGS} 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'sent/s', d['ms_per_step'], 'ms/step')" >> $OUT
  done
 done
done
cat $OUT
