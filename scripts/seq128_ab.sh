#!/bin/bash
# A/B of library builds on the default encode step (seq 128) and the variable-length leg
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
OUT=gpurun_out/seq128_ab.txt; : > $OUT
for rnd in 1 2 3; do
 for lib in ${LIBS:-libsgpt_hip.so}; do
    echo -n "$lib r$rnd: " >> $OUT
    SGPT_HIP_LIB=$R/sgpt_amd/lib/$lib python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-1m 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'sent/s', d['ms_per_step'], 'ms/step; varlen', d.get('varlen',{}).get('sentences_per_sec', d.get('varlen')))" >> $OUT
 done
done
cat $OUT
