#!/bin/bash
# Encode-call size sweep: does a smaller per-call working set keep producer->consumer intermediates in the
# 256 MiB Infinity Cache?  (documents per sgpt_encode call; 4096 documents per step either way)
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in 1024 512 256 128; do
  ( timeout 300 python bench.py --steps 6 --warmup 1 --call $c --no-1m --no-cpu-baseline ) 2>/dev/null | grep '^{' | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('call', $c, 'sent/s', d['value'], 'gemm TF/s', d['roofline']['achieved'], 'gemm_share', d['roofline']['gemm_share_of_step'])" | tee -a gpurun_out/call_sweep.log
done
