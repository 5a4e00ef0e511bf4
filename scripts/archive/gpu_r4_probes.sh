#!/bin/bash
# Round 4 probes: CU-partitioned two-call pipeline; slab residency (call-size sweep with HBM bytes).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
( for cap in 0 240 224 192; do for call in 1024 512; do CAP=$cap CALL=$call CALLS=$((16384 / call)) python scripts/dual_stream_probe.py 2>&1 | grep -v amdgpu | tail -2; done; done ) | tee gpurun_out/r4_dual_stream_cu_cap.txt
( for c in 1024 512 256 128 64; do CALL=$c python scripts/slab_probe.py 2>&1 | grep CALL; done ) | tee gpurun_out/r4_slab_probe.txt
cd /tmp
for c in 1024 128 64; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$C; mkdir -p /tmp/pmc_$C
    CALL=$c timeout 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$C -o pmc -- python $R/scripts/slab_probe.py > /tmp/pmc_$C.log 2>&1
    echo -n "CALL=$c "; python $R/scripts/pmc_totals.py /tmp/pmc_$C/pmc_results.db $C $( [ $C = FETCH_SIZE ] && echo 2048 || echo 1024 ) | head -4
  done
done 2>&1 | tee -a $R/gpurun_out/r4_slab_probe.txt
