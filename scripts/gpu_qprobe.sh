#!/bin/bash
# builds (here, cross-compiled) and runs (on the GPU box) the qgemm tile sweep: MS="32 96 352 ..." bash scripts/gpu_qprobe.sh [build|run]
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
C=sgpt_amd/csrc
if [ "$1" = "build" ]; then
  F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -I $C"
  /opt/rocm/bin/hipcc $F -c scripts/micro/qgemm_probe.hip -o sgpt_amd/lib/qgemm_probe.o && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -o sgpt_amd/lib/qgemm_probe.bin sgpt_amd/lib/qgemm_probe.o sgpt_amd/lib/qgemm.o sgpt_amd/lib/elementwise.o
  exit $?
fi
mkdir -p gpurun_out
./sgpt_amd/lib/qgemm_probe.bin ${MS:-32 64 96 160 352 512 768 1024 1536 2304 2816 3584} 2>&1 | tee gpurun_out/qprobe.txt
