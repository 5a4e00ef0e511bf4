#!/bin/bash
# builds (here, cross-compiled) and runs (on the GPU box) the qgemm probe: SHAPES="M,N,K,epi,ln ..." bash scripts/gpu_qprobe.sh [build|run]
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
C=sgpt_amd/csrc
if [ "$1" = "build" ]; then
  F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -I $C"
  /opt/rocm/bin/hipcc $F -DSGPT_QSTAMPS -c $C/qgemm.hip -o sgpt_amd/lib/qgemm_stamps.o && \
  /opt/rocm/bin/hipcc $F -c scripts/micro/qgemm_probe.hip -o sgpt_amd/lib/qgemm_probe.o && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -o sgpt_amd/lib/qgemm_probe.bin sgpt_amd/lib/qgemm_probe.o sgpt_amd/lib/qgemm_stamps.o sgpt_amd/lib/elementwise.o
  exit $?
fi
mkdir -p gpurun_out
for sh in ${SHAPES:-32,2304,768,7,1 32,3072,768,1,1 32,768,768,2,0 32,768,3072,2,0 352,2304,768,7,1 352,3072,768,1,1 352,768,768,2,0 352,768,3072,2,0}; do
  ./sgpt_amd/lib/qgemm_probe.bin ${sh//,/ } 2>&1
done | tee gpurun_out/qprobe.txt
