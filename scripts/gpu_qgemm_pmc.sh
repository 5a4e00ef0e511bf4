#!/bin/bash
# SQ counter passes over the query-sized projection kernels (scripts/micro/qgemm_probe.hip at M token rows): where do the wave cycles go?
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/gpurun_out
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_MISC" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  rm -rf /tmp/qpmc$i
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/qpmc$i -o pmc -- $R/sgpt_amd/lib/qgemm_probe.bin ${MS:-32 2304} > $R/gpurun_out/qgemm_pmc_$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $R
python scripts/pmc_summary.py /tmp/qpmc1/pmc_results.db /tmp/qpmc2/pmc_results.db /tmp/qpmc3/pmc_results.db 2>&1 | grep -E "^kernel|qgemm" | cut -c1-300 | tee gpurun_out/qgemm_pmc_summary.csv
