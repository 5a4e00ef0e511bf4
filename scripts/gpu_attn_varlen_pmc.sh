#!/bin/bash
# SQ counter passes over encode-only steps of variable-length (and fixed-128) documents: where do the attention launches of the
# short calls spend their wave cycles?  (scripts/varlen_profile.py; one kernel name per call shape with the fitted blocks)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/gpurun_out
cd /tmp
: > $R/gpurun_out/attn_varlen_pmc.csv
for L in ${VPROF_LENS:-var fixed}; do
  i=0
  for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rm -rf /tmp/vpmc_${L}_$i
    LENS=$L STEPS=2 timeout 600 rocprofv3 --pmc $C --kernel-trace -d /tmp/vpmc_${L}_$i -o pmc -- python $R/scripts/varlen_profile.py > $R/gpurun_out/attn_varlen_pmc_${L}_$i.log 2>&1
    echo "LENS=$L pass $i rc=$?"
  done
  echo "# LENS=$L" >> $R/gpurun_out/attn_varlen_pmc.csv
  python $R/scripts/pmc_summary.py /tmp/vpmc_${L}_1/pmc_results.db /tmp/vpmc_${L}_2/pmc_results.db /tmp/vpmc_${L}_3/pmc_results.db | grep -E "^kernel|attn16" >> $R/gpurun_out/attn_varlen_pmc.csv
done
cat $R/gpurun_out/attn_varlen_pmc.csv | cut -c40-200
