#!/bin/bash
# SQ counter passes over a seq-512 encode step: where do attn16_lds_kernel's wave cycles go?
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/gpurun_out
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+" | sort -u > $R/gpurun_out/sq_counters.txt
wc -l $R/gpurun_out/sq_counters.txt
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rm -rf /tmp/apmc$i
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d /tmp/apmc$i -o pmc -- python $R/bench.py --steps 1 --warmup 1 --seq ${SEQ:-512} --chunk ${CHUNK:-1024} --call ${CALL:-256} --no-cpu-baseline --no-1m --no-varlen > $R/gpurun_out/attn_pmc_$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $R
python scripts/pmc_summary.py /tmp/apmc1/pmc_results.db /tmp/apmc2/pmc_results.db | grep -E "^kernel|attn16|layernorm" | tee gpurun_out/attn_pmc_summary.csv
