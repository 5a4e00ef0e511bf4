#!/bin/bash
# PMC passes (own runs, kernel-trace only): HBM read / write bytes and MFMA busy per kernel.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $C | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmc_$tag; mkdir -p $R/gpurun_out/pmc_$tag
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/pmc_$tag -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-1m --no-varlen --no-modes > $R/gpurun_out/pmc_$tag.log 2>&1
  echo "$tag rc=$?"
done
cd $R
python scripts/pmc_summary.py gpurun_out/pmc_FETCH_SIZE/pmc_results.db gpurun_out/pmc_WRITE_SIZE/pmc_results.db "gpurun_out/pmc_SQ_VALU_MFMA_BUSY_CYCLES/pmc_results.db" | tee gpurun_out/pmc_summary.csv
python scripts/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE/pmc_results.db gpurun_out/pmc_WRITE_SIZE/pmc_results.db gpurun_out/pmc_SQ_VALU_MFMA_BUSY_CYCLES/pmc_results.db > gpurun_out/pmc_traffic.json || echo "pmc_traffic failed"
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_SQ_VALU_MFMA_BUSY_CYCLES
