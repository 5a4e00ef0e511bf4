#!/bin/bash
# Round-2 artefacts in one gpurun call: tests, default bench line, rocprofv3 kernel stats, PMC traffic, GEMM A/B,
# per-tile timeline, clock trace, other model sizes, fp8 numbers.  Everything lands in gpurun_out/ (copied to profiles/).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -q --timeout=600 ) > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest.log
# clock trace beside the default bench run
( for i in $(seq 1 400); do rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/clock_trace.txt 2>&1 &
CLK=$!
( timeout 900 python bench.py ) > gpurun_out/bench_full.log 2>&1; echo "bench rc=$?"; grep '^{' gpurun_out/bench_full.log > gpurun_out/bench_n1.json; cut -c1-500 gpurun_out/bench_n1.json
kill $CLK 2>/dev/null
BENCH_STEPS=3 bash scripts/gpu_prof.sh > gpurun_out/prof.log 2>&1; head -16 gpurun_out/prof_summary.csv
bash scripts/gpu_pmc.sh > gpurun_out/pmc.log 2>&1; tail -3 gpurun_out/pmc.log; head -c 600 gpurun_out/pmc_traffic.json
( VARIANTS=0,1 DTYPES=f16,bf16 ROUNDS=3 python scripts/gemm_bench.py; VARIANTS=0 DTYPES=f16 SKEWS=0,6000,24000 ROUNDS=2 python scripts/gemm_bench.py ) 2>&1 | grep -v amdgpu.ids > gpurun_out/gemm_ab.txt; grep "block total" gpurun_out/gemm_ab.txt
SGPT_GEMM_DBG=1 python scripts/gemm_dbg.py 2>&1 | grep -v amdgpu.ids > gpurun_out/gemm_timeline.txt
( ./gpurun_in/f8_probe; python scripts/gemm_fp8_bench.py ) 2>&1 | grep -v amdgpu.ids > gpurun_out/fp8_gemm.txt; tail -8 gpurun_out/fp8_gemm.txt
bash scripts/score_prof.sh > gpurun_out/score_prof.log 2>&1; head -8 gpurun_out/score_prof_summary.csv 2>/dev/null
( python -m pytest tests/test_gpu_fp8mfma.py -m gpu -q -s ) 2>&1 | grep -v amdgpu.ids > gpurun_out/fp8_tests.txt; tail -3 gpurun_out/fp8_tests.txt
# query-sized batches: 64-row scorer tile A/B, encode latency (default / low-latency mode), time split, per-kernel profile
( bash scripts/score_small_ab.sh; for kg in 1 2; do for nq in 1 16 128; do echo -n "SGPT_KGROUPS=$kg "; SGPT_KGROUPS=$kg NQ=$nq python scripts/small_batch_profile.py 2>&1 | grep "per encode"; done; done
  python scripts/query_side_breakdown.py ) 2>&1 | grep -v amdgpu.ids > gpurun_out/query_side.txt; tail -12 gpurun_out/query_side.txt
( cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/small -o small -- python $R/scripts/small_batch_profile.py ) > /dev/null 2>&1
python scripts/prof_summary.py gpurun_out/small/small_results.db 14 > gpurun_out/small_batch_kernel_stats.csv 2>&1; rm -rf gpurun_out/small
: > gpurun_out/models.jsonl
for spec in "125m bf16" "125m fp8mfma" "1.3b f16" "2.7b f16" "5.8b f16" "5.8b fp8mfma" "bloom-7b1 f16" "bloom-7b1 bf16" "bloom-7b1 fp8" "bloom-7b1 fp8mfma"; do
  set -- $spec
  ch=4096; [ "$1" != "125m" ] && ch=1024
  timeout 900 python bench.py --model $1 --dtype $2 --steps 3 --warmup 1 --chunk $ch --no-cpu-baseline --no-1m --no-varlen 2>&1 | grep '^{' >> gpurun_out/models.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/models.jsonl"):
    d = json.loads(l)
    print(d["config"]["workload"][:40], d["dtype"], d["value"], d["roofline"]["achieved"], d["roofline"]["end_to_end_frac_of_mfma_roofline"])
PY
