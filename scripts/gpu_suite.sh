#!/bin/bash
# One gpurun call, named stages (replaces the per-round gpu_rN_{first..last}.sh one-shots):
#   bash scripts/gpu_suite.sh tests smoke bench yardstick prof pmc shapes score query models seq
# Every stage writes under gpurun_out/ (merged back by gpurun); copy what is to be judged into profiles/rNN_*.
#   tests      whole GPU suite with the parity log            -> pytest_full.log, parity.jsonl, parity_numbers.txt
#   smoke      __graft_entry__.smoke()
#   bench      default bench line                              -> bench_n1.json
#   yardstick  vendor GEMM vs gemm256d on the five shapes      -> hipblaslt_yardstick.txt
#   prof       rocprofv3 --kernel-trace --stats of the bench   -> prof_summary.csv
#   pmc        separate --pmc passes (FETCH, WRITE, MFMA busy) -> pmc_summary.csv, pmc_traffic.json
#   shapes     gemm_bench.py over the five launch shapes       -> gemm_shapes.txt
#   score      scorer passes (shards, nq, drift, k)            -> score_bench.txt
#   query      query-sized encode latency + time split         -> query_side.txt
#   models     every SGPT size / mode with its parity column   -> models.jsonl
#   seq        seq 300 / 512 bench lines                       -> seq_lengths.txt
# LIBS="a b" (library tags built with SGPT_LIB_TAG): stages `ab_bench` / `ab_shapes` alternate the default library and each tag
# on this box (same-box A/B: boxes differ by +-3 %).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
libpath() { if [ "$1" = "default" ]; then echo ""; else echo "$R/sgpt_amd/lib/libsgpt_hip_$1.so"; fi; }
for stage in "$@"; do
  echo "=== stage $stage ($(date +%T))"
  case $stage in
    tests)
      rm -f gpurun_out/parity.jsonl
      ( SGPT_PARITY_LOG=$R/gpurun_out/parity.jsonl timeout 1800 python -m pytest tests -m gpu -q --timeout=900 -rA ${PYTEST_ARGS} ) > gpurun_out/pytest_full.log 2>&1
      echo "pytest rc=$?"; tail -3 gpurun_out/pytest_full.log
      grep -E "^(cfg|outlier|f16 range|default-mode)" gpurun_out/pytest_full.log | cut -c1-700 > gpurun_out/parity_numbers.txt
      grep -E "passed|failed" gpurun_out/pytest_full.log | tail -1 >> gpurun_out/parity_numbers.txt
      grep -E "^(FAILED|ERROR)" gpurun_out/pytest_full.log | cut -c1-300 ;;
    smoke) python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -6 ;;
    bench)
      ( timeout 900 python bench.py ${BENCH_ARGS} ) > gpurun_out/bench_full.log 2>&1; echo "bench rc=$?"
      grep '^{' gpurun_out/bench_full.log > gpurun_out/bench_n1.json; cut -c1-900 gpurun_out/bench_n1.json
      python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/bench_n1.json"))
    keys = ("value", "ms_per_step", "value_incl_host_pack_and_h2d", "queries_per_sec_at_1M_corpus", "queries_per_sec_at_1M_corpus_fp32_scorer",
            "queries_per_sec_at_1M_corpus_incl_query_encode_by_nq", "queries_per_sec_at_1M_corpus_k1001", "projected_8gpu", "precision_modes", "varlen")
    for k in keys:
        print(k, "=", json.dumps(d.get(k))[:900])
    print("roofline =", json.dumps({k: v for k, v in d["roofline"].items() if k not in ("kernel", "traffic_source", "mfma_busy_note")}))
    print("cpu parity =", json.dumps((d.get("cpu_baseline") or {}).get("parity"))[:600])
except Exception as e:
    print("bench line unreadable:", e)
PY
      ;;
    yardstick) ( timeout 600 python scripts/hipblaslt_yardstick.py ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/hipblaslt_yardstick.txt ;;
    prof) BENCH_STEPS=3 bash scripts/gpu_prof.sh > gpurun_out/prof.log 2>&1; head -16 gpurun_out/prof_summary.csv ;;
    pmc) bash scripts/gpu_pmc.sh > gpurun_out/pmc.log 2>&1; tail -3 gpurun_out/pmc.log; head -c 300 gpurun_out/pmc_traffic.json; echo; grep -E "busy|clock|error|durations" gpurun_out/pmc_traffic.json ;;
    shapes) ( VARIANTS=0 DTYPES=f16,bf16 ROUNDS=3 python scripts/gemm_bench.py ) 2>&1 | grep -v amdgpu.ids > gpurun_out/gemm_shapes.txt; grep -E "block total|f16 mfma16" gpurun_out/gemm_shapes.txt ;;
    score)
      ( for n in 1000000 500000 250000 125000; do N=$n python scripts/score_bench.py; done
        for nq in 128 64 16; do NQ=$nq python scripts/score_bench.py; done
        for dr in 0.1 0.6 0.9; do echo -n "DRIFT=$dr "; DRIFT=$dr python scripts/score_bench.py; done
        echo -n "K=101 "; K=101 python scripts/score_bench.py; echo -n "K=1001 "; K=1001 python scripts/score_bench.py ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/score_bench.txt ;;
    query) ( for nq in 1 16 128; do LL=0 NQ=$nq python scripts/small_batch_profile.py 2>&1 | grep "per encode"; done; python scripts/query_side_breakdown.py ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/query_side.txt ;;
    models) python scripts/models_table.py gpurun_out/parity.jsonl gpurun_out/models.jsonl > gpurun_out/models.log 2>&1; cut -c1-300 gpurun_out/models.jsonl ;;
    seq)
      ( for s in 300 512; do timeout 300 python bench.py --seq $s --call $((131072 / s)) --chunk $((4 * (131072 / s))) --steps 4 --warmup 1 --no-cpu-baseline --no-1m --no-varlen --no-modes 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seq', $s, d['value'], 'sent/s', d['roofline']['end_to_end_frac_of_mfma_roofline'])"; done ) | tee gpurun_out/seq_lengths.txt ;;
    ab_bench)
      : > gpurun_out/ab_bench.txt
      for rnd in 1 2 3; do for tag in default ${LIBS}; do
        L=$(libpath $tag)
        v=$( SGPT_HIP_LIB=$L timeout 600 python bench.py --steps ${AB_STEPS:-10} --warmup 3 --no-cpu-baseline --no-1m --no-varlen --no-modes 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], 'sent/s  gemm', r['achieved'], 'TF  launches', r['launches'], 'avg_ms', r['avg_launch_ms'])" )
        echo "round $rnd lib $tag: $v" | tee -a gpurun_out/ab_bench.txt
      done; done ;;
    ab_score)
      : > gpurun_out/ab_score.txt
      for rnd in 1 2 3; do for tag in default ${LIBS}; do
        L=$(libpath $tag)
        for n in 125000 1000000; do echo -n "round $rnd lib $tag: " | tee -a gpurun_out/ab_score.txt; SGPT_HIP_LIB=$L N=$n python scripts/score_bench.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/ab_score.txt; done
      done; done ;;
    ab_varlen)
      : > gpurun_out/ab_varlen.txt
      for rnd in 1 2 3; do for mode in "" "--equal-calls"; do
        v=$( timeout 600 python bench.py ${VARLEN_BENCH_ARGS:---steps 9 --warmup 2 --no-1m} --no-cpu-baseline --no-modes $mode 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); v=d['varlen']; print(v['sentences_per_s'], 'sent/s', v['end_to_end_frac_of_mfma_roofline'], v['rows_per_call'], '| fixed-128', d['value'])" )
        echo "round $rnd calls ${mode:-round-aware}: $v" | tee -a gpurun_out/ab_varlen.txt
      done; done ;;
    ab_varlen_lib)
      : > gpurun_out/ab_varlen_lib.txt
      for rnd in 1 2 3; do for tag in default ${LIBS}; do
        v=$( SGPT_HIP_LIB=$(libpath $tag) timeout 600 python bench.py ${VARLEN_BENCH_ARGS:---steps 9 --warmup 2 --no-1m} --no-cpu-baseline --no-modes 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); v=d['varlen']; print(v['sentences_per_s'], 'sent/s', v['end_to_end_frac_of_mfma_roofline'], v['rows_per_call'], '| fixed-128', d['value'])" )
        echo "round $rnd lib $tag: $v" | tee -a gpurun_out/ab_varlen_lib.txt
      done; done ;;
    varlen_prof)
      for L in ${VPROF_LENS:-var fixed}; do
        rm -rf gpurun_out/vprof_$L; mkdir -p gpurun_out/vprof_$L
        ( cd /tmp && LENS=$L timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/vprof_$L -o trace -- python $R/scripts/varlen_profile.py ) > gpurun_out/vprof_$L.log 2>&1
        grep "LENS=" gpurun_out/vprof_$L.log
        python scripts/prof_summary.py gpurun_out/vprof_$L/trace_results.db 14 > gpurun_out/varlen_prof_$L.csv
        python scripts/prof_by_grid.py gpurun_out/vprof_$L/trace_results.db attn16 | tee gpurun_out/varlen_attn_by_call_$L.csv; rm -rf gpurun_out/vprof_$L
        head -14 gpurun_out/varlen_prof_$L.csv | cut -c1-160
      done ;;
    ab_score_prof)
      for tag in default ${LIBS}; do
        L=$(libpath $tag); rm -rf gpurun_out/sprof_$tag; mkdir -p gpurun_out/sprof_$tag
        ( cd /tmp && SGPT_HIP_LIB=$L N=${SPROF_N:-125000} REPS=20 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/sprof_$tag -o trace -- python $R/scripts/score_bench.py ) > gpurun_out/sprof_$tag.log 2>&1
        echo "--- lib $tag: $(grep 'per pass' gpurun_out/sprof_$tag.log)"
        python scripts/prof_summary.py gpurun_out/sprof_$tag/trace_results.db 12 | cut -c1-170 | tee gpurun_out/score_prof_$tag.csv; rm -rf gpurun_out/sprof_$tag
      done ;;
    select_probe) ( python scripts/select_probe.py ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/select_probe.txt ;;
    graph_probe) ( python scripts/score_graph_probe.py ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/score_graph_probe.txt ;;
    probe) ( python scripts/score_shape_probe.py ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/score_shape_probe.txt ;;
    ab_shapes)
      : > gpurun_out/ab_shapes.txt
      for tag in default ${LIBS}; do
        L=$(libpath $tag); echo "--- lib $tag" | tee -a gpurun_out/ab_shapes.txt
        ( SGPT_HIP_LIB=$L VARIANTS=0 DTYPES=f16 ROUNDS=3 python scripts/gemm_bench.py ) 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/ab_shapes.txt | grep -E "block total"
      done ;;
    *) echo "unknown stage $stage" ;;
  esac
done
echo "=== done ($(date +%T))"
