#!/bin/bash
# Round 4, third GPU call: the reworked scorer (exactness tests, shard timings, drift), default-mode parity of the GPT-Neo fixtures.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_search.py tests/test_gpu_dist.py -q --timeout=600 ) > gpurun_out/r4c_tests.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/r4c_tests.log | cut -c1-300
( for n in 125000 250000 500000 1000000; do N=$n python scripts/score_bench.py; done; for nq in 128 64 16; do NQ=$nq python scripts/score_bench.py; done; NQ=16 N=125000 python scripts/score_bench.py; NQ=128 N=125000 python scripts/score_bench.py; for dr in 0.1 0.5 0.9; do echo -n "DRIFT=$dr "; DRIFT=$dr python scripts/score_bench.py; done; echo -n "K=101 "; K=101 python scripts/score_bench.py;  echo -n "K=1001 "; K=1001 python scripts/score_bench.py ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4c_score_bench.txt
rm -f gpurun_out/parity.jsonl
( SGPT_PARITY_LOG=$R/gpurun_out/parity.jsonl timeout 1500 python -m pytest tests/test_gpu_parity_large.py -q --timeout=900 -rA -k "neo13b_specb-f16] or neo27b-f16] or outlier" ) > gpurun_out/r4c_parity_large.log 2>&1; echo "parity rc=$?"
grep -E "^(cfg|outlier)" gpurun_out/r4c_parity_large.log | cut -c1-330; grep -E "passed|failed" gpurun_out/r4c_parity_large.log | tail -1
( timeout 900 python bench.py --steps 10 --no-cpu-baseline --no-varlen ) 2>/dev/null | grep '^{' > gpurun_out/r4c_bench.json; python -c "
import json; d=json.load(open('gpurun_out/r4c_bench.json')); print(d['value'], d['queries_per_sec_at_1M_corpus'], json.dumps(d['projected_8gpu']), json.dumps(d['queries_per_sec_at_1M_corpus_k1001']), d['queries_per_sec_at_1M_corpus_incl_query_encode_by_nq'])"
