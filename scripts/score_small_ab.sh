for nq in 16 64 128; do NQ=$nq python scripts/score_bench.py; NQ=$nq SGPT_SCORE_NO64=1 python scripts/score_bench.py; done 2>&1 | grep -v amdgpu
