#!/usr/bin/env python3
"""profiles/rNN_models.jsonl: throughput of every SGPT size / operand mode (bench.py --model, short runs) WITH the parity
figure measured at that shape beside it ("a parity column").  Parity comes from the GPU suite's log
(SGPT_PARITY_LOG of tests/test_gpu_parity_cfg2.py and tests/test_gpu_parity_large.py): max |cos - ref| and max |normalised
embedding - ref| against the reference stack's fixtures, through the same kernels, in the same mode.
usage: models_table.py PARITY.jsonl OUT.jsonl"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = {"125m": "cfg2_125m_1024x128", "1.3b": "cfg3_neo13b_specb", "2.7b": "cfg_neo27b", "5.8b": "cfg4_gptj6b", "bloom-7b1": "cfg5_bloom7b1"}
# (1.3b / 2.7b f16: the model default there is the structural precise_qk rule; "f16-qk" rows time the plain projection,
#  "f16-x3" rows every operand as a hi + lo pair)
SPECS = ["125m f16", "125m bf16", "125m fp8mfma", "125m f16-x3", "1.3b f16", "1.3b f16-qk", "1.3b f16-x3", "1.3b bf16", "2.7b f16", "2.7b f16-qk",
         "2.7b f16-x3", "5.8b f16", "5.8b bf16", "5.8b fp8mfma", "bloom-7b1 f16", "bloom-7b1 bf16", "bloom-7b1 fp8", "bloom-7b1 fp8mfma"]
parity = {}
for ln in open(sys.argv[1]):
    d = json.loads(ln)
    if "dtype" in d:                 # (the default-mode search() lines carry score_function / scorer instead)
        parity[(d["case"], d["dtype"])] = d
with open(sys.argv[2], "w") as out:
    for spec in SPECS:
        model, dtype = spec.split()
        ch = "4096" if model == "125m" else "1024"
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--model", model, "--dtype", dtype.split("-")[0],
                            "--precise-qk", "off" if dtype.endswith("-qk") else "auto",
                            "--precision", "x3" if dtype.endswith("-x3") else ("plain" if dtype.endswith("-qk") else "default"),
                            "--steps", "3", "--warmup", "1",
                            "--chunk", ch, "--no-cpu-baseline", "--no-1m", "--no-varlen", "--no-modes"], capture_output=True, text=True, timeout=900)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if not line:
            print(spec, "FAILED", r.stderr[-300:])
            continue
        b = json.loads(line[-1])
        fx = FIXTURE[model]
        p = parity.get((fx, dtype)) if fx else None
        row = {"model": model, "dtype": dtype, "sentences_per_s": b["value"], "gemm_tflops": b["roofline"]["achieved"],
               "end_to_end_frac_of_16bit_mfma_roofline": b["roofline"]["end_to_end_frac_of_mfma_roofline"],
               "parity_fixture": fx,
               "parity_max_abs_cos": None if p is None else p["max_abs_cos"],
               "parity_max_abs_normalised_emb": None if p is None else p["max_abs_norm_emb"],
               "precise_qk": b["config"]["workload"].split("precise_qk=")[1].split(" ")[0].rstrip(",") if "precise_qk=" in b["config"]["workload"] else None,
               "parity_note": None if p is not None else "no parity-log line for this mode"}
        out.write(json.dumps(row) + "\n")
        print(json.dumps(row))
