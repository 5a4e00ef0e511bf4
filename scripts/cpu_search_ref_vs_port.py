#!/usr/bin/env python3
"""Build-container only (needs /root/reference): the reference's own DenseRetrievalExactSearch.search
(biencoder/beir/custommodels/exact_search.py, loaded from its file with a stub `beir` as tests/golden/make_golden.py does)
timed beside the port bench.py uses for its `cpu_baseline.search` leg (oracle.exact_search, backend='torch'), on the same inputs:
nq = 128 queries against 100k x 768 fp32 documents, chunks of 50k, cos_sim, top-10.  The reference file cannot travel to the
GPU box, so bench.py reports the port (`kind: "port"`); this script shows what the port stands in for.

    python scripts/cpu_search_ref_vs_port.py > profiles/r03_cpu_search_ref_vs_port.txt
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from oracle import sgpt_oracle as O  # noqa: E402
import make_golden as G  # noqa: E402

U = G.load_ref_util()
ES = G.load_ref_exact_search(U)
rng = np.random.default_rng(11)
nd, nq, d, topk = 100_000, 128, 768, 10
cemb = rng.standard_normal((nd, d)).astype(np.float32)
cemb[:, :3] *= 30.0
qemb = rng.standard_normal((nq, d)).astype(np.float32)
qemb[:, :3] *= 30.0
corpus = {f"d{i}": {"title": "", "text": "x"} for i in range(nd)}
queries = {f"q{i}": "q" for i in range(nq)}


class Fake:
    def encode_queries(self, qq, batch_size, **kw):
        return torch.from_numpy(qemb[[int(q[1:]) for q, _ in qq]])

    def encode_corpus(self, cs, batch_size, **kw):
        return torch.from_numpy(cemb[[int(c[1:]) for c, _ in cs]])


def med(f, n=3):
    ts = []
    for _ in range(n):
        t = time.perf_counter()
        r = f()
        ts.append(time.perf_counter() - t)
    return float(np.median(ts)), r


torch.set_num_threads(os.cpu_count() or 1)
t_ref, res = med(lambda: ES.DenseRetrievalExactSearch(Fake(), batch_size=128, corpus_chunk_size=50_000).search(corpus, queries, topk, "cos_sim"))
cids = sorted(corpus, key=lambda k: len(corpus[k].get("title", "") + corpus[k].get("text", "")), reverse=True)
order = [int(c[1:]) for c in cids]
t_port, ores = med(lambda: O.exact_search(qemb, list(queries), cemb[order], cids, topk, "cos_sim", chunk_size=50_000, backend="torch"))
t_np, nres = med(lambda: O.exact_search(qemb, list(queries), cemb[order], cids, topk, "cos_sim", chunk_size=50_000))
same = all(set(res[q]) == set(ores[q]) == set(nres[q]) for q in queries)
print(f"cpus: {os.cpu_count()}  torch threads: {torch.get_num_threads()}")
print(f"reference exact_search.py (torch CPU, incl. its fake-encoder lookups): {t_ref:.3f} s -> {nq / t_ref:.1f} queries/s")
print(f"port, torch CPU primitives (oracle.exact_search backend='torch'; bench.py): {t_port:.3f} s -> {nq / t_port:.1f} queries/s")
print(f"port, numpy lines (oracle.exact_search, the parity checker):                {t_np:.3f} s -> {nq / t_np:.1f} queries/s")
print(f"identical hit sets for all {nq} queries: {same}")
