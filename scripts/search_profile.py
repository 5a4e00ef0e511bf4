#!/usr/bin/env python3
"""cProfile of DenseRetrievalExactSearch.search on the text API (word-level fast tokenizer, SGPT-125M shape): where the host
time of a multi-chunk search goes."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tokenizers import Tokenizer, models, pre_tokenizers
from transformers import PreTrainedTokenizerFast
from sgpt_amd import SGPTConfig, SGPTModel, synthetic_weights
from sgpt_amd.beir import CustomEmbedder, DenseRetrievalExactSearch
words = [f"w{i}" for i in range(5000)] + ["[", "]", "{", "}", "[UNK]"]
tk = Tokenizer(models.WordLevel({w: i for i, w in enumerate(words)}, unk_token="[UNK]"))
tk.pre_tokenizer = pre_tokenizers.Whitespace()
tok = PreTrainedTokenizerFast(tokenizer_object=tk, unk_token="[UNK]", eos_token="[UNK]")
tok.pad_token = tok.eos_token
cfg = SGPTConfig()
m = SGPTModel(cfg, synthetic_weights(cfg, seed=1), device="cuda:0", dtype="f16")
os.chdir("/tmp")
emb = CustomEmbedder(model_name="synthetic/125m", model=m, tokenizer=tok, method="weightedmean", specb=True, maxseqlen=128, dataset="unit")
rng = np.random.default_rng(4)
mk = lambda n, lo, hi: [" ".join(f"w{j}" for j in rng.integers(0, 5000, size=int(rng.integers(lo, hi)))) for _ in range(n)]
N = int(os.environ.get("N", 60000))
corpus = {f"d{i}": {"title": "", "text": t} for i, t in enumerate(mk(N, 60, 140))}
queries = {f"q{i}": t for i, t in enumerate(mk(200, 4, 20))}
dres = DenseRetrievalExactSearch(emb, corpus_chunk_size=int(os.environ.get("CHUNK", 15000)), score_dtype=torch.float16)
dres.search(corpus, queries, 10, "cos_sim")
torch.cuda.synchronize(); t = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
dres.search(corpus, queries, 10, "cos_sim")
pr.disable(); torch.cuda.synchronize()
dt = time.perf_counter() - t
print(f"{N} documents: {dt:.3f} s = {N / dt:,.0f} documents/s end to end (text in, ranked ids out)")
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
