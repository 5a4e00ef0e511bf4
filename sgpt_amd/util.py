"""Drop-in for the scoring helpers the reference imports
(sentence_transformers/util.py:24-91,197-258 == beir.util.cos_sim/dot_score used at
biencoder/beir/custommodels/exact_search.py:9,27,96-98), computed by the HIP scorer."""
from typing import Callable, List

import numpy as np
import torch

from .runtime import get_context


def _ctx_for(*xs):
    for x in xs:
        if isinstance(x, torch.Tensor) and x.is_cuda:
            return get_context(x.device), x.device
    return get_context(), torch.device("cpu")


def _wrap(a):
    if not isinstance(a, torch.Tensor):
        a = torch.tensor(np.asarray(a))
    if a.dim() == 1:
        a = a.unsqueeze(0)
    return a


def _as_tensor(a):
    return a if isinstance(a, torch.Tensor) else torch.tensor(np.asarray(a))


def normalize_embeddings(embeddings: torch.Tensor) -> torch.Tensor:
    """util.normalize_embeddings (util.py:66-70): rows scaled to unit L2 norm."""
    ctx, home = _ctx_for(embeddings)
    out = ctx.l2_normalize(_wrap(embeddings))
    return out if home.type == "cuda" else out.cpu()


def cos_sim(a, b) -> torch.Tensor:
    """util.cos_sim (util.py:24-43): res[i][j] = cos(a[i], b[j]); exact-fp32 MFMA."""
    ctx, home = _ctx_for(a, b)
    out = ctx.scores(ctx.l2_normalize(_wrap(a)), ctx.l2_normalize(_wrap(b)))
    return out if home.type == "cuda" else out.cpu()


pytorch_cos_sim = cos_sim


def dot_score(a, b) -> torch.Tensor:
    """util.dot_score (util.py:46-63)."""
    ctx, home = _ctx_for(a, b)
    out = ctx.scores(ctx._dev_f32(_wrap(a)), ctx._dev_f32(_wrap(b)))
    return out if home.type == "cuda" else out.cpu()


def pairwise_dot_score(a, b) -> torch.Tensor:
    """util.pairwise_dot_score (util.py:66-76): res[i] = dot(a[i], b[i])."""
    ctx, home = _ctx_for(a, b)
    a, b = _as_tensor(a), _as_tensor(b)
    lead = a.shape[:-1]
    out = ctx.pairwise_scores(a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1]), cosine=False).reshape(lead)
    return out if home.type == "cuda" else out.cpu()


def pairwise_cos_sim(a, b) -> torch.Tensor:
    """util.pairwise_cos_sim (util.py:79-91): res[i] = cos_sim(a[i], b[i]) -- the product sum of the normalised rows."""
    ctx, home = _ctx_for(a, b)
    a, b = _as_tensor(a), _as_tensor(b)
    out = ctx.pairwise_scores(a, b, cosine=True)         # (normalize_embeddings is dim=1: matrices, as in the reference)
    return out if home.type == "cuda" else out.cpu()


def semantic_search(query_embeddings, corpus_embeddings, query_chunk_size: int = 100,
                    corpus_chunk_size: int = 500000, top_k: int = 10,
                    score_function: Callable = cos_sim) -> List[List[dict]]:
    """util.semantic_search (util.py:197-258).  Same result contract (per query a list of
    {'corpus_id','score'} sorted by decreasing score); the chunk loops collapse into one fused
    score + running-top-k pass on the GPU when score_function is cos_sim / dot_score."""
    if isinstance(query_embeddings, list):
        query_embeddings = torch.stack(query_embeddings)
    if isinstance(corpus_embeddings, list):
        corpus_embeddings = torch.stack(corpus_embeddings)
    q, c = _wrap(query_embeddings), _wrap(corpus_embeddings)
    ctx, _ = _ctx_for(c, q)
    if score_function in (cos_sim, pytorch_cos_sim):
        qd, cd = ctx.l2_normalize(q), ctx.l2_normalize(c)
    elif score_function is dot_score:
        qd, cd = ctx._dev_f32(q), ctx._dev_f32(c)
    else:  # arbitrary callable: materialise its scores chunk by chunk, top-k on the GPU
        out = [[] for _ in range(len(q))]
        for qs in range(0, len(q), query_chunk_size):
            for cs in range(0, len(c), corpus_chunk_size):
                sc = score_function(q[qs:qs + query_chunk_size], c[cs:cs + corpus_chunk_size])
                v, i = ctx.topk(sc, min(top_k, sc.shape[1]), idx_base=cs)
                for r, (vv, ii) in enumerate(zip(v.cpu().tolist(), i.cpu().tolist())):
                    out[qs + r] += [{"corpus_id": j, "score": s} for j, s in zip(ii, vv)]
        return [sorted(o, key=lambda x: x["score"], reverse=True)[:top_k] for o in out]
    k = min(top_k, len(c))
    val, idx, n = ctx.score_topk(qd, cd, k)
    val, idx = val.cpu().tolist(), idx.cpu().tolist()
    return [[{"corpus_id": j, "score": s} for j, s in zip(ii[:n], vv[:n])] for vv, ii in zip(val, idx)]
