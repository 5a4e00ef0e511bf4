"""Drop-in adapters for biencoder/beir of the reference.

  CustomEmbedder              <- biencoder/beir/beir_dense_retriever.py:106-348
  DenseRetrievalExactSearch   <- biencoder/beir/custommodels/exact_search.py:21-134
  SentenceBERTBOSEOS          <- biencoder/beir/custommodels/sentence_bert_asym.py:21-79

Same constructor arguments, method names, input conventions, return types and error
behaviour; the device work goes through the HIP layer (sgpt_amd.model / sgpt_amd.runtime)
and embeddings never leave the GPU between encode and top-k."""
import logging
import os
import pathlib
import pickle
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from .model import SGPTModel
from .runtime import get_context
from .tokenization import TextPipeline, load_tokenizer

logger = logging.getLogger(__name__)

SCORE_SPLIT_CREST = 10.0      # normalised embedding rows: crest factor above which a 16-bit corpus is kept as hi + lo pairs
SINGLE_LAYER_METHODS = ("mean", "weightedmean", "lasttoken")
ALL_LAYER_METHODS = {"meanmean": "mean", "lasttokenmean": "lasttoken"}


class CustomEmbedder:
    """Raw-HF-path embedder of the reference (beir_dense_retriever.py:106-348): per-text
    tokenise/truncate/specb brackets on the host, then forward + pooling on the GPU in one C call
    (the reference copies all L+1 hidden states to the CPU and pools there, :221-304)."""

    def __init__(self, model_name="EleutherAI/gpt-neo-1.3B", batch_size=250, device="cuda:0", save_emb=False,
                 reinit=False, layeridx=-1, method="mean", dataset="scifact", specb=False, maxseqlen=None,
                 model: Optional[SGPTModel] = None, tokenizer=None, dtype="f16", **kwargs):
        if reinit:
            raise NotImplementedError("reinit (random re-initialisation ablation) is not part of the hot path")
        self.device = torch.device(device)
        self.model = model if model is not None else SGPTModel.from_pretrained(model_name, device=device, dtype=dtype)
        self.tokenizer = tokenizer if tokenizer is not None else load_tokenizer(model_name)
        self.max_token_len = maxseqlen if maxseqlen else self.model.cfg.max_position_embeddings   # :128
        self.pipe = TextPipeline(self.tokenizer, self.max_token_len, specb=specb)
        self.max_token_len = self.pipe.max_token_len
        self.batch_size = batch_size
        self.save_emb = save_emb
        self.layeridx = layeridx
        self.method = method
        self.specb = specb
        if method not in SINGLE_LAYER_METHODS and method not in ALL_LAYER_METHODS:
            raise ValueError(f"unknown method {method}")   # 'poolout' needs a pooler GPT models do not have
        self.base_path = f"embeddings/{model_name.split('/')[-1]}/{self.method}/{dataset}"     # :155
        if save_emb:
            pathlib.Path(self.base_path).parent.mkdir(parents=True, exist_ok=True)

    # -- device leg --------------------------------------------------------------------------
    def tokenize(self, sentences: List[str], is_query: bool) -> List[List[int]]:
        """Host leg only (text -> truncated, bracketed id lists).  Separate from the device leg so that a caller can run
        it for the NEXT corpus chunk on a worker thread while the GPU encodes the current one (HF fast tokenizers
        release the GIL; DenseRetrievalExactSearch.search does exactly that)."""
        before = (self.pipe.docs_truncated, self.pipe.toks_truncated)
        seqs = self.pipe.batch(sentences, is_query)
        if self.pipe.docs_truncated > before[0]:
            logging.warning(f"Truncated {self.pipe.docs_truncated - before[0]} out of {len(sentences)} documents "
                            f"by {self.pipe.toks_truncated - before[1]} tokens.")                # :216-219
        return seqs

    def tokenize_corpus(self, corpus) -> List[List[int]]:
        return self.tokenize([t for (_, t) in self._corpus_texts(corpus)], False)

    def embed_device(self, sentences: List[str], is_query: bool, normalize: bool = False) -> torch.Tensor:
        return self.embed_ids_device(self.tokenize(sentences, is_query), normalize=normalize)

    def embed_ids_device(self, seqs: List[List[int]], normalize: bool = False) -> torch.Tensor:
        L = self.model.cfg.num_layers
        if abs(self.layeridx) > L + 1:
            raise ValueError(f"Layer Idx {self.layeridx} is larger than the {L + 1} hidden states")   # :234-235
        if self.method in SINGLE_LAYER_METHODS:
            return self.model.encode_ids(seqs, mode=self.method, normalize=normalize, layer_idx=self.layeridx)
        # meanmean / lasttokenmean (:243-257, 284-301): average of the per-layer pooled vectors over all
        # L+1 hidden states (equal token counts per layer make the two formulations identical) -- pooled on
        # the way through ONE forward (sgpt_encode_layers) instead of copying L+1 hidden states to the host
        acc = self.model.encode_ids_all_layers(seqs, mode=ALL_LAYER_METHODS[self.method])
        return get_context(self.model.device).l2_normalize(acc) if normalize else acc

    # -- reference surface -------------------------------------------------------------------
    def embed_batcher(self, texts: List[Tuple[int, str]], is_query, out_name=None, **kwargs) -> Dict:
        ids, sentences = zip(*texts)
        emb = self.embed_device(list(sentences), is_query).cpu().numpy()
        all_embeddings = {i: e for i, e in zip(ids, emb)}
        assert len(texts) == len(all_embeddings)
        if self.save_emb:
            pickle.dump(all_embeddings, open(out_name, "wb"))                                   # :311-312
        return all_embeddings

    def encode_queries(self, queries: List[Tuple[str, str]], batch_size: int = None, **kwargs) -> np.ndarray:
        path = f"{self.base_path}_queries.pickle"
        if os.path.exists(path):
            embeddings = pickle.load(open(path, "rb"))                                          # :319-321
        else:
            embeddings = self.embed_batcher(texts=queries, out_name=path, is_query=True, **kwargs)
        embeddings = np.array([embeddings[i] for (i, _) in queries])
        logger.info(f"Produced embeddings of shape {embeddings.shape}")
        return embeddings

    @staticmethod
    def _corpus_texts(corpus):
        # (title + " " + text).strip() (:341); docs without a title keep their text (appendix A.1 latent bug not copied)
        return [(i, (d["title"] + " " + d["text"]).strip() if "title" in d else d["text"].strip()) for (i, d) in corpus]

    def encode_corpus(self, corpus: List[Tuple[str, Dict[str, str]]], batch_size: int = None, batch_num="",
                      **kwargs) -> np.ndarray:
        path = f"{self.base_path}_corpus{batch_num}.pickle"
        if os.path.exists(path):
            embeddings = pickle.load(open(path, "rb"))                                          # :336-338
        else:
            embeddings = self.embed_batcher(texts=self._corpus_texts(corpus), out_name=path, is_query=False, **kwargs)
        embeddings = np.array([embeddings[i] for (i, _) in corpus])
        logger.info(f"Produced embeddings of shape {embeddings.shape}")
        return embeddings

    # device-resident variants used by DenseRetrievalExactSearch below (no D2H of embeddings)
    def encode_queries_device(self, queries, normalize=False, use_cache=True):
        """use_cache=False: never touch `{base_path}_queries.pickle` (the sharded search encodes a slice per rank)."""
        if use_cache and (os.path.exists(f"{self.base_path}_queries.pickle") or self.save_emb):
            emb = torch.from_numpy(self.encode_queries(queries)).to(self.model.device)
            return get_context(self.model.device).l2_normalize(emb) if normalize else emb
        return self.embed_device([q for (_, q) in queries], True, normalize=normalize)

    def uses_embedding_cache(self, batch_num="") -> bool:
        return os.path.exists(f"{self.base_path}_corpus{batch_num}.pickle") or self.save_emb

    def encode_corpus_device(self, corpus, batch_num="", normalize=False, seqs=None):
        """seqs: the chunk's id lists when the caller tokenised it ahead of time (tokenize_corpus)."""
        if self.uses_embedding_cache(batch_num):
            emb = torch.from_numpy(self.encode_corpus(corpus, batch_num=batch_num)).to(self.model.device)
            return get_context(self.model.device).l2_normalize(emb) if normalize else emb
        if seqs is None:
            seqs = self.tokenize_corpus(corpus)
        return self.embed_ids_device(seqs, normalize=normalize)


class DenseRetrievalExactSearch:
    """custommodels/exact_search.py:21-134 with the scoring / top-k / chunk merge on the GPU.

    Result contract kept bit-for-bit in structure: Dict[qid, Dict[doc_id, float]] holding, per
    query, the (k+1) best of {per-chunk top-(k+1) minus corpus_id == query_id} (:102-132)."""

    def __init__(self, model, batch_size: int = 128, corpus_chunk_size: int = 50000, score_dtype=torch.float32,
                 prefetch_tokenize: bool = True, ctx=None, group=None, distributed: Optional[bool] = None,
                 score_split: Optional[bool] = None, exact_scorer: str = "brute", **kwargs):
        self.model = model
        self.prefetch_tokenize = prefetch_tokenize
        self.batch_size = batch_size
        self.score_function_desc = {"cos_sim": "Cosine Similarity", "dot": "Dot Product"}
        self.corpus_chunk_size = corpus_chunk_size
        self.show_progress_bar = True
        self.convert_to_tensor = True
        self.score_dtype = score_dtype          # torch.float32: fp32 scores (below); torch.float16 / bfloat16: 16-bit corpus in HBM
        # score_dtype=torch.float32 (the default: what the reference computes, util.py:41-43 in fp32), how:
        #   "brute" (default): the exact-fp32 MFMA scorer over every pair (1/16 of the 16-bit rate; 68 k queries/s against 1 M rows);
        #   "refined" (cos_sim only): the 16-bit filtered scorer proposes k + head-room candidates per query, every candidate is
        #             re-scored in exact fp32, and a device-side check (worst case |s16 - s32| <= 1.1e-3 for unit rows) proves the
        #             fp32 top-k is among them -- else the chunk is redone by the exact pass, predicated on the device flag
        #             (sgpt_score_topk_refined).  Pays when the k-th best score has fewer than ~50 rivals within 2.5e-3 (well-spread
        #             embeddings, shallow k: up to 8x); concentrated cosine distributions -- SGPT's anisotropic embeddings on a large
        #             corpus, or the driver's k = 1001 -- trip the check and pay both passes, hence opt-in.  (Against the encode of
        #             the corpus chunk it scores, either scorer is < 0.1 % of a search() call.)
        if exact_scorer not in ("refined", "brute"):
            raise ValueError("exact_scorer must be 'refined' or 'brute'")
        self.exact_scorer = exact_scorer
        self.ctx = ctx                          # device context (default: the model's GPU)
        self.group = group                      # torch.distributed process group of a multi-GPU search (default: WORLD)
        self.distributed = distributed          # None: sharded search iff the group has more than one rank; True: whenever a
                                                # group is initialised (a world of one runs the same collectives: tests)
        # 16-bit score_dtype only: keep query / corpus rows as split-precision (hi + lo) pairs and score over 3 d columns on
        # the same kernels (runtime.Context.split16).  None = decided per search from the normalised query embeddings (the
        # same on every rank): rows that a few channels dominate (crest factor above SCORE_SPLIT_CREST) lose up to 4e-4 of
        # cosine to the 16-bit format alone; well-spread rows lose ~1e-5 and stay plain.  fp32 score_dtype is exact already.
        self.score_split = score_split
        self.last_score_split = False
        self.results = {}
        self.last_shard = None                  # (rank, world, first document, one-past-last document) of the last search

    def _to_dev(self, ctx, x):
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(np.asarray(x))
        return x.to(device=ctx.device, dtype=torch.float32)

    def _encode_queries(self, ctx, qlist, use_cache=True):
        if not qlist:
            return None
        if hasattr(self.model, "encode_queries_device"):
            q_emb = self.model.encode_queries_device(qlist, use_cache=use_cache)
        else:
            q_emb = self.model.encode_queries(qlist, batch_size=self.batch_size,
                                              show_progress_bar=self.show_progress_bar,
                                              convert_to_tensor=self.convert_to_tensor)
        return self._to_dev(ctx, q_emb)

    def search(self, corpus: Dict[str, Dict[str, str]], queries: Dict[str, str], top_k: int, score_function: str,
               return_sorted: bool = False, **kwargs) -> Dict[str, Dict[str, float]]:
        """Single process: the reference's loop (exact_search.py:34-134) with scoring / top-k / merge on the GPU.

        Under an initialised torch.distributed group of N > 1 ranks (one process per GPU, `torchrun ... beir_dense_retriever`)
        the SAME call is the corpus-sharded search of SURVEY 8e: every rank takes one contiguous range of the
        length-sorted corpus (cuts balanced on document length, dist.balanced_cuts -- equal counts would hand rank 0 the
        longest documents), tokenises, encodes and scores only that range (its embeddings never leave its HBM); the
        queries are encoded sharded and all-gathered once over RCCL; the per-rank top-(k+1) lists are exchanged and merged
        on the device (C ABI: sgpt_allgather_rows, sgpt_exchange_topk).  Every rank returns the same, complete dict --
        equal to the single-process result (the encoder is batch-invariant bit for bit, ties go to the lower index)."""
        if score_function not in self.score_function_desc:
            raise ValueError(
                "score function: {} must be either (cos_sim) for cosine similarity or (dot) for dot product".format(
                    score_function))
        from .dist import balanced_cuts, get_comm, is_distributed
        ctx = self.ctx if self.ctx is not None else get_context(getattr(getattr(self.model, "model", None), "device", None))
        import torch.distributed as tdist
        sharded = is_distributed(self.group) if self.distributed is None else (
            bool(self.distributed) and tdist.is_available() and tdist.is_initialized())
        comm = get_comm(ctx, self.group) if sharded else None
        world, rank = (comm.world, comm.rank) if comm is not None else (1, 0)

        logger.info("Encoding Queries...")
        query_ids = list(queries.keys())
        nq = len(query_ids)
        self.results = {qid: {} for qid in query_ids}
        qlist = [(qid, queries[qid]) for qid in queries]
        sgpt = getattr(self.model, "model", None)
        if comm is not None and hasattr(sgpt, "sync_precision") and hasattr(self.model, "tokenize") and qlist:
            # precision='auto' settles from the first sequences a process encodes -- a different query slice and corpus shard
            # on every rank.  Decide ONCE for the group, before anything is encoded: every rank probes a bounded sample of its
            # own queries, the flags are max-reduced, every rank installs the same plan (and logs it).
            from .dist import max_reducer
            own = qlist[rank::world][:64] or qlist[:64]
            sgpt.sync_precision(self.model.tokenize([q for (_, q) in own], True), max_reducer(ctx, self.group))
        if getattr(sgpt, "precision_report", None):
            rep = sgpt.precision_report
            logger.info("Encoder precision: %s (%s; %d operand classes flagged)", rep["decided"], rep["probed"], rep["flagged"])
        if comm is not None and (nq >= 4 * world or world == 1):
            # this rank's contiguous slice of the queries (cuts balanced on text length, never empty: an empty slice has no
            # embedding width to contribute and would leave the other ranks waiting in the collective), then ONE all-gather.
            # The `{base_path}_queries.pickle` cache of the reference holds ALL queries under one name: a sharded run neither
            # reads nor writes it (every rank would write its partial dict to the same path).
            qcuts = balanced_cuts([len(q[1]) + 1 for q in qlist], world, min_one=True)
            q_loc = self._encode_queries(ctx, qlist[qcuts[rank]: qcuts[rank + 1]], use_cache=False)
            q_emb = comm.all_gather_rows(q_loc, np.diff(qcuts).tolist())
        else:
            q_emb = self._encode_queries(ctx, qlist)       # (fewer queries than a handful per rank: every rank encodes them)
        if score_function == "cos_sim":
            q_emb = ctx.l2_normalize(q_emb)                                       # util.py:41 (once, not per chunk)
        split = False
        if self.score_dtype in (torch.float16, torch.bfloat16) and score_function == "cos_sim" and hasattr(ctx, "split16"):
            split = (ctx.row_crest(q_emb) > SCORE_SPLIT_CREST) if self.score_split is None else bool(self.score_split)
        self.last_score_split = split
        q_op = ctx.split16(q_emb, "query", self.score_dtype) if split else q_emb

        logger.info("Sorting Corpus by document length (Longest first)...")
        doc_len = {k: len(corpus[k].get("title", "") + corpus[k].get("text", "")) for k in corpus}
        corpus_ids = sorted(corpus, key=doc_len.__getitem__, reverse=True)        # :66-70
        pos_of = {cid: i for i, cid in enumerate(corpus_ids)}
        # corpus_id != query_id (:118) as an index: position of the doc carrying the query's id, or -1
        self_idx = torch.tensor([pos_of.get(qid, -1) for qid in query_ids], dtype=torch.int64, device=ctx.device)
        if comm is not None:
            ccuts = balanced_cuts([doc_len[c] + 1 for c in corpus_ids], world)
            lo, hi = int(ccuts[rank]), int(ccuts[rank + 1])
        else:
            lo, hi = 0, len(corpus_ids)
        self.last_shard = (rank, world, lo, hi)
        clist = [(cid, corpus[cid]) for cid in corpus_ids[lo:hi]]                # this rank's documents only

        run_val = run_idx = None
        itr = range(0, len(clist), self.corpus_chunk_size)
        # The host leg (tokenise + truncate + brackets) of chunk i+1 runs on a worker thread while the GPU encodes and
        # scores chunk i (the reference does both serially, exact_search.py:80-93).  Embedding caches bypass it.
        ahead = self.prefetch_tokenize and hasattr(self.model, "tokenize_corpus")
        pool = ThreadPoolExecutor(max_workers=1) if ahead else None
        fut = None
        # embedding-cache files are per chunk of the WHOLE sorted corpus in the reference; a sharded search numbers its
        # chunks per rank, so the cache names carry the rank
        tag = (lambda b: b) if comm is None else (lambda b: f"_r{rank}of{world}_{b}")

        def tokens_for(batch_num, start):
            if not ahead or self.model.uses_embedding_cache(tag(batch_num)) or start >= len(clist):
                return None
            return pool.submit(self.model.tokenize_corpus, clist[start: start + self.corpus_chunk_size])
        try:
            fut = tokens_for(0, 0)
            for batch_num, start in enumerate(itr):
                logger.info("Encoding Batch {}/{}...".format(batch_num + 1, len(itr)))
                end = min(start + self.corpus_chunk_size, len(clist))
                if ahead:
                    seqs = fut.result() if fut is not None else None
                    fut = tokens_for(batch_num + 1, end)
                    sub = self.model.encode_corpus_device(clist[start:end], batch_num=tag(batch_num), seqs=seqs)
                elif hasattr(self.model, "encode_corpus_device"):
                    sub = self.model.encode_corpus_device(clist[start:end], batch_num=tag(batch_num))
                else:
                    sub = self.model.encode_corpus(clist[start:end], batch_size=self.batch_size,
                                                   show_progress_bar=self.show_progress_bar,
                                                   convert_to_tensor=self.convert_to_tensor, batch_num=tag(batch_num))
                sub = self._to_dev(ctx, sub)
                kk = min(top_k + 1, end - start)                                      # :104
                refined = (self.score_dtype == torch.float32 and self.exact_scorer == "refined" and score_function == "cos_sim"
                           and hasattr(ctx, "score_topk_refined") and sub.shape[1] % 8 == 0)
                if score_function == "cos_sim":
                    sub = (ctx.split16(ctx.l2_normalize(sub), "doc", self.score_dtype) if split
                           else ctx.l2_normalize(sub, out_dtype=self.score_dtype))    # util.py:42
                if refined:     # fp32 scores of the fp32 top-k, found through f16 copies of the rows (:96-108, NaN -> -1)
                    val, idx, _ = ctx.score_topk_refined(q_op, sub, None, kk, idx_base=lo + start)
                else:
                    val, idx, _ = ctx.score_topk(q_op, sub, kk, idx_base=lo + start, dtype=self.score_dtype)   # :96-108 (NaN -> -1)
                if run_val is None:
                    cand_v, cand_i = val, idx
                else:
                    cand_v, cand_i = torch.cat([run_val, val], dim=1), torch.cat([run_idx, idx], dim=1)
                keep = min(top_k + 1, cand_v.shape[1])                                # :126
                run_val, run_idx = ctx.topk_merge(cand_v, cand_i, keep, exclude_idx=self_idx)          # :118,121-132
        finally:
            if pool is not None:      # an exception in encode / score must not leave the worker (and its tokenizer) running
                if fut is not None:
                    fut.cancel()
                pool.shutdown(wait=True)

        if comm is not None:
            # exchange: every rank's top-(k+1) list (padded with (-inf, -1); an empty range contributes only padding)
            k1 = top_k + 1
            pv = torch.full((nq, k1), float("-inf"), dtype=torch.float32, device=ctx.device)
            pi = torch.full((nq, k1), -1, dtype=torch.int64, device=ctx.device)
            if run_val is not None:
                pv[:, : run_val.shape[1]] = run_val
                pi[:, : run_idx.shape[1]] = run_idx
            run_val, run_idx = comm.exchange_topk(pv, pi, min(k1, len(corpus_ids)), exclude_idx=self_idx)
        if run_val is not None:
            self.results = assemble_results(query_ids, corpus_ids, run_val.cpu().numpy(), run_idx.cpu().numpy())
        return self.results


def _host_ext():
    """sgpt_amd/lib/_sgpt_host.so (csrc/host_assemble.c, built by sgpt_amd.build), or None."""
    global _HOST_EXT
    if _HOST_EXT is None:
        _HOST_EXT = False
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "_sgpt_host.so")
        if os.path.exists(path):
            import importlib.util
            try:
                spec = importlib.util.spec_from_file_location("_sgpt_host", path)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                _HOST_EXT = mod
            except Exception as e:          # built for another interpreter / not loadable: the Python form below gives the same dict
                logger.warning("host extension %s not loadable (%s): result dicts are assembled in Python", path, e)
    return _HOST_EXT or None


_HOST_EXT = None


def assemble_results(query_ids, corpus_ids, vals: np.ndarray, idxs: np.ndarray, native: Optional[bool] = None) -> Dict[str, Dict[str, float]]:
    """[nq, k] (score, corpus position) arrays -> the reference's result dict (exact_search.py:109-132: {qid: {doc_id: score}}).
    The reference driver asks for k = 1000 (+1): a million entries per 1000 queries -- at that depth this function, not the GPU,
    is the search leg (3.3 ms of device pass).  native (default: when built): the construction in C (csrc/host_assemble.c:
    pre-sized dicts, id strings prefetched ahead of the insert, no intermediate lists), 2.4x the Python form, which stays as
    the fallback and as the definition: one object-array gather and one tolist per row, dict(zip(...)).  Positions < 0 are
    padding and are skipped."""
    ext = _host_ext() if native in (None, True) else None
    if native is True and ext is None:
        raise RuntimeError("assemble_results(native=True): sgpt_amd/lib/_sgpt_host.so is not built (python -m sgpt_amd.build)")
    if ext is not None and len(query_ids) > 0:
        qs = query_ids if isinstance(query_ids, list) else list(query_ids)
        cs = corpus_ids if isinstance(corpus_ids, list) else list(corpus_ids)
        return ext.assemble(qs, cs, np.ascontiguousarray(vals, dtype=np.float32), np.ascontiguousarray(idxs, dtype=np.int64))
    cid = np.asarray(corpus_ids, dtype=object)
    ok_all = idxs >= 0
    full = bool(ok_all.all())
    out = {}
    for qi, qid in enumerate(query_ids):
        if full:
            out[qid] = dict(zip(cid[idxs[qi]].tolist(), vals[qi].tolist()))
        else:
            ok = ok_all[qi]
            out[qid] = dict(zip(cid[idxs[qi][ok]].tolist(), vals[qi][ok].tolist()))
    return out


class SentenceBERTBOSEOS:
    """custommodels/sentence_bert_asym.py:21-79: the `--usest --specb / --speca` adapter.  The reference
    prefixes "[SOS]" / "{SOS}" marker tokens that Transformer.tokenize_bos_eos keeps (speca) or swaps for
    the bracket ids (specb) and closes with the EOS marker / bracket (Transformer.py:131-153); here the
    ids are added directly on the id lists (same ids, no marker round trip; the content is cut to
    max_seq_length - 3 tokens exactly as the marker-inclusive truncation of :135 does).  speca uses four
    ADDED vocabulary rows, so the checkpoint's embedding table must already hold them (the reference
    resizes it, sentence_bert_asym.py:54-55).  Inputs follow upstream BEIR's DRES: List[str] queries,
    List[{"title","text"}] corpus."""

    def __init__(self, model_path=None, sep: str = " ", speca=False, specb=False, model: Optional[SGPTModel] = None,
                 tokenizer=None, max_seq_length: int = 300, method: str = "weightedmean", dtype="f16",
                 device="cuda:0", **kwargs):
        self.sep = sep
        self.speca, self.specb = speca, specb
        if not (specb or speca):
            raise NameError("name 'sentences' is not defined")   # the reference fails the same way (appendix A.2)
        self.model = model if model is not None else SGPTModel.from_pretrained(model_path, device=device, dtype=dtype)
        tok = tokenizer if tokenizer is not None else load_tokenizer(model_path)
        self.pipe = TextPipeline(tok, max_seq_length, specb=specb and not speca, speca=speca, st_path=True)
        if speca and max(self.pipe.bos_q + self.pipe.eos_q + self.pipe.bos_d + self.pipe.eos_d) >= self.model.cfg.vocab_size:
            raise ValueError("speca: the checkpoint's embedding table has no rows for the added [SOS]/[EOS]/{SOS}/{EOS} ids")
        self.method = method

    def _encode(self, texts, is_query, convert_to_tensor=False, normalize_embeddings=False, **kwargs):
        seqs = self.pipe.batch([str(t).strip() for t in texts], is_query)
        emb = self.model.encode_ids(seqs, mode=self.method, normalize=normalize_embeddings)
        return emb if convert_to_tensor else emb.cpu().numpy()

    def encode_queries(self, queries: List[str], batch_size: int = 16, **kwargs):
        queries = [q[1] if isinstance(q, tuple) else q for q in queries]   # custom DRES passes (id, text)
        return self._encode(queries, True, **kwargs)

    def encode_corpus(self, corpus: List[Dict[str, str]], batch_size: int = 8, **kwargs):
        corpus = [c[1] if isinstance(c, tuple) else c for c in corpus]
        kwargs.pop("batch_num", None)
        sentences = [(doc["title"] + self.sep + doc["text"]).strip() if "title" in doc else doc["text"].strip()
                     for doc in corpus]
        return self._encode(sentences, False, **kwargs)


class SentenceBERTAsym:
    """custommodels/sentence_bert_asym.py:8-19: asymmetric two-tower model -- queries go through the `QRY`
    Transformer, documents through the `DOCPOS` Transformer of models/Asym.py (train_bi-encoder_mnrl.py:139),
    one shared weight-less Pooling.  Two SGPTModel weight sets on the same GPU, the same kernels."""

    def __init__(self, model_path=None, sep: str = " ", query_model: Optional[SGPTModel] = None,
                 doc_model: Optional[SGPTModel] = None, tokenizer=None, max_seq_length: int = 300,
                 method: str = "weightedmean", dtype="f16", device="cuda:0", **kwargs):
        self.sep = sep
        if query_model is None or doc_model is None:
            from .formats import read_st_folder
            spec = read_st_folder(model_path)
            if not spec.asymmetric or "QRY" not in spec.transformer_dirs or "DOCPOS" not in spec.transformer_dirs:
                raise ValueError(f"{model_path} is not an Asym model with QRY / DOCPOS towers")
            query_model = SGPTModel.from_pretrained(spec.transformer_dirs["QRY"], device=device, dtype=dtype)
            doc_model = SGPTModel.from_pretrained(spec.transformer_dirs["DOCPOS"], device=device, dtype=dtype)
            method, max_seq_length = spec.pooling_mode, spec.max_seq_length or max_seq_length
            tokenizer = tokenizer if tokenizer is not None else load_tokenizer(spec.transformer_dirs["QRY"])
        self.query_model, self.doc_model = query_model, doc_model
        self.pipe = TextPipeline(tokenizer, max_seq_length)
        self.method = method

    def _encode(self, model, texts, convert_to_tensor=False, normalize_embeddings=False, **kwargs):
        seqs = self.pipe.batch([str(t).strip() for t in texts], True)
        emb = model.encode_ids(seqs, mode=self.method, normalize=normalize_embeddings)
        return emb if convert_to_tensor else emb.cpu().numpy()

    def encode_queries(self, queries: List[str], batch_size: int = 16, **kwargs):
        queries = [q[1] if isinstance(q, tuple) else q for q in queries]
        return self._encode(self.query_model, queries, **kwargs)

    def encode_corpus(self, corpus: List[Dict[str, str]], batch_size: int = 8, **kwargs):
        corpus = [c[1] if isinstance(c, tuple) else c for c in corpus]
        kwargs.pop("batch_num", None)
        sentences = [(doc["title"] + self.sep + doc["text"]).strip() if "title" in doc else doc["text"].strip()
                     for doc in corpus]
        return self._encode(self.doc_model, sentences, **kwargs)
