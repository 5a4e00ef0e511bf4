"""SentenceTransformer-style surface (sentence_transformers/SentenceTransformer.py:107-255) over
the HIP encoder: `encode(sentences, batch_size, ..., normalize_embeddings)` with the reference's
return-type rules, plus its torch.distributed data-parallel branch (:153-175) re-done with
token-balanced shards and ONE RCCL all-gather (C ABI: sgpt_allgather_rows) instead of the two-step
pad-to-max gather of util.mismatched_sizes_all_gather (util.py:326-347)."""
from typing import List, Optional, Union

import numpy as np
import torch

from .dist import balanced_cuts, get_comm, is_distributed, shard_sizes  # noqa: F401  (shard_sizes: re-exported)
from .model import ALIGN, SGPTModel
from .runtime import get_context  # noqa: F401
from .tokenization import TextPipeline


class SentenceTransformerSGPT:
    """Transformer + Pooling(weightedmean) [+ Normalize] as one module
    (the assembly of training_nli_v2.py:85-123 / modules.json)."""

    def __init__(self, model: SGPTModel, tokenizer, max_seq_length: int = 300, pooling_mode: str = "weightedmean",
                 specb: bool = False, normalize: bool = False):
        self.model = model
        self.tokenizer = tokenizer
        self.max_seq_length = max_seq_length
        self.pooling_mode = pooling_mode
        self.specb = specb
        self.normalize = normalize          # a Normalize module in modules.json (models/Normalize.py)
        self.pipe = TextPipeline(tokenizer, max_seq_length, specb=specb)

    @classmethod
    def from_pretrained(cls, path: str, tokenizer=None, device=None, dtype: str = "f16", specb: bool = False,
                        **model_kw) -> "SentenceTransformerSGPT":
        """SentenceTransformer(model_path) for an SGPT folder (SentenceTransformer._load_sbert_model, :903-936):
        modules.json -> Transformer weights, Pooling / WeightedMeanPooling mode, optional Normalize."""
        from .formats import read_st_folder
        from .tokenization import load_tokenizer
        spec = read_st_folder(path)
        if spec.asymmetric:
            raise ValueError("asymmetric (Asym) folders load through sgpt_amd.beir.SentenceBERTAsym")
        model = SGPTModel.from_pretrained(spec.transformer_dirs[""], device=device, dtype=dtype, **model_kw)
        if spec.position_weights_file:
            model.set_position_weights(torch.load(spec.position_weights_file, map_location="cpu", weights_only=True)["position_weights"])
        tok = tokenizer if tokenizer is not None else load_tokenizer(spec.transformer_dirs[""])
        return cls(model, tok, max_seq_length=spec.max_seq_length or model.cfg.max_position_embeddings,
                   pooling_mode=spec.pooling_mode, specb=specb, normalize=spec.normalize)

    def get_sentence_embedding_dimension(self) -> int:
        return self.model.cfg.hidden_size

    def encode(self, sentences: Union[str, List[str]], batch_size: int = 32, show_progress_bar: bool = None,
               output_value: str = "sentence_embedding", convert_to_numpy: bool = True,
               convert_to_tensor: bool = False, device: str = None, normalize_embeddings: bool = False,
               num_proc=None, is_query: bool = True):
        """SentenceTransformer.encode (SentenceTransformer.py:107-215), same arguments and return-type rules:
          output_value 'sentence_embedding' -> ndarray [n, d] (convert_to_numpy, the default) | Tensor [n, d] (convert_to_tensor)
                        | list of Tensors; 'token_embeddings' -> list of [len_i, d] Tensors (:233-241); None -> list of dicts
                        {input_ids, attention_mask, token_embeddings, sentence_embedding} per sentence (:242-246; not
                        normalised, as in the reference; rows carry their own length instead of the batch's padding);
                        for both, convert_to_numpy / convert_to_tensor are ignored (:133-135);
          device        the reference moves the module there on every call (:180-183).  The weights here are resident on one
                        GPU behind the C ABI: None or that GPU is accepted, anything else is refused (build a second
                        SGPTModel on the other device);
          num_proc      None / 1: this process.  > 1 in the reference pickles the module into a process pool (:190-203); a
                        device handle cannot be pickled, so that form is refused with a pointer to the equivalents
                        (start_multi_process_pool(model_factory) / encode_multi_process, or torchrun + this method);
          batch_size / show_progress_bar are accepted for signature compatibility (batches are cut by token budget).
        A single string returns a single vector / dict (:143-146, 212-213)."""
        if device is not None:
            dev, mine = torch.device(device), torch.device(self.model.device)
            if dev.type != mine.type or dev.index not in (None, mine.index):
                raise ValueError(f"encode(device={device!r}): this model's weights are resident on {mine} (C-ABI handle); "
                                 "load a second SGPTModel on the other device instead of moving this one per call")
        if num_proc is not None and int(num_proc) > 1:
            raise ValueError("encode(num_proc > 1): a device-resident model cannot be pickled into a process pool; use "
                             "sgpt_amd.st.start_multi_process_pool(model_factory) + encode_multi_process, or launch one "
                             "process per GPU with torchrun (encode() then shards the sentences itself)")
        if convert_to_tensor:
            convert_to_numpy = False
        if output_value != "sentence_embedding":                                 # :133-135
            convert_to_tensor = convert_to_numpy = False
        normalize_embeddings = normalize_embeddings or self.normalize
        input_was_string = False
        if isinstance(sentences, str) or not hasattr(sentences, "__len__"):     # :143-146
            sentences = [sentences]
            input_was_string = True
        seqs = self.pipe.batch([str(s).strip() for s in sentences], is_query)

        if output_value == "token_embeddings":                                  # :233-241
            embs = self.model.token_embeddings(seqs)
            return embs[0] if input_was_string else embs
        if output_value is None:                                                # :242-246: every output of the module chain
            toks = self.model.token_embeddings(seqs)
            sent = self.model.encode_ids(seqs, mode=self.pooling_mode, normalize=self.normalize)
            rows = [{"input_ids": torch.tensor(s, dtype=torch.int64, device=t.device),
                     "attention_mask": torch.ones(len(s), dtype=torch.int64, device=t.device),
                     "token_embeddings": t, "sentence_embedding": e} for s, t, e in zip(seqs, toks, sent)]
            return rows[0] if input_was_string else rows
        if output_value != "sentence_embedding":
            raise ValueError("output_value must be 'sentence_embedding', 'token_embeddings' or None")

        if is_distributed():
            emb = self.encode_ids_distributed(seqs, normalize_embeddings)
        else:
            emb = self.model.encode_ids(seqs, mode=self.pooling_mode, normalize=normalize_embeddings)

        if convert_to_numpy:
            emb = emb.cpu().numpy()
        elif not convert_to_tensor:
            emb = [e for e in emb]                                               # list of tensors (:207-210)
        if input_was_string:
            emb = emb[0]
        return emb

    def encode_ids_distributed(self, seqs, normalize_embeddings: bool = False, group=None) -> torch.Tensor:
        """The torch.distributed data-parallel branch of SentenceTransformer.encode (:153-175): every rank sorts the
        SAME list by length (longest first, :148-149 -- appendix A.10: the un-sort below relies on all ranks agreeing
        on this order, so the sort is stable and keyed on the token count only) and encodes one CONTIGUOUS shard of the
        sorted list.  The reference cuts equal sentence counts (:159-163), which gives rank 0 the longest sentences
        (U{16..128} tokens, 8 ranks: 1.7x the mean token load); here the cuts balance the token rows each rank packs
        (dist.balanced_cuts: max / mean <= 1 + one sentence).  ONE all-gather (every rank knows every count) puts the
        shards back together in sorted order, and the inverse permutation restores the input order (:205).  Every rank
        returns the full [n, d] matrix."""
        ctx = getattr(self.model, "ctx", None)
        comm = get_comm(ctx, group)
        lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
        order = np.argsort(-lens, kind="stable")
        cuts = balanced_cuts((lens[order] + ALIGN - 1) // ALIGN * ALIGN, comm.world)
        counts = np.diff(cuts).tolist()
        mine = [seqs[i] for i in order[cuts[comm.rank]: cuts[comm.rank + 1]]]
        d = self.get_sentence_embedding_dimension()
        if mine:
            local = self.model.encode_ids(mine, mode=self.pooling_mode, normalize=normalize_embeddings)
        else:                                                     # more ranks than sentences
            local = torch.empty((0, d), dtype=torch.float32, device=getattr(self.model, "device", "cpu"))
        gathered = comm.all_gather_rows(local, counts)
        emb = torch.empty_like(gathered)
        emb[torch.from_numpy(order).to(gathered.device)] = gathered              # un-sort (:205)
        return emb


# ---- multi-process pool: SentenceTransformer.start_multi_process_pool / encode_multi_process (:257-353) ----
def _pool_worker(target_device, model_factory, input_queue, results_queue):
    """One process per device (SentenceTransformer._encode_multi_process_worker, :335-353): pulls
    [chunk_id, batch_size, sentences] items, pushes [chunk_id, embeddings ndarray]."""
    import queue as _q
    model = model_factory(target_device)
    while True:
        try:
            item = input_queue.get()
            if item is None:
                break
            chunk_id, batch_size, sentences = item
            emb = model.encode(sentences, batch_size=batch_size, convert_to_numpy=True, show_progress_bar=False)
            results_queue.put([chunk_id, emb])
        except _q.Empty:
            break


def start_multi_process_pool(model_factory, target_devices: List[str] = None):
    """SentenceTransformer.start_multi_process_pool (:257-285): one spawned process per target device
    (default: every visible GPU).  `model_factory(device) -> SentenceTransformerSGPT` is called inside each
    worker (device handles cannot be pickled across processes; the reference re-moves the pickled model)."""
    import torch.multiprocessing as mp
    if target_devices is None:
        target_devices = [f"cuda:{i}" for i in range(torch.cuda.device_count())]
    ctx = mp.get_context("spawn")
    input_queue, output_queue, processes = ctx.Queue(), ctx.Queue(), []
    for dev in target_devices:
        p = ctx.Process(target=_pool_worker, args=(dev, model_factory, input_queue, output_queue), daemon=True)
        p.start()
        processes.append(p)
    return {"input": input_queue, "output": output_queue, "processes": processes}


def stop_multi_process_pool(pool):
    """SentenceTransformer.stop_multi_process_pool (:288-301)."""
    for _ in pool["processes"]:
        pool["input"].put(None)
    for p in pool["processes"]:
        p.join(60)
        if p.is_alive():
            p.terminate()
    pool["input"].close()
    pool["output"].close()


def encode_multi_process(sentences: List[str], pool, batch_size: int = 32, chunk_size: int = None) -> np.ndarray:
    """SentenceTransformer.encode_multi_process (:304-332): sentences are cut into chunks
    (min(ceil(n / procs / 10), 5000) by default, :317-318), sent to the workers, results concatenated in order."""
    import math
    if chunk_size is None:
        chunk_size = min(math.ceil(len(sentences) / len(pool["processes"]) / 10), 5000)
    chunk_size = max(1, chunk_size)
    n_chunks = 0
    for start in range(0, len(sentences), chunk_size):
        pool["input"].put([n_chunks, batch_size, sentences[start:start + chunk_size]])
        n_chunks += 1
    results = sorted([pool["output"].get() for _ in range(n_chunks)], key=lambda x: x[0])
    return np.concatenate([r[1] for r in results])
