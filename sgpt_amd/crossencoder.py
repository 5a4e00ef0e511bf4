"""Cross-encoder re-ranking surface of the reference (crossencoder/beir/sgptce.py): the score of a (query, document)
pair is log P(query tokens | prompt(document)) under the causal LM -- `GPTRanker.predict` -> `_loglikelihood_tokens`.

Same forward kernels as the bi-encoder; what is new on the device is the LM head on the continuation rows only
(sgpt_lm_logprobs: gather rows -> exact-fp32 logits GEMM -> log-softmax + gather), instead of materialising
[batch, seq, vocab] logits and moving them to the host (sgptce.py:233)."""
from typing import List, Sequence, Tuple

import numpy as np
import torch

from .model import ALIGN, SGPTModel


def _ids(tokenizer, text: str) -> List[int]:
    return list(tokenizer.encode(text, add_special_tokens=False))


def encode(requests: Sequence[Tuple[str, str]], tokenizer):
    """(query, prompted document) pairs -> ((document, query), document ids, query ids): the request triples
    `loglikelihood_tokens` scores.  An empty document stands for the end-of-text id alone (sgptce.py:77-91)."""
    out = []
    for query, doc in requests:
        doc_ids = _ids(tokenizer, doc) if doc != "" else [tokenizer.eos_token_id]
        out.append(((doc, query), doc_ids, _ids(tokenizer, query)))
    return out


def model_input(context_enc: List[int], continuation_enc: List[int], max_length: int, instruction_len: int = 0) -> List[int]:
    """sgptce.py:204-211: keep the instruction, truncate the rest from the LEFT to max_length + 1 - instruction_len
    tokens, drop the final token (it is only ever a target)."""
    assert len(context_enc) > 0 and len(continuation_enc) > 0
    assert len(continuation_enc) <= max_length, f"Got {len(continuation_enc)} but max len is only {max_length}"
    rest = (context_enc[instruction_len:] + continuation_enc)[-(max_length + 1 - instruction_len):]
    return (context_enc[:instruction_len] + rest)[:-1]


def loglikelihood_tokens(requests, model: SGPTModel, max_length: int, instruction_len: int = 0,
                         max_tokens_per_call: int = 65536) -> List[float]:
    """`_loglikelihood_tokens` (sgptce.py:150-262): for every ((context, continuation), context_enc, continuation_enc)
    the sum of log P(continuation token | everything before it).  Requests are batched by a token budget in the
    packed var-len layout (causal attention: a sequence never sees its batch); results come back in request order."""
    inps, spans = [], []
    for _, ctx_enc, cont_enc in requests:
        inp = model_input(list(ctx_enc), list(cont_enc), max_length, instruction_len)
        if len(cont_enc) > len(inp):
            raise ValueError("continuation longer than the model input left after truncation")
        inps.append(inp)
        spans.append((len(inp) - len(cont_enc), len(inp), list(cont_enc)))
    order = np.argsort([-len(x) for x in inps], kind="stable")        # longest first, as the reference's Reorderer
    # under the model's range guard: an f16 activation class that overflows gets its power-of-two shift raised and the
    # requests are scored again (otherwise inf / NaN log-probabilities would come back silently)
    model.ensure_precision_plan(inps)           # precision='auto': probe the checkpoint on the first requests
    return model.guarded(lambda: _score_batches(inps, spans, order, model, max_tokens_per_call))


def _score_batches(inps, spans, order, model: SGPTModel, max_tokens_per_call: int) -> List[float]:
    res = [0.0] * len(inps)
    start = 0
    while start < len(order):
        tok, end = 0, start
        while end < len(order) and (end == start or tok + (len(inps[order[end]]) + ALIGN - 1) // ALIGN * ALIGN <= max_tokens_per_call):
            tok += (len(inps[order[end]]) + ALIGN - 1) // ALIGN * ALIGN
            end += 1
        sel = order[start:end]
        pb = model.pack([inps[i] for i in sel])
        _, hidden = model.encode_packed(pb, return_hidden=True)       # post-ln_f hidden states [T_pad, d]
        off = pb.seq_off.cpu().numpy()
        rows, tgts, owner = [], [], []
        for b, i in enumerate(sel):
            lo, hi, cont = spans[i]
            rows.extend(range(int(off[b]) + lo, int(off[b]) + hi))    # logits[inplen - contlen : inplen]
            tgts.extend(cont)
            owner.extend([b] * (hi - lo))
        lp = model.lm_logprobs(hidden, rows, tgts).cpu().numpy().astype(np.float64)
        sums = np.zeros(len(sel), dtype=np.float64)
        np.add.at(sums, np.asarray(owner), lp)
        for b, i in enumerate(sel):
            res[i] = float(sums[b])
        start = end
    return res


class GPTRanker:
    """sgptce.py:265-331 (`Rerank(GPTRanker(...))` in BEIR): predict([(query, doc), ...]) -> log-probabilities.
    The text in front of the document slot of `prompt_doc` (and the few-shot block, if any) is the instruction: it is never
    truncated away, and its token count is fixed at construction."""

    def __init__(self, model: SGPTModel, tokenizer, max_length: int = None, use_prompt: bool = True,
                 prompt_doc: str = "{}\n", prompt_doc_start: str = "{}\n{}\n", fewshots=""):
        self.model, self.tokenizer = model, tokenizer
        self.max_length = max_length if max_length else model.cfg.max_position_embeddings
        self.use_prompt, self.prompt_doc = use_prompt, prompt_doc
        n_tok = lambda text: len(tokenizer.tokenize(text))  # noqa: E731
        self.fewshots = prompt_doc_start.format(fewshots[0], fewshots[1]) if fewshots else ""   # one formatted block
        head = prompt_doc.split("{", 1)[0]
        if "{" not in prompt_doc:
            raise ValueError("prompt_doc needs a {} slot for the document")
        self.instruction_len = n_tok(head) + (n_tok(self.fewshots) if self.fewshots else 0)

    def predict(self, sentences: List[Tuple[str, str]], batch_size: int = 0, **kwargs) -> List[float]:
        pairs = list(sentences)
        if self.use_prompt:
            pairs = [(query, self.fewshots + self.prompt_doc.format(doc)) for query, doc in pairs]
        return loglikelihood_tokens(encode(pairs, self.tokenizer), self.model, self.max_length,
                                    instruction_len=self.instruction_len)
