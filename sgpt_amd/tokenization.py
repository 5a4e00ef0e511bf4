"""Host-side tokenisation leg: text -> ids, truncation, specb brackets.
Stays on the host (integer work on Python strings); the kernel boundary starts at the id lists.

Restates CustomEmbedder.embed's per-text loop (biencoder/beir/beir_dense_retriever.py:167-198)
and Transformer.tokenize_bos_eos (sentence_transformers/models/Transformer.py:131-153)."""
import re
import zlib
from typing import List, Optional, Sequence

SPECB_QUE_BOS, SPECB_QUE_EOS = "[", "]"     # beir_dense_retriever.py:100-104
SPECB_DOC_BOS, SPECB_DOC_EOS = "{", "}"


class SyntheticTokenizer:
    """Deterministic stand-in used when no vocabulary files exist (this image has no tokenizer
    files and no network): whitespace/punctuation split, token -> crc32 % vocab.  It duck-types
    the three HF tokenizer calls the reference makes: tokenize(), convert_tokens_to_ids(),
    encode().  Bracket characters map to their GPT-2 BPE ids ([ ] { } = 58 60 90 92)."""
    _FIXED = {"[": 58, "]": 60, "{": 90, "}": 92}

    def __init__(self, vocab_size: int = 50257):
        self.vocab_size = vocab_size
        self.eos_token_id = min(50256, vocab_size - 1)
        self.pad_token_id = self.eos_token_id

    def tokenize(self, text: str) -> List[str]:
        return re.findall(r"\w+|[^\w\s]", text)

    def convert_tokens_to_ids(self, tokens: Sequence[str]) -> List[int]:
        return [self._FIXED[t] if t in self._FIXED else zlib.crc32(t.encode()) % (self.vocab_size - 1) for t in tokens]

    def encode(self, text: str, add_special_tokens: bool = False) -> List[int]:
        return self.convert_tokens_to_ids(self.tokenize(text))


def load_tokenizer(model_name_or_path: str):
    """AutoTokenizer.from_pretrained (beir_dense_retriever.py:138) with pad = eos for GPT models (:140-141)."""
    from transformers import AutoTokenizer
    tok = AutoTokenizer.from_pretrained(model_name_or_path)
    if "gpt" in model_name_or_path.lower() and tok.pad_token is None:
        tok.pad_token = tok.eos_token
    return tok


class TextPipeline:
    """text -> truncated id list (+ brackets)."""

    def __init__(self, tokenizer, max_token_len: int, specb: bool = False):
        self.tok = tokenizer
        self.specb = specb
        self.max_token_len = max_token_len - 2 if specb else max_token_len   # :134-136
        if specb:
            self.bos_q = list(tokenizer.encode(SPECB_QUE_BOS))
            self.eos_q = list(tokenizer.encode(SPECB_QUE_EOS))
            self.bos_d = list(tokenizer.encode(SPECB_DOC_BOS))
            self.eos_d = list(tokenizer.encode(SPECB_DOC_EOS))
        self.docs_truncated = 0
        self.toks_truncated = 0

    def ids(self, txt: str, is_query: bool) -> List[int]:
        txt = txt.replace("\n", " ")                                           # :169
        tokens = self.tok.convert_tokens_to_ids(self.tok.tokenize(txt))        # :172-173
        n = len(tokens)
        if n > self.max_token_len:
            self.docs_truncated += 1
            self.toks_truncated += n - self.max_token_len
        elif n == 0:
            raise ValueError("Empty items should be cleaned prior to running")  # :180-181
        tokens = list(tokens[: self.max_token_len])
        if self.specb:                                                          # :186-191
            tokens = (self.bos_q + tokens + self.eos_q) if is_query else (self.bos_d + tokens + self.eos_d)
        return tokens

    def batch(self, texts: Sequence[str], is_query: bool) -> List[List[int]]:
        return [self.ids(t, is_query) for t in texts]
