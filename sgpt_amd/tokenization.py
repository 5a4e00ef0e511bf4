"""Host-side tokenisation leg: text -> ids, truncation, specb brackets.
Stays on the host (integer work on Python strings); the kernel boundary starts at the id lists.

Restates CustomEmbedder.embed's per-text loop (biencoder/beir/beir_dense_retriever.py:167-198)
and Transformer.tokenize_bos_eos (sentence_transformers/models/Transformer.py:131-153)."""
import re
import zlib
from typing import List, Optional, Sequence

SPECB_QUE_BOS, SPECB_QUE_EOS = "[", "]"     # beir_dense_retriever.py:100-104
SPECB_DOC_BOS, SPECB_DOC_EOS = "{", "}"
SPECA_TOKENS = ["[SOS]", "[EOS]", "{SOS}", "{EOS}"]   # sentence_bert_asym.py:52-53: added vocabulary rows


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
        self.added = {}

    def __len__(self):
        return self.vocab_size + len(self.added)

    def add_tokens(self, tokens: Sequence[str], special_tokens: bool = False) -> int:
        """HF add_tokens: new ids are appended after the base vocabulary (the model's embedding is resized to match,
        sentence_bert_asym.py:36-37,54-55)."""
        new = [t for t in tokens if t not in self.added]
        for t in new:
            self.added[t] = self.vocab_size + len(self.added)
        return len(new)

    def tokenize(self, text: str) -> List[str]:
        if self.added:
            pat = "|".join(re.escape(t) for t in sorted(self.added, key=len, reverse=True))
            return re.findall(pat + r"|\w+|[^\w\s]", text)
        return re.findall(r"\w+|[^\w\s]", text)

    def convert_tokens_to_ids(self, tokens: Sequence[str]) -> List[int]:
        return [self.added[t] if t in self.added else self._FIXED[t] if t in self._FIXED
                else zlib.crc32(t.encode()) % (self.vocab_size - 1) for t in tokens]

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
    """text -> truncated id list (+ brackets).

    specb: `[`..`]` around queries, `{`..`}` around documents, existing vocabulary ids.
    speca: `[SOS]`..`[EOS]` / `{SOS}`..`{EOS}`, ids ADDED to the vocabulary (the checkpoint carries the extra
           embedding rows); only on the sentence-transformers path, as in the reference (beir_dense_retriever.py:415-416).
    st_path: the marker travels inside the text through the tokenizer call of Transformer.tokenize_bos_eos
           (Transformer.py:131-135, `max_length = max_seq_length - 2` INCLUDING the marker), so the content is cut to
           max_seq_length - 3 tokens; the raw-HF path (beir_dense_retriever.py:134-136,172-191) cuts the content to
           max_seq_length - 2 and adds both brackets afterwards."""

    def __init__(self, tokenizer, max_token_len: int, specb: bool = False, speca: bool = False, st_path: bool = False):
        if speca and specb:
            raise ValueError("speca and specb are mutually exclusive")
        self.tok = tokenizer
        self.specb, self.speca, self.st_path = specb, speca, st_path
        bracketed = specb or speca
        self.max_token_len = max_token_len - (3 if st_path else 2) if bracketed else max_token_len   # :134-136
        if specb:
            self.bos_q = list(tokenizer.encode(SPECB_QUE_BOS))
            self.eos_q = list(tokenizer.encode(SPECB_QUE_EOS))
            self.bos_d = list(tokenizer.encode(SPECB_DOC_BOS))
            self.eos_d = list(tokenizer.encode(SPECB_DOC_EOS))
        if speca:
            tokenizer.add_tokens(SPECA_TOKENS, special_tokens=True)             # sentence_bert_asym.py:52-54
            enc = [list(tokenizer.encode(t, add_special_tokens=False)) for t in SPECA_TOKENS]
            if any(len(e) != 1 for e in enc):
                raise ValueError("speca markers must be single added tokens of the tokenizer")
            self.bos_q, self.eos_q, self.bos_d, self.eos_d = enc
        self.docs_truncated = 0
        self.toks_truncated = 0

    def ids(self, txt: str, is_query: bool) -> List[int]:
        if not self.st_path:                                                    # the ST path hands the text to the
            txt = txt.replace("\n", " ")                                       # tokenizer untouched; raw path: :169
        tokens = self.tok.convert_tokens_to_ids(self.tok.tokenize(txt))        # :172-173
        n = len(tokens)
        if n > self.max_token_len:
            self.docs_truncated += 1
            self.toks_truncated += n - self.max_token_len
        elif n == 0:
            raise ValueError("Empty items should be cleaned prior to running")  # :180-181
        tokens = list(tokens[: self.max_token_len])
        if self.specb or self.speca:                                            # :186-191
            tokens = (self.bos_q + tokens + self.eos_q) if is_query else (self.bos_d + tokens + self.eos_d)
        return tokens

    def batch(self, texts: Sequence[str], is_query: bool) -> List[List[int]]:
        """All texts of a call at once.  A HF *fast* tokenizer takes the whole list in ONE call (its Rust core
        tokenises the batch in parallel and returns the ids directly: the id sequence of
        `convert_tokens_to_ids(tokenize(t))`, beir_dense_retriever.py:172-173); truncation and brackets are then list
        slices.  Any other tokenizer object (slow HF tokenizers, SyntheticTokenizer) goes through the per-text calls."""
        if not getattr(self.tok, "is_fast", False) or len(texts) < 2:
            return [self.ids(t, is_query) for t in texts]
        if not self.st_path:
            texts = [t.replace("\n", " ") for t in texts]                        # :169
        enc = self.tok(list(texts), add_special_tokens=False, padding=False, truncation=False,
                       return_attention_mask=False, return_token_type_ids=False)["input_ids"]
        bracket = self.specb or self.speca
        out = []
        lim = self.max_token_len
        for tokens in enc:
            n = len(tokens)
            if n > lim:
                self.docs_truncated += 1
                self.toks_truncated += n - lim
                tokens = tokens[:lim]
            elif n == 0:
                raise ValueError("Empty items should be cleaned prior to running")  # :180-181
            if bracket:                                                             # :186-191
                tokens = (self.bos_q + tokens + self.eos_q) if is_query else (self.bos_d + tokens + self.eos_d)
            out.append(tokens)
        return out
