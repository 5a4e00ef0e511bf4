"""USEB drop-in surface (biencoder/useb/useb_dense_retriever.py:174-309,455-546):
`semb_fn(sentences, dataset_name=None, add_name="", idx=None, **kw) -> torch.Tensor[len, d]` on the
CPU, as useb/useb/useb/evaluators/base.py:33 calls it (batches of 32; AskUbuntu: 21)."""
from typing import List, Optional

import torch

from .beir import ALL_LAYER_METHODS, SINGLE_LAYER_METHODS
from .model import SGPTModel
from .tokenization import TextPipeline


class CustomEmbedder:
    """useb_dense_retriever.py:76-309: encode(sentences, ...) -> list of lists; same pooling
    methods as the BEIR embedder plus `learntmean` (:253-270: trained position weights, read from
    <model>/1_WeightedMeanPooling by SGPTModel.from_pretrained or set with model.set_position_weights)."""

    def __init__(self, model: SGPTModel, tokenizer, layeridx: int = -1, method: str = "weightedmean",
                 specb: bool = False, maxseqlen: Optional[int] = None):
        if method not in SINGLE_LAYER_METHODS and method not in ALL_LAYER_METHODS and method != "learntmean":
            raise ValueError(f"unknown method {method}")
        self.model = model
        self.layeridx = layeridx
        self.method = method
        self.pipe = TextPipeline(tokenizer, maxseqlen or model.cfg.max_position_embeddings, specb=specb)

    def encode_device(self, sentences: List[str], is_query: bool = True) -> torch.Tensor:
        seqs = self.pipe.batch(sentences, is_query)
        L = self.model.cfg.num_layers
        if self.method in SINGLE_LAYER_METHODS or self.method == "learntmean":
            return self.model.encode_ids(seqs, mode=self.method, layer_idx=self.layeridx)
        return self.model.encode_ids_all_layers(seqs, mode=ALL_LAYER_METHODS[self.method])

    def encode(self, sentences, **kwargs):
        return self.encode_device(list(sentences)).cpu().tolist()


def make_semb_fn(embedder: CustomEmbedder):
    """The METHOD_TO_FN closures of useb_dense_retriever.py:455-527."""
    def semb_fn(sentences, dataset_name=None, add_name="", idx=None, **kwargs) -> torch.Tensor:
        return embedder.encode_device(list(sentences)).cpu()
    return semb_fn
