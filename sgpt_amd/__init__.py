"""sgpt_amd -- MI355X (gfx950) native SGPT bi-encoder retrieval hot path.

token ids -> GPT-Neo forward -> position-weighted mean pool -> cosine/dot scoring -> top-k,
as hand-written HIP kernels behind the C ABI of include/sgpt_hip.h, with Python adapters that
keep the reference's encode_queries / encode_corpus / encode / semb_fn / cos_sim / search
surface (biencoder/beir, biencoder/useb of Muennighoff/sgpt)."""
__version__ = "0.1.0"

from .model import EncodeGraph, SGPTConfig, SGPTModel, synthetic_weights  # noqa: F401
from .runtime import Context, get_context  # noqa: F401
