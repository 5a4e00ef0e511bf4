"""On-disk formats of the reference that sit either side of the hot path, host side only:

* sentence-transformers model folders -- `modules.json` + per-module sub-folders as written by
  SentenceTransformer.save (sentence_transformers/SentenceTransformer.py:389-430) and read back by
  `_load_sbert_model` (:903-936): Transformer (`sentence_bert_config.json`, models/Transformer.py:158-175),
  Pooling (`config.json`, models/Pooling.py:172-185), WeightedMeanPooling (`config.json` +
  `pytorch_model.bin`, models/WeightedMeanPooling.py:47-62), Normalize (models/Normalize.py), and Asym with
  one Transformer tower per key (models/Asym.py:62-122; train_bi-encoder_mnrl.py:139);
* the embedding pickle cache `{id: ndarray}` (beir_dense_retriever.py:311-348) and the results JSON
  `{qid: {doc_id: score}}` (:434-441).

Nothing here touches the GPU: the readers return plain descriptions that sgpt_amd.st / sgpt_amd.beir turn into
SGPTModel instances."""
import json
import os
import pickle
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

_POOL_FLAGS = {"pooling_mode_mean_tokens": "mean", "pooling_mode_weightedmean_tokens": "weightedmean",
               "pooling_mode_lasttoken": "lasttoken"}
_UNSUPPORTED_POOL_FLAGS = ("pooling_mode_cls_token", "pooling_mode_max_tokens", "pooling_mode_mean_sqrt_len_tokens")
_SBERT_CONFIG_NAMES = ("sentence_bert_config.json", "sentence_roberta_config.json", "sentence_distilbert_config.json",
                       "sentence_camembert_config.json", "sentence_albert_config.json",
                       "sentence_xlm-roberta_config.json", "sentence_xlnet_config.json")   # Transformer.py:168


@dataclass
class STFolder:
    """What a sentence-transformers folder says about the pipeline."""
    transformer_dirs: Dict[str, str]          # key -> folder of an HF checkpoint; {"": dir} for a symmetric model
    max_seq_length: Optional[int] = None
    do_lower_case: bool = False
    pooling_mode: str = "mean"
    position_weights_file: Optional[str] = None   # learntmean: <dir>/pytorch_model.bin holding `position_weights`
    normalize: bool = False
    modules: List[dict] = field(default_factory=list)

    @property
    def asymmetric(self) -> bool:
        return list(self.transformer_dirs) != [""]


def _read_json(path):
    with open(path) as f:
        return json.load(f)


def _transformer_cfg(folder: str) -> dict:
    for name in _SBERT_CONFIG_NAMES:
        p = os.path.join(folder, name)
        if os.path.exists(p):
            return _read_json(p)
    return {}


def pooling_mode_from_config(cfg: dict) -> str:
    """Pooling/config.json -> the single pooling mode the GPU path implements."""
    bad = [k for k in _UNSUPPORTED_POOL_FLAGS if cfg.get(k)]
    if bad:
        raise NotImplementedError(f"pooling flags {bad} are not used by any SGPT checkpoint")
    on = [mode for flag, mode in _POOL_FLAGS.items() if cfg.get(flag)]
    if len(on) != 1:
        raise NotImplementedError(f"exactly one pooling mode expected, config enables {on or 'none'} "
                                  "(concatenated poolings are not used by SGPT)")
    return on[0]


def read_st_folder(path: str) -> STFolder:
    mj = os.path.join(path, "modules.json")
    if not os.path.exists(mj):
        # plain HF checkpoint: the reference falls back to Transformer + MEAN pooling (SentenceTransformer.py:893-900)
        if not os.path.exists(os.path.join(path, "config.json")):
            raise FileNotFoundError(f"{path}: neither modules.json nor config.json")
        return STFolder({"": path}, pooling_mode="mean")
    modules = _read_json(mj)
    out = STFolder({}, modules=modules)
    for mod in sorted(modules, key=lambda m: m["idx"]):
        kind = mod["type"].rsplit(".", 1)[-1]
        folder = os.path.join(path, mod["path"])
        if kind == "Transformer":
            out.transformer_dirs[""] = folder
            tc = _transformer_cfg(folder)
            out.max_seq_length, out.do_lower_case = tc.get("max_seq_length"), bool(tc.get("do_lower_case", False))
        elif kind == "Asym":
            ac = _read_json(os.path.join(folder, "config.json"))
            for key, ids in ac["structure"].items():
                if len(ids) != 1 or ac["types"][ids[0]].rsplit(".", 1)[-1] != "Transformer":
                    raise NotImplementedError("Asym towers other than one Transformer per key are not used by SGPT")
                tower = os.path.join(folder, ids[0])
                out.transformer_dirs[key] = tower
                tc = _transformer_cfg(tower)
                out.max_seq_length = tc.get("max_seq_length", out.max_seq_length)
                out.do_lower_case = bool(tc.get("do_lower_case", out.do_lower_case))
        elif kind == "Pooling":
            out.pooling_mode = pooling_mode_from_config(_read_json(os.path.join(folder, "config.json")))
        elif kind == "WeightedMeanPooling":
            out.pooling_mode = "learntmean"
            out.position_weights_file = os.path.join(folder, "pytorch_model.bin")
        elif kind == "Normalize":
            out.normalize = True
        else:
            raise NotImplementedError(f"sentence-transformers module {mod['type']} is not part of the SGPT pipelines")
    if not out.transformer_dirs:
        raise ValueError(f"{path}: modules.json lists no Transformer")
    return out


def write_st_folder(path: str, hf_config: dict, state_dict: Dict[str, "np.ndarray"], pooling_mode: str = "weightedmean",
                    max_seq_length: int = 300, normalize: bool = False, position_weights=None) -> None:
    """Writes the layout SentenceTransformer.save produces for [Transformer, Pooling|WeightedMeanPooling, (Normalize)]
    (transformer in the root folder, :418-419), so a folder written here loads in the reference and vice versa.
    Tokenizer files are the caller's business (none exist offline)."""
    import torch
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(hf_config, f, indent=2)
    save_file({k: torch.as_tensor(np.asarray(v)).contiguous() for k, v in state_dict.items()},
              os.path.join(path, "model.safetensors"))
    with open(os.path.join(path, "sentence_bert_config.json"), "w") as f:
        json.dump({"max_seq_length": max_seq_length, "do_lower_case": False}, f, indent=2)
    modules = [{"idx": 0, "name": "0", "path": "", "type": "sentence_transformers.models.Transformer"}]
    d = hf_config.get("hidden_size", hf_config.get("n_embd"))
    if pooling_mode == "learntmean":
        sub = "1_WeightedMeanPooling"
        os.makedirs(os.path.join(path, sub), exist_ok=True)
        pw = torch.as_tensor(np.asarray(position_weights, dtype=np.float32))
        with open(os.path.join(path, sub, "config.json"), "w") as f:
            json.dump({"word_embedding_dimension": d, "position_start": 0, "num_positions": int(pw.numel()) - 1}, f, indent=2)
        torch.save({"position_weights": pw}, os.path.join(path, sub, "pytorch_model.bin"))
        modules.append({"idx": 1, "name": "1", "path": sub, "type": "sentence_transformers.models.WeightedMeanPooling"})
    else:
        if pooling_mode not in _POOL_FLAGS.values():
            raise ValueError(f"unknown pooling mode {pooling_mode}")
        sub = "1_Pooling"
        os.makedirs(os.path.join(path, sub), exist_ok=True)
        cfg = {"word_embedding_dimension": d, "pooling_mode_cls_token": False, "pooling_mode_mean_tokens": False,
               "pooling_mode_max_tokens": False, "pooling_mode_mean_sqrt_len_tokens": False,
               "pooling_mode_weightedmean_tokens": False, "pooling_mode_lasttoken": False}
        cfg[[k for k, v in _POOL_FLAGS.items() if v == pooling_mode][0]] = True
        with open(os.path.join(path, sub, "config.json"), "w") as f:
            json.dump(cfg, f, indent=2)
        modules.append({"idx": 1, "name": "1", "path": sub, "type": "sentence_transformers.models.Pooling"})
    if normalize:
        os.makedirs(os.path.join(path, "2_Normalize"), exist_ok=True)
        modules.append({"idx": 2, "name": "2", "path": "2_Normalize", "type": "sentence_transformers.models.Normalize"})
    with open(os.path.join(path, "modules.json"), "w") as f:
        json.dump(modules, f, indent=2)


# ---- embedding cache + results -----------------------------------------------------------------------------------
def save_embedding_cache(path: str, ids, embeddings) -> None:
    """`pickle.dump({id: ndarray})` of embed_batcher (beir_dense_retriever.py:306-312)."""
    emb = np.asarray(embeddings)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        pickle.dump({i: e for i, e in zip(ids, emb)}, f)


def load_embedding_cache(path: str, ids) -> np.ndarray:
    """Rows in the order of `ids` (encode_queries / encode_corpus, :319-323, 336-346)."""
    with open(path, "rb") as f:
        table = pickle.load(f)
    return np.array([table[i] for i in ids])


def save_results_json(path: str, results: Dict[str, Dict[str, float]]) -> None:
    """`json.dump(results)` of beir_dense_retriever.py:439-441: {qid: {doc_id: score}}, plain floats."""
    with open(path, "w") as f:
        json.dump({q: {d: float(s) for d, s in hits.items()} for q, hits in results.items()}, f)


def load_results_json(path: str) -> Dict[str, Dict[str, float]]:
    return _read_json(path)
