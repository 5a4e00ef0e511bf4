"""Thin Python host over the C ABI: a `Context` per GPU.  PyTorch is used only to own device
memory (torch.empty / .data_ptr()) and to name the current HIP stream -- every computation is
a call into libsgpt_hip.so.  No torch compute op stands in for a kernel here."""
import ctypes as C
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import SGPT_BF16, SGPT_F16, SGPT_F32, POOL_MODES

# torch dtype <-> element-type code of include/sgpt_hip.h
DT_CODE = {torch.float32: SGPT_F32, torch.bfloat16: SGPT_BF16, torch.float16: SGPT_F16}


def _stream_ptr(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _p(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


_contexts = {}


def get_context(device=None) -> "Context":
    """One Context per HIP device per process (one process per GPU in multi-GPU runs)."""
    if not torch.cuda.is_available():
        raise _lib.SgptHipError("no HIP device visible: sgpt_amd runs its hot path only as gfx950 kernels "
                                "(there is no CPU path)")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type != "cuda":
        raise _lib.SgptHipError(f"sgpt_amd needs a HIP device, got {dev}")
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _contexts:
        _contexts[idx] = Context(idx)
    return _contexts[idx]


class Context:
    def __init__(self, device_index: int):
        self.lib = _lib.load()
        self.device = torch.device("cuda", device_index)
        h = C.c_void_p()
        st = self.lib.sgpt_ctx_create(device_index, C.byref(h))
        if st != 0:
            raise _lib.SgptHipError(f"sgpt_ctx_create(device={device_index}) failed with status {st}")
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self.lib.sgpt_ctx_destroy(self.handle)
            self.handle = None

    def _chk(self, st, what):
        _lib.check(self.handle, st, what)

    # ---- helpers ----
    def _dev_f32(self, a) -> torch.Tensor:
        """Tensor / ndarray / list -> contiguous fp32 tensor on this device (the coercions of
        util.cos_sim, sentence_transformers/util.py:29-39)."""
        if not isinstance(a, torch.Tensor):
            a = torch.as_tensor(np.asarray(a))
        if a.dim() == 1:
            a = a.unsqueeze(0)
        return a.to(device=self.device, dtype=torch.float32).contiguous()

    # ---- a4: stand-alone pooling ----
    def pool(self, hidden: torch.Tensor, mask: torch.Tensor, mode: str = "weightedmean",
             position_weights: Optional[torch.Tensor] = None) -> torch.Tensor:
        """mode 'learntmean' takes `position_weights` (WeightedMeanPooling.py:21-39), indexed by the padded position."""
        if mode not in POOL_MODES:
            raise ValueError(f"unknown pooling mode {mode}")
        hidden = hidden.to(self.device)
        if hidden.dtype not in DT_CODE:
            hidden = hidden.float()
        hidden = hidden.contiguous()
        B, S, d = hidden.shape
        m = mask.to(device=self.device, dtype=torch.int32).contiguous()
        out = torch.empty((B, d), dtype=torch.float32, device=self.device)
        dt = DT_CODE[hidden.dtype]
        if mode == "learntmean":
            if position_weights is None or position_weights.numel() < S:
                raise ValueError("learntmean needs position_weights covering the sequence length")
            pw = position_weights.to(device=self.device, dtype=torch.float32).contiguous()
            self._chk(self.lib.sgpt_pool_learnt(self.handle, _p(hidden), dt, _p(m), B, S, d, _p(pw), _p(out),
                                                _stream_ptr(self.device)), "sgpt_pool_learnt")
            return out
        self._chk(self.lib.sgpt_pool(self.handle, _p(hidden), dt, _p(m), B, S, d, POOL_MODES[mode], _p(out),
                                     _stream_ptr(self.device)), "sgpt_pool")
        return out

    # ---- normalise / convert ----
    def l2_normalize(self, x: torch.Tensor, out_dtype=torch.float32) -> torch.Tensor:
        x = self._dev_f32(x)
        n, d = x.shape
        out = torch.empty((n, d), dtype=out_dtype, device=self.device)
        self._chk(self.lib.sgpt_l2_normalize(self.handle, _p(x), n, d, _p(out), DT_CODE[out_dtype],
                                             _stream_ptr(self.device)), "sgpt_l2_normalize")
        return out

    def pairwise_scores(self, a: torch.Tensor, b: torch.Tensor, cosine: bool) -> torch.Tensor:
        """out[i] = dot(a[i], b[i]) (cosine: of the L2-normalised rows): include/sgpt_hip.h::sgpt_pairwise_scores."""
        a, b = self._dev_f32(a), self._dev_f32(b)
        if a.shape != b.shape or a.dim() != 2:
            raise ValueError(f"pairwise scores need two [n, d] matrices of one shape, got {tuple(a.shape)} and {tuple(b.shape)}")
        n, d = a.shape
        out = torch.empty((n,), dtype=torch.float32, device=self.device)
        if n:
            self._chk(self.lib.sgpt_pairwise_scores(self.handle, _p(a), _p(b), n, d, 1 if cosine else 0, _p(out),
                                                    _stream_ptr(self.device)), "sgpt_pairwise_scores")
        return out

    def to_16(self, x: torch.Tensor, dtype=torch.float16, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """fp32 -> bf16 / f16 (RNE) on the device: the scorer's 16-bit operand format."""
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        if out is None:
            out = torch.empty(x.shape, dtype=dtype, device=self.device)
        self._chk(self.lib.sgpt_f32_to_16(self.handle, _p(x), x.numel(), _p(out), DT_CODE[dtype],
                                          _stream_ptr(self.device)), "sgpt_f32_to_16")
        return out

    def row_crest(self, x: torch.Tensor) -> float:
        """max over the rows of max|v| / rms(v) (include/sgpt_hip.h::sgpt_row_crest; fp32 rows are rounded to f16 first).
        ~3.5-5 for well-spread embedding rows, ~sqrt(d / 2) when two channels carry the row.  Syncs the stream."""
        if x.dtype not in (torch.float16, torch.bfloat16):
            x = self.to_16(self._dev_f32(x), torch.float16)
        x = x.to(self.device).contiguous()
        n, d = x.shape
        out = C.c_float(0)
        self._chk(self.lib.sgpt_row_crest(self.handle, _p(x), DT_CODE[x.dtype], n, d, d, C.byref(out),
                                          _stream_ptr(self.device)), "sgpt_row_crest")
        return float(out.value)

    def split16(self, x: torch.Tensor, role: str, dtype=torch.float16) -> torch.Tensor:
        """fp32 rows [n, d] -> split-precision 16-bit rows [n, 3 d] for the scorer (include/sgpt_hip.h::sgpt_split16):
        role 'doc' = [hi | lo | hi], role 'query' = [hi | hi | lo]; `scores` / `score_topk` over the 3 d columns then give
        q_hi.c_hi + q_hi.c_lo + q_lo.c_hi -- cosine scores to ~1e-6 on the 16-bit scorer kernels, for embeddings that a few
        channels dominate (the plain 16-bit corpus format alone costs those up to 4e-4)."""
        if role not in ("doc", "query"):
            raise ValueError("role must be 'doc' or 'query'")
        x = self._dev_f32(x)
        n, d = x.shape
        out = torch.empty((n, 3 * d), dtype=dtype, device=self.device)
        self._chk(self.lib.sgpt_split16(self.handle, _p(x), n, d, 0 if role == "doc" else 1, _p(out), DT_CODE[dtype],
                                        _stream_ptr(self.device)), "sgpt_split16")
        return out

    def to_bf16(self, x: torch.Tensor) -> torch.Tensor:
        return self.to_16(x, torch.bfloat16)

    def range_check(self, reset: bool = True) -> int:
        """Range guard of the ctx-level stand-alone ops (`linear` with f16 output; syncs the stream): bit 0 = an f16 value
        left the half range since the last reset.  Models carry their own guard word (SGPTModel.check_range)."""
        flagged = C.c_int32(0)
        self._chk(self.lib.sgpt_range_check(self.handle, C.byref(flagged), 1 if reset else 0, _stream_ptr(self.device)),
                  "sgpt_range_check")
        return int(flagged.value)

    def generation(self) -> int:
        """Changes when a library-owned buffer captured graphs point into was re-allocated (EncodeGraph)."""
        return int(self.lib.sgpt_ctx_generation(self.handle))

    def set_low_latency(self, on: bool) -> bool:
        """Per context: k-groups for query-sized GEMM launches (include/sgpt_hip.h::sgpt_ctx_set_low_latency): ~1 % off a
        16-query encode since round 3 (16 % before), at the price of bit-identical embeddings across batch sizes.  Returns the previous setting."""
        return bool(self.lib.sgpt_ctx_set_low_latency(self.handle, 1 if on else 0))

    def set_gemm_cu_cap(self, n: int) -> int:
        """Per context: at most n workgroups per launch of the persistent 256x256 projection kernel (0 = one per CU), for two
        contexts pipelined on two streams (include/sgpt_hip.h::sgpt_ctx_set_gemm_cu_cap).  Returns the previous value."""
        return int(self.lib.sgpt_ctx_set_gemm_cu_cap(self.handle, int(n)))

    def set_tile_policy(self, force_256) -> int:
        """Per context: True / 1 = keep the 256x256 LDS-DMA GEMM tiles even where the small-tile rule would apply (kernel tests of
        single-tile shapes); 2 = no query- / mid-sized kernels (csrc/qgemm.hip): the bulk path's small-tile kernels on every layout
        (A/Bs); identical bits whatever the policy.  Returns the previous policy (0 | 1 | 2)."""
        return int(self.lib.sgpt_ctx_set_tile_policy(self.handle, int(force_256)))

    def reserve(self, encode_bytes: int = 0, score_bytes: int = 0) -> None:
        self._chk(self.lib.sgpt_ctx_reserve(self.handle, encode_bytes, score_bytes), "sgpt_ctx_reserve")

    # ---- fp8 (e4m3fn, power-of-two per-row scales) weight storage: building blocks of dtype="fp8" models ----
    def fp8_quantize_rows(self, w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        w = w.to(device=self.device, dtype=torch.float32).contiguous()
        rows, cols = w.shape
        codes = torch.empty((rows, cols), dtype=torch.uint8, device=self.device)
        scale = torch.empty((rows,), dtype=torch.float32, device=self.device)
        self._chk(self.lib.sgpt_fp8_quantize_rows(self.handle, _p(w), rows, cols, _p(codes), _p(scale),
                                                  _stream_ptr(self.device)), "sgpt_fp8_quantize_rows")
        return codes, scale

    def fp8_dequantize_rows(self, codes: torch.Tensor, scale: torch.Tensor, out_dtype=torch.float32) -> torch.Tensor:
        codes = codes.to(device=self.device, dtype=torch.uint8).contiguous()
        scale = scale.to(device=self.device, dtype=torch.float32).contiguous()
        rows, cols = codes.shape
        out = torch.empty((rows, cols), dtype=out_dtype, device=self.device)
        self._chk(self.lib.sgpt_fp8_dequantize_rows(self.handle, _p(codes), _p(scale), rows, cols, _p(out),
                                                    DT_CODE[out_dtype],
                                                    _stream_ptr(self.device)), "sgpt_fp8_dequantize_rows")
        return out

    def _operand(self, x: torch.Tensor, dtype) -> torch.Tensor:
        if x.dtype == dtype and x.device == self.device and x.is_contiguous():
            return x
        if dtype in (torch.bfloat16, torch.float16):
            return self.to_16(x, dtype)
        return x.to(device=self.device, dtype=torch.float32).contiguous()

    # ---- the projection GEMM with a fused epilogue (kernel-level tests, custom blocks) ----
    def linear(self, a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, epi: str = "store",
               resid: Optional[torch.Tensor] = None, out_dtype=None) -> torch.Tensor:
        """epi: 'store' | 'gelu' (gelu_new(a.w^T + bias)) | 'resid' (resid + a.w^T + bias, fp32) | 'vt' (transposed)."""
        code = {"store": 0, "gelu": 1, "resid": 2, "vt": 4}[epi]
        M, K = a.shape
        N = w.shape[0]
        if out_dtype is None:
            out_dtype = torch.float32 if epi == "resid" else a.dtype
        out = torch.empty((N, M) if epi == "vt" else (M, N), dtype=out_dtype, device=self.device)
        b = None if bias is None else bias.to(device=self.device, dtype=torch.float32).contiguous()
        r = None if resid is None else resid.to(device=self.device, dtype=torch.float32).contiguous()
        self._chk(self.lib.sgpt_linear(self.handle, DT_CODE[a.dtype], code, DT_CODE[out_dtype], _p(a.contiguous()),
                                       _p(w.contiguous()), _p(b), _p(r), _p(out), M, N, K, _stream_ptr(self.device)),
                  "sgpt_linear")
        return out

    def linear_query(self, w: torch.Tensor, a: Optional[torch.Tensor] = None, x: Optional[torch.Tensor] = None, ln=None,
                     bias: Optional[torch.Tensor] = None, epi: str = "store", resid: Optional[torch.Tensor] = None, n_split: int = 0):
        """The query- / mid-sized projection kernels stand-alone (include/sgpt_hip.h::sgpt_linear_query).  a: [M, K] 16-bit operand, or
        x: fp32 [M, K] with ln = (gamma, beta, eps) -- the LayerNorm prologue.  epi 'store' | 'gelu' | 'resid' | 'qkv' (returns
        (q|k [M, n_split], V^T [N - n_split, M]))."""
        code = {"store": 0, "gelu": 1, "resid": 2, "qkv": 7}[epi]
        N, K = w.shape
        src = a if a is not None else x
        M = src.shape[0]
        odt = torch.float32 if epi == "resid" else w.dtype
        out = torch.empty((M, n_split if epi == "qkv" else N), dtype=odt, device=self.device)
        vt = torch.empty((N - n_split, M), dtype=odt, device=self.device) if epi == "qkv" else None
        f32 = lambda t_: None if t_ is None else t_.to(device=self.device, dtype=torch.float32).contiguous()  # noqa: E731
        g, b, eps = (f32(ln[0]), f32(ln[1]), float(ln[2])) if ln is not None else (None, None, 0.0)
        b32, r32, x32 = f32(bias), f32(resid), f32(x)
        self._chk(self.lib.sgpt_linear_query(self.handle, DT_CODE[w.dtype], code, _p(None if a is None else a.contiguous()), _p(x32), _p(g), _p(b), eps,
                                             _p(w.contiguous()), _p(b32), _p(r32), _p(out), _p(vt), int(n_split), M, N, K,
                                             _stream_ptr(self.device)), "sgpt_linear_query")
        return (out, vt) if epi == "qkv" else out

    def linear_split(self, a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, epi: str = "store",
                     triple: bool = False) -> torch.Tensor:
        """The split store epilogues (include/sgpt_hip.h::sgpt_linear_split): epi 'store' | 'gelu' -> [M, 2 N] = [hi | lo], or
        with triple=True [M, 3 N] = [hi | lo | hi] (the row layout a split consumer contracts over); 'vt' -> [2, N, M]
        (hi, lo transposed)."""
        code = {"store": 0, "gelu": 1, "vt": 4}[epi]
        M, K = a.shape
        N = w.shape[0]
        b = None if bias is None else bias.to(device=self.device, dtype=torch.float32).contiguous()
        if epi == "vt":
            out = torch.zeros((2, N, M), dtype=a.dtype, device=self.device)
            ldo, lo, hi2 = M, N * M, 0
        else:
            out = torch.zeros((M, (3 if triple else 2) * N), dtype=a.dtype, device=self.device)
            ldo, lo, hi2 = out.shape[1], N, (2 * N if triple else 0)
        self._chk(self.lib.sgpt_linear_split(self.handle, DT_CODE[a.dtype], code, _p(a.contiguous()), _p(w.contiguous()), _p(b),
                                             _p(out), ldo, lo, hi2, M, N, K, _stream_ptr(self.device)), "sgpt_linear_split")
        return out

    # ---- fp8-MFMA building blocks (dtype='fp8mfma'): quantising LayerNorm, e4m3 x e4m3 projection ----
    def layernorm_fp8(self, x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5):
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        T, d = x.shape
        codes = torch.empty((T, d), dtype=torch.uint8, device=self.device)
        scale = torch.empty((T,), dtype=torch.float32, device=self.device)
        g = gamma.to(device=self.device, dtype=torch.float32).contiguous()
        b = beta.to(device=self.device, dtype=torch.float32).contiguous()
        self._chk(self.lib.sgpt_layernorm_fp8(self.handle, _p(x), _p(g), _p(b), T, d, eps, _p(codes), _p(scale),
                                              _stream_ptr(self.device)), "sgpt_layernorm_fp8")
        return codes, scale

    def linear_fp8(self, a8: torch.Tensor, w8: torch.Tensor, w_scale: torch.Tensor, bias: torch.Tensor, epi: str,
                   a_scale: Optional[torch.Tensor] = None, a_scalar: float = 1.0, resid: Optional[torch.Tensor] = None,
                   out_scale: float = 1.0, out_dtype=None) -> torch.Tensor:
        """epi 'gelu': u8 codes of gelu_new(acc + bias) / out_scale;  'resid': fp32 resid + acc + bias;
        'store' / 'vt': 16-bit (out_dtype) acc (+ bias), row-major / transposed."""
        M, K = a8.shape
        N = w8.shape[0]
        code = {"store": 0, "gelu": 1, "resid": 2, "vt": 4}[epi]
        odt = {"gelu": torch.uint8, "resid": torch.float32}.get(epi, out_dtype if out_dtype is not None else torch.bfloat16)
        out = torch.empty((N, M) if epi == "vt" else (M, N), dtype=odt, device=self.device)
        r = None if resid is None else resid.to(device=self.device, dtype=torch.float32).contiguous()
        b = None if bias is None else bias.contiguous()
        self._chk(self.lib.sgpt_linear_fp8(self.handle, code, DT_CODE.get(odt, 0), _p(a8.contiguous()), _p(a_scale), a_scalar,
                                           _p(w8.contiguous()), _p(w_scale.contiguous()), _p(b), _p(r), _p(out),
                                           out_scale, M, N, K, _stream_ptr(self.device)), "sgpt_linear_fp8")
        return out

    # ---- a7: dense score matrix (cos_sim / dot_score) ----
    def scores(self, a: torch.Tensor, b: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
        a, b = self._operand(a, dtype), self._operand(b, dtype)
        na, d = a.shape
        nb, d2 = b.shape
        if d != d2:
            raise ValueError(f"embedding dims differ: {d} vs {d2}")
        ldo = (nb + 3) // 4 * 4
        out = torch.empty((na, ldo), dtype=torch.float32, device=self.device)
        self._chk(self.lib.sgpt_scores(self.handle, _p(a), _p(b), DT_CODE[dtype],
                                       na, nb, d, _p(out), ldo, _stream_ptr(self.device)), "sgpt_scores")
        return out[:, :nb]

    # ---- a7+a8: fused chunked score + running top-k ----
    def score_topk(self, q: torch.Tensor, corpus: torch.Tensor, k: int, idx_base: int = 0,
                   run: Optional[Tuple[torch.Tensor, torch.Tensor, int]] = None,
                   dtype=None) -> Tuple[torch.Tensor, torch.Tensor, int]:
        """-> (values fp32[nq,k], indices int64[nq,k], n_valid); rows sorted by descending score."""
        if dtype is None:
            dtype = corpus.dtype if corpus.dtype in DT_CODE else torch.float32
        q, corpus = self._operand(q, dtype), self._operand(corpus, dtype)
        nq, d = q.shape
        N, d2 = corpus.shape
        if d != d2:
            raise ValueError(f"embedding dims differ: {d} vs {d2}")
        if run is None:
            val = torch.empty((nq, k), dtype=torch.float32, device=self.device)
            idx = torch.empty((nq, k), dtype=torch.int64, device=self.device)
            n_run = 0
        else:
            val, idx, n_run = run
        n_out = C.c_int32(0)
        self._chk(self.lib.sgpt_score_topk(self.handle, _p(q), _p(corpus), DT_CODE[dtype], nq, N, d, k,
                                           idx_base, _p(val), _p(idx), n_run, C.byref(n_out),
                                           _stream_ptr(self.device)), "sgpt_score_topk")
        return val, idx, int(n_out.value)

    REFINE_MARGIN_UNIT_F16 = 2.5e-3      # 2 eps for L2-normalised rows with IEEE-half stage-1 copies (include/sgpt_hip.h)

    def score_topk_refined(self, q: torch.Tensor, corpus32: torch.Tensor, corpus16: Optional[torch.Tensor], k: int,
                           idx_base: int = 0, run: Optional[Tuple[torch.Tensor, torch.Tensor, int]] = None,
                           margin: Optional[float] = None, report: bool = False):
        """The fp32 top-k (fp32 scores, the fp32 set) at the 16-bit scorer's speed: sgpt_score_topk_refined.  q, corpus32: fp32
        rows (L2-normalised for the default margin); corpus16: their f16 rounding (None: made here).  report=True synchronises
        and returns a fourth value: 1 if the exact fallback pass ran, 0 if not, -1 if the call was the exact pass outright."""
        q = q.to(device=self.device, dtype=torch.float32).contiguous()
        corpus32 = corpus32.to(device=self.device, dtype=torch.float32).contiguous()
        if corpus16 is None:
            corpus16 = self.to_16(corpus32, torch.float16)
        if corpus16.dtype not in (torch.float16, torch.bfloat16) or corpus16.shape != corpus32.shape or not corpus16.is_contiguous():
            raise ValueError("corpus16 must be the contiguous 16-bit rounding of corpus32")
        if margin is None:
            if corpus16.dtype != torch.float16:
                raise ValueError("the default margin is the bound for IEEE-half copies of unit rows: pass margin= for bf16 copies")
            margin = self.REFINE_MARGIN_UNIT_F16
        nq, d = q.shape
        N = corpus32.shape[0]
        if run is None:
            val = torch.empty((nq, k), dtype=torch.float32, device=self.device)
            idx = torch.empty((nq, k), dtype=torch.int64, device=self.device)
            n_run = 0
        else:
            val, idx, n_run = run
        n_out, fb = C.c_int32(0), C.c_int32(0)
        self._chk(self.lib.sgpt_score_topk_refined(self.handle, _p(q), _p(corpus32), _p(corpus16), DT_CODE[corpus16.dtype], nq, N, d, k,
                                                   idx_base, float(margin), _p(val), _p(idx), n_run, C.byref(n_out),
                                                   C.byref(fb) if report else None, _stream_ptr(self.device)), "sgpt_score_topk_refined")
        return (val, idx, int(n_out.value), int(fb.value)) if report else (val, idx, int(n_out.value))

    def topk_merge(self, val: torch.Tensor, idx: torch.Tensor, k: int,
                   exclude_idx: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        val = val.to(device=self.device, dtype=torch.float32).contiguous()
        idx = idx.to(device=self.device, dtype=torch.int64).contiguous()
        nq, m = val.shape
        ov = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        oi = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        ex = None if exclude_idx is None else exclude_idx.to(device=self.device, dtype=torch.int64).contiguous()
        self._chk(self.lib.sgpt_topk_merge(self.handle, _p(val), _p(idx), nq, m, k, _p(ex), _p(ov), _p(oi),
                                           _stream_ptr(self.device)), "sgpt_topk_merge")
        return ov, oi

    def fold_gathered_topk(self, gathered_val: torch.Tensor, gathered_idx: torch.Tensor, k_out: int,
                           exclude_idx: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """[world, nq, k] all-gathered per-rank lists (rank order) -> the k_out best per query: the rank-local half of
        sgpt_exchange_topk (include/sgpt_hip.h::sgpt_fold_gathered_topk), no communicator involved."""
        gv = gathered_val.to(device=self.device, dtype=torch.float32).contiguous()
        gi = gathered_idx.to(device=self.device, dtype=torch.int64).contiguous()
        world, nq, k = gv.shape
        ov = torch.empty((nq, k_out), dtype=torch.float32, device=self.device)
        oi = torch.empty((nq, k_out), dtype=torch.int64, device=self.device)
        ex = None if exclude_idx is None else exclude_idx.to(device=self.device, dtype=torch.int64).contiguous()
        self._chk(self.lib.sgpt_fold_gathered_topk(self.handle, _p(gv), _p(gi), world, nq, k, k_out, _p(ex), _p(ov), _p(oi),
                                                   _stream_ptr(self.device)), "sgpt_fold_gathered_topk")
        return ov, oi

    def topk(self, scores: torch.Tensor, k: int, idx_base: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
        """torch.topk(scores, k, dim=1) with NaN -> -1 first (exact_search.py:99-108); sorted descending."""
        scores = scores.to(device=self.device, dtype=torch.float32)
        if scores.stride(1) != 1:
            scores = scores.contiguous()
        nq, n = scores.shape
        ov = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        oi = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        self._chk(self.lib.sgpt_topk(self.handle, _p(scores), nq, n, scores.stride(0), k, idx_base, _p(ov), _p(oi),
                                     _stream_ptr(self.device)), "sgpt_topk")
        return ov, oi

    # ---- measurement hooks (bench.py) ----
    def prof_enable(self, on: bool):
        self._chk(self.lib.sgpt_prof_enable(self.handle, 1 if on else 0), "sgpt_prof_enable")

    def prof_read(self, reset=True):
        n, ms, fl = C.c_int64(0), C.c_double(0), C.c_double(0)
        self._chk(self.lib.sgpt_prof_read(self.handle, C.byref(n), C.byref(ms), C.byref(fl), 1 if reset else 0),
                  "sgpt_prof_read")
        return int(n.value), float(ms.value), float(fl.value)
