"""ctypes binding of the C ABI in include/sgpt_hip.h (libsgpt_hip.so, built in-tree by
sgpt_amd/build.py).  There is NO fallback: if the library is missing or fails to load,
every product entry point raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SGPT_HIP_LIB") or os.path.join(HERE, "lib", "libsgpt_hip.so")   # env: A/B builds of the same ABI

SGPT_F32, SGPT_BF16, SGPT_FP8W, SGPT_F16, SGPT_FP8M = 0, 1, 2, 3, 4
SGPT_ABI_VERSION = 8
SGPT_PREC_CLASSES = 5                      # precision-plan classes per block: LN1, ATT, CTX, LN2, H (include/sgpt_hip.h)
PC_LN1, PC_ATT, PC_CTX, PC_LN2, PC_H = 0, 1, 2, 3, 4
SGPT_ERR_RANGE = -5
SGPT_ERR_COMM = -6
SGPT_COMM_ID_BYTES = 128
SGPT_ARCH_GPTNEO, SGPT_ARCH_GPTJ, SGPT_ARCH_BLOOM = 0, 1, 2
POOL_MODES = {"weightedmean": 0, "mean": 1, "lasttoken": 2, "learntmean": 3}


class ModelDesc(C.Structure):
    _fields_ = [("arch", C.c_int32), ("n_layers", C.c_int32), ("d_model", C.c_int32), ("n_heads", C.c_int32),
                ("d_ffn", C.c_int32), ("vocab", C.c_int32), ("max_pos", C.c_int32), ("window", C.c_int32),
                ("ln_eps", C.c_float), ("attn_scale", C.c_float), ("compute_dtype", C.c_int32),
                ("layer_is_local", C.POINTER(C.c_uint8)), ("rotary_dim", C.c_int32), ("qk_split", C.c_int32),
                ("split_weights", C.c_int32)]


class TensorView(C.Structure):
    _fields_ = [("name", C.c_char_p), ("ptr", C.c_void_p), ("numel", C.c_int64)]


# name -> (restype, argtypes); mirrors include/sgpt_hip.h one to one
SIGNATURES = {
    "sgpt_abi_version": (C.c_int, []),
    "sgpt_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "sgpt_ctx_destroy": (None, [C.c_void_p]),
    "sgpt_last_error": (C.c_char_p, [C.c_void_p]),
    "sgpt_model_load": (C.c_int, [C.c_void_p, C.POINTER(ModelDesc), C.POINTER(TensorView), C.c_size_t,
                                  C.POINTER(C.c_void_p)]),
    "sgpt_model_free": (None, [C.c_void_p]),
    "sgpt_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                              C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                              C.c_void_p, C.c_void_p, C.c_void_p]),
    "sgpt_encode_layers": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "sgpt_model_set_pool_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "sgpt_pool_learnt": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    "sgpt_fp8_quantize_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                         C.c_void_p]),
    "sgpt_fp8_dequantize_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                           C.c_int32, C.c_void_p]),
    "sgpt_lm_logprobs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    "sgpt_pool": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                            C.c_int32, C.c_void_p, C.c_void_p]),
    "sgpt_l2_normalize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "sgpt_pairwise_scores": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "sgpt_f32_to_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "sgpt_f32_to_16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]),
    "sgpt_range_check": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.c_void_p]),
    "sgpt_ctx_generation": (C.c_uint64, [C.c_void_p]),
    "sgpt_ctx_reserve": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t]),
    "sgpt_scores": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int32,
                              C.c_void_p, C.c_int64, C.c_void_p]),
    "sgpt_score_topk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int32,
                                  C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32),
                                  C.c_void_p]),
    "sgpt_score_topk_refined": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int32,
                                           C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                           C.c_void_p]),
    "sgpt_topk_merge": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "sgpt_topk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int64,
                            C.c_void_p, C.c_void_p, C.c_void_p]),
    "sgpt_linear": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sgpt_linear_query": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sgpt_model_calibrate_begin": (C.c_int, [C.c_void_p]),
    "sgpt_model_calibrate_end": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p]),
    "sgpt_model_set_act_scales": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "sgpt_layernorm_fp8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "sgpt_linear_fp8": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sgpt_comm_unique_id": (C.c_int, [C.c_void_p]),
    "sgpt_comm_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "sgpt_comm_destroy": (C.c_int, [C.c_void_p]),
    "sgpt_comm_world": (C.c_int32, [C.c_void_p]),
    "sgpt_comm_rank": (C.c_int32, [C.c_void_p]),
    "sgpt_allgather_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "sgpt_allgather_rows_padded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "sgpt_exchange_topk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    "sgpt_fold_gathered_topk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sgpt_ctx_set_low_latency": (C.c_int32, [C.c_void_p, C.c_int32]),
    "sgpt_ctx_set_tile_policy": (C.c_int32, [C.c_void_p, C.c_int32]),
    "sgpt_ctx_set_gemm_cu_cap": (C.c_int32, [C.c_void_p, C.c_int32]),
    "sgpt_model_range_check": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.c_void_p]),
    "sgpt_model_range_adapt": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    "sgpt_model_get_range_shifts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "sgpt_model_set_range_shifts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "sgpt_model_set_precision": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "sgpt_model_get_precision": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "sgpt_model_release_split_weights": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sgpt_model_precision_probe_begin": (C.c_int, [C.c_void_p]),
    "sgpt_model_precision_probe_end": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sgpt_row_crest": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int64, C.POINTER(C.c_float), C.c_void_p]),
    "sgpt_split16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "sgpt_linear_split": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sgpt_bench_gemm": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_int32, C.POINTER(C.c_float)]),
    "sgpt_prof_enable": (C.c_int, [C.c_void_p, C.c_int32]),
    "sgpt_prof_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                 C.c_int32]),
}

_lib = None


class SgptHipError(RuntimeError):
    pass


class SgptRangeError(SgptHipError):
    """dtype='f16': a weight is outside the IEEE-half range, or an activation class is beyond every power-of-two range
    shift (use dtype='bf16'); dtype='fp8mfma': an e4m3 code saturated (re-calibrate)."""

# symbols that exist only in the experiment build (SGPT_EXPERIMENTS=1 python -m sgpt_amd.build -> libsgpt_hip_exp.so,
# selected with SGPT_HIP_LIB): A/B knobs of scripts/, never part of the product ABI
EXPERIMENT_SIGNATURES = {
    "sgpt_exp_set_gemm_skew": (C.c_int32, [C.c_int32]),
    "sgpt_exp_set_gemm_w": (C.c_int32, [C.c_int32]),
}


def load():
    """dlopen libsgpt_hip.so and declare every prototype.  Raises (never falls back)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SgptHipError(f"{LIB_PATH} is missing: build it with `python -m sgpt_amd.build` "
                           "(hipcc --offload-arch=gfx950); there is no CPU/PyTorch fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in EXPERIMENT_SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
    if lib.sgpt_abi_version() != SGPT_ABI_VERSION:
        raise SgptHipError("libsgpt_hip.so ABI version mismatch: rebuild with `python -m sgpt_amd.build --force`")
    _lib = lib
    return lib


def check(ctx_handle, status, what=""):
    if status == 0:
        return
    msg = load().sgpt_last_error(ctx_handle)
    msg = msg.decode() if msg else ""
    if status == -1:
        raise ValueError(f"{what}: {msg}")
    if status == SGPT_ERR_RANGE:
        raise SgptRangeError(f"{what}: {msg}")
    raise SgptHipError(f"{what}: status {status}: {msg}")
