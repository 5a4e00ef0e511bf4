// Shared device helpers for the gfx950 kernels (wave = 64 lanes, MFMA 16x16 fragments).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 bits in memory
// raw IEEE binary16 bits in memory.  A distinct type (not a second typedef of uint16_t) so that kernels templated on the
// 16-bit operand format pick the matching conversion / MFMA; pointer arithmetic on either 16-bit type is the same.
struct f16_t { uint16_t bits; };
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// element-type codes of include/sgpt_hip.h
constexpr int DT_F32 = 0, DT_BF16 = 1, DT_FP8W = 2, DT_F16 = 3;
__host__ __device__ constexpr bool dt_is16(int dt) { return dt == DT_BF16 || dt == DT_F16; }

#define WAVE 64

// LDS scratch is written and re-read through different vector types: exempt from strict aliasing
typedef uint2 __attribute__((may_alias)) uint2_a;
typedef uint4 __attribute__((may_alias)) uint4_a;
typedef float4 __attribute__((may_alias)) float4_a;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// fp32 -> bf16, round-to-nearest-even: one v_cvt_pk_bf16_f32 (gfx950), same result as
// torch.Tensor.to(bfloat16) (checked bit-for-bit in tests/test_gpu_kernels.py::test_l2_normalize)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.0f) & 0xffffu); }

// fp32 -> f16, round-to-nearest-even (v_cvt_f16_f32 under the default rounding mode; NOT v_cvt_pkrtz), bit-identical to
// torch.Tensor.to(float16).  Overflow gives +-inf: every kernel that rounds activations to f16 tracks max|v| and raises
// the context's range flag (see RangeTrack) so a model whose activations leave the f16 range fails loudly.
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}

// four fp32 -> four OCP e4m3fn codes (RNE, v_cvt_pk_fp8_f32), saturating at +-448
__device__ __forceinline__ uint32_t pack_fp8x4(float a, float b, float c, float d) {
    a = __builtin_fminf(__builtin_fmaxf(a, -448.f), 448.f); b = __builtin_fminf(__builtin_fmaxf(b, -448.f), 448.f);
    c = __builtin_fminf(__builtin_fmaxf(c, -448.f), 448.f); d = __builtin_fminf(__builtin_fmaxf(d, -448.f), 448.f);
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (uint32_t)w;
}

// 16-bit operand format traits: H = bf16_t | f16_t
template <typename H> struct Half;
template <> struct Half<bf16_t> {
    static constexpr bool is_f16 = false;
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return pack_bf16x2(lo, hi); }
    static __device__ __forceinline__ float lo(uint32_t u) { return __uint_as_float(u << 16); }
    static __device__ __forceinline__ float hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
    static __device__ __forceinline__ f32x4 mfma16(const uint4& a, const uint4& b, const f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(const uint4& a, const uint4& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Half<f16_t> {
    static constexpr bool is_f16 = true;
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return pack_f16x2(lo, hi); }
    static __device__ __forceinline__ float lo(uint32_t u) {
        return (float)__builtin_bit_cast(f16x2_t, u)[0];
    }
    static __device__ __forceinline__ float hi(uint32_t u) {
        return (float)__builtin_bit_cast(f16x2_t, u)[1];
    }
    static __device__ __forceinline__ f32x4 mfma16(const uint4& a, const uint4& b, const f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(const uint4& a, const uint4& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
template <typename H> __device__ __forceinline__ uint16_t f32_to_h(float f) { return (uint16_t)(Half<H>::pack2(f, 0.0f) & 0xffffu); }

// f16 range guard.  Kernels that round fp32 activations to f16 feed every value through note(); finish() raises the
// context's device flag when a magnitude reached RANGE_LIMIT (half of the f16 maximum: head-room for the GPT-J rotary
// rotation, which can grow a pair by sqrt(2)).  bf16 has the fp32 exponent range: the tracker compiles to nothing.
constexpr float RANGE_LIMIT = 32768.0f;
template <typename H> struct RangeTrack {
    float amax = 0.f;
    __device__ __forceinline__ void note(float a, float b) {
        if constexpr (Half<H>::is_f16) amax = fmaxf(amax, fmaxf(fabsf(a), fabsf(b)));   // v_max3_f32 with |.| modifiers
    }
    // amax_slot (optional): the offending magnitude (fp32 bits of a non-negative float: unsigned order == float order) is
    // folded into the slot of this launch's (block, operand class) so that the host can raise that class's power-of-two
    // down-shift by the right amount and re-run (sgpt_model_range_adapt) instead of giving up on the checkpoint
    __device__ __forceinline__ void finish(int* flag, unsigned* amax_slot = nullptr) const {
        if constexpr (Half<H>::is_f16) {
            if (flag != nullptr && !(amax < RANGE_LIMIT)) {                             // also catches +-inf
                atomicOr(flag, 1);
                if (amax_slot != nullptr) atomicMax(amax_slot, __float_as_uint(amax));
            }
        }
    }
};

// 16-byte global store.  NT = non-temporal (`global_store_dwordx4 ... nt`): for streaming outputs the producing
// kernel never re-reads.  Measured on the GEMM epilogues: plain stores write-allocate in the 4 MiB XCD L2 and
// evict the activation / weight panels the k-loop is sharing (QK projection 369 -> 297 us with nt).
template <bool NT>
__device__ __forceinline__ void gstore16(void* ptr, uint4 v) {
    if constexpr (NT) {
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        const u32x4_t w = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(w, reinterpret_cast<u32x4_t*>(ptr));
    } else {
        *reinterpret_cast<uint4*>(ptr) = v;
    }
}

template <bool NT>
__device__ __forceinline__ uint4 ldg16u(const void* ptr) {
    // dword-aligned source (sequences start on even rows of the token axis: a V^T tile row begins on a 4-byte boundary, not a
    // 16-byte one): the vector type carries aligned(4), so the compiler may not assume more; global_load_dwordx4 needs no more
    typedef uint32_t u32x4v_t __attribute__((ext_vector_type(4), aligned(4)));
    u32x4v_t w;
    if constexpr (NT) w = __builtin_nontemporal_load(reinterpret_cast<const u32x4v_t*>(ptr));
    else w = *reinterpret_cast<const u32x4v_t*>(ptr);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

template <bool NT>
__device__ __forceinline__ float4 ldg16(const float* ptr) {
    if constexpr (NT) {
        typedef float f32x4v_t __attribute__((ext_vector_type(4)));
        const f32x4v_t w = __builtin_nontemporal_load(reinterpret_cast<const f32x4v_t*>(ptr));
        return make_float4(w[0], w[1], w[2], w[3]);
    } else {
        return *reinterpret_cast<const float4*>(ptr);
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// wave_sum with the partner values fetched by VALU lane exchanges (DPP quad permutes / row rotate, v_permlane16/32_swap) instead
// of six ds_bpermute round trips through the LDS crossbar: the SAME butterfly (partners lane ^ 32, 16, 8, 4, 2, 1 in that order),
// and a + b == b + a bit for bit, so the result equals wave_sum's on every lane.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum_valu(float v) {
    { const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
      v = __uint_as_float(r[0]) + __uint_as_float(r[1]); }                         // lane ^ 32
    { const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
      v = __uint_as_float(r[0]) + __uint_as_float(r[1]); }                         // lane ^ 16
    v += dpp_mov<0x128>(v);                                                         // row_ror:8 = lane ^ 8 within the 16-lane row
    v += dpp_mov<0x1B>(dpp_mov<0x141>(v));                                          // row_half_mirror then quad reverse = lane ^ 4
    v += dpp_mov<0x4E>(v);                                                          // quad_perm [2,3,0,1] = lane ^ 2
    v += dpp_mov<0xB1>(v);                                                          // quad_perm [1,0,3,2] = lane ^ 1
    return v;
}
// R independent wave sums, the butterfly steps interleaved across the values (step k of every value before step k + 1 of any): each
// value goes through exactly wave_sum_valu's operations -- same bits -- but the dependent DPP / permlane chain of one value (~12 ops
// that each wait for the previous one) now overlaps with the others' (round 6: the LayerNorm prologue of a query-sized projection
// normalised its four rows per wave one after the other, ~1 200 cycles each)
template <int R>
__device__ __forceinline__ void wave_sum_valu_multi(float (&v)[R]) {
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[u]), __float_as_uint(v[u]), false, false);
        v[u] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[u]), __float_as_uint(v[u]), false, false);
        v[u] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int u = 0; u < R; ++u) v[u] += dpp_mov<0x128>(v[u]);
#pragma unroll
    for (int u = 0; u < R; ++u) v[u] += dpp_mov<0x1B>(dpp_mov<0x141>(v[u]));
#pragma unroll
    for (int u = 0; u < R; ++u) v[u] += dpp_mov<0x4E>(v[u]);
#pragma unroll
    for (int u = 0; u < R; ++u) v[u] += dpp_mov<0xB1>(v[u]);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// transformers.activations.NewGELUActivation (HF gelu_new)
__device__ __forceinline__ float gelu_new(float u) {
    const float c = 0.7978845608028654f;  // sqrt(2/pi)
    float t = c * (u + 0.044715f * u * u * u);
    return 0.5f * u * (1.0f + tanhf(t));
}

// bf16 path: 0.5u(1+tanh(t)) == u * sigmoid(2t) = u / (1 + 2^(-2t*log2e)), with the constants folded:
// 3 mul/fma + v_exp_f32 + add + v_rcp_f32 + mul (the `1/x` spelled __frcp_rn costs an 11-instruction IEEE divide).
__device__ __forceinline__ float gelu_new_fast(float u) {
    const float k0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;
    const float k1 = k0 * 0.044715f;
    const float z = u * __builtin_fmaf(u * u, k1, k0);
    return u * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z));
}

// the same with the f16 range shift folded in: returns gelu_new(u) * m for m = 1 / inv (a power of two).  (1 + e) * inv is
// one fma (inv = 1: e + 1 exactly as above, so the default path keeps its bits), and the reciprocal of a power-of-two
// multiple is the power-of-two multiple of the reciprocal.
__device__ __forceinline__ float gelu_new_fast_scaled(float u, float inv) {
    const float k0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;
    const float k1 = k0 * 0.044715f;
    const float z = u * __builtin_fmaf(u * u, k1, k0);
    return u * __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_amdgcn_exp2f(z), inv, inv));
}

// Filtered scorer epilogues: is there still room in query row's candidate list?  Called only AFTER a lane has found a
// survivor (rare).  An ATOMIC relaxed load on purpose: a plain load is hoisted by the compiler in front of the
// `max > threshold` test, which put a global load (and its wait) on every row of every tile -- measured 0.27 -> 0.37 ms on
// a 90 k-document pass.  A list already over capacity is recomputed anyway (cand_merge flags raw > cap), so its atomics are
// skipped: a chunk in which every document beats the threshold otherwise spends milliseconds in them.
__device__ __forceinline__ bool cand_room(const int* cnt, int cap) {
    return __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= cap;
}

// ---- launch descriptors shared between the .hip translation units and api.cpp ----
enum GemmEpi {
    EPI_STORE = 0,        // out[m][n] = acc                          (row-major, OutT)
    EPI_BIAS_GELU = 1,    // out[m][n] = gelu_new(acc + bias[n])      (row-major, OutT)
    EPI_BIAS_RESID = 2,   // out[m][n] = resid[m][n] + acc + bias[n]  (fp32, may alias)
    EPI_SCORE = 3,        // out[m][n] = isnan(acc) ? -1 : acc        (fp32)
    EPI_VT = 4,           // out[n][m] = acc  (transposed store, bf16: V^T for the attention B-operand)
    EPI_NONE = 5,         // micro-benchmark only: no stores (accumulators kept live)
    EPI_SCORE_FILTER = 6, // scorer, chunks after the first: append (score, index) of every score > thr[m] to a
                          // per-query candidate list instead of materialising the score tile
    EPI_SCORE_TOP2 = 8,   // scorer's threshold sample (round 5): per query row and (document tile, wave) the two best scores of the wave's
                          // 64 documents -> out[m][8 * (n0 / 256) + 2 * wn + {0, 1}]; nothing else leaves the registers (256x256 kernel only)
    EPI_QKV = 7,          // fused QKV projection: columns < n_split -> out[m][n] (q | k, row-major), the rest -> out2[n - n_split][m]
                          // (V^T); one launch for query-sized batches, split into EPI_STORE + EPI_VT launches otherwise
};

struct GemmArgs {
    const void* A;      // [M][K] row-major, leading dim lda (elements)
    const void* W;      // [N][K] row-major, leading dim ldw (elements)  -> C = A * W^T
    void* out;
    const float* bias;  // [N] or null
    const float* resid; // [M][ldo] fp32 or null
    int M, N, K;
    long lda, ldw, ldo;
    int m_valid;        // rows >= m_valid are computed from clamped reads and not stored
    int skew;           // persistent kernel: start-up stagger (shader cycles per phase), see gemm256_kernel
    int gm, gn;         // gemm256d: supertile shape in tiles (major x minor); 0 = default 4 x 8
    int balanced;       // gemm256d: 1 = few-round launch -- every XCD takes a contiguous, equally long run of the tile list (set by the launcher)
    const int* pred;    // device flag or null: the kernel exits at once when *pred == 0 (sync-free fallback launches)
    // EPI_SCORE_FILTER
    const float* thr;   // per-query threshold thr[m * thr_ld] (the running k-th best score; -inf = keep all)
    long thr_ld;
    float* cand_val;    // [M][cand_cap]
    long long* cand_idx;
    int* cand_cnt;      // [M] appended so far (may exceed cand_cap: overflow, detected by the merge)
    int cand_cap;
    long idx_base;      // global index of W row 0
    long long* dbg;     // optional s_memtime stamps of workgroup 0 / wave 0 (micro-benchmark diagnostics)
    int* range_flag;    // f16 outputs: device flag raised when a stored magnitude reaches RANGE_LIMIT (or null)
    unsigned* range_amax;   // with range_flag: slot that collects the offending magnitude (fp32 bits, atomicMax) or null
    // Power-of-two range shifts of the f16 path (exact in fp32; 0 is read as 1 by the launchers, so a zero-initialised
    // GemmArgs means "no scaling"):  out = epilogue(acc * in_mul) * out_mul
    //   in_mul  = product of the operands' up-shifts (an operand stored as v * 2^-k contributes 2^k),
    //   out_mul = 2^-k of the tensor this launch writes (EPI_QKV: out_mul for q | k, out_mul2 for V^T).
    float in_mul, out_mul, out_mul2;
    int k_algo;         // 0, or the K of the un-split problem when K carries split-precision blocks (profiling counts 2*M*N*k_algo)
    int kgroups;        // per-ctx low-latency mode: 2 = k-groups for under-filled small-tile launches (0 / 1 = off)
    int force256;       // per-ctx tile policy: 1 = keep 256x256 tiles even where the small-tile rule would apply (kernel tests)
    int cu_cap;         // per-ctx: the persistent 256x256 kernel launches at most this many workgroups (0 = one per CU)
    // Split-precision outputs (16-bit OutT; 0 = off).  Besides out (hi = round16(v)) the epilogue writes lo = round16(v - hi) at
    // element offset lo_delta from the hi element and, when hi2_delta != 0, a second copy of hi at hi2_delta: an activation
    // that enters the next GEMM as a [hi | lo | hi] row of 3 K (lo_delta = K, hi2_delta = 2 K, ldo = 3 K), or q | k / V^T
    // kept as separate hi and lo buffers for the split-precision attention (lo_delta = the buffers' distance).
    // lo_delta2: the same for out2 (EPI_QKV's V^T part).
    long lo_delta, hi2_delta, lo_delta2;
    // EPI_QKV
    void* out2;         // V^T [N - n_split][ldo2]
    long ldo2;
    int n_split;        // multiple of 128
    // EPI_SCORE_FILTER on the 256x256 kernel (round 5): documents [n_valid, N) of the last column tile do not exist -- their
    // rows are fetched from the last valid one and their scores masked, so that a ragged shard needs no second launch.  0 = N.
    int n_valid;
    // fp8 operands (gemm256q.hip): C = (A8 . W8^T) * a_scale[m] * a_scalar * w_scale[n]
    const float* a_scale;   // [M] per-row scale of the A codes, or null (1)
    const float* w_scale;   // [N] per-output-channel scale of the W codes
    float a_scalar;         // per-tensor scale of the A codes (1 when a_scale is used)
    float out_scale;        // EPI_BIAS_GELU: the fp8 output holds gelu(...) / out_scale
};

void launch_gemm(int dtype, int epi, int out_dtype, const GemmArgs& a, hipStream_t s);
// qgemm.hip: the query-sized projections (at most QGEMM_MAX_ROWS token rows, M % 32 == 0, K % 128 == 0): LDS-DMA ring with the
// whole reach in flight, optional LayerNorm prologue (x != null: A = LayerNorm(x) * ln_mul rounded to the operand format, K = d).
// Same k-ascending MFMA chain and epilogue arithmetic as the other GEMM kernels: identical bits.
constexpr int QGEMM_MAX_ROWS = 4096;
struct QGemmArgs {
    GemmArgs g;
    const float* x;        // LayerNorm prologue: fp32 [M][K] residual stream (g.A unused), or null
    const float* ln_g;
    const float* ln_b;
    float eps, ln_mul;     // ln_mul: power-of-two range shift of the normalised rows (0 is read as 1)
    int group;             // LayerNorm prologue: column tiles per workgroup (set by the launcher)
    int tile;              // 0 = the launcher's choice; k > 0 = its k-th candidate tile (scripts/micro/qgemm_probe.hip: tile sweeps)
};
bool qgemm_shape_ok(int M, int N, int K, int epi, int n_split);
bool qgemm_ln_ok(int M, int N, int d, int epi, int n_split);
// false = shape / options not served (nothing launched): the caller takes launch_gemm
bool launch_qgemm(int dtype, int epi, int out_dtype, const QGemmArgs& a, hipStream_t s);

bool gemm_qkv_one_launch(int M, int n_split, bool force256);   // 16-bit operands: does EPI_QKV apply (query-sized batch)?
bool gemm_score_tail_foldable(int M, long N, int K);   // scorer: may the < 256 trailing documents ride in the 256x256 launch (GemmArgs.n_valid)?
bool gemm_qkv_bulk(int M, int N, int K, int n_split, bool force256);   // 16-bit operands: EPI_QKV on the 256x256 kernel (bulk batch)?
#ifdef SGPT_EXPERIMENTS        // A/B knobs of the measurement scripts (libsgpt_hip_exp.so only)
int set_gemm_skew(int cycles);  // start-up stagger of the persistent 256^2 kernel (shader cycles per phase); returns the previous value
int set_gemm_use_w(int on);     // 1: the 32x32x16-MFMA re-tiling of the 256^2 kernel (gemm256w.hip); returns the previous value
#endif
// gemm256q.hip: fp8 (e4m3fn) x fp8 on v_mfma_f32_16x16x128_f8f6f4; epi = EPI_BIAS_GELU (fp8 out) | EPI_BIAS_RESID (fp32) |
// EPI_STORE / EPI_VT (16-bit out) | EPI_NONE
bool gemm_fp8_shape_ok(int M, int N, int K);
void launch_gemm_fp8(int epi, int out_dtype, const GemmArgs& a, hipStream_t s);   // out_dtype: EPI_STORE / EPI_VT only
// gemm256w.hip: the 256x256 LDS-DMA kernel on v_mfma_f32_32x32x16 (16-bit operands and outputs as gemm256d_kernel)
void launch_gemm256w(int dtype, int epi, const GemmArgs& a, hipStream_t s, bool deep_a);

struct AttnArgs {
    const void* q;       // [T][ldq]  (bf16 path: qk buffer; fp32 path: qkv buffer)
    const void* k;       // same buffer, column offset applied by the caller
    const void* v;       // bf16 path: V^T [d][ldvt]; fp32 path: [T][ldq]
    void* ctx;           // [T][d]
    const int* seq_off;  // [B+1]
    int B, H, dh;
    long ldq, ldvt, ldo;
    int window;          // 0 = global causal
    float scale;
    int max_alloc_len;
    const float* alibi;  // [H] ALiBi slopes (BLOOM) or null: score += slope_h * key_index (HF:bloom:45-89)
    int dtype;           // DT_BF16 | DT_F16 (16-bit path)
    // SGPT_FP8M: the context is written as e4m3 codes of ctx / out_scale ([T][ldo] bytes: the A operand of the fp8
    // out-projection); a saturated code raises bit 1 of *range_flag
    int out_fp8;
    float out_scale;
    int* range_flag;
    // split precision (attn.hip, MODE 1 / 2): x3 = 1: q | k, V^T and the probabilities as hi + lo pairs (lo halves qk_lo_delta /
    // v_lo_delta elements behind the hi halves); ctx_lo_delta != 0: the context leaves as hi (ctx), lo (ctx + ctx_lo_delta) and,
    // when ctx_hi2_delta != 0, a second hi (ctx + ctx_hi2_delta)
    int x3;
    long qk_lo_delta, v_lo_delta, ctx_lo_delta, ctx_hi2_delta;
};
inline bool attn_x3_supported(int dh) { return dh == 64 || dh == 128; }
void launch_attn_bf16(const AttnArgs& a, hipStream_t s);   // 16-bit MFMA path (bf16 or f16 by a.dtype)
void launch_attn_f32(const AttnArgs& a, hipStream_t s);

void launch_embed(const int* ids, const int* pos, const float* wte, const float* wpe, float* x, int T, int d, int vocab,
                  int max_pos, hipStream_t s);
void launch_layernorm(const float* x, const float* g, const float* b, void* out, int out_dtype, int T, int d,
                      float eps, hipStream_t s, float out_mul = 1.0f);   // out_mul: f16 range shift (power of two)
// split-precision Q / K projection (elementwise.hip): LayerNorm output as [hi | lo | hi] rows of 3 * d, weights as
// [W_hi | W_hi | W_lo] rows of 3 * cols
void launch_layernorm_split(const float* x, const float* g, const float* b, void* out, int out_dtype, int T, int d,
                            float eps, hipStream_t s, float out_mul);
void launch_pack_split_rows(const float* src, long rows, long cols, void* dst, int out_dtype, hipStream_t s);
// precision probe: max over rows of max|v| / rms(v) of a 16-bit operand [T][cols] (leading dim ld) folded into *out_bits
void launch_crest16(const void* in, int T, int cols, long ld, int dtype, unsigned* out_bits, hipStream_t s);
// fp32 [n][d] -> 16-bit [n][3 d]: layout 0 = [hi | lo | hi] (documents / activations), 1 = [hi | hi | lo] (queries / weights)
void launch_split16_rows(const float* in, long n, int d, int layout, void* out, int out_dtype, hipStream_t s);
// LayerNorm -> e4m3fn codes q[T,d] + one power-of-two scale per row (+ optionally the same rows in a 16-bit format)
void launch_layernorm_q8(const float* x, const float* g, const float* b, void* q, float* scale, void* out16, int out16_dtype,
                         int T, int d, float eps, hipStream_t s);
void launch_absmax16(const void* in, long numel, int dtype, unsigned* out_bits, hipStream_t s);   // max|x| of a bf16 / f16 array
void launch_lnf_pool(const float* x, const float* g, const float* b, const int* seq_off, const int* seq_len,
                     const int* pad_left, int B, int d, float eps, int apply_ln, int mode, int normalize,
                     const float* pos_weights, int pos_weights_n, float* out, hipStream_t s,
                     int* nonfinite_flag = nullptr);   // f16 models: raised (bit 0) when a pooled row is not finite
void launch_pool(const void* hidden, int dtype, const int* mask, int B, int S, int d, int mode,
                 const float* pos_weights, float* out, hipStream_t s);
// fp8 e4m3fn weight storage, one power-of-two scale per row (output channel)
void launch_fp8_quant_rows(const float* w, long rows, long cols, void* q, float* scale, hipStream_t s);
void launch_fp8_dequant_rows(const void* q, const float* scale, long rows, long cols, void* out, int out_dtype,
                             hipStream_t s);
void launch_l2norm(const float* in, long n, int d, void* out, int out_dtype, hipStream_t s);
void launch_pairwise(const float* a, const float* b, long n, int d, bool cosine, float* out, hipStream_t s);
// cross-encoder scoring (sgptce.py): gather hidden rows; log_softmax + gather target + argmax per logits row
void launch_gather_rows(const float* src, const int* row_idx, int n, int d, float* dst, hipStream_t s);
void launch_logprob_rows(const float* logits, long ld, int V, const int* targets, int n, float* out_lp, int* out_arg,
                         hipStream_t s);
// out[i] = mean_j in[j][i], in fp32 [n0][n] (layer average of the meanmean / lasttokenmean methods)
void launch_mean_over_axis0(const float* in, int n0, long n, float* out, hipStream_t s);
void launch_f32_to_bf16(const float* in, long numel, void* out, hipStream_t s);
void launch_f32_to_16(const float* in, long numel, void* out, int out_dtype, hipStream_t s);   // DT_BF16 | DT_F16, RNE
// max |in[i]| folded into *out_bits (the fp32 bit pattern of a non-negative float, atomicMax; zero it first)
void launch_absmax(const float* in, long numel, unsigned* out_bits, hipStream_t s);
void launch_fill_f32(float* p, long n, float v, hipStream_t s);
// GPT-J rotary embedding, in place on the q / k columns of the projection buffer
void launch_rope(void* qk, int dtype, long ld, long k_off, const int* pos, const float* sin_t, const float* cos_t, int T,
                 int H, int dh, int rotary_dim, hipStream_t s);
// BLOOM fused QKV rows [n_head, 3, head_dim] -> [q rows | k rows | v rows] (row_len floats per row; 1 for the bias)
void launch_qkv_deinterleave(const float* src, float* dst, int H, int dh, long row_len, hipStream_t s);
void launch_fill_rand(void* p, long n, int dtype, unsigned seed, float scale, hipStream_t s);

// top-k: one block per query row over a virtual row = [scores(n) | prev(n_prev)]
void launch_topk_select(const float* scores, long ld, long n, long idx_base, const float* prev_val,
                        const int64_t* prev_idx, int n_prev, long prev_ld, int nq, int k, int nan_to_m1,
                        const int64_t* exclude_idx, float* out_val, int64_t* out_idx, hipStream_t s,
                        const int* pred = nullptr, float* thr_out = nullptr,    // thr_out[q] = the k-th best, one ulp lower (or null)
                        int gather_k = 0);   // > 0: prev_val / prev_idx are a [world][nq][gather_k] stack of per-rank lists (n_prev = world * gather_k)
// refined scorer (sgpt_score_topk_refined): exact fp32 scores of stage-1 candidates, the guarantee check, list concatenation
void launch_rescore(const float* q, const float* corpus, const int64_t* idx, long ld_idx, int nq, int m, int d, long idx_base,
                    long n_docs, float* out_val, int64_t* out_idx, long ld_out, hipStream_t s);
void launch_refine_check(const float* v16, int k, int kp, float margin, int nq, int* flag, hipStream_t s);
void launch_list_append(const float* in_val, const int64_t* in_idx, long ld_in, int n_run, int nq, float* out_val, int64_t* out_idx,
                        long ld_out, int col0, hipStream_t s);
// scorer pass prologue in one launch: q -> zero-padded qpad (byte counts, multiples of 16), counters[n] = 0, idx_list[n] = -1
void launch_score_prep(const void* q, void* qpad, long q_bytes, long qpad_bytes, int* counters, long n_counters, long long* idx_list,
                       long n_idx, hipStream_t s);
// thr[q] = nextafter(list[q][k-1], -inf): the inclusive form of a sampled threshold for the strict `>` filter (topk.hip)
void launch_thr_below(const float* list, int k, int nq, float* thr, hipStream_t s);
// k best of {running top-k} U {candidate list of the filtered score GEMM}; resets cnt[q], raises *overflow when a
// list was longer than cap (candidates were dropped: the caller's predicated fallback recomputes)
void launch_cand_merge(const float* run_val, const int64_t* run_idx, const float* cand_val, const int64_t* cand_idx,
                       int* cand_cnt, int cap, int nq, int k, float* out_val, int64_t* out_idx, int* overflow,
                       hipStream_t s, float* thr_io = nullptr);   // thr_io: dense per-query threshold, raised to the new k-th best
