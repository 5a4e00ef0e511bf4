// HBM-bound streaming kernels of the encoder: embedding gather-add, LayerNorm, the fused
// final-LayerNorm + position-weighted mean pool, stand-alone pooling, L2 normalise.
// All are one-wave-per-row (64 lanes x float4 = 1 KiB per instruction, coalesced) with
// wavefront-shuffle reductions; no LDS except the cross-wave combine of the pool.
#include "common.h"
#include "rowln.h"


namespace {

// ---- wte[ids] + wpe[pos]  (HF:gpt_neo/modeling_gpt_neo.py:444,462-463) ----
__global__ __launch_bounds__(256) void embed_kernel(const int* __restrict__ ids, const int* __restrict__ pos,
                                                    const float* __restrict__ wte, const float* __restrict__ wpe,
                                                    float* __restrict__ x, int T, int d, int vocab, int max_pos) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const int lane = threadIdx.x & 63;
    // ids / positions are clamped into the tables: a C-ABI caller's bad id reads a wrong row, never out of bounds
    // (the Python host validates and raises before the call, model.py::SGPTModel.pack)
    int id = ids[row]; id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const float4* a = reinterpret_cast<const float4*>(wte + (long)id * d);
    float4* o = reinterpret_cast<float4*>(x + (long)row * d);
    if (wpe == nullptr) {  // GPT-J: no learned position embedding (HF:gptj:484)
        for (int c = lane; c < d / 4; c += 64) o[c] = a[c];
        return;
    }
    int ps = pos[row]; ps = ps < 0 ? 0 : (ps >= max_pos ? max_pos - 1 : ps);
    const float4* b = reinterpret_cast<const float4*>(wpe + (long)ps * d);
    for (int c = lane; c < d / 4; c += 64) {
        const float4 u = a[c], v = b[c];
        o[c] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    }
}

// out_mul: power-of-two range shift of an f16 output (the consuming GEMMs multiply their accumulators by 1 / out_mul);
// 1 for every other format and for models without shifts (v * 1 == v: the default path keeps its bits)
template <typename OutT, int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                        const float* __restrict__ b, OutT* __restrict__ out, int T,
                                                        int d, float eps, float out_mul) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const int lane = threadIdx.x & 63;
    RowLN<NV> r;
    r.load(x + (long)row * d, d, lane);
    r.normalize(g, b, d, eps, lane);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < d) {
            if constexpr (sizeof(OutT) == 4) {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (long)row * d + c) = r.v[i];
            } else {
                *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(out) + (long)row * d + c) =
                    make_uint2(Half<OutT>::pack2(r.v[i].x * out_mul, r.v[i].y * out_mul),
                               Half<OutT>::pack2(r.v[i].z * out_mul, r.v[i].w * out_mul));
            }
        }
    }
}

// LayerNorm for the split-precision Q / K projection (sgpt_model_desc.qk_split): the normalised row a is written as
// [hi | lo | hi] with hi = round16(a), lo = round16(a - hi) -- three K blocks of a [T, 3d] operand.  Against weights packed as
// [W_hi | W_hi | W_lo] (pack_split_rows_kernel) one ordinary GEMM over K' = 3d computes a_hi.W_hi + a_lo.W_hi + a_hi.W_lo:
// the product of the two operands to ~2^-22 instead of 2^-11 each, on the 16-bit MFMA, with no kernel of its own.  The V
// projection (and GPT-J's MLP) read the first block alone (lda = 3d, K = d): the plain 16-bit LayerNorm output.
template <typename OutT, int NV>
__global__ __launch_bounds__(256) void layernorm_split_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                              const float* __restrict__ b, uint16_t* __restrict__ out, int T,
                                                              int d, float eps, float out_mul) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const int lane = threadIdx.x & 63;
    RowLN<NV> r;
    r.load(x + (long)row * d, d, lane);
    r.normalize(g, b, d, eps, lane);
    uint16_t* o = out + (long)row * 3 * d;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < d) {
            const float v0 = r.v[i].x * out_mul, v1 = r.v[i].y * out_mul, v2 = r.v[i].z * out_mul, v3 = r.v[i].w * out_mul;
            const uint32_t h01 = Half<OutT>::pack2(v0, v1), h23 = Half<OutT>::pack2(v2, v3);
            const uint32_t l01 = Half<OutT>::pack2(v0 - Half<OutT>::lo(h01), v1 - Half<OutT>::hi(h01));
            const uint32_t l23 = Half<OutT>::pack2(v2 - Half<OutT>::lo(h23), v3 - Half<OutT>::hi(h23));
            *reinterpret_cast<uint2*>(o + c) = make_uint2(h01, h23);
            *reinterpret_cast<uint2*>(o + d + c) = make_uint2(l01, l23);
            *reinterpret_cast<uint2*>(o + 2 * d + c) = make_uint2(h01, h23);
        }
    }
}

// weights of the split-precision projection: src fp32 [rows, cols] -> dst 16-bit [rows, 3 * cols] = [W_hi | W_hi | W_lo]
template <typename H>
__global__ __launch_bounds__(256) void pack_split_rows_kernel(const float* __restrict__ src, long rows, long cols,
                                                              uint16_t* __restrict__ dst) {
    const long n = rows * cols, stride = (long)gridDim.x * 256;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += stride) {
        const long r = e / cols, c = e - r * cols;
        const float w = src[e];
        const uint32_t hi = Half<H>::pack2(w, 0.0f);
        const uint16_t lo = f32_to_h<H>(w - Half<H>::lo(hi));
        uint16_t* o = dst + r * 3 * cols + c;
        o[0] = (uint16_t)(hi & 0xffffu); o[cols] = (uint16_t)(hi & 0xffffu); o[2 * cols] = lo;
    }
}

// LayerNorm whose output feeds an fp8-MFMA GEMM (gemm256q.hip): the normalised row is quantised to OCP e4m3fn codes under
// ONE power-of-two scale per row -- the smallest 2^k with max|row| / 2^k <= 448 (the rule of fp8_quant_rows_kernel) --
// which factors out of the GEMM's k-sum and is applied to its accumulators.  Optionally the same row is also written in
// a 16-bit format (GPT-J: ln_1 feeds the attention projections and the MLP).
template <int NV, typename H16>
__global__ __launch_bounds__(256) void layernorm_q8_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                           const float* __restrict__ b, uint8_t* __restrict__ q,
                                                           float* __restrict__ scale, uint16_t* __restrict__ out16, int T,
                                                           int d, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const int lane = threadIdx.x & 63;
    RowLN<NV> r;
    r.load(x + (long)row * d, d, lane);
    r.normalize(g, b, d, eps, lane);
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < d) amax = fmaxf(amax, fmaxf(fmaxf(fabsf(r.v[i].x), fabsf(r.v[i].y)), fmaxf(fabsf(r.v[i].z), fabsf(r.v[i].w))));
    }
    amax = wave_max(amax);
    float sc = 1.0f;
    if (amax > 0.f && amax < 3.0e38f) {
        const uint32_t u = __float_as_uint(amax);
        const int e = (int)((u >> 23) & 0xffu) - 127;
        int k = e - ((u & 0x7fffffu) > 0x600000u ? 7 : 8);               // amax / 2^k in (224, 448]
        k = k < -126 ? -126 : (k > 127 ? 127 : k);
        sc = __uint_as_float((uint32_t)(k + 127) << 23);
    }
    if (lane == 0) scale[row] = sc;
    const float inv = 1.0f / sc;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < d) {
            int w = __builtin_amdgcn_cvt_pk_fp8_f32(r.v[i].x * inv, r.v[i].y * inv, 0, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(r.v[i].z * inv, r.v[i].w * inv, w, true);
            *reinterpret_cast<uint32_t*>(q + (long)row * d + c) = (uint32_t)w;
            if (out16 != nullptr)
                *reinterpret_cast<uint2*>(out16 + (long)row * d + c) =
                    make_uint2(Half<H16>::pack2(r.v[i].x, r.v[i].y), Half<H16>::pack2(r.v[i].z, r.v[i].w));
        }
    }
}

// ---- ln_f + pooling + optional L2 normalise, one workgroup (4 waves) per sequence ----
// weightedmean: sum_t (P+t+1) * h_t / clamp(sum_t (P+t+1), 1e-9), P = pad_left
//   (Pooling.py:99-125; beir_dense_retriever.py:258-270; weights follow the PADDED index)
// mean: Pooling.py:117-125 / beir_dense_retriever.py:238-242;  lasttoken: :271-282 (index len-1)
// learntmean (mode 3): w_t = position_weights[P+t], clamp 1e-9 (WeightedMeanPooling.py:21-39)
template <int NV>
__global__ __launch_bounds__(256) void lnf_pool_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                       const float* __restrict__ b, const int* __restrict__ seq_off,
                                                       const int* __restrict__ seq_len,
                                                       const int* __restrict__ pad_left, int d, float eps,
                                                       int apply_ln, int mode, int normalize,
                                                       const float* __restrict__ pw, int pw_n, float* __restrict__ out,
                                                       int* __restrict__ nonfinite_flag) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // [4][d] + 8
    const int sq = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s0 = seq_off[sq], len = seq_len[sq], P = pad_left ? pad_left[sq] : 0;

    float4 acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    float den = 0.f;
    const int t_lo = mode == 2 ? (len > 0 ? len - 1 : 0) : 0;
    // A wave's rows t = wave, wave + 4, ... are accumulated in that order; their LOADS go out LPB rows at a time (round 6: one
    // sequence of 30 tokens is eight dependent load -> normalise -> accumulate round trips per wave otherwise -- 14 us of a
    // 0.43 ms single-query encode; a bulk call has thousands of workgroups to hide the latency behind).  Same sums, same bits.
    constexpr int LPB = NV <= 4 ? 4 : 2;
    for (int t = t_lo + wave; t < len; t += 4 * LPB) {
        RowLN<NV> r[LPB];
#pragma unroll
        for (int u = 0; u < LPB; ++u)
            if (t + 4 * u < len) r[u].load(x + (long)(s0 + t + 4 * u) * d, d, lane);
#pragma unroll
        for (int u = 0; u < LPB; ++u) {
            const int tt = t + 4 * u;
            if (tt >= len) break;
            if (apply_ln) r[u].template normalize<true>(g, b, d, eps, lane);
            // mode 3 (learntmean): trained per-position weights, indexed like the padded position
            // (WeightedMeanPooling.py:21-39; useb_dense_retriever.py:253-270)
            const float w = mode == 0 ? (float)(P + tt + 1) : (mode == 3 ? pw[P + tt < pw_n ? P + tt : pw_n - 1] : 1.0f);   // index clamped: no OOB read
            den += w;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                acc[i].x += w * r[u].v[i].x; acc[i].y += w * r[u].v[i].y;
                acc[i].z += w * r[u].v[i].z; acc[i].w += w * r[u].v[i].w;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < d) *reinterpret_cast<float4*>(sm + wave * d + c) = acc[i];
    }
    float* red = sm + 4 * d;
    if (lane == 0) red[wave] = den;
    __syncthreads();
    float dsum = (red[0] + red[1]) + (red[2] + red[3]);
    if (mode != 2) dsum = fmaxf(dsum, 1e-9f);  // Pooling.py:122
    else dsum = 1.0f;
    float ss = 0.f;
    for (int c = threadIdx.x; c < d; c += 256) {
        const float e = ((sm[c] + sm[d + c]) + (sm[2 * d + c] + sm[3 * d + c])) / dsum;
        sm[c] = e;
        ss += e * e;
    }
    // f16 models: a NaN / inf that reached the pooled embedding (an under-shifted LayerNorm output overflowing to inf gives
    // inf - inf in the next GEMM, and the per-launch range trackers' fmaxf drops NaN) raises the model's guard word: never silent
    if (nonfinite_flag != nullptr && !(ss < INFINITY)) atomicOr(nonfinite_flag, 1);
    if (normalize) {  // F.normalize(p=2, dim=1), SentenceTransformer.py:248-249
        ss = wave_sum(ss);
        __syncthreads();
        if (lane == 0) red[4 + wave] = ss;
        __syncthreads();
        const float nrm = fmaxf(sqrtf((red[4] + red[5]) + (red[6] + red[7])), 1e-12f);
        for (int c = threadIdx.x; c < d; c += 256) out[(long)sq * d + c] = sm[c] / nrm;
    } else {
        for (int c = threadIdx.x; c < d; c += 256) out[(long)sq * d + c] = sm[c];
    }
}

// ---- stand-alone pooling over [B,S,d] hidden states + {0,1} mask (any padding side) ----
template <typename T>
__device__ __forceinline__ float4 load4(const T* p);
template <> __device__ __forceinline__ float4 load4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 load4<bf16_t>(const bf16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}

template <> __device__ __forceinline__ float4 load4<f16_t>(const f16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(Half<f16_t>::lo(u.x), Half<f16_t>::hi(u.x), Half<f16_t>::lo(u.y), Half<f16_t>::hi(u.y));
}

template <typename T>
__global__ __launch_bounds__(256) void pool_kernel(const T* __restrict__ h, const int* __restrict__ mask, int S, int d,
                                                   int mode, const float* __restrict__ pw, float* __restrict__ out) {
    const int bq = blockIdx.x;
    const int c = (blockIdx.y * 256 + threadIdx.x) * 4;
    if (c >= d) return;
    const T* hb = h + (long)bq * S * d + c;
    const int* mb = mask + (long)bq * S;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (mode == 2) {
        int last = 0;
        for (int t = 0; t < S; ++t) if (mb[t] != 0) last = t;
        acc = load4<T>(hb + (long)last * d);
    } else {
        float den = 0.f;
#pragma unroll 4
        for (int t = 0; t < S; ++t) {
            const float w = mb[t] != 0 ? (mode == 0 ? (float)(t + 1) : (mode == 3 ? pw[t] : 1.0f)) : 0.0f;
            const float4 v = load4<T>(hb + (long)t * d);
            den += w;
            acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
        }
        den = fmaxf(den, 1e-9f);
        acc.x /= den; acc.y /= den; acc.z /= den; acc.w /= den;
    }
    *reinterpret_cast<float4*>(out + (long)bq * d + c) = acc;
}

// ---- x / max(||x||, 1e-12) per row; fp32 or bf16 output ----
template <typename OutT>
__global__ __launch_bounds__(256) void l2norm_kernel(const float* __restrict__ in, long n, int d,
                                                     OutT* __restrict__ out) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int lane = threadIdx.x & 63;
    const float* xr = in + row * d;
    float s = 0.f;
    for (int c = lane; c < d; c += 64) { const float v = xr[c]; s += v * v; }
    const float nrm = fmaxf(sqrtf(wave_sum(s)), 1e-12f);
    for (int c = lane; c < d; c += 64) {
        const float v = xr[c] / nrm;
        if constexpr (sizeof(OutT) == 4) reinterpret_cast<float*>(out)[row * d + c] = v;
        else reinterpret_cast<uint16_t*>(out)[row * d + c] = f32_to_h<OutT>(v);
    }
}

// ---- row-wise scores of two equally shaped matrices: util.pairwise_dot_score / pairwise_cos_sim (util.py:66-91) ----
// out[i] = sum_j a[i][j] b[i][j]            (COS = false: `(a * b).sum(dim=-1)`)
//        = sum_j (a[i][j] / max(|a[i]|, 1e-12)) (b[i][j] / max(|b[i]|, 1e-12))   (COS = true: both sides through
//          normalize_embeddings first, then the same product sum -- the reference's order of operations, not dot / (|a| |b|)).
// One wave per row pair, float4 loads where the row allows, wavefront butterflies for the three sums.
template <bool COS>
__global__ __launch_bounds__(256) void pairwise_kernel(const float* __restrict__ a, const float* __restrict__ b, long n, int d,
                                                       float* __restrict__ out) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int lane = threadIdx.x & 63;
    const float* ar = a + row * d;
    const float* br = b + row * d;
    float na = 1.f, nb = 1.f;
    if constexpr (COS) {
        float sa = 0.f, sb = 0.f;
        for (int c = lane; c < d; c += 64) { const float x = ar[c], y = br[c]; sa += x * x; sb += y * y; }
        na = fmaxf(sqrtf(wave_sum(sa)), 1e-12f);
        nb = fmaxf(sqrtf(wave_sum(sb)), 1e-12f);
    }
    float s = 0.f;
    for (int c = lane; c < d; c += 64) {
        const float x = COS ? ar[c] / na : ar[c], y = COS ? br[c] / nb : br[c];
        s += x * y;
    }
    s = wave_sum(s);
    if (lane == 0) out[row] = s;
}

// ---- GPT-J rotary position embedding (HF:gptj/modeling_gptj.py:57-67,190-210) ----
// For every token, head and pair i < rotary_dim/2 of the head's leading dims:
//   (x[2i], x[2i+1]) <- (x[2i] cos - x[2i+1] sin, x[2i+1] cos + x[2i] sin),  angle = pos * inv_freq[i]
// applied in place to q (column 0) and k (column k_off) of the projection buffer.  One thread per pair.
template <typename T>
__global__ __launch_bounds__(256) void rope_kernel(T* __restrict__ buf, long ld, long k_off, const int* __restrict__ pos,
                                                   const float* __restrict__ sin_t, const float* __restrict__ cos_t,
                                                   int Tn, int H, int dh, int half) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long per_tok = (long)H * half;
    if (idx >= (long)Tn * per_tok) return;
    const int tkn = (int)(idx / per_tok);
    const int rem = (int)(idx - (long)tkn * per_tok);
    const int h = rem / half, i = rem - h * half;
    const float sn = sin_t[(long)pos[tkn] * half + i], cs = cos_t[(long)pos[tkn] * half + i];
    T* q = buf + (long)tkn * ld + (long)h * dh + 2 * i;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        T* ptr = q + which * k_off;
        if constexpr (sizeof(T) == 4) {
            const float2 v = *reinterpret_cast<const float2*>(ptr);
            *reinterpret_cast<float2*>(ptr) = make_float2(v.x * cs - v.y * sn, v.y * cs + v.x * sn);
        } else {
            const uint32_t u = *reinterpret_cast<const uint32_t*>(ptr);
            const float x0 = Half<T>::lo(u), x1 = Half<T>::hi(u);   // f16: |rotated| <= sqrt(2) * RANGE_LIMIT < 65504
            *reinterpret_cast<uint32_t*>(ptr) = Half<T>::pack2(x0 * cs - x1 * sn, x1 * cs + x0 * sn);
        }
    }
}

// BLOOM stores the QKV projection fused and head-interleaved: row h*3*dh + which*dh + c (HF:bloom:214).
// Re-order to row which*d + h*dh + c so the Q/K and V projections are the same contiguous blocks as for GPT-Neo.
__global__ __launch_bounds__(256) void qkv_deinterleave_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                               int H, int dh, long row_len) {
    const int drow = blockIdx.x;                       // destination row in [0, 3*H*dh)
    const int d = H * dh;
    const int which = drow / d, rem = drow - which * d, h = rem / dh, c = rem - h * dh;
    const long srow = (long)h * 3 * dh + (long)which * dh + c;
    for (long i = threadIdx.x; i < row_len; i += 256) dst[(long)drow * row_len + i] = src[srow * row_len + i];
}

// ---- fp8 (OCP e4m3fn) weight storage with one power-of-two fp32 scale per output channel (row) ----
// scale[r] = smallest 2^k with amax_r / 2^k <= 448; code = RNE(w / scale) in e4m3fn.  Dividing by a power of
// two is exact, and code * scale is exactly representable in bf16 (4 significant bits), so the bf16 MFMA path
// runs on precisely the de-quantised weights the oracle uses in fp32.  Software encode/decode: load-time and
// once-per-layer work, and independent of any hardware conversion / saturation mode.
__device__ __forceinline__ uint32_t e4m3fn_encode(float x) {
    const uint32_t sign = (__float_as_uint(x) >> 24) & 0x80u;
    const float a = fabsf(x);
    if (!(a < 464.f)) return sign | (a != a ? 0x7fu : 0x7eu);        // saturate to 448; NaN stays NaN
    if (a < 0.015625f) return sign | (uint32_t)rintf(a * 512.f);      // subnormals: multiples of 2^-9 (8 -> min normal)
    uint32_t u = __float_as_uint(a);
    u += 0x7ffffu + ((u >> 20) & 1u);                                 // RNE to 3 mantissa bits
    u >>= 20;
    const uint32_t code = u - (120u << 3);                            // re-bias 127 -> 7
    return sign | (code > 0x7eu ? 0x7eu : code);
}
__device__ __forceinline__ float e4m3fn_decode(uint32_t c) {
    const uint32_t e = (c >> 3) & 15u, m = c & 7u;
    const float a = e ? __uint_as_float(((e + 120u) << 23) | (m << 20)) : (float)m * 0.001953125f;
    return (c & 0x80u) ? -a : a;
}

// one wave per row: amax -> scale, then encode
__global__ __launch_bounds__(256) void fp8_quant_rows_kernel(const float* __restrict__ w, long rows, long cols,
                                                             uint8_t* __restrict__ q, float* __restrict__ scale) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* wr = w + row * cols;
    float amax = 0.f;
    for (long c = lane; c < cols; c += 64) amax = fmaxf(amax, fabsf(wr[c]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
    float sc = 1.0f;
    if (amax > 0.f && amax < 3.0e38f) {
        const uint32_t u = __float_as_uint(amax);
        const int e = (int)((u >> 23) & 0xffu) - 127;                 // amax = m * 2^e, 1 <= m < 2 (subnormal amax: e = -127)
        const uint32_t frac = u & 0x7fffffu;
        int k = e - (frac > 0x600000u ? 7 : 8);                       // m <= 1.75 -> 2^(e-8), else 2^(e-7)
        k = k < -126 ? -126 : (k > 127 ? 127 : k);
        sc = __uint_as_float((uint32_t)(k + 127) << 23);
    }
    if (lane == 0) scale[row] = sc;
    const float inv = 1.0f / sc;                                      // exact: power of two
    for (long c = lane; c < cols; c += 64) q[row * cols + c] = (uint8_t)e4m3fn_encode(wr[c] * inv);
}

template <typename OutT>
__global__ __launch_bounds__(256) void fp8_dequant_rows_kernel(const uint8_t* __restrict__ q, const float* __restrict__ scale,
                                                               long rows, long cols, OutT* __restrict__ out) {
    // cols % 4 == 0: each thread decodes 4 codes of one row
    const long n4 = rows * cols / 4, stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const uint32_t u = reinterpret_cast<const uint32_t*>(q)[i];
        const float sc = scale[i * 4 / cols];
        const float v0 = e4m3fn_decode(u & 0xffu) * sc, v1 = e4m3fn_decode((u >> 8) & 0xffu) * sc;
        const float v2 = e4m3fn_decode((u >> 16) & 0xffu) * sc, v3 = e4m3fn_decode(u >> 24) * sc;
        if constexpr (sizeof(OutT) == 4) reinterpret_cast<float4*>(out)[i] = make_float4(v0, v1, v2, v3);
        else reinterpret_cast<uint2*>(out)[i] = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
    }
}

template <typename H>
__global__ __launch_bounds__(256) void cvt16_kernel(const float* __restrict__ in, long numel, uint16_t* __restrict__ out) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < numel; i += stride) out[i] = f32_to_h<H>(in[i]);
}

// max |x| of an fp32 array folded into *out_bits by atomicMax on the bit pattern (non-negative floats order like
// unsigned integers; NaN maps above +inf and is therefore caught by a `< limit` test on the result)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ in, long numel, unsigned* __restrict__ out_bits) {
    const long stride = (long)gridDim.x * 256;
    unsigned m = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < numel; i += stride) {
        const unsigned b = __float_as_uint(in[i]) & 0x7fffffffu;
        m = b > m ? b : m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out_bits, m);
}

__global__ __launch_bounds__(256) void mean_axis0_kernel(const float* __restrict__ in, int n0, long n,
                                                         float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float acc = 0.f;
    for (int j = 0; j < n0; ++j) acc += in[(long)j * n + i];
    out[i] = acc / (float)n0;
}

// ---- cross-encoder scoring: rows of the last hidden state -> log P(target | prefix) (crossencoder/beir/sgptce.py:150-262) ----
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ row_idx, int n,
                                                          int d, float* __restrict__ dst) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const int lane = threadIdx.x & 63;
    const float4* s4 = reinterpret_cast<const float4*>(src + (long)row_idx[r] * d);
    float4* d4 = reinterpret_cast<float4*>(dst + (long)r * d);
    for (int c = lane; c < d / 4; c += 64) d4[c] = s4[c];
}

// one workgroup per row of logits [n, ld]: log_softmax over the V entries, gathered at the target id
// (F.log_softmax + torch.gather, sgptce.py:233,255), and the greedy token (argmax, first maximum, :243)
__global__ __launch_bounds__(256) void logprob_rows_kernel(const float* __restrict__ logits, long ld, int V,
                                                           const int* __restrict__ targets, float* __restrict__ out_lp,
                                                           int* __restrict__ out_arg) {
    __shared__ float s_f[4];
    __shared__ int s_i[4];
    const int r = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float* x = logits + (long)r * ld;
    float mx = -INFINITY;
    int am = 0x7fffffff;
    for (int i = t; i < V; i += 256) {
        const float v = x[i];
        if (v > mx || (v == mx && i < am)) { mx = v; am = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(mx, o, 64);
        const int oi = __shfl_xor(am, o, 64);
        if (ov > mx || (ov == mx && oi < am)) { mx = ov; am = oi; }
    }
    if (lane == 0) { s_f[wave] = mx; s_i[wave] = am; }
    __syncthreads();
    mx = s_f[0]; am = s_i[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (s_f[w] > mx || (s_f[w] == mx && s_i[w] < am)) { mx = s_f[w]; am = s_i[w]; }
    __syncthreads();
    float sum = 0.f;
    for (int i = t; i < V; i += 256) sum += expf(x[i] - mx);
    sum = wave_sum(sum);
    if (lane == 0) s_f[wave] = sum;
    __syncthreads();
    if (t == 0) {
        const float tot = (s_f[0] + s_f[1]) + (s_f[2] + s_f[3]);
        out_lp[r] = (x[targets[r]] - mx) - logf(tot);
        if (out_arg) out_arg[r] = am;
    }
}

__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ p, long n, float v) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = v;
}

// deterministic pseudo-random fill in [-scale, scale) (micro-benchmarks: never time MFMA on zeros, DVFS)
template <typename T>
__global__ __launch_bounds__(256) void fill_rand_kernel(T* __restrict__ p, long n, unsigned seed, float scale) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
        const float v = ((float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;
        if constexpr (sizeof(T) == 4) p[i] = v; else reinterpret_cast<uint16_t*>(p)[i] = f32_to_h<T>(v);
    }
}

// max |x| of a 16-bit array (fp8 activation-scale calibration: the range of a block's GELU output)
template <typename H>
__global__ __launch_bounds__(256) void absmax16_kernel(const uint32_t* __restrict__ in, long n2, unsigned* __restrict__ out_bits) {
    const long stride = (long)gridDim.x * 256;
    float m = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride) {
        const uint32_t u = in[i];
        m = fmaxf(m, fmaxf(fabsf(Half<H>::lo(u)), fabsf(Half<H>::hi(u))));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out_bits, __float_as_uint(m));
}

// Precision probe (sgpt_model_precision_probe_begin / _end): the crest factor max|v| / rms(v) of every row of a 16-bit
// operand [T][cols] (leading dimension ld), its maximum over the rows folded into *out_bits (fp32 bits, atomicMax).  A row of
// 11-bit values whose energy sits in one or two entries (outlier channels / hidden units of real GPT-Neo checkpoints: crest
// ~ sqrt(cols / 2); a well-conditioned row: 4-9) puts the whole relative rounding error of those entries on the dot product.
// One wave per row.
template <typename H>
__global__ __launch_bounds__(256) void crest16_kernel(const uint16_t* __restrict__ in, int T, int cols, long ld,
                                                      unsigned* __restrict__ out_bits) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const int lane = threadIdx.x & 63;
    const uint32_t* r = reinterpret_cast<const uint32_t*>(in + (long)row * ld);
    float mx = 0.f, ss = 0.f;
    for (int c = lane; c < cols / 2; c += 64) {
        const uint32_t u = r[c];
        const float a = Half<H>::lo(u), b = Half<H>::hi(u);
        mx = fmaxf(mx, fmaxf(fabsf(a), fabsf(b)));
        ss += a * a + b * b;
    }
    mx = wave_max(mx);
    ss = wave_sum(ss);
    if (lane == 0 && ss > 0.f && mx < INFINITY) {
        const float crest = mx / sqrtf(ss / (float)cols);
        atomicMax(out_bits, __float_as_uint(crest));
    }
}

// fp32 rows [n][d] -> split-precision 16-bit rows [n][3 d]: layout 0 = [hi | lo | hi] (the streamed operand: activations,
// documents), layout 1 = [hi | hi | lo] (the resident operand: weights, queries); hi = round16(v), lo = round16(v - hi).
// One contraction over 3 d of a layout-0 row with a layout-1 row is hi.hi + lo.hi + hi.lo: the product to ~2^-22.
template <typename H>
__global__ __launch_bounds__(256) void split16_rows_kernel(const float* __restrict__ in, long n, int d, int layout,
                                                           uint16_t* __restrict__ out) {
    const long total = n * (long)(d / 4), stride = (long)gridDim.x * 256;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
        const long r = e / (d / 4);
        const int c = (int)(e - r * (d / 4)) * 4;
        const float4 v = *reinterpret_cast<const float4*>(in + r * d + c);
        const uint32_t h01 = Half<H>::pack2(v.x, v.y), h23 = Half<H>::pack2(v.z, v.w);
        const uint2 hi = make_uint2(h01, h23);
        const uint2 lo = make_uint2(Half<H>::pack2(v.x - Half<H>::lo(h01), v.y - Half<H>::hi(h01)),
                                    Half<H>::pack2(v.z - Half<H>::lo(h23), v.w - Half<H>::hi(h23)));
        uint16_t* o = out + r * 3 * d + c;
        *reinterpret_cast<uint2*>(o) = hi;
        *reinterpret_cast<uint2*>(o + d) = layout == 0 ? lo : hi;
        *reinterpret_cast<uint2*>(o + 2 * d) = layout == 0 ? hi : lo;
    }
}

inline int cap_grid(long blocks) { return (int)(blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks)); }

}  // namespace

void launch_embed(const int* ids, const int* pos, const float* wte, const float* wpe, float* x, int T, int d, int vocab,
                  int max_pos, hipStream_t s) {
    hipLaunchKernelGGL(embed_kernel, dim3((T + 3) / 4), dim3(256), 0, s, ids, pos, wte, wpe, x, T, d, vocab, max_pos);
}

void launch_layernorm(const float* x, const float* g, const float* b, void* out, int out_dtype, int T, int d,
                      float eps, hipStream_t s, float out_mul) {
#define LN_CASE(NV)                                                                                              \
    if (out_dtype == DT_BF16)                                                                                    \
        hipLaunchKernelGGL((layernorm_kernel<bf16_t, NV>), dim3((T + 3) / 4), dim3(256), 0, s, x, g, b,          \
                           (bf16_t*)out, T, d, eps, 1.0f);                                                       \
    else if (out_dtype == DT_F16)                                                                                \
        hipLaunchKernelGGL((layernorm_kernel<f16_t, NV>), dim3((T + 3) / 4), dim3(256), 0, s, x, g, b,           \
                           (f16_t*)out, T, d, eps, out_mul);                                                     \
    else                                                                                                         \
        hipLaunchKernelGGL((layernorm_kernel<float, NV>), dim3((T + 3) / 4), dim3(256), 0, s, x, g, b,           \
                           (float*)out, T, d, eps, 1.0f);
    const int nv = (d + 255) / 256;
    if (nv <= 1) { LN_CASE(1) } else if (nv <= 2) { LN_CASE(2) } else if (nv <= 3) { LN_CASE(3) }
    else if (nv <= 4) { LN_CASE(4) } else if (nv <= 8) { LN_CASE(8) } else if (nv <= 10) { LN_CASE(10) }
    else { LN_CASE(16) }
#undef LN_CASE
}

void launch_layernorm_split(const float* x, const float* g, const float* b, void* out, int out_dtype, int T, int d,
                            float eps, hipStream_t s, float out_mul) {
#define LS_CASE(NV)                                                                                              \
    if (out_dtype == DT_F16)                                                                                     \
        hipLaunchKernelGGL((layernorm_split_kernel<f16_t, NV>), dim3((T + 3) / 4), dim3(256), 0, s, x, g, b,     \
                           (uint16_t*)out, T, d, eps, out_mul);                                                  \
    else                                                                                                         \
        hipLaunchKernelGGL((layernorm_split_kernel<bf16_t, NV>), dim3((T + 3) / 4), dim3(256), 0, s, x, g, b,    \
                           (uint16_t*)out, T, d, eps, 1.0f);
    const int nv = (d + 255) / 256;
    if (nv <= 1) { LS_CASE(1) } else if (nv <= 2) { LS_CASE(2) } else if (nv <= 3) { LS_CASE(3) }
    else if (nv <= 4) { LS_CASE(4) } else if (nv <= 8) { LS_CASE(8) } else if (nv <= 10) { LS_CASE(10) }
    else { LS_CASE(16) }
#undef LS_CASE
}

void launch_crest16(const void* in, int T, int cols, long ld, int dtype, unsigned* out_bits, hipStream_t s) {
    if (dtype == DT_F16) hipLaunchKernelGGL(crest16_kernel<f16_t>, dim3((T + 3) / 4), dim3(256), 0, s, (const uint16_t*)in, T, cols, ld, out_bits);
    else hipLaunchKernelGGL(crest16_kernel<bf16_t>, dim3((T + 3) / 4), dim3(256), 0, s, (const uint16_t*)in, T, cols, ld, out_bits);
}

void launch_split16_rows(const float* in, long n, int d, int layout, void* out, int out_dtype, hipStream_t s) {
    const int grid = cap_grid((n * (d / 4) + 255) / 256);
    if (out_dtype == DT_F16) hipLaunchKernelGGL(split16_rows_kernel<f16_t>, dim3(grid), dim3(256), 0, s, in, n, d, layout, (uint16_t*)out);
    else hipLaunchKernelGGL(split16_rows_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, in, n, d, layout, (uint16_t*)out);
}

void launch_pack_split_rows(const float* src, long rows, long cols, void* dst, int out_dtype, hipStream_t s) {
    const int grid = cap_grid((rows * cols + 255) / 256);
    if (out_dtype == DT_F16) hipLaunchKernelGGL(pack_split_rows_kernel<f16_t>, dim3(grid), dim3(256), 0, s, src, rows, cols, (uint16_t*)dst);
    else hipLaunchKernelGGL(pack_split_rows_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, src, rows, cols, (uint16_t*)dst);
}

void launch_layernorm_q8(const float* x, const float* g, const float* b, void* q, float* scale, void* out16, int out16_dtype,
                         int T, int d, float eps, hipStream_t s) {
#define LQ_CASE(NV)                                                                                                  \
    if (out16_dtype == DT_F16)                                                                                       \
        hipLaunchKernelGGL((layernorm_q8_kernel<NV, f16_t>), dim3((T + 3) / 4), dim3(256), 0, s, x, g, b, (uint8_t*)q, scale, \
                           (uint16_t*)out16, T, d, eps);                                                             \
    else                                                                                                             \
        hipLaunchKernelGGL((layernorm_q8_kernel<NV, bf16_t>), dim3((T + 3) / 4), dim3(256), 0, s, x, g, b, (uint8_t*)q, scale, \
                           (uint16_t*)out16, T, d, eps);
    const int nv = (d + 255) / 256;
    if (nv <= 1) { LQ_CASE(1) } else if (nv <= 2) { LQ_CASE(2) } else if (nv <= 3) { LQ_CASE(3) }
    else if (nv <= 4) { LQ_CASE(4) } else if (nv <= 8) { LQ_CASE(8) } else if (nv <= 10) { LQ_CASE(10) }
    else { LQ_CASE(16) }
#undef LQ_CASE
}

void launch_absmax16(const void* in, long numel, int dtype, unsigned* out_bits, hipStream_t s) {
    const int grid = cap_grid((numel / 2 + 255) / 256);
    if (dtype == DT_F16) hipLaunchKernelGGL(absmax16_kernel<f16_t>, dim3(grid), dim3(256), 0, s, (const uint32_t*)in, numel / 2, out_bits);
    else hipLaunchKernelGGL(absmax16_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const uint32_t*)in, numel / 2, out_bits);
}

void launch_lnf_pool(const float* x, const float* g, const float* b, const int* seq_off, const int* seq_len,
                     const int* pad_left, int B, int d, float eps, int apply_ln, int mode, int normalize,
                     const float* pos_weights, int pos_weights_n, float* out, hipStream_t s, int* nonfinite_flag) {
    const size_t sm = (size_t)(4 * d + 8) * sizeof(float);
#define LP_CASE(NV)                                                                                              \
    hipLaunchKernelGGL((lnf_pool_kernel<NV>), dim3(B), dim3(256), sm, s, x, g, b, seq_off, seq_len, pad_left, d, \
                       eps, apply_ln, mode, normalize, pos_weights, pos_weights_n, out, nonfinite_flag);
    const int nv = (d + 255) / 256;
    if (nv <= 1) { LP_CASE(1) } else if (nv <= 2) { LP_CASE(2) } else if (nv <= 3) { LP_CASE(3) }
    else if (nv <= 4) { LP_CASE(4) } else if (nv <= 8) { LP_CASE(8) } else if (nv <= 10) { LP_CASE(10) }
    else { LP_CASE(16) }
#undef LP_CASE
}

void launch_pool(const void* hidden, int dtype, const int* mask, int B, int S, int d, int mode,
                 const float* pos_weights, float* out, hipStream_t s) {
    dim3 grid(B, (d / 4 + 255) / 256);
    if (dtype == DT_BF16)
        hipLaunchKernelGGL(pool_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)hidden, mask, S, d, mode, pos_weights, out);
    else if (dtype == DT_F16)
        hipLaunchKernelGGL(pool_kernel<f16_t>, grid, dim3(256), 0, s, (const f16_t*)hidden, mask, S, d, mode, pos_weights, out);
    else
        hipLaunchKernelGGL(pool_kernel<float>, grid, dim3(256), 0, s, (const float*)hidden, mask, S, d, mode, pos_weights, out);
}

void launch_fp8_quant_rows(const float* w, long rows, long cols, void* q, float* scale, hipStream_t s) {
    hipLaunchKernelGGL(fp8_quant_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, w, rows, cols,
                       (uint8_t*)q, scale);
}

void launch_fp8_dequant_rows(const void* q, const float* scale, long rows, long cols, void* out, int out_dtype,
                             hipStream_t s) {
    const int grid = cap_grid((rows * cols / 4 + 255) / 256);
    if (out_dtype == 1)
        hipLaunchKernelGGL(fp8_dequant_rows_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const uint8_t*)q, scale, rows,
                           cols, (bf16_t*)out);
    else
        hipLaunchKernelGGL(fp8_dequant_rows_kernel<float>, dim3(grid), dim3(256), 0, s, (const uint8_t*)q, scale, rows,
                           cols, (float*)out);
}

void launch_gather_rows(const float* src, const int* row_idx, int n, int d, float* dst, hipStream_t s) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, s, src, row_idx, n, d, dst);
}

void launch_logprob_rows(const float* logits, long ld, int V, const int* targets, int n, float* out_lp, int* out_arg,
                         hipStream_t s) {
    hipLaunchKernelGGL(logprob_rows_kernel, dim3(n), dim3(256), 0, s, logits, ld, V, targets, out_lp, out_arg);
}

void launch_mean_over_axis0(const float* in, int n0, long n, float* out, hipStream_t s) {
    hipLaunchKernelGGL(mean_axis0_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, n0, n, out);
}

void launch_pairwise(const float* a, const float* b, long n, int d, bool cosine, float* out, hipStream_t s) {
    const unsigned grid = (unsigned)((n + 3) / 4);
    if (cosine) hipLaunchKernelGGL(pairwise_kernel<true>, dim3(grid), dim3(256), 0, s, a, b, n, d, out);
    else hipLaunchKernelGGL(pairwise_kernel<false>, dim3(grid), dim3(256), 0, s, a, b, n, d, out);
}

void launch_l2norm(const float* in, long n, int d, void* out, int out_dtype, hipStream_t s) {
    const int grid = (int)((n + 3) / 4);
    if (out_dtype == DT_BF16) hipLaunchKernelGGL(l2norm_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, in, n, d, (bf16_t*)out);
    else if (out_dtype == DT_F16) hipLaunchKernelGGL(l2norm_kernel<f16_t>, dim3(grid), dim3(256), 0, s, in, n, d, (f16_t*)out);
    else hipLaunchKernelGGL(l2norm_kernel<float>, dim3(grid), dim3(256), 0, s, in, n, d, (float*)out);
}

void launch_f32_to_16(const float* in, long numel, void* out, int out_dtype, hipStream_t s) {
    const int grid = cap_grid((numel + 255) / 256);
    if (out_dtype == DT_F16) hipLaunchKernelGGL(cvt16_kernel<f16_t>, dim3(grid), dim3(256), 0, s, in, numel, (uint16_t*)out);
    else hipLaunchKernelGGL(cvt16_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, in, numel, (uint16_t*)out);
}
void launch_f32_to_bf16(const float* in, long numel, void* out, hipStream_t s) { launch_f32_to_16(in, numel, out, DT_BF16, s); }

void launch_absmax(const float* in, long numel, unsigned* out_bits, hipStream_t s) {
    hipLaunchKernelGGL(absmax_kernel, dim3(cap_grid((numel + 255) / 256)), dim3(256), 0, s, in, numel, out_bits);
}

void launch_fill_f32(float* p, long n, float v, hipStream_t s) {
    hipLaunchKernelGGL(fill_kernel, dim3(cap_grid((n + 255) / 256)), dim3(256), 0, s, p, n, v);
}

void launch_fill_rand(void* p, long n, int dtype, unsigned seed, float scale, hipStream_t s) {
    if (dtype == DT_BF16)
        hipLaunchKernelGGL(fill_rand_kernel<bf16_t>, dim3(cap_grid((n + 255) / 256)), dim3(256), 0, s, (bf16_t*)p, n, seed, scale);
    else if (dtype == DT_F16)
        hipLaunchKernelGGL(fill_rand_kernel<f16_t>, dim3(cap_grid((n + 255) / 256)), dim3(256), 0, s, (f16_t*)p, n, seed, scale);
    else
        hipLaunchKernelGGL(fill_rand_kernel<float>, dim3(cap_grid((n + 255) / 256)), dim3(256), 0, s, (float*)p, n, seed, scale);
}

void launch_rope(void* qk, int dtype, long ld, long k_off, const int* pos, const float* sin_t, const float* cos_t, int T,
                 int H, int dh, int rotary_dim, hipStream_t s) {
    const int half = rotary_dim / 2;
    const long n = (long)T * H * half;
    const int grid = (int)((n + 255) / 256);
    if (dtype == DT_BF16)
        hipLaunchKernelGGL(rope_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (bf16_t*)qk, ld, k_off, pos, sin_t, cos_t, T, H, dh, half);
    else if (dtype == DT_F16)
        hipLaunchKernelGGL(rope_kernel<f16_t>, dim3(grid), dim3(256), 0, s, (f16_t*)qk, ld, k_off, pos, sin_t, cos_t, T, H, dh, half);
    else
        hipLaunchKernelGGL(rope_kernel<float>, dim3(grid), dim3(256), 0, s, (float*)qk, ld, k_off, pos, sin_t, cos_t, T, H, dh, half);
}

void launch_qkv_deinterleave(const float* src, float* dst, int H, int dh, long row_len, hipStream_t s) {
    hipLaunchKernelGGL(qkv_deinterleave_kernel, dim3(3 * H * dh), dim3(256), 0, s, src, dst, H, dh, row_len);
}
