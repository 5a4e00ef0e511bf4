// One-wave-per-row LayerNorm in registers, shared by the streaming kernels (elementwise.hip) and by the query-sized
// projections that normalise their own A rows (qgemm.hip): the SAME loads, reductions and arithmetic, so a LayerNorm that
// runs inside a projection's prologue produces the bits of the stand-alone kernel.
#pragma once
#include "common.h"

#ifndef SGPT_LN_NT_LOAD
#define SGPT_LN_NT_LOAD 1   // LayerNorm / ln_f+pool read the residual stream with non-temporal loads: x is not needed again
                            // before the next residual epilogue, and leaving the caches to `a` (the next GEMM's operand) is
                            // +2.9 % end to end (36.07 k -> 37.11 k sentences/s, A/B on one box)
#endif

// row statistics + normalise in registers: nn.LayerNorm(eps) (HF:gpt_neo:317-319,385,492)
// NT = non-temporal row loads (the bulk kernels); the query-sized prologues re-read a row once per column tile: cached loads
template <int NV, bool NT = (SGPT_LN_NT_LOAD != 0)>
struct RowLN {
    float4 v[NV];
    __device__ __forceinline__ void load(const float* __restrict__ xr, int d, int lane) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            v[i] = c < d ? ldg16<NT>(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // statistics of the row (the butterfly of wave_sum; FAST: the same partners and order through VALU lane exchanges -- a + b
    // is commutative, so the bits do not depend on which instruction fetched the partner's value)
    template <bool FAST = false>
    __device__ __forceinline__ void stats(int d, int lane, float& mean, float& rstd, float eps) const {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        mean = (FAST ? wave_sum_valu(s) : wave_sum(s)) / (float)d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < d) {
                const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
                q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
        }
        rstd = 1.0f / sqrtf((FAST ? wave_sum_valu(q) : wave_sum(q)) / (float)d + eps);
    }
    __device__ __forceinline__ void scale(int i, float mean, float rstd, const float4& gg, const float4& bb) {
        v[i].x = (v[i].x - mean) * rstd * gg.x + bb.x;
        v[i].y = (v[i].y - mean) * rstd * gg.y + bb.y;
        v[i].z = (v[i].z - mean) * rstd * gg.z + bb.z;
        v[i].w = (v[i].w - mean) * rstd * gg.w + bb.w;
    }
    // FAST: the statistics' butterflies on the VALU (same partners, same order, same bits) -- for launches whose few waves wait on
    // the reductions themselves (ln_f + pool of a handful of query sequences)
    template <bool FAST = false>
    __device__ __forceinline__ void normalize(const float* __restrict__ g, const float* __restrict__ b, int d,
                                              float eps, int lane) {
        float mean, rstd;
        stats<FAST>(d, lane, mean, rstd, eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < d) scale(i, mean, rstd, *reinterpret_cast<const float4*>(g + c), *reinterpret_cast<const float4*>(b + c));
        }
    }
    // the same with gamma / beta already in registers (gg[i], bb[i] = the lane's float4 of column block i) and the VALU butterfly:
    // the query-sized prologues fetch every operand of the tile's rows in ONE memory round trip (qgemm.hip)
    __device__ __forceinline__ void normalize_pre(const float4* gg, const float4* bb, int d, float eps, int lane) {
        float mean, rstd;
        stats<true>(d, lane, mean, rstd, eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < d) scale(i, mean, rstd, gg[i], bb[i]);
        }
    }
};

// R rows normalised together (gamma / beta in registers): RowLN::normalize_pre on each row, with the rows' reductions interleaved
// (wave_sum_valu_multi) -- the same operations on every row, the same bits.
template <int NV, int R, bool NT>
__device__ __forceinline__ void rowln_normalize_rows(RowLN<NV, NT> (&r)[R], const float4* gg, const float4* bb, int d, float eps, int lane) {
    float s[R], q[R], mean[R], rstd[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
        s[u] = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) s[u] += (r[u].v[i].x + r[u].v[i].y) + (r[u].v[i].z + r[u].v[i].w);
    }
    wave_sum_valu_multi<R>(s);
#pragma unroll
    for (int u = 0; u < R; ++u) {
        mean[u] = s[u] / (float)d;
        q[u] = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < d) {
                const float a0 = r[u].v[i].x - mean[u], a1 = r[u].v[i].y - mean[u], a2 = r[u].v[i].z - mean[u], a3 = r[u].v[i].w - mean[u];
                q[u] += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
        }
    }
    wave_sum_valu_multi<R>(q);
#pragma unroll
    for (int u = 0; u < R; ++u) rstd[u] = 1.0f / sqrtf(q[u] / (float)d + eps);
#pragma unroll
    for (int u = 0; u < R; ++u)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < d) r[u].scale(i, mean[u], rstd[u], gg[i], bb[i]);
        }
}
