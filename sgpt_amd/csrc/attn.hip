// Causal (+ sliding-window) self-attention over packed variable-length sequences:
// GPTNeoSelfAttention._attn (HF:gpt_neo/modeling_gpt_neo.py:105-130): scores = q.k^T (NO
// 1/sqrt(dh) for GPT-Neo; `scale` carries it for other families), causal / local-window mask,
// fp32 softmax, P.V.  Right/left padding never reaches a real token (pad keys are masked,
// :119-120), so only the rows of each sequence's own span are touched.
//
// 16-bit path (attn16_lds_kernel<H>, H = bf16 or f16 operands): one wave per 16 query rows.
//   S^T = K.Q^T via v_mfma_f32_16x16x32_{bf16,f16} with K rows as the A-operand: a lane then owns ONE
//   query (column lane&15) and keys {4g+r} of each 16-key tile, so
//     - the row max / row sum are in-lane reductions + 2 shuffles (xor 16, 32),
//     - the 8 probabilities a lane holds for a 32-key step ARE the B-operand fragment of
//       O^T = V^T.P^T under the k-slot permutation  slot(g,j) <-> key 16*(j>>2) + 4g + (j&3);
//       the A-operand (V^T rows, written transposed by the QKV GEMM epilogue) applies the same
//       permutation with two 8-byte loads along the token axis.  No P round trip through LDS.
//   Online softmax (running max m, running sum l) over 64-key tiles; masked scores are -inf and
//   m starts at -1e30 so a fully masked step contributes exp(-inf) = 0 without NaN.
//
// fp32 path (attn_f32_kernel): exact-fp32 VALU kernel for the parity gate, one wave per query row.
#include <cstdlib>

#include "common.h"

// Build-time A/B switches (SGPT_LIB_TAG=x SGPT_EXTRA_FLAGS=-D... python -m sgpt_amd.build; scripts/seq_ab.sh, mid_ab.sh); the
// defaults are what the measurements in profiles/r03_attn_pmc.txt and r03_mid_batch_ab.txt selected.
#ifndef SGPT_ATTN_PERMLANE
#define SGPT_ATTN_PERMLANE 1
#endif
#ifndef SGPT_ATTN_SHORT
#define SGPT_ATTN_SHORT 1
#endif
#ifndef SGPT_ATTN_W16
#define SGPT_ATTN_W16 1
#endif
#ifndef SGPT_ATTN_Q32
#define SGPT_ATTN_Q32 0   // 1: also build / use the two-fragments-per-wave variant for head_dim 64 (A/B builds)
#endif
#ifndef SGPT_ATTN_Q32_WAVES
#define SGPT_ATTN_Q32_WAVES 3   // 4-wave blocks: resident blocks per CU
#endif
#ifndef SGPT_ATTN_Q32_MINLEN
#define SGPT_ATTN_Q32_MINLEN 64
#endif
#ifndef SGPT_ATTN_STAGES
#define SGPT_ATTN_STAGES 1
#endif
#ifndef SGPT_ATTN_NT_LOAD
#define SGPT_ATTN_NT_LOAD 1   // K / V^T tiles are read once per (sequence, head): non-temporal loads (+0.2 % end to end)
#endif
constexpr bool ATTN_NT_LOAD = SGPT_ATTN_NT_LOAD != 0;
#ifndef SGPT_ATTN_WAVES_PER_SIMD
#define SGPT_ATTN_WAVES_PER_SIMD 4   // register budget of attn16_lds_kernel: 4 waves per SIMD = two 512-thread workgroups per CU
#endif
constexpr int ATTN_WAVES_PER_SIMD = SGPT_ATTN_WAVES_PER_SIMD;

namespace {

#ifndef SGPT_ATTN_NT
#define SGPT_ATTN_NT 1
#endif
constexpr bool ATTN_NT = SGPT_ATTN_NT != 0;

// max over the four 16-lane rows of a wave (lane ^ 16, lane ^ 32) with gfx950's VALU row swaps instead of two ds_bpermute
// round trips through the LDS queue: v_permlane16_swap exchanges the odd rows of its first operand with the even rows of
// its second, v_permlane32_swap the upper half of the first with the lower half of the second.
#if SGPT_ATTN_PERMLANE
__device__ __forceinline__ float xor16_max(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
#else
__device__ __forceinline__ float xor16_max(float v) { return fmaxf(v, __shfl_xor(v, 16, 64)); }
__device__ __forceinline__ float xor32_max(float v) { return fmaxf(v, __shfl_xor(v, 32, 64)); }
#endif

// ---- K / V^T tiles of 64 keys staged in LDS and shared by the block's 8 waves ----
// (A direct-from-global predecessor re-fetched every K/V fragment once per wave with 16 scattered 64-byte pieces
// per instruction and one dependent global round trip per 8 MFMAs: 2.4 % MFMA-busy; removed in round 2.)
// A 512-thread block (128 queries of one (sequence, head)) loads
// each 64-key tile ONCE with row-contiguous 16-byte loads into LDS
//     Ks[64 keys][DH]   (16-B chunk index XOR (key & 7): conflict-free ds_read_b128 fragments)
//     Vs[DH][64 keys]   (same swizzle; keys k-slot-permuted inside each 32-key block: one ds_read_b128 per P.V fragment)
// and every wave takes its MFMA operands from there.  Same math / lane maps as above.
// H = bf16_t | f16_t: the 16-bit format of q / k / V^T, of the probabilities fed to the P.V MFMA and of the context
// OUT8: the context leaves as e4m3 codes of ctx / out_scale (fp8 out-projection operand) instead of the 16-bit format
// NQ = 16-query fragments per wave.  NQ = 1: 8 waves x 16 queries.  NQ = 2 (DH = 64): 4 waves x 32 queries -- every K / V^T
// fragment read from LDS feeds two MFMAs, halving the LDS bytes per query; the SQ counters at S = 512 put the LDS array
// right behind the VALU as this kernel's busiest unit (profiles/r03_attn_pmc.txt).
// MODE 0: the kernel as described.  MODE 1: the context additionally leaves as a split-precision pair (hi at ctx, lo = round16(v - hi)
// at ctx + ctx_lo_delta, a second hi at ctx + ctx_hi2_delta: the [hi | lo | hi] row the split out-projection contracts over).
// MODE 2 ("x3" attention): q, k, V^T and the probabilities ALL enter their MFMAs as hi + lo pairs of 16-bit values --
// S = k_hi.q_hi + k_lo.q_hi + k_hi.q_lo and O += v_hi.p_hi + v_lo.p_hi + v_hi.p_lo, i.e. products to ~2^-22 on the 16-bit MFMA;
// the lo halves of q | k and V^T sit qk_lo_delta / v_lo_delta elements behind the hi halves (written by the projections'
// split epilogues).  Twice the LDS and three times the MFMAs of an attention that is < 2 % of a block's FLOPs; 2 waves per
// SIMD (the second fragment set does not fit 128 VGPRs).  Context split as MODE 1 when ctx_lo_delta != 0.
template <typename H, int DH, bool OUT8, int NQ, int NW = (NQ == 1 ? 8 : 4), int MODE = 0>
__global__ __launch_bounds__(64 * NW, MODE == 2 ? 2 : (NQ == 2 ? SGPT_ATTN_Q32_WAVES : (NW <= 2 ? 3 : ATTN_WAVES_PER_SIMD)))   // (NW = 2: 4 staging loads per thread)
void attn16_lds_kernel(const AttnArgs p) {
    constexpr bool X3 = MODE == 2;
    static_assert(!(MODE != 0 && OUT8), "split-precision modes write 16-bit contexts");
    constexpr int KS = DH / 32, DT = DH / 16, CPR = DH / 8;  // CPR = 16-B chunks per K row
    constexpr int NT = 64 * NW, QW = 16 * NQ, QB = NW * QW;    // NW waves per block, QW queries per wave, QB per block
    constexpr int ORS = DH * 2 + 16;                           // output-transpose row stride (bytes)
    constexpr int NB = (DH <= 64 && SGPT_ATTN_STAGES == 2) ? 2 : 1;   // LDS stages of the K / V^T tiles
    __shared__ __attribute__((aligned(16))) uint4 Ks[NB][64 * CPR];
    __shared__ __attribute__((aligned(16))) uint4 Vs[NB][DH * 8];
    __shared__ __attribute__((aligned(16))) uint4 KsL[X3 ? 64 * CPR : 1];     // lo halves of the key / V^T tile (MODE 2)
    __shared__ __attribute__((aligned(16))) uint4 VsL[X3 ? DH * 8 : 1];
    __shared__ __attribute__((aligned(16))) char Os[NW][16 * ORS];
    const int sq = blockIdx.z, head = blockIdx.y;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int s0 = p.seq_off[sq];
    const int alloc = p.seq_off[sq + 1] - s0;
    const int qb0 = blockIdx.x * QB;
    if (qb0 >= alloc) return;                       // whole block out of range (uniform)
    const int q0 = qb0 + wave * QW;
    const bool wave_on = q0 < alloc;
    const int fr = lane & 15, g = lane >> 4;

    const bf16_t* __restrict__ qb = static_cast<const bf16_t*>(p.q) + (long)head * DH;
    const bf16_t* __restrict__ kb = static_cast<const bf16_t*>(p.k) + (long)head * DH;
    const bf16_t* __restrict__ vt = static_cast<const bf16_t*>(p.v) + (long)head * DH * p.ldvt;

    // (query rows past the allocation belong to the next sequence or to the padding of the token axis: loaded, never stored)
    uint4 qf[NQ][KS];
#pragma unroll
    for (int f = 0; f < NQ; ++f)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            qf[f][ks] = *reinterpret_cast<const uint4*>(qb + (long)(s0 + q0 + 16 * f + fr) * p.ldq + ks * 32 + 8 * g);
    uint4 qfl[X3 ? NQ : 1][X3 ? KS : 1];
    if constexpr (X3) {
#pragma unroll
        for (int f = 0; f < NQ; ++f)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                qfl[f][ks] = *reinterpret_cast<const uint4*>(qb + p.qk_lo_delta + (long)(s0 + q0 + 16 * f + fr) * p.ldq + ks * 32 + 8 * g);
    }

    f32x4 o[NQ][DT];
    float m_run[NQ], l_run[NQ];
#pragma unroll
    for (int f = 0; f < NQ; ++f) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[f][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        m_run[f] = -1e30f; l_run[f] = 0.f;
    }
    const float slope = p.alibi ? p.alibi[head] : 0.f;

    int j_lo = 0;
    if (p.window > 0) { j_lo = qb0 - p.window + 1; j_lo = j_lo < 0 ? 0 : (j_lo & ~63); }
    int j_hi = qb0 + QB - 1;                         // last key any query of the block may see
    if (j_hi > alloc - 1) j_hi = alloc - 1;
    // ---- cooperative tile load (row-contiguous 16-B pieces), one tile ahead: the global loads of key tile j+1 are
    // issued before tile j is consumed from LDS, so a sequence of several key tiles (S >= 128) does not pay one
    // un-hidden global-load round trip per tile ----
    constexpr int KU = (64 * CPR + NT - 1) / NT, VU = (DH * 8 + NT - 1) / NT;
    uint4 kreg[KU], vreg[VU], kregl[X3 ? KU : 1], vregl[X3 ? VU : 1];
    auto tile_load = [&](int j0) {
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int c = t + NT * u, row = c / CPR, ch = c % CPR;
            if (c < 64 * CPR) {
                kreg[u] = ldg16u<ATTN_NT_LOAD>(kb + (long)(s0 + j0 + row) * p.ldq + ch * 8);
                if constexpr (X3) kregl[u] = ldg16u<ATTN_NT_LOAD>(kb + p.qk_lo_delta + (long)(s0 + j0 + row) * p.ldq + ch * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < VU; ++u) {
            const int c = t + NT * u, row = c >> 3, ch = c & 7;
            if (c < DH * 8) {
                vreg[u] = ldg16u<ATTN_NT_LOAD>(vt + (long)row * p.ldvt + s0 + j0 + ch * 8);
                if constexpr (X3) vregl[u] = ldg16u<ATTN_NT_LOAD>(vt + p.v_lo_delta + (long)row * p.ldvt + s0 + j0 + ch * 8);
            }
        }
    };
    auto tile_store = [&](int b) {
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int c = t + NT * u, row = c / CPR, ch = c % CPR;
            if (c < 64 * CPR) {
                Ks[b][row * CPR + (ch ^ (row & 7))] = kreg[u];
                if constexpr (X3) KsL[row * CPR + (ch ^ (row & 7))] = kregl[u];
            }
        }
#pragma unroll
        for (int u = 0; u < VU; ++u) {
            const int c = t + NT * u, row = c >> 3, ch = c & 7;
            if (c < DH * 8) {
                // keys are stored k-slot-permuted inside each 32-key block: the 16-byte chunk 4*step + g of a row holds keys
                // 32*step + 4g..4g+3 and 32*step + 16 + 4g..4g+3 -- exactly the eight k-slots lane group g feeds to the P.V
                // MFMA, so a fragment is ONE ds_read_b128 (it was two ds_read_b64 from chunks two apart, and with them most
                // of this kernel's time at S >= 256).  The 16 loaded bytes (keys 8ch..8ch+7) go to two chunks, 8 bytes each.
                const int blk = ch >> 2, w = ch & 3, hf = w >> 1, g0 = 2 * (w & 1);
                char* vrow = reinterpret_cast<char*>(&Vs[b][row * 8]);
                *reinterpret_cast<uint2*>(vrow + ((4 * blk + g0) ^ (row & 7)) * 16 + hf * 8) = make_uint2(vreg[u].x, vreg[u].y);
                *reinterpret_cast<uint2*>(vrow + ((4 * blk + g0 + 1) ^ (row & 7)) * 16 + hf * 8) = make_uint2(vreg[u].z, vreg[u].w);
                if constexpr (X3) {
                    char* vrl = reinterpret_cast<char*>(&VsL[row * 8]);
                    *reinterpret_cast<uint2*>(vrl + ((4 * blk + g0) ^ (row & 7)) * 16 + hf * 8) = make_uint2(vregl[u].x, vregl[u].y);
                    *reinterpret_cast<uint2*>(vrl + ((4 * blk + g0 + 1) ^ (row & 7)) * 16 + hf * 8) = make_uint2(vregl[u].z, vregl[u].w);
                }
            }
        }
    };
    // NB = 2: two LDS stages, ONE barrier per key tile -- tile j+1 is written into the other stage behind tile j's MFMAs,
    // and the barrier at the end of the step both publishes it and retires stage j.  Measured: no gain (the waves do not
    // wait for the barriers, profiles/r03_attn_pmc.txt), so NB = 1 is what ships: store, barrier, consume, barrier.
    int cur = 0;
    if (j_lo <= j_hi) {
        tile_load(j_lo);
        if constexpr (NB == 2) { tile_store(0); __syncthreads(); }
    }
    const float c2 = p.scale * 1.44269504088896341f;
    const float s2 = slope * 1.44269504088896341f;
    for (int j0 = j_lo; j0 <= j_hi; j0 += 64) {
        if constexpr (NB == 1) {
            __syncthreads();                         // previous tile fully consumed
            tile_store(0);
            __syncthreads();
        }
        const bool more = j0 + 64 <= j_hi;
        if (more) tile_load(j0 + 64);                // next tile's loads fly under this tile's MFMAs and softmax
        if (wave_on && j0 <= q0 + QW - 1) {          // else: nothing visible for this wave in this tile
        const uint4* __restrict__ Kc = Ks[cur];
        const uint4* __restrict__ Vc = Vs[cur];
        // ---- S^T = K.Q^T : 4 tiles of [16 keys][16 queries] per fragment ----
        f32x4 s[NQ][4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
            for (int f = 0; f < NQ; ++f) s[f][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int row = nt * 16 + fr;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const uint4 kv = Kc[row * CPR + ((ks * 4 + g) ^ (row & 7))];
#pragma unroll
                for (int f = 0; f < NQ; ++f) s[f][nt] = Half<H>::mfma16(kv, qf[f][ks], s[f][nt]);
                if constexpr (X3) {
                    const uint4 kl = KsL[row * CPR + ((ks * 4 + g) ^ (row & 7))];
#pragma unroll
                    for (int f = 0; f < NQ; ++f) {
                        s[f][nt] = Half<H>::mfma16(kl, qf[f][ks], s[f][nt]);
                        s[f][nt] = Half<H>::mfma16(kv, qfl[f][ks], s[f][nt]);
                    }
                }
            }
        }
        // Softmax in the log2 domain: t = s * (scale * log2 e) [+ alibi * log2 e], p = 2^(t - m).  SQ counters at S = 512
        // (profiles/r03_attn_pmc.txt) put this kernel's VALU at 77 % busy with 4 waves per SIMD -- 255 VALU instructions per
        // 64-key tile and wave, half of them the per-score mask (key index, two compares, select, int -> float for the
        // ALiBi term).  A tile every query of the fragment sees whole (all but the diagonal tile of a fragment, and the
        // window's low edge) takes the lean path: one multiply per score.  Elsewhere the compares run against
        // compile-time offsets of one per-lane distance.
        uint32_t pw[NQ][8], pwl[X3 ? NQ : 1][8];
#pragma unroll
        for (int f = 0; f < NQ; ++f) {
            const int qf0 = q0 + 16 * f, qi = qf0 + fr;
            // (DH = 128: the second code path costs 14 spilled VGPRs at 4 waves per SIMD -- lean path for DH = 64 only)
            const bool full = DH <= 64 && (j0 + 63 <= qf0) && (p.window <= 0 || j0 > qf0 + 15 - p.window);
            float mx = -INFINITY;
            if (full && slope == 0.f) {                  // wave-uniform
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[f][nt][r] *= c2;
                    mx = fmaxf(mx, fmaxf(fmaxf(s[f][nt][0], s[f][nt][1]), fmaxf(s[f][nt][2], s[f][nt][3])));
                }
            } else {
                const int dq = qi - (j0 + 4 * g);         // key offset o = 16 nt + r is visible iff o <= dq (and o > dq - window)
                const int dw = p.window > 0 ? dq - p.window : -(1 << 30);
                const float ab = s2 * (float)(j0 + 4 * g);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int oo = nt * 16 + r;
                        const bool vis = (oo <= dq) && (oo > dw);
                        const float v = vis ? __builtin_fmaf(s[f][nt][r], c2, __builtin_fmaf(s2, (float)oo, ab)) : -INFINITY;
                        s[f][nt][r] = v;
                        mx = fmaxf(mx, v);
                    }
            }
            mx = xor16_max(mx);                      // the other three lane groups' keys of this query: two VALU lane swaps
            mx = xor32_max(mx);                      // (ds_bpermute round trips sat on every tile's dependent chain)
            const float m_new = fmaxf(m_run[f], mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run[f] - m_new);
            m_run[f] = m_new;
            float ps = 0.f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const float e0 = __builtin_amdgcn_exp2f(s[f][nt][0] - m_new), e1 = __builtin_amdgcn_exp2f(s[f][nt][1] - m_new);
                const float e2 = __builtin_amdgcn_exp2f(s[f][nt][2] - m_new), e3 = __builtin_amdgcn_exp2f(s[f][nt][3] - m_new);
                ps += (e0 + e1) + (e2 + e3);
                pw[f][nt * 2] = Half<H>::pack2(e0, e1);          // probabilities in [0, 1]: inside either format's range
                pw[f][nt * 2 + 1] = Half<H>::pack2(e2, e3);
                if constexpr (X3) {
                    pwl[f][nt * 2] = Half<H>::pack2(e0 - Half<H>::lo(pw[f][nt * 2]), e1 - Half<H>::hi(pw[f][nt * 2]));
                    pwl[f][nt * 2 + 1] = Half<H>::pack2(e2 - Half<H>::lo(pw[f][nt * 2 + 1]), e3 - Half<H>::hi(pw[f][nt * 2 + 1]));
                }
            }
            l_run[f] = l_run[f] * alpha + ps;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                o[f][dt][0] *= alpha; o[f][dt][1] *= alpha; o[f][dt][2] *= alpha; o[f][dt][3] *= alpha;
            }
        }
        // ---- O^T += V^T . P^T, two 32-key steps; k-slot j <-> key 32*step + 16*(j>>2) + 4g + (j&3) ----
#pragma unroll
        for (int step = 0; step < 2; ++step) {
            uint4 pu[NQ], pul[X3 ? NQ : 1];
#pragma unroll
            for (int f = 0; f < NQ; ++f) {
                pu[f].x = pw[f][step * 4]; pu[f].y = pw[f][step * 4 + 1]; pu[f].z = pw[f][step * 4 + 2]; pu[f].w = pw[f][step * 4 + 3];
                if constexpr (X3) {
                    pul[f].x = pwl[f][step * 4]; pul[f].y = pwl[f][step * 4 + 1]; pul[f].z = pwl[f][step * 4 + 2]; pul[f].w = pwl[f][step * 4 + 3];
                }
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int row = dt * 16 + fr;
                // chunk 4*step + g of the k-slot-permuted row = this lane group's eight k-slots (see the staging store)
                const uint4 vu = Vc[row * 8 + ((4 * step + g) ^ (row & 7))];
#pragma unroll
                for (int f = 0; f < NQ; ++f) o[f][dt] = Half<H>::mfma16(vu, pu[f], o[f][dt]);
                if constexpr (X3) {
                    const uint4 vl = VsL[row * 8 + ((4 * step + g) ^ (row & 7))];
#pragma unroll
                    for (int f = 0; f < NQ; ++f) {
                        o[f][dt] = Half<H>::mfma16(vl, pu[f], o[f][dt]);
                        o[f][dt] = Half<H>::mfma16(vu, pul[f], o[f][dt]);
                    }
                }
            }
        }
        }
        if constexpr (NB == 2) {
            if (more) tile_store(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }
    if (!wave_on) return;
    // O^T tile dt: lane holds head-dim elements dt*16 + 4g + r of query fr.  Transpose through a per-wave
    // LDS slice so every store instruction writes whole rows with 16 B per lane (the direct
    // 8-B-per-lane form wrote 3.6x the bytes: 32-B pieces of 16 different lines per instruction).
    // (The slice is private to the wave and LDS operations of one wave execute in order: fragment f + 1's writes queue
    // behind fragment f's reads.)
    char* os = Os[wave];
#pragma unroll
    for (int f = 0; f < NQ; ++f) {
        const int qf0 = q0 + 16 * f;
        if (qf0 >= alloc) break;
        float lr = l_run[f];
        lr += __shfl_xor(lr, 16, 64);
        lr += __shfl_xor(lr, 32, 64);
        const float inv = 1.0f / lr;
        if constexpr (OUT8) {
            constexpr int ORS8 = DH + 16, CPR8 = DH / 16, RPI8 = 64 / CPR8;
            const float sc = inv / p.out_scale;
            float amax = 0.f;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const float v0 = o[f][dt][0] * sc, v1 = o[f][dt][1] * sc, v2 = o[f][dt][2] * sc, v3 = o[f][dt][3] * sc;
                amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v0), fabsf(v1)), fmaxf(fabsf(v2), fabsf(v3))));
                *reinterpret_cast<uint32_t*>(os + fr * ORS8 + dt * 16 + 4 * g) = pack_fp8x4(v0, v1, v2, v3);
            }
            if (p.range_flag != nullptr && !(amax <= 448.f)) atomicOr(p.range_flag, 4);   // bit 2: attention context (bit 1: GELU output)
            uint8_t* obase = static_cast<uint8_t*>(p.ctx) + (long)(s0 + qf0) * p.ldo + (long)head * DH;
#pragma unroll
            for (int h = 0; h < 16 / RPI8; ++h) {
                const int row = h * RPI8 + lane / CPR8, ch = lane % CPR8;
                const uint4 v = *reinterpret_cast<const uint4_a*>(os + row * ORS8 + ch * 16);
                if (qf0 + row < alloc) gstore16<ATTN_NT>(obase + (long)row * p.ldo + ch * 16, v);   // rows past the allocation are the next sequence's
            }
        } else {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                *reinterpret_cast<uint2_a*>(os + fr * ORS + (dt * 16 + 4 * g) * 2) =
                    make_uint2(Half<H>::pack2(o[f][dt][0] * inv, o[f][dt][1] * inv), Half<H>::pack2(o[f][dt][2] * inv, o[f][dt][3] * inv));   // a convex
            // combination of V rows: bounded by max|V|, which the V projection's epilogue already range-checked (f16)
            constexpr int RPI = 64 / CPR;                    // rows per store instruction
            bf16_t* obase = static_cast<bf16_t*>(p.ctx) + (long)(s0 + qf0) * p.ldo + (long)head * DH;
#pragma unroll
            for (int h = 0; h < 16 / RPI; ++h) {
                const int row = h * RPI + lane / CPR, ch = lane % CPR;
                const uint4 v = *reinterpret_cast<const uint4_a*>(os + row * ORS + ch * 16);
                if (qf0 + row < alloc) gstore16<ATTN_NT>(obase + (long)row * p.ldo + ch * 8, v);   // (allocations are multiples of 8 rows)
            }
            if constexpr (MODE != 0) {
                if (p.ctx_lo_delta != 0) {
                    // split-precision context: lo = round16(v - hi) through the same per-wave slice (in order behind the reads
                    // above), then the second copy of hi -- the [hi | lo | hi] row of the split out-projection
#pragma unroll
                    for (int pass = 1; pass < 3; ++pass) {
                        if (pass == 2 && p.ctx_hi2_delta == 0) break;
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt) {
                            const float v0 = o[f][dt][0] * inv, v1 = o[f][dt][1] * inv, v2 = o[f][dt][2] * inv, v3 = o[f][dt][3] * inv;
                            const uint32_t h01 = Half<H>::pack2(v0, v1), h23 = Half<H>::pack2(v2, v3);
                            *reinterpret_cast<uint2_a*>(os + fr * ORS + (dt * 16 + 4 * g) * 2) = pass == 2 ? make_uint2(h01, h23) :
                                make_uint2(Half<H>::pack2(v0 - Half<H>::lo(h01), v1 - Half<H>::hi(h01)),
                                           Half<H>::pack2(v2 - Half<H>::lo(h23), v3 - Half<H>::hi(h23)));
                        }
                        bf16_t* ob2 = obase + (pass == 1 ? p.ctx_lo_delta : p.ctx_hi2_delta);
#pragma unroll
                        for (int h = 0; h < 16 / RPI; ++h) {
                            const int row = h * RPI + lane / CPR, ch = lane % CPR;
                            const uint4 v = *reinterpret_cast<const uint4_a*>(os + row * ORS + ch * 16);
                            if (qf0 + row < alloc) gstore16<ATTN_NT>(ob2 + (long)row * p.ldo + ch * 8, v);
                        }
                    }
                }
            }
        }
    }
}

// Exact fp32: one wave per query row.  Scores in LDS (max 2048 keys per wave).
constexpr int F32_MAXKEYS = 2048;
__global__ __launch_bounds__(256) void attn_f32_kernel(const AttnArgs p) {
    __shared__ float sc[4][F32_MAXKEYS];
    __shared__ float qs[4][256];
    const int sq = blockIdx.z, head = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s0 = p.seq_off[sq];
    const int alloc = p.seq_off[sq + 1] - s0;
    const int qi = blockIdx.x * 4 + wave;
    if (qi >= alloc) return;
    const int dh = p.dh;
    const float* __restrict__ qb = static_cast<const float*>(p.q) + (long)head * dh;
    const float* __restrict__ kb = static_cast<const float*>(p.k) + (long)head * dh;
    const float* __restrict__ vb = static_cast<const float*>(p.v) + (long)head * dh;
    for (int c = lane; c < dh; c += 64) qs[wave][c] = qb[(long)(s0 + qi) * p.ldq + c];
    int lo = 0;
    if (p.window > 0) { lo = qi - p.window + 1; lo = lo < 0 ? 0 : lo; }
    const int nkeys = qi - lo + 1;
    const float slope = p.alibi ? p.alibi[head] : 0.f;
    float mx = -INFINITY;
    for (int jj = lane; jj < nkeys; jj += 64) {
        const float* kr = kb + (long)(s0 + lo + jj) * p.ldq;
        float a = 0.f;
        for (int c = 0; c < dh; c += 4) {
            const float4 kv = *reinterpret_cast<const float4*>(kr + c);
            a = fmaf(qs[wave][c], kv.x, a); a = fmaf(qs[wave][c + 1], kv.y, a);
            a = fmaf(qs[wave][c + 2], kv.z, a); a = fmaf(qs[wave][c + 3], kv.w, a);
        }
        a = a * p.scale + slope * (float)(lo + jj);
        sc[wave][jj] = a;
        mx = fmaxf(mx, a);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int jj = lane; jj < nkeys; jj += 64) {
        const float e = expf(sc[wave][jj] - mx);
        sc[wave][jj] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    float* orow = static_cast<float*>(p.ctx) + (long)(s0 + qi) * p.ldo + (long)head * dh;
    for (int c = lane; c < dh; c += 64) {
        float a = 0.f;
        for (int jj = 0; jj < nkeys; ++jj) a = fmaf(sc[wave][jj], vb[(long)(s0 + lo + jj) * p.ldq + c], a);
        orow[c] = a * inv;
    }
}

}  // namespace

void launch_attn_bf16(const AttnArgs& a, hipStream_t s) {
    dim3 grid((a.max_alloc_len + 127) / 128, a.H, a.B);
    if (a.x3 || a.ctx_lo_delta != 0) {
        // split-precision variants (MODE 2: hi + lo q / k / V^T / p; MODE 1: 16-bit attention, split context).  head_dim 256
        // has no MODE 2 (its lo tiles do not fit the LDS): the caller keeps x3 = 0 there (attn_x3_supported)
        if (a.out_fp8 || (a.x3 && a.dh == 256)) abort();
#define ATTN_SPLIT_CASE(H)                                                                                                      \
        if (a.x3) {                                                                                                             \
            if (a.dh == 64) hipLaunchKernelGGL((attn16_lds_kernel<H, 64, false, 1, 8, 2>), grid, dim3(512), 0, s, a);           \
            else if (a.dh == 128) hipLaunchKernelGGL((attn16_lds_kernel<H, 128, false, 1, 8, 2>), grid, dim3(512), 0, s, a);    \
            else abort();                                                                                                       \
        } else {                                                                                                                \
            if (a.dh == 64) hipLaunchKernelGGL((attn16_lds_kernel<H, 64, false, 1, 8, 1>), grid, dim3(512), 0, s, a);           \
            else if (a.dh == 128) hipLaunchKernelGGL((attn16_lds_kernel<H, 128, false, 1, 8, 1>), grid, dim3(512), 0, s, a);    \
            else if (a.dh == 256) hipLaunchKernelGGL((attn16_lds_kernel<H, 256, false, 1, 8, 1>), grid, dim3(512), 0, s, a);    \
            else abort();                                                                                                       \
        }
        if (a.dtype == DT_F16) { ATTN_SPLIT_CASE(f16_t) } else { ATTN_SPLIT_CASE(bf16_t) }
#undef ATTN_SPLIT_CASE
        return;
    }
    // Long sequences, head_dim 64: 16-wave blocks of 256 queries stage every K / V^T tile once per 256 queries instead of
    // once per 128 (seq 512: 118.9 -> 117.4 ms per step; 64-query blocks, the other direction: -2 ... -8 %).  Only where the
    // last block of a sequence is at least half full (seq 300 = 256 + 48 queries: -1.4 %).
    if (SGPT_ATTN_W16 && a.dh == 64 && !a.out_fp8 && a.max_alloc_len > 384 && (a.max_alloc_len - 1) % 256 >= 128) {
        dim3 g16((a.max_alloc_len + 255) / 256, a.H, a.B);
        if (a.dtype == DT_F16) hipLaunchKernelGGL((attn16_lds_kernel<f16_t, 64, false, 1, 16>), g16, dim3(1024), 0, s, a);
        else hipLaunchKernelGGL((attn16_lds_kernel<bf16_t, 64, false, 1, 16>), g16, dim3(1024), 0, s, a);
        return;
    }
    // Short sequences, head_dim 64 (query batches: 4..32 tokens each): blocks of 2 / 4 waves instead of 8 -- a 128-query block
    // on a 24-token sequence launches six waves that only ever wait at barriers (1000 queries: 76 us per launch, 11 % of
    // the encode).  One block still covers a whole sequence.
    if (SGPT_ATTN_SHORT && a.dh == 64 && !a.out_fp8 && a.max_alloc_len <= 64) {
        const bool w2 = a.max_alloc_len <= 32;
        dim3 gs(1, a.H, a.B);
        if (a.dtype == DT_F16) {
            if (w2) hipLaunchKernelGGL((attn16_lds_kernel<f16_t, 64, false, 1, 2>), gs, dim3(128), 0, s, a);
            else hipLaunchKernelGGL((attn16_lds_kernel<f16_t, 64, false, 1, 4>), gs, dim3(256), 0, s, a);
        } else {
            if (w2) hipLaunchKernelGGL((attn16_lds_kernel<bf16_t, 64, false, 1, 2>), gs, dim3(128), 0, s, a);
            else hipLaunchKernelGGL((attn16_lds_kernel<bf16_t, 64, false, 1, 4>), gs, dim3(256), 0, s, a);
        }
        return;
    }
#if SGPT_ATTN_Q32
    // head_dim 64: two 16-query fragments per wave (4-wave blocks of the same 128 queries) once sequences have more than one
    // 64-key tile.  Measured against eight 16-query waves: seq 512 +0.6 %, seq 300 -0.4 %, seq 128 -0.3 % at three blocks per
    // CU, -5 % at two (profiles/r03_attn_pmc.txt) -- what the second fragment adds in independent work per wave it takes
    // away in resident waves.  Not used by default.
    const bool q32 = a.max_alloc_len > SGPT_ATTN_Q32_MINLEN;
#define ATTN_Q32_CASE(H, O8) if (a.dh == 64 && q32) hipLaunchKernelGGL((attn16_lds_kernel<H, 64, O8, 2>), grid, dim3(256), 0, s, a); else
#else
#define ATTN_Q32_CASE(H, O8)
#endif
#define ATTN_CASE(H, O8)                                                                                          \
    ATTN_Q32_CASE(H, O8)                                                                                          \
    if (a.dh == 64) hipLaunchKernelGGL((attn16_lds_kernel<H, 64, O8, 1>), grid, dim3(512), 0, s, a);          \
    else if (a.dh == 128) hipLaunchKernelGGL((attn16_lds_kernel<H, 128, O8, 1>), grid, dim3(512), 0, s, a);        \
    else if (a.dh == 256) hipLaunchKernelGGL((attn16_lds_kernel<H, 256, O8, 1>), grid, dim3(512), 0, s, a);        \
    else abort();
    if (a.out_fp8) { ATTN_CASE(bf16_t, true) }
    else if (a.dtype == DT_F16) { ATTN_CASE(f16_t, false) } else { ATTN_CASE(bf16_t, false) }
#undef ATTN_CASE
#undef ATTN_Q32_CASE
}

void launch_attn_f32(const AttnArgs& a, hipStream_t s) {
    dim3 grid((a.max_alloc_len + 3) / 4, a.H, a.B);
    hipLaunchKernelGGL(attn_f32_kernel, grid, dim3(256), 0, s, a);
}
