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
#ifndef SGPT_ATTN_ZZ
#define SGPT_ATTN_ZZ 1   // zig-zag fragment pairs (see attn16_lds_kernel) for head_dim 64, sequences of > SGPT_ATTN_ZZ_MINLEN rows
#endif
#ifndef SGPT_ATTN_ZZ_MINLEN
#define SGPT_ATTN_ZZ_MINLEN 128
#endif
#ifndef SGPT_ATTN_ZZ_NW
#define SGPT_ATTN_ZZ_NW 0   // A/B builds: 8 / 16 = always blocks of that many waves, whatever the pair count
#endif
#ifndef SGPT_ATTN_STAGES
#define SGPT_ATTN_STAGES 1
#endif
#ifndef SGPT_ATTN_PIPE
#define SGPT_ATTN_PIPE 0   // 1: software-pipelined schedule (next tile's K.Q^T under this tile's softmax) for head_dim 64, sequences of > SGPT_ATTN_PIPE_MINLEN rows
#endif
#ifndef SGPT_ATTN_PIPE_ALT
#define SGPT_ATTN_PIPE_ALT 1   // pipelined schedule: half of each SIMD's waves take (next scores | softmax + P.V) in the opposite order
#endif
#ifndef SGPT_ATTN_WINSKIP
#define SGPT_ATTN_WINSKIP 1   // local layers: a wave skips key tiles wholly below the window of its first query
#endif
#ifndef SGPT_ATTN_PIPE_MINLEN
#define SGPT_ATTN_PIPE_MINLEN 128
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
// NQ = 16-query fragments per wave, processed one after the other inside a key tile.  NQ = 1: fragment = queries q0 .. q0+15.
// ZZ (NQ = 2, "zig-zag"): wave w of block b owns the PAIR of fragments (i, nf-1-i), i = b NW + w, of a sequence of nf
// fragments.  Under the causal mask fragment f needs f/4 + 1 key tiles, so a block of consecutive fragments leaves its low
// waves idle behind barriers while the high ones walk the diagonal (S = 512: 36 of 48 wave-steps used; S = 300 in 128-query
// blocks: 55 of 88); the mirror pair makes every wave's work (i/4 + 1) + ((nf-1-i)/4 + 1) ~ constant, gives a wave two
// independent dependency chains wherever both fragments see the tile, and halves the blocks (and K / V^T stagings) per
// sequence.  Same arithmetic per (fragment, tile) in the same tile order: bits identical to NQ = 1.  Launch rule and numbers:
// launch_attn_bf16.  (Round 3's other two-fragment variant -- adjacent fragments sharing every K / V^T LDS read, +0.6 % at
// S = 512, build option SGPT_ATTN_Q32 -- was removed when this one replaced it.)
// MODE 0: the kernel as described.  MODE 1: the context additionally leaves as a split-precision pair (hi at ctx, lo = round16(v - hi)
// at ctx + ctx_lo_delta, a second hi at ctx + ctx_hi2_delta: the [hi | lo | hi] row the split out-projection contracts over).
// MODE 2 ("x3" attention): q, k, V^T and the probabilities ALL enter their MFMAs as hi + lo pairs of 16-bit values --
// S = k_hi.q_hi + k_lo.q_hi + k_hi.q_lo and O += v_hi.p_hi + v_lo.p_hi + v_hi.p_lo, i.e. products to ~2^-22 on the 16-bit MFMA;
// the lo halves of q | k and V^T sit qk_lo_delta / v_lo_delta elements behind the hi halves (written by the projections'
// split epilogues).  Twice the LDS and three times the MFMAs of an attention that is < 2 % of a block's FLOPs; 2 waves per
// SIMD (the second fragment set does not fit 128 VGPRs).  Context split as MODE 1 when ctx_lo_delta != 0.
template <typename H, int DH, bool OUT8, int NQ, int NW = 8, int MODE = 0, bool PIPE = false, bool ZZ = false>
__global__ __launch_bounds__(64 * NW, MODE == 2 ? 2 : (NW <= 2 ? 3 : ATTN_WAVES_PER_SIMD))   // (NW = 2: 4 staging loads per thread)
void attn16_lds_kernel(const AttnArgs p) {
    constexpr bool X3 = MODE == 2;
    static_assert(!(MODE != 0 && OUT8), "split-precision modes write 16-bit contexts");
    static_assert(!PIPE || (NQ == 1 && !X3 && DH <= 128), "pipelined variant: one fragment per wave, 16-bit operands");
    static_assert(ZZ ? NQ == 2 : NQ == 1, "two fragments per wave = the zig-zag pairing");
    constexpr int KS = DH / 32, DT = DH / 16, CPR = DH / 8;  // CPR = 16-B chunks per K row
    constexpr int NT = 64 * NW, QW = 16 * NQ, QB = NW * QW;    // NW waves per block, QW queries per wave, QB per block
    constexpr int ORS = DH * 2 + 16;                           // output-transpose row stride (bytes)
    constexpr int NB = (PIPE || (DH <= 64 && SGPT_ATTN_STAGES == 2)) ? 2 : 1;   // LDS stages of the K / V^T tiles
    __shared__ __attribute__((aligned(16))) uint4 Ks[NB][64 * CPR];
    __shared__ __attribute__((aligned(16))) uint4 Vs[NB][DH * 8];
    __shared__ __attribute__((aligned(16))) uint4 KsL[X3 ? 64 * CPR : 1];     // lo halves of the key / V^T tile (MODE 2)
    __shared__ __attribute__((aligned(16))) uint4 VsL[X3 ? DH * 8 : 1];
    __shared__ __attribute__((aligned(16))) char Os[NW][16 * ORS];
    const int sq = blockIdx.z, head = blockIdx.y;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int s0 = p.seq_off[sq];
    const int alloc = p.seq_off[sq + 1] - s0;
    const int fr = lane & 15, g = lane >> 4;
    // qpos[f] = first query row of the wave's fragment f (-1: none); qb0 = lowest query row of the block
    int qpos[NQ], qb0;
    bool wave_on;
    if constexpr (ZZ) {
        const int nf = (alloc + 15) >> 4, pi0 = blockIdx.x * NW, pi = pi0 + wave;
        if (2 * pi0 > nf - 1) return;                // whole block out of range (uniform)
        wave_on = 2 * pi <= nf - 1;
        qpos[0] = 16 * pi;
        qpos[1] = (nf - 1 - pi > pi) ? 16 * (nf - 1 - pi) : -1;   // (the middle fragment of an odd count is a pair of one)
        qb0 = 16 * pi0;
    } else {
        qb0 = blockIdx.x * QB;
        if (qb0 >= alloc) return;                    // whole block out of range (uniform)
        qpos[0] = qb0 + wave * QW;
        wave_on = qpos[0] < alloc;
    }
    const int q0 = qpos[0];

    const bf16_t* __restrict__ qb = static_cast<const bf16_t*>(p.q) + (long)head * DH;
    const bf16_t* __restrict__ kb = static_cast<const bf16_t*>(p.k) + (long)head * DH;
    const bf16_t* __restrict__ vt = static_cast<const bf16_t*>(p.v) + (long)head * DH * p.ldvt;

    // (query rows past the allocation belong to the next sequence or to the padding of the token axis: loaded, never stored)
    uint4 qf[NQ][KS];
#pragma unroll
    for (int f = 0; f < NQ; ++f)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            qf[f][ks] = *reinterpret_cast<const uint4*>(qb + (long)(s0 + (qpos[f] < 0 ? 0 : qpos[f]) + fr) * p.ldq + ks * 32 + 8 * g);
    uint4 qfl[X3 ? NQ : 1][X3 ? KS : 1];
    if constexpr (X3) {
#pragma unroll
        for (int f = 0; f < NQ; ++f)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                qfl[f][ks] = *reinterpret_cast<const uint4*>(qb + p.qk_lo_delta + (long)(s0 + (qpos[f] < 0 ? 0 : qpos[f]) + fr) * p.ldq + ks * 32 + 8 * g);
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
    int j_hi = ZZ ? 16 * (((alloc + 15) >> 4) - 1 - blockIdx.x * NW) + 15 : qb0 + QB - 1;   // last key any query of the block may see
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
    const float c2 = p.scale * 1.44269504088896341f;
    const float s2 = slope * 1.44269504088896341f;
    // ---- the three phases of one 64-key tile, for fragment f of the wave (f is a compile-time index after unrolling) ----
    // S^T = K.Q^T : 4 tiles of [16 keys][16 queries]
    auto qk_frag = [&](const uint4* __restrict__ Kc, const uint4 (&q)[KS], const uint4 (&ql)[X3 ? KS : 1], f32x4 (&s)[4]) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            s[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int row = nt * 16 + fr;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const uint4 kv = Kc[row * CPR + ((ks * 4 + g) ^ (row & 7))];
                s[nt] = Half<H>::mfma16(kv, q[ks], s[nt]);
                if constexpr (X3) {
                    const uint4 kl = KsL[row * CPR + ((ks * 4 + g) ^ (row & 7))];
                    s[nt] = Half<H>::mfma16(kl, q[ks], s[nt]);
                    s[nt] = Half<H>::mfma16(kv, ql[ks], s[nt]);
                }
            }
        }
    };
    // Softmax in the log2 domain: t = s * (scale * log2 e) [+ alibi * log2 e], p = 2^(t - m).  SQ counters at S = 512
    // (profiles/r03_attn_pmc.txt) put this kernel's VALU at 77 % busy with 4 waves per SIMD -- 255 VALU instructions per
    // 64-key tile and wave, half of them the per-score mask (key index, two compares, select, int -> float for the
    // ALiBi term).  A tile every query of the fragment sees whole (all but the diagonal tile of a fragment, and the
    // window's low edge) takes the lean path: one multiply per score.  Elsewhere the compares run against
    // compile-time offsets of one per-lane distance.
    auto softmax_frag = [&](int j0, int qf0, f32x4 (&s)[4], float& m_r, float& l_r, f32x4 (&oo_)[DT], uint32_t (&pw)[8], uint32_t (&pwl)[8]) {
        const int qi = qf0 + fr;
        // (DH = 128: the second code path costs 14 spilled VGPRs at 4 waves per SIMD -- lean path for DH = 64 only)
        const bool full = DH <= 64 && (j0 + 63 <= qf0) && (p.window <= 0 || j0 > qf0 + 15 - p.window);
        float mx = -INFINITY;
        if (full && slope == 0.f) {                  // wave-uniform
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s[nt][r] *= c2;
                mx = fmaxf(mx, fmaxf(fmaxf(s[nt][0], s[nt][1]), fmaxf(s[nt][2], s[nt][3])));
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
                    const float v = vis ? __builtin_fmaf(s[nt][r], c2, __builtin_fmaf(s2, (float)oo, ab)) : -INFINITY;
                    s[nt][r] = v;
                    mx = fmaxf(mx, v);
                }
        }
        mx = xor16_max(mx);                      // the other three lane groups' keys of this query: two VALU lane swaps
        mx = xor32_max(mx);                      // (ds_bpermute round trips sat on every tile's dependent chain)
        const float m_new = fmaxf(m_r, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_r - m_new);
        m_r = m_new;
        float ps = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const float e0 = __builtin_amdgcn_exp2f(s[nt][0] - m_new), e1 = __builtin_amdgcn_exp2f(s[nt][1] - m_new);
            const float e2 = __builtin_amdgcn_exp2f(s[nt][2] - m_new), e3 = __builtin_amdgcn_exp2f(s[nt][3] - m_new);
            ps += (e0 + e1) + (e2 + e3);
            pw[nt * 2] = Half<H>::pack2(e0, e1);          // probabilities in [0, 1]: inside either format's range
            pw[nt * 2 + 1] = Half<H>::pack2(e2, e3);
            if constexpr (X3) {
                pwl[nt * 2] = Half<H>::pack2(e0 - Half<H>::lo(pw[nt * 2]), e1 - Half<H>::hi(pw[nt * 2]));
                pwl[nt * 2 + 1] = Half<H>::pack2(e2 - Half<H>::lo(pw[nt * 2 + 1]), e3 - Half<H>::hi(pw[nt * 2 + 1]));
            }
        }
        l_r = l_r * alpha + ps;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            oo_[dt][0] *= alpha; oo_[dt][1] *= alpha; oo_[dt][2] *= alpha; oo_[dt][3] *= alpha;
        }
    };
    // O^T += V^T . P^T, two 32-key steps; k-slot j <-> key 32*step + 16*(j>>2) + 4g + (j&3)
    auto pv_frag = [&](const uint4* __restrict__ Vc, f32x4 (&oo_)[DT], const uint32_t (&pw)[8], const uint32_t (&pwl)[8]) {
#pragma unroll
        for (int step = 0; step < 2; ++step) {
            uint4 pu, pul;
            pu.x = pw[step * 4]; pu.y = pw[step * 4 + 1]; pu.z = pw[step * 4 + 2]; pu.w = pw[step * 4 + 3];
            if constexpr (X3) { pul.x = pwl[step * 4]; pul.y = pwl[step * 4 + 1]; pul.z = pwl[step * 4 + 2]; pul.w = pwl[step * 4 + 3]; }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int row = dt * 16 + fr;
                // chunk 4*step + g of the k-slot-permuted row = this lane group's eight k-slots (see the staging store)
                const uint4 vu = Vc[row * 8 + ((4 * step + g) ^ (row & 7))];
                oo_[dt] = Half<H>::mfma16(vu, pu, oo_[dt]);
                if constexpr (X3) {
                    const uint4 vl = VsL[row * 8 + ((4 * step + g) ^ (row & 7))];
                    oo_[dt] = Half<H>::mfma16(vl, pu, oo_[dt]);
                    oo_[dt] = Half<H>::mfma16(vu, pul, oo_[dt]);
                }
            }
        }
    };
    // fragment at query row qf0 has work in the key tile at j0 iff some query of it sees some key of the tile: not past the
    // diagonal, and (local layers) not wholly below the window of its first query -- a skipped tile would contribute p = 0
    // everywhere (same bits)
    auto frag_visible = [&](int qf0, int j0) {
        return wave_on && qf0 >= 0 && j0 <= j_hi && j0 <= qf0 + 15 && (!SGPT_ATTN_WINSKIP || p.window <= 0 || j0 + 63 > qf0 - p.window);
    };
    auto tile_compute = [&](const uint4* __restrict__ Kc, const uint4* __restrict__ Vc, int j0) {
#pragma unroll
        for (int f = 0; f < NQ; ++f) {
            if (frag_visible(qpos[f], j0)) {
                f32x4 s[4];
                uint32_t pw[8], pwl[8];
                qk_frag(Kc, qf[f], qfl[X3 ? f : 0], s);
                softmax_frag(j0, qpos[f], s, m_run[f], l_run[f], o[f], pw, pwl);
                pv_frag(Vc, o[f], pw, pwl);
            }
        }
    };
    if constexpr (PIPE) {
        // Software-pipelined schedule for sequences of several key tiles: a wave issues the K.Q^T MFMAs of
        // tile i+1 BEFORE the softmax of tile i, so the LDS fragment reads and the MFMA latency of the next scores run under
        // this tile's ~150 VALU instructions instead of in front of them.  K runs one tile ahead of V^T through two LDS
        // stages each: during step i the LDS holds K_{i+1} (Ks[(i+1)&1]) and V_i (Vs[i&1]); K_{i+2} and V_{i+1} (global
        // loads issued a step earlier) are written into the two retired stages at the start of the step and published by
        // the ONE barrier that ends it.  Same arithmetic per tile, same order over tiles: bits identical to the plain schedule.
        // Measured (profiles/r04_attn_ab.txt): S = 512 377 -> 413 us per launch, S = 300 315 -> 317 -- the waves are not
        // waiting for their own MFMAs.  Build option SGPT_ATTN_PIPE, off.  (What they WERE waiting for was found afterwards:
        // the plain loop's s_waitcnt vmcnt(0) in front of its K.Q^T MFMAs, see the explicit wait below.)
        uint4 kreg2[KU];
        auto k_load = [&](int j0, uint4 (&r)[KU]) {
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const int c = t + NT * u, row = c / CPR, ch = c % CPR;
                if (c < 64 * CPR) r[u] = ldg16u<ATTN_NT_LOAD>(kb + (long)(s0 + j0 + row) * p.ldq + ch * 8);
            }
        };
        auto k_store = [&](int b, const uint4 (&r)[KU]) {
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const int c = t + NT * u, row = c / CPR, ch = c % CPR;
                if (c < 64 * CPR) Ks[b][row * CPR + (ch ^ (row & 7))] = r[u];
            }
        };
        auto v_load = [&](int j0) {
#pragma unroll
            for (int u = 0; u < VU; ++u) {
                const int c = t + NT * u, row = c >> 3, ch = c & 7;
                if (c < DH * 8) vreg[u] = ldg16u<ATTN_NT_LOAD>(vt + (long)row * p.ldvt + s0 + j0 + ch * 8);
            }
        };
        auto v_store = [&](int b) {
#pragma unroll
            for (int u = 0; u < VU; ++u) {
                const int c = t + NT * u, row = c >> 3, ch = c & 7;
                if (c < DH * 8) {
                    const int blk = ch >> 2, w = ch & 3, hf = w >> 1, g0 = 2 * (w & 1);   // (k-slot permutation: see tile_store)
                    char* vrow = reinterpret_cast<char*>(&Vs[b][row * 8]);
                    *reinterpret_cast<uint2*>(vrow + ((4 * blk + g0) ^ (row & 7)) * 16 + hf * 8) = make_uint2(vreg[u].x, vreg[u].y);
                    *reinterpret_cast<uint2*>(vrow + ((4 * blk + g0 + 1) ^ (row & 7)) * 16 + hf * 8) = make_uint2(vreg[u].z, vreg[u].w);
                }
            }
        };
        if (j_lo <= j_hi) {
            k_load(j_lo, kreg); v_load(j_lo);
            if (j_lo + 64 <= j_hi) k_load(j_lo + 64, kreg2);
            k_store(0, kreg); v_store(0);
            if (j_lo + 64 <= j_hi) { k_store(1, kreg2); v_load(j_lo + 64); }
            if (j_lo + 128 <= j_hi) k_load(j_lo + 128, kreg);
        }
        __syncthreads();
        f32x4 sa[4], sb[4];
        if (frag_visible(q0, j_lo)) qk_frag(Ks[0], qf[0], qfl[0], sa);
        __syncthreads();
        // one step: tile at j0 in stage b (scores already in sc), next scores into sn
        auto pipe_step = [&](int j0, int b, f32x4 (&sc)[4], f32x4 (&sn)[4]) {
            if (j0 + 128 <= j_hi) k_store(b, kreg);          // K_{i+2} over K_i (consumed a step ago)
            if (j0 + 64 <= j_hi) v_store(b ^ 1);             // V_{i+1} over V_{i-1}
            if (j0 + 192 <= j_hi) k_load(j0 + 192, kreg);
            if (j0 + 128 <= j_hi) v_load(j0 + 128);
            // The barriers phase-lock a block's waves: left alone they all read K fragments, then all issue MFMAs, then all run
            // the softmax's VALU work -- LDS, MFMA and VALU time ADD UP (S = 512: ~2 300 + 1 000 + 3 500 cycles per step and
            // CU against the 6 500 measured).  The next tile's scores do not depend on this tile's softmax, so half of the
            // waves of every SIMD (waves w, w+4, w+8, w+12 share one) take the two pieces in the opposite order: while one
            // half reads LDS and feeds the MFMA pipe, the other half is in its softmax.
            const bool next_first = !SGPT_ATTN_PIPE_ALT || ((wave >> 2) & 1) == 0;
            if (next_first && frag_visible(q0, j0 + 64)) qk_frag(Ks[b ^ 1], qf[0], qfl[0], sn);
            if (frag_visible(q0, j0)) {
                uint32_t pw[8], pwl[8];
                softmax_frag(j0, q0, sc, m_run[0], l_run[0], o[0], pw, pwl);
                pv_frag(Vs[b], o[0], pw, pwl);
            }
            if (!next_first && frag_visible(q0, j0 + 64)) qk_frag(Ks[b ^ 1], qf[0], qfl[0], sn);
            __syncthreads();
        };
        for (int j0 = j_lo; j0 <= j_hi; j0 += 128) {
            pipe_step(j0, 0, sa, sb);
            if (j0 + 64 <= j_hi) pipe_step(j0 + 64, 1, sb, sa);
        }
    } else {
    // NB = 2: two LDS stages, ONE barrier per key tile -- tile j+1 is written into the other stage behind tile j's MFMAs,
    // and the barrier at the end of the step both publishes it and retires stage j.  Measured: no gain (the waves do not
    // wait for the barriers, profiles/r03_attn_pmc.txt), so NB = 1 is what ships: store, barrier, consume, barrier.
    int cur = 0;
    if (j_lo <= j_hi) {
        tile_load(j_lo);
        if constexpr (NB == 2) { tile_store(0); __syncthreads(); }
    }
    // The query fragments (global loads issued above) are first used by the MFMAs inside the loop.  Left to the compiler's
    // counter bookkeeping, the loop's back edge keeps them "pending" and every iteration waited s_waitcnt vmcnt(0) in front of
    // its K.Q^T MFMAs -- i.e. for the NEXT tile's prefetch it had just issued: a full global round trip per tile and wave, on
    // the dependent chain.  Retiring everything that is in flight once, here (the first tile is needed by the first
    // tile_store anyway), leaves only the prefetch pending inside the loop, and that is waited for where it is stored.
    __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0), expcnt / lgkmcnt untouched (gfx9 encoding)
    for (int j0 = j_lo; j0 <= j_hi; j0 += 64) {
        if constexpr (NB == 1) {
            __syncthreads();                         // previous tile fully consumed
            tile_store(0);
            __syncthreads();
        }
        const bool more = j0 + 64 <= j_hi;
        if (more) tile_load(j0 + 64);                // next tile's loads fly under this tile's MFMAs and softmax
        tile_compute(Ks[cur], Vs[cur], j0);
        if constexpr (NB == 2) {
            if (more) tile_store(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
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
        const int qf0 = qpos[f];
        if (qf0 < 0 || qf0 >= alloc) continue;
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
                if (qf0 + row < alloc) gstore16<ATTN_NT>(obase + (long)row * p.ldo + ch * 8, v);   // (rows past the allocation -- an even row count since round 4 -- are the next sequence's)
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
    // Zig-zag fragment pairs (head_dim 64, sequences of more than 128 rows; block b of NW waves owns pairs b NW .. b NW + NW - 1).
    // Attention launch alone at 131 072 token rows, same box per column (profiles/r04_attn_ab.txt), us:
    //     rows                            160    200    256    300    384    512
    //     consecutive fragments, 8 waves  267    288    217    294    235    (372: 16 waves)
    //     pairs, 16-wave blocks           344    322    256    286    259    271
    //     pairs, 8-wave blocks            207    213    169    306    271    255
    // -> 8-wave blocks of pairs wherever the last block is at least three quarters full (always up to 256 rows: one block, and
    //    two of them share a CU); else consecutive fragments if THEIR last block is that full (384 rows: 3 x 8); else one
    //    16-wave block of pairs up to 512 rows (300 rows: 10 of 16 waves have a pair); else the round-3 shapes below.
    const int zz_nf = (a.max_alloc_len + 15) / 16, zz_pairs = (zz_nf + 1) / 2;
    if (SGPT_ATTN_ZZ && a.dh == 64 && !a.out_fp8 && a.max_alloc_len > SGPT_ATTN_ZZ_MINLEN) {
        const bool fit8 = zz_pairs <= 8 || zz_pairs % 8 == 0 || zz_pairs % 8 >= 6;
        const bool consecutive_fit = zz_nf % 8 == 0 || zz_nf % 8 >= 6;
        const int zw = SGPT_ATTN_ZZ_NW ? SGPT_ATTN_ZZ_NW : (fit8 ? 8 : ((!consecutive_fit && zz_pairs <= 16) ? 16 : 0));
        if (zw != 0) {
            dim3 gz((zz_pairs + zw - 1) / zw, a.H, a.B);
            if (zw == 8) {
                if (a.dtype == DT_F16) hipLaunchKernelGGL((attn16_lds_kernel<f16_t, 64, false, 2, 8, 0, false, true>), gz, dim3(512), 0, s, a);
                else hipLaunchKernelGGL((attn16_lds_kernel<bf16_t, 64, false, 2, 8, 0, false, true>), gz, dim3(512), 0, s, a);
            } else {
                if (a.dtype == DT_F16) hipLaunchKernelGGL((attn16_lds_kernel<f16_t, 64, false, 2, 16, 0, false, true>), gz, dim3(1024), 0, s, a);
                else hipLaunchKernelGGL((attn16_lds_kernel<bf16_t, 64, false, 2, 16, 0, false, true>), gz, dim3(1024), 0, s, a);
            }
            return;
        }
    }
    constexpr bool PIPE = SGPT_ATTN_PIPE != 0;
    const bool pipe = PIPE && a.dh == 64 && !a.out_fp8 && a.max_alloc_len > SGPT_ATTN_PIPE_MINLEN;
    if (SGPT_ATTN_W16 && a.dh == 64 && !a.out_fp8 && a.max_alloc_len > 384 && (a.max_alloc_len - 1) % 256 >= 128) {
        dim3 g16((a.max_alloc_len + 255) / 256, a.H, a.B);
        if (a.dtype == DT_F16) hipLaunchKernelGGL((attn16_lds_kernel<f16_t, 64, false, 1, 16, 0, PIPE>), g16, dim3(1024), 0, s, a);
        else hipLaunchKernelGGL((attn16_lds_kernel<bf16_t, 64, false, 1, 16, 0, PIPE>), g16, dim3(1024), 0, s, a);
        return;
    }
    if (pipe) {
        if (a.dtype == DT_F16) hipLaunchKernelGGL((attn16_lds_kernel<f16_t, 64, false, 1, 8, 0, PIPE>), grid, dim3(512), 0, s, a);
        else hipLaunchKernelGGL((attn16_lds_kernel<bf16_t, 64, false, 1, 8, 0, PIPE>), grid, dim3(512), 0, s, a);
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
#define ATTN_CASE(H, O8)                                                                                          \
    if (a.dh == 64) hipLaunchKernelGGL((attn16_lds_kernel<H, 64, O8, 1>), grid, dim3(512), 0, s, a);          \
    else if (a.dh == 128) hipLaunchKernelGGL((attn16_lds_kernel<H, 128, O8, 1>), grid, dim3(512), 0, s, a);        \
    else if (a.dh == 256) hipLaunchKernelGGL((attn16_lds_kernel<H, 256, O8, 1>), grid, dim3(512), 0, s, a);        \
    else abort();
    if (a.out_fp8) { ATTN_CASE(bf16_t, true) }
    else if (a.dtype == DT_F16) { ATTN_CASE(f16_t, false) } else { ATTN_CASE(bf16_t, false) }
#undef ATTN_CASE
}

void launch_attn_f32(const AttnArgs& a, hipStream_t s) {
    dim3 grid((a.max_alloc_len + 3) / 4, a.H, a.B);
    hipLaunchKernelGGL(attn_f32_kernel, grid, dim3(256), 0, s, a);
}
