// gemm256w: the 256x256x64 persistent LDS-DMA GEMM of gemm.hip re-tiled for v_mfma_f32_32x32x16_{bf16,f16}.
//
// Why a second MFMA shape: the 16x16x32 instruction issues at ~17-20 cycles per 16 pipe cycles
// of work (guide: 2075 TFLOP/s micro-benchmark ceiling), the 32x32x16 one at 32.4 per 32 (2382-2495): the k-loop of
// gemm256d_kernel (20.3 cycles per 16x16x32 MFMA, 79 % pipe) sits AT the ceiling of its instruction.  Same operand
// bytes per FLOP from LDS (24 ds_read_b128 per wave and k-step either way), half the MFMA instructions.
//
// Geometry (identical to gemm256d_kernel unless noted)
//   * 8 waves as 2 (M) x 4 (N); a wave owns 128 x 64 = 4 x 2 blocks of 32 x 32, fp32 accumulators f32x16[4][2];
//   * LDS ring: streamed ("deep") operand 3 x 32-KiB slots fetched two k-steps ahead, resident ("shallow") operand
//     2 slots one step ahead, filled by global_load_lds_dwordx4 (1 KiB = 8 rows x 128 B per instruction);
//   * a k-step (64 deep) = 4 sub-steps of 16; fragment of sub-step s: lane (r = lane & 31, h = lane >> 5) reads the
//     16-byte chunk 2s + h of row r of its block -- both operands K-contiguous, so the A- and the B-fragment are the
//     same read and any k-permutation inside the instruction is common to both;
//   * swizzle: physical chunk = logical chunk ^ ((row >> 1) & 7).  ds_read_b128 is serviced in lane groups
//     {0-3,12-15,20-27} / {4-11,16-19,28-31} (+32): the 8 rows of one parity inside a group then hit 8 distinct
//     chunk slots and the two parities the two 128-byte halves of the 256-byte bank row: conflict-free
//     (scripts/lds_layout_check.py enumerates it).  The DMA writes LDS lane-linearly, so the same XOR is applied to the
//     per-lane SOURCE chunk: (lane & 7) ^ (lane >> 4) ^ 4 * (piece & 1);
//   * fragments are double-buffered by sub-step (next sub-step's 6 reads are issued before the current one's 8 MFMAs,
//     behind sched_barrier fences); the k-step's one barrier sits in front of its LAST sub-step, and sub-step 0 of the
//     next k-step is read behind it; DMA pieces ride behind every second MFMA of sub-steps 0 (shallow) and 1 (deep);
//     `s_waitcnt vmcnt(4)` at the barrier = everything but the newest four pieces (deep(s+2)) has landed.
//   * C layout of 32x32: lane (r, h) holds column r and rows 8q + 4h + e (register 4q + e).  With the weight fragment
//     as the A-operand (SWAP) that is ONE token row m and 4 consecutive n per register quad: the epilogues transpose
//     32 rows at a time through the wave's 8-KiB LDS scratch and store whole 128 / 256-byte rows (dwordx4).
#include <cstdlib>

#include "common.h"

namespace {

constexpr int CH = 8;                 // 16-byte chunks per row per k-step
constexpr int TM = 256, TN = 256;
constexpr int SLOT = TM * CH;         // uint4 per 32-KiB slot

template <typename OutT> struct OutRange { typedef RangeTrack<bf16_t> type; };
template <> struct OutRange<f16_t> { typedef RangeTrack<f16_t> type; };

template <typename T, int EPI, typename OutT, bool SWAP, bool DEEP_A>
__global__ __launch_bounds__(512, 2) void gemm256w_kernel(const GemmArgs p) {
    if (p.pred != nullptr && *p.pred == 0) return;
    typedef __attribute__((address_space(3))) char* lds_cptr_t;
    __shared__ __attribute__((aligned(16))) uint4 lds[5 * SLOT];    // [deep 0..2 | shallow 0..1] = 160 KiB

    const int N = p.N, K = p.K;
    const int MT = p.M / TM, NT = N / TN;
    const int GM = p.gm > 0 ? p.gm : 4, GN = p.gn > 0 ? p.gn : 8;
    const bool m_major = MT >= NT;
    const int AT = m_major ? MT : NT, BT = m_major ? NT : MT;
    const int per_band = GM * BT;
    const int tiles_total = ((AT + 7) / 8 + GM - 1) / GM * GM * 8 * BT;
    auto tile_coords = [&](int tile, int& m0, int& n0) -> bool {
        const int xcd = tile & 7, local = tile >> 3;
        const int band = local / per_band, inb = local % per_band;
        const int ng = inb / (GM * GN);
        const int gn = (BT - ng * GN) < GN ? (BT - ng * GN) : GN;
        const int r = inb - ng * GM * GN;
        const int at = xcd + 8 * (band * GM + r / gn), bt = ng * GN + r % gn;
        m0 = (m_major ? at : bt) * TM; n0 = (m_major ? bt : at) * TN;
        return at < AT;
    };

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int r32 = lane & 31, h = lane >> 5;

    const bf16_t* __restrict__ Ag = static_cast<const bf16_t*>(p.A);
    const bf16_t* __restrict__ Wg = static_cast<const bf16_t*>(p.W);
    const unsigned lds_base = (unsigned)(size_t)(lds_cptr_t)(&lds[0]);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // LDS-DMA source: lane l of piece q fills LDS slot (row 8q + (l >> 3), chunk l & 7) from global chunk
    // (l & 7) ^ swz(row), swz(row) = (row >> 1) & 7 = ((l >> 4) | 4 (q & 1)): two per-lane byte offsets per operand
    const int lc0 = (lane & 7) ^ (lane >> 4), lc1 = lc0 ^ 4;
    const unsigned a_loff0 = (unsigned)(((lane >> 3) * p.lda + lc0 * 8) * 2), a_loff1 = (unsigned)(((lane >> 3) * p.lda + lc1 * 8) * 2);
    const unsigned w_loff0 = (unsigned)(((lane >> 3) * p.ldw + lc0 * 8) * 2), w_loff1 = (unsigned)(((lane >> 3) * p.ldw + lc1 * 8) * 2);
    // wave-uniform 64-bit base in SGPRs + per-lane 32-bit offset, M0 = LDS destination.  M0 is not declared clobbered
    // (hipcc rejects reserved registers on clobber lists with a warning); nothing the compiler emits for this kernel
    // uses M0 -- tests/test_host_logic.py::test_m0_only_written_by_the_dma_idiom checks the generated ISA.
    auto dma16 = [&](const char* base_uniform, unsigned lane_off, unsigned dst_byte) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :
                     : "v"(lane_off), "s"(base_uniform), "s"(dst_byte)
                     : "memory");
    };
    auto deep_off = [&](int sd) { return (unsigned)(sd * SLOT * 16); };
    auto shal_off = [&](int ss) { return (unsigned)((3 + ss) * SLOT * 16); };
    auto piece = [&](const bf16_t* src, long ld, unsigned loff0, unsigned loff1, int kt, unsigned slot_off, int q) {
        const unsigned row_off = (unsigned)((wave_u * 32 + q * 8) * CH * 16);
        dma16(reinterpret_cast<const char*>(src + (long)(wave_u * 32 + q * 8) * ld + kt * 64), (q & 1) ? loff1 : loff0,
              lds_base + slot_off + row_off);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = K / 64;   // >= 2 (launcher)
    OutT* __restrict__ out = static_cast<OutT*>(p.out);

    int tile = blockIdx.x, m0 = 0, n0 = 0;
    while (tile < tiles_total && !tile_coords(tile, m0, n0)) tile += gridDim.x;
    if (tile >= tiles_total) return;
    const bf16_t* asrc = Ag + (long)m0 * p.lda;          // wave-uniform tile bases
    const bf16_t* wsrc = Wg + (long)n0 * p.ldw;
    const long dld = DEEP_A ? p.lda : p.ldw, sld = DEEP_A ? p.ldw : p.lda;
    const unsigned dloff0 = DEEP_A ? a_loff0 : w_loff0, dloff1 = DEEP_A ? a_loff1 : w_loff1;
    const unsigned sloff0 = DEEP_A ? w_loff0 : a_loff0, sloff1 = DEEP_A ? w_loff1 : a_loff1;
    int sd = 0, ss = 0;      // ring slots of the k-step about to be computed
    typename OutRange<OutT>::type range;

    {   // prologue: deep(0), shallow(0), deep(1) -- in the order the waits assume
        const bf16_t* dsrc = DEEP_A ? asrc : wsrc;
        const bf16_t* ssrc = DEEP_A ? wsrc : asrc;
#pragma unroll
        for (int q = 0; q < 4; ++q) piece(dsrc, dld, dloff0, dloff1, 0, deep_off(0), q);
#pragma unroll
        for (int q = 0; q < 4; ++q) piece(ssrc, sld, sloff0, sloff1, 0, shal_off(0), q);
#pragma unroll
        for (int q = 0; q < 4; ++q) piece(dsrc, dld, dloff0, dloff1, 1, deep_off(1), q);
    }
    // fragment addressing (uint4 index inside a slot): row * 8 + ((2s + h) ^ swz), swz = (r32 >> 1) & 7 for every block
    // of this lane (block row offsets are multiples of 16)
    const int swz = (r32 >> 1) & 7;
    const int a_row = (wm * 128 + r32) * CH, w_row = (wn * 64 + r32) * CH;
    int cs[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) cs[s] = (2 * s + h) ^ swz;
    uint4 af[2][4], wf[2][2];
    auto ld_frags = [&](const uint4* la, const uint4* lw, int s, int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) wf[buf][j] = lw[w_row + j * 32 * CH + cs[s]];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[buf][i] = la[a_row + i * 32 * CH + cs[s]];
    };
    {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // deep(0), shallow(0) landed
        __syncthreads();
        const uint4* la = lds + (DEEP_A ? sd * SLOT : (3 + ss) * SLOT);
        const uint4* lw = lds + (DEEP_A ? (3 + ss) * SLOT : sd * SLOT);
        ld_frags(la, lw, 0, 0);
    }
    while (true) {
        int ntile = tile + gridDim.x, nm0 = 0, nn0 = 0;
        while (ntile < tiles_total && !tile_coords(ntile, nm0, nn0)) ntile += gridDim.x;
        const bool has_next = ntile < tiles_total;
        const bf16_t* nasrc = has_next ? Ag + (long)nm0 * p.lda : asrc;   // past the end: harmless re-fetch
        const bf16_t* nwsrc = has_next ? Wg + (long)nn0 * p.ldw : wsrc;
        const bf16_t* d_cur = DEEP_A ? asrc : wsrc, *d_nxt = DEEP_A ? nasrc : nwsrc;
        const bf16_t* s_cur = DEEP_A ? wsrc : asrc, *s_nxt = DEEP_A ? nwsrc : nasrc;
        for (int kt = 0; kt < nk; ++kt) {
            if (p.dbg && blockIdx.x == 0 && t == 0 && kt < 12) p.dbg[64 + kt] = (long long)__builtin_amdgcn_s_memtime();
            const bool s_in = kt + 1 < nk, d_in = kt + 2 < nk;
            const bf16_t* sp = s_in ? s_cur : s_nxt;  const int skt = s_in ? kt + 1 : 0;
            const bf16_t* dp = d_in ? d_cur : d_nxt;  const int dkt = d_in ? kt + 2 : kt + 2 - nk;
            const unsigned s_dst = shal_off(ss ^ 1);
            const int sd2 = sd + 2 >= 3 ? sd - 1 : sd + 2;
            const unsigned d_dst = deep_off(sd2);
            const int sdn = sd + 1 >= 3 ? 0 : sd + 1;
            const uint4* la = lds + (DEEP_A ? sd * SLOT : (3 + ss) * SLOT);
            const uint4* lw = lds + (DEEP_A ? (3 + ss) * SLOT : sd * SLOT);
            const uint4* nla = lds + (DEEP_A ? sdn * SLOT : (3 + (ss ^ 1)) * SLOT);
            const uint4* nlw = lds + (DEEP_A ? (3 + (ss ^ 1)) * SLOT : sdn * SLOT);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s < 3) {
                    ld_frags(la, lw, s + 1, (s + 1) & 1);
                } else {                        // every read of this stage has returned; stage kt+1 has landed
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(4)" ::: "memory");
                    __syncthreads();
                    ld_frags(nla, nlw, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if constexpr (SWAP) acc[i][j] = Half<T>::mfma32(wf[s & 1][j], af[s & 1][i], acc[i][j]);
                        else acc[i][j] = Half<T>::mfma32(af[s & 1][i], wf[s & 1][j], acc[i][j]);
                    }
                    // one 1-KiB DMA piece behind every second MFMA: shallow x4 in sub-step 0, deep x4 in sub-step 1
                    if (s == 0) piece(sp, sld, sloff0, sloff1, skt, s_dst, i);
                    if (s == 1) piece(dp, dld, dloff0, dloff1, dkt, d_dst, i);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            ss ^= 1;
            sd = sdn;
        }
        if (p.dbg && blockIdx.x == 0 && t == 0) p.dbg[0] = (long long)__builtin_amdgcn_s_memtime();
        // ------------------------------------------ epilogue ------------------------------------------
        {
            const int sdf = sd + 2 >= 3 ? sd - 1 : sd + 2;     // the deep / shallow slots of the stage just consumed are free
            char* scr = reinterpret_cast<char*>(lds) + (wave < 4 ? deep_off(sdf) : shal_off(ss ^ 1)) + (wave & 3) * 8192;
            if constexpr (EPI == EPI_SCORE_FILTER) {
                // Scorer chunks after the first: only scores STRICTLY above the query's running k-th best can enter its
                // top-k; they are appended to the query's candidate list, nothing else leaves the registers.
                // Lane (r32, h): query row m of block i, document columns 32j + 8q + 4h + e.
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = m0 + wm * 128 + i * 32 + r32;
                    const float th = m < p.m_valid ? p.thr[(long)m * p.thr_ld] : INFINITY;
                    float mx = -INFINITY;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            float v = acc[i][j][e];
                            v = v != v ? -1.0f : v;                      // cos_scores[isnan] = -1 (exact_search.py:99)
                            acc[i][j][e] = v;
                            mx = fmaxf(mx, v);
                        }
                    if (mx > th && cand_room(p.cand_cnt + m, p.cand_cap)) {   // (over capacity: recomputed anyway)
                        int c = 0;
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int e = 0; e < 16; ++e) c += acc[i][j][e] > th ? 1 : 0;
                        int slot = atomicAdd(p.cand_cnt + m, c);
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                const float v = acc[i][j][e];
                                if (v > th) {
                                    if (slot < p.cand_cap) {
                                        p.cand_val[(long)m * p.cand_cap + slot] = v;
                                        p.cand_idx[(long)m * p.cand_cap + slot] =
                                            p.idx_base + n0 + wn * 64 + j * 32 + 8 * (e >> 2) + 4 * h + (e & 3);
                                    }
                                    ++slot;
                                }
                            }
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
                }
                (void)scr;
            } else if constexpr (EPI == EPI_NONE) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        asm volatile("" ::"v"(acc[i][j]));
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
                    }
                (void)scr;
            } else if constexpr (SWAP && sizeof(OutT) == 2) {
                // 16-bit row-major: 32 rows x 128 B per round, LDS row stride 144 B
                constexpr int RS = 144;
                float4 bb[2][4];
                const bool has_bias = EPI == EPI_BIAS_GELU || p.bias != nullptr;   // plain store + bias: BLOOM Q/K projection
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        bb[j][q] = has_bias ? *reinterpret_cast<const float4*>(p.bias + n0 + wn * 64 + j * 32 + 8 * q + 4 * h)
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
                const int rrow = lane >> 3, rchunk = lane & 7;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float v0 = acc[i][j][4 * q] + bb[j][q].x, v1 = acc[i][j][4 * q + 1] + bb[j][q].y;
                            float v2 = acc[i][j][4 * q + 2] + bb[j][q].z, v3 = acc[i][j][4 * q + 3] + bb[j][q].w;
                            if constexpr (EPI == EPI_BIAS_GELU) {
                                v0 = gelu_new_fast(v0); v1 = gelu_new_fast(v1); v2 = gelu_new_fast(v2); v3 = gelu_new_fast(v3);
                            }
                            range.note(v0, v1); range.note(v2, v3);
                            *reinterpret_cast<uint2_a*>(scr + r32 * RS + (j * 32 + 8 * q + 4 * h) * 2) =
                                make_uint2(Half<OutT>::pack2(v0, v1), Half<OutT>::pack2(v2, v3));
                            acc[i][j][4 * q] = 0.f; acc[i][j][4 * q + 1] = 0.f; acc[i][j][4 * q + 2] = 0.f; acc[i][j][4 * q + 3] = 0.f;
                        }
#pragma unroll
                    for (int k8 = 0; k8 < 4; ++k8) {
                        const int row = k8 * 8 + rrow;
                        const uint4 v = *reinterpret_cast<const uint4_a*>(scr + row * RS + rchunk * 16);
                        const int m = m0 + wm * 128 + i * 32 + row;
                        gstore16<true>(reinterpret_cast<bf16_t*>(out) + (long)m * p.ldo + n0 + wn * 64 + rchunk * 8, v);
                    }
                }
            } else if constexpr (SWAP) {
                // fp32 row-major (+bias +residual | score): 32 rows x 256 B per round; the scratch rows are 256 B, unpadded,
                // 16-byte chunk c of row r stored at chunk c ^ (r & 15)
                const int rrow = lane >> 4, rchunk = lane & 15;
                float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (EPI == EPI_BIAS_RESID) bb = *reinterpret_cast<const float4*>(p.bias + n0 + wn * 64 + rchunk * 4);
                const long gbase = (long)(m0 + wm * 128 + rrow) * p.ldo + n0 + wn * 64 + rchunk * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // scratch <- the block row's accumulators
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int c = j * 8 + 2 * q + h;
                            *reinterpret_cast<float4_a*>(scr + r32 * 256 + ((c ^ (r32 & 15)) << 4)) =
                                make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                            acc[i][j][4 * q] = 0.f; acc[i][j][4 * q + 1] = 0.f; acc[i][j][4 * q + 2] = 0.f; acc[i][j][4 * q + 3] = 0.f;
                        }
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        float4 rr[4];
                        if constexpr (EPI == EPI_BIAS_RESID) {   // 16 residual rows in flight per half round (32 would spill)
#pragma unroll
                            for (int k4 = 0; k4 < 4; ++k4)
                                rr[k4] = *reinterpret_cast<const float4*>(p.resid + gbase + (long)(i * 32 + (half * 4 + k4) * 4) * p.ldo);
                        }
#pragma unroll
                        for (int k4 = 0; k4 < 4; ++k4) {
                            const int row = (half * 4 + k4) * 4 + rrow;
                            float4 v = *reinterpret_cast<const float4_a*>(scr + row * 256 + ((rchunk ^ (row & 15)) << 4));
                            if constexpr (EPI == EPI_BIAS_RESID) {
                                // acc + (bias + resid): the association of every GEMM kernel in this library, bit for bit
                                v.x += bb.x + rr[k4].x; v.y += bb.y + rr[k4].y; v.z += bb.z + rr[k4].z; v.w += bb.w + rr[k4].w;
                            }
                            if constexpr (EPI == EPI_SCORE) {   // cos_scores[isnan] = -1 (exact_search.py:99); padded query rows skipped
                                v.x = v.x != v.x ? -1.0f : v.x; v.y = v.y != v.y ? -1.0f : v.y;
                                v.z = v.z != v.z ? -1.0f : v.z; v.w = v.w != v.w ? -1.0f : v.w;
                                if (m0 + wm * 128 + i * 32 + row >= p.m_valid) continue;
                            }
                            gstore16<false>(reinterpret_cast<float*>(out) + gbase + (long)(i * 32 + (half * 4 + k4) * 4) * p.ldo,
                                            __builtin_bit_cast(uint4, v));
                        }
                    }
                }
            } else {
                // V^T (16-bit, out[n][m]): the activation fragment was the A-operand, so lane (r32, h) holds output row
                // n = 32j + r32 and 4 consecutive m = 32i + 8q + 4h + e per register quad.  32 n-rows x 256 B (128 m)
                // per round, scratch rows 256 B with the chunk ^ (row & 15) swizzle
                const int rrow = lane >> 4, rchunk = lane & 15;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float bn = p.bias ? p.bias[n0 + wn * 64 + j * 32 + r32] : 0.f;   // BLOOM: V projection bias
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float v0 = acc[i][j][4 * q] + bn, v1 = acc[i][j][4 * q + 1] + bn;
                            const float v2 = acc[i][j][4 * q + 2] + bn, v3 = acc[i][j][4 * q + 3] + bn;
                            range.note(v0, v1); range.note(v2, v3);
                            const int c = i * 4 + q;
                            *reinterpret_cast<uint2_a*>(scr + r32 * 256 + ((c ^ (r32 & 15)) << 4) + 8 * h) =
                                make_uint2(Half<OutT>::pack2(v0, v1), Half<OutT>::pack2(v2, v3));
                            acc[i][j][4 * q] = 0.f; acc[i][j][4 * q + 1] = 0.f; acc[i][j][4 * q + 2] = 0.f; acc[i][j][4 * q + 3] = 0.f;
                        }
#pragma unroll
                    for (int k4 = 0; k4 < 8; ++k4) {
                        const int row = k4 * 4 + rrow;
                        const uint4 v = *reinterpret_cast<const uint4_a*>(scr + row * 256 + ((rchunk ^ (row & 15)) << 4));
                        const int n = n0 + wn * 64 + j * 32 + row;
                        gstore16<true>(reinterpret_cast<bf16_t*>(out) + (long)n * p.ldo + m0 + wm * 128 + rchunk * 8, v);
                    }
                }
            }
        }
        if (p.dbg && blockIdx.x == 0 && t == 0) p.dbg[1] = (long long)__builtin_amdgcn_s_memtime();
        __syncthreads();       // every wave is done with its scratch before the next tile's DMA re-uses those slots
        if (!has_next) break;
        tile = ntile; m0 = nm0; n0 = nn0; asrc = nasrc; wsrc = nwsrc;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the run-ahead DMA before the LDS is released
    range.finish(p.range_flag);
}

template <typename T, int EPI, typename OutT, bool SWAP>
void launch256w(const GemmArgs& a, hipStream_t s, bool deep_a) {
    const int MT = a.M / 256, NT = a.N / 256;
    const int AT = MT >= NT ? MT : NT, BT = MT >= NT ? NT : MT;
    static const int ncu = [] {
        int dev = 0, n = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n / 8 * 8;
    }();
    GemmArgs b = a;
    static const int env_gm = getenv("SGPT_GM") ? atoi(getenv("SGPT_GM")) : 0, env_gn = getenv("SGPT_GN") ? atoi(getenv("SGPT_GN")) : 0;
    if (env_gm > 0) b.gm = env_gm;
    if (env_gn > 0) b.gn = env_gn;
    const int gm = b.gm > 0 ? b.gm : 4;
    const int tiles_pad = ((AT + 7) / 8 + gm - 1) / gm * gm * 8 * BT;
    const int grid = tiles_pad < ncu ? tiles_pad : ncu;
    if (deep_a) hipLaunchKernelGGL((gemm256w_kernel<T, EPI, OutT, SWAP, true>), dim3(grid), dim3(512), 0, s, b);
    else hipLaunchKernelGGL((gemm256w_kernel<T, EPI, OutT, SWAP, false>), dim3(grid), dim3(512), 0, s, b);
}

template <typename H>
void dispatch(int epi, const GemmArgs& a, hipStream_t s, bool deep_a) {
    if (epi == EPI_SCORE) return launch256w<H, EPI_SCORE, float, true>(a, s, deep_a);
    if (epi == EPI_SCORE_FILTER) return launch256w<H, EPI_SCORE_FILTER, float, true>(a, s, deep_a);
    if (epi == EPI_STORE) return launch256w<H, EPI_STORE, H, true>(a, s, deep_a);
    if (epi == EPI_VT) return launch256w<H, EPI_VT, H, false>(a, s, deep_a);
    if (epi == EPI_BIAS_GELU) return launch256w<H, EPI_BIAS_GELU, H, true>(a, s, deep_a);
    if (epi == EPI_BIAS_RESID) return launch256w<H, EPI_BIAS_RESID, float, true>(a, s, deep_a);
    if (epi == EPI_NONE) return launch256w<H, EPI_NONE, H, true>(a, s, deep_a);
    abort();
}

}  // namespace

// 16-bit operands, M % 256 == 0, N % 256 == 0, K % 64 == 0, K >= 128; EPI_STORE means a 16-bit output here
void launch_gemm256w(int dtype, int epi, const GemmArgs& a, hipStream_t s, bool deep_a) {
    if (dtype == DT_F16) dispatch<f16_t>(epi, a, s, deep_a);
    else dispatch<bf16_t>(epi, a, s, deep_a);
}
