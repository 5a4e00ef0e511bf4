// C[M,N] = A[M,K] * W[N,K]^T with fused epilogues -- the MFMA hot loop of the encoder
// (QKV / out-proj / MLP projections: HF:gpt_neo/modeling_gpt_neo.py:141-143,155,304-309)
// and of the scorer (torch.mm(a, b.T): sentence_transformers/util.py:43,63).
//
// gfx950 design
//   * 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave =
//     4x4 MFMA 16x16 fragments, fp32 accumulators in 64 VGPRs);
//   * both operands are K-contiguous ("B^T input"), so A- and B-fragments are the same
//     16-byte-per-lane read: lane (r = lane&15, g = lane>>4) holds 16 B of row r at
//     K-chunk g.  One k-step = 128 B of K per row (64 bf16 / 32 fp32);
//       bf16: one v_mfma_f32_16x16x32_bf16 per fragment pair per 32-wide k slice,
//       fp32: four v_mfma_f32_16x16x4_f32 (element e of the 16-B chunk <-> k = 4*chunk+e,
//             the same k permutation on both operands, so the contraction is exact);
//   * LDS: 2 (double buffer) x 2 (A,W) x 128 rows x 128 B = 64 KiB, rows XOR-swizzled at
//     16-B granularity (chunk ^= row&7) -> conflict-free ds_write_b128 / ds_read_b128;
//   * register-staged software pipeline: global loads of tile k+1 are issued before the
//     MFMAs of tile k and written to the other LDS buffer after them (one barrier / k-step);
//   * XCD-aware block order: the 8 XCDs own interleaved M-tiles, and the N-tiles of one
//     M-tile run back-to-back on the same XCD, so the activation panel is fetched into one
//     L2 only (weights are small and live in every L2);
//   * SWAP=true feeds the weight fragment as the MFMA A-operand, so a lane ends up with
//     4 consecutive n for one m: row-major stores are 8 B (bf16) / 16 B (fp32) per lane.
//     SWAP=false gives 4 consecutive m for one n: used for the transposed V^T store.
#include <cstdlib>

#include "common.h"

namespace {

#ifndef SGPT_RESID_NT
#define SGPT_RESID_NT 0
#endif
constexpr bool RESID_NT = SGPT_RESID_NT != 0;
#ifndef SGPT_RESID_LD_NT
#define SGPT_RESID_LD_NT 0
#endif
constexpr bool RESID_LD_NT = SGPT_RESID_LD_NT != 0;
#ifndef SGPT_FOLD_TAIL_SCORE
#define SGPT_FOLD_TAIL_SCORE 1   // the same for the materialised-score launch (api.hip has the same switch)
#endif
#ifndef SGPT_FOLD_TAIL
#define SGPT_FOLD_TAIL 1     // 0: A/B builds without the ragged-tail handling of the filtered scorer launch (api.hip has the same switch)
#endif
#ifndef SGPT_THV_UNCOND
#define SGPT_THV_UNCOND 1
#endif
#ifndef SGPT_SMALL_PF
#define SGPT_SMALL_PF 2
#endif
#ifndef SGPT_FEW_TILES
#define SGPT_FEW_TILES 128   // 256x256 tiles from more than half a wave of them on
#endif
#ifndef SGPT_DEEP_TILES
#define SGPT_DEEP_TILES 512
#endif
#ifndef SGPT_KG16_MIN
#define SGPT_KG16_MIN 6
#endif
constexpr int CH = 8;  // 16-byte chunks per row per k-step

// A/B switches of the measurement scripts exist only in the experiment build (`SGPT_EXPERIMENTS=1 python -m sgpt_amd.build`
// -> libsgpt_hip_exp.so, loaded through SGPT_HIP_LIB): the shipped library reads no environment variable and holds no
// process-global mutable state -- the two run-time policies it has (tile policy, k-groups) are per-ctx and arrive in GemmArgs.
#ifdef SGPT_EXPERIMENTS
inline const char* exp_env(const char* name) { return getenv(name); }
int g_skew = getenv("SGPT_SKEW") ? atoi(getenv("SGPT_SKEW")) : 0;     // start-up stagger of gemm256d_kernel, shader cycles per phase
int g_use_w = getenv("SGPT_GEMM_W") ? (atoi(getenv("SGPT_GEMM_W")) & 1) : 0;   // 1: the 32x32x16-MFMA re-tiling (gemm256w.hip)
#else
inline const char* exp_env(const char*) { return nullptr; }
#endif

template <typename T> struct ElemTraits;
template <> struct ElemTraits<bf16_t> { static constexpr int EPC = 8; };
template <> struct ElemTraits<f16_t> { static constexpr int EPC = 8; };
template <> struct ElemTraits<float> { static constexpr int EPC = 4; };

template <typename T, bool SWAP>
__device__ __forceinline__ void mma(f32x4& acc, const uint4& act, const uint4& wgt) {
    if constexpr (sizeof(T) == 2) {
        if constexpr (SWAP) acc = Half<T>::mfma16(wgt, act, acc);
        else acc = Half<T>::mfma16(act, wgt, acc);
    } else {
        const float a[4] = {__uint_as_float(act.x), __uint_as_float(act.y), __uint_as_float(act.z),
                            __uint_as_float(act.w)};
        const float w[4] = {__uint_as_float(wgt.x), __uint_as_float(wgt.y), __uint_as_float(wgt.z),
                            __uint_as_float(wgt.w)};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (SWAP) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[e], a[e], acc, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], w[e], acc, 0, 0, 0);
        }
    }
}

template <typename OutT> __device__ __forceinline__ void store4(OutT* p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void store4<float>(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
}
template <> __device__ __forceinline__ void store4<f16_t>(f16_t* p, float a, float b, float c, float d) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_f16x2(a, b), pack_f16x2(c, d));
}
template <typename OutT> __device__ __forceinline__ void store1(OutT* p, float a);
template <> __device__ __forceinline__ void store1<float>(float* p, float a) { *p = a; }
template <> __device__ __forceinline__ void store1<bf16_t>(bf16_t* p, float a) { *p = f32_to_bf16(a); }
template <> __device__ __forceinline__ void store1<f16_t>(f16_t* p, float a) { p->bits = f32_to_h<f16_t>(a); }
// range tracker of the output format (a no-op unless OutT = f16_t)
template <typename OutT> struct OutRange { typedef RangeTrack<bf16_t> type; };
template <> struct OutRange<f16_t> { typedef RangeTrack<f16_t> type; };

// NI = 16-row MFMA fragments per wave and dimension: 4 -> 128x128 tile (64x64 per wave), 2 -> 64x64 tile (32x32 per wave)
// for problems too small to fill the chip with larger tiles.  Every GEMM kernel in this file feeds each output element the
// same sequence of v_mfma_f32_16x16x32_bf16 operations (k ascending, same k-slot mapping) and the same epilogue
// arithmetic, so the choice of kernel / tile never changes a bit of the result (batch invariance of the embeddings) --
// with the one opt-in exception of the k-groups below.
//
// KG = k-groups per workgroup (split-K inside the workgroup).  A query-sized launch has fewer tiles than the chip has
// CUs and a long serial k-loop per tile (16 queries, fc2: 96 tiles x 48 k-steps, each step a load -> LDS -> MFMA
// round trip of ~0.7 us that nothing overlaps).  With KG > 1 the workgroup carries KG groups of 4 waves, each with its
// own double-buffered LDS stage, walking its own contiguous 1/KG of the k-steps concurrently; at the end groups 1..KG-1
// park their fp32 accumulators in LDS and group 0 adds them in group order and runs the epilogue.  Deterministic (fixed
// order, no atomics, nothing leaves the CU); the sum differs from the k-ascending one only by fp32 rounding.
// (A split over several workgroups per tile with a ticket counter was tried first: the device-scope fences it needs
// write back the XCD's L2 once per workgroup and cost 58 us per launch -- 4x slower than not splitting.)
template <typename T, int EPI, typename OutT, bool SWAP, int NI, int KG, int CHS = CH>
__global__ __launch_bounds__(256 * KG, KG == 1 ? 2 : 1) void gemm_kernel(const GemmArgs p) {
    // CHS = 16-byte chunks per LDS row = k-step depth (8: 64 16-bit elements; 16: 128, the query-sized launches' setting)
    constexpr int BM = 32 * NI, BN = 32 * NI;
    constexpr int NP = NI * CHS / 8;          // load passes: 256 threads cover 256 / CHS rows x CHS chunks per pass
    constexpr int RPP = 256 / CHS;
    if (p.pred != nullptr && *p.pred == 0) return;   // predicated (fallback) launch that is not needed
    constexpr int EPC = ElemTraits<T>::EPC;
    constexpr int BK = CHS * EPC;
    static_assert(KG == 1 || NI == 2, "k-groups are for the 64x64 tile (32 KiB of LDS per group)");
    __shared__ __attribute__((aligned(16))) uint4 lds_all[KG][2][2][BM * CHS];  // 64 KiB (128x128) / 32 or 64 KiB per group (64x64)
    const int kg = KG == 1 ? 0 : (int)(threadIdx.x >> 8);
    auto& lds = lds_all[kg];

    const int M = p.M, N = p.N, K = p.K;
    const int MT = (M + BM - 1) / BM, NT = (N + BN - 1) / BN;
    // XCD-aware block order (block b runs on XCD b%8, ~64 blocks resident per XCD).
    // More than one resident round (MT NT > 512): XCD x owns M-tiles x, x+8, ...; its blocks walk GM x GN supertiles so the ~64
    // co-resident blocks share GM activation panels and GN weight panels (<= 16 x 192 KiB, fits the 4 MiB L2).
    // Query-sized launches (at most 512 tiles, everything fits the L2s): XCD x takes a contiguous run of the M-major tile list,
    // runs equal to within one tile -- the supertile order hands XCD x NT x ceil or floor(MT / 8) tiles (8 x 12 tiles: 12 per
    // XCD either way, but 44 x 12: 72 against 60), and a launch that fits one round must not spill into a second on half of
    // the chip.  Round 4, same box: 16 queries 0.793 -> 0.764 ms; with the runs also on larger launches 300 queries 2.61 -> 2.70 ms
    // (a run of 64 consecutive tiles of a 48-tile-wide launch streams the whole weight matrix through the L2), hence the limit.
    const int b = blockIdx.x;
    const int xcd = b & 7, local = b >> 3;
    int mt, nt;
    if (MT * NT <= 512) {
        const int R = MT * NT, c0 = R >> 3, rem = R & 7;
        if (local >= c0 + (xcd < rem ? 1 : 0)) return;
        const int gi = xcd * c0 + (xcd < rem ? xcd : rem) + local;
        mt = gi / NT; nt = gi - mt * NT;
    } else {
        constexpr int GM = 8, GN = 8;
        const int per_band = GM * NT;              // blocks per band of GM M-tiles
        const int band = local / per_band, inb = local % per_band;
        const int ng = inb / (GM * GN);            // N-group inside the band
        const int gn = (NT - ng * GN) < GN ? (NT - ng * GN) : GN;   // width of this N-group
        const int r = inb - ng * GM * GN;
        const int mi = r / gn, ni = r % gn;
        mt = xcd + 8 * (band * GM + mi); nt = ng * GN + ni;
        if (mt >= MT) return;
    }
    const int m0 = mt * BM, n0 = nt * BN;

    const int t = threadIdx.x & 255;
    const int lr = t / CHS, lc = t % CHS;
    const T* __restrict__ Ag = static_cast<const T*>(p.A);
    const T* __restrict__ Wg = static_cast<const T*>(p.W);
    // group kg covers k-steps [kb, kb + nk); the launcher picks KG > 1 only when the steps divide evenly, so every group
    // runs the same number of barriers
    const int nk = (K + BK - 1) / BK / KG;
    const int kb = kg * nk;

    long arow[NP], wrow[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        int ra = m0 + lr + RPP * i; ra = ra < M ? ra : M - 1;
        int rw = n0 + lr + RPP * i; rw = rw < N ? rw : N - 1;
        arow[i] = (long)ra * p.lda;
        wrow[i] = (long)rw * p.ldw;
    }

    // PF k-steps of global loads in flight per thread (register sets used round-robin).  The small-problem launches this
    // kernel serves run <= 1 workgroup per CU, so nothing else hides the ~1.3 us global-load round trip: with one step
    // in flight every k-step cost exactly that (fc2 at 512 tokens: 48 steps = 63 us for 2.4 GFLOP); with PF in flight
    // the steady state is latency / PF.  Order of the MFMA operations per output element is unchanged (k ascending).
    constexpr int PF = SGPT_SMALL_PF;
    uint4 ra_[PF][NP], rw_[PF][NP];
    auto gload = [&](int set, int kt) {
        kt += kb;
        const int kc = kt * BK + lc * EPC;
        if (kt * BK + BK <= K) {  // block-uniform: full k-step (every encoder GEMM)
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                ra_[set][i] = *reinterpret_cast<const uint4*>(Ag + arow[i] + kc);
                rw_[set][i] = *reinterpret_cast<const uint4*>(Wg + wrow[i] + kc);
            }
        } else {  // K tail (scoring with d not a multiple of the k-step): zero-fill chunks past K
            const bool ok = kc < K;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                ra_[set][i] = ok ? *reinterpret_cast<const uint4*>(Ag + arow[i] + kc) : make_uint4(0, 0, 0, 0);
                rw_[set][i] = ok ? *reinterpret_cast<const uint4*>(Wg + wrow[i] + kc) : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto gload_full = [&](int set, int kt) {
        const int kc = (kb + kt) * BK + lc * EPC;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            ra_[set][i] = *reinterpret_cast<const uint4*>(Ag + arow[i] + kc);
            rw_[set][i] = *reinterpret_cast<const uint4*>(Wg + wrow[i] + kc);
        }
    };
    auto lstore = [&](int set, int buf) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int row = lr + RPP * i;
            const int off = row * CHS + (lc ^ (row & (CHS - 1)));
            lds[buf][0][off] = ra_[set][i];
            lds[buf][1][off] = rw_[set][i];
        }
    };

    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, g = lane >> 4;

    f32x4 acc[NI][NI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < CHS / 4; ++ks) {
            uint4 af[NI], wf[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int row = wm * (16 * NI) + i * 16 + fr;
                af[i] = lds[buf][0][row * CHS + ((4 * ks + g) ^ (row & (CHS - 1)))];
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int row = wn * (16 * NI) + j * 16 + fr;
                wf[j] = lds[buf][1][row * CHS + ((4 * ks + g) ^ (row & (CHS - 1)))];
            }
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) mma<T, SWAP>(acc[i][j], af[i], wf[j]);
        }
    };

    // Query-sized launches (64x64 tiles) are a serial chain of a few microseconds.  The epilogue's bias / residual operands do
    // not depend on the product, so they are fetched HERE, under the k-loop, instead of one dependent L2 round trip per 16x16
    // piece at the end (the residual is updated in place: every piece's load also waited for the previous piece's store).
    // Same values, same arithmetic: identical bits.
    constexpr bool PRE = NI == 2 && SWAP && (EPI == EPI_BIAS_RESID || EPI == EPI_BIAS_GELU || EPI == EPI_STORE || EPI == EPI_QKV);
    float4 bpre[PRE ? NI : 1], rpre[(PRE && EPI == EPI_BIAS_RESID) ? NI : 1][NI];
    if constexpr (PRE) {
        if (kg == 0) {
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int n = n0 + wn * (16 * NI) + j * 16 + 4 * g;
                bpre[j] = (p.bias != nullptr && n + 3 < N) ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (EPI == EPI_BIAS_RESID) {
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        const int m = m0 + wm * (16 * NI) + i * 16 + fr;
                        rpre[i][j] = (m < p.m_valid && n + 3 < N) ? *reinterpret_cast<const float4*>(p.resid + (long)m * p.ldo + n)
                                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int d = 0; d < PF; ++d)
        if (d < nk) gload(d, d);
    lstore(0, 0);
    __syncthreads();
    int k0 = 0;
    // steady state: every step prefetches a full k-step and hands one to LDS -- no conditions inside, so the compiler's
    // s_waitcnt placement stays counted (vmcnt(2 * NP * (PF - 1))) instead of collapsing to vmcnt(0) at block merges
    for (; k0 + 2 * PF < nk; k0 += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {        // register-set indices are compile-time after unrolling
            gload_full(d, k0 + d + PF);       // set d went to LDS during the previous step: free again
            compute((k0 + d) & 1);
            lstore((d + 1) % PF, (k0 + d + 1) & 1);
            __syncthreads();
        }
    }
    for (; k0 < nk; k0 += PF) {               // drain (and the K tail, if any)
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const int kt = k0 + d;
            if (kt < nk) {
                if (kt + PF < nk) gload(d, kt + PF);
                compute(kt & 1);
                if (kt + 1 < nk) lstore((d + 1) % PF, (kt + 1) & 1);
                __syncthreads();
            }
        }
    }

    // ---------------- k-groups: group 0 collects the other groups' accumulators in group order ----------------
    if constexpr (KG > 1) {
        f32x4* stage = reinterpret_cast<f32x4*>(&lds_all[kg][0][0][0]);     // this group's LDS stage is free after the last barrier
        if (kg != 0) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) stage[(i * NI + j) * 256 + t] = acc[i][j];
        }
        __syncthreads();
        if (kg != 0) return;
#pragma unroll
        for (int g2 = 1; g2 < KG; ++g2) {
            const f32x4* src = reinterpret_cast<const f32x4*>(&lds_all[g2][0][0][0]);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const f32x4 u = src[(i * NI + j) * 256 + t];
                    acc[i][j][0] += u[0]; acc[i][j][1] += u[1]; acc[i][j][2] += u[2]; acc[i][j][3] += u[3];
                }
        }
    }

    // ---------------- epilogue ----------------
    OutT* __restrict__ out = static_cast<OutT*>(p.out);
    const int mv = p.m_valid;
    typename OutRange<OutT>::type range;
    if constexpr (EPI == EPI_SCORE_FILTER) {
        // the ragged tail (< 256 documents) of a threshold-filtered scorer pass: same rule as the 256x256 kernel's epilogue
        // (only scores STRICTLY above the query's running k-th best are appended; NaN -> -1 first), columns >= N masked
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int m = m0 + wm * (16 * NI) + i * 16 + fr;
            if (m >= mv) continue;
            const float th = p.thr[(long)m * p.thr_ld];
            // (capacity is checked per survivor below: a check up here would put a global load on every row's path)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + wn * (16 * NI) + j * 16 + 4 * g + r;
                    float v = acc[i][j][r];
                    v = v != v ? -1.0f : v;
                    if (n < N && v > th && cand_room(p.cand_cnt + m, p.cand_cap)) {
                        const int slot = atomicAdd(p.cand_cnt + m, 1);
                        if (slot < p.cand_cap) {
                            p.cand_val[(long)m * p.cand_cap + slot] = v;
                            p.cand_idx[(long)m * p.cand_cap + slot] = p.idx_base + n;
                        }
                    }
                }
        }
    } else if constexpr (SWAP) {
        // lane: m = .. + fr ; n = .. + 4g + r  -> 4 consecutive n, row-major vector store
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int m = m0 + wm * (16 * NI) + i * 16 + fr;
            if (m >= mv) continue;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int n = n0 + wn * (16 * NI) + j * 16 + 4 * g;
                if (n >= N) continue;
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                const bool full = n + 3 < N;
                // f16 range shifts, the arithmetic of gemm256_epilogue.inc bit for bit (all factors 1 unless a model carries
                // shifts): store (acc * in_mul + bias) * out_mul as one fma; gelu(fma(acc, in_mul, bias)) * out_mul
                float om = p.out_mul;
                if constexpr (EPI == EPI_QKV) om = n >= p.n_split ? p.out_mul2 : p.out_mul;
                const float cs = (EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RESID) ? p.in_mul : p.in_mul * om;
                const float bsc = EPI == EPI_BIAS_GELU ? 1.0f : om;
                if constexpr (EPI == EPI_BIAS_RESID) {
                    // acc (* in_mul) + (bias + resid): the association of gemm256_kernel's epilogue, bit for bit
                    float4 bb, rr;
                    if constexpr (PRE) { bb = bpre[j]; rr = rpre[i][j]; }
                    else {
                        bb = *reinterpret_cast<const float4*>(p.bias + n);
                        rr = *reinterpret_cast<const float4*>(p.resid + (long)m * p.ldo + n);
                    }
                    v[0] = __builtin_fmaf(v[0], cs, bb.x + rr.x); v[1] = __builtin_fmaf(v[1], cs, bb.y + rr.y);
                    v[2] = __builtin_fmaf(v[2], cs, bb.z + rr.z); v[3] = __builtin_fmaf(v[3], cs, bb.w + rr.w);
                } else if (EPI == EPI_BIAS_GELU || ((EPI == EPI_STORE || EPI == EPI_QKV) && p.bias != nullptr)) {
                    if (full) {
                        float4 bb;
                        if constexpr (PRE) bb = bpre[j]; else bb = *reinterpret_cast<const float4*>(p.bias + n);
                        v[0] = __builtin_fmaf(v[0], cs, bb.x * bsc); v[1] = __builtin_fmaf(v[1], cs, bb.y * bsc);
                        v[2] = __builtin_fmaf(v[2], cs, bb.z * bsc); v[3] = __builtin_fmaf(v[3], cs, bb.w * bsc);
                    } else {       // ragged last column group (LM head: vocab % 4 != 0): no read past the bias array
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(v[r], cs, (n + r < N) ? p.bias[n + r] * bsc : 0.f);
                    }
                } else if constexpr (sizeof(OutT) == 2) {       // 16-bit store without bias (GPT-Neo / GPT-J q | k | v)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(v[r], cs, 0.f);
                }
                if constexpr (EPI == EPI_BIAS_GELU) {
                    const float ginv = 1.0f / om;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = sizeof(T) == 2 ? gelu_new_fast_scaled(v[r], ginv) : gelu_new(v[r]);
                }
                if constexpr (EPI == EPI_SCORE) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (v[r] != v[r]) ? -1.0f : v[r];  // exact_search.py:99
                }
                range.note(v[0], v[1]); range.note(v[2], v[3]);
                if constexpr (EPI == EPI_QKV) {
                    if (n >= p.n_split) {        // V columns: transposed (scattered 2-byte stores; query-sized batches only)
                        OutT* vt = static_cast<OutT*>(p.out2) + (long)(n - p.n_split) * p.ldo2 + m;
#pragma unroll
                        for (int r = 0; r < 4; ++r) store1<OutT>(vt + (long)r * p.ldo2, v[r]);
                        if constexpr (sizeof(OutT) == 2 && sizeof(T) == 2) {
                            if (p.lo_delta2 != 0) {
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                    store1<OutT>(vt + (long)r * p.ldo2 + p.lo_delta2, v[r] - Half<OutT>::lo(Half<OutT>::pack2(v[r], 0.f)));
                            }
                        }
                        continue;
                    }
                }
                OutT* dst = out + (long)m * p.ldo + n;
                if (full) {
                    store4<OutT>(dst, v[0], v[1], v[2], v[3]);
                    if constexpr (sizeof(OutT) == 2 && sizeof(T) == 2) {
                        if (p.lo_delta != 0) {    // split-precision output: lo = round16(v - hi) (+ a second copy of hi), see GemmArgs
                            const uint32_t h01 = Half<OutT>::pack2(v[0], v[1]), h23 = Half<OutT>::pack2(v[2], v[3]);
                            store4<OutT>(dst + p.lo_delta, v[0] - Half<OutT>::lo(h01), v[1] - Half<OutT>::hi(h01),
                                         v[2] - Half<OutT>::lo(h23), v[3] - Half<OutT>::hi(h23));
                            if (p.hi2_delta != 0) store4<OutT>(dst + p.hi2_delta, v[0], v[1], v[2], v[3]);
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n + r < N) store1<OutT>(dst + r, v[r]);
                }
            }
        }
    } else {
        // lane: m = .. + 4g + r ; n = .. + fr  -> 4 consecutive m: transposed store out[n][m..m+3]
        static_assert(SWAP || EPI == EPI_VT, "non-swapped orientation is only used for the V^T store");
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = n0 + wn * (16 * NI) + j * 16 + fr;
            if (n >= N) continue;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int m = m0 + wm * (16 * NI) + i * 16 + 4 * g;
                if (m >= M) continue;  // M (token axis) is padded to a multiple of 128 by the caller
                const float csv = p.in_mul * p.out_mul;                   // f16 range shifts (1 by default)
                const float bn = (p.bias ? p.bias[n] : 0.f) * p.out_mul;   // BLOOM: V projection has a bias
                const float v0 = __builtin_fmaf(acc[i][j][0], csv, bn), v1 = __builtin_fmaf(acc[i][j][1], csv, bn);
                const float v2 = __builtin_fmaf(acc[i][j][2], csv, bn), v3 = __builtin_fmaf(acc[i][j][3], csv, bn);
                range.note(v0, v1); range.note(v2, v3);
                store4<OutT>(out + (long)n * p.ldo + m, v0, v1, v2, v3);
                if constexpr (sizeof(OutT) == 2 && sizeof(T) == 2) {
                    if (p.lo_delta != 0) {
                        const uint32_t h01 = Half<OutT>::pack2(v0, v1), h23 = Half<OutT>::pack2(v2, v3);
                        store4<OutT>(out + (long)n * p.ldo + m + p.lo_delta, v0 - Half<OutT>::lo(h01), v1 - Half<OutT>::hi(h01),
                                     v2 - Half<OutT>::lo(h23), v3 - Half<OutT>::hi(h23));
                    }
                }
            }
        }
    }
    range.finish(p.range_flag, p.range_amax);
}


// ------------------------------------------------------------------------------------------
// Variant D ("deep"): gemm256_kernel with an ASYMMETRIC LDS ring.  One operand of every encoder / scorer GEMM is
// small and L2-resident (weights, queries), the other streams from HBM (activations, corpus).  With two symmetric
// 64-KiB stages the streamed operand gets one k-step (~1.3 us) to arrive; a loaded-HBM round trip is longer, and
// the k-steps stretch from 2.7 k to 3.0-4.0 k ticks (fc2: 1 144 TFLOP/s without stores where the vendor GEMM
// reaches 1 308; the filtered scorer GEMM 1 035).  Here the streamed ("deep") operand has THREE 32-KiB slots and
// is fetched two k-steps ahead, the resident ("shallow") one keeps two slots: 96 + 64 = 160 KiB, the whole LDS.
// Per k-step s a wave issues shallow(s+1) x4 pieces and THEN deep(s+2) x4 pieces; DMA completes in order, so
// `s_waitcnt vmcnt(4)` at the step's barrier means "everything but the newest four pieces", i.e. shallow(s+1) and
// deep(s+1) have landed while deep(s+2) stays in flight.  The barrier sits in front of the step's last row pair and
// the next step's first fragments are read behind it (see the loop): 2 615 instead of 2 700 cycles per k-step.  Epilogue scratch: after a tile's last step one deep and one
// shallow slot are free (32 KiB each): waves 0-3 transpose through the first, waves 4-7 through the second.
// LO: the 16-bit store epilogues also write lo = round16(v - hi) (GemmArgs.lo_delta / hi2_delta) -- split-precision operands
// for the next GEMM or the attention kernel; a template parameter so that the default kernels keep their exact code.
template <typename T, int EPI, typename OutT, bool SWAP, bool DEEP_A, bool LO = false>
__global__ __launch_bounds__(512, 2) void gemm256d_kernel(const GemmArgs p) {
    if (p.pred != nullptr && *p.pred == 0) return;
    typedef __attribute__((address_space(3))) char* lds_cptr_t;
    constexpr int TM = 256, TN = 256;
    constexpr int SLOT = TM * CH;                                    // uint4 per 32-KiB slot (256 rows x 8 chunks)
    __shared__ __attribute__((aligned(16))) uint4 lds[5 * SLOT];    // [deep 0..2 | shallow 0..1] = 160 KiB

    const int N = p.N, K = p.K;
    const int MT = p.M / TM, NT = N / TN;
    const int GM = p.gm > 0 ? p.gm : 4, GN = p.gn > 0 ? p.gn : 8;
    const bool m_major = MT >= NT;
    const int AT = m_major ? MT : NT, BT = m_major ? NT : MT;
    const int per_band = GM * BT;
    // Few-round launches (p.balanced, mid-size batches): the supertile order gives XCD x the A-tiles x, x+8, ... -- BT x
    // ceil or floor(AT / 8) tiles -- and a launch of 85 x 3 tiles put 33 on five XCDs' 32 CUs: a second round for 3 tiles
    // (the "lone round costs 1.67 x" of round 3).  Here XCD x takes a contiguous run of the A-major tile list instead, runs that
    // differ by at most one tile; the B-tiles of one A-tile still sit next to each other on one XCD.
    const int R = AT * BT, c0 = R >> 3, rem = R & 7;
    const int tiles_total = p.balanced ? 8 * ((R + 7) >> 3) : ((AT + 7) / 8 + GM - 1) / GM * GM * 8 * BT;
    auto tile_coords = [&](int tile, int& m0, int& n0) -> bool {
        if (p.balanced) {
            const int xcd = tile & 7, l = tile >> 3;
            if (l >= c0 + (xcd < rem ? 1 : 0)) return false;
            const int gi = xcd * c0 + (xcd < rem ? xcd : rem) + l;
            const int at = gi / BT, bt = gi - at * BT;
            m0 = (m_major ? at : bt) * TM; n0 = (m_major ? bt : at) * TN;
            return true;
        }
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
    const int fr = lane & 15, g = lane >> 4;

    const bf16_t* __restrict__ Ag = static_cast<const bf16_t*>(p.A);
    const bf16_t* __restrict__ Wg = static_cast<const bf16_t*>(p.W);
    const int lchunk = (lane & 7) ^ (lane >> 3);
    const unsigned lds_base = (unsigned)(size_t)(lds_cptr_t)(&lds[0]);
    // LDS-DMA piece: wave-uniform 64-bit base in SGPRs + a per-lane 32-bit byte offset, M0 = LDS destination.
    // (A per-lane 64-bit pointer per piece costs two VALU adds each -- 14 of the ~30 VALU instructions of a k-step,
    // and VALU issue competes with MFMA issue; M0 is declared clobbered instead of being saved and restored.)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned a_loff = (unsigned)(((lane >> 3) * p.lda + lchunk * 8) * 2);
    const unsigned w_loff = (unsigned)(((lane >> 3) * p.ldw + lchunk * 8) * 2);
    auto dma16 = [&](const char* base_uniform, unsigned lane_off, unsigned dst_byte) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :
                     : "v"(lane_off), "s"(base_uniform), "s"(dst_byte)
                     : "memory");
    };
    // slot byte offsets: deep slot sd in [0,3), shallow slot ss in [0,2)
    auto deep_off = [&](int sd) { return (unsigned)(sd * SLOT * 16); };
    auto shal_off = [&](int ss) { return (unsigned)((3 + ss) * SLOT * 16); };
    // piece q (0..3) of this wave's 32 rows of one operand; `src` = wave-uniform pointer to the operand tile's row 0
    auto piece = [&](const bf16_t* src, long ld, unsigned loff, int kt, unsigned slot_off, int q) {
        const unsigned row_off = (unsigned)((wave_u * 32 + q * 8) * CH * 16);
        dma16(reinterpret_cast<const char*>(src + (long)(wave_u * 32 + q * 8) * ld + kt * 64), loff, lds_base + slot_off + row_off);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = K / 64;   // >= 2 (launcher)
    OutT* __restrict__ out = static_cast<OutT*>(p.out);

    int tile = blockIdx.x, m0 = 0, n0 = 0;
    while (tile < tiles_total && !tile_coords(tile, m0, n0)) tile += gridDim.x;
    if (tile >= tiles_total) return;
    // EPI_QKV (the bulk fused QKV projection, N = 3 d in ONE launch): a tile of the q | k columns (n0 < n_split) is the plain
    // store tile; a tile of the V columns is computed with the operand ROLES SWAPPED -- the V weight rows take the "A" side, the
    // token rows the "W" side (lda == ldw, checked by the launcher: the per-lane DMA offsets are the same) -- so its accumulators
    // hold V^T[n][m] in the row-major store orientation and the SAME 16-bit store epilogue writes whole 128-byte runs of
    // out2[n][m..] (what EPI_VT does with the non-swapped MFMA orientation; same k-ascending sums per element: same bits).
    auto tile_src = [&](int tm0, int tn0, const bf16_t*& a_, const bf16_t*& w_) {
        if constexpr (EPI == EPI_QKV) {
            if (tn0 >= p.n_split) { a_ = Wg + (long)tn0 * p.ldw; w_ = Ag + (long)tm0 * p.lda; return; }
        }
        a_ = Ag + (long)tm0 * p.lda; w_ = Wg + (long)tn0 * p.ldw;
    };
    // per-lane source rows of the two operands for the current tile
    const bf16_t *asrc, *wsrc;                            // wave-uniform tile bases
    tile_src(m0, n0, asrc, wsrc);
    const long dld = DEEP_A ? p.lda : p.ldw, sld = DEEP_A ? p.ldw : p.lda;
    const unsigned dloff = DEEP_A ? a_loff : w_loff, sloff = DEEP_A ? w_loff : a_loff;
    int sd = 0, ss = 0;      // ring slots of the k-step about to be computed
    int dbg_tile = 0;
    // Ragged document count (EPI_SCORE_FILTER, documents = the streamed W operand): the last column tile holds p.n_valid - tail_n0
    // real rows.  Its DMA pieces read CLAMPED rows (per-lane byte offsets from the tile's row 0, computed once; the normal tiles
    // use the same form so that the k-loop carries one uniform select per deep piece and no branch) and the epilogue masks the
    // columns that do not exist.  The separate small-tile launch for the < 256 trailing documents (9 us of a 0.27 ms shard pass,
    // plus a second no-op launch in the fallback) is gone.
    // (EPI_SCORE, the materialise-and-select pieces: same clamped rows, and the columns that do not exist are stored as copies of
    // the last real document's score into the padding of the score tile's row -- the select reads n_valid columns)
    constexpr bool TAIL = SGPT_FOLD_TAIL != 0 && (EPI == EPI_SCORE_FILTER || (SGPT_FOLD_TAIL_SCORE != 0 && EPI == EPI_SCORE)) && !DEEP_A;
    const bool has_tail = TAIL && p.n_valid > 0 && p.n_valid < N;
    const int tail_n0 = N - TN, tail_rows = has_tail ? p.n_valid - tail_n0 : TN;
    unsigned wnorm_off[TAIL ? 4 : 1], wtail_off[TAIL ? 4 : 1];
    if constexpr (TAIL) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = wave * 32 + q * 8 + (lane >> 3);
            const int crow = row < tail_rows ? row : tail_rows - 1;
            wnorm_off[q] = (unsigned)((row * p.ldw + lchunk * 8) * 2);
            wtail_off[q] = (unsigned)((crow * p.ldw + lchunk * 8) * 2);
        }
    }
    auto dpiece = [&](const bf16_t* src, bool tail, int kt, unsigned slot_off, int q) {
        if constexpr (TAIL) {
            const unsigned row_off = (unsigned)((wave_u * 32 + q * 8) * CH * 16);
            dma16(reinterpret_cast<const char*>(src + kt * 64), tail ? wtail_off[q] : wnorm_off[q], lds_base + slot_off + row_off);
        } else {
            (void)tail;
            piece(src, dld, dloff, kt, slot_off, q);
        }
    };
    typename OutRange<OutT>::type range;
#ifdef SGPT_EXPERIMENTS
#define STAMP(k)                                                                                   \
    if (p.dbg && blockIdx.x == 0 && t == 0 && dbg_tile < 8) p.dbg[dbg_tile * 8 + (k)] = (long long)__builtin_amdgcn_s_memtime()
#else
#define STAMP(k) (void)0
#endif

    // Start-up stagger (p.skew shader cycles per phase, 4 phases per XCD): every workgroup walks equally long tiles, so
    // without it all 256 CUs reach their store / read-modify-write epilogues at the same moment and the chip alternates
    // between an HBM-bound burst and an MFMA-bound phase.
#ifdef SGPT_EXPERIMENTS
    if (p.skew > 0) {
        const int phase = (blockIdx.x >> 3) & 3;
        const long until = (long)__builtin_amdgcn_s_memtime() + (long)phase * p.skew;
        while (phase && (long)__builtin_amdgcn_s_memtime() < until) __builtin_amdgcn_s_sleep(32);
    }
#endif
    {   // prologue: deep(0), shallow(0), deep(1) -- in the order the waits assume
        const bf16_t* dsrc = DEEP_A ? asrc : wsrc;
        const bf16_t* ssrc = DEEP_A ? wsrc : asrc;
        const bool t0 = has_tail && n0 == tail_n0;
#pragma unroll
        for (int q = 0; q < 4; ++q) dpiece(dsrc, t0, 0, deep_off(0), q);
#pragma unroll
        for (int q = 0; q < 4; ++q) piece(ssrc, sld, sloff, 0, shal_off(0), q);
#pragma unroll
        for (int q = 0; q < 4; ++q) dpiece(dsrc, t0, 1, deep_off(1), q);
    }
    // Late-barrier pipeline: the k-step's barrier sits in front of its LAST row pair, and the first fragments of the
    // next k-step are read behind it, under that pair's MFMAs -- no ds_read latency is left between a barrier and the
    // first MFMA.  Fragments roll through two W sets (one per 32-wide slice) and two A row pairs.
    uint4 wf[2][4], af[2][2];
    auto ld_w = [&](const uint4* lw, int ks, int j) { const int row = wn * 64 + j * 16 + fr; return lw[row * CH + ((4 * ks + g) ^ (row & 7))]; };
    auto ld_a = [&](const uint4* la, int ks, int i) { const int row = wm * 128 + i * 16 + fr; return la[row * CH + ((4 * ks + g) ^ (row & 7))]; };
    {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // deep(0), shallow(0) landed
        __syncthreads();
        const uint4* la = lds + (DEEP_A ? sd * SLOT : (3 + ss) * SLOT);
        const uint4* lw = lds + (DEEP_A ? (3 + ss) * SLOT : sd * SLOT);
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[0][j] = ld_w(lw, 0, j);
        af[0][0] = ld_a(la, 0, 0); af[0][1] = ld_a(la, 0, 1);
    }
    // The tile after this one is looked up one tile ahead, inside the k-loop (its integer divisions run in the shadow of
    // k-step 0's MFMAs instead of sitting between two tiles: ~1 k cycles per tile on the critical path otherwise).
    int ntile = tile + gridDim.x, nm0 = 0, nn0 = 0;
    while (ntile < tiles_total && !tile_coords(ntile, nm0, nn0)) ntile += gridDim.x;
    // EPI_SCORE_FILTER: the thresholds of this tile's rows, loaded under k-step 0 -- at the head of the epilogue they were an
    // un-hidden L2 round trip per tile (behind the next tile's run-ahead DMA, which the compiler's vmcnt wait also retires)
    float thv_pre[8];
    // (Round 5 probe, removed again: the GELU launch's bias fragments fetched under k-step 0 by inline-asm loads the compiler's
    // s_waitcnt bookkeeping does not see -- so that the epilogue's head carries no `vmcnt(0)` behind the next tile's run-ahead
    // DMA -- changed nothing: fc1 628.9 -> 643.4 us median, 628.7 -> 630.9 best, same box.  The wait is not where the epilogue's
    // time goes; neither was the `vmcnt(0)` the conditional threshold loads of the filtered scorer launch had put in front of
    // every tile's second k-step -- removed, 0.28 / 1.47 ms per pass either way, profiles/r05_score_thv_ab.txt.)
    while (true) {
        const bool has_next = ntile < tiles_total;
        int n2tile = ntile, n2m0 = 0, n2n0 = 0;
        const bf16_t *nasrc = asrc, *nwsrc = wsrc;                         // past the end: harmless re-fetch
        if (has_next) tile_src(nm0, nn0, nasrc, nwsrc);
        const bf16_t* d_cur = DEEP_A ? asrc : wsrc, *d_nxt = DEEP_A ? nasrc : nwsrc;
        const bf16_t* s_cur = DEEP_A ? wsrc : asrc, *s_nxt = DEEP_A ? nwsrc : nasrc;
        const bool tail_cur = has_tail && n0 == tail_n0, tail_nxt = has_tail && has_next && nn0 == tail_n0;
        for (int kt = 0; kt < nk; ++kt) {
#ifdef SGPT_EXPERIMENTS
            if (p.dbg && blockIdx.x == 0 && t == 0 && dbg_tile < 4 && kt < 12)
                p.dbg[64 + dbg_tile * 16 + kt] = (long long)__builtin_amdgcn_s_memtime();
#endif
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
            for (int q = 0; q < 8; ++q) {
                const int ks = q >> 2, pr = q & 3;
                if (q < 7) {                    // fragments of the next row pair (and the next slice's W set)
                    const int nks = (q + 1) >> 2, npr = (q + 1) & 3;
                    af[(q + 1) & 1][0] = ld_a(la, nks, 2 * npr);
                    af[(q + 1) & 1][1] = ld_a(la, nks, 2 * npr + 1);
                    if (npr == 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) wf[nks][j] = ld_w(lw, nks, j);
                    }
                } else {                        // every read of this stage has returned; stage kt+1 has landed
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(4)" ::: "memory");
                    __syncthreads();
                    af[0][0] = ld_a(nla, 0, 0);
                    af[0][1] = ld_a(nla, 0, 1);
#pragma unroll
                    for (int j = 0; j < 4; ++j) wf[0][j] = ld_w(nlw, 0, j);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int i = 2 * pr + h;
#pragma unroll
                    for (int j = 0; j < 4; ++j) mma<T, SWAP>(acc[i][j], af[q & 1][h], wf[ks][j]);
                    if (ks == 0) {   // one 1-KiB piece behind every 4 MFMAs: shallow x4 first, then deep x4
                        if (i < 4) piece(sp, sld, sloff, skt, s_dst, i);
                        else dpiece(dp, d_in ? tail_cur : tail_nxt, dkt, d_dst, i - 4);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            ss ^= 1;
            sd = sdn;
            if constexpr (EPI == EPI_SCORE_FILTER) {
                if (kt == 0) {
                    // UNCONDITIONAL loads from a clamped row (padded query rows are masked in the epilogue): written as
                    // `m < m_valid ? load : inf`, every tile's k-step 0 began with `v_mov inf` into registers whose previous
                    // loads the compiler's counter model still held pending across the k-loop's back edge -- it cannot count the
                    // inline-asm DMA pieces, so the write-after-write wait came out as `s_waitcnt vmcnt(0)`: a full memory round
                    // trip, the deep operand's run-ahead pieces included, in front of every tile's second k-step (round 5;
                    // the same compiler bookkeeping that stalled the attention tile loop until round 4).  Load after load to the
                    // same register needs no wait: VMEM loads return in order.
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        int m = m0 + wm * 128 + i * 16 + fr;
#if SGPT_THV_UNCOND
                        m = m < p.m_valid ? m : p.m_valid - 1;
                        thv_pre[i] = p.thr[(long)m * p.thr_ld];
#else
                        thv_pre[i] = m < p.m_valid ? p.thr[(long)m * p.thr_ld] : INFINITY;     // (A/B build: rounds 3-4)
#endif
                    }
                }
            }
            if (kt == 0 && has_next) {          // look up the tile after the next one (wave-uniform scalar work)
                n2tile = ntile + gridDim.x;
                while (n2tile < tiles_total && !tile_coords(n2tile, n2m0, n2n0)) n2tile += gridDim.x;
            }
        }
        STAMP(0);
        STAMP(1);
        STAMP(2);
        {
            const int sdf = sd + 2 >= 3 ? sd - 1 : sd + 2;
            char* scr = reinterpret_cast<char*>(lds) + (wave < 4 ? deep_off(sdf) : shal_off(ss ^ 1)) + (wave & 3) * 8192;
#include "gemm256_epilogue.inc"
        }
        STAMP(3);
        ++dbg_tile;
        __syncthreads();       // every wave is done with its scratch before the next tile's DMA re-uses those slots
        if (!has_next) break;
        tile = ntile; m0 = nm0; n0 = nn0; asrc = nasrc; wsrc = nwsrc;
        ntile = n2tile; nm0 = n2m0; nn0 = n2n0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the run-ahead DMA before the LDS is released
    range.finish(p.range_flag, p.range_amax);
#undef STAMP
}

template <typename T, int EPI, typename OutT, bool SWAP, bool LO = false>
void launch256d(const GemmArgs& a, hipStream_t s, bool deep_a) {
    const int MT = a.M / 256, NT = a.N / 256;
    const int AT = MT >= NT ? MT : NT, BT = MT >= NT ? NT : MT;
    static const int ncu = [] {
        int dev = 0, n = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n / 8 * 8;
    }();
    GemmArgs b = a;
#ifdef SGPT_EXPERIMENTS
    static const int env_gm = getenv("SGPT_GM") ? atoi(getenv("SGPT_GM")) : 0, env_gn = getenv("SGPT_GN") ? atoi(getenv("SGPT_GN")) : 0;
    if (env_gm > 0) b.gm = env_gm;
    if (env_gn > 0) b.gn = env_gn;
#endif
    const int gm = b.gm > 0 ? b.gm : 4;
    b.balanced = (long)AT * BT <= 4L * ncu ? 1 : 0;        // up to ~4 rounds: XCD-balanced tile runs (see the kernel)
    const int tiles_pad = b.balanced ? 8 * ((AT * BT + 7) / 8) : ((AT + 7) / 8 + gm - 1) / gm * gm * 8 * BT;
    // a.cu_cap (per ctx): leave CUs to a second pipeline on another stream (its LayerNorm / attention / embed kernels run on
    // the free CUs while this launch is in its k-loops); multiples of 8 keep the XCD interleave of the tile order
    const int cus = (a.cu_cap >= 8 && a.cu_cap < ncu) ? a.cu_cap / 8 * 8 : ncu;
    const int grid = tiles_pad < cus ? tiles_pad : cus;
#ifdef SGPT_EXPERIMENTS
    b.skew = tiles_pad >= 4 * grid ? g_skew : 0;       // fewer than ~4 tiles per workgroup: the delay is not amortised
#else
    b.skew = 0;
#endif
    if (deep_a) hipLaunchKernelGGL((gemm256d_kernel<T, EPI, OutT, SWAP, true, LO>), dim3(grid), dim3(512), 0, s, b);
    else hipLaunchKernelGGL((gemm256d_kernel<T, EPI, OutT, SWAP, false, LO>), dim3(grid), dim3(512), 0, s, b);
}


// ------------------------------------------------------------------------------------------
// Scorer tile for SHORT query batches (nq <= 64): 64 query rows x 256 documents per workgroup.
// With the 256-row tile above a 16-query search pays for 256 padded query rows -- 16x the MFMA work of a pass that is
// HBM-bound by nature (1 M x 768 x 2 B = 1.5 GB of corpus per pass: 0.3 ms at 5 TB/s; measured 0.76 ms).
// Here a wave owns 32 documents (all 64 queries: 4 x 2 fragments, 16 MFMAs per k-step), the corpus streams through
// the same three 32-KiB slots two k-steps ahead, the queries (8 KiB per k-step, L2-resident) through two small slots.
// Scores leave the registers directly (one query row, 4 consecutive documents per lane): EPI_SCORE stores them,
// EPI_SCORE_FILTER appends the ones above the running k-th best to the candidate lists.  Same MFMA sequence per
// output element as every other kernel in this file.
template <typename T, int EPI>
__global__ __launch_bounds__(512, 2) void score64_kernel(const GemmArgs p) {
    if (p.pred != nullptr && *p.pred == 0) return;
    typedef __attribute__((address_space(3))) char* lds_cptr_t;
    constexpr int DSLOT = 256 * CH, QSLOT = 64 * CH;                           // uint4 per slot
    __shared__ __attribute__((aligned(16))) uint4 lds[3 * DSLOT + 2 * QSLOT];  // 96 + 16 KiB
    __shared__ __attribute__((aligned(16))) uint2 stage_all[EPI == EPI_SCORE_FILTER ? 8 * 256 : 1];   // filtered epilogue: 256 staged survivors per wave
    const int K = p.K, nk = K / 64, NT = p.N / 256;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const bf16_t* __restrict__ Qg = static_cast<const bf16_t*>(p.A);          // [64][K] (padded query rows)
    const bf16_t* __restrict__ Dg = static_cast<const bf16_t*>(p.W);          // [N][K] documents
    const unsigned lds_base = (unsigned)(size_t)(lds_cptr_t)(&lds[0]);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int lchunk = (lane & 7) ^ (lane >> 3);
    const unsigned q_loff = (unsigned)(((lane >> 3) * p.lda + lchunk * 8) * 2);
    const unsigned d_loff = (unsigned)(((lane >> 3) * p.ldw + lchunk * 8) * 2);
    auto dma16 = [&](const char* base_uniform, unsigned lane_off, unsigned dst_byte) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :
                     : "v"(lane_off), "s"(base_uniform), "s"(dst_byte)
                     : "memory");
    };
    auto doc_pieces = [&](const bf16_t* dsrc, int kt, int slot) {             // this wave's own 32 document rows
#pragma unroll
        for (int q = 0; q < 4; ++q)
            dma16(reinterpret_cast<const char*>(dsrc + (long)(wave_u * 32 + q * 8) * p.ldw + kt * 64), d_loff,
                  lds_base + (unsigned)((slot * DSLOT + (wave_u * 32 + q * 8) * CH) * 16));
    };
    auto qry_piece = [&](int kt, int slot) {                                   // query rows 8w .. 8w+7
        dma16(reinterpret_cast<const char*>(Qg + (long)(wave_u * 8) * p.lda + kt * 64), q_loff,
              lds_base + (unsigned)((3 * DSLOT + slot * QSLOT + wave_u * 8 * CH) * 16));
    };
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    int tile = blockIdx.x;
    if (tile >= NT) return;
    const bf16_t* dsrc = Dg + (long)tile * 256 * p.ldw;
    // prologue, in the order the waits assume: docs(0), queries(0), docs(1)
    doc_pieces(dsrc, 0, 0);
    qry_piece(0, 0);
    doc_pieces(dsrc, 1, 1);
    int sd = 0, ss = 0;
    while (true) {
        const int ntile = tile + gridDim.x;
        const bool has_next = ntile < NT;
        const bf16_t* ndsrc = has_next ? Dg + (long)ntile * 256 * p.ldw : dsrc;
        for (int kt = 0; kt < nk; ++kt) {
            // stage kt has landed: everything but the newest four pieces (docs(kt+1)) is complete
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __syncthreads();                                   // ... for every wave; the slots of stage kt-1 are free
            {   // prefetch: queries(kt+1) then docs(kt+2) -- across the tile boundary into the next tile's first steps
                const bool q_in = kt + 1 < nk, d_in = kt + 2 < nk;
                qry_piece(q_in ? kt + 1 : 0, ss ^ 1);
                doc_pieces(d_in ? dsrc : ndsrc, d_in ? kt + 2 : kt + 2 - nk, sd + 2 >= 3 ? sd - 1 : sd + 2);
            }
            const uint4* lq = lds + 3 * DSLOT + ss * QSLOT;
            const uint4* ld = lds + sd * DSLOT;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 qf[4], df[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int row = wave * 32 + j * 16 + fr;
                    df[j] = ld[row * CH + ((4 * ks + g) ^ (row & 7))];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = i * 16 + fr;
                    qf[i] = lq[row * CH + ((4 * ks + g) ^ (row & 7))];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) mma<T, true>(acc[i][j], qf[i], df[j]);   // document fragment = A-operand
            }
            ss ^= 1;
            sd = sd + 1 >= 3 ? 0 : sd + 1;
        }
        // ---- epilogue, straight from the registers: lane = query row i*16 + fr, documents n0 + 32 w + 16 j + 4 g .. + 3 ----
        const long n0 = (long)tile * 256 + wave * 32;
        if constexpr (EPI == EPI_SCORE_FILTER) {
            // survivors staged in LDS by ballot positions, one flush per tile (see gemm256_epilogue.inc)
            uint2_a* stage = reinterpret_cast<uint2_a*>(&stage_all[wave * 256]);
            int staged = 0;                                                // wave-uniform
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = i * 16 + fr;
                const float th = m < p.m_valid ? p.thr[(long)m * p.thr_ld] : INFINITY;
                float mx = -INFINITY;                                // fast reject on the raw accumulators
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, acc[i][j][r]);
                if (__ballot(th < -1.0f) != 0) {                      // thresholds below -1 (dot scores): the per-lane path, NaN -> -1 may survive
                    if ((mx > th || th < -1.0f) && cand_room(p.cand_cnt + m, p.cand_cap)) {
                        int c = 0;
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float v = acc[i][j][r];
                                acc[i][j][r] = v != v ? -1.0f : v;       // cos_scores[isnan] = -1 (exact_search.py:99)
                                c += acc[i][j][r] > th ? 1 : 0;
                            }
                        int slot = atomicAdd(p.cand_cnt + m, c);
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float v = acc[i][j][r];
                                if (v > th) {
                                    if (slot < p.cand_cap) {
                                        p.cand_val[(long)m * p.cand_cap + slot] = v;
                                        p.cand_idx[(long)m * p.cand_cap + slot] = p.idx_base + n0 + j * 16 + 4 * g + r;
                                    }
                                    ++slot;
                                }
                            }
                    }
                } else if (__ballot(mx > th) != 0) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = acc[i][j][r];
                            const bool sv = v > th;
                            const unsigned long long mask = __ballot(sv);
                            if (mask != 0) {
                                if (sv) {
                                    const int pos = staged + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                                    if (pos < 256) stage[pos] = make_uint2((unsigned)(m << 16) | (unsigned)(j * 16 + 4 * g + r), __float_as_uint(v));
                                }
                                staged += __builtin_popcountll(mask);
                            }
                        }
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (staged > 0) {
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (staged <= 256) {                                       // one global round trip per flush (see gemm256_epilogue.inc)
                    for (int e = lane; e < staged; e += 64) {
                        const uint2 ent = stage[e];
                        const int m = (int)(ent.x >> 16);
                        const int slot = atomicAdd(p.cand_cnt + m, 1);
                        if (slot < p.cand_cap) {
                            p.cand_val[(long)m * p.cand_cap + slot] = __uint_as_float(ent.y);
                            p.cand_idx[(long)m * p.cand_cap + slot] = p.idx_base + n0 + (int)(ent.x & 0xffffu);
                        }
                    }
                } else if (lane == 0) {
                    atomicAdd(p.cand_cnt + (int)(stage[0].x >> 16), p.cand_cap + 1);   // more than the scratch holds: force the fallback
                }
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = i * 16 + fr;
                if (m < p.m_valid) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                        v.x = v.x != v.x ? -1.0f : v.x; v.y = v.y != v.y ? -1.0f : v.y;
                        v.z = v.z != v.z ? -1.0f : v.z; v.w = v.w != v.w ? -1.0f : v.w;
                        *reinterpret_cast<float4*>(static_cast<float*>(p.out) + (long)m * p.ldo + n0 + j * 16 + 4 * g) = v;
                    }
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        if (!has_next) break;
        tile = ntile; dsrc = ndsrc;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the run-ahead DMA before the LDS is released
}

template <typename T, int EPI>
void launch_score64(const GemmArgs& a, hipStream_t s) {
    static const int ncu = [] {
        int dev = 0, n = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n;
    }();
    const int NT = a.N / 256;
    hipLaunchKernelGGL((score64_kernel<T, EPI>), dim3(NT < ncu ? NT : ncu), dim3(512), 0, s, a);
}

template <typename T, int EPI, typename OutT, bool SWAP>
void launch(const GemmArgs& a, hipStream_t s) {
    // 64x64 tiles when 128x128 ones would leave most of the 256 CUs idle (short query batches, USEB's 21-32 sentence
    // calls: M = 1024 x N = 768 is 48 tiles of 128^2 but 192 of 64^2).  Switch point measured on whole encodes
    // (SGPT_T128_MIN sweep): 1536 token rows prefer 64^2 tiles up to fc1's 288 tiles of 128^2 (1.26 -> 1.15 ms), 6912 rows
    // prefer 128^2 tiles from the N = 768 launches' 324 on.
    const long t128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    // Round 4 (scripts/small_tile_sweep.py, profiles/r04_small_tile_sweep.txt): with a short k-loop (K <= 1024: 12 k-steps) the
    // 64^2 tiles keep winning up to ~600 tiles of 128^2 -- 576 of them are 1.1 rounds of the 512 resident slots, the second
    // round 12 % full (fc1 at 3072 rows: 37.6 us on 128^2 tiles, 32.4 on 64^2; Q/K at 4096 rows 25.0 -> 22.5; out-projection
    // at 8192 rows 31.8 -> 29.2); the 48-step k-loop of fc2 keeps the 300 (8192 rows: 64.1 us on 128^2, 69.5 on 64^2).
    static const long t128_env = exp_env("SGPT_T128_MIN") ? atol(exp_env("SGPT_T128_MIN")) : -1;
    const long t128_min = t128_env >= 0 ? t128_env : (a.K <= 1024 ? 600 : 300);
    const bool small = t128 < t128_min;
    const int B = small ? 64 : 128;
    const int MT = (a.M + B - 1) / B, NT = (a.N + B - 1) / B;
    const int mt_per_xcd = (MT + 7) / 8;
    const int bands = (mt_per_xcd + 7) / 8;
    // (at most 512 tiles: XCD x = blocks x, x+8, ... walks its own run of the tile list; else the supertile order -- gemm_kernel)
    const int grid = (long)MT * NT <= 512 ? 8 * ((MT * NT + 7) / 8) : 8 * bands * 8 * NT;
    // Query-sized launches (at most one resident round of 64x64 tiles: 2 workgroups x 256 CUs) walk 128-element k-steps
    // (CHS = 16): half as many load -> LDS -> MFMA round trips on each tile's serial chain, same MFMA order per output
    // element, so results stay bit-identical.  16-query encode (512 token rows) 0.97 -> 0.82 ms; with more tiles than that
    // the 64-KiB stages cost occupancy instead (3072 rows: 1.77 -> 1.85 ms), so those keep 64-element steps.  Threshold
    // sweep 256 / 512 / 1024 tiles and 256-element steps (128 KiB of LDS, slower everywhere): profiles/r03_small_kstep_ab.txt.
    const long tiles = (long)MT * NT;
    const bool deep = small && tiles <= SGPT_DEEP_TILES;
    // two k-groups (split-K inside the workgroup, see gemm_kernel) when the 64x64 tiles leave workgroup slots empty (two
    // 512-thread workgroups fit a CU).  OFF by default: it trades the bit-identical-across-batch-sizes property for
    // latency (per ctx: sgpt_ctx_set_low_latency -> GemmArgs.kgroups).
    const int kgmax = a.kgroups;
    constexpr bool splittable = EPI != EPI_SCORE && EPI != EPI_SCORE_FILTER;
    constexpr int EPC = ElemTraits<T>::EPC;
    const int nkt16 = (a.K + 16 * EPC - 1) / (16 * EPC);
    if constexpr (splittable) {
        if (deep && kgmax >= 2) {
            if (nkt16 % 2 == 0 && nkt16 / 2 >= SGPT_KG16_MIN) {
                hipLaunchKernelGGL((gemm_kernel<T, EPI, OutT, SWAP, 2, 2, 16>), dim3(grid), dim3(512), 0, s, a); return;
            }
        }
    }
    if (deep) hipLaunchKernelGGL((gemm_kernel<T, EPI, OutT, SWAP, 2, 1, 16>), dim3(grid), dim3(256), 0, s, a);
    else if (small) hipLaunchKernelGGL((gemm_kernel<T, EPI, OutT, SWAP, 2, 1, 8>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((gemm_kernel<T, EPI, OutT, SWAP, 4, 1, 8>), dim3(grid), dim3(256), 0, s, a);
}

// 16-bit operand format H (bf16_t | f16_t): the 256x256 LDS-DMA kernel where the shape allows, else the register-staged one
template <typename H>
void launch_gemm16(int epi, int out_dtype, const GemmArgs& a, hipStream_t s) {
    const bool o16 = dt_is16(out_dtype);
    // A folded ragged document tail (n_valid > 0: N = whole 256-document tiles + 256, the last tile's rows clamped / columns masked)
    // exists in the 256x256 scorer launches only.  The callers ask gemm_score_tail_foldable() -- the one predicate -- before they
    // fold; a launch that arrives here folded without satisfying it (an experiment build's switches, a future edit) is split into
    // the two launches the fold replaced instead of reaching a kernel that ignores n_valid (round 6, ADVICE r05: was abort()).
    if (a.n_valid > 0) {
        const bool scorer_ = epi == EPI_SCORE || epi == EPI_SCORE_FILTER || epi == EPI_SCORE_TOP2;
        if (!(scorer_ && gemm_score_tail_foldable(a.M, a.N, a.K))) {
            const int whole = a.N - 256;
            GemmArgs b = a;
            b.N = whole; b.n_valid = 0;
            if (whole > 0) launch_gemm16<H>(epi, out_dtype, b, s);
            GemmArgs t = a;
            t.W = static_cast<const char*>(a.W) + (size_t)whole * a.ldw * 2; t.N = a.n_valid - whole; t.n_valid = 0;
            if (a.out != nullptr) t.out = static_cast<float*>(a.out) + whole;
            t.idx_base = a.idx_base + whole;
            if (t.N > 0) launch_gemm16<H>(epi, out_dtype, t, s);
            return;
        }
    }
    // Small problems (short query batches, USEB's 21-32 sentence calls): fewer than half a wave of 256x256 tiles
    // leaves most of the 256 CUs idle; the register-staged kernel's 128x128 / 64x64 tiles fill the chip instead.
    // Bit-identical results (see gemm_kernel), so the embeddings stay batch-invariant.  SGPT_NO_SMALL_TILE=1: A/B.
    static const bool small_tiles = exp_env("SGPT_NO_SMALL_TILE") == nullptr;
    const bool scorer = epi == EPI_SCORE || epi == EPI_SCORE_FILTER || epi == EPI_SCORE_TOP2;
    // a.force256 (per ctx, sgpt_ctx_set_tile_policy): keep 256x256 tiles for problems the small-tile rule would hand to
    // the register-staged kernel -- kernel-level tests of single-tile shapes
    // (the GELU launch -- N = 4 d, a VALU-heavy epilogue -- wants a whole round of 256x256 tiles before the LDS-DMA kernel pays:
    //  fc1 at 3072 / 4096 rows is 144 / 192 tiles, 43.3 / 43.5 us there against 32.4 / 38.8 us on the small tiles; the other
    //  launches are faster on 256x256 tiles from half a round on -- Q/K at 6144 rows: 22.6 us against 31.7)
    const long few_limit = epi == EPI_BIAS_GELU ? 255 : SGPT_FEW_TILES;
    const bool few = small_tiles && !a.force256 && !scorer && (long)(a.M / 256) * (a.N / 256) <= few_limit;
    static const bool use256 = exp_env("SGPT_GEMM128") == nullptr;
    const bool shape256 = a.M % 256 == 0 && a.N % 256 == 0 && a.K % 64 == 0 && a.K >= 128;
    // short query batches: the 64-row scorer tile (M = 64 padded query rows, N % 256 == 0, K % 64 == 0, K >= 128)
    if (scorer && a.M == 64 && a.N % 256 == 0 && a.K % 64 == 0 && a.K >= 128) {
        if (epi == EPI_SCORE) return launch_score64<H, EPI_SCORE>(a, s);
        return launch_score64<H, EPI_SCORE_FILTER>(a, s);
    }
    if (epi == EPI_SCORE_FILTER && !shape256) return launch<H, EPI_SCORE_FILTER, float, true>(a, s);   // ragged document tail
    if (epi == EPI_QKV) {
        // bulk: the 256x256 kernel with per-tile operand roles (caller checked gemm_qkv_bulk()); else the query-sized form
        if (gemm_qkv_bulk(a.M, a.N, a.K, a.n_split, a.force256 != 0) && a.lda == a.ldw && a.bias == nullptr && a.lo_delta == 0 &&
            a.lo_delta2 == 0 && a.m_valid == a.M) {
            GemmArgs b = a;
#if defined(SGPT_QKV_GM) && defined(SGPT_QKV_GN)
            b.gm = SGPT_QKV_GM; b.gn = SGPT_QKV_GN;       // (A/B builds: supertile shape of the nine-column-tile launch)
#endif
            return launch256d<H, EPI_QKV, H, true>(b, s, true);
        }
        return launch<H, EPI_QKV, H, true>(a, s);                   // caller checked gemm_qkv_one_launch()
    }
    if (use256 && shape256 && (scorer || (!few && a.m_valid == a.M))) {
        const bool deep_a = a.M >= a.N;          // the longer axis is the streamed operand (tokens / documents)
#ifdef SGPT_EXPERIMENTS
        // experiment build only: the 32x32x16-MFMA re-tiling of the same kernel (gemm256w.hip; measured slower)
        if (g_use_w && (epi != EPI_STORE || o16))
            return launch_gemm256w(Half<H>::is_f16 ? DT_F16 : DT_BF16, epi, a, s, deep_a);
#endif
        if (a.lo_delta != 0) {                   // split-precision (hi + lo) outputs: 16-bit store epilogues only
            if (epi == EPI_STORE && o16) return launch256d<H, EPI_STORE, H, true, true>(a, s, deep_a);
            if (epi == EPI_VT) return launch256d<H, EPI_VT, H, false, true>(a, s, deep_a);
            if (epi == EPI_BIAS_GELU) return launch256d<H, EPI_BIAS_GELU, H, true, true>(a, s, deep_a);
            abort();
        }
        if (epi == EPI_SCORE) return launch256d<H, EPI_SCORE, float, true>(a, s, deep_a);
        if (epi == EPI_SCORE_FILTER) return launch256d<H, EPI_SCORE_FILTER, float, true>(a, s, deep_a);
        if (epi == EPI_SCORE_TOP2) return launch256d<H, EPI_SCORE_TOP2, float, true>(a, s, deep_a);
        if (epi == EPI_STORE && o16) return launch256d<H, EPI_STORE, H, true>(a, s, deep_a);
        if (epi == EPI_VT) return launch256d<H, EPI_VT, H, false>(a, s, deep_a);
        if (epi == EPI_BIAS_GELU) return launch256d<H, EPI_BIAS_GELU, H, true>(a, s, deep_a);
        if (epi == EPI_BIAS_RESID) return launch256d<H, EPI_BIAS_RESID, float, true>(a, s, deep_a);
        if (epi == EPI_NONE) return launch256d<H, EPI_NONE, H, true>(a, s, deep_a);
    }
    if (epi == EPI_STORE && o16) return launch<H, EPI_STORE, H, true>(a, s);
    if (epi == EPI_STORE && !o16) return launch<H, EPI_STORE, float, true>(a, s);
    if (epi == EPI_VT) return launch<H, EPI_VT, H, false>(a, s);
    if (epi == EPI_BIAS_GELU) return launch<H, EPI_BIAS_GELU, H, true>(a, s);
    if (epi == EPI_BIAS_RESID) return launch<H, EPI_BIAS_RESID, float, true>(a, s);
    if (epi == EPI_SCORE) return launch<H, EPI_SCORE, float, true>(a, s);
    abort();
}

}  // namespace

// The fused QKV projection as ONE launch (EPI_QKV: q | k row-major, V^T scattered) when the q | k part alone would take
// the small-tile kernel -- query-sized batches, where a launch costs ~8 us and the scatter stores are few; larger batches
// keep the two launches with their own store epilogues.  Same sums either way: identical bits.
bool gemm_qkv_one_launch(int M, int n_split, bool force256) {
    static const bool small_tiles = exp_env("SGPT_NO_SMALL_TILE") == nullptr && exp_env("SGPT_QKV_TWO") == nullptr;
    return small_tiles && !force256 && n_split % 128 == 0 && (long)(M / 256) * (n_split / 256) * 2 <= 256;
}
// Bulk batches: q | k | V^T from ONE launch of the 256x256 kernel (N = 3 d: nine column tiles per row tile at d = 768; the
// LayerNorm output panel is fetched once instead of once per launch and a launch boundary goes away).  Shapes the 256x256 kernel
// takes with more than half a wave of q | k tiles; n_split a multiple of 256.  The caller adds: no bias (GPT-Neo / GPT-J), plain
// operands (lda == ldw == K), no split outputs.  -DSGPT_QKV_BULK=0 builds the two-launch form for same-box A/Bs.
#ifndef SGPT_QKV_BULK
#define SGPT_QKV_BULK 1
#endif
bool gemm_qkv_bulk(int M, int N, int K, int n_split, bool force256) {
    (void)force256;
    if (!SGPT_QKV_BULK || exp_env("SGPT_QKV_TWO") != nullptr || exp_env("SGPT_GEMM128") != nullptr) return false;
    return M % 256 == 0 && N % 256 == 0 && n_split % 256 == 0 && n_split < N && K % 64 == 0 && K >= 128 &&
           (long)(M / 256) * (n_split / 256) > SGPT_FEW_TILES;
}
// THE predicate of the folded ragged tail (api.hip asks it before folding; launch_gemm16 re-checks it): a scorer launch of M padded
// query rows against N = (whole 256-document tiles + 256) documents of width K takes the 256x256 kernel with the documents as its
// streamed operand -- the only kernel that honours GemmArgs.n_valid.
bool gemm_score_tail_foldable(int M, long N, int K) {
    if (exp_env("SGPT_GEMM128") != nullptr) return false;
    return M >= 256 && M != 64 && M % 256 == 0 && N % 256 == 0 && N >= 512 && K % 64 == 0 && K >= 128 && (long)M < N;
}
#ifdef SGPT_EXPERIMENTS
int set_gemm_skew(int cycles) { const int old = g_skew; g_skew = cycles; return old; }
int set_gemm_use_w(int on) { const int old = g_use_w; g_use_w = on ? 1 : 0; return old; }
#endif

void launch_gemm(int dtype, int epi, int out_dtype, const GemmArgs& a0, hipStream_t s) {
    GemmArgs a = a0;          // a zero-initialised descriptor means "no range shifts"
    if (a.in_mul == 0.f) a.in_mul = 1.f;
    if (a.out_mul == 0.f) a.out_mul = 1.f;
    if (a.out_mul2 == 0.f) a.out_mul2 = 1.f;
    if (a.kgroups < 1) a.kgroups = 1;
    if (dtype == DT_BF16) return launch_gemm16<bf16_t>(epi, out_dtype, a, s);
    if (dtype == DT_F16) return launch_gemm16<f16_t>(epi, out_dtype, a, s);
    if (epi == EPI_STORE) return launch<float, EPI_STORE, float, true>(a, s);
    if (epi == EPI_BIAS_GELU) return launch<float, EPI_BIAS_GELU, float, true>(a, s);
    if (epi == EPI_BIAS_RESID) return launch<float, EPI_BIAS_RESID, float, true>(a, s);
    if (epi == EPI_SCORE) return launch<float, EPI_SCORE, float, true>(a, s);
    abort();
}
