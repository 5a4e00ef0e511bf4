// Query- and mid-sized projections: C = A W^T for batches of one to a few thousand token rows -- one query
// (`SentenceTransformer.encode("one query")`: SentenceTransformer.py:143-146, README.md:309-330), USEB's 21-sentence batches
// (useb/useb/useb/evaluators/askubuntu.py:144-148), a rank's eighth of a query set.  (Round 6.)
//
// What bounds these launches is not arithmetic (a 16-query fc2 is 1.6 GFLOP) but how fast ONE workgroup per CU can pull its
// (BM + BN) K 2 operand bytes, and the serial chain of a tile.  Measured on this part (profiles/r06_dma_stream_probe.txt): 64
// workgroups streaming 1 MiB each reach 4.2 TB/s through `global_load_lds_dwordx4` and 7.7 TB/s through `global_load_dwordx4`
// into registers + `ds_write_b128` (256 workgroups: 6.4 against 17.8) -- the LDS-DMA path that feeds the 256x256 bulk kernel at
// the 40 GB/s per CU it needs is the slower one when nothing but the stream matters.  So these kernels stage through registers:
//   * every thread keeps D ring stages (128 or 256 k-elements each) of its 16-byte chunks in flight in VGPRs -- 48-96 KiB per
//     CU requested before the first MFMA, refilled one stage per k-step; LDS holds two stages (write stage s while stage s-1 is
//     being read: one barrier per stage);
//   * tiles from 32 x 16 to 128 x 128, chosen per launch so that one round of workgroups covers the chip with the fewest
//     operand bytes per workgroup; tile list column-group-major in contiguous per-XCD runs (a weight panel lands in one L2);
//   * LN_A: the projection normalises its own A rows.  At K = d the rows a workgroup needs ARE whole rows of the residual stream:
//     the prologue runs the one-wave-per-row LayerNorm of elementwise.hip (rowln.h: same loads, reductions, arithmetic, same
//     bits) on the tile's rows -- their loads issued BEFORE the weight stages, so the statistics do not wait for the weights --
//     and writes the 16-bit panel straight into LDS; the LayerNorm launch, its round trip through HBM and its boundary are gone
//     (LN1 -> QKV and LN2 -> fc1: two of seven launches per block);
//   * every output element still receives the k-ascending chain of v_mfma_f32_16x16x32 and the epilogue arithmetic of the
//     other GEMM kernels: a sequence's embedding keeps its bits whichever kernel its batch size selects.
// Why not ONE persistent kernel per forward: a device-wide barrier with the release / acquire it needs costs 43 us across 256
// workgroups on this part (3.9 us without the cache maintenance that makes the hand-off visible), a dependent kernel boundary
// ~5 us including the hand-off (profiles/r06_grid_barrier_probe.txt).
// LDS image of a stage: rows of 256 B (512 B at KD = 256), 16-byte chunk c of row r at position c ^ (r & 15): the four 16-lane
// groups of a ds_read_b128 fragment read (rows fr, chunks 4 ks + g) cover all 64 banks once (scripts/lds_layout_check.py).
// The image is thread-linear (thread t writes bytes [16 t, 16 t + 16) of each 8-KiB slab), so the XOR is applied to each
// thread's SOURCE chunk.
#include <atomic>
#include <type_traits>

#include "common.h"
#include "rowln.h"

namespace {

constexpr int QNW = 8;       // waves per workgroup
constexpr int QNT = 64 * QNW;
// the staging registers are a clang vector type: an array of HIP's uint4 (a struct around a union) that lives across the k-loop is
// not split into registers by the compiler -- it went to scratch memory, 80-400 bytes per lane, with a wait after every load
typedef unsigned int qv4 __attribute__((ext_vector_type(4)));

template <typename OutT> __device__ __forceinline__ void q_store4(OutT* p, float a, float b, float c, float d) {
    if constexpr (sizeof(OutT) == 4) *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
    else *reinterpret_cast<uint2*>(p) = make_uint2(Half<OutT>::pack2(a, b), Half<OutT>::pack2(c, d));
}
template <typename OutT> struct QRange { typedef RangeTrack<bf16_t> type; };
template <> struct QRange<f16_t> { typedef RangeTrack<f16_t> type; };

// EPI: EPI_STORE (16-bit or fp32 out, optional bias) | EPI_QKV (q | k row-major, V^T through the role-swapped MFMA orientation)
//      | EPI_BIAS_GELU | EPI_BIAS_RESID.   NV = float4 loads per lane of a LayerNorm row (LN_A; d <= 256 NV).
// KD = k-elements per ring stage (128 | 256); D = stages in flight in registers (K / KD is a multiple of D).
// LN_A workgroups keep their normalised A panel and walk `q.group` consecutive column tiles, the weight stream running across the
// tile boundaries (one LayerNorm and one pipeline fill per workgroup).
template <typename T, int EPI, typename OutT, int BM, int BN, bool LN_A, int NV, int KD, int D>
__global__ __launch_bounds__(QNT) void qgemm_kernel(const QGemmArgs q) {
    const GemmArgs& p = q.g;
    constexpr int ROWB = KD * 2, CPR = KD / 8;                  // bytes per staged row, 16-B chunks per row
    constexpr int NFM = BM / 16, NFN = BN / 16;
    constexpr int WGN = NFN >= 4 ? 4 : NFN, WGM = QNW / WGN;    // waves along N / M
    constexpr int FN = NFN / WGN, FM = (NFM + WGM - 1) / WGM;   // fragments per wave
    static_assert(NFN % WGN == 0 && (NFM % WGM == 0 || NFM < WGM), "tile / wave grid");
    constexpr int A_ROWS = LN_A ? 0 : BM;
    constexpr int STAGE = (A_ROWS + BN) * ROWB;
    constexpr int CH = STAGE / (QNT * 16);                      // 16-byte chunks per thread and stage
    constexpr int RPJ = QNT / CPR;                              // rows one slab of QNT chunks covers
    static_assert(STAGE % (QNT * 16) == 0 && CH >= 1 && A_ROWS % RPJ == 0, "a stage is whole slabs; a slab is all A or all W");
    constexpr bool HAS_BIAS = EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RESID;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int K = p.K, nk = K / KD;
    const int G = LN_A ? q.group : 1;
    const int MT = (p.M + BM - 1) / BM, NT = p.N / BN, NG = (NT + G - 1) / G;
    // workgroup list column-group-major (the workgroups that share weight panels are neighbours), XCD x = blocks x, x + 8, ... takes a
    // contiguous run
    const int R = MT * NG, c0 = R >> 3, rem = R & 7;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    if (local >= c0 + (xcd < rem ? 1 : 0)) return;
    const int gi = xcd * c0 + (xcd < rem ? xcd : rem) + local;
    const int ng = gi / MT, mt = gi - ng * MT;
    const int m0 = mt * BM, nt0 = ng * G;
    const int ntiles = NT - nt0 < G ? NT - nt0 : G;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int wn = wave % WGN, wm = wave / WGN;

    const int panel_bytes = LN_A ? nk * BM * ROWB : 0;           // LN_A: [nk][BM rows][ROWB]
    char* const ring = smem + panel_bytes;                       // two stages
    const int bias_off = panel_bytes + 2 * STAGE;                // LN_A: fp32 bias of the workgroup's column tiles [G][BN]

    // this thread's chunks of a stage: chunk ci = j QNT + t -> row ci / CPR of the stage image (A rows first), position ci % CPR
    const char* srcp[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int ci = j * QNT + t;
        const int r = ci / CPR, pos = ci % CPR;
        const int chunk = pos ^ (r & 15);                        // XOR on the low four chunk bits: stays in the 256-byte half
        if (j * RPJ < A_ROWS) {
            const int row = m0 + r < p.M ? m0 + r : p.M - 1;     // ragged last row tile: clamped loads, predicated stores
            srcp[j] = reinterpret_cast<const char*>(static_cast<const T*>(p.A) + (long)row * p.lda) + chunk * 16;
        } else {
            srcp[j] = reinterpret_cast<const char*>(static_cast<const T*>(p.W) + (long)(nt0 * BN + r - A_ROWS) * p.ldw) + chunk * 16;
        }
    }
    const long wtile = (long)BN * p.ldw * 2;                     // bytes between two column tiles' weight rows
    qv4 stg[D][CH];
    auto load_into = [&](qv4 (&dst)[CH], int ti, int kt) __attribute__((always_inline)) {
        const long aoff = (long)kt * ROWB, woff = (long)ti * wtile + aoff;
#pragma unroll
        for (int j = 0; j < CH; ++j) dst[j] = *reinterpret_cast<const qv4*>(srcp[j] + (j * RPJ < A_ROWS ? aoff : woff));
    };

    // LayerNorm operands first (so that the statistics wait for these loads only), then the first D weight stages
    constexpr int RPW = BM / QNW;                    // LN_A: rows per wave
    constexpr int RB = RPW > 4 ? 4 : RPW;            //       rows per batch (registers: RB NV float4 in flight; all 8 rows of a 64-row tile at once
                                                     //       was measured: 12.3 us against 12.0 -- the rows' arithmetic is the cost, not the second round trip)
    float4 gg[LN_A ? NV : 1], bb[LN_A ? NV : 1];
    RowLN<NV, false> rl[LN_A ? RB : 1];
    auto ln_rows_load = [&](int batch) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int row = m0 + wave * RPW + batch * RB + u;
            rl[u].load(q.x + (long)(row < p.M ? row : p.M - 1) * K, K, lane);
        }
    };
    float bias_r[2] = {0.f, 0.f};                    // LN_A: the column tiles' bias, on its way to LDS (requested first: in-order returns)
    if constexpr (LN_A) {
        if (p.bias != nullptr) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (u * QNT + t < ntiles * BN) bias_r[u] = p.bias[nt0 * BN + u * QNT + t];
        }
        ln_rows_load(0);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            gg[i] = c < K ? *reinterpret_cast<const float4*>(q.ln_g + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            bb[i] = c < K ? *reinterpret_cast<const float4*>(q.ln_b + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
#pragma unroll
    for (int j = 0; j < D; ++j) {                                // (nk >= D: the launcher's rule)
        load_into(stg[j], 0, j);
        __builtin_amdgcn_sched_barrier(0);                       // stage order = issue order = arrival order (the scheduler would
    }                                                            //  cluster the loads by address register: stage 0 complete last)

    // epilogue operands that do not depend on the product (plain kernels, one tile per workgroup): fetched under the k-loop
    float4 bpre[FN], rpre[EPI == EPI_BIAS_RESID ? FM : 1][FN];
    if constexpr (!LN_A) {
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = nt0 * BN + (wn + WGN * j) * 16 + 4 * g;
            bpre[j] = (HAS_BIAS || p.bias != nullptr) ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (EPI == EPI_BIAS_RESID) {
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int fm = wm + WGM * i;
                    const int m = m0 + (fm < NFM ? fm : 0) * 16 + fr;
                    rpre[i][j] = *reinterpret_cast<const float4*>(p.resid + (long)(m < p.M ? m : p.M - 1) * p.ldo + n);
                }
            }
        }
    }

    if constexpr (LN_A) {
        // the column tiles' bias into LDS (no load inside the tile loop); visible after the first stage barrier
        if (p.bias != nullptr) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (u * QNT + t < ntiles * BN) reinterpret_cast<float*>(smem + bias_off)[u * QNT + t] = bias_r[u];
            for (int i = 2 * QNT + t; i < ntiles * BN; i += QNT) reinterpret_cast<float*>(smem + bias_off)[i] = p.bias[nt0 * BN + i];
        }
        // LayerNorm of the tile's BM rows, one wave per row (rowln.h), 16-bit result into the A panel
#pragma unroll
        for (int batch = 0; batch < RPW / RB; ++batch) {
            if (batch > 0) ln_rows_load(batch);
            // (the rows' reduction chains interleaved: same bits.  What the prologue costs is VALU throughput, not latency: ~210
            //  instructions per row, 4 rows per wave, two waves per SIMD -- 2.5 us of a 7.7 us launch at 32 rows, measured by compiling
            //  the arithmetic out (5.2 us); the rows' loads are 0.4 us of it.  profiles/r06_qgemm_ln_ablation.txt)
            rowln_normalize_rows<NV, RB, false>(rl, gg, bb, K, q.eps, lane);
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int row = wave * RPW + batch * RB + u;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int c = (i * 64 + lane) * 4;
                    if (c < K) {
                        const int kt = c / KD, cc = c % KD;
                        const int ch = cc >> 3;
                        const unsigned off = (unsigned)(kt * BM * ROWB + row * ROWB + ((ch ^ (row & 15)) << 4) + ((cc >> 2) & 1) * 8);
                        const uint2 v = make_uint2(Half<T>::pack2(rl[u].v[i].x * q.ln_mul, rl[u].v[i].y * q.ln_mul),
                                                   Half<T>::pack2(rl[u].v[i].z * q.ln_mul, rl[u].v[i].w * q.ln_mul));
                        *reinterpret_cast<uint2_a*>(smem + off) = v;
                    }
                }
            }
        }
    }

    f32x4 acc[FM][FN];
    int par = 0;                                                 // LDS stage the next k-step is written to
    // one ring stage: registers -> LDS, the same registers re-armed with the stage D steps ahead (LOAD: unconditional, so that the
    // compiler's `s_waitcnt vmcnt` counts are exact -- a load under a branch makes every later wait assume it was not issued),
    // barrier, MFMAs
    auto stage = [&](qv4 (&reg)[CH], int kt, auto load_tag, int lti, int lkt, auto swap_tag) __attribute__((always_inline)) {
        constexpr bool SW = decltype(swap_tag)::value;
        char* const st = ring + par * STAGE;
#pragma unroll
        for (int j = 0; j < CH; ++j) *reinterpret_cast<qv4*>(st + (j * QNT + t) * 16) = reg[j];
        if constexpr (decltype(load_tag)::value) load_into(reg, lti, lkt);
        __syncthreads();          // stage visible; every wave is done with the OTHER slot's previous contents two steps back
        const char* ab = LN_A ? smem + kt * BM * ROWB : st;
        const char* wb = st + A_ROWS * ROWB;
#pragma unroll
        for (int ks = 0; ks < KD / 32; ++ks) {
            uint4 af[FM], wf[FN];
            const int co = ((4 * ks + g) ^ fr) << 4;             // (4 ks + g < 32; the XOR touches the low four bits only)
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int fm = wm + WGM * i;
                af[i] = *reinterpret_cast<const uint4_a*>(ab + ((fm < NFM ? fm : 0) * 16 + fr) * ROWB + co);
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) wf[j] = *reinterpret_cast<const uint4_a*>(wb + ((wn + WGN * j) * 16 + fr) * ROWB + co);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    if constexpr (SW) acc[i][j] = Half<T>::mfma16(wf[j], af[i], acc[i][j]);
                    else acc[i][j] = Half<T>::mfma16(af[i], wf[j], acc[i][j]);
                }
        }
        par ^= 1;
    };
    // D consecutive stages kt0 .. kt0 + D - 1 of this tile; their registers are re-armed with stages lkt0 .. of tile lti
    auto group = [&](int kt0, auto load_tag, int lti, int lkt0, auto swap_tag) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < D; ++j) stage(stg[j], kt0 + j, load_tag, lti, lkt0 + j, swap_tag);
    };
    auto ktile = [&](int ti, auto swap_tag) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        int kt0 = 0;
        for (; kt0 + D < nk; kt0 += D) group(kt0, std::true_type{}, ti, kt0 + D, swap_tag);
        if (ti + 1 < ntiles) group(kt0, std::true_type{}, ti + 1, 0, swap_tag);       // the next column tile's first stages
        else group(kt0, std::false_type{}, 0, 0, swap_tag);                           // the stream ends here
    };

    typename QRange<OutT>::type range;
    for (int ti = 0; ti < ntiles; ++ti) {
        const int n0 = (nt0 + ti) * BN;
        const bool vt_tile = EPI == EPI_QKV && n0 >= p.n_split;
        if (vt_tile) ktile(ti, std::false_type{}); else ktile(ti, std::true_type{});
        // ---------------- epilogue (the arithmetic of gemm_kernel / gemm256_epilogue.inc, bit for bit) ----------------
        if (vt_tile) {
            if constexpr (EPI == EPI_QKV) {
                // lane: m = .. + 4 g + r, n = .. + fr: four consecutive token rows of one V column -> V^T[n][m .. m+3]
                OutT* __restrict__ vt = static_cast<OutT*>(p.out2);
                const float csv = p.in_mul * p.out_mul2;
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int nl = (wn + WGN * j) * 16 + fr, n = n0 + nl;
                    float bn = 0.f;
                    if (p.bias != nullptr) bn = (LN_A ? reinterpret_cast<const float*>(smem + bias_off)[ti * BN + nl] : p.bias[n]) * p.out_mul2;
#pragma unroll
                    for (int i = 0; i < FM; ++i) {
                        const int fm = wm + WGM * i;
                        if (fm >= NFM) continue;
                        const int m = m0 + fm * 16 + 4 * g;
                        if (m >= p.M) continue;                  // (M % 4 == 0: the four rows are in or out together)
                        const float v0 = __builtin_fmaf(acc[i][j][0], csv, bn), v1 = __builtin_fmaf(acc[i][j][1], csv, bn);
                        const float v2 = __builtin_fmaf(acc[i][j][2], csv, bn), v3 = __builtin_fmaf(acc[i][j][3], csv, bn);
                        range.note(v0, v1); range.note(v2, v3);
                        q_store4<OutT>(vt + (long)(n - p.n_split) * p.ldo2 + m, v0, v1, v2, v3);
                    }
                }
            }
        } else {
            OutT* __restrict__ out = static_cast<OutT*>(p.out);
            const float om = p.out_mul;
            const float cs = HAS_BIAS ? p.in_mul : p.in_mul * om;
            const float bsc = EPI == EPI_BIAS_GELU ? 1.0f : om;
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int fm = wm + WGM * i;
                if (fm >= NFM) continue;
                const int m = m0 + fm * 16 + fr;
                if (m >= p.M) continue;
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int nl = (wn + WGN * j) * 16 + 4 * g, n = n0 + nl;
                    float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    float4 bb4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if constexpr (LN_A) {
                        if (HAS_BIAS || p.bias != nullptr) bb4 = *reinterpret_cast<const float4_a*>(smem + bias_off + (ti * BN + nl) * 4);
                    } else {
                        bb4 = bpre[j];
                    }
                    if constexpr (EPI == EPI_BIAS_RESID) {
                        const float4 rr = rpre[i][j];
                        v[0] = __builtin_fmaf(v[0], cs, bb4.x + rr.x); v[1] = __builtin_fmaf(v[1], cs, bb4.y + rr.y);
                        v[2] = __builtin_fmaf(v[2], cs, bb4.z + rr.z); v[3] = __builtin_fmaf(v[3], cs, bb4.w + rr.w);
                    } else if (EPI == EPI_BIAS_GELU || p.bias != nullptr) {
                        v[0] = __builtin_fmaf(v[0], cs, bb4.x * bsc); v[1] = __builtin_fmaf(v[1], cs, bb4.y * bsc);
                        v[2] = __builtin_fmaf(v[2], cs, bb4.z * bsc); v[3] = __builtin_fmaf(v[3], cs, bb4.w * bsc);
                    } else if constexpr (sizeof(OutT) == 2) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(v[r], cs, 0.f);
                    }
                    if constexpr (EPI == EPI_BIAS_GELU) {
                        const float ginv = 1.0f / om;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = gelu_new_fast_scaled(v[r], ginv);
                    }
                    range.note(v[0], v[1]); range.note(v[2], v[3]);
                    q_store4<OutT>(out + (long)m * p.ldo + n, v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
    range.finish(p.range_flag, p.range_amax);
}

// 16-byte chunks per thread and ring stage of a tile
constexpr int q_chunks(int bm, int bn, bool ln_a, int kd) { return ((ln_a ? 0 : bm) + bn) * kd * 2 / (QNT * 16); }
// THE depth rule (used by the kernel instantiations and by the launcher's feasibility check alike): stages in flight, as many as the
// registers allow -- 24 chunks per thread up to 4 chunks per stage, 18 / 16 above (128-row tiles carry 32-64 accumulators) -- in two
// classes: SIX for K / KD a multiple of 6 (or 3), FOUR for a multiple of 4 (or 2)
constexpr int q_depth_rule(int ch, bool six) { return ch >= 8 ? 2 : six ? (ch * 6 <= 24 ? 6 : 3) : (ch * 4 <= 16 ? 4 : 2); }
template <int BM, int BN, bool LN_A, int KD, bool SIX>
constexpr int q_depth() { return q_depth_rule(q_chunks(BM, BN, LN_A, KD), SIX); }

template <typename T, int EPI, typename OutT, int BM, int BN, bool LN_A, int NV, int KD, int D>
void qlaunch(const QGemmArgs& a, int group, hipStream_t s) {
    auto* kern = qgemm_kernel<T, EPI, OutT, BM, BN, LN_A, NV, KD, D>;
    // more than 64 KiB of dynamic LDS is a per-function, per-DEVICE opt-in: once per device this process launches on
    static std::atomic<unsigned> attr_done{0u};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned bit = 1u << (dev & 31);
    if (!(attr_done.load(std::memory_order_relaxed) & bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done.fetch_or(bit, std::memory_order_relaxed);
    }
    QGemmArgs b = a;
    b.group = group;
    const int nk = a.g.K / KD;
    const size_t lds = (size_t)(LN_A ? nk * BM * KD * 2 + group * BN * 4 : 0) + (size_t)2 * ((LN_A ? 0 : BM) + BN) * KD * 2;
    const int NT = a.g.N / BN, R = ((a.g.M + BM - 1) / BM) * ((NT + group - 1) / group);
    hipLaunchKernelGGL(kern, dim3(8 * ((R + 7) / 8)), dim3(QNT), lds, s, b);
}

inline int q_ncu() {
    static const int ncu = [] {
        int dev = 0, n = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n;
    }();
    return ncu;
}

// ---- tile choice (plain kernels): one launch = rounds x (BM + BN) K 2 bytes per workgroup; the fewest wins, ties to the larger tile ----
struct QTile { int bm, bn, kd; };
constexpr QTile Q_TILES[] = {{32, 16, 256}, {32, 32, 128}, {32, 64, 128}, {64, 32, 128}, {64, 64, 128}, {128, 64, 128}, {128, 128, 128}};
// (measured and dropped for the 32-row tiles, fc2 at 32 rows = 8.1 us per launch: 512- / 256-element stages 8.5; a software-pipelined
//  k-loop -- the MFMAs of stage s under stage s + 1's LDS fill and barrier, their first fragments fetched a stage ahead, three LDS
//  stages -- 8.0; all twelve stages in flight from the start (D = 12, 288 KiB per workgroup) 9.2.  Stamps: the first stage is
//  released 1.9 us after entry, then one 256-element stage (24 KiB) every 0.44 us = 55 GB/s per CU, half of what a free-running
//  stream reaches -- whatever the depth, the stage length or the LDS schedule.  profiles/r06_qgemm_tile_sweep.txt,
//  r06_qgemm_pipe_probe.txt, r06_qgemm_stage_stamps.txt)
constexpr int Q_NTILES = sizeof(Q_TILES) / sizeof(Q_TILES[0]);

// depth class of a tile at this K: 6 -> SIX kernels, 4 -> the others, 0 -> not served
inline int q_class(const QTile& tl, bool ln_a, int K) {
    if (K % tl.kd) return 0;
    const int nk = K / tl.kd;
    const int ch = q_chunks(tl.bm, tl.bn, ln_a, tl.kd);
    const int d6 = q_depth_rule(ch, true), d4 = q_depth_rule(ch, false);
    const bool ok6 = nk % d6 == 0, ok4 = nk % d4 == 0;
    if (ok6 && (!ok4 || d6 >= d4)) return 6;
    return ok4 ? 4 : 0;
}

inline int q_pick_plain(int M, int N, int K, int epi, int n_split, int force = 0) {
    const int ncu = q_ncu();
    int best = -1; long best_cost = 0;
    for (int i = 0; i < Q_NTILES; ++i) {
        const QTile& tl = Q_TILES[i];
        if (force > 0 && i != force - 1) continue;
        if (N % tl.bn || !q_class(tl, false, K)) continue;
        if (epi == EPI_QKV && (n_split % tl.bn || tl.bn < 16)) continue;
        if (tl.bm > 32 && M <= 32) continue;
        const long tiles = (long)((M + tl.bm - 1) / tl.bm) * (N / tl.bn);
        const long rounds = (tiles + ncu - 1) / ncu;
        const long cost = rounds * (tl.bm + tl.bn);
        if (best < 0 || cost < best_cost || (cost == best_cost && tl.bm * tl.bn > Q_TILES[best].bm * Q_TILES[best].bn)) { best = i; best_cost = cost; }
    }
    return best;
}

template <typename T, int EPI, typename OutT, int BM, int BN, int KD>
bool qrun_plain(const QGemmArgs& a, int cls, hipStream_t s) {
    if (cls == 6) qlaunch<T, EPI, OutT, BM, BN, false, 1, KD, q_depth<BM, BN, false, KD, true>()>(a, 1, s);
    else qlaunch<T, EPI, OutT, BM, BN, false, 1, KD, q_depth<BM, BN, false, KD, false>()>(a, 1, s);
    return true;
}

template <typename T, int EPI, typename OutT>
bool qdispatch_plain(const QGemmArgs& a, int epi, hipStream_t s) {
    const int i = q_pick_plain(a.g.M, a.g.N, a.g.K, epi, a.g.n_split, a.tile);
    if (i < 0) return false;
    const int cls = q_class(Q_TILES[i], false, a.g.K);
    switch (i) {
        case 0: return qrun_plain<T, EPI, OutT, 32, 16, 256>(a, cls, s);
        case 1: return qrun_plain<T, EPI, OutT, 32, 32, 128>(a, cls, s);
        case 2: return qrun_plain<T, EPI, OutT, 32, 64, 128>(a, cls, s);
        case 3: return qrun_plain<T, EPI, OutT, 64, 32, 128>(a, cls, s);
        case 4: return qrun_plain<T, EPI, OutT, 64, 64, 128>(a, cls, s);
        case 5: return qrun_plain<T, EPI, OutT, 128, 64, 128>(a, cls, s);
        case 6: return qrun_plain<T, EPI, OutT, 128, 128, 128>(a, cls, s);
    }
    return false;
}

// ---- LayerNorm-prologue kernels: (BM, BN) in {(32, 32), (32, 64), (64, 64)}, KD = 128; a workgroup walks `group` column tiles ----
struct QLnPlan { int bm, bn, group; };
inline bool q_plan_ln(int M, int N, int d, int epi, int n_split, QLnPlan* out, int force = 0) {
    if (d % 128 || N % 32) return false;
    const int nk = d / 128;
    if (nk % 6 != 0 && nk % 4 != 0) return false;
    const int ncu = q_ncu();
    QLnPlan best{0, 0, 0}; long best_cost = 0;
    const int cands[3][2] = {{32, 32}, {32, 64}, {64, 64}};
    for (int ci = 0; ci < 3; ++ci) {
        const int* c = cands[ci];
        if (force > 0 && ci != force - 1) continue;
        const int bm = c[0], bn = c[1];
        if (N % bn || (epi == EPI_QKV && n_split % bn)) continue;
        if (bm > 32 && M <= 32) continue;
        const int mt = (M + bm - 1) / bm, nt = N / bn;
        int gq = (int)(((long)mt * nt + ncu - 1) / ncu);
        gq = gq < 1 ? 1 : gq;
        while ((long)mt * ((nt + gq - 1) / gq) > ncu) ++gq;      // ONE round of the chip (44 x 6 = 264 workgroups ran two)
        const long lds = (long)nk * bm * 256 + 2l * bn * 256 + (long)gq * bn * 4;
        if (lds > 160 * 1024) continue;
        // launch time in ns, fitted to the tile sweep of scripts/micro/qgemm_probe.hip (profiles/r06_qgemm_tile_sweep.txt, d = 768):
        // a fixed part (LayerNorm of the tile's rows + pipeline fill: 6.2 us at 32 rows, 9.4 at 64 -- two batches of rows) + the
        // serial chain of barrier-locked stages, one per 128 k-elements and column tile (0.25 / 0.35 / 0.44 us by tile)
        const long cost = (bm == 32 ? 6200 : 9400) + (long)gq * nk * (bn == 32 ? 250 : bm == 32 ? 350 : 440);
        if (best.bm == 0 || cost < best_cost) { best = QLnPlan{bm, bn, gq}; best_cost = cost; }
    }
    if (best.bm == 0) return false;
    *out = best;
    return true;
}

template <typename T, int EPI, typename OutT, int NV, bool SIX>
bool qdispatch_ln(const QGemmArgs& a, const QLnPlan& pl, hipStream_t s) {
    if (pl.bm == 32 && pl.bn == 32) qlaunch<T, EPI, OutT, 32, 32, true, NV, 128, q_depth<32, 32, true, 128, SIX>()>(a, pl.group, s);
    else if (pl.bm == 32 && pl.bn == 64) qlaunch<T, EPI, OutT, 32, 64, true, NV, 128, q_depth<32, 64, true, 128, SIX>()>(a, pl.group, s);
    else if (pl.bm == 64 && pl.bn == 64) qlaunch<T, EPI, OutT, 64, 64, true, NV, 128, q_depth<64, 64, true, 128, SIX>()>(a, pl.group, s);
    else return false;
    return true;
}

template <typename T>
bool qgemm16(int epi, int out_dtype, const QGemmArgs& a, hipStream_t s) {
    const bool o16 = dt_is16(out_dtype);
    if (a.x != nullptr) {           // LayerNorm prologue (K = d)
        const int d = a.g.K;
        QLnPlan pl;
        if (!o16 || !q_plan_ln(a.g.M, a.g.N, d, epi, a.g.n_split, &pl, a.tile)) return false;
        const bool six = (d / 128) % 6 == 0;
#define QLN(NVV)                                                                                                          \
        do {                                                                                                               \
            if (epi == EPI_QKV) return six ? qdispatch_ln<T, EPI_QKV, T, NVV, true>(a, pl, s) : qdispatch_ln<T, EPI_QKV, T, NVV, false>(a, pl, s);   \
            if (epi == EPI_BIAS_GELU) return six ? qdispatch_ln<T, EPI_BIAS_GELU, T, NVV, true>(a, pl, s) : qdispatch_ln<T, EPI_BIAS_GELU, T, NVV, false>(a, pl, s); \
            return false;                                                                                                  \
        } while (0)
        if (d == 768) QLN(3);
        if (d == 1024 || d == 512) QLN(4);          // (d = 512: NV = 2 would do; one instantiation less)
        return false;
#undef QLN
    }
    if (epi == EPI_STORE && o16) return qdispatch_plain<T, EPI_STORE, T>(a, epi, s);
    if (epi == EPI_STORE && !o16) return qdispatch_plain<T, EPI_STORE, float>(a, epi, s);
    if (epi == EPI_QKV && o16) return qdispatch_plain<T, EPI_QKV, T>(a, epi, s);
    if (epi == EPI_BIAS_GELU && o16) return qdispatch_plain<T, EPI_BIAS_GELU, T>(a, epi, s);
    if (epi == EPI_BIAS_RESID && !o16) return qdispatch_plain<T, EPI_BIAS_RESID, float>(a, epi, s);
    return false;
}

}  // namespace

bool qgemm_shape_ok(int M, int N, int K, int epi, int n_split) {
    if (M <= 0 || M > QGEMM_MAX_ROWS || M % 32 || K < 128 || N % 16) return false;
    if (epi == EPI_QKV && (n_split % 32 || n_split <= 0 || n_split >= N || N % 32)) return false;
    if (!(epi == EPI_STORE || epi == EPI_QKV || epi == EPI_BIAS_GELU || epi == EPI_BIAS_RESID)) return false;
    return q_pick_plain(M, N, K, epi, n_split) >= 0;
}

bool qgemm_ln_ok(int M, int N, int d, int epi, int n_split) {
    if (M <= 0 || M > QGEMM_MAX_ROWS || M % 32 || (epi != EPI_QKV && epi != EPI_BIAS_GELU)) return false;
    if (epi == EPI_QKV && (n_split % 32 || n_split <= 0 || n_split >= N)) return false;
    if (d != 768 && d != 1024 && d != 512) return false;
    QLnPlan pl;
    return q_plan_ln(M, N, d, epi, n_split, &pl);
}

bool launch_qgemm(int dtype, int epi, int out_dtype, const QGemmArgs& a0, hipStream_t s) {
    QGemmArgs a = a0;
    if (a.g.in_mul == 0.f) a.g.in_mul = 1.f;
    if (a.g.out_mul == 0.f) a.g.out_mul = 1.f;
    if (a.g.out_mul2 == 0.f) a.g.out_mul2 = 1.f;
    if (a.ln_mul == 0.f) a.ln_mul = 1.f;
    if (a.g.lo_delta != 0 || a.g.lo_delta2 != 0 || a.g.hi2_delta != 0 || a.g.m_valid != a.g.M || a.g.pred != nullptr) return false;
    if (a.x != nullptr ? !qgemm_ln_ok(a.g.M, a.g.N, a.g.K, epi, a.g.n_split) : !qgemm_shape_ok(a.g.M, a.g.N, a.g.K, epi, a.g.n_split)) return false;
    if (dtype == DT_BF16) return qgemm16<bf16_t>(epi, out_dtype, a, s);
    if (dtype == DT_F16) return qgemm16<f16_t>(epi, out_dtype, a, s);
    return false;
}
