// Query-sized projections: C = A W^T for batches of at most a few hundred token rows -- one query
// (`SentenceTransformer.encode("one query")`: SentenceTransformer.py:143-146, README.md:309-330), USEB's 21-sentence batches
// (useb/useb/useb/evaluators/askubuntu.py:144-148), a rank's eighth of a query set.  (Round 6.)
//
// What bounds these launches is not arithmetic (a 16-query fc2 is 1.6 GFLOP) but the serial chain of one tile: the
// register-staged kernel of gemm.hip walks load -> LDS -> MFMA round trips of ~1.3 us with two k-steps in flight, so a
// K = 768 tile costs ~8 us and a K = 3072 tile ~20 us whatever the row count (profiles/r06_query_kernel_stats_before.csv).
// This kernel puts the WHOLE reach of the ring in flight before the first MFMA:
//   * operands reach LDS by `global_load_lds_dwordx4` (no registers held): a ring of up to 8 stages of 128 k-elements, every
//     stage issued up front, refilled one stage per k-step behind a counted `s_waitcnt vmcnt` -- 64-128 KiB in flight per CU,
//     which is what an L2 / MALL stream needs to run at the CU's ~130 GB/s;
//   * small tiles (32 rows x 16 / 32 / 64 columns) so that even one query's launch covers 24-96 CUs, tile list nt-major in
//     contiguous per-XCD runs so that a weight panel is fetched into one L2;
//   * LN_A: the projection normalises its own A rows.  At K = d the rows a workgroup needs ARE whole rows of the residual stream:
//     the prologue runs the one-wave-per-row LayerNorm of elementwise.hip (rowln.h: same loads, reductions, arithmetic, same
//     bits) on the tile's 32 rows and writes the 16-bit panel straight into LDS; the LayerNorm launch, its round trip through
//     HBM and its boundary are gone (LN1 -> QKV and LN2 -> fc1: two of seven launches per block);
//   * every output element still receives the k-ascending chain of v_mfma_f32_16x16x32 and the epilogue arithmetic of the
//     other GEMM kernels: a sequence's embedding keeps its bits whichever kernel its batch size selects.
// LDS image of a stage: rows of 256 B (16 chunks of 16 B), chunk c of row r at position c ^ (r & 15): the four 16-lane
// groups of a ds_read_b128 fragment read (rows fr, chunks 4 ks + g) cover all 64 banks once (scripts/lds_layout_check.py).
// With LDS-DMA the image is lane-linear, so the XOR is applied to each lane's SOURCE chunk.
#include <type_traits>

#include "common.h"
#include "rowln.h"

namespace {

typedef __attribute__((address_space(3))) char* lds_cptr_t;
constexpr int QNW = 8;       // waves per workgroup
constexpr int QKD = 128;     // k-elements per ring stage (LayerNorm-prologue kernels; the plain ones: 128 | 256)
constexpr int QROWB = 256;   // bytes per staged row

__device__ __forceinline__ void q_dma16(const char* base_uniform, unsigned lane_off, unsigned dst_byte) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(lane_off), "s"(base_uniform), "s"(dst_byte) : "memory");
}

// everything but the newest k * PW LDS-DMA pieces of this wave has landed (k = stages still in flight behind the one needed,
// wave-uniform; PW = pieces per wave and stage): `s_waitcnt vmcnt` takes an immediate, hence the switch
template <int PW>
__device__ __forceinline__ void q_wait_stages(int k) {
    constexpr int KMAX = 63 / PW < 7 ? 63 / PW : 7;      // vmcnt is a 6-bit counter; a smaller count waits for more, never for less
    k = k > KMAX ? KMAX : k;
#define QW(i) case i: asm volatile("s_waitcnt vmcnt(%0)" : : "n"((i <= KMAX ? i : KMAX) * PW) : "memory"); break;
    switch (k) { QW(0) QW(1) QW(2) QW(3) QW(4) QW(5) QW(6) default: asm volatile("s_waitcnt vmcnt(%0)" : : "n"(KMAX * PW) : "memory"); }
#undef QW
}

template <typename OutT> __device__ __forceinline__ void q_store4(OutT* p, float a, float b, float c, float d) {
    if constexpr (sizeof(OutT) == 4) *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
    else *reinterpret_cast<uint2*>(p) = make_uint2(Half<OutT>::pack2(a, b), Half<OutT>::pack2(c, d));
}
template <typename OutT> struct QRange { typedef RangeTrack<bf16_t> type; };
template <> struct QRange<f16_t> { typedef RangeTrack<f16_t> type; };

// s_memtime stamps of workgroup 0 (every wave's lane 0 -> dbg[wave * 16 + k]): probe build only (scripts/micro/qgemm_probe.hip)
#ifdef SGPT_QSTAMPS
#define QSTAMP(k) do { if (p.dbg != nullptr && blockIdx.x == 0 && lane == 0) p.dbg[wave * 16 + (k)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define QSTAMP(k) (void)0
#endif

// EPI: EPI_STORE (16-bit or fp32 out, optional bias) | EPI_QKV (q | k row-major, V^T through the role-swapped MFMA orientation)
//      | EPI_BIAS_GELU | EPI_BIAS_RESID.   NV = float4 loads per lane of a LayerNorm row (LN_A; d <= 256 NV).
// KD = k-elements per ring stage (128 | 256: rows of 256 / 512 B; the XOR stays inside a row's 256-byte halves).
// LN_A workgroups keep their normalised A panel and walk `q.group` consecutive column tiles, the weight ring running across the
// tile boundaries (one LayerNorm and one pipeline fill per workgroup, and a launch of 11 x 48 tiles fits one round of the chip).
template <typename T, int EPI, typename OutT, int BM, int BN, bool LN_A, int NV, int KD>
__global__ __launch_bounds__(64 * QNW) void qgemm_kernel(const QGemmArgs q) {
    const GemmArgs& p = q.g;
    constexpr int ROWB = KD * 2, CPR = KD / 8, RPP = 64 / CPR;  // bytes per staged row, 16-B chunks per row, rows per 1-KiB DMA piece
    constexpr int NFM = BM / 16, NFN = BN / 16;
    constexpr int WGN = NFN >= 4 ? 4 : NFN, WGM = QNW / WGN;    // waves along N / M
    constexpr int FN = NFN / WGN, FM = (NFM + WGM - 1) / WGM;   // fragments per wave
    static_assert(NFN % WGN == 0 && (NFM % WGM == 0 || NFM < WGM), "tile / wave grid");
    constexpr int A_STAGE = LN_A ? 0 : BM * ROWB;
    constexpr int STAGE = A_STAGE + BN * ROWB;
    // Wave roles.  A tile of NFM x NFN fragments keeps NACT = min(QNW, NFM NFN) waves busy with MFMAs (waves 0 .. NACT-1); with at
    // least four waves left over, THOSE issue every DMA piece (the price of a piece -- 60-185 cycles of issue -- then sits beside the
    // MFMA chain instead of inside it); otherwise all waves share the pieces.
    constexpr int NACT = NFM * NFN < QNW ? NFM * NFN : QNW;
    constexpr bool SPEC = QNW - NACT >= 4;
    constexpr int NL = SPEC ? QNW - NACT : QNW, L0 = SPEC ? NACT : 0;   // loader waves: L0 .. L0 + NL - 1
    constexpr int NPIECE = STAGE / 1024, PW = NPIECE / NL;       // 1-KiB DMA pieces per stage / per loader wave
    static_assert(NPIECE % NL == 0 && PW >= 1, "pieces per stage must split evenly over the loader waves");
    constexpr bool HAS_BIAS = EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RESID;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int K = p.K, nk = K / KD, ns = q.ns;
    const int G = LN_A ? q.group : 1;
    const int MT = p.M / BM, NT = p.N / BN, NG = (NT + G - 1) / G;
    // workgroup list column-group-major (the workgroups that share weight panels are neighbours), XCD x = blocks x, x + 8, ... takes a
    // contiguous run
    const int R = MT * NG, c0 = R >> 3, rem = R & 7;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    if (local >= c0 + (xcd < rem ? 1 : 0)) return;
    const int gi = xcd * c0 + (xcd < rem ? xcd : rem) + local;
    const int ng = gi / MT, mt = gi - ng * MT;
    const int m0 = mt * BM, nt0 = ng * G;
    const int ntiles = NT - nt0 < G ? NT - nt0 : G;
    const int S = ntiles * nk;                                   // ring stages this workgroup consumes

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int wn = wave % WGN, wm = wave / WGN;
    const bool loader = wave >= L0;
    const int lw = loader ? wave - L0 : 0;                       // index among the loader waves

    const T* __restrict__ Ag = static_cast<const T*>(p.A);
    const T* __restrict__ Wg = static_cast<const T*>(p.W);
    const unsigned lds_base = (unsigned)(size_t)(lds_cptr_t)smem;
    const int panel_bytes = LN_A ? nk * BM * ROWB : 0;           // LN_A: [nk][BM rows][ROWB]
    const unsigned ring0 = lds_base + (unsigned)panel_bytes;
    const int bias_off = panel_bytes + ns * STAGE;               // LN_A: fp32 bias of the workgroup's column tiles [G][BN]

    // this wave's DMA pieces of a stage: piece pi = lw + NL j covers RPP rows of one operand
    int prow[PW];
    bool pis_a[PW];
    unsigned poff[PW], pdst[PW];
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        const int pi = lw + NL * j;
        pis_a[j] = !LN_A && pi < BM / RPP;
        prow[j] = pis_a[j] ? pi * RPP : (pi - (LN_A ? 0 : BM / RPP)) * RPP;
        const long ld = pis_a[j] ? p.lda : p.ldw;
        const int r = lane / CPR, pos = lane % CPR;
        const int chunk = pos ^ ((prow[j] + r) & 15);            // XOR on the low four chunk bits: stays in the 256-byte half
        poff[j] = (unsigned)((r * ld + chunk * 8) * 2);
        pdst[j] = (unsigned)((pis_a[j] ? 0 : A_STAGE) + prow[j] * ROWB);
    }
    int iss = 0, iss_ti = 0, iss_kt = 0;                         // next stage to issue: its tile and k-step
    auto issue_next = [&]() {
        const unsigned dst = ring0 + (unsigned)((iss % ns) * STAGE);
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            if (!loader) break;
            const T* src = pis_a[j] ? Ag + (long)(m0 + prow[j]) * p.lda : Wg + (long)((nt0 + iss_ti) * BN + prow[j]) * p.ldw;
            q_dma16(reinterpret_cast<const char*>(src) + (long)iss_kt * ROWB, poff[j], dst + pdst[j]);
        }
        ++iss;
        if (++iss_kt == nk) { iss_kt = 0; ++iss_ti; }
    };
    if constexpr (LN_A) {
        // the column tiles' bias through the same queue (no compiler-counted load inside the tile loop), ahead of every stage:
        // whichever stage wait comes first covers it
        if (p.bias != nullptr && wave == L0 && lane * 4 < ntiles * BN)
            q_dma16(reinterpret_cast<const char*>(p.bias + nt0 * BN), (unsigned)(lane * 16), lds_base + bias_off);
    }
    QSTAMP(0);
    // ring of ns stages; when everything fits (ns >= S) all of it is issued here and nothing is refilled
    const int ns_pro = ns >= S ? S : ns - 1;
    for (int i = 0; i < ns_pro; ++i) issue_next();
    QSTAMP(1);

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
                    rpre[i][j] = *reinterpret_cast<const float4*>(p.resid + (long)m * p.ldo + n);
                }
            }
        }
    }

    if constexpr (LN_A) {
        // LayerNorm of the tile's BM rows, one wave per row (rowln.h), 16-bit result into the A panel.  Every operand -- the rows,
        // gamma, beta -- is requested before the first use: one memory round trip for the whole prologue.
        constexpr int RPW = BM / QNW;                    // rows per wave
        const int d = K;
        float4 gg[NV], bb[NV];
        RowLN<NV, false> rl[RPW];
#pragma unroll
        for (int u = 0; u < RPW; ++u) rl[u].load(q.x + (long)(m0 + wave * RPW + u) * d, d, lane);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            gg[i] = c < d ? *reinterpret_cast<const float4*>(q.ln_g + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            bb[i] = c < d ? *reinterpret_cast<const float4*>(q.ln_b + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < RPW; ++u) {
            rl[u].normalize_pre(gg, bb, d, q.eps, lane);
            const int row = wave * RPW + u;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (i * 64 + lane) * 4;
                if (c < d) {
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

    QSTAMP(2);
    f32x4 acc[FM][FN];
    int gs = 0;                                                  // ring stage about to be consumed
    auto ktile = [&](auto swap_tag) {
        constexpr bool SW = decltype(swap_tag)::value;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nk; ++kt, ++gs) {
            if (loader) q_wait_stages<PW>(iss - gs - 1);         // stages in flight behind this one
            __syncthreads();
            if (gs < 8) QSTAMP(8 + gs);
            if (iss < S) issue_next();                           // refills the slot consumed one step ago
            const char* st = smem + panel_bytes + (gs % ns) * STAGE;
            const char* ab = LN_A ? smem + kt * BM * ROWB : st;
            const char* wb = st + A_STAGE;
#pragma unroll
            for (int ks = 0; ks < KD / 32; ++ks) {
                uint4 af[FM], wf[FN];
                const int co = ((4 * ks + g) ^ fr) << 4;         // (4 ks + g < 32; the XOR touches the low four bits only)
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
        }
    };

    typename QRange<OutT>::type range;
    for (int ti = 0; ti < ntiles; ++ti) {
        const int n0 = (nt0 + ti) * BN;
        const bool vt_tile = EPI == EPI_QKV && n0 >= p.n_split;
        if (vt_tile) ktile(std::false_type{}); else ktile(std::true_type{});
        if (ti == 0) QSTAMP(3);
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
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int nl = (wn + WGN * j) * 16 + 4 * g, n = n0 + nl;
                    float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
                    if constexpr (LN_A) {
                        if (HAS_BIAS || p.bias != nullptr) bb = *reinterpret_cast<const float4_a*>(smem + bias_off + (ti * BN + nl) * 4);
                    } else {
                        bb = bpre[j];
                    }
                    if constexpr (EPI == EPI_BIAS_RESID) {
                        const float4 rr = rpre[i][j];
                        v[0] = __builtin_fmaf(v[0], cs, bb.x + rr.x); v[1] = __builtin_fmaf(v[1], cs, bb.y + rr.y);
                        v[2] = __builtin_fmaf(v[2], cs, bb.z + rr.z); v[3] = __builtin_fmaf(v[3], cs, bb.w + rr.w);
                    } else if (EPI == EPI_BIAS_GELU || p.bias != nullptr) {
                        v[0] = __builtin_fmaf(v[0], cs, bb.x * bsc); v[1] = __builtin_fmaf(v[1], cs, bb.y * bsc);
                        v[2] = __builtin_fmaf(v[2], cs, bb.z * bsc); v[3] = __builtin_fmaf(v[3], cs, bb.w * bsc);
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
    QSTAMP(4);
    range.finish(p.range_flag, p.range_amax);
#undef QSTAMP
}

template <typename T, int EPI, typename OutT, int BM, int BN, bool LN_A, int NV, int KD>
void qlaunch(const QGemmArgs& a, int ns, int group, hipStream_t s) {
    auto* kern = qgemm_kernel<T, EPI, OutT, BM, BN, LN_A, NV, KD>;
    static const bool attr_set = [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
    }();
    (void)attr_set;
    QGemmArgs b = a;
    b.ns = ns; b.group = group;
    const int nk = a.g.K / KD;
    const size_t lds = (size_t)(LN_A ? nk * BM * KD * 2 + group * BN * 4 : 0) + (size_t)ns * ((LN_A ? 0 : BM * KD * 2) + BN * KD * 2);
    const int NT = a.g.N / BN, R = (a.g.M / BM) * ((NT + group - 1) / group);
    hipLaunchKernelGGL(kern, dim3(8 * ((R + 7) / 8)), dim3(64 * QNW), lds, s, b);
}

// ring depth: everything when it fits `budget` bytes of LDS, else as many stages as fit (at most 8, at least 3; 0 = does not fit)
inline int ring_stages(int total_stages, int stage_bytes, int budget) {
    int ns = budget / stage_bytes;
    ns = ns > 8 ? 8 : ns;
    if (ns >= total_stages) return total_stages;
    return ns < 3 ? 0 : ns;
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

template <typename T, int EPI, typename OutT>
bool qdispatch_plain(const QGemmArgs& a, hipStream_t s) {
    const int M = a.g.M, N = a.g.N, K = a.g.K;
    const int mt = M / 32, ncu = q_ncu();
    // BN_: column tile.  A launch that fits one workgroup per CU may take the whole LDS for its ring (and, at long K, 256-element
    // stages: half the barriers and waits of a tile's serial chain); a larger one keeps two workgroups per CU resident.
#define QPLAIN(BN_)                                                                                                  \
    do {                                                                                                             \
        const long tiles = (long)mt * (N / BN_);                                                                     \
        const int budget = tiles <= ncu ? 160 * 1024 : 80 * 1024;                                                    \
        if (K % 256 == 0 && K >= 2048 && tiles <= ncu) {                                                             \
            const int ns = ring_stages(K / 256, (32 + BN_) * 512, budget);                                           \
            if (ns) { qlaunch<T, EPI, OutT, 32, BN_, false, 1, 256>(a, ns, 1, s); return true; }                     \
        }                                                                                                            \
        const int ns = ring_stages(K / 128, (32 + BN_) * 256, budget);                                               \
        if (ns) { qlaunch<T, EPI, OutT, 32, BN_, false, 1, 128>(a, ns, 1, s); return true; }                         \
    } while (0)
    // column tile: as narrow as it takes to cover the chip's CUs with one round of 32-row tiles
    if (N % 64 == 0 && (long)mt * (N / 64) >= 192) QPLAIN(64);
    if (N % 32 == 0 && ((long)mt * (N / 32) >= 96 || N % 16 != 0)) QPLAIN(32);
    if (N % 16 == 0) QPLAIN(16);
#undef QPLAIN
    return false;
}

template <typename T, int EPI, typename OutT, int NV>
bool qdispatch_ln(const QGemmArgs& a, hipStream_t s) {
    const int M = a.g.M, N = a.g.N, nk = a.g.K / QKD;
    const int panel = nk * 32 * QROWB;
    const int mt = M / 32;
    const int ncu = q_ncu();
    // column tiles per workgroup: as few as keep the launch inside one round of the chip (one workgroup per CU: the panel and the
    // ring take most of the LDS)
    auto group_of = [&](int nt) { int gq = (mt * nt + ncu - 1) / ncu; return gq < 1 ? 1 : gq; };
    if (N % 64 == 0 && (EPI != EPI_QKV || a.g.n_split % 64 == 0) && (long)mt * (N / 64) >= 128) {
        const int gq = group_of(N / 64);
        const int ns = ring_stages(gq * nk, 64 * QROWB, 160 * 1024 - panel - gq * 64 * 4);
        if (ns) { qlaunch<T, EPI, OutT, 32, 64, true, NV, QKD>(a, ns, gq, s); return true; }
    }
    if (N % 32 == 0 && (EPI != EPI_QKV || a.g.n_split % 32 == 0)) {
        const int gq = group_of(N / 32);
        const int ns = ring_stages(gq * nk, 32 * QROWB, 160 * 1024 - panel - gq * 32 * 4);
        if (ns) { qlaunch<T, EPI, OutT, 32, 32, true, NV, QKD>(a, ns, gq, s); return true; }
    }
    return false;
}

template <typename T>
bool qgemm16(int epi, int out_dtype, const QGemmArgs& a, hipStream_t s) {
    const bool o16 = dt_is16(out_dtype);
    if (a.x != nullptr) {           // LayerNorm prologue (K = d)
        const int d = a.g.K;
        if (d % 256 != 0 && d != 128) return false;
#define QLN(NVV)                                                                                         \
        do {                                                                                              \
            if (epi == EPI_QKV && o16) return qdispatch_ln<T, EPI_QKV, T, NVV>(a, s);                      \
            if (epi == EPI_BIAS_GELU && o16) return qdispatch_ln<T, EPI_BIAS_GELU, T, NVV>(a, s);          \
            return false;                                                                                 \
        } while (0)
        if (d <= 256) QLN(1);
        if (d <= 512) QLN(2);
        if (d <= 768) QLN(3);
        if (d <= 1024) QLN(4);
        return false;
#undef QLN
    }
    if (epi == EPI_STORE && o16) return qdispatch_plain<T, EPI_STORE, T>(a, s);
    if (epi == EPI_STORE && !o16) return qdispatch_plain<T, EPI_STORE, float>(a, s);
    if (epi == EPI_QKV && o16) return qdispatch_plain<T, EPI_QKV, T>(a, s);
    if (epi == EPI_BIAS_GELU && o16) return qdispatch_plain<T, EPI_BIAS_GELU, T>(a, s);
    if (epi == EPI_BIAS_RESID && !o16) return qdispatch_plain<T, EPI_BIAS_RESID, float>(a, s);
    return false;
}

}  // namespace

bool qgemm_shape_ok(int M, int N, int K, int epi, int n_split) {
    if (M <= 0 || M > QGEMM_MAX_ROWS || M % 32 || K % QKD || K < QKD || N % 16) return false;
    if (epi == EPI_QKV && (n_split % 32 || n_split <= 0 || n_split >= N || N % 32)) return false;
    return epi == EPI_STORE || epi == EPI_QKV || epi == EPI_BIAS_GELU || epi == EPI_BIAS_RESID;
}

bool qgemm_ln_ok(int M, int N, int d, int epi, int n_split) {
    if (!qgemm_shape_ok(M, N, d, epi, n_split) || (epi != EPI_QKV && epi != EPI_BIAS_GELU)) return false;
    if (d > 1024 || (d % 256 != 0 && d != 128) || N % 32) return false;
    return (d / QKD) * 32 * QROWB + 3 * 32 * QROWB + 1024 <= 160 * 1024;
}

bool launch_qgemm(int dtype, int epi, int out_dtype, const QGemmArgs& a0, hipStream_t s) {
    QGemmArgs a = a0;
    if (a.g.in_mul == 0.f) a.g.in_mul = 1.f;
    if (a.g.out_mul == 0.f) a.g.out_mul = 1.f;
    if (a.g.out_mul2 == 0.f) a.g.out_mul2 = 1.f;
    if (a.ln_mul == 0.f) a.ln_mul = 1.f;
    if (a.g.lo_delta != 0 || a.g.lo_delta2 != 0 || a.g.hi2_delta != 0 || a.g.m_valid != a.g.M || a.g.pred != nullptr) return false;
    if (!qgemm_shape_ok(a.g.M, a.g.N, a.g.K, epi, a.g.n_split)) return false;
    if (a.x != nullptr && !qgemm_ln_ok(a.g.M, a.g.N, a.g.K, epi, a.g.n_split)) return false;
    if (dtype == DT_BF16) return qgemm16<bf16_t>(epi, out_dtype, a, s);
    if (dtype == DT_F16) return qgemm16<f16_t>(epi, out_dtype, a, s);
    return false;
}
