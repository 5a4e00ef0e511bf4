// Per-query top-k selection with running merge: torch.topk(k, largest, sorted=False) of
// custommodels/exact_search.py:102-108 plus the heapq.nlargest merge of :121-132.
//
// One 256-thread workgroup per query scans a VIRTUAL row = [chunk scores (n) | previous best
// (n_prev)].  Output: the k best, sorted by descending score, ties by ascending index
// (deterministic), unused tail (-inf, -1).
//
//   fast path (k <= 128, row >= 4096): ONE pass.  A strided 1024-element sample is bitonic-sorted
//     in LDS; its k-th largest value tau is a lower bound of the row's k-th largest, so every
//     element >= tau (expected k*row/1024 of them) is appended to an LDS candidate list (rare LDS
//     atomics), and the k best are the head of the sorted candidate list.  Falls back to the radix
//     path if the list overflows (adversarial order / massive ties).
//   radix path: 4 passes of 8 bits (MSB first) over the order-preserving uint32 image of the fp32
//     score find the exact k-th largest key.  Scores cluster in a few exponent/mantissa bins, so
//     histogram updates are wave-aggregated (ballot-peel the 4 most common digits of the wave, one
//     LDS atomic each; only the stragglers issue their own).  A gather pass collects the elements
//     above the threshold; ties AT the threshold are taken in ascending position by a parallel
//     count + prefix-sum pass.
#include <cstdlib>

#include "common.h"

namespace {

__device__ __forceinline__ uint32_t f2key(float f) {  // ascending uint order == ascending float order
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

constexpr int TK_THREADS = 256;
constexpr int TK_SORT_MAX = 2048;   // k <= 2048 sorted in LDS (the driver's k+1 = 1001 fits)
constexpr int TK_FAST_KMAX = 64;
constexpr int TK_MERGE_KMAX = 1024;  // cand_merge_kernel: running top-k kept in LDS
constexpr long TK_FAST_MIN_ROW = 4096;

struct Row {
    const float* sc; long n; long idx_base;
    const float* pv; const int64_t* pi; int n_prev; int nan_to_m1; int64_t excl;
    // gk > 0 ("gathered" previous leg, round 5): pv / pi point at a [world][nq][gk] stack of per-rank lists as ncclAllGather
    // delivers them (NOT at this row); entry j of the row is list j / gk, column j % gk -- the rank-major -> row-major permute of
    // the cross-rank fold done by the select's own loads instead of a kernel of its own
    int gk; long gstride;           // gstride = nq * gk elements between two ranks' lists
    __device__ __forceinline__ long total() const { return n + n_prev; }
    __device__ __forceinline__ long paddr(long j) const { return gk > 0 ? (j / gk) * gstride + (j % gk) : j; }
    // Both legs are always addressed in-bounds (clamped index, never-null base): hipcc may hoist /
    // speculate these loads out of their guards (seen: pi[i-n] issued for i < n with pi == nullptr).
    __device__ __forceinline__ float val(long i) const {
        const bool isp = i >= n;
        const long j = isp ? i - n : 0;
        float v;
        if (isp) {
            const long a = paddr(j);
            const int64_t id = pi[a];
            v = (id < 0 || id == excl) ? -INFINITY : pv[a];   // empty slot / excluded (self) id
        } else {
            v = sc[i];
        }
        if (nan_to_m1 && v != v) v = -1.0f;                   // exact_search.py:99
        return v;
    }
    __device__ __forceinline__ int64_t idx(long i) const {
        if (i < n) return idx_base + i;
        const int64_t id = pi[paddr(i - n)];
        return id == excl ? -1 : id;
    }
};

// The order of every path of this file (round 6: ONE order -- ADVICE r05).  Higher score first, ties -> lower index.  It is
// taken on the order-preserving integer image of the score, so it is total over NaNs too: a positive NaN ranks above +inf and a
// negative one below -inf (torch.topk's convention for the former), exactly as the register-key rounds (wave_topk_keys) and
// the radix path rank them; -0.0 is folded into +0.0 (equal scores tie on the index).  For every other pair of floats this is
// `va > vb || (va == vb && ia < ib)`.
__device__ __forceinline__ bool sorts_before(float va, int64_t ia, float vb, int64_t ib) {
    const uint32_t ka = f2key(va + 0.0f), kb = f2key(vb + 0.0f);
    return ka > kb || (ka == kb && ia < ib);
}

// Wave-wide arg-max of (value, index, position) triples under sorts_before (ties -> lower index; a total order, so the result does
// not depend on the pairing), every lane ending with the winner.  Round 5: the butterfly runs on the VALU -- DPP quad permutes
// (lane ^ 1, lane ^ 2), row_half_mirror / row_mirror (the other four lanes, the other eight of a row of 16) and gfx950's
// v_permlane16_swap / v_permlane32_swap (the other rows) -- instead of `__shfl_xor`, i.e. ds_bpermute round trips through the LDS
// queue: four dependent ~100-cycle trips per step, 6 steps, k rounds on ONE wave were 16 of the 20 us of the sample's select and
// most of cand_merge's 18 us.  `valid` (pos >= 0) lets empty lanes lose every comparison.
template <int CTRL>
__device__ __forceinline__ void argmax_step_dpp(float& bv, int64_t& bi, int& bp) {
    const float ov = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(bv), CTRL, 0xf, 0xf, false));
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)((unsigned long long)bi & 0xffffffffull), CTRL, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)((unsigned long long)bi >> 32), CTRL, 0xf, 0xf, false);
    const int op = __builtin_amdgcn_update_dpp(0, bp, CTRL, 0xf, 0xf, false);
    const int64_t oi = (int64_t)(((unsigned long long)hi << 32) | lo);
    if (op >= 0 && (bp < 0 || sorts_before(ov, oi, bv, bi))) { bv = ov; bi = oi; bp = op; }
}
template <bool ROW32>
__device__ __forceinline__ void argmax_step_rows(float& bv, int64_t& bi, int& bp) {
    auto other = [](unsigned x) -> unsigned {
        // the swap hands every lane the partner row's value in ONE of the two results (the other one is its own): pick the one that differs
        // in position by construction -- results r[0] / r[1] hold (own, partner) for one half of the lanes and (partner, own) for the other
        const auto r = ROW32 ? __builtin_amdgcn_permlane32_swap(x, x, false, false) : __builtin_amdgcn_permlane16_swap(x, x, false, false);
        const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        const bool upper = ROW32 ? (lane & 32u) != 0 : (lane & 16u) != 0;
        return upper ? r[0] : r[1];
    };
    const float ov = __uint_as_float(other(__float_as_uint(bv)));
    const unsigned lo = other((unsigned)((unsigned long long)bi & 0xffffffffull));
    const unsigned hi = other((unsigned)((unsigned long long)bi >> 32));
    const int op = (int)other((unsigned)bp);
    const int64_t oi = (int64_t)(((unsigned long long)hi << 32) | lo);
    if (op >= 0 && (bp < 0 || sorts_before(ov, oi, bv, bi))) { bv = ov; bi = oi; bp = op; }
}
__device__ __forceinline__ void wave_argmax(float& bv, int64_t& bi, int& bp) {
    argmax_step_dpp<0xB1>(bv, bi, bp);      // quad_perm [1,0,3,2]: lane ^ 1
    argmax_step_dpp<0x4E>(bv, bi, bp);      // quad_perm [2,3,0,1]: lane ^ 2
    argmax_step_dpp<0x141>(bv, bi, bp);     // row_half_mirror: lane i <-> 7 - i (the other quad of the eight)
    argmax_step_dpp<0x140>(bv, bi, bp);     // row_mirror: lane i <-> 15 - i (the other eight of the row)
    argmax_step_rows<false>(bv, bi, bp);    // the neighbouring row of 16
    argmax_step_rows<true>(bv, bi, bp);     // the other half of the wave
}

// The k best of <= 64 * SLOTS candidates held in LDS, by k rounds of a wave-wide maximum over REGISTER-resident 64-bit keys
// (round 5).  key = f2key(value) << 32 | (0xffffffff - index): one unsigned 64-bit compare is the whole comparator (higher score
// first, ties -> lower index), keys are unique per row (indices are), 0 = an empty slot.  A round is a lane-local maximum over the
// lane's SLOTS keys, a six-step VALU butterfly on two dwords, and the owner clearing its key -- no LDS access and no 64-bit index
// plumbing inside the loop.  Equal (score, index) pairs (a caller's lists handed to sgpt_topk_merge may repeat an id) are all
// emitted, one per round: only ONE slot -- lowest lane, lowest slot -- is cleared per round (round 6; clearing every key equal to
// the round's best collapsed such pairs into one output and left the last slot empty).  A negative index is an empty slot on
// every path (it would decode to 2^32 - 1 + id).  (The LDS-scanning rounds this replaces cost ~0.8 us + 0.5 us per 256 candidates EACH -- two dependent
// LDS loads per 64 candidates, four-dword shuffles -- 17 us of launch for the k = 11 best of 251, scripts/select_probe.py.)
// Needs indices < 2^32 - 1 (the caller checks, block-uniformly, and keeps the generic rounds otherwise); -0.0 is folded into +0.0
// so that equal scores tie on the index as they do under the float comparator.  emit(round, value, index, valid) runs on lane 0.
__device__ __forceinline__ unsigned long long u64_dpp_max_step(unsigned long long x, unsigned long long o) { return o > x ? o : x; }
template <int CTRL>
__device__ __forceinline__ unsigned long long u64_dpp(unsigned long long x) {
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(x & 0xffffffffull), CTRL, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(x >> 32), CTRL, 0xf, 0xf, false);
    return ((unsigned long long)hi << 32) | lo;
}
template <bool ROW32>
__device__ __forceinline__ unsigned long long u64_rows(unsigned long long x, bool upper) {
    const unsigned xl = (unsigned)(x & 0xffffffffull), xh = (unsigned)(x >> 32);
    const auto rl = ROW32 ? __builtin_amdgcn_permlane32_swap(xl, xl, false, false) : __builtin_amdgcn_permlane16_swap(xl, xl, false, false);
    const auto rh = ROW32 ? __builtin_amdgcn_permlane32_swap(xh, xh, false, false) : __builtin_amdgcn_permlane16_swap(xh, xh, false, false);
    return ((unsigned long long)(upper ? rh[0] : rh[1]) << 32) | (upper ? rl[0] : rl[1]);
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long x, int lane) {
    x = u64_dpp_max_step(x, u64_dpp<0xB1>(x));
    x = u64_dpp_max_step(x, u64_dpp<0x4E>(x));
    x = u64_dpp_max_step(x, u64_dpp<0x141>(x));
    x = u64_dpp_max_step(x, u64_dpp<0x140>(x));
    x = u64_dpp_max_step(x, u64_rows<false>(x, (lane & 16) != 0));
    x = u64_dpp_max_step(x, u64_rows<true>(x, (lane & 32) != 0));
    return x;
}
template <int SLOTS, typename Emit>
__device__ __forceinline__ void wave_topk_keys(const float* s_val, const int64_t* s_idx, int cnt, int rounds, int lane, Emit emit) {
    unsigned long long key[SLOTS];
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
        const int c = lane + 64 * u;
        unsigned long long kk_ = 0;
        if (c < cnt) {
            const float v = s_val[c] + 0.0f;
            const int64_t id = s_idx[c];
            if (id >= 0) kk_ = ((unsigned long long)f2key(v) << 32) | (0xffffffffu - (unsigned)id);
        }
        key[u] = kk_;
    }
    for (int round = 0; round < rounds; ++round) {
        unsigned long long best = key[0];
#pragma unroll
        for (int u = 1; u < SLOTS; ++u) best = key[u] > best ? key[u] : best;
        best = wave_max_u64(best, lane);
        bool mine = false;
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) mine |= key[u] == best;
        const int owner = __builtin_amdgcn_readfirstlane(__ffsll((long long)__ballot(mine)) - 1);   // (best == 0: every lane; harmless)
        if (lane == owner) {
            bool cleared = false;
#pragma unroll
            for (int u = 0; u < SLOTS; ++u) {
                const bool hit = !cleared && key[u] == best;
                key[u] = hit ? 0ull : key[u];
                cleared |= hit;
            }
        }
        if (lane == 0) {
            const uint32_t kv = (uint32_t)(best >> 32);
            const float v = __uint_as_float((kv & 0x80000000u) ? (kv & 0x7fffffffu) : ~kv);
            emit(round, v, (int64_t)(0xffffffffu - (unsigned)(best & 0xffffffffull)), best != 0ull);
        }
    }
}
template <typename Emit>
__device__ __forceinline__ void wave_topk_keys_any(const float* s_val, const int64_t* s_idx, int cnt, int rounds, int lane, Emit emit) {
    if (cnt <= 256) wave_topk_keys<4>(s_val, s_idx, cnt, rounds, lane, emit);
    else if (cnt <= 1024) wave_topk_keys<16>(s_val, s_idx, cnt, rounds, lane, emit);
    else wave_topk_keys<32>(s_val, s_idx, cnt, rounds, lane, emit);
}
// block-uniform: does every (valid) index of s_idx[0, cnt) fit the 32-bit key field?  (ends with a barrier)
__device__ __forceinline__ bool idx_fit_keys(const int64_t* s_idx, int cnt, int t, int nthreads) {
    int big = 0;
    for (int i = t; i < cnt; i += nthreads) big |= (s_idx[i] >= 0xffffffffLL) ? 1 : 0;
    return __syncthreads_or(big) == 0;
}

// in-LDS bitonic sort of np2 (power of two) entries: descending value, ascending index
template <bool WITH_IDX>
__device__ __forceinline__ void bitonic_desc(float* s_val, int64_t* s_idx, int np2, int t) {
    for (int size = 2; size <= np2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = t; i < np2; i += TK_THREADS) {
                const int j = i ^ stride;
                if (j > i) {
                    const bool up = (i & size) == 0;  // this run ends in final (descending) order
                    const float va = s_val[i], vb = s_val[j];
                    const int64_t ia = WITH_IDX ? s_idx[i] : 0, ib = WITH_IDX ? s_idx[j] : 0;
                    const bool swap = up ? sorts_before(vb, ib, va, ia) : sorts_before(va, ia, vb, ib);
                    if (swap) {
                        s_val[i] = vb; s_val[j] = va;
                        if (WITH_IDX) { s_idx[i] = ib; s_idx[j] = ia; }
                    }
                }
            }
            __syncthreads();
        }
    }
}

// wave-aggregated histogram update: peel up to 4 distinct digits with ballots, rest by plain atomics
__device__ __forceinline__ void hist_add(unsigned* hist, bool match, unsigned dgt, int lane) {
    unsigned long long todo = __ballot(match);
#pragma unroll 1
    for (int it = 0; it < 4 && todo; ++it) {
        const int leader = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
        const unsigned d0 = (unsigned)__builtin_amdgcn_readlane((int)dgt, leader);
        const unsigned long long same = __ballot(match && dgt == d0) & todo;
        if (lane == leader) atomicAdd(&hist[d0 & 0xff], (unsigned)__popcll(same));
        todo &= ~same;
    }
    if ((todo >> lane) & 1ull) atomicAdd(&hist[dgt & 0xff], 1u);
}

// one ulp towards -inf: the INCLUSIVE form of a filter threshold (`v > below(t)` is `v >= t`); -inf / NaN stay as they are
__device__ __forceinline__ float thr_below(float t) {
    if (!(t == t) || !(t > -INFINITY)) return t;
    if (t == 0.0f) return -__uint_as_float(1u);                       // below +-0: the smallest negative subnormal
    const uint32_t u = __float_as_uint(t);
    return __uint_as_float((u & 0x80000000u) ? u + 1 : u - 1);
}

// thr_out (optional): the row's k-th best, lowered by one ulp, as the scorer's dense filter threshold -- written by the select
// itself (round 5; a launch of its own before: ~5 us of a 0.28 ms shard pass)
__global__ __launch_bounds__(TK_THREADS) void topk_select_kernel(const float* scores, long ld, long n, long idx_base,
                                                                const float* prev_val, const int64_t* prev_idx,
                                                                int n_prev, long prev_ld, int k, int nan_to_m1,
                                                                const int64_t* exclude_idx, float* out_val,
                                                                int64_t* out_idx, const int* pred, float* thr_out, int gather_k,
                                                                int gather_nq) {
    if (pred && *pred == 0) return;   // predicated fallback launch that is not needed
    __shared__ unsigned hist[256];
    __shared__ unsigned sh_prefix, sh_krem, sh_cnt, sh_cnt2;
    __shared__ float s_val[TK_SORT_MAX];
    __shared__ int64_t s_idx[TK_SORT_MAX];

    const int qrow = blockIdx.x, t = threadIdx.x, lane = t & 63;
    // no previous list: alias the prev pointers to valid memory (n_prev = 0, never selected)
    const bool has_prev = prev_val != nullptr && prev_idx != nullptr && n_prev > 0;
    const bool gath = has_prev && gather_k > 0;
    Row r{scores + (long)qrow * ld, n, idx_base,
          has_prev ? prev_val + (long)qrow * (gath ? gather_k : prev_ld) : scores,
          has_prev ? prev_idx + (long)qrow * (gath ? gather_k : prev_ld) : reinterpret_cast<const int64_t*>(out_idx),
          has_prev ? n_prev : 0, nan_to_m1, exclude_idx ? exclude_idx[qrow] : (int64_t)-1,
          gath ? gather_k : 0, gath ? (long)gather_nq * gather_k : 0};
    const long total = r.total();
    float* ov = out_val + (long)qrow * k;
    int64_t* oi = out_idx + (long)qrow * k;
    const int kk = total < k ? (int)total : k;  // number of real outputs
    const bool in_lds = k <= TK_SORT_MAX;
    int n_sorted = kk;                           // entries of s_val/s_idx to sort at the end

    bool done = false;
    if (total <= k) {  // everything survives
        for (long i = t; i < k; i += TK_THREADS) {
            const bool ok = i < total;
            const float v = ok ? r.val(ok ? i : 0) : -INFINITY;
            const int64_t id = ok ? r.idx(ok ? i : 0) : -1;
            if (in_lds) { s_val[i] = v; s_idx[i] = id; } else { ov[i] = v; oi[i] = id; }
        }
        done = true;
    } else if (k <= TK_FAST_KMAX && total <= TK_SORT_MAX && in_lds) {
        // ---------------- short rows (cross-chunk / cross-rank merges: 2 k ... world * k candidates): everything into LDS,
        // the k best by k rounds of a wave-wide arg-max -- ~2 us where the radix path below (4 histogram passes + tie
        // handling, built for rows of 10^4 ... 10^6 scores) took ~25: the fold of 8 ranks' lists, the chunk-loop merge ----
        for (long i = t; i < total; i += TK_THREADS) { s_val[i] = r.val(i); s_idx[i] = r.idx(i); }
        __syncthreads();
        const bool keys_ok = idx_fit_keys(s_idx, (int)total, t, TK_THREADS);
        if (t < 64 && keys_ok) {
            wave_topk_keys_any(s_val, s_idx, (int)total, kk, lane, [&](int round, float bv, int64_t bi, bool valid) {
                const bool ok = valid && (bv > -INFINITY || bv != bv);                // (-inf: an empty / excluded slot -> (-inf, -1) like every path)
                ov[round] = ok ? bv : -INFINITY; oi[round] = ok ? bi : -1;
                if (thr_out && round == k - 1) thr_out[qrow] = thr_below(ok ? bv : -INFINITY);
            });
            for (int i = kk + lane; i < k; i += 64) { ov[i] = -INFINITY; oi[i] = -1; }
            if (thr_out && kk < k && lane == 0) thr_out[qrow] = -INFINITY;
        } else if (t < 64) {
            const int cnt = (int)total;
            for (int round = 0; round < kk; ++round) {
                float bv = -INFINITY; int64_t bi = 0x7fffffffffffffffLL; int bp = -1;
                for (int c0 = lane; c0 < cnt; c0 += 64) {
                    const float v = s_val[c0]; const int64_t id = s_idx[c0];
                    if (id >= 0 && (bp < 0 || sorts_before(v, id, bv, bi))) { bv = v; bi = id; bp = c0; }
                }
                wave_argmax(bv, bi, bp);
                if (lane == 0) {
                    const bool ok = bp >= 0 && (bv > -INFINITY || bv != bv);          // (-inf: an empty / excluded slot -> (-inf, -1) like every path)
                    ov[round] = ok ? bv : -INFINITY; oi[round] = ok ? bi : -1;
                    if (thr_out && round == k - 1) thr_out[qrow] = thr_below(ok ? bv : -INFINITY);
                    if (bp >= 0) s_idx[bp] = -1;
                }
            }
            for (int i = kk + lane; i < k; i += 64) { ov[i] = -INFINITY; oi[i] = -1; }
            if (thr_out && kk < k && lane == 0) thr_out[qrow] = -INFINITY;
        }
        return;                                    // block-uniform exit: output written
    } else if (k <= TK_FAST_KMAX && total >= TK_FAST_MIN_ROW) {
        // ---------------- fast path: sampled threshold + one pass ----------------
        // Threshold: every thread takes the max of 16 strided samples; each wave sorts its 64 thread-maxima
        // with shuffles (no LDS, no barrier); the k-th largest of them is a lower bound of the row's k-th
        // largest (they are 64 distinct row elements), and so is the max over the 4 waves.
        const long stride = total / (TK_THREADS * 16);
        float smax = -INFINITY;
#pragma unroll 4
        for (int u = 0; u < 16; ++u) smax = fmaxf(smax, r.val(((long)u * TK_THREADS + t) * stride));
#pragma unroll
        for (int size = 2; size <= 64; size <<= 1)
#pragma unroll
            for (int sd = size >> 1; sd > 0; sd >>= 1) {
                const float other = __shfl_xor(smax, sd, 64);
                const bool keep_max = ((lane & sd) == 0) == ((lane & size) == 0);
                smax = keep_max ? fmaxf(smax, other) : fminf(smax, other);
            }
        const float tau_w = __shfl(smax, k - 1, 64);   // lane i holds the i-th largest of the wave
        if (lane == 0) hist[t >> 6] = __float_as_uint(tau_w);
        if (t == 0) sh_cnt = 0;
        __syncthreads();
        const float tau = fmaxf(fmaxf(__uint_as_float(hist[0]), __uint_as_float(hist[1])),
                                fmaxf(__uint_as_float(hist[2]), __uint_as_float(hist[3])));
        if (tau > -INFINITY) {
            auto take = [&](float v, long i) {
                if (nan_to_m1 && v != v) v = -1.0f;
                if (v >= tau) {
                    const unsigned slot = atomicAdd(&sh_cnt, 1u);
                    if (slot < TK_SORT_MAX) { s_val[slot] = v; s_idx[slot] = r.idx(i); }
                }
            };
            // fresh-score leg: 16-byte loads, two per thread in flight (row base and n are multiples of 4
            // for the scorer's chunks; anything else takes the scalar tail below)
            const bool vec_ok = ((reinterpret_cast<size_t>(r.sc) & 15) == 0);
            const long nv = vec_ok ? (n / 4) : 0;
            const float4* sc4 = reinterpret_cast<const float4*>(r.sc);
            // eight 16-byte loads per thread in flight: with few query rows (nq = 16: 16 workgroups on the chip) the pass
            // is bound by one workgroup's memory-level parallelism (two loads in flight: 3 GB/s per row, 168 us for 125 k
            // scores), not by bandwidth
            constexpr int UL = 8;
            long i0 = 0;
            for (; i0 + (long)UL * TK_THREADS <= nv; i0 += (long)UL * TK_THREADS) {
                float4 v[UL];
#pragma unroll
                for (int u = 0; u < UL; ++u) v[u] = sc4[i0 + (long)u * TK_THREADS + t];
#pragma unroll
                for (int u = 0; u < UL; ++u) {
                    const long ia = i0 + (long)u * TK_THREADS + t;
                    take(v[u].x, 4 * ia); take(v[u].y, 4 * ia + 1); take(v[u].z, 4 * ia + 2); take(v[u].w, 4 * ia + 3);
                }
            }
            for (; i0 < nv; i0 += 2 * TK_THREADS) {
                const long ia = i0 + t, ib = ia + TK_THREADS;
                float4 a = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY), b = a;
                if (ia < nv) a = sc4[ia];
                if (ib < nv) b = sc4[ib];
                if (ia < nv) { take(a.x, 4 * ia); take(a.y, 4 * ia + 1); take(a.z, 4 * ia + 2); take(a.w, 4 * ia + 3); }
                if (ib < nv) { take(b.x, 4 * ib); take(b.y, 4 * ib + 1); take(b.z, 4 * ib + 2); take(b.w, 4 * ib + 3); }
            }
            for (long i = 4 * nv + t; i < total; i += TK_THREADS) take(r.val(i), i);   // tail + previous-best leg
            __syncthreads();
            const unsigned cnt = sh_cnt;          // cnt >= k always: k sampled elements are >= tau
            if (cnt <= 1024 && in_lds) {
                // k best of the candidates by k rounds of wave-wide arg-max (wave 0; sorted, ties -> lower index)
                const bool keys_ok = idx_fit_keys(s_idx, (int)cnt, t, TK_THREADS);
                if (t < 64 && keys_ok) {
                    wave_topk_keys_any(s_val, s_idx, (int)cnt, kk, lane, [&](int round, float bv, int64_t bi, bool) {
                        ov[round] = bv; oi[round] = bi;
                        if (thr_out && round == k - 1) thr_out[qrow] = thr_below(bv);
                    });
                    for (int i = kk + lane; i < k; i += 64) { ov[i] = -INFINITY; oi[i] = -1; }
                    if (thr_out && kk < k && lane == 0) thr_out[qrow] = -INFINITY;
                } else if (t < 64) {
                    for (int round = 0; round < kk; ++round) {
                        float bv = -INFINITY; int64_t bi = 0x7fffffffffffffffLL; int bp = -1;
                        for (unsigned c0 = lane; c0 < cnt; c0 += 64) {
                            const float v = s_val[c0]; const int64_t id = s_idx[c0];
                            if (sorts_before(v, id, bv, bi)) { bv = v; bi = id; bp = (int)c0; }
                        }
                        wave_argmax(bv, bi, bp);
                        if (lane == 0) {
                            ov[round] = bv; oi[round] = bi; s_val[bp] = -INFINITY; s_idx[bp] = 0x7fffffffffffffffLL;
                            if (thr_out && round == k - 1) thr_out[qrow] = thr_below(bv);
                        }
                    }
                    for (int i = kk + lane; i < k; i += 64) { ov[i] = -INFINITY; oi[i] = -1; }
                    if (thr_out && kk < k && lane == 0) thr_out[qrow] = -INFINITY;
                }
                return;                            // block-uniform exit: output written
            }
            if (cnt <= TK_SORT_MAX) { n_sorted = (int)cnt; done = true; }
        }
        __syncthreads();
    }
    if (!done) {
        // ---------------- radix select: key of the k-th largest ----------------
        unsigned prefix = 0, krem = (unsigned)k;
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            hist[t] = 0;
            __syncthreads();
            const unsigned himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
            for (long i0 = 0; i0 < total; i0 += TK_THREADS) {
                const long i = i0 + t;
                const bool act = i < total;
                const uint32_t key = f2key(r.val(act ? i : 0));
                hist_add(hist, act && (key & himask) == prefix, (key >> shift) & 0xff, lane);
            }
            __syncthreads();
            if (t == 0) {
                unsigned acc = 0;
                int dsel = 0;
                for (int dgt = 255; dgt >= 0; --dgt) {
                    const unsigned c = hist[dgt];
                    if (acc + c >= krem) { dsel = dgt; break; }
                    acc += c;
                }
                sh_prefix = prefix | ((unsigned)dsel << shift);
                sh_krem = krem - acc;
                sh_cnt2 = hist[dsel];            // after the last pass: number of elements equal to the threshold
            }
            __syncthreads();
            prefix = sh_prefix;
            krem = sh_krem;
            __syncthreads();
        }
        const uint32_t thr = prefix;  // exact key of the k-th largest; krem = how many == thr to keep
        // (k - krem elements are strictly above the threshold key)
        // ---- gather everything above the threshold ----
        if (t == 0) sh_cnt = 0;
        __syncthreads();
        for (long i = t; i < total; i += TK_THREADS) {
            const float v = r.val(i);
            if (f2key(v) > thr) {
                const unsigned slot = atomicAdd(&sh_cnt, 1u);
                if (in_lds) { s_val[slot] = v; s_idx[slot] = r.idx(i); } else { ov[slot] = v; oi[slot] = r.idx(i); }
            }
        }
        // ---- ties at the threshold: the krem LOWEST INDICES among the equal scores ----
        // (Position in the virtual row is not index order: the previous-best leg sits behind the chunk leg but
        // holds earlier documents, and a cross-rank merge has no order at all.)  If the row holds exactly krem
        // such elements they are all taken; otherwise a second radix select, over the 64-bit index of the tied
        // elements (8 passes of 8 bits, ascending), finds the krem-th smallest index.  Indices are unique in a row.
        __syncthreads();
        const unsigned n_tie = sh_cnt2;
        unsigned long long cutoff = ~0ull;
        if (n_tie > krem) {
            unsigned long long pfx = 0;
            unsigned rem = krem;
            for (int pass = 0; pass < 8; ++pass) {
                const int shift = 56 - 8 * pass;
                hist[t] = 0;
                __syncthreads();
                const unsigned long long himask = pass == 0 ? 0ull : (~0ull << (shift + 8));
                for (long i0 = 0; i0 < total; i0 += TK_THREADS) {
                    const long i = i0 + t;
                    const bool act = i < total;
                    const bool tie = act && f2key(r.val(act ? i : 0)) == thr;
                    const unsigned long long key = (unsigned long long)r.idx(act ? i : 0);
                    hist_add(hist, tie && (key & himask) == pfx, (unsigned)(key >> shift) & 0xff, lane);
                }
                __syncthreads();
                if (t == 0) {
                    unsigned acc = 0;
                    int dsel = 255;
                    for (int dgt = 0; dgt < 256; ++dgt) {
                        const unsigned cdg = hist[dgt];
                        if (acc + cdg >= rem) { dsel = dgt; break; }
                        acc += cdg;
                    }
                    sh_prefix = (unsigned)dsel;
                    sh_krem = rem - acc;
                }
                __syncthreads();
                pfx |= (unsigned long long)sh_prefix << shift;
                rem = sh_krem;
                __syncthreads();
            }
            cutoff = pfx;
        }
        for (long i = t; i < total; i += TK_THREADS) {
            const float v = r.val(i);
            if (f2key(v) == thr && (unsigned long long)r.idx(i) <= cutoff) {
                const unsigned slot = atomicAdd(&sh_cnt, 1u);       // continues behind the k - krem gathered elements
                if (slot < (unsigned)k) {
                    if (in_lds) { s_val[slot] = v; s_idx[slot] = r.idx(i); } else { ov[slot] = v; oi[slot] = r.idx(i); }
                }
            }
        }
        n_sorted = kk;
    }
    __syncthreads();
    if (!in_lds) return;  // unsorted output for very large k
    // ---------------- sort the survivors ----------------
    int np2 = 1;
    while (np2 < n_sorted || np2 < k) np2 <<= 1;
    if (np2 > TK_SORT_MAX) np2 = TK_SORT_MAX;
    for (int i = n_sorted + t; i < np2; i += TK_THREADS) { s_val[i] = -INFINITY; s_idx[i] = 0x7fffffffffffffffLL; }
    __syncthreads();
    bitonic_desc<true>(s_val, s_idx, np2, t);
    for (int i = t; i < k; i += TK_THREADS) {
        const bool ok = i < kk;
        ov[i] = ok ? s_val[i] : -INFINITY;
        oi[i] = ok ? s_idx[i] : -1;
        if (thr_out && i == k - 1) thr_out[qrow] = thr_below(ok ? s_val[i] : -INFINITY);
    }
}

// Merge of the filtered scorer: one workgroup per query.  The running top-k arrives sorted (descending score, ties by
// ascending index; unused tail = (-inf, -1)); the <= cap appended candidates are sorted in LDS (bitonic over the next
// power of two of their COUNT -- typically ~k/2..k entries, not cap), then the two sorted lists are merged by rank:
// every element's output position is its own position plus the number of elements of the other list that sort
// before it (binary search; the order is total because indices are unique).  The candidate list holds every score of
// the chunk that beat the running k-th best, so the result equals the full-row selection.
__device__ __forceinline__ int count_before(const float* v, const int64_t* id, int n, float qv, int64_t qi) {
    int lo = 0, hi = n;                       // first position whose element does NOT sort before (qv, qi)
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sorts_before(v[mid], id[mid], qv, qi)) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(TK_THREADS) void cand_merge_kernel(const float* __restrict__ run_val,
                                                               const int64_t* __restrict__ run_idx,
                                                               const float* __restrict__ cand_val,
                                                               const int64_t* __restrict__ cand_idx,
                                                               int* __restrict__ cand_cnt, int cap, int k,
                                                               float* __restrict__ out_val, int64_t* __restrict__ out_idx,
                                                               int* __restrict__ overflow, float* __restrict__ thr_io) {
    // thr_io (optional, sampled scorer schedule): the query's dense filter threshold; raised to the merged list's k-th best
    // when that is higher.  It starts as the (inclusive) k-th best of a strided sample of the whole shard and never drops
    // below it: a first chunk whose documents all score under the sample's k-th best leaves fewer than k entries in the
    // running list, and a threshold read from that list would be -inf -- every document of the next chunk a survivor.
    __shared__ float s_val[TK_SORT_MAX];      // candidates
    __shared__ int64_t s_idx[TK_SORT_MAX];
    __shared__ float r_val[TK_MERGE_KMAX];    // running top-k
    __shared__ int64_t r_idx[TK_MERGE_KMAX];
    const int q = blockIdx.x, t = threadIdx.x;
    const int raw = cand_cnt[q];
    const int cnt = raw < cap ? raw : cap;
    int np2 = 1;
    while (np2 < cnt) np2 <<= 1;
    for (int i = t; i < np2; i += TK_THREADS) {
        const bool ok = i < cnt;
        s_val[i] = ok ? cand_val[(long)q * cap + i] : -INFINITY;
        s_idx[i] = ok ? cand_idx[(long)q * cap + i] : 0x7fffffffffffffffLL;
    }
    int nrun = 0;                              // valid entries of the running list (they come first)
    for (int i = t; i < k; i += TK_THREADS) {
        const int64_t id = run_idx[(long)q * k + i];
        r_val[i] = id >= 0 ? run_val[(long)q * k + i] : -INFINITY;
        r_idx[i] = id >= 0 ? id : 0x7fffffffffffffffLL;
    }
    __syncthreads();
    if (t == 0) {
        cand_cnt[q] = 0;                       // ready for the next filtered chunk
        if (raw > cap) atomicOr(overflow, 1);
    }
    // the candidates, sorted: small k with many candidates (the sampled schedule: ~180 of them for k = 11) -> the k best by k
    // rounds of a wave-wide arg-max (wave 0; ~1 us) instead of a bitonic sort of all of them (36 barrier stages for 256
    // entries: 29 us per launch at nq = 1000, profiles/r04_shard_profile.txt); everything else is sorted whole
    const float* cv = s_val;
    const int64_t* ci = s_idx;
    int ccnt = cnt;
    // (every thread ranking its own candidate against all of them -- cnt LDS reads each, all threads in parallel -- was measured
    // too: 72.8 us per launch against 17.8 for the arg-max rounds at ~180 candidates, profiles/r04_shard_profile.txt)
    if (k <= TK_FAST_KMAX && cnt > 2 * k) {
        float* tv_ = r_val + TK_MERGE_KMAX / 2;          // (k <= 64: the upper half of the running-list arrays is free)
        int64_t* ti_ = r_idx + TK_MERGE_KMAX / 2;
        const bool keys_ok = idx_fit_keys(s_idx, cnt, t, TK_THREADS);
        if (t < 64 && keys_ok) {
            wave_topk_keys_any(s_val, s_idx, cnt, k, t, [&](int round, float bv, int64_t bi, bool valid) {
                tv_[round] = valid ? bv : -INFINITY; ti_[round] = valid ? bi : 0x7fffffffffffffffLL;
            });
        } else if (t < 64) {
            for (int round = 0; round < k; ++round) {
                float bv = -INFINITY; int64_t bi = 0x7fffffffffffffffLL; int bp = -1;
                for (int c0 = t; c0 < cnt; c0 += 64) {
                    const float v = s_val[c0]; const int64_t id = s_idx[c0];
                    if (sorts_before(v, id, bv, bi)) { bv = v; bi = id; bp = c0; }
                }
                wave_argmax(bv, bi, bp);
                if (t == 0) { tv_[round] = bv; ti_[round] = bi; if (bp >= 0) { s_val[bp] = -INFINITY; s_idx[bp] = 0x7fffffffffffffffLL; } }
            }
        }
        __syncthreads();
        cv = tv_; ci = ti_; ccnt = k;
    } else if (cnt > 1) {
        bitonic_desc<true>(s_val, s_idx, np2, t);
    }
    nrun = count_before(r_val, r_idx, k, -INFINITY, 0x7fffffffffffffffLL);   // sentinels sort last
    const int total = nrun + ccnt;
    for (int i = t; i < nrun; i += TK_THREADS) {
        const int pos = i + count_before(cv, ci, ccnt, r_val[i], r_idx[i]);
        if (pos < k) { out_val[(long)q * k + pos] = r_val[i]; out_idx[(long)q * k + pos] = r_idx[i]; }
        if (pos == k - 1 && thr_io != nullptr && r_val[i] > thr_io[q]) thr_io[q] = r_val[i];     // (exactly one element lands on k - 1)
    }
    for (int j = t; j < ccnt; j += TK_THREADS) {
        const int pos = j + count_before(r_val, r_idx, nrun, cv[j], ci[j]);
        if (pos < k) { out_val[(long)q * k + pos] = cv[j]; out_idx[(long)q * k + pos] = ci[j]; }
        if (pos == k - 1 && thr_io != nullptr && cv[j] > thr_io[q]) thr_io[q] = cv[j];
    }
    for (int i = total + t; i < k; i += TK_THREADS) { out_val[(long)q * k + i] = -INFINITY; out_idx[(long)q * k + i] = -1; }
}

// thr[q] = the largest float strictly below list[q][k - 1] (the k-th best of a SAMPLE of the shard): the filtered scorer keeps
// scores STRICTLY above its threshold, which is right when the threshold comes from earlier documents (a later document never
// displaces an equal score) but not when it comes from a sample that also holds LATER documents -- an equal score with a lower
// index must survive.  `v > nextdown(t)` is `v >= t`.  -inf (fewer than k real scores) stays -inf: everything survives.
__global__ __launch_bounds__(256) void thr_below_kernel(const float* __restrict__ list, int k, int nq, float* __restrict__ thr) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    thr[q] = thr_below(list[(long)q * k + (k - 1)]);
}

// One launch in front of a scorer pass instead of four stream operations (each a dispatch of its own with a ~2.5 us gap):
// the query rows copied into their padded tile (pad rows zero), the candidate counters and overflow flags cleared, and --
// sampled schedule without an incoming running list -- the running list's indices set to -1 (an empty slot whatever its value).
__global__ __launch_bounds__(256) void score_prep_kernel(const uint4* __restrict__ q, uint4* __restrict__ qpad, long q_vec, long qpad_vec,
                                                         int* __restrict__ counters, long n_counters, long long* __restrict__ idx_list,
                                                         long n_idx) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < qpad_vec; i += stride) qpad[i] = i < q_vec ? q[i] : make_uint4(0, 0, 0, 0);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_counters; i += stride) counters[i] = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_idx; i += stride) idx_list[i] = -1;
}

// Stage 2 of the refined scorer (sgpt_score_topk_refined): the EXACT fp32 score of every stage-1 candidate.  One wave per
// (query, candidate): lane l sums the float4 chunks l, l + 64, ... of the two rows, the wave adds its 64 partial sums in a fixed
// butterfly -- deterministic, independent of nq / m / the launch shape.  idx < 0 (an unused slot) -> (-inf, -1); NaN -> -1
// (exact_search.py:99).  Values and indices land in the first m columns of a [nq, ld_out] list that the caller completes with
// the running list and hands to topk_select_kernel.
__global__ __launch_bounds__(256) void rescore_kernel(const float* __restrict__ q, const float* __restrict__ corpus,
                                                      const int64_t* __restrict__ idx, long ld_idx, int m, int d, long idx_base,
                                                      long n_docs, float* __restrict__ out_val, int64_t* __restrict__ out_idx, long ld_out) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), qi = blockIdx.y, lane = threadIdx.x & 63;
    if (j >= m) return;
    const int64_t id = idx[(long)qi * ld_idx + j];
    const long row = (long)(id - idx_base);
    float v = -INFINITY;
    if (id >= 0 && row >= 0 && row < n_docs) {
        const float4* a = reinterpret_cast<const float4*>(q + (long)qi * d);
        const float4* b = reinterpret_cast<const float4*>(corpus + row * d);
        float acc = 0.f;
        for (int c = lane; c < d / 4; c += 64) {
            const float4 x = a[c], y = b[c];
            acc = __builtin_fmaf(x.x, y.x, acc); acc = __builtin_fmaf(x.y, y.y, acc);
            acc = __builtin_fmaf(x.z, y.z, acc); acc = __builtin_fmaf(x.w, y.w, acc);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        v = acc != acc ? -1.0f : acc;
    }
    if (lane == 0) {
        out_val[(long)qi * ld_out + j] = v;
        out_idx[(long)qi * ld_out + j] = v > -INFINITY ? id : (int64_t)-1;
    }
}

// The guarantee of the refined scorer, checked per query on the stage-1 (16-bit) scores, sorted descending: every document
// outside the kp candidates scores at most v16[kp-1]; if that is more than `margin` = 2 eps below the k-th best 16-bit score,
// no outsider can reach the fp32 top-k (|s16 - s32| <= eps for every pair).  A query for which it does not hold -- a mass of
// near-equal scores: duplicated documents -- raises the flag that switches the predicated exact pass on.
__global__ __launch_bounds__(256) void refine_check_kernel(const float* __restrict__ v16, int k, int kp, float margin, int nq,
                                                           int* __restrict__ flag) {
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    const float t = v16[(long)qi * kp + (k - 1)], u = v16[(long)qi * kp + (kp - 1)];
    if (u > -INFINITY && !(u < t - margin)) atomicOr(flag, 1);
}

// running list [nq, n_run] (row stride ld_in) -> columns [col0, col0 + n_run) of a [nq, ld_out] list
__global__ __launch_bounds__(256) void list_append_kernel(const float* __restrict__ in_val, const int64_t* __restrict__ in_idx, long ld_in,
                                                          int n_run, int nq, float* __restrict__ out_val, int64_t* __restrict__ out_idx,
                                                          long ld_out, int col0) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)nq * n_run) return;
    const long qi = i / n_run, j = i - qi * n_run;
    out_val[qi * ld_out + col0 + j] = in_val[qi * ld_in + j];
    out_idx[qi * ld_out + col0 + j] = in_idx[qi * ld_in + j];
}

}  // namespace

void launch_rescore(const float* q, const float* corpus, const int64_t* idx, long ld_idx, int nq, int m, int d, long idx_base,
                    long n_docs, float* out_val, int64_t* out_idx, long ld_out, hipStream_t s) {
    hipLaunchKernelGGL(rescore_kernel, dim3((m + 3) / 4, nq), dim3(256), 0, s, q, corpus, idx, ld_idx, m, d, idx_base, n_docs,
                       out_val, out_idx, ld_out);
}

void launch_refine_check(const float* v16, int k, int kp, float margin, int nq, int* flag, hipStream_t s) {
    hipLaunchKernelGGL(refine_check_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, v16, k, kp, margin, nq, flag);
}

void launch_list_append(const float* in_val, const int64_t* in_idx, long ld_in, int n_run, int nq, float* out_val, int64_t* out_idx,
                        long ld_out, int col0, hipStream_t s) {
    if (n_run <= 0) return;
    hipLaunchKernelGGL(list_append_kernel, dim3((unsigned)(((long)nq * n_run + 255) / 256)), dim3(256), 0, s, in_val, in_idx, ld_in,
                       n_run, nq, out_val, out_idx, ld_out, col0);
}

void launch_score_prep(const void* q, void* qpad, long q_bytes, long qpad_bytes, int* counters, long n_counters, long long* idx_list,
                       long n_idx, hipStream_t s) {
    long work = qpad_bytes / 16;
    if (n_counters > work) work = n_counters;
    if (n_idx > work) work = n_idx;
    const long blocks = (work + 255) / 256;
    hipLaunchKernelGGL(score_prep_kernel, dim3((unsigned)(blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks))), dim3(256), 0, s,
                       (const uint4*)q, (uint4*)qpad, q_bytes / 16, qpad_bytes / 16, counters, n_counters, idx_list, n_idx);
}

void launch_thr_below(const float* list, int k, int nq, float* thr, hipStream_t s) {
    hipLaunchKernelGGL(thr_below_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, list, k, nq, thr);
}

void launch_topk_select(const float* scores, long ld, long n, long idx_base, const float* prev_val,
                        const int64_t* prev_idx, int n_prev, long prev_ld, int nq, int k, int nan_to_m1,
                        const int64_t* exclude_idx, float* out_val, int64_t* out_idx, hipStream_t s, const int* pred,
                        float* thr_out, int gather_k) {
    hipLaunchKernelGGL(topk_select_kernel, dim3(nq), dim3(TK_THREADS), 0, s, scores, ld, n, idx_base, prev_val,
                       prev_idx, n_prev, prev_ld, k, nan_to_m1, exclude_idx, out_val, out_idx, pred, thr_out, gather_k, nq);
}

void launch_cand_merge(const float* run_val, const int64_t* run_idx, const float* cand_val, const int64_t* cand_idx,
                       int* cand_cnt, int cap, int nq, int k, float* out_val, int64_t* out_idx, int* overflow,
                       hipStream_t s, float* thr_io) {
    hipLaunchKernelGGL(cand_merge_kernel, dim3(nq), dim3(TK_THREADS), 0, s, run_val, run_idx, cand_val, cand_idx, cand_cnt,
                       cap, k, out_val, out_idx, overflow, thr_io);
}
