// Per-query top-k selection with running merge: torch.topk(k, largest, sorted=False) of
// custommodels/exact_search.py:102-108 plus the heapq.nlargest merge of :121-132.
//
// One 256-thread workgroup per query scans a VIRTUAL row = [chunk scores (n) | previous best
// (n_prev)]: 4 radix passes (8 bits each, LDS histogram, MSB first) on the order-preserving
// uint32 image of the fp32 score find the exact k-th largest key; a final pass gathers every
// element above it plus the lowest-index ties, and the k survivors are bitonic-sorted in LDS
// (descending score, ascending index) so the output is deterministic.  The row is re-read
// 5x from L2 / Infinity Cache (the score chunk is sized to stay on-die by the caller).
#include "common.h"

namespace {

__device__ __forceinline__ uint32_t f2key(float f) {  // ascending uint order == ascending float order
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

constexpr int TK_THREADS = 256;
constexpr int TK_SORT_MAX = 2048;  // k <= 2048 sorted in LDS (the driver's k+1 = 1001 fits)

struct Row {
    const float* sc; long n; long idx_base;
    const float* pv; const int64_t* pi; int n_prev; int nan_to_m1; int64_t excl;
    __device__ __forceinline__ long total() const { return n + n_prev; }
    __device__ __forceinline__ float val(long i) const {
        float v = i < n ? sc[i] : pv[i - n];
        if (i >= n && (pi[i - n] < 0 || pi[i - n] == excl)) v = -INFINITY;  // empty slot / excluded (self) id
        if (nan_to_m1 && v != v) v = -1.0f;                   // exact_search.py:99
        return v;
    }
    __device__ __forceinline__ int64_t idx(long i) const {
        if (i < n) return idx_base + i;
        const int64_t id = pi[i - n];
        return id == excl ? -1 : id;
    }
};

__global__ __launch_bounds__(TK_THREADS) void topk_select_kernel(const float* __restrict__ scores, long ld, long n,
                                                                long idx_base, const float* __restrict__ prev_val,
                                                                const int64_t* __restrict__ prev_idx, int n_prev,
                                                                long prev_ld, int k, int nan_to_m1,
                                                                const int64_t* __restrict__ exclude_idx,
                                                                float* __restrict__ out_val,
                                                                int64_t* __restrict__ out_idx) {
    __shared__ unsigned hist[256];
    __shared__ unsigned sh_prefix, sh_krem, sh_cnt_gt, sh_cnt_eq;
    __shared__ float s_val[TK_SORT_MAX];
    __shared__ int64_t s_idx[TK_SORT_MAX];

    const int qrow = blockIdx.x, t = threadIdx.x;
    Row r{scores + (long)qrow * ld, n, idx_base, prev_val ? prev_val + (long)qrow * prev_ld : nullptr,
          prev_idx ? prev_idx + (long)qrow * prev_ld : nullptr, prev_val ? n_prev : 0, nan_to_m1,
          exclude_idx ? exclude_idx[qrow] : (int64_t)-1};
    const long total = r.total();
    float* ov = out_val + (long)qrow * k;
    int64_t* oi = out_idx + (long)qrow * k;
    const int kk = total < k ? (int)total : k;  // number of real outputs

    if (total <= k) {  // everything survives
        for (long i = t; i < k; i += TK_THREADS) {
            const bool ok = i < total;
            const float v = ok ? r.val(i) : -INFINITY;
            const int64_t id = ok ? r.idx(i) : -1;
            if (k <= TK_SORT_MAX) { s_val[i] = v; s_idx[i] = id; } else { ov[i] = v; oi[i] = id; }
        }
    } else {
        // ---- radix select: find key of the k-th largest ----
        unsigned prefix = 0, krem = (unsigned)k;
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            hist[t] = 0;
            __syncthreads();
            const unsigned himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
            for (long i = t; i < total; i += TK_THREADS) {
                const uint32_t key = f2key(r.val(i));
                if ((key & himask) == prefix) atomicAdd(&hist[(key >> shift) & 0xff], 1u);
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
            }
            __syncthreads();
            prefix = sh_prefix;
            krem = sh_krem;
            __syncthreads();
        }
        const uint32_t thr = prefix;  // exact key of the k-th largest; krem = how many == thr to keep
        // ---- gather: all > thr, then the krem lowest-index == thr ----
        if (t == 0) { sh_cnt_gt = 0; sh_cnt_eq = 0; }
        __syncthreads();
        const unsigned n_gt = (unsigned)k - krem;
        // ties: deterministic lowest-index choice needs an ordered scan; ties at the threshold are
        // rare, so thread 0 resolves them serially only when there are more ties than slots.
        for (long i = t; i < total; i += TK_THREADS) {
            const float v = r.val(i);
            const uint32_t key = f2key(v);
            if (key > thr) {
                const unsigned slot = atomicAdd(&sh_cnt_gt, 1u);
                if (k <= TK_SORT_MAX) { s_val[slot] = v; s_idx[slot] = r.idx(i); } else { ov[slot] = v; oi[slot] = r.idx(i); }
            } else if (key == thr) {
                atomicAdd(&sh_cnt_eq, 1u);
            }
        }
        __syncthreads();
        const unsigned n_eq = sh_cnt_eq;
        __syncthreads();
        if (n_eq == krem) {  // common case: take every tie
            if (t == 0) sh_cnt_eq = 0;
            __syncthreads();
            for (long i = t; i < total; i += TK_THREADS) {
                const float v = r.val(i);
                if (f2key(v) == thr) {
                    const unsigned slot = n_gt + atomicAdd(&sh_cnt_eq, 1u);
                    if (k <= TK_SORT_MAX) { s_val[slot] = v; s_idx[slot] = r.idx(i); } else { ov[slot] = v; oi[slot] = r.idx(i); }
                }
            }
        } else if (t == 0) {  // more ties than slots: lowest positions first (serial, rare)
            unsigned taken = 0;
            for (long i = 0; i < total && taken < krem; ++i) {
                const float v = r.val(i);
                if (f2key(v) == thr) {
                    const unsigned slot = n_gt + taken++;
                    if (k <= TK_SORT_MAX) { s_val[slot] = v; s_idx[slot] = r.idx(i); } else { ov[slot] = v; oi[slot] = r.idx(i); }
                }
            }
        }
    }
    __syncthreads();
    if (k > TK_SORT_MAX) return;  // unsorted output for very large k
    // ---- bitonic sort of the survivors: descending value, ascending index ----
    int np2 = 1;
    while (np2 < k) np2 <<= 1;
    for (int i = kk + t; i < np2; i += TK_THREADS)
        if (i < TK_SORT_MAX) { s_val[i] = -INFINITY; s_idx[i] = 0x7fffffffffffffffLL; }
    __syncthreads();
    auto before = [](float va, int64_t ia, float vb, int64_t ib) {  // a sorts before b
        return va > vb || (va == vb && ia < ib);
    };
    for (int size = 2; size <= np2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = t; i < np2; i += TK_THREADS) {
                const int j = i ^ stride;
                if (j > i) {
                    const bool up = (i & size) == 0;  // "up" = this run is in final (descending) order
                    const float va = s_val[i], vb = s_val[j];
                    const int64_t ia = s_idx[i], ib = s_idx[j];
                    const bool swap = up ? before(vb, ib, va, ia) : before(va, ia, vb, ib);
                    if (swap) { s_val[i] = vb; s_val[j] = va; s_idx[i] = ib; s_idx[j] = ia; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = t; i < k; i += TK_THREADS) {
        const bool ok = i < kk;
        ov[i] = ok ? s_val[i] : -INFINITY;
        oi[i] = ok ? s_idx[i] : -1;
    }
}

}  // namespace

void launch_topk_select(const float* scores, long ld, long n, long idx_base, const float* prev_val,
                        const int64_t* prev_idx, int n_prev, long prev_ld, int nq, int k, int nan_to_m1,
                        const int64_t* exclude_idx, float* out_val, int64_t* out_idx, hipStream_t s) {
    hipLaunchKernelGGL(topk_select_kernel, dim3(nq), dim3(TK_THREADS), 0, s, scores, ld, n, idx_base, prev_val,
                       prev_idx, n_prev, prev_ld, k, nan_to_m1, exclude_idx, out_val, out_idx);
}
