// Multi-GPU exchange steps of the search path on RCCL (one process per GPU, xGMI), behind the C ABI.
//
// The reference's only inference-path collectives are the two torch.distributed all-gathers of
// util.mismatched_sizes_all_gather (sentence_transformers/util.py:326-347: sizes first, then rows padded to the maximum).
// Here every rank knows every rank's row count up front (shards are a pure function of the global length list), so one
// ncclAllGather moves the rows; the top-k exchange of the corpus-sharded search gathers the per-rank
// [nq, k] (score, index) lists and folds them with the library's own merge kernel on the same stream.
//
// RCCL is bound LAZILY (round 4; rccl.h supplies the types only, the library is not linked against librccl): the first
// communicator call dlopens it, preferring the copy that is already in the process -- torch ships its own librccl.so.1, and a
// process must never hold two -- and checks that its NCCL major version is the one of the header this file was compiled
// against.  A single-GPU user needs no RCCL to build or load libsgpt_hip.so, and the order of `import torch` and the dlopen of
// this library no longer decides which RCCL the collectives run on.  torch.distributed is only the bootstrap that carries the
// 128-byte unique id from rank 0 to the other ranks (sgpt_amd/dist.py).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sgpt_hip.h"
#include "common.h"
#include "ctx.h"

static_assert(SGPT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique-id size");

namespace {

// The seven RCCL entry points this file calls, resolved once per process (an immutable table after the first use).
struct Rccl {
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;          // empty: usable
    std::string path;
};

const Rccl& rccl() {
    static const Rccl table = [] {
        Rccl r;
        void* h = nullptr;
        std::vector<std::string> names = {"librccl.so.1", "librccl.so"};
        for (const auto& n : names) if (!h) h = dlopen(n.c_str(), RTLD_NOW | RTLD_NOLOAD);        // the copy already in the process
        if (const char* rp = getenv("ROCM_PATH")) { names.push_back(std::string(rp) + "/lib/librccl.so.1"); }
        names.push_back("/opt/rocm/lib/librccl.so.1");
        for (const auto& n : names) if (!h) h = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL);
        if (!h) {
            const char* e = dlerror();      // (one call: dlerror() clears the state it returns)
            r.err = std::string("librccl.so.1 not found (") + (e ? e : "") + "): the multi-GPU exchange steps need RCCL";
            return r;
        }
#define SGPT_SYM(field, name)                                                                   \
        r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, #name));                            \
        if (!r.field && r.err.empty()) r.err = std::string("librccl does not export ") + #name;
        SGPT_SYM(GetVersion, ncclGetVersion) SGPT_SYM(GetUniqueId, ncclGetUniqueId) SGPT_SYM(CommInitRank, ncclCommInitRank)
        SGPT_SYM(CommDestroy, ncclCommDestroy) SGPT_SYM(AllGather, ncclAllGather) SGPT_SYM(GroupStart, ncclGroupStart)
        SGPT_SYM(GroupEnd, ncclGroupEnd) SGPT_SYM(GetErrorString, ncclGetErrorString)
#undef SGPT_SYM
        Dl_info info;
        if (r.GetVersion && dladdr(reinterpret_cast<void*>(r.GetVersion), &info) && info.dli_fname) r.path = info.dli_fname;
        int v = 0;
        if (r.err.empty() && (r.GetVersion(&v) != ncclSuccess || v / 10000 != NCCL_MAJOR))
            r.err = "librccl at " + r.path + " reports NCCL version " + std::to_string(v) + ", this library was compiled against major " +
                    std::to_string(NCCL_MAJOR);
        return r;
    }();
    return table;
}

#define HIPC(ctx, call)                                                                       \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                   \
            return SGPT_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)
#define NCCLC(ctx, call)                                                                      \
    do {                                                                                      \
        ncclResult_t r_ = (call);                                                             \
        if (r_ != ncclSuccess) {                                                              \
            (ctx)->err = std::string(#call) + ": " + rccl().GetErrorString(r_);               \
            return SGPT_ERR_COMM;                                                             \
        }                                                                                     \
    } while (0)

sgpt_status cfail(sgpt_ctx* c, sgpt_status st, const char* m) { c->err = m; return st; }

// [world][nq][k] (rank-major, as ncclAllGather delivers) -> [nq][world * k] (candidates of one query contiguous)
__global__ __launch_bounds__(256) void gather_to_rows_kernel(const float* __restrict__ gv, const long long* __restrict__ gi,
                                                             float* __restrict__ ov, long long* __restrict__ oi, int world,
                                                             int nq, int k) {
    const long total = (long)world * nq * k;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int j = (int)(e % k);
        const long t = e / k;
        const int q = (int)(t % nq), r = (int)(t / nq);
        const long o = ((long)q * world + r) * k + j;
        ov[o] = gv[e];
        oi[o] = gi[e];
    }
}

sgpt_status ensure3(sgpt_ctx* c, size_t need) {
    if (c->ws3_bytes >= need) return SGPT_OK;
    if (c->ws3) { HIPC(c, hipDeviceSynchronize()); HIPC(c, hipFree(c->ws3)); c->ws3 = nullptr; c->ws3_bytes = 0; }
    need = (need + (need >> 2) + ((size_t)1 << 20) - 1) >> 20 << 20;
    if (hipMalloc(&c->ws3, need) != hipSuccess) { c->ws3 = nullptr; return cfail(c, SGPT_ERR_OOM, "hipMalloc exchange workspace failed"); }
    c->ws3_bytes = need;
    c->generation++;
    return SGPT_OK;
}

}  // namespace

extern "C" {

sgpt_status sgpt_comm_unique_id(uint8_t* id) {
    if (!id) return SGPT_ERR_INVALID;
    if (!rccl().err.empty()) return SGPT_ERR_COMM;
    ncclUniqueId u;
    if (rccl().GetUniqueId(&u) != ncclSuccess) return SGPT_ERR_COMM;
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return SGPT_OK;
}

sgpt_status sgpt_comm_init(sgpt_ctx* c, const uint8_t* id, int32_t rank, int32_t world) {
    if (!c) return SGPT_ERR_INVALID;
    if (!id || world <= 0 || rank < 0 || rank >= world) return cfail(c, SGPT_ERR_INVALID, "sgpt_comm_init: bad rank / world");
    if (c->comm) return cfail(c, SGPT_ERR_INVALID, "sgpt_comm_init: this ctx already has a communicator (sgpt_comm_destroy first)");
    if (!rccl().err.empty()) { c->err = rccl().err; return SGPT_ERR_COMM; }
    HIPC(c, hipSetDevice(c->device));
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    NCCLC(c, rccl().CommInitRank(&comm, world, u, rank));
    c->comm = comm;
    c->comm_rank = rank;
    c->comm_world = world;
    return SGPT_OK;
}

sgpt_status sgpt_comm_destroy(sgpt_ctx* c) {
    if (!c) return SGPT_ERR_INVALID;
    if (c->comm) {
        (void)hipSetDevice(c->device);
        (void)hipDeviceSynchronize();
        (void)rccl().CommDestroy((ncclComm_t)c->comm);
        c->comm = nullptr;
        c->comm_world = 0;
        c->comm_rank = 0;
    }
    return SGPT_OK;
}

int32_t sgpt_comm_world(const sgpt_ctx* c) { return c && c->comm ? c->comm_world : 0; }
int32_t sgpt_comm_rank(const sgpt_ctx* c) { return c && c->comm ? c->comm_rank : -1; }

static sgpt_status allgather_rows_impl(sgpt_ctx* c, const void* local, const int64_t* counts, int64_t row_bytes, void* out,
                                       void* stream, bool force_padded) {
    if (!c) return SGPT_ERR_INVALID;
    if (!c->comm) return cfail(c, SGPT_ERR_INVALID, "sgpt_allgather_rows: no communicator (sgpt_comm_init)");
    if (!counts || !out || row_bytes <= 0) return cfail(c, SGPT_ERR_INVALID, "sgpt_allgather_rows: bad arguments");
    const int world = c->comm_world, rank = c->comm_rank;
    int64_t mx = 0, total = 0;
    bool equal = true;
    for (int r = 0; r < world; ++r) {
        if (counts[r] < 0) return cfail(c, SGPT_ERR_INVALID, "sgpt_allgather_rows: negative row count");
        mx = counts[r] > mx ? counts[r] : mx;
        equal = equal && counts[r] == counts[0];
        total += counts[r];
    }
    if (total == 0) return SGPT_OK;
    if (counts[rank] > 0 && !local) return cfail(c, SGPT_ERR_INVALID, "sgpt_allgather_rows: local rows missing");
    HIPC(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t blk = (size_t)mx * row_bytes;
    if (equal && !force_padded) {            // the common case (shard sizes differ by at most one row only when n % world != 0)
        NCCLC(c, rccl().AllGather(local, out, blk, ncclInt8, (ncclComm_t)c->comm, s));
        return SGPT_OK;
    }
    // ragged: pad this rank's block to the largest one, gather into the exchange workspace, compact in rank order
    sgpt_status st = ensure3(c, blk * ((size_t)world + 1));
    if (st != SGPT_OK) return st;
    char* send = (char*)c->ws3;
    char* recv = send + blk;
    const size_t mine = (size_t)counts[rank] * row_bytes;
    if (mine) HIPC(c, hipMemcpyAsync(send, local, mine, hipMemcpyDeviceToDevice, s));
    if (mine < blk) HIPC(c, hipMemsetAsync(send + mine, 0, blk - mine, s));
    NCCLC(c, rccl().AllGather(send, recv, blk, ncclInt8, (ncclComm_t)c->comm, s));
    size_t o = 0;
    for (int r = 0; r < world; ++r) {
        const size_t b = (size_t)counts[r] * row_bytes;
        if (b) HIPC(c, hipMemcpyAsync((char*)out + o, recv + (size_t)r * blk, b, hipMemcpyDeviceToDevice, s));
        o += b;
    }
    return SGPT_OK;
}

sgpt_status sgpt_allgather_rows(sgpt_ctx* c, const void* local, const int64_t* counts, int64_t row_bytes, void* out,
                                void* stream) {
    return allgather_rows_impl(c, local, counts, row_bytes, out, stream, false);
}

sgpt_status sgpt_allgather_rows_padded(sgpt_ctx* c, const void* local, const int64_t* counts, int64_t row_bytes, void* out,
                                       void* stream) {
    return allgather_rows_impl(c, local, counts, row_bytes, out, stream, true);
}

// The rank-local half of the exchange: [world][nq][k] gathered lists -> the k_out best per query.  Needs no communicator:
// a caller with its own transport uses it directly, and the tests drive it on one GPU with a simulated world.
static sgpt_status fold_gathered(sgpt_ctx* c, const float* gv, const long long* gi, int world, int nq, int k, int k_out,
                                 const int64_t* exclude_idx, float* out_val, int64_t* out_idx, float* rv, long long* ri,
                                 hipStream_t s) {
    const size_t nw = (size_t)nq * k * world;
    const float* mv = gv;
    const long long* mi = gi;
    // the k_out best of the world * k candidates per query; idx < 0 and idx == exclude_idx[q] are skipped
    // (exact_search.py:118, 121-132); ties -> lowest index: identical on every rank.  The rank-major -> row-major regrouping is done
    // by the select's own loads (round 5: its "gathered" previous leg; the permute kernel of round 3 was a launch of its own)
    const int mcand = world * k;
    (void)rv; (void)ri; (void)nw;
    launch_topk_select(mv, mcand, 0, 0, mv, (const int64_t*)mi, mcand, mcand, nq, k_out, 0, exclude_idx, out_val, out_idx, s,
                       nullptr, nullptr, world > 1 ? k : 0);
    HIPC(c, hipGetLastError());
    return SGPT_OK;
}

sgpt_status sgpt_fold_gathered_topk(sgpt_ctx* c, const float* gathered_val, const int64_t* gathered_idx, int32_t world, int32_t nq,
                                    int32_t k, int32_t k_out, const int64_t* exclude_idx, float* out_val, int64_t* out_idx,
                                    void* stream) {
    if (!c) return SGPT_ERR_INVALID;
    if (!gathered_val || !gathered_idx || !out_val || !out_idx || world <= 0 || nq <= 0 || k <= 0 || k_out <= 0)
        return cfail(c, SGPT_ERR_INVALID, "sgpt_fold_gathered_topk: bad arguments");
    HIPC(c, hipSetDevice(c->device));
    const size_t nw = (size_t)nq * k * world;
    const size_t vb = (nw * 4 + 255) / 256 * 256, ib = (nw * 8 + 255) / 256 * 256;
    sgpt_status st = ensure3(c, vb + ib);
    if (st != SGPT_OK) return st;
    char* base = (char*)c->ws3;
    return fold_gathered(c, gathered_val, (const long long*)gathered_idx, world, nq, k, k_out, exclude_idx, out_val, out_idx,
                         (float*)base, (long long*)(base + vb), (hipStream_t)stream);
}

sgpt_status sgpt_exchange_topk(sgpt_ctx* c, const float* val, const int64_t* idx, int32_t nq, int32_t k, int32_t k_out,
                               const int64_t* exclude_idx, float* out_val, int64_t* out_idx, void* stream) {
    if (!c) return SGPT_ERR_INVALID;
    if (!c->comm) return cfail(c, SGPT_ERR_INVALID, "sgpt_exchange_topk: no communicator (sgpt_comm_init)");
    if (!val || !idx || !out_val || !out_idx || nq <= 0 || k <= 0 || k_out <= 0)
        return cfail(c, SGPT_ERR_INVALID, "sgpt_exchange_topk: bad arguments");
    HIPC(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const int world = c->comm_world;
    const size_t n1 = (size_t)nq * k, nw = n1 * world;
    // workspace: gathered values | gathered indices | row-major values | row-major indices
    const size_t vb = (nw * 4 + 255) / 256 * 256, ib = (nw * 8 + 255) / 256 * 256;
    sgpt_status st = ensure3(c, 2 * (vb + ib));
    if (st != SGPT_OK) return st;
    char* base = (char*)c->ws3;
    float* gv = (float*)base;
    long long* gi = (long long*)(base + vb);
    float* rv = (float*)(base + vb + ib);
    long long* ri = (long long*)(base + 2 * vb + ib);
    NCCLC(c, rccl().GroupStart());
    ncclResult_t r1 = rccl().AllGather(val, gv, n1, ncclFloat32, (ncclComm_t)c->comm, s);
    ncclResult_t r2 = rccl().AllGather(idx, gi, n1, ncclInt64, (ncclComm_t)c->comm, s);
    NCCLC(c, rccl().GroupEnd());
    NCCLC(c, r1);
    NCCLC(c, r2);
    return fold_gathered(c, gv, gi, world, nq, k, k_out, exclude_idx, out_val, out_idx, rv, ri, s);
}

}  // extern "C"
