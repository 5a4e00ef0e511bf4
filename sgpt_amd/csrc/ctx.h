// The context object behind `sgpt_ctx*` (include/sgpt_hip.h), shared by api.hip and comm.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <utility>
#include <vector>

struct sgpt_ctx {
    int device = 0;
    std::string err;
    // grow-only workspaces
    void* ws = nullptr; size_t ws_bytes = 0;        // encoder activations
    void* ws2 = nullptr; size_t ws2_bytes = 0;      // scorer: score chunk + ping-pong top-k
    void* ws3 = nullptr; size_t ws3_bytes = 0;      // multi-GPU exchange staging (comm.hip)
    void* ws4 = nullptr; size_t ws4_bytes = 0;      // refined scorer: 16-bit queries, stage-1 lists, candidate list, saved running list, flag
    void* comm = nullptr;                           // ncclComm_t of this ctx (sgpt_comm_init), one per process / GPU
    int comm_rank = 0, comm_world = 0;
    // bumped whenever a library-owned buffer that launched kernels point into is re-allocated (workspace growth,
    // learnt pooling weights): a hipGraph captured earlier holds stale pointers once this moves (sgpt_ctx_generation)
    uint64_t generation = 0;
    int* range_flag = nullptr;                      // device int: an f16 activation left the representable range (ctx-level ops: sgpt_linear*)
    int kgroups = 1;                                // low-latency mode: 2 = k-groups for query-sized launches (sgpt_ctx_set_low_latency)
    int force256 = 0;                               // tile policy: 1 = 256x256 tiles even for small problems (sgpt_ctx_set_tile_policy)
    int no_qpath = 0;                               // tile policy 2: query- / mid-sized layouts keep the bulk path's small-tile kernels (A/B, tests)
    int cu_cap = 0;                                 // persistent 256x256 GEMM: workgroups per launch at most (0 = one per CU; sgpt_ctx_set_gemm_cu_cap)
    // GEMM profiling (bench.py roofline)
    bool prof = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
    std::vector<double> ev_flops;
    int64_t prof_launches = 0; double prof_ms = 0, prof_flops = 0;
};
