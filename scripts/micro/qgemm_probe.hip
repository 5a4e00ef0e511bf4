// Stand-alone probe of the query-sized projection kernels (sgpt_amd/csrc/qgemm.hip built with -DSGPT_QSTAMPS): per-launch time of
// one shape in a back-to-back loop and the s_memtime stamps of workgroup 0 (cycles relative to its entry):
//   1 ring prologue issued   2 LayerNorm prologue done (LN shapes)   8+i stage i released by its barrier   3 tile 0 k-loop done   4 stores issued
// Build + run: scripts/gpu_qprobe.sh.   argv: M N K epi(0 store, 1 gelu, 2 resid, 7 qkv) ln(0|1) [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "common.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: qgemm_probe M N K epi ln [iters]\n"); return 2; }
    const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]), epi = atoi(argv[4]), ln = atoi(argv[5]);
    const int iters = argc > 6 ? atoi(argv[6]) : 200;
    void *A, *W, *O, *O2; float *x, *g, *b, *bias, *R; long long* dbg;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&O, (size_t)M * N * 4)); CK(hipMalloc(&O2, (size_t)M * N * 4));
    CK(hipMalloc((void**)&x, (size_t)M * K * 4)); CK(hipMalloc((void**)&g, K * 4)); CK(hipMalloc((void**)&b, K * 4));
    CK(hipMalloc((void**)&bias, N * 4)); CK(hipMalloc((void**)&R, (size_t)M * N * 4)); CK(hipMalloc((void**)&dbg, 8 * 16 * 8));
    launch_fill_rand(A, (long)M * K, DT_F16, 1u, 1.0f, 0); launch_fill_rand(W, (long)N * K, DT_F16, 2u, 0.05f, 0);
    launch_fill_rand(x, (long)M * K, DT_F32, 3u, 1.0f, 0); launch_fill_rand(g, K, DT_F32, 4u, 1.0f, 0); launch_fill_rand(b, K, DT_F32, 5u, 0.1f, 0);
    launch_fill_rand(bias, N, DT_F32, 6u, 0.1f, 0); launch_fill_rand(R, (long)M * N, DT_F32, 7u, 1.0f, 0);
    CK(hipMemset(dbg, 0, 8 * 16 * 8));
    QGemmArgs q{};
    q.g.A = A; q.g.lda = K; q.g.W = W; q.g.ldw = K; q.g.M = M; q.g.m_valid = M; q.g.N = N; q.g.K = K; q.g.out = O; q.g.ldo = N;
    q.g.bias = (epi == EPI_BIAS_GELU || epi == EPI_BIAS_RESID) ? bias : nullptr; q.g.resid = epi == EPI_BIAS_RESID ? R : nullptr;
    if (epi == EPI_BIAS_RESID) q.g.out = R;
    if (epi == EPI_QKV) { q.g.n_split = N / 3 * 2; q.g.ldo = q.g.n_split; q.g.out2 = O2; q.g.ldo2 = M; }
    if (ln) { q.x = x; q.ln_g = g; q.ln_b = b; q.eps = 1e-5f; }
    q.g.dbg = dbg;
    const int odt = epi == EPI_BIAS_RESID ? DT_F32 : DT_F16;
    if (!launch_qgemm(DT_F16, epi, odt, q, 0)) { fprintf(stderr, "shape not served\n"); return 3; }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 10; ++i) launch_qgemm(DT_F16, epi, odt, q, 0);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) launch_qgemm(DT_F16, epi, odt, q, 0);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    long long h[128]; CK(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
    printf("M %d N %d K %d epi %d ln %d: %.2f us per launch (back to back)\n", M, N, K, epi, ln, ms / iters * 1e3);
    for (int w = 0; w < 8; ++w) {
        const long long* r = h + w * 16;
        if (!r[0]) continue;
        printf("  wave %d: pro %lld  ln %lld  stages", w, r[1] - r[0], r[2] - r[0]);
        for (int i = 0; i < 8; ++i) if (r[8 + i]) printf(" %lld", r[8 + i] - r[0]);
        printf("  kloop %lld  end %lld\n", r[3] - r[0], r[4] - r[0]);
    }
    return 0;
}
