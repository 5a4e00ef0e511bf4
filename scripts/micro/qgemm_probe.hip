// Stand-alone tile sweep of the query- / mid-sized projection kernels (sgpt_amd/csrc/qgemm.hip): for M token rows, the four
// projections of an SGPT-125M block (LayerNorm + QKV, out-projection + residual, LayerNorm + fc1 + GELU, fc2 + residual), every
// candidate tile (QGemmArgs.tile) and the launcher's own choice (tile 0): microseconds per launch, back to back, the weights
// rotating over 12 copies (a forward walks 12 blocks: a launch finds its weights in the Infinity Cache, not in its L2).
// Build + run: scripts/gpu_qprobe.sh.   argv: M [M ...]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "common.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int d = 768, ffn = 3072, NL = 12, iters = 240;
    const int MMAX = 4096;
    void *A, *H, *O, *O2, *W[4]; float *x, *g, *b, *bias, *R;
    CK(hipMalloc(&A, (size_t)MMAX * d * 2)); CK(hipMalloc(&H, (size_t)MMAX * ffn * 2));
    CK(hipMalloc(&O, (size_t)MMAX * ffn * 4)); CK(hipMalloc(&O2, (size_t)MMAX * d * 4));
    const size_t wsz[4] = {(size_t)3 * d * d, (size_t)d * d, (size_t)ffn * d, (size_t)d * ffn};
    for (int i = 0; i < 4; ++i) { CK(hipMalloc(&W[i], wsz[i] * 2 * NL)); launch_fill_rand(W[i], (long)wsz[i] * NL, DT_F16, 2u + i, 0.05f, 0); }
    CK(hipMalloc((void**)&x, (size_t)MMAX * d * 4)); CK(hipMalloc((void**)&g, d * 4)); CK(hipMalloc((void**)&b, d * 4));
    CK(hipMalloc((void**)&bias, ffn * 4)); CK(hipMalloc((void**)&R, (size_t)MMAX * d * 4));
    launch_fill_rand(A, (long)MMAX * d, DT_F16, 1u, 1.0f, 0); launch_fill_rand(H, (long)MMAX * ffn, DT_F16, 9u, 1.0f, 0);
    launch_fill_rand(x, (long)MMAX * d, DT_F32, 3u, 1.0f, 0); launch_fill_rand(g, d, DT_F32, 4u, 1.0f, 0); launch_fill_rand(b, d, DT_F32, 5u, 0.1f, 0);
    launch_fill_rand(bias, ffn, DT_F32, 6u, 0.1f, 0); launch_fill_rand(R, (long)MMAX * d, DT_F32, 7u, 1.0f, 0);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[4] = {"LN+QKV  N 2304 K  768", "out+res N  768 K  768", "LN+fc1  N 3072 K  768", "fc2+res N  768 K 3072"};
    const char* plain_tiles[] = {"auto", "32x16", "32x32", "32x64", "64x32", "64x64", "128x64", "128x128"};
    const char* ln_tiles[] = {"auto", "32x32", "32x64", "64x64"};
    for (int ai = 1; ai < argc; ++ai) {
        const int M = atoi(argv[ai]);
        if (M <= 0 || M > MMAX || M % 32) continue;
        printf("M = %d token rows\n", M);
        for (int k = 0; k < 4; ++k) {
            const bool ln = k == 0 || k == 2;
            const int nt = ln ? 4 : 8;
            printf("  %s:", names[k]);
            for (int tile = 0; tile < nt; ++tile) {
                QGemmArgs q{};
                q.tile = tile;
                int epi, odt;
                auto setw = [&](int layer) {
                    q.g.W = (const char*)W[k] + wsz[k] * 2 * layer;
                };
                if (k == 0) { q.x = x; q.ln_g = g; q.ln_b = b; q.eps = 1e-5f; q.g.N = 3 * d; q.g.K = d; q.g.ldw = d; q.g.n_split = 2 * d; q.g.out = O; q.g.ldo = 2 * d; q.g.out2 = O2; q.g.ldo2 = M; epi = EPI_QKV; odt = DT_F16; }
                else if (k == 1) { q.g.A = A; q.g.lda = d; q.g.N = d; q.g.K = d; q.g.ldw = d; q.g.out = R; q.g.ldo = d; q.g.resid = R; q.g.bias = bias; epi = EPI_BIAS_RESID; odt = DT_F32; }
                else if (k == 2) { q.x = x; q.ln_g = g; q.ln_b = b; q.eps = 1e-5f; q.g.N = ffn; q.g.K = d; q.g.ldw = d; q.g.out = O; q.g.ldo = ffn; q.g.bias = bias; epi = EPI_BIAS_GELU; odt = DT_F16; }
                else { q.g.A = H; q.g.lda = ffn; q.g.N = d; q.g.K = ffn; q.g.ldw = ffn; q.g.out = R; q.g.ldo = d; q.g.resid = R; q.g.bias = bias; epi = EPI_BIAS_RESID; odt = DT_F32; }
                q.g.M = M; q.g.m_valid = M;
                setw(0);
                if (!launch_qgemm(DT_F16, epi, odt, q, 0)) { printf("  %s -", ln ? ln_tiles[tile] : plain_tiles[tile]); continue; }
                for (int i = 0; i < 12; ++i) { setw(i % NL); launch_qgemm(DT_F16, epi, odt, q, 0); }
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < iters; ++i) { setw(i % NL); launch_qgemm(DT_F16, epi, odt, q, 0); }
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("  %s %.1f", ln ? ln_tiles[tile] : plain_tiles[tile], ms / iters * 1e3);
            }
            printf("\n");
        }
    }
    return 0;
}
