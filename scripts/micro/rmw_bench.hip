// Micro-benchmark: fp32 read-modify-write of x[T][768] in the access shapes a GEMM residual epilogue can use.
//   mode 0: per wave-instruction 4 rows x 256 B (lane>>4 = row, lane&15 = 16-B chunk)  -- today's per-wave epilogue
//   mode 1: per wave-instruction 1 row x 1 KiB                                          -- WG-level staging
//   mode 2: per wave-instruction 1 KiB contiguous, whole 3 KiB rows per wave            -- LayerNorm-like streaming
// Tiles are visited like the persistent GEMM does: workgroup b handles 256x256 tiles b, b+256, ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(512) void rmw(float* __restrict__ x, int T, int d, int mode, int tiles_n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles_m = T / 256, tiles = tiles_m * tiles_n;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
        const int m0 = tm * 256, n0 = tn * 256;
        if (mode == 0) {
            const int wm = wave >> 2, wn = wave & 3;          // 2 x 4 waves, 128 x 64 each
            const int rrow = lane >> 4, rchunk = lane & 15;
            float* base = x + (long)(m0 + wm * 128 + rrow) * d + n0 + wn * 64 + rchunk * 4;
#pragma unroll 2
            for (int i = 0; i < 8; ++i) {
                float4 v[4];
#pragma unroll
                for (int h = 0; h < 4; ++h) v[h] = *reinterpret_cast<const float4*>(base + (long)(i * 16 + h * 4) * d);
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    v[h].x += 1.f; v[h].y += 1.f; v[h].z += 1.f; v[h].w += 1.f;
                    *reinterpret_cast<float4*>(base + (long)(i * 16 + h * 4) * d) = v[h];
                }
            }
        } else if (mode == 1) {
            float* base = x + (long)(m0 + wave * 32) * d + n0 + lane * 4;   // 8 waves x 32 rows, 1 KiB per instruction
#pragma unroll 2
            for (int i = 0; i < 8; ++i) {
                float4 v[4];
#pragma unroll
                for (int h = 0; h < 4; ++h) v[h] = *reinterpret_cast<const float4*>(base + (long)(i * 4 + h) * d);
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    v[h].x += 1.f; v[h].y += 1.f; v[h].z += 1.f; v[h].w += 1.f;
                    *reinterpret_cast<float4*>(base + (long)(i * 4 + h) * d) = v[h];
                }
            }
        }
    }
    if (mode == 2) {   // streaming: whole rows, one wave per row
        for (long row = (long)blockIdx.x * 8 + wave; row < T; row += (long)gridDim.x * 8) {
            float* r = x + row * d;
            float4 v[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = *reinterpret_cast<const float4*>(r + c * 256 + lane * 4);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                v[c].x += 1.f; v[c].y += 1.f; v[c].z += 1.f; v[c].w += 1.f;
                *reinterpret_cast<float4*>(r + c * 256 + lane * 4) = v[c];
            }
        }
    }
}

int main() {
    const int T = 131072, d = 768;
    float* x;
    hipMalloc(&x, (size_t)T * d * 4);
    hipMemset(x, 0, (size_t)T * d * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode)
        for (int grid : {256, 512, 1024}) {
            for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(rmw, dim3(grid), dim3(512), 0, 0, x, T, d, mode, 3);
            hipEventRecord(e0, 0);
            for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(rmw, dim3(grid), dim3(512), 0, 0, x, T, d, mode, 3);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            ms /= 10;
            printf("mode %d grid %4d: %7.1f us  %.2f TB/s (read+write)\n", mode, grid, ms * 1e3, 2.0 * T * d * 4 / (ms * 1e-3) / 1e12);
        }
    return 0;
}
