// Probe (round 6): the corpus stream of the short-query scorer (score64_kernel: 256 documents x 64 k-elements per step = a 128-byte
// piece of each of 256 rows, rows 1536 B apart, 12 steps per tile) against other ways of walking the same 1.5 GB:
//   mode 0: the scorer's pattern (128-byte pieces, 12 passes over a 384-KiB tile)
//   mode 1: 512-byte pieces (KD = 256: 3 passes per tile)
//   mode 2: whole rows (each wave instruction reads 1 KiB contiguous: 2/3 of a row; one pass)
// Persistent grid of one workgroup per CU over 1 M rows x 1536 B, 8 waves, `depth` x 16 B in flight per thread.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int DEPTH>
__global__ __launch_bounds__(512) void walk(const char* __restrict__ src, long n_tiles, int mode, float* sink) {
    const int t = threadIdx.x;
    float acc = 0.f;
    for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const char* base = src + tile * (256l * 1536);
        // 384 KiB per tile = 24576 chunks of 16 B = 48 per thread
        for (int c0 = 0; c0 < 48; c0 += DEPTH) {
            float4 v[DEPTH];
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                const int ci = (c0 + u) * 512 + t;                     // chunk index in walk order
                long off;
                if (mode == 0) {          // step s = ci / 2048 (2048 chunks = 256 rows x 8), row = (ci % 2048) / 8, pos = ci % 8
                    const int s = ci >> 11, r = (ci & 2047) >> 3, pos = ci & 7;
                    off = (long)r * 1536 + s * 128 + pos * 16;
                } else if (mode == 1) {   // step s = ci / 8192 (256 rows x 32), row = (ci % 8192) / 32, pos = ci % 32
                    const int s = ci >> 13, r = (ci & 8191) >> 5, pos = ci & 31;
                    off = (long)r * 1536 + s * 512 + pos * 16;
                } else {
                    off = (long)ci * 16;
                }
                v[u] = *reinterpret_cast<const float4*>(base + off);
            }
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) acc += v[u].x;
        }
    }
    if (acc == 123456.f) sink[0] = acc;
}

int main() {
    int dev = 0, ncu = 0;
    CK(hipGetDevice(&dev)); CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    const long n_tiles = 3906, bytes = n_tiles * 256l * 1536;      // 1 M documents x 768 x 2 B
    char* src; float* sink;
    CK(hipMalloc((void**)&src, bytes)); CK(hipMemset(src, 1, bytes)); CK(hipMalloc((void**)&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[] = {"128-byte pieces, 12 passes (score64)", "512-byte pieces, 3 passes", "contiguous"};
    for (int depth : {8, 16}) {
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(e0, 0));
                if (depth == 8) hipLaunchKernelGGL(walk<8>, dim3(ncu), dim3(512), 0, 0, src, n_tiles, mode, sink);
                else hipLaunchKernelGGL(walk<16>, dim3(ncu), dim3(512), 0, 0, src, n_tiles, mode, sink);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            printf("%2d loads in flight per thread, %-40s %7.1f us  %5.2f TB/s\n", depth, names[mode], best * 1e3, (double)bytes / (best * 1e-3) / 1e12);
        }
    }
    return 0;
}
