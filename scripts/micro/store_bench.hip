// Micro-benchmark: non-temporal bf16 tile stores in the shapes a GEMM epilogue can produce, [T][N] row-major output,
// 256x256 tiles walked like the persistent GEMM (workgroup b: tiles b, b+256, ...), 8 waves per workgroup.
//   mode 0: wave sub-tile 128 rows x 64 cols : per instruction 8 rows x 128 B   (today's epilogue)
//   mode 1: wave sub-tile  64 rows x 128 cols: per instruction 4 rows x 256 B
//   mode 2: wave sub-tile  32 rows x 256 cols: per instruction 2 rows x 512 B
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void st(uint16_t* __restrict__ out, int T, int N, int mode) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles_n = N / 256, tiles = (T / 256) * tiles_n;
    const u32x4_t v = {(uint32_t)threadIdx.x, 2u, 3u, 4u};
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 256;
        if (mode == 0) {
            const int wm = wave >> 2, wn = wave & 3, rrow = lane >> 3, ch = lane & 7;
            for (int i = 0; i < 16; ++i) {
                const long m = m0 + wm * 128 + i * 8 + rrow;
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(out + m * N + n0 + wn * 64 + ch * 8));
            }
        } else if (mode == 1) {
            const int wm = wave >> 1, wn = wave & 1, rrow = lane >> 4, ch = lane & 15;
            for (int i = 0; i < 16; ++i) {
                const long m = m0 + wm * 64 + i * 4 + rrow;
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(out + m * N + n0 + wn * 128 + ch * 8));
            }
        } else {
            const int rrow = lane >> 5, ch = lane & 31;
            for (int i = 0; i < 16; ++i) {
                const long m = m0 + wave * 32 + i * 2 + rrow;
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(out + m * N + n0 + ch * 8));
            }
        }
    }
}

int main() {
    const int T = 131072;
    uint16_t* o;
    hipMalloc(&o, (size_t)T * 3072 * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int N : {1536, 3072})
        for (int mode = 0; mode < 3; ++mode) {
            for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(st, dim3(256), dim3(512), 0, 0, o, T, N, mode);
            hipEventRecord(e0, 0);
            for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(st, dim3(256), dim3(512), 0, 0, o, T, N, mode);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            ms /= 10;
            printf("N=%d mode %d: %7.1f us  %.2f TB/s\n", N, mode, ms * 1e3, (double)T * N * 2 / (ms * 1e-3) / 1e12);
        }
    return 0;
}
