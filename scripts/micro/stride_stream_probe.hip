// Probe (round 6): does the ROW-STRIDED shape of a projection tile's operand stream cost bandwidth against a contiguous one?
// A query-sized tile reads, per ring stage, a 512-byte (or 256-byte) piece of each of its 48 rows, rows `stride` bytes apart
// (K x 2: 6144 for fc2, 1536 for the K = 768 projections) -- 48 interleaved sequential streams.  Same bytes, three address patterns:
//   mode 0: contiguous (piece p of the workgroup's region at p x 1 KiB)
//   mode 1: the tile's pattern: stage s, rows r = 0..47: bytes [s x 512, s x 512 + 512) of row r at r x stride
//   mode 2: the same rows in a BLOCKED layout: stage-major, [stage][row][512 B] contiguous (what a re-laid-out weight copy would give)
// Every workgroup owns a private region of 48 rows x stride bytes (weights of its column tile); 8 waves, 6 KiB in flight per wave.
// Build: hipcc --offload-arch=gfx950 -O3 -o stride_stream_probe.bin scripts/micro/stride_stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(512) void stream(const char* __restrict__ src, long region, int stride, int mode, float* sink) {
    const int t = threadIdx.x;
    const char* base = src + (long)blockIdx.x * region;
    const int nstage = stride / 512;                  // 512-byte pieces per row
    float acc = 0.f;
    // per stage: 48 rows x 512 B = 24 KiB = 3 x 16 B per thread (chunk ci = j * 512 + t: row ci / 32, 16-byte position ci % 32)
    for (int s0 = 0; s0 < nstage; s0 += 2) {
        float4 v[6];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int s = s0 + u;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int ci = j * 512 + t, r = ci >> 5, pos = ci & 31;
                long off;
                if (mode == 0) off = ((long)s * 1536 + ci) * 16;
                else if (mode == 1) off = (long)r * stride + (long)s * 512 + pos * 16;
                else off = ((long)s * 48 + r) * 512 + pos * 16;
                v[u * 3 + j] = *reinterpret_cast<const float4*>(base + off);
            }
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) acc += v[j].x;
    }
    if (acc == 123456.f) sink[0] = acc;
}

int main() {
    int dev = 0, ncu = 0;
    CK(hipGetDevice(&dev)); CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    char* src; float* sink;
    const long total = 1l << 30;
    CK(hipMalloc((void**)&src, total)); CK(hipMemset(src, 1, total)); CK(hipMalloc((void**)&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[] = {"contiguous", "row-strided (the tile's pattern)", "blocked [stage][row][512 B]"};
    for (int stride : {6144, 1536, 8192, 16384}) {
        for (int nb : {48, 96, 256}) {
            const long region = 48l * stride;
            for (int mode = 0; mode < 3; ++mode) {
                float best = 1e9;
                for (int rep = 0; rep < 5; ++rep) {
                    // rotate over disjoint sets of regions so that a launch does not find its bytes in L2 (12 sets, like 12 blocks' weights)
                    const char* p = src + (long)(rep % 12) * nb * region;
                    CK(hipEventRecord(e0, 0));
                    hipLaunchKernelGGL(stream, dim3(nb), dim3(512), 0, 0, p, region, stride, mode, sink);
                    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep > 0 && ms < best) best = ms;
                }
                printf("row stride %5d B, %3d workgroups x %3ld KiB, %-34s %6.2f us  %6.1f GB/s per CU\n", stride, nb, region / 1024, names[mode], best * 1e3,
                       (double)region / (best * 1e-3) / 1e9);
            }
        }
    }
    return 0;
}
