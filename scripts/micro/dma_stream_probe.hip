// Probe (round 6): how fast can ONE workgroup per CU pull bytes into LDS -- the floor of a query- / mid-sized projection, whose
// workgroups stream (BM + BN) K 2 bytes each while the MFMA work is negligible.
//   mode 0: global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave instruction), `nw` of the 8 waves issuing, ring of `depth` KiB per wave
//   mode 1: global_load_dwordx4 into registers + ds_write_b128 (the register-staged path), 8 outstanding loads per wave
// Every workgroup reads its own contiguous region of `kib` KiB (regions are disjoint: the sum over workgroups is what the memory
// side serves); pass 1 comes from HBM (cold), later passes from MALL / L2 when the total fits.
// Build: hipcc --offload-arch=gfx950 -O3 -o dma_stream_probe.bin scripts/micro/dma_stream_probe.hip ; run: ./dma_stream_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef __attribute__((address_space(3))) char* lds_cptr_t;

__device__ __forceinline__ void dma16(const char* base_uniform, unsigned lane_off, unsigned dst_byte) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(lane_off), "s"(base_uniform), "s"(dst_byte) : "memory");
}

// each issuing wave walks its share of the region in 1-KiB pieces, `DEPTH` pieces in flight
template <int DEPTH>
__global__ __launch_bounds__(512) void dma_kernel(const char* __restrict__ src, long region_bytes, int nw, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    if (wave < nw) {
        const char* base = src + (long)blockIdx.x * region_bytes + (long)wave * (region_bytes / nw);
        const unsigned lds0 = (unsigned)(size_t)(lds_cptr_t)smem + (unsigned)wave * DEPTH * 1024;
        const long pieces = region_bytes / nw / 1024;
        for (long p = 0; p < pieces; ++p) {
            dma16(base + p * 1024, (unsigned)lane * 16, lds0 + (unsigned)(p % DEPTH) * 1024);
            if (p >= DEPTH - 1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(DEPTH - 1) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (threadIdx.x == 0) cyc[blockIdx.x] = (long long)__builtin_amdgcn_s_memtime() - t0;
}

__global__ __launch_bounds__(512) void reg_kernel(const char* __restrict__ src, long region_bytes, int nw, long long* cyc, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    float acc = 0.f;
    if (wave < nw) {
        const char* base = src + (long)blockIdx.x * region_bytes + (long)wave * (region_bytes / nw);
        const long pieces = region_bytes / nw / 1024;
        char* dst = smem + wave * 8 * 1024;
        for (long p = 0; p + 8 <= pieces; p += 8) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(base + (p + j) * 1024 + lane * 16);
#pragma unroll
            for (int j = 0; j < 8; ++j) *reinterpret_cast<float4*>(dst + j * 1024 + lane * 16) = v[j];
            acc += v[0].x;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) cyc[blockIdx.x] = (long long)__builtin_amdgcn_s_memtime() - t0;
    if (acc == 123456.f) sink[0] = acc;
}

int main(int argc, char** argv) {
    int dev = 0, ncu = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    const long max_total = 1l << 30;
    char* src; long long* cyc; float* sink;
    CK(hipMalloc((void**)&src, max_total)); CK(hipMemset(src, 1, max_total));
    CK(hipMalloc((void**)&cyc, ncu * 8)); CK(hipMalloc((void**)&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dma_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dma_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dma_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(reg_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    printf("CUs %d.  region = KiB per workgroup; rate = region x workgroups / launch time; per CU in bytes per shader-clock tick of workgroup 0\n", ncu);
    for (int nb : {ncu, 64}) {
        for (long kib : {256l, 1024l}) {
            const long region = kib * 1024;
            for (int mode = 0; mode < 4; ++mode) {
                for (int nw : {2, 4, 8}) {
                    float best = 1e9; long long c0 = 0;
                    for (int rep = 0; rep < 4; ++rep) {
                        CK(hipEventRecord(e0, 0));
                        if (mode == 0) hipLaunchKernelGGL(dma_kernel<4>, dim3(nb), dim3(512), 8 * 4 * 1024, 0, src, region, nw, cyc);
                        else if (mode == 1) hipLaunchKernelGGL(dma_kernel<8>, dim3(nb), dim3(512), 8 * 8 * 1024, 0, src, region, nw, cyc);
                        else if (mode == 2) hipLaunchKernelGGL(dma_kernel<16>, dim3(nb), dim3(512), 8 * 16 * 1024, 0, src, region, nw, cyc);
                        else hipLaunchKernelGGL(reg_kernel, dim3(nb), dim3(512), 8 * 8 * 1024, 0, src, region, nw, cyc, sink);
                        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                        if (rep > 0 && ms < best) { best = ms; CK(hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost)); }
                    }
                    const char* names[] = {"LDS-DMA, 4 KiB in flight per wave", "LDS-DMA, 8 KiB in flight per wave", "LDS-DMA, 16 KiB in flight per wave",
                                           "registers + ds_write, 8 KiB per wave"};
                    printf("%3d workgroups x %4ld KiB, %-38s %d waves: %7.2f us  %6.2f TB/s  %5.1f B/tick per CU\n", nb, kib, names[mode], nw,
                           best * 1e3, (double)region * nb / (best * 1e-3) / 1e12, (double)region / (double)(c0 ? c0 : 1));
                }
            }
        }
    }
    return 0;
}
