// Probe for a persistent query-sized forward (round 6): what does a device-wide phase boundary cost INSIDE one kernel, against
// the kernel boundary it would replace?
//   1. grid barrier across one workgroup per CU (monotonic counter in device memory; release fence + atomic add + acquire spin),
//      flat and two-level (one counter per XCD slot -> one global counter);
//   2. the same with a hand-off: every workgroup writes 3 KiB before the barrier and reads 48 KiB written by OTHER workgroups
//      (other XCDs' L2s) after it -- the A panel of a 32-row projection -- and checks it;
//   3. back-to-back launches of an empty kernel and of the same write / read pair as two dependent kernels.
// Every spin is bounded (s_memtime budget): a barrier that cannot complete sets a flag and the kernel ends.
// Build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier_probe.bin scripts/micro/grid_barrier_probe.hip ; run: ./grid_barrier_probe.bin [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Bar {
    unsigned* global;      // one counter
    unsigned* xcd;         // 8 counters, 64 B apart
    int* failed;
};

__device__ __forceinline__ bool spin_until(unsigned* p, unsigned target, int* failed) {
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((long long)__builtin_amdgcn_s_memtime() - t0 > (1ll << 28)) { *failed = 1; return false; }
    }
    return true;
}

// mode 0: flat, fences; 1: flat, no fences (atomics only); 2: two-level (XCD slot = blockIdx & 7), fences
template <int MODE>
__device__ __forceinline__ void grid_barrier(const Bar& b, unsigned& epoch, unsigned nblocks) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (MODE != 1) __atomic_thread_fence(__ATOMIC_RELEASE);      // hipcc: agent-scope release = buffer_wbl2 sc1 + waits
        epoch += 1;
        if (MODE == 2) {
            unsigned* xc = b.xcd + (blockIdx.x & 7) * 16;
            const unsigned per = nblocks / 8;
            const unsigned old = __hip_atomic_fetch_add(xc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == epoch * per) __hip_atomic_fetch_add(b.global, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            spin_until(b.global, epoch * 8, b.failed);
        } else {
            __hip_atomic_fetch_add(b.global, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            spin_until(b.global, epoch * nblocks, b.failed);
        }
        if (MODE != 1) __atomic_thread_fence(__ATOMIC_ACQUIRE);      // buffer_inv sc1
    }
    __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(512) void barrier_only(Bar b, int iters, long long* cyc) {
    unsigned epoch = 0;
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) grid_barrier<MODE>(b, epoch, gridDim.x);
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = (long long)__builtin_amdgcn_s_memtime() - t0;
}

// hand-off: phase A: workgroup w writes rows [w * ROWS_W, ...) of a [nblocks * 3 KiB] buffer (value = iteration + index);
// barrier; phase B: every workgroup reads 48 KiB starting at another XCD's region and sums it; barrier (buffer reuse).
__global__ __launch_bounds__(512) void handoff(Bar b, int iters, float* buf, float* sums, long long* cyc, int* bad) {
    unsigned epoch = 0;
    const int t = threadIdx.x, nb = gridDim.x;
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    float acc = 0.f;
    for (int i = 0; i < iters; ++i) {
        // 3 KiB = 768 floats per workgroup
        for (int j = t; j < 768; j += 512) buf[(long)blockIdx.x * 768 + j] = (float)(i + 1);
        grid_barrier<0>(b, epoch, nb);
        // 48 KiB = 12288 floats = 16 workgroups' regions, starting 3 workgroups away (another XCD: XCD = block & 7)
        float s = 0.f;
        const long start = ((long)blockIdx.x + 3) % nb * 768;
        const long total = (long)nb * 768;
        for (int j = t; j < 12288; j += 512) s += buf[(start + j) % total];
        acc += s;
        // check: every value read must be i + 1
        const float want = (float)(i + 1) * (12288 / 512);
        if (s != want) atomicAdd(bad, 1);
        grid_barrier<0>(b, epoch, nb);
    }
    if (t == 0) sums[blockIdx.x] = acc;
    if (blockIdx.x == 0 && t == 0) cyc[0] = (long long)__builtin_amdgcn_s_memtime() - t0;
}

__global__ __launch_bounds__(512) void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ __launch_bounds__(512) void write_kernel(float* buf, int it) {
    for (int j = threadIdx.x; j < 768; j += 512) buf[(long)blockIdx.x * 768 + j] = (float)(it + 1);
}
__global__ __launch_bounds__(512) void read_kernel(const float* buf, float* sums, int it, int* bad) {
    const int t = threadIdx.x, nb = gridDim.x;
    float s = 0.f;
    const long start = ((long)blockIdx.x + 3) % nb * 768, total = (long)nb * 768;
    for (int j = t; j < 12288; j += 512) s += buf[(start + j) % total];
    if (s != (float)(it + 1) * (12288 / 512)) atomicAdd(bad, 1);
    if (t == 0) sums[blockIdx.x] = s;
}

template <typename K, typename... Args>
static int coop(K kern, int nblocks, Args... args) {
    void* params[] = {(void*)&args...};
    hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(kern), dim3(nblocks), dim3(512), params, 0, 0);
    if (e != hipSuccess) { fprintf(stderr, "cooperative launch: %s\n", hipGetErrorString(e)); return 1; }
    return 0;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    int dev = 0, ncu = 0, clk = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    CK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, dev));
    printf("CUs %d, clock %d kHz\n", ncu, clk);
    unsigned *gl, *xc; int *failed, *bad; long long* cyc; float *buf, *sums;
    CK(hipMalloc((void**)&gl, 256)); CK(hipMalloc((void**)&xc, 8 * 64)); CK(hipMalloc((void**)&failed, 4)); CK(hipMalloc((void**)&bad, 4));
    CK(hipMalloc((void**)&cyc, 64)); CK(hipMalloc((void**)&buf, (size_t)ncu * 768 * 4)); CK(hipMalloc((void**)&sums, (size_t)ncu * 4));
    Bar b{gl, xc, failed};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto reset = [&]() { hipMemset(gl, 0, 256); hipMemset(xc, 0, 512); hipMemset(failed, 0, 4); hipMemset(bad, 0, 4); hipDeviceSynchronize(); };
    auto report = [&](const char* name, int per_iter) -> int {
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        long long c = 0; int f = 0, bd = 0;
        CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&f, failed, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&bd, bad, 4, hipMemcpyDeviceToHost));
        printf("%-58s %8.2f us per iteration (%d barrier(s) each; %lld s_memtime ticks per iteration)%s%s\n", name, ms / iters * 1e3, per_iter,
               c / iters, f ? "  BARRIER TIMED OUT" : "", bd ? "  STALE DATA READ" : "");
        return 0;
    };
    for (int nb : {ncu, ncu / 2, 32}) {
        printf("-- %d workgroups of 512 threads\n", nb);
        reset(); CK(hipEventRecord(e0, 0)); if (coop(barrier_only<0>, nb, b, iters, cyc)) return 1; CK(hipEventRecord(e1, 0));
        report("grid barrier, flat counter, release + acquire fences", 1);
        reset(); CK(hipEventRecord(e0, 0)); if (coop(barrier_only<1>, nb, b, iters, cyc)) return 1; CK(hipEventRecord(e1, 0));
        report("grid barrier, flat counter, atomics only (no fences)", 1);
        if (nb % 8 == 0) {
            reset(); CK(hipEventRecord(e0, 0)); if (coop(barrier_only<2>, nb, b, iters, cyc)) return 1; CK(hipEventRecord(e1, 0));
            report("grid barrier, two-level (8 slots), fences", 1);
        }
        reset(); CK(hipEventRecord(e0, 0)); if (coop(handoff, nb, b, iters, buf, sums, cyc, bad)) return 1; CK(hipEventRecord(e1, 0));
        report("write 3 KiB | barrier | read 48 KiB of others | barrier", 2);
        // kernel boundaries instead
        reset(); CK(hipMemset(cyc, 0, 8));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(empty_kernel, dim3(nb), dim3(512), 0, 0, (int*)nullptr);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(empty_kernel, dim3(nb), dim3(512), 0, 0, (int*)nullptr);
        CK(hipEventRecord(e1, 0));
        report("empty kernel, back-to-back launches", 0);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) {
            hipLaunchKernelGGL(write_kernel, dim3(nb), dim3(512), 0, 0, buf, i);
            hipLaunchKernelGGL(read_kernel, dim3(nb), dim3(512), 0, 0, buf, sums, i, bad);
        }
        CK(hipEventRecord(e1, 0));
        report("write kernel -> read kernel (two dependent launches)", 0);
    }
    return 0;
}
