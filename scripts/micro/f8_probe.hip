// Probe of v_mfma_f32_16x16x128_f8f6f4 (unscaled, OCP e4m3 x e4m3) on gfx950: operand lane/byte layout and issue rate.
// hipcc --offload-arch=gfx950 -O3 scripts/micro/f8_probe.hip -o gpurun_in/f8_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void one(const uint8_t* A, const uint8_t* B, float* D) {   // A [16][128], B [16][128] (n-major), D [16][16]
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    i32x8 a = *reinterpret_cast<const i32x8*>(A + r * 128 + 32 * g);
    i32x8 b = *reinterpret_cast<const i32x8*>(B + r * 128 + 32 * g);
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
    for (int e = 0; e < 4; ++e) D[(4 * g + e) * 16 + r] = c[e];     // assumed: row = 4 (lane >> 4) + e (A rows), col = lane & 15 (B rows)
}

template <int F8>
__global__ __launch_bounds__(256) void rate(float* out, long long* cyc, int n) {
    const int l = threadIdx.x;
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + l * 0x01010101 * (i & 1); b[i] = 0x3c343c34 ^ (l << 8); }
    f32x4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = f32x4{0, 0, 0, 0};
    typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
    typedef __attribute__((ext_vector_type(4))) int i32x4;
    i32x4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (F8) c[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[i], 0, 0, 0, 0, 0, 0);
            else c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a4), __builtin_bit_cast(bf16x8, b4), c[i], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][3];
    out[blockIdx.x * 256 + l] = s;
    if (l == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

static uint8_t enc(int v) {   // small integers -> e4m3fn (exact for |v| <= 16)
    if (v == 0) return 0;
    uint8_t s = v < 0 ? 0x80 : 0; v = std::abs(v);
    int e = 0; while ((v >> (e + 1)) != 0) ++e;           // v = 1.m * 2^e
    int m = ((v << 3) >> e) & 7;
    return s | (uint8_t)(((e + 7) << 3) | m);
}

int main() {
    std::vector<uint8_t> A(16 * 128), B(16 * 128);
    std::vector<int> Ai(16 * 128), Bi(16 * 128);
    for (int r = 0; r < 16; ++r) for (int k = 0; k < 128; ++k) {
        Ai[r * 128 + k] = ((r * 7 + k * 3) % 9) - 4; Bi[r * 128 + k] = ((r * 5 + k * 11) % 13) - 6;
        A[r * 128 + k] = enc(Ai[r * 128 + k]); B[r * 128 + k] = enc(Bi[r * 128 + k]);
    }
    uint8_t *dA, *dB; float* dD;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 256 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    one<<<1, 64>>>(dA, dB, dD);
    float D[256]; hipMemcpy(D, dD, sizeof(D), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
        long ref = 0; for (int k = 0; k < 128; ++k) ref += (long)Ai[m * 128 + k] * Bi[n * 128 + k];
        if (D[m * 16 + n] != (float)ref) { if (bad < 5) printf("mismatch D[%d][%d] = %g want %ld\n", m, n, D[m * 16 + n], ref); ++bad; }
    }
    printf("layout probe (lane r = l & 15 holds 32 consecutive k of row r at k = 32 (l >> 4); D row = 4 (l >> 4) + e, col = l & 15): %s (%d mismatches)\n",
           bad ? "WRONG" : "OK", bad);
    float* dO; long long* dC; hipMalloc(&dO, 1024 * 256 * 4); hipMalloc(&dC, 8);
    for (int f8 = 0; f8 < 2; ++f8) for (int blocks : {1, 256, 512}) {
        const int n = 2000; long long cyc = 0;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        if (f8) rate<1><<<blocks, 256>>>(dO, dC, 10); else rate<0><<<blocks, 256>>>(dO, dC, 10);
        hipEventRecord(e0);
        if (f8) rate<1><<<blocks, 256>>>(dO, dC, n); else rate<0><<<blocks, 256>>>(dO, dC, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&cyc, dC, 8, hipMemcpyDeviceToHost);
        const double flop = 2.0 * 16 * 16 * (f8 ? 128 : 32) * 8.0 * n * 4 * blocks;
        printf("%s blocks=%d (1 wave/SIMD x %d): %.1f cycles per MFMA per wave, %.1f TFLOP/s\n", f8 ? "f8f6f4 16x16x128" : "bf16   16x16x32 ",
               blocks, blocks > 256 ? 2 : 1, (double)cyc / (8.0 * n), flop / (ms * 1e-3) / 1e12);
    }
    return bad != 0;
}
