// Does f32 VALU FMA work run concurrently with f32 MFMA work on one SIMD of gfx950?
// 8 waves per workgroup (2 per SIMD).  Roles by wave index: M = MFMA-only chain, V = VALU-only FMA
// chains.  Configs: MM (both waves MFMA), VV (both VALU), MV (one of each per SIMD).
// If the pipes are independent, MV finishes in ~max(M, V) with both streams at full rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ float mfma_work(const float *w, int iters, int lane)
{
    float a = w[lane], b = w[64 + lane];
    f32x16 acc[4];
    for (int i = 0; i < 4; i++) for (int q = 0; q < 16; q++) acc[i][q] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float r = 0;
    for (int i = 0; i < 4; i++) for (int q = 0; q < 16; q++) r += acc[i][q];
    return r;
}

__device__ __forceinline__ float valu_work(const float *w, int iters, int lane)
{
    float x[16], a = w[lane], b = w[64 + lane] * 1e-3f;
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = w[128 + i * 64 + lane];
    // one MFMA = 4096 flop = 64 lanes x 32 FMA: issue 32 v_fma per "MFMA equivalent"
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < 16; i++) x[i] = __builtin_fmaf(x[i], a, b);
    }
    float r = 0;
    for (int i = 0; i < 16; i++) r += x[i];
    return r;
}

// mode: 0 = all waves MFMA, 1 = all waves VALU, 2 = waves 0-3 MFMA + waves 4-7 VALU
__global__ __launch_bounds__(512, 2) void k(const float *w, float *out, int iters, int mode)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool is_mfma = mode == 0 || (mode == 2 && wave < 4);
    // wave-uniform branch
    float r;
    if (__builtin_amdgcn_readfirstlane(is_mfma)) r = mfma_work(w, iters, lane);
    else r = valu_work(w, iters, lane);          // per iteration: 4 MFMA-equivalents (4 x 32 FMAs = 128 v_fma)
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

int main()
{
    float *d_w, *d_out;
    CHK(hipMalloc(&d_w, 2048 * 4));
    CHK(hipMalloc(&d_out, 256 * 512 * 4));
    float h[2048];
    for (int i = 0; i < 2048; i++) h[i] = 0.5f + 0.0001f * (i % 97);
    CHK(hipMemcpy(d_w, h, sizeof h, hipMemcpyHostToDevice));
    const int iters = 20000;
    const char *names[3] = {"MM (2 MFMA waves/SIMD)", "VV (2 VALU waves/SIMD)", "MV (1 MFMA + 1 VALU wave/SIMD)"};
    for (int rep = 0; rep < 2; rep++)
        for (int mode = 0; mode < 3; mode++) {
            hipEvent_t a, b;
            CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d_w, d_out, 100, mode);
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(a));
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d_w, d_out, iters, mode);
            CHK(hipEventRecord(b));
            CHK(hipEventSynchronize(b));
            float ms;
            CHK(hipEventElapsedTime(&ms, a, b));
            // flop per wave = iters * 4 * 4096 in both roles
            const double waves = 256.0 * 8, flop = waves * iters * 4.0 * 4096.0;
            printf("%-32s %.3f ms  total %.1f TFLOP/s (f32)\n", names[mode], ms, flop / (ms * 1e-3) / 1e12);
        }
    return 0;
}
