// Microbenchmark for the encoder's MFMA pattern on gfx950: how many cycles per
// v_mfma_f32_32x32x2_f32 does each way of chaining cost?  Each "tile" = 40 layer-1 MFMAs into
// 5 accumulators + 80 layer-2 MFMAs whose B operand is relu(layer-1 accumulator register).
//   V0 relu inline (v_max between dependent MFMAs)          -- what enc_kernel does
//   V1 relu batched: 16 v_max first, then 16 MFMAs back to back
//   V2 like V1 with two layer-2 accumulators (even/odd unit tiles), summed at the end
//   V3 pure MFMA: no relu at all (upper bound)
// build: hipcc --offload-arch=gfx950 -O3 mfma_chain_bench.hip -o mfma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int V, int WPS>
__global__ __launch_bounds__(256, WPS) void chain(const float *w, float *out, int tiles)
{
    const int lane = threadIdx.x & 63;
    float w1[40], w2[80], f[8];
#pragma unroll
    for (int i = 0; i < 40; i++) w1[i] = w[i * 64 + lane];
#pragma unroll
    for (int i = 0; i < 80; i++) w2[i] = w[(40 + i) * 64 + lane];
#pragma unroll
    for (int i = 0; i < 8; i++) f[i] = w[(120 + i) * 64 + lane];
    float zsum = 0.f;
    for (int t = 0; t < tiles; ++t) {
        f32x16 acc2, acc2b, h1a, h1b;
#pragma unroll
        for (int q = 0; q < 16; q++) { acc2[q] = 0.f; acc2b[q] = 0.f; h1a[q] = 0.f; }
#pragma unroll
        for (int st = 0; st < 8; st++) h1a = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[st], f[st], h1a, 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 5; m++) {
            f32x16 &cur = (m & 1) ? h1b : h1a;
            f32x16 &nxt = (m & 1) ? h1a : h1b;
            if (m < 4) {
#pragma unroll
                for (int q = 0; q < 16; q++) nxt[q] = 0.f;
#pragma unroll
                for (int st = 0; st < 8; st++)
                    nxt = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[(m + 1) * 8 + st], f[st], nxt, 0, 0, 0);
            }
            if (V == 0) {
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const float hq = fmaxf(cur[q], 0.f);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[m * 16 + q], hq, acc2, 0, 0, 0);
                }
            } else if (V == 1 || V == 2) {
                float h[16];
#pragma unroll
                for (int q = 0; q < 16; q++) h[q] = __builtin_fmaxf(cur[q], 0.f);
                __builtin_amdgcn_sched_barrier(0);
                f32x16 &a2 = (V == 2 && (m & 1)) ? acc2b : acc2;
#pragma unroll
                for (int q = 0; q < 16; q++)
                    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[m * 16 + q], h[q], a2, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            } else {
#pragma unroll
                for (int q = 0; q < 16; q++)
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[m * 16 + q], cur[q], acc2, 0, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < 16; q++) zsum += acc2[q] + acc2b[q];
        f[0] += zsum * 1e-30f;     // loop-carried so tiles cannot be merged or hoisted
    }
    out[blockIdx.x * 256 + threadIdx.x] = zsum;
}

template <int V, int WPS>
void run(const char *name, const float *d_w, float *d_out)
{
    const int tiles = 300;
    const int blocks = 256 * WPS;        // WPS blocks of 4 waves per CU
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    hipLaunchKernelGGL((chain<V, WPS>), dim3(blocks), dim3(256), 0, 0, d_w, d_out, 10);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a));
    hipLaunchKernelGGL((chain<V, WPS>), dim3(blocks), dim3(256), 0, 0, d_w, d_out, tiles);
    CHK(hipEventRecord(b));
    CHK(hipEventSynchronize(b));
    float ms;
    CHK(hipEventElapsedTime(&ms, a, b));
    const double mfma = (double)blocks * 4 * tiles * 120;
    const double tflops = mfma * 4096 / (ms * 1e-3) / 1e12;
    printf("%-44s waves/SIMD=%d  %.3f ms  %.1f TFLOP/s executed (%.0f%% of 157.3)  ns/tile/wave %.0f\n", name, WPS, ms, tflops,
           100 * tflops / 157.3, ms * 1e6 / tiles);
}

int main()
{
    float *d_w, *d_out;
    CHK(hipMalloc(&d_w, 136 * 64 * 4));
    CHK(hipMalloc(&d_out, 256 * 8 * 256 * 4));
    float h[136 * 64];
    for (int i = 0; i < 136 * 64; i++) h[i] = 0.01f * ((i * 37) % 19 - 9);
    CHK(hipMemcpy(d_w, h, sizeof h, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; rep++) {
        run<0, 1>("V0 relu inline", d_w, d_out);
        run<0, 2>("V0 relu inline", d_w, d_out);
        run<1, 1>("V1 relu batched, MFMAs back to back", d_w, d_out);
        run<1, 2>("V1 relu batched, MFMAs back to back", d_w, d_out);
        run<2, 1>("V2 batched + two layer-2 accumulators", d_w, d_out);
        run<2, 2>("V2 batched + two layer-2 accumulators", d_w, d_out);
        run<3, 1>("V3 no relu (upper bound)", d_w, d_out);
        run<3, 2>("V3 no relu (upper bound)", d_w, d_out);
    }
    return 0;
}
