// Microbenchmark: random LDS gathers from a 20-entry bag, as pool_table_kernel does.
// Variants: element width (b32 / b64 / b128) and entry stride.  Prints ns per wave-gather and
// effective gathers/clk/CU.  build: hipcc --offload-arch=gfx950 -O3 lds_gather_bench.hip -o lds_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename T, int STRIDE_B, int NIDX>
__global__ __launch_bounds__(256) void gather(const unsigned *idx, float *out, int iters, int n_entries)
{
    __shared__ __attribute__((aligned(256))) char lds[4][4096];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    char *base = lds[w];
    for (int i = lane; i < 1024; i += 64) ((float *)base)[i] = 1.0f + 1e-7f * i;
    unsigned o[NIDX];
#pragma unroll
    for (int i = 0; i < NIDX; i++) o[i] = (idx[(blockIdx.x * 256 + threadIdx.x) * NIDX + i] % n_entries) * STRIDE_B;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    T acc;
    float *af = (float *)&acc;
    for (unsigned k = 0; k < sizeof(T) / 4; k++) af[k] = 1.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NIDX; i++) {
            asm volatile("" : "+v"(o[i]));          // opaque: no hoisting of the LDS load
            const T v = *(const T *)(base + o[i]);
            const float *vf = (const float *)&v;
#pragma unroll
            for (unsigned k = 0; k < sizeof(T) / 4; k++) af[k] *= vf[k];
        }
    }
    float r = 0;
    for (unsigned k = 0; k < sizeof(T) / 4; k++) r += af[k];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <typename T, int STRIDE_B>
void run(const char *name, int n_entries, int blocks, const unsigned *d_idx, float *d_out)
{
    constexpr int NIDX = 20;
    const int iters = 2000;
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    hipLaunchKernelGGL((gather<T, STRIDE_B, NIDX>), dim3(blocks), dim3(256), 0, 0, d_idx, d_out, 10, n_entries);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a));
    hipLaunchKernelGGL((gather<T, STRIDE_B, NIDX>), dim3(blocks), dim3(256), 0, 0, d_idx, d_out, iters, n_entries);
    CHK(hipEventRecord(b));
    CHK(hipEventSynchronize(b));
    float ms;
    CHK(hipEventElapsedTime(&ms, a, b));
    const double wave_gathers = (double)blocks * 4 * iters * NIDX;
    const double per_cu_per_clk = wave_gathers / 256.0 / (ms * 1e-3 * 2.2e9);   // wave-instr / clk / CU at ~2.2 GHz
    printf("%-34s entries=%2d blocks=%5d  %.3f ms  %.3f wave-gathers/clk/CU  -> %.1f clk per wave-gather (lane-elems/clk/CU %.1f)\n",
           name, n_entries, blocks, ms, per_cu_per_clk, 1.0 / per_cu_per_clk, per_cu_per_clk * 64 * (sizeof(T) / 4));
}

int main()
{
    const int max_blocks = 256 * 8;
    std::vector<unsigned> h((size_t)max_blocks * 256 * 20);
    unsigned s = 12345;
    for (auto &x : h) { s = s * 1664525u + 1013904223u; x = s >> 8; }
    unsigned *d_idx; float *d_out;
    CHK(hipMalloc(&d_idx, h.size() * 4));
    CHK(hipMalloc(&d_out, (size_t)max_blocks * 256 * 4));
    CHK(hipMemcpy(d_idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (int blocks : {256 * 2, 256 * 4, 256 * 8}) {
        run<float, 4>("b32 stride4 (20 entries)", 20, blocks, d_idx, d_out);
        run<float2, 8>("b64 stride8 (20 entries)", 20, blocks, d_idx, d_out);
        run<float2, 8>("b64 stride8 (1 entry: broadcast)", 1, blocks, d_idx, d_out);
        run<float2, 8>("b64 stride8 (32 entries)", 32, blocks, d_idx, d_out);
        run<float4, 16>("b128 stride16 (16 entries)", 16, blocks, d_idx, d_out);
        run<float4, 16>("b128 stride16 (20 entries)", 20, blocks, d_idx, d_out);
        run<float, 4>("b32 stride4 (64 entries)", 64, blocks, d_idx, d_out);
    }
    return 0;
}
