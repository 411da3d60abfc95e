// Probes behind the ragged index-table pooling kernel (pool_rtab_kernel):
//   1. 2-byte-aligned global_load_dwordx4 / dwordx2 (the u16 index rows start at any even byte): correct?
//   2. ds_read_b32 gathers at random indices of an n-entry bag, n = 20..1000: LDS cycles per wave-gather
//      (bank conflicts of random addresses), 8 waves per SIMD.
//   3. ds_bpermute_b32 rate (a conflict-free crossbar gather for bags <= 64).
//   4. (round 5) the bag stored TWICE, the copy picked by lane parity: copy B rotated by 16 banks
//      (MODE 2) or by 0 banks at +64 dwords (MODE 3), and the pair layout read with ds_read_b64 + select (MODE 4: entries
//      i and i + 32 side by side, so any bag <= 64 touches each bank pair once).  Do any of them beat one copy for 32 < n <= 64?
// build: hipcc --offload-arch=gfx950 -O3 ragged_gather_probe.hip -o rg_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct __attribute__((packed, aligned(2))) U16x20 { uint16_t v[20]; };

__global__ void unaligned_kernel(const uint16_t *tab, int start, uint32_t *out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const U16x20 r = *(const U16x20 *)(tab + start + 20 * t);
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 20; k++) s = s * 31u + r.v[k];
    out[t] = s;
}

template <int MODE>
__global__ __launch_bounds__(256) void gather_kernel(const uint32_t *idx, float *out, int iters, int n)
{
    __shared__ float lds[4][1024];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float *bag = lds[w];
    for (int i = lane; i < 1024; i += 64) bag[i] = 1.0f + 1e-7f * i;
    uint32_t o[20];
#pragma unroll
    for (int i = 0; i < 20; i++) {
        o[i] = idx[(blockIdx.x * 256 + threadIdx.x) * 20 + i] % (uint32_t)n;
        if (MODE == 2) o[i] = (lane & 1) ? 64 + ((o[i] + 16) & 63) : o[i];      // copy B at +64 dwords, rotated by 16 banks
        if (MODE == 3) o[i] = (lane & 1) ? 64 + o[i] : o[i];                    // copy B at +64 dwords, same banks
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    float acc = 1.0f;
    const float mine = bag[lane];
    for (int it = 0; it < iters; ++it) {
        float g[20];
#pragma unroll
        for (int i = 0; i < 20; i++) {
            asm volatile("" : "+v"(o[i]));
            if (MODE == 0 || MODE == 2 || MODE == 3) g[i] = bag[o[i]];
            else if (MODE == 4) {                                               // pair (i % 32) holds entries i % 32 and i % 32 + 32
                const float2 pr = *(const float2 *)(bag + 2 * (o[i] & 31));
                g[i] = (o[i] & 32) ? pr.y : pr.x;
            } else g[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)(o[i] << 2), __builtin_bit_cast(int, mine)));
        }
#pragma unroll
        for (int i = 0; i < 20; i++) acc *= g[i];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main()
{
    // ---- 1. unaligned loads
    {
        const int N = 64 * 20 + 64;
        std::vector<uint16_t> h(N);
        for (int i = 0; i < N; i++) h[i] = (uint16_t)(i * 2654435761u >> 16);
        uint16_t *d; uint32_t *o;
        CHK(hipMalloc(&d, N * 2)); CHK(hipMalloc(&o, 64 * 4));
        CHK(hipMemcpy(d, h.data(), N * 2, hipMemcpyHostToDevice));
        int bad = 0;
        for (int start : {0, 1, 2, 3, 5, 7, 33}) {
            hipLaunchKernelGGL(unaligned_kernel, dim3(1), dim3(64), 0, 0, d, start, o);
            uint32_t r[64];
            CHK(hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost));
            for (int t = 0; t < 64; t++) {
                uint32_t s = 0;
                for (int k = 0; k < 20; k++) s = s * 31u + h[start + 20 * t + k];
                bad += s != r[t];
            }
        }
        printf("unaligned 40-byte row loads at 2-byte alignment: %s\n", bad ? "WRONG" : "ok");
    }
    // ---- 2/3. gather rates
    const int blocks = 256 * 8;
    std::vector<uint32_t> h((size_t)blocks * 256 * 20);
    uint32_t s = 12345;
    for (auto &x : h) { s = s * 1664525u + 1013904223u; x = s >> 8; }
    uint32_t *d_idx; float *d_out;
    CHK(hipMalloc(&d_idx, h.size() * 4));
    CHK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
    CHK(hipMemcpy(d_idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    const int iters = 1000;
    const bool extra = getenv("RG_PROBE_COPIES") != nullptr;       // bench.py runs the probe for modes 0 / 1 only
    static const char *names[] = {"ds_read_b32   ", "ds_bpermute_b32", "2 copies, B rotated 16 banks", "2 copies, same banks", "pairs, ds_read_b64 + select"};
    for (int mode = 0; mode < (extra ? 5 : 2); mode++)
        for (int n : {20, 32, 40, 50, 64, 100, 128, 200, 275, 400, 500, 1000}) {
            if (mode >= 1 && n > 64) continue;
            auto launch = [&](int it) {
                if (mode == 0) hipLaunchKernelGGL(gather_kernel<0>, dim3(blocks), dim3(256), 0, 0, d_idx, d_out, it, n);
                else if (mode == 1) hipLaunchKernelGGL(gather_kernel<1>, dim3(blocks), dim3(256), 0, 0, d_idx, d_out, it, n);
                else if (mode == 2) hipLaunchKernelGGL(gather_kernel<2>, dim3(blocks), dim3(256), 0, 0, d_idx, d_out, it, n);
                else if (mode == 3) hipLaunchKernelGGL(gather_kernel<3>, dim3(blocks), dim3(256), 0, 0, d_idx, d_out, it, n);
                else hipLaunchKernelGGL(gather_kernel<4>, dim3(blocks), dim3(256), 0, 0, d_idx, d_out, it, n);
            };
            launch(10);
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(a));
            launch(iters);
            CHK(hipEventRecord(b));
            CHK(hipEventSynchronize(b));
            float ms;
            CHK(hipEventElapsedTime(&ms, a, b));
            const double wave_gathers = (double)blocks * 4 * iters * 20;
            const double clk = ms * 1e-3 * 2.4e9 * 256.0 / wave_gathers;
            printf("%s n=%4d  %.3f ms  %.2f clk per wave-gather per CU (at 2.4 GHz)  %.2f T gathers/s\n",
                   names[mode], n, ms, clk, wave_gathers * 64 / (ms * 1e-3) / 1e12);
        }
    return 0;
}
