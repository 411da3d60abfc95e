// What does a VALU instruction cost next to f32 MFMAs on one gfx950 SIMD, as a function of HOW it is placed?
// (round 6; the encoder's floor: 116 v_mfma_f32_32x32x2_f32 = 7 424 SIMD cycles per 32-read tile + ~190 other VALU instructions that
// profiles/r06_encoder_floor.txt prices at 5.9 cycles each where the datapath needs 2.)
//
// A wave runs groups of  G dependent-or-alternating MFMAs + M VALU instructions  in a loop; lane 0 stamps s_memtime around the loop.
// Knobs: waves per SIMD (1 or 2, forced through the workgroup's LDS size), burst vs spread placement of the VALU instructions,
// one accumulator (every MFMA depends on the one before, as layer 2 of the encoder) or two in alternation, s_setprio around the
// VALU part, the VALU opcode, and whether the VALU results are independent, one chain, or the MFMAs' B operands.  Printed per configuration: shader cycles per group per wave (median over all waves), the share of
// SIMD cycles the matrix pipe is busy  = W * G * 64 / cycles,  and  (cycles / W - 64 G) / M  = SIMD cycles one VALU instruction cost.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_valu_mix.hip -o mfma_valu_mix ; run: ./mfma_valu_mix > out.json
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int VOP>
__device__ __forceinline__ void valu(float &x, f32x2 &xp, float c1, float c2, f32x2 p1, f32x2 p2)
{
    if (VOP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2));
    if (VOP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(xp) : "v"(p1), "v"(p2));
    if (VOP == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(c1));
    if (VOP == 3) asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(x) : "v"(c1), "v"(c2));
    if (VOP == 4 || VOP == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2));
}

// PRIO: 0 none; 1 VALU part at priority 0, MFMA part at 3; 2 VALU part at 3, MFMA part at 0
template <int G, int M, int SPREAD, int NACC, int PRIO, int VOP>
__global__ __launch_bounds__(256) void mix(const float *w, float *out, unsigned long long *cyc, int groups)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    if (groups < 0) lds[threadIdx.x] = 0.f;                    // keeps the allocation
    const float a = w[lane], b = w[64 + lane], c1 = w[128 + lane], c2 = w[192 + lane] * 1e-3f;
    const f32x2 p1 = {c1, c1}, p2 = {c2, c2};
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int q = 0; q < 16; q++) acc[i][q] = 0.f;
    constexpr int MV = M > 0 ? M : 1;
    float x[MV];
    f32x2 xp[MV];
#pragma unroll
    for (int i = 0; i < MV; i++) { x[i] = w[256 + i * 64 + lane]; xp[i] = f32x2{x[i], x[i]}; }
    const unsigned long long t0 = __builtin_readcyclecounter();           // s_memtime
    constexpr int U = 64 / G > 1 ? 64 / G : 2;                           // groups per loop trip: the branch is paid once per >= 64 MFMAs
    for (int g = 0; g < groups; g += U) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (SPREAD == 0) {
            if (PRIO == 1) __builtin_amdgcn_s_setprio(3);
            if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int i = 0; i < G; i++) acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, VOP == 5 ? x[i % MV] : b, acc[i % NACC], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
            if (PRIO == 2) __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int i = 0; i < M; i++) valu<VOP>(x[VOP == 4 ? 0 : i], xp[i], c1, c2, p1, p2);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            // spread: M / G VALU instructions after every MFMA (M a multiple of G), or one after every G / M MFMAs
            constexpr int per = M >= G ? M / G : 1, every = M >= G ? 1 : G / MV;
#pragma unroll
            for (int i = 0; i < G; i++) {
                acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i % NACC], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (M > 0 && i % every == every - 1) {
#pragma unroll
                    for (int j = 0; j < per; j++) valu<VOP>(x[((i / every) * per + j) % MV], xp[((i / every) * per + j) % MV], c1, c2, p1, p2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
      }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int q = 0; q < 16; q++) r += acc[i][q];
#pragma unroll
    for (int i = 0; i < MV; i++) r += x[i] + xp[i].x + xp[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static float *d_w, *d_out;
static unsigned long long *d_cyc;
static bool first = true;

template <int G, int M, int SPREAD, int NACC, int PRIO, int VOP>
static void run(int wps)
{
    const int blocks = 256 * wps, groups = 16384 / G;        // a multiple of every U
    const size_t lds = wps == 1 ? 96 * 1024 : 64 * 1024;                // 160 KB per CU: one or two workgroups fit
    auto kern = mix<G, M, SPREAD, NACC, PRIO, VOP>;
    CHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    std::vector<unsigned long long> h(blocks * 4);
    double med = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d_w, d_out, d_cyc, groups);
        CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        med = (double)h[h.size() / 2] / groups;
    }
    const double busy = wps * G * 64.0 / med, per_valu = M ? (med / wps - 64.0 * G) / M : 0.0;
    static const char *vn[] = {"v_fma_f32", "v_pk_fma_f32", "v_add_f32", "v_fma_f32 clamp", "v_fma_f32, one dependent chain", "v_fma_f32 -> B operands of the group's MFMAs"};
    printf("%s{\"waves_per_simd\": %d, \"mfma_per_group\": %d, \"valu_per_group\": %d, \"placement\": \"%s\", \"accumulators\": %d, \"prio\": \"%s\", "
           "\"valu\": \"%s\", \"cycles_per_group_per_wave\": %.1f, \"matrix_pipe_busy\": %.4f, \"simd_cycles_per_valu\": %.2f}",
           first ? "" : ",\n", wps, G, M, SPREAD ? "spread" : "burst", NACC, PRIO == 0 ? "none" : PRIO == 1 ? "valu 0 / mfma 3" : "valu 3 / mfma 0",
           vn[VOP], med, busy, per_valu);
    first = false;
}

// The encoder's own block (layer2_with_bn in m6a_kernels.hip): four clamped fmas on (alpha, beta) pairs that came from LDS one block earlier, the NEXT block's two
// ds_read_b128, four MFMAs whose B operands are the fmas' results.  LOADS = 0: the same without the LDS reads (pairs stay in registers).
template <int LOADS>
__global__ __launch_bounds__(256) void encblock(const float *w, float *out, unsigned long long *cyc, int groups)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = w[i & 1023];
    __syncthreads();
    const float a = w[lane];
    float h[4];
#pragma unroll
    for (int i = 0; i < 4; i++) h[i] = w[256 + i * 64 + lane];
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; q++) acc[q] = 0.f;
    const float *base = lds + (lane >> 5) * 32;
    float4 p0 = *(const float4 *)(base), p1 = *(const float4 *)(base + 4);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int g = 0; g < groups; g += 16) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(h[0]) : "v"(p0.x), "v"(p0.y));
            asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(h[1]) : "v"(p0.z), "v"(p0.w));
            asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(h[2]) : "v"(p1.x), "v"(p1.y));
            asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(h[3]) : "v"(p1.z), "v"(p1.w));
            if (LOADS == 1) {
                p0 = *(const float4 *)(base + 64 * ((u + 1) & 15));
                p1 = *(const float4 *)(base + 64 * ((u + 1) & 15) + 4);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, h[0], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (LOADS >= 2) {                       // behind the block's FIRST MFMA: the wave waits there for the dependent second one anyway
                p0 = *(const float4 *)(base + 64 * ((u + 1) & 15));
                p1 = *(const float4 *)(base + 64 * ((u + 1) & 15) + 4);
                __builtin_amdgcn_sched_barrier(0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, h[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, h[2], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (LOADS == 3) {                       // ... and the wait for them behind the third
                __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0), vmcnt / expcnt untouched (gfx9 encoding)
                __builtin_amdgcn_sched_barrier(0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, h[3], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = p0.x + p1.y;
#pragma unroll
    for (int q = 0; q < 16; q++) r += acc[q];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int LOADS>
static void run_encblock(int wps)
{
    const int blocks = 256 * wps, groups = 4096;
    const size_t lds = wps == 1 ? 96 * 1024 : 64 * 1024;
    auto kern = encblock<LOADS>;
    CHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    std::vector<unsigned long long> h(blocks * 4);
    double med = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d_w, d_out, d_cyc, groups);
        CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        med = (double)h[h.size() / 2] / groups;
    }
    printf(",\n{\"waves_per_simd\": %d, \"mfma_per_group\": 4, \"valu_per_group\": 4, \"placement\": \"the encoder's block: 4 clamped fmas%s, 4 MFMAs on their results\", "
           "\"cycles_per_group_per_wave\": %.1f, \"matrix_pipe_busy\": %.4f, \"simd_cycles_per_block_beyond_its_mfmas\": %.2f}",
           wps, LOADS == 0 ? " (pairs in registers, no LDS)" : LOADS == 1 ? " + the next block's two ds_read_b128" : LOADS == 2 ? ", first MFMA, the next block's two ds_read_b128 behind it" : ", first MFMA, the two ds_read_b128 behind it, s_waitcnt behind the third", med, wps * 256.0 / med, med / wps - 256.0);
}

template <int G, int M, int SPREAD, int NACC, int PRIO, int VOP>
static void both() { run<G, M, SPREAD, NACC, PRIO, VOP>(1); run<G, M, SPREAD, NACC, PRIO, VOP>(2); }

int main()
{
    CHK(hipMalloc(&d_w, 64 * 64 * 4));
    CHK(hipMalloc(&d_out, 512 * 256 * 4));
    CHK(hipMalloc(&d_cyc, 512 * 4 * 8));
    std::vector<float> h(64 * 64);
    for (size_t i = 0; i < h.size(); i++) h[i] = 0.5f + 0.0001f * (float)(i % 97);
    CHK(hipMemcpy(d_w, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    printf("{\"what\": \"tools/mfma_valu_mix: groups of G v_mfma_f32_32x32x2_f32 (64 SIMD cycles each) + M VALU instructions per wave, 1 or 2 waves per SIMD; "
           "cycles = s_memtime, median over all waves\", \"configs\": [\n");
    // pure MFMA: the ceiling of the harness
    both<4, 0, 0, 1, 0, 0>();
    both<16, 0, 0, 1, 0, 0>();
    // bursts of M independent VALU instructions after G dependent MFMAs (the encoder's blocks are G = 4, M = 4): fixed cost and slope
    both<4, 1, 0, 1, 0, 0>();
    both<4, 2, 0, 1, 0, 0>();
    both<4, 4, 0, 1, 0, 0>();
    both<4, 8, 0, 1, 0, 0>();
    both<4, 16, 0, 1, 0, 0>();
    both<8, 8, 0, 1, 0, 0>();
    both<8, 16, 0, 1, 0, 0>();
    both<16, 16, 0, 1, 0, 0>();
    both<16, 32, 0, 1, 0, 0>();
    both<16, 64, 0, 1, 0, 0>();
    both<32, 32, 0, 1, 0, 0>();
    // other opcodes in the encoder's block shape
    both<4, 4, 0, 1, 0, 3>();
    both<4, 4, 0, 1, 0, 2>();
    both<4, 2, 0, 1, 0, 1>();
    both<16, 8, 0, 1, 0, 1>();
    // the VALU results are the B operands of the group's MFMAs (batch norm -> layer 2)
    both<4, 4, 0, 1, 0, 5>();
    both<8, 8, 0, 1, 0, 5>();
    both<16, 16, 0, 1, 0, 5>();
    // one dependent chain of M instructions (the epilogue's shape)
    both<16, 16, 0, 1, 0, 4>();
    both<16, 64, 0, 1, 0, 4>();
    // spread between the MFMAs instead
    both<4, 4, 1, 1, 0, 0>();
    both<16, 16, 1, 1, 0, 0>();
    both<16, 4, 1, 1, 0, 0>();
    // two accumulators in alternation (an independent MFMA could be queued behind the running one)
    both<4, 4, 0, 2, 0, 0>();
    both<16, 16, 0, 2, 0, 0>();
    both<16, 16, 1, 2, 0, 0>();
    // priorities (two waves per SIMD is where they could act)
    both<4, 4, 0, 1, 1, 0>();
    both<4, 4, 0, 1, 2, 0>();
    both<16, 64, 0, 1, 1, 0>();
    run_encblock<0>(1); run_encblock<0>(2); run_encblock<1>(1); run_encblock<1>(2); run_encblock<2>(1); run_encblock<2>(2); run_encblock<3>(1); run_encblock<3>(2);
    printf("\n]}\n");
    return 0;
}
