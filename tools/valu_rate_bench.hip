// Chain-free float32 VALU throughput on gfx950: what does one SIMD deliver for v_mul_f32 / v_fma_f32 /
// v_pk_mul_f32 / v_pk_fma_f32 when NOTHING but issue can limit it -- 16 independent accumulators per lane (a dependent
// instruction is 16 issues away), operands in distinct VGPR banks (bank = register number mod 4), 1 / 2 / 4 / 8 resident
// waves per SIMD -- and how much of that a dependent chain or a shared bank takes away?  The guide's constant is 2 cycles
// per wave64 v_fma_f32 (MI355X_MICROARCH.md, per-instruction table); tools/gpr_variants measured 3.05 per v_mul_f32 and
// 6.1 per v_pk_mul_f32 in a 20-deep product chain.  This settles which of the two prices pool_reg_kernel's inner loop.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/valu_rate_bench.hip -o tools/valu_rate_bench && tools/valu_rate_bench
//
// Per variant and occupancy: cycles per wave-instruction as the SIMD sees them = (median over waves of the s_memtime
// delta) / instructions per wave / resident waves per SIMD, and the chip-wide rate from HIP events.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int INSTR_PER_ITER = 64;

// register map of every variant: sources v[8:15] (all banks, as scalars or aligned pairs), accumulators v[16:47]
#define PROLOGUE                                                                   \
    "  s_mov_b32 s36, %[n]\n"                                                      \
    "  .set i, 0\n  .rept 8\n  v_mov_b32 v[8+i], %[a]\n  .set i, i+1\n  .endr\n"   \
    "  .set i, 0\n  .rept 32\n  v_mov_b32 v[16+i], 1.0\n  .set i, i+1\n  .endr\n"  \
    "  s_memtime s[38:39]\n  s_waitcnt lgkmcnt(0)\n"                               \
    "1:\n"
// the same prologue with per-lane random mantissas (values in [0.999, 1.001]) in sources and accumulators: DVFS gives
// back clock for low-toggle data, and pool_reg_kernel multiplies real probabilities
#define PROLOGUE_RAND                                                              \
    "  s_mov_b32 s36, %[n]\n"                                                      \
    "  v_mov_b32 v8, %[r0]\n  v_mov_b32 v9, %[r1]\n  v_mov_b32 v10, %[r2]\n  v_mov_b32 v11, %[r3]\n"   \
    "  v_mov_b32 v12, %[r1]\n  v_mov_b32 v13, %[r2]\n  v_mov_b32 v14, %[r3]\n  v_mov_b32 v15, %[r0]\n" \
    "  .set i, 0\n  .rept 8\n  v_mov_b32 v[16+4*i], %[r2]\n  v_mov_b32 v[17+4*i], %[r3]\n  v_mov_b32 v[18+4*i], %[r0]\n  v_mov_b32 v[19+4*i], %[r1]\n  .set i, i+1\n  .endr\n"  \
    "  s_memtime s[38:39]\n  s_waitcnt lgkmcnt(0)\n"                               \
    "1:\n"
#define EPILOGUE                                                                   \
    "  s_sub_u32 s36, s36, 1\n  s_cmp_lg_u32 s36, 0\n  s_cbranch_scc1 1b\n"        \
    "  s_memtime s[40:41]\n  s_waitcnt lgkmcnt(0)\n"                               \
    "  s_sub_u32 s40, s40, s38\n  s_subb_u32 s41, s41, s39\n"                      \
    "  v_mov_b32 %[c0], s40\n  v_mov_b32 %[c1], s41\n"                             \
    "  v_mov_b32 %[o], v16\n"                                                      \
    "  .set i, 1\n  .rept 31\n  v_add_f32 %[o], %[o], v[16+i]\n  .set i, i+1\n  .endr\n"
#define CLOBBERS "s36", "s38", "s39", "s40", "s41", "scc", "memory",                                                  \
    "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23",      \
    "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39",    \
    "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47"

// 16 accumulators v[16+i]; the source sits one bank over: v[8 + ((i+1) & 3)]
#define BODY_MUL     "  .rept 4\n  .set i, 0\n  .rept 16\n  v_mul_f32 v[16+i], v[8+((i+1)&3)], v[16+i]\n  .set i, i+1\n  .endr\n  .endr\n"
// same, the source in the accumulator's own bank
#define BODY_MUL_BC  "  .rept 4\n  .set i, 0\n  .rept 16\n  v_mul_f32 v[16+i], v[8+(i&3)], v[16+i]\n  .set i, i+1\n  .endr\n  .endr\n"
// one accumulator: a 64-deep dependent chain per iteration
#define BODY_MUL_DEP "  .rept 64\n  v_mul_f32 v16, v9, v16\n  .endr\n"
// four accumulators: a dependent instruction is 4 issues away
#define BODY_MUL_DEP4 "  .rept 16\n  .set i, 0\n  .rept 4\n  v_mul_f32 v[16+i], v[8+((i+1)&3)], v[16+i]\n  .set i, i+1\n  .endr\n  .endr\n"
#define BODY_FMA     "  .rept 4\n  .set i, 0\n  .rept 16\n  v_fma_f32 v[16+i], v[8+((i+1)&3)], v[12+((i+2)&3)], v[16+i]\n  .set i, i+1\n  .endr\n  .endr\n"
// 16 pair accumulators v[16+2i : 17+2i] (banks {0,1} or {2,3}); the source pair in the other two banks
#define BODY_PKMUL   "  .rept 4\n  .set i, 0\n  .rept 16\n  v_pk_mul_f32 v[16+2*i:17+2*i], v[8+2*((i+1)&1):9+2*((i+1)&1)], v[16+2*i:17+2*i]\n  .set i, i+1\n  .endr\n  .endr\n"
#define BODY_PKMUL_BC "  .rept 4\n  .set i, 0\n  .rept 16\n  v_pk_mul_f32 v[16+2*i:17+2*i], v[8+2*(i&1):9+2*(i&1)], v[16+2*i:17+2*i]\n  .set i, i+1\n  .endr\n  .endr\n"
#define BODY_PKMUL_DEP "  .rept 64\n  v_pk_mul_f32 v[16:17], v[10:11], v[16:17]\n  .endr\n"
// two pair accumulators = pool_reg_kernel's situation (two product pairs per wave, each 20 deep)
#define BODY_PKMUL_DEP2 "  .rept 32\n  v_pk_mul_f32 v[16:17], v[10:11], v[16:17]\n  v_pk_mul_f32 v[18:19], v[8:9], v[18:19]\n  .endr\n"
#define BODY_PKFMA   "  .rept 4\n  .set i, 0\n  .rept 16\n  v_pk_fma_f32 v[16+2*i:17+2*i], v[8+2*((i+1)&1):9+2*((i+1)&1)], v[12+2*((i+1)&1):13+2*((i+1)&1)], v[16+2*i:17+2*i]\n  .set i, i+1\n  .endr\n  .endr\n"
// the kernel's two chains with the VGPR index mode on (SRC0-relative, index fixed at 0 / switched every instruction pair)
#define BODY_PKMUL_IDXON "  s_mov_b32 s37, 0\n  s_set_gpr_idx_on s37, gpr_idx(SRC0)\n  .rept 32\n  v_pk_mul_f32 v[16:17], v[10:11], v[16:17]\n  v_pk_mul_f32 v[18:19], v[8:9], v[18:19]\n  .endr\n  s_set_gpr_idx_off\n"
#define BODY_PKMUL_IDXSW "  s_mov_b32 s37, 0\n  s_set_gpr_idx_on s37, gpr_idx(SRC0)\n  .rept 32\n  s_set_gpr_idx_idx s37\n  v_pk_mul_f32 v[16:17], v[10:11], v[16:17]\n  v_pk_mul_f32 v[18:19], v[8:9], v[18:19]\n  .endr\n  s_set_gpr_idx_off\n"
// integer and move for scale: is it the f32 datapath or VALU issue in general?
#define BODY_ADDU    "  .rept 4\n  .set i, 0\n  .rept 16\n  v_add_u32 v[16+i], v[8+((i+1)&3)], v[16+i]\n  .set i, i+1\n  .endr\n  .endr\n"
#define BODY_MAXF    "  .rept 4\n  .set i, 0\n  .rept 16\n  v_max_f32 v[16+i], v[8+((i+1)&3)], v[16+i]\n  .set i, i+1\n  .endr\n  .endr\n"
#define BODY_MED3    "  .rept 4\n  .set i, 0\n  .rept 16\n  v_med3_f32 v[16+i], v[8+((i+1)&3)], v[16+i], v[12+((i+2)&3)]\n  .set i, i+1\n  .endr\n  .endr\n"
#define BODY_MAXI    "  .rept 4\n  .set i, 0\n  .rept 16\n  v_max_i32 v[16+i], v[8+((i+1)&3)], v[16+i]\n  .set i, i+1\n  .endr\n  .endr\n"

#define KERNEL(name, BODY)                                                                                          \
    __global__ void __launch_bounds__(256) name(int n, float a, float *out, unsigned long long *cyc)               \
    {                                                                                                               \
        extern __shared__ char lds_[];                                                                              \
        float o; unsigned c0, c1;                                                                                   \
        asm volatile(PROLOGUE BODY EPILOGUE : [o] "=&v"(o), [c0] "=&v"(c0), [c1] "=&v"(c1) : [n] "s"(n), [a] "v"(a) : "s37", CLOBBERS); \
        const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;                                             \
        if (o == 12345.678f) out[g] = o + lds_[threadIdx.x];                                                        \
        if ((threadIdx.x & 63) == 0) cyc[g >> 6] = ((unsigned long long)c1 << 32) | c0;                             \
    }

#define KERNEL_RAND(name, BODY)                                                                                     \
    __global__ void __launch_bounds__(256) name(int n, float a, float *out, unsigned long long *cyc)               \
    {                                                                                                               \
        extern __shared__ char lds_[];                                                                              \
        float o; unsigned c0, c1;                                                                                   \
        const unsigned g32 = blockIdx.x * blockDim.x + threadIdx.x;                                                 \
        float r[4];                                                                                                 \
        for (int i = 0; i < 4; i++) { unsigned h = (g32 * 4 + i) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; \
                                      r[i] = a * (0.999f + 0.002f * (float)(h >> 8) * (1.0f / 16777216.0f)); }      \
        asm volatile(PROLOGUE_RAND BODY EPILOGUE : [o] "=&v"(o), [c0] "=&v"(c0), [c1] "=&v"(c1)                     \
                     : [n] "s"(n), [r0] "v"(r[0]), [r1] "v"(r[1]), [r2] "v"(r[2]), [r3] "v"(r[3]) : "s37", CLOBBERS); \
        const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;                                             \
        if (o == 12345.678f) out[g] = o + lds_[threadIdx.x];                                                        \
        if ((threadIdx.x & 63) == 0) cyc[g >> 6] = ((unsigned long long)c1 << 32) | c0;                             \
    }

// pool_reg_kernel's own situation: sources in v[128:129] / v[192:193] (a 256-register wave: two per SIMD by register
// budget alone), products in v[56:59], index mode on, random mantissas; as 64-thread and as 256-thread workgroups
#define KLIKE(name, THREADS, IDXSW) KLIKE2(name, THREADS, "", IDXSW)
#define KLIKE2(name, THREADS, INIT, IDXSW) KLIKE3(name, THREADS, INIT, IDXSW, "192", "193", "v255")
// SRCB0/SRCB1: the second source pair; VMAX: the highest register the wave claims (v255: two waves per SIMD, v167: three)
#define KLIKE3(name, THREADS, INIT, IDXSW, SRCB0, SRCB1, VMAX)                                                                                  \
    __global__ void __launch_bounds__(THREADS) name(int n, float a, float *out, unsigned long long *cyc)             \
    {                                                                                                               \
        float o; unsigned c0, c1;                                                                                   \
        const unsigned g32 = blockIdx.x * blockDim.x + threadIdx.x;                                                 \
        float r[4];                                                                                                 \
        for (int i = 0; i < 4; i++) { unsigned h = (g32 * 4 + i) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; \
                                      r[i] = a * (0.999f + 0.002f * (float)(h >> 8) * (1.0f / 16777216.0f)); }      \
        asm volatile(                                                                                               \
            "  s_mov_b32 s36, %[n]\n  s_mov_b32 s37, 0\n" INIT                                                     \
            "  v_mov_b32 v128, %[r0]\n  v_mov_b32 v129, %[r1]\n  v_mov_b32 v" SRCB0 ", %[r2]\n  v_mov_b32 v" SRCB1 ", %[r3]\n"  \
            "  v_mov_b32 v56, %[r2]\n  v_mov_b32 v57, %[r3]\n  v_mov_b32 v58, %[r0]\n  v_mov_b32 v59, %[r1]\n"      \
            "  s_memtime s[38:39]\n  s_waitcnt lgkmcnt(0)\n"                                                       \
            "1:\n"                                                                                                  \
            "  s_set_gpr_idx_on s37, gpr_idx(SRC0)\n"                                                               \
            "  .rept 32\n" IDXSW "  v_pk_mul_f32 v[56:57], v[128:129], v[56:57]\n  v_pk_mul_f32 v[58:59], v[" SRCB0 ":" SRCB1 "], v[58:59]\n  .endr\n" \
            "  s_set_gpr_idx_off\n"                                                                                 \
            "  s_sub_u32 s36, s36, 1\n  s_cmp_lg_u32 s36, 0\n  s_cbranch_scc1 1b\n"                                 \
            "  s_memtime s[40:41]\n  s_waitcnt lgkmcnt(0)\n"                                                       \
            "  s_sub_u32 s40, s40, s38\n  s_subb_u32 s41, s41, s39\n"                                               \
            "  v_mov_b32 %[c0], s40\n  v_mov_b32 %[c1], s41\n"                                                      \
            "  v_add_f32 %[o], v56, v57\n  v_add_f32 %[o], %[o], v58\n  v_add_f32 %[o], %[o], v59\n"                \
            : [o] "=&v"(o), [c0] "=&v"(c0), [c1] "=&v"(c1)                                                          \
            : [n] "s"(n), [r0] "v"(r[0]), [r1] "v"(r[1]), [r2] "v"(r[2]), [r3] "v"(r[3])                            \
            : "s36", "s37", "s38", "s39", "s40", "s41", "scc", "memory", "v56", "v57", "v58", "v59", "v128", "v129", "v" SRCB0, "v" SRCB1, VMAX); \
        const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;                                             \
        if (o == 12345.678f) out[g] = o;                                                                            \
        if ((threadIdx.x & 63) == 0) cyc[g >> 6] = ((unsigned long long)c1 << 32) | c0;                             \
    }
KLIKE(k_klike64, 64, "")
KLIKE(k_klike256, 256, "")
KLIKE(k_klike64_sw, 64, "  s_set_gpr_idx_idx s37\n")
KLIKE(k_klike64_sw2, 64, "  s_lshr_b32 s37, s37, 8\n  s_set_gpr_idx_idx s37\n")
// round 5's draw: ONE scalar instruction -- the table entry is the draw's M0 (SRC0 enable | 2 x index), written straight into M0
KLIKE2(k_klike64_m0, 64, "  s_mov_b32 s37, 0x10001000\n", "  s_lshr_b32 m0, s37, 16\n")
// the same draw in a 168-register wave (three per SIMD): what a 20-read-only register map would buy
KLIKE3(k_klike64_m0_3w, 64, "  s_mov_b32 s37, 0x10001000\n", "  s_lshr_b32 m0, s37, 16\n", "160", "161", "v167")

KERNEL_RAND(k_pkmul_rand, BODY_PKMUL)
KERNEL_RAND(k_pkmul_dep2_rand, BODY_PKMUL_DEP2)
KERNEL_RAND(k_mul_rand, BODY_MUL)
KERNEL_RAND(k_pkmul_idxon_rand, BODY_PKMUL_IDXON)
KERNEL_RAND(k_pkmul_idxsw_rand, BODY_PKMUL_IDXSW)
KERNEL(k_pkmul_idxon, BODY_PKMUL_IDXON)
KERNEL(k_mul, BODY_MUL)
KERNEL(k_mul_bank, BODY_MUL_BC)
KERNEL(k_mul_dep, BODY_MUL_DEP)
KERNEL(k_mul_dep4, BODY_MUL_DEP4)
KERNEL(k_fma, BODY_FMA)
KERNEL(k_pkmul, BODY_PKMUL)
KERNEL(k_pkmul_bank, BODY_PKMUL_BC)
KERNEL(k_pkmul_dep, BODY_PKMUL_DEP)
KERNEL(k_pkmul_dep2, BODY_PKMUL_DEP2)
KERNEL(k_pkfma, BODY_PKFMA)
KERNEL(k_addu, BODY_ADDU)
KERNEL(k_maxi, BODY_MAXI)
KERNEL(k_maxf, BODY_MAXF)
KERNEL(k_med3, BODY_MED3)

typedef void (*kern_t)(int, float, float *, unsigned long long *);
struct Variant { const char *name; kern_t fn; int lane_ops; const char *what; };

int main(int argc, char **argv)
{
    const bool quick = argc > 1 && std::string(argv[1]) == "--ceiling";     // bench.py: the rows pool_roofline quotes, ~0.1 s
    const int iters = quick ? 4000 : argc > 1 ? atoi(argv[1]) : 16000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const Variant vs[] = {
        {"v_mul_f32      16 indep, banks apart", k_mul, 1, ""},
        {"v_mul_f32      16 indep, same bank", k_mul_bank, 1, ""},
        {"v_mul_f32      4 indep", k_mul_dep4, 1, ""},
        {"v_mul_f32      1 chain", k_mul_dep, 1, ""},
        {"v_fma_f32      16 indep, banks apart", k_fma, 1, ""},
        {"v_pk_mul_f32   16 indep, banks apart", k_pkmul, 2, ""},
        {"v_pk_mul_f32   16 indep, same banks", k_pkmul_bank, 2, ""},
        {"v_pk_mul_f32   2 chains (the kernel's)", k_pkmul_dep2, 2, ""},
        {"v_pk_mul_f32   1 chain", k_pkmul_dep, 2, ""},
        {"v_pk_fma_f32   16 indep, banks apart", k_pkfma, 2, ""},
        {"v_mul_f32      16 indep, random mantissas", k_mul_rand, 1, ""},
        {"v_pk_mul_f32   16 indep, random mantissas", k_pkmul_rand, 2, ""},
        {"v_pk_mul_f32   2 chains, random mantissas", k_pkmul_dep2_rand, 2, ""},
        {"v_pk_mul_f32   2 chains, index mode on", k_pkmul_idxon, 2, ""},
        {"v_pk_mul_f32   2 chains, index mode on, random mantissas", k_pkmul_idxon_rand, 2, ""},
        {"v_pk_mul_f32   2 chains, index switched per pair, random", k_pkmul_idxsw_rand, 2, ""},
        {"v_add_u32      16 indep", k_addu, 1, ""},
        {"v_max_i32      16 indep", k_maxi, 1, ""},
        {"v_max_f32      16 indep", k_maxf, 1, ""},
        {"v_med3_f32     16 indep", k_med3, 1, ""},
    };
    float *out; unsigned long long *cyc;
    CHECK(hipMalloc(&out, (size_t)16384 * 256 * 4));
    CHECK(hipMalloc(&cyc, (size_t)16384 * 4 * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("{\"device\": \"%s\", \"cus\": %d, \"iters\": %d, \"instr_per_wave\": %d, \"rows\": [\n", prop.gcnArchName, cus, iters, iters * INSTR_PER_ITER);
    bool first = true;
    for (const Variant &v : vs) {
        if (quick && v.fn != (kern_t)k_mul_rand && v.fn != (kern_t)k_pkmul_rand) continue;
        for (int W : {1, 2, 4, 8}) {
            if (quick && W != 2 && W != 8) continue;
            // exactly W workgroups of 4 waves (one per SIMD) fit a CU: each takes 1/W of the 160 KB of LDS
            const size_t lds = (size_t)(160 * 1024 / W) - (W == 1 ? 0 : 512);
            CHECK(hipFuncSetAttribute((const void *)v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int blocks = cus * W;
            hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(256), lds, 0, iters, 1.0000001f, out, cyc);      // warm (clocks included)
            CHECK(hipDeviceSynchronize());
            float ms = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(256), lds, 0, iters, 1.0000001f, out, cyc);
                CHECK(hipEventRecord(e1));
                CHECK(hipDeviceSynchronize());
                float m = 0;
                CHECK(hipEventElapsedTime(&m, e0, e1));
                ms = std::min(ms, m);
            }
            std::vector<unsigned long long> c((size_t)blocks * 4);
            CHECK(hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost));
            std::sort(c.begin(), c.end());
            const double n_instr = (double)iters * INSTR_PER_ITER;
            const double med = (double)c[c.size() / 2], mx = (double)c.back();
            const double cyc_per_instr_simd = med / n_instr / W;
            const double chip = (double)blocks * 4 * n_instr * 64 * v.lane_ops / (ms * 1e-3);
            // from the events: SIMD cycles per wave-instruction at an ASSUMED 2.4 GHz (the clock is not observable from here)
            const double cyc_events = (ms * 1e-3) * 2.4e9 / (n_instr * W);
            printf("%s {\"variant\": \"%s\", \"waves_per_simd\": %d, \"T_lane_ops_per_s\": %.2f, \"cycles_per_wave_instr_at_2.4GHz\": %.3f, "
                   "\"s_memtime_ticks_per_wave_instr_per_simd\": %.3f, \"wave_ticks_median\": %.0f, \"wave_ticks_max\": %.0f, \"ms\": %.4f}",
                   first ? " " : ",\n", v.name, W, chip / 1e12, cyc_events, cyc_per_instr_simd, med, mx, ms);
            first = false;
        }
    }
    // the kernel-like variants: occupancy comes from their 256 registers (2 waves per SIMD), no LDS
    struct KL { const char *name; kern_t fn; int threads; int wps = 2; } kl[] = {
        {"kernel-like, 64-thread workgroups, index mode on", k_klike64, 64},
        {"kernel-like, 256-thread workgroups, index mode on", k_klike256, 256},
        {"kernel-like, 64-thread, 1 SALU (idx) per draw", k_klike64_sw, 64},
        {"kernel-like, 64-thread, 2 SALU (shift + idx) per draw", k_klike64_sw2, 64},
        {"kernel-like, 64-thread, 1 SALU (M0 write: round 5's draw) per draw", k_klike64_m0, 64},
        {"kernel-like, 1 SALU (M0 write) per draw, 168-register waves: 3 per SIMD", k_klike64_m0_3w, 64, 3},
    };
    for (const KL &v : kl) {
        if (quick && v.wps != 2) continue;
        const int waves = cus * 4 * v.wps, blocks = waves * 64 / v.threads;
        hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(v.threads), 0, 0, iters, 1.0000001f, out, cyc);
        CHECK(hipDeviceSynchronize());
        float ms = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(v.threads), 0, 0, iters, 1.0000001f, out, cyc);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float m = 0;
            CHECK(hipEventElapsedTime(&m, e0, e1));
            ms = std::min(ms, m);
        }
        const double n_instr = (double)iters * INSTR_PER_ITER;
        const double chip = (double)waves * n_instr * 64 * 2 / (ms * 1e-3);
        printf(",\n {\"variant\": \"%s\", \"waves_per_simd\": %d, \"T_lane_ops_per_s\": %.2f, \"cycles_per_wave_instr_at_2.4GHz\": %.3f, "
               "\"s_memtime_ticks_per_wave_instr_per_simd\": 0, \"ms\": %.4f}", v.name, v.wps, chip / 1e12, (ms * 1e-3) * 2.4e9 / (n_instr * v.wps), ms);
    }
    // pool_reg_kernel's launch shape: how does a kernel of N single-wave, 256-register workgroups (the chip holds 2 048)
    // scale with N when one wave takes ~0.2 ms?  (tools/pool_reg_rounds.py: the kernel itself is 0.36 ms at 2 048 waves but
    // only +0.17 ms per further 2 048)
    for (int waves : {512, 1024, 2048, 3936, 4096, 6144, 8192, 16384}) {
        if (quick) break;
        const int it = 700;
        hipLaunchKernelGGL(k_klike64_sw2, dim3(waves), dim3(64), 0, 0, it, 1.0000001f, out, cyc);
        CHECK(hipDeviceSynchronize());
        float ms = 1e30f;
        for (int rep = 0; rep < 5; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_klike64_sw2, dim3(waves), dim3(64), 0, 0, it, 1.0000001f, out, cyc);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float m = 0;
            CHECK(hipEventElapsedTime(&m, e0, e1));
            ms = std::min(ms, m);
        }
        printf(",\n {\"variant\": \"launch shape: %d single-wave 256-register workgroups, 44 800 v_pk_mul each\", \"waves_per_simd\": 2, \"T_lane_ops_per_s\": %.2f, "
               "\"cycles_per_wave_instr_at_2.4GHz\": 0, \"s_memtime_ticks_per_wave_instr_per_simd\": 0, \"waves\": %d, \"ms\": %.4f}",
               waves, (double)waves * it * 64 * 128 / (ms * 1e-3) / 1e12, waves, ms);
    }
    printf("\n]}\n");
    return 0;
}
