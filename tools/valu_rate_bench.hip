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
// integer and move for scale: is it the f32 datapath or VALU issue in general?
#define BODY_ADDU    "  .rept 4\n  .set i, 0\n  .rept 16\n  v_add_u32 v[16+i], v[8+((i+1)&3)], v[16+i]\n  .set i, i+1\n  .endr\n  .endr\n"
#define BODY_MAXI    "  .rept 4\n  .set i, 0\n  .rept 16\n  v_max_i32 v[16+i], v[8+((i+1)&3)], v[16+i]\n  .set i, i+1\n  .endr\n  .endr\n"

#define KERNEL(name, BODY)                                                                                          \
    __global__ void __launch_bounds__(256) name(int n, float a, float *out, unsigned long long *cyc)               \
    {                                                                                                               \
        extern __shared__ char lds_[];                                                                              \
        float o; unsigned c0, c1;                                                                                   \
        asm volatile(PROLOGUE BODY EPILOGUE : [o] "=&v"(o), [c0] "=&v"(c0), [c1] "=&v"(c1) : [n] "s"(n), [a] "v"(a) : CLOBBERS); \
        const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;                                             \
        if (o == 12345.678f) out[g] = o + lds_[threadIdx.x];                                                        \
        if ((threadIdx.x & 63) == 0) cyc[g >> 6] = ((unsigned long long)c1 << 32) | c0;                             \
    }

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

typedef void (*kern_t)(int, float, float *, unsigned long long *);
struct Variant { const char *name; kern_t fn; int lane_ops; const char *what; };

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
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
        {"v_add_u32      16 indep", k_addu, 1, ""},
        {"v_max_i32      16 indep", k_maxi, 1, ""},
    };
    float *out; unsigned long long *cyc;
    CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    CHECK(hipMalloc(&cyc, (size_t)cus * 8 * 4 * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("{\"device\": \"%s\", \"cus\": %d, \"iters\": %d, \"instr_per_wave\": %d, \"rows\": [\n", prop.gcnArchName, cus, iters, iters * INSTR_PER_ITER);
    bool first = true;
    for (const Variant &v : vs) {
        for (int W : {1, 2, 4, 8}) {
            // exactly W workgroups of 4 waves (one per SIMD) fit a CU: each takes 1/W of the 160 KB of LDS
            const size_t lds = (size_t)(160 * 1024 / W) - (W == 1 ? 0 : 512);
            CHECK(hipFuncSetAttribute((const void *)v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int blocks = cus * W;
            hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(256), lds, 0, 64, 1.0000001f, out, cyc);      // warm
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(256), lds, 0, iters, 1.0000001f, out, cyc);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> c((size_t)blocks * 4);
            CHECK(hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost));
            std::sort(c.begin(), c.end());
            const double n_instr = (double)iters * INSTR_PER_ITER;
            const double med = (double)c[c.size() / 2], mx = (double)c.back();
            const double cyc_per_instr_simd = med / n_instr / W;
            const double chip = (double)blocks * 4 * n_instr * 64 * v.lane_ops / (ms * 1e-3);
            printf("%s {\"variant\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_wave_instr_per_simd\": %.3f, \"lane_ops_per_simd_cycle\": %.2f, "
                   "\"wave_cycles_median\": %.0f, \"wave_cycles_max\": %.0f, \"ms\": %.4f, \"T_lane_ops_per_s\": %.2f}",
                   first ? " " : ",\n", v.name, W, cyc_per_instr_simd, 64.0 * v.lane_ops / cyc_per_instr_simd, med, mx, ms, chip / 1e12);
            first = false;
        }
    }
    printf("\n]}\n");
    return 0;
}
