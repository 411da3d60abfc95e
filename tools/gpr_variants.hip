#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(64) void k_pk(const unsigned *tab, int T, float *out)
{
    const float seed = (float)(threadIdx.x + 1) * 1e-5f + (float)blockIdx.x * 1e-9f;
    float o0, o1, o2, o3;
    const unsigned long long tp = (unsigned long long)tab;
    asm volatile(
        "  s_mov_b64 s[38:39], %[tab]\n"
        "  s_mov_b32 s36, %[T]\n"
        "  v_mov_b32 v8, 0\n"
        "  v_mov_b32 v9, 0\n"
        "  v_mov_b32 v10, 0\n"
        "  v_mov_b32 v11, 0\n"
        "  .set i, 0\n  .rept 128\n  v_mov_b32 v3, i+1\n  v_cvt_f32_i32 v3, v3\n  v_fma_f32 v[128+i], %[seed], v3, 0.5\n  .set i, i+1\n  .endr\n"
        "1:\n"
        "  s_load_dwordx16 s[16:31], s[38:39], 0x0\n  s_load_dwordx4 s[32:35], s[38:39], 0x40\n"
        "  v_mov_b32 v4, 1.0\n"
        "  v_mov_b32 v5, 1.0\n"
        "  v_mov_b32 v6, 1.0\n"
        "  v_mov_b32 v7, 1.0\n"
        "  s_waitcnt lgkmcnt(0)\n"
        "  s_set_gpr_idx_on s16, gpr_idx(SRC0)\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  s_set_gpr_idx_idx s17\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  s_set_gpr_idx_idx s18\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  s_set_gpr_idx_idx s19\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  s_set_gpr_idx_idx s20\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  s_set_gpr_idx_idx s21\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  s_set_gpr_idx_idx s22\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  s_set_gpr_idx_idx s23\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  s_set_gpr_idx_idx s24\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  s_set_gpr_idx_idx s25\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  s_set_gpr_idx_idx s26\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  s_set_gpr_idx_idx s27\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  s_set_gpr_idx_idx s28\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  s_set_gpr_idx_idx s29\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  s_set_gpr_idx_idx s30\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  s_set_gpr_idx_idx s31\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  s_set_gpr_idx_idx s32\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  s_set_gpr_idx_idx s33\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  s_set_gpr_idx_idx s34\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  s_set_gpr_idx_idx s35\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  s_set_gpr_idx_off\n"
        "  v_pk_add_f32 v[4:5], v[4:5], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n  v_pk_add_f32 v[6:7], v[6:7], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n"
        "  v_pk_add_f32 v[8:9], v[8:9], v[4:5]\n  v_pk_add_f32 v[10:11], v[10:11], v[6:7]\n"
        "  s_add_u32 s38, s38, 80\n  s_addc_u32 s39, s39, 0\n  s_sub_u32 s36, s36, 1\n  s_cmp_lg_u32 s36, 0\n  s_cbranch_scc1 1b\n"
        "  v_mov_b32 %[o0], v8\n"
        "  v_mov_b32 %[o1], v9\n"
        "  v_mov_b32 %[o2], v10\n"
        "  v_mov_b32 %[o3], v11\n"
        : [o0] "=v"(o0), [o1] "=v"(o1), [o2] "=v"(o2), [o3] "=v"(o3)
        : [tab] "s"(tp), [T] "s"(T), [seed] "v"(seed)
        : "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255", "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "s36", "s37", "s38", "s39", "s40", "m0", "scc", "memory");
    float *o = out + ((size_t)blockIdx.x * 64 + threadIdx.x) * 4;
    o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3;
}
__global__ __launch_bounds__(64) void k_pk_noidx(const unsigned *tab, int T, float *out)
{
    const float seed = (float)(threadIdx.x + 1) * 1e-5f + (float)blockIdx.x * 1e-9f;
    float o0, o1, o2, o3;
    const unsigned long long tp = (unsigned long long)tab;
    asm volatile(
        "  s_mov_b64 s[38:39], %[tab]\n"
        "  s_mov_b32 s36, %[T]\n"
        "  v_mov_b32 v8, 0\n"
        "  v_mov_b32 v9, 0\n"
        "  v_mov_b32 v10, 0\n"
        "  v_mov_b32 v11, 0\n"
        "  .set i, 0\n  .rept 128\n  v_mov_b32 v3, i+1\n  v_cvt_f32_i32 v3, v3\n  v_fma_f32 v[128+i], %[seed], v3, 0.5\n  .set i, i+1\n  .endr\n"
        "1:\n"
        "  s_load_dwordx16 s[16:31], s[38:39], 0x0\n  s_load_dwordx4 s[32:35], s[38:39], 0x40\n"
        "  v_mov_b32 v4, 1.0\n"
        "  v_mov_b32 v5, 1.0\n"
        "  v_mov_b32 v6, 1.0\n"
        "  v_mov_b32 v7, 1.0\n"
        "  s_waitcnt lgkmcnt(0)\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "  v_pk_add_f32 v[4:5], v[4:5], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n  v_pk_add_f32 v[6:7], v[6:7], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n"
        "  v_pk_add_f32 v[8:9], v[8:9], v[4:5]\n  v_pk_add_f32 v[10:11], v[10:11], v[6:7]\n"
        "  s_add_u32 s38, s38, 80\n  s_addc_u32 s39, s39, 0\n  s_sub_u32 s36, s36, 1\n  s_cmp_lg_u32 s36, 0\n  s_cbranch_scc1 1b\n"
        "  v_mov_b32 %[o0], v8\n"
        "  v_mov_b32 %[o1], v9\n"
        "  v_mov_b32 %[o2], v10\n"
        "  v_mov_b32 %[o3], v11\n"
        : [o0] "=v"(o0), [o1] "=v"(o1), [o2] "=v"(o2), [o3] "=v"(o3)
        : [tab] "s"(tp), [T] "s"(T), [seed] "v"(seed)
        : "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255", "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "s36", "s37", "s38", "s39", "s40", "m0", "scc", "memory");
    float *o = out + ((size_t)blockIdx.x * 64 + threadIdx.x) * 4;
    o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3;
}
__global__ __launch_bounds__(64) void k_pk_noidx_nc(const unsigned *tab, int T, float *out)
{
    const float seed = (float)(threadIdx.x + 1) * 1e-5f + (float)blockIdx.x * 1e-9f;
    float o0, o1, o2, o3;
    const unsigned long long tp = (unsigned long long)tab;
    asm volatile(
        "  s_mov_b64 s[38:39], %[tab]\n"
        "  s_mov_b32 s36, %[T]\n"
        "  v_mov_b32 v8, 0\n"
        "  v_mov_b32 v9, 0\n"
        "  v_mov_b32 v10, 0\n"
        "  v_mov_b32 v11, 0\n"
        "  .set i, 0\n  .rept 128\n  v_mov_b32 v3, i+1\n  v_cvt_f32_i32 v3, v3\n  v_fma_f32 v[128+i], %[seed], v3, 0.5\n  .set i, i+1\n  .endr\n"
        "1:\n"
        "  s_load_dwordx16 s[16:31], s[38:39], 0x0\n  s_load_dwordx4 s[32:35], s[38:39], 0x40\n"
        "  v_mov_b32 v4, 1.0\n"
        "  v_mov_b32 v5, 1.0\n"
        "  v_mov_b32 v6, 1.0\n"
        "  v_mov_b32 v7, 1.0\n"
        "  s_waitcnt lgkmcnt(0)\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[194:195], v[4:5]\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[194:195], v[4:5]\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[194:195], v[4:5]\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[194:195], v[4:5]\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[194:195], v[4:5]\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[194:195], v[4:5]\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[194:195], v[4:5]\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[194:195], v[4:5]\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[194:195], v[4:5]\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[194:195], v[4:5]\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[194:195], v[4:5]\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[194:195], v[4:5]\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[194:195], v[4:5]\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[194:195], v[4:5]\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[194:195], v[4:5]\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[194:195], v[4:5]\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[194:195], v[4:5]\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[194:195], v[4:5]\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[194:195], v[4:5]\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[194:195], v[4:5]\n"
        "  v_pk_add_f32 v[4:5], v[4:5], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n  v_pk_add_f32 v[6:7], v[6:7], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n"
        "  v_pk_add_f32 v[8:9], v[8:9], v[4:5]\n  v_pk_add_f32 v[10:11], v[10:11], v[6:7]\n"
        "  s_add_u32 s38, s38, 80\n  s_addc_u32 s39, s39, 0\n  s_sub_u32 s36, s36, 1\n  s_cmp_lg_u32 s36, 0\n  s_cbranch_scc1 1b\n"
        "  v_mov_b32 %[o0], v8\n"
        "  v_mov_b32 %[o1], v9\n"
        "  v_mov_b32 %[o2], v10\n"
        "  v_mov_b32 %[o3], v11\n"
        : [o0] "=v"(o0), [o1] "=v"(o1), [o2] "=v"(o2), [o3] "=v"(o3)
        : [tab] "s"(tp), [T] "s"(T), [seed] "v"(seed)
        : "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255", "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "s36", "s37", "s38", "s39", "s40", "m0", "scc", "memory");
    float *o = out + ((size_t)blockIdx.x * 64 + threadIdx.x) * 4;
    o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3;
}
__global__ __launch_bounds__(64) void k_pk_noidx_c(const unsigned *tab, int T, float *out)
{
    const float seed = (float)(threadIdx.x + 1) * 1e-5f + (float)blockIdx.x * 1e-9f;
    float o0, o1, o2, o3;
    const unsigned long long tp = (unsigned long long)tab;
    asm volatile(
        "  s_mov_b64 s[38:39], %[tab]\n"
        "  s_mov_b32 s36, %[T]\n"
        "  v_mov_b32 v8, 0\n"
        "  v_mov_b32 v9, 0\n"
        "  v_mov_b32 v10, 0\n"
        "  v_mov_b32 v11, 0\n"
        "  .set i, 0\n  .rept 128\n  v_mov_b32 v3, i+1\n  v_cvt_f32_i32 v3, v3\n  v_fma_f32 v[128+i], %[seed], v3, 0.5\n  .set i, i+1\n  .endr\n"
        "1:\n"
        "  s_load_dwordx16 s[16:31], s[38:39], 0x0\n  s_load_dwordx4 s[32:35], s[38:39], 0x40\n"
        "  v_mov_b32 v4, 1.0\n"
        "  v_mov_b32 v5, 1.0\n"
        "  v_mov_b32 v6, 1.0\n"
        "  v_mov_b32 v7, 1.0\n"
        "  s_waitcnt lgkmcnt(0)\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[194:195], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[194:195], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[194:195], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[194:195], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[194:195], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[194:195], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[194:195], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[194:195], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[194:195], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[194:195], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[194:195], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[194:195], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[194:195], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[194:195], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[194:195], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[194:195], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[194:195], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[194:195], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[194:195], v[6:7]\n"
        "  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n  v_pk_mul_f32 v[6:7], v[194:195], v[6:7]\n"
        "  v_pk_add_f32 v[4:5], v[4:5], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n  v_pk_add_f32 v[6:7], v[6:7], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n"
        "  v_pk_add_f32 v[8:9], v[8:9], v[4:5]\n  v_pk_add_f32 v[10:11], v[10:11], v[6:7]\n"
        "  s_add_u32 s38, s38, 80\n  s_addc_u32 s39, s39, 0\n  s_sub_u32 s36, s36, 1\n  s_cmp_lg_u32 s36, 0\n  s_cbranch_scc1 1b\n"
        "  v_mov_b32 %[o0], v8\n"
        "  v_mov_b32 %[o1], v9\n"
        "  v_mov_b32 %[o2], v10\n"
        "  v_mov_b32 %[o3], v11\n"
        : [o0] "=v"(o0), [o1] "=v"(o1), [o2] "=v"(o2), [o3] "=v"(o3)
        : [tab] "s"(tp), [T] "s"(T), [seed] "v"(seed)
        : "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255", "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "s36", "s37", "s38", "s39", "s40", "m0", "scc", "memory");
    float *o = out + ((size_t)blockIdx.x * 64 + threadIdx.x) * 4;
    o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3;
}
__global__ __launch_bounds__(64) void k_pk_nc(const unsigned *tab, int T, float *out)
{
    const float seed = (float)(threadIdx.x + 1) * 1e-5f + (float)blockIdx.x * 1e-9f;
    float o0, o1, o2, o3;
    const unsigned long long tp = (unsigned long long)tab;
    asm volatile(
        "  s_mov_b64 s[38:39], %[tab]\n"
        "  s_mov_b32 s36, %[T]\n"
        "  v_mov_b32 v8, 0\n"
        "  v_mov_b32 v9, 0\n"
        "  v_mov_b32 v10, 0\n"
        "  v_mov_b32 v11, 0\n"
        "  .set i, 0\n  .rept 128\n  v_mov_b32 v3, i+1\n  v_cvt_f32_i32 v3, v3\n  v_fma_f32 v[128+i], %[seed], v3, 0.5\n  .set i, i+1\n  .endr\n"
        "1:\n"
        "  s_load_dwordx16 s[16:31], s[38:39], 0x0\n  s_load_dwordx4 s[32:35], s[38:39], 0x40\n"
        "  v_mov_b32 v4, 1.0\n"
        "  v_mov_b32 v5, 1.0\n"
        "  v_mov_b32 v6, 1.0\n"
        "  v_mov_b32 v7, 1.0\n"
        "  s_waitcnt lgkmcnt(0)\n"
        "  s_lshl_b32 s40, s16, 1\n  s_set_gpr_idx_on s40, gpr_idx(SRC0)\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "  s_lshl_b32 s40, s17, 1\n  s_set_gpr_idx_idx s40\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "  s_lshl_b32 s40, s18, 1\n  s_set_gpr_idx_idx s40\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "  s_lshl_b32 s40, s19, 1\n  s_set_gpr_idx_idx s40\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "  s_lshl_b32 s40, s20, 1\n  s_set_gpr_idx_idx s40\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "  s_lshl_b32 s40, s21, 1\n  s_set_gpr_idx_idx s40\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "  s_lshl_b32 s40, s22, 1\n  s_set_gpr_idx_idx s40\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "  s_lshl_b32 s40, s23, 1\n  s_set_gpr_idx_idx s40\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "  s_lshl_b32 s40, s24, 1\n  s_set_gpr_idx_idx s40\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "  s_lshl_b32 s40, s25, 1\n  s_set_gpr_idx_idx s40\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "  s_lshl_b32 s40, s26, 1\n  s_set_gpr_idx_idx s40\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "  s_lshl_b32 s40, s27, 1\n  s_set_gpr_idx_idx s40\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "  s_lshl_b32 s40, s28, 1\n  s_set_gpr_idx_idx s40\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "  s_lshl_b32 s40, s29, 1\n  s_set_gpr_idx_idx s40\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "  s_lshl_b32 s40, s30, 1\n  s_set_gpr_idx_idx s40\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "  s_lshl_b32 s40, s31, 1\n  s_set_gpr_idx_idx s40\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "  s_lshl_b32 s40, s32, 1\n  s_set_gpr_idx_idx s40\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "  s_lshl_b32 s40, s33, 1\n  s_set_gpr_idx_idx s40\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "  s_lshl_b32 s40, s34, 1\n  s_set_gpr_idx_idx s40\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "  s_lshl_b32 s40, s35, 1\n  s_set_gpr_idx_idx s40\n"
        "  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n  v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "  s_set_gpr_idx_off\n"
        "  v_pk_add_f32 v[4:5], v[4:5], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n  v_pk_add_f32 v[6:7], v[6:7], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n"
        "  v_pk_add_f32 v[8:9], v[8:9], v[4:5]\n  v_pk_add_f32 v[10:11], v[10:11], v[6:7]\n"
        "  s_add_u32 s38, s38, 80\n  s_addc_u32 s39, s39, 0\n  s_sub_u32 s36, s36, 1\n  s_cmp_lg_u32 s36, 0\n  s_cbranch_scc1 1b\n"
        "  v_mov_b32 %[o0], v8\n"
        "  v_mov_b32 %[o1], v9\n"
        "  v_mov_b32 %[o2], v10\n"
        "  v_mov_b32 %[o3], v11\n"
        : [o0] "=v"(o0), [o1] "=v"(o1), [o2] "=v"(o2), [o3] "=v"(o3)
        : [tab] "s"(tp), [T] "s"(T), [seed] "v"(seed)
        : "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255", "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "s36", "s37", "s38", "s39", "s40", "m0", "scc", "memory");
    float *o = out + ((size_t)blockIdx.x * 64 + threadIdx.x) * 4;
    o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3;
}
int main()
{
    const int T = 1000, K = 20;
    std::vector<unsigned> tab((size_t)T * K);
    unsigned s = 12345;
    for (auto &v : tab) { s = s * 1664525u + 1013904223u; v = 2 * ((s >> 16) % 20); }
    unsigned *d_tab; float *d_out;
    const int blocks = 256 * 8 * 4;
    CHECK(hipMalloc(&d_tab, tab.size() * 4));
    CHECK(hipMalloc(&d_out, (size_t)blocks * 64 * 4 * 4));
    CHECK(hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_pk, dim3(blocks), dim3(64), 0, 0, d_tab, T, d_out);
        hipEventRecord(e1);
        CHECK(hipDeviceSynchronize());
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("%-12s %.3f ms  %.1f T draw-sites/s  %.2f cycles per draw (4 sites) per SIMD at 2.4 GHz\n", "pk", ms, (double)blocks * 256 * T * K / ms / 1e9, ms * 1e-3 * 2.4e9 * 1024 / ((double)blocks * T * K));
    }
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_pk_noidx, dim3(blocks), dim3(64), 0, 0, d_tab, T, d_out);
        hipEventRecord(e1);
        CHECK(hipDeviceSynchronize());
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("%-12s %.3f ms  %.1f T draw-sites/s  %.2f cycles per draw (4 sites) per SIMD at 2.4 GHz\n", "pk_noidx", ms, (double)blocks * 256 * T * K / ms / 1e9, ms * 1e-3 * 2.4e9 * 1024 / ((double)blocks * T * K));
    }
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_pk_noidx_nc, dim3(blocks), dim3(64), 0, 0, d_tab, T, d_out);
        hipEventRecord(e1);
        CHECK(hipDeviceSynchronize());
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("%-12s %.3f ms  %.1f T draw-sites/s  %.2f cycles per draw (4 sites) per SIMD at 2.4 GHz\n", "pk_noidx_nc", ms, (double)blocks * 256 * T * K / ms / 1e9, ms * 1e-3 * 2.4e9 * 1024 / ((double)blocks * T * K));
    }
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_pk_noidx_c, dim3(blocks), dim3(64), 0, 0, d_tab, T, d_out);
        hipEventRecord(e1);
        CHECK(hipDeviceSynchronize());
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("%-12s %.3f ms  %.1f T draw-sites/s  %.2f cycles per draw (4 sites) per SIMD at 2.4 GHz\n", "pk_noidx_c", ms, (double)blocks * 256 * T * K / ms / 1e9, ms * 1e-3 * 2.4e9 * 1024 / ((double)blocks * T * K));
    }
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_pk_nc, dim3(blocks), dim3(64), 0, 0, d_tab, T, d_out);
        hipEventRecord(e1);
        CHECK(hipDeviceSynchronize());
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("%-12s %.3f ms  %.1f T draw-sites/s  %.2f cycles per draw (4 sites) per SIMD at 2.4 GHz\n", "pk_nc", ms, (double)blocks * 256 * T * K / ms / 1e9, ms * 1e-3 * 2.4e9 * 1024 / ((double)blocks * T * K));
    }
    return 0;
}
