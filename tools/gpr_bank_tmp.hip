#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(64) void k_old(const unsigned *tab, int T, float *out)
{
    const float seed = (float)(threadIdx.x + 1) * 1e-5f;
    float o0, o1;
    const unsigned long long tp = (unsigned long long)tab;
    asm volatile(
        "s_mov_b64 s[38:39], %[tab]\n"
        "s_mov_b32 s36, %[T]\n"
        "v_mov_b32 v8, 0\n"
        "v_mov_b32 v9, 0\n"
        ".set i, 0\n"
        ".rept 128\n"
        "v_mov_b32 v3, i+1\n"
        "v_cvt_f32_i32 v3, v3\n"
        "v_fma_f32 v[128+i], %[seed], v3, 0.5\n"
        ".set i, i+1\n"
        ".endr\n"
        "s_load_dwordx16 s[16:31], s[38:39], 0x0\n"
        "s_load_dwordx4 s[40:43], s[38:39], 0x40\n"
        "s_waitcnt lgkmcnt(0)\n"
        "1:\n"
        "v_mov_b32 v4, 1.0\n"
        "v_mov_b32 v5, 1.0\n"
        "v_mov_b32 v6, 1.0\n"
        "v_mov_b32 v7, 1.0\n"
        "s_set_gpr_idx_on s16, gpr_idx(SRC0)\n"
        "v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n"
        "v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "s_set_gpr_idx_idx s17\n"
        "v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n"
        "v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "s_set_gpr_idx_idx s18\n"
        "v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n"
        "v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "s_set_gpr_idx_idx s19\n"
        "v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n"
        "v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "s_set_gpr_idx_idx s20\n"
        "v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n"
        "v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "s_set_gpr_idx_idx s21\n"
        "v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n"
        "v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "s_set_gpr_idx_idx s22\n"
        "v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n"
        "v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "s_set_gpr_idx_idx s23\n"
        "v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n"
        "v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "s_set_gpr_idx_idx s24\n"
        "v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n"
        "v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "s_set_gpr_idx_idx s25\n"
        "v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n"
        "v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "s_set_gpr_idx_idx s26\n"
        "v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n"
        "v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "s_set_gpr_idx_idx s27\n"
        "v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n"
        "v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "s_set_gpr_idx_idx s28\n"
        "v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n"
        "v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "s_set_gpr_idx_idx s29\n"
        "v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n"
        "v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "s_set_gpr_idx_idx s30\n"
        "v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n"
        "v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "s_set_gpr_idx_idx s31\n"
        "v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n"
        "v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "s_set_gpr_idx_idx s40\n"
        "v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n"
        "v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "s_set_gpr_idx_idx s41\n"
        "v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n"
        "v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "s_set_gpr_idx_idx s42\n"
        "v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n"
        "v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "s_set_gpr_idx_idx s43\n"
        "v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\n"
        "v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\n"
        "s_set_gpr_idx_off\n"
        "v_pk_add_f32 v[8:9], v[8:9], v[4:5]\n"
        "v_pk_add_f32 v[8:9], v[8:9], v[6:7]\n"
        "s_sub_u32 s36, s36, 1\n"
        "s_cmp_lg_u32 s36, 0\n"
        "s_cbranch_scc1 1b\n"
        "v_mov_b32 %[o0], v8\n"
        "v_mov_b32 %[o1], v9\n"
        : [o0] "=v"(o0), [o1] "=v"(o1)
        : [tab] "s"(tp), [T] "s"(T), [seed] "v"(seed)
        : "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255", "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "scc", "memory");
    out[(size_t)blockIdx.x * 64 + threadIdx.x] = o0 + o1;
}

__global__ __launch_bounds__(64) void k_new(const unsigned *tab, int T, float *out)
{
    const float seed = (float)(threadIdx.x + 1) * 1e-5f;
    float o0, o1;
    const unsigned long long tp = (unsigned long long)tab;
    asm volatile(
        "s_mov_b64 s[38:39], %[tab]\n"
        "s_mov_b32 s36, %[T]\n"
        "v_mov_b32 v8, 0\n"
        "v_mov_b32 v9, 0\n"
        ".set i, 0\n"
        ".rept 128\n"
        "v_mov_b32 v3, i+1\n"
        "v_cvt_f32_i32 v3, v3\n"
        "v_fma_f32 v[128+i], %[seed], v3, 0.5\n"
        ".set i, i+1\n"
        ".endr\n"
        "s_load_dwordx16 s[16:31], s[38:39], 0x0\n"
        "s_load_dwordx4 s[40:43], s[38:39], 0x40\n"
        "s_waitcnt lgkmcnt(0)\n"
        "1:\n"
        "v_mov_b32 v4, 1.0\n"
        "v_mov_b32 v5, 1.0\n"
        "v_mov_b32 v6, 1.0\n"
        "v_mov_b32 v7, 1.0\n"
        "s_set_gpr_idx_on s16, gpr_idx(SRC0)\n"
        "v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n"
        "v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "s_set_gpr_idx_idx s17\n"
        "v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n"
        "v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "s_set_gpr_idx_idx s18\n"
        "v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n"
        "v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "s_set_gpr_idx_idx s19\n"
        "v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n"
        "v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "s_set_gpr_idx_idx s20\n"
        "v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n"
        "v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "s_set_gpr_idx_idx s21\n"
        "v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n"
        "v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "s_set_gpr_idx_idx s22\n"
        "v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n"
        "v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "s_set_gpr_idx_idx s23\n"
        "v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n"
        "v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "s_set_gpr_idx_idx s24\n"
        "v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n"
        "v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "s_set_gpr_idx_idx s25\n"
        "v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n"
        "v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "s_set_gpr_idx_idx s26\n"
        "v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n"
        "v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "s_set_gpr_idx_idx s27\n"
        "v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n"
        "v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "s_set_gpr_idx_idx s28\n"
        "v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n"
        "v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "s_set_gpr_idx_idx s29\n"
        "v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n"
        "v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "s_set_gpr_idx_idx s30\n"
        "v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n"
        "v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "s_set_gpr_idx_idx s31\n"
        "v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n"
        "v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "s_set_gpr_idx_idx s40\n"
        "v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n"
        "v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "s_set_gpr_idx_idx s41\n"
        "v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n"
        "v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "s_set_gpr_idx_idx s42\n"
        "v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n"
        "v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "s_set_gpr_idx_idx s43\n"
        "v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\n"
        "v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\n"
        "s_set_gpr_idx_off\n"
        "v_pk_add_f32 v[8:9], v[8:9], v[4:5]\n"
        "v_pk_add_f32 v[8:9], v[8:9], v[6:7]\n"
        "s_sub_u32 s36, s36, 1\n"
        "s_cmp_lg_u32 s36, 0\n"
        "s_cbranch_scc1 1b\n"
        "v_mov_b32 %[o0], v8\n"
        "v_mov_b32 %[o1], v9\n"
        : [o0] "=v"(o0), [o1] "=v"(o1)
        : [tab] "s"(tp), [T] "s"(T), [seed] "v"(seed)
        : "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255", "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "scc", "memory");
    out[(size_t)blockIdx.x * 64 + threadIdx.x] = o0 + o1;
}

template <class F> void run(const char *name, F f, const unsigned *tab, float *out)
{
    const int T = 1000, blocks = 256 * 4 * 8;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(f, dim3(blocks), dim3(64), 0, 0, tab, T, out);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double steps = (double)blocks / 1024 * T * 20;
        if (rep) printf("%s: %.3f ms -> %.1f cycles per draw-step per SIMD (2 pk_mul)\n", name, ms, ms * 1e-3 * 2.4e9 / steps);
    }
}
int main()
{
    std::vector<unsigned> told(1000 * 20 + 64), tnew(1000 * 20 + 64);
    unsigned s = 12345;
    for (size_t i = 0; i < told.size(); i++) { s = s * 1664525u + 1013904223u; unsigned e = (s >> 16) % 20; told[i] = 2 * e; tnew[i] = 4 * e; }
    unsigned *d_old, *d_new; float *d_out;
    (void)hipMalloc(&d_old, told.size() * 4); (void)hipMalloc(&d_new, tnew.size() * 4); (void)hipMalloc(&d_out, 256 * 4 * 8 * 64 * 4);
    (void)hipMemcpy(d_old, told.data(), told.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_new, tnew.data(), tnew.size() * 4, hipMemcpyHostToDevice);
    run("pairs at 2e / 2e+64 (index 2e)", k_old, d_old, d_out);
    run("quads at 4e, products in the opposite banks (index 4e)", k_new, d_new, d_out);
    return 0;
}
