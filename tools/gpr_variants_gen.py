#!/usr/bin/env python3
"""SUPERSEDED by tools/valu_rate_bench.hip (round 4): this probe waits for a scalar load in every iteration with two waves to
cover it, so its 6-7 cycles per v_pk_mul_f32 / 3.05 per v_mul_f32 were its own loop, not the part (HISTORY.md section 4.2r); kept
because profiles/r02_gpr_variants.txt and r03_gpr_variants.txt came from it.

Generates tools/gpr_variants.hip: the inner loop of pool_reg_kernel (bags in VGPRs, wave-uniform draw index through
the VGPR index mode) with the multiply written six ways -- is a different instruction, or no index switching at all, any
faster than the kernel's two indexed v_pk_mul_f32 per draw of four sites?  (No: profiles/r02_gpr_variants.txt.)

Round 3 adds the occupancy experiment VERDICT r2 asked for: every variant above runs at 2 resident waves per SIMD (256
VGPRs, 4 sites per lane).  `s2_*` keep 2 sites per lane in v[64:127] (128 VGPRs -> 4 waves per SIMD, one packed multiply per
draw), `s1_*` keep 1 site per lane in v[32:63] (64 VGPRs -> 8 waves per SIMD, one scalar multiply per draw): if the ~7
cycles of a v_pk_mul_f32 were a per-wave issue limit rather than datapath occupancy, more waves would hide them.

    python tools/gpr_variants_gen.py && hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/gpr_variants.hip -o tools/gpr_variants
"""
import os
def body(variant):
    L=[]
    L.append('  s_mov_b64 s[38:39], %[tab]\\n')
    L.append('  s_mov_b32 s36, %[T]\\n')
    for r in (8,9,10,11): L.append('  v_mov_b32 v%d, 0\\n'%r)
    L.append('  .set i, 0\\n  .rept 128\\n  v_mov_b32 v3, i+1\\n  v_cvt_f32_i32 v3, v3\\n  v_fma_f32 v[128+i], %[seed], v3, 0.5\\n  .set i, i+1\\n  .endr\\n')
    L.append('1:\\n')
    L.append('  s_load_dwordx16 s[16:31], s[38:39], 0x0\\n  s_load_dwordx4 s[32:35], s[38:39], 0x40\\n')
    for r in (4,5,6,7): L.append('  v_mov_b32 v%d, 1.0\\n'%r)
    L.append('  s_waitcnt lgkmcnt(0)\\n')
    def mul():
        if variant in ('pkfma','pkfma_noidx'):
            return '  v_pk_fma_f32 v[4:5], v[128:129], v[4:5], 0 op_sel_hi:[1,1,0]\\n  v_pk_fma_f32 v[6:7], v[192:193], v[6:7], 0 op_sel_hi:[1,1,0]\\n'
        if variant in ('fma4','fma4_noidx'):
            return '  v_fma_f32 v4, v128, v4, 0\\n  v_fma_f32 v5, v129, v5, 0\\n  v_fma_f32 v6, v192, v6, 0\\n  v_fma_f32 v7, v193, v7, 0\\n'
        if variant == 'mac4':
            return '  v_mul_f32_e32 v4, v128, v4\\n  v_mul_f32_e32 v5, v129, v5\\n  v_mul_f32_e32 v6, v192, v6\\n  v_mul_f32_e32 v7, v193, v7\\n'
        if variant == 'pk_noidx_nc':      # no VGPR bank shared between the two 64-bit sources (bank = register number mod 4)
            return '  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\\n  v_pk_mul_f32 v[4:5], v[194:195], v[4:5]\\n'
        if variant == 'pk_noidx_c':       # both sources in the same two banks
            return '  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\\n  v_pk_mul_f32 v[6:7], v[194:195], v[6:7]\\n'
        if variant == 'pk_nc':            # indexed, bag pairs at a stride of four registers: always banks {0,1} / {2,3}
            return '  v_pk_mul_f32 v[6:7], v[128:129], v[6:7]\\n  v_pk_mul_f32 v[4:5], v[130:131], v[4:5]\\n'
        if variant in ('pk','pk_fixed','pk_noidx'):
            return '  v_pk_mul_f32 v[4:5], v[128:129], v[4:5]\\n  v_pk_mul_f32 v[6:7], v[192:193], v[6:7]\\n'
        return '  v_mul_f32 v4, v128, v4\\n  v_mul_f32 v5, v129, v5\\n  v_mul_f32 v6, v192, v6\\n  v_mul_f32 v7, v193, v7\\n'
    idx = variant in ('pk','mul4','pkfma','fma4','mac4','pk_nc')
    fixed = variant in ('pk_fixed','mul4_fixed')
    if variant == 'pk_nc':
        L.append('  s_lshl_b32 s40, s16, 1\\n  s_set_gpr_idx_on s40, gpr_idx(SRC0)\\n')
    elif idx or fixed: L.append('  s_set_gpr_idx_on s16, gpr_idx(SRC0)\\n')
    L.append(mul())
    for k in range(17,36):
        if variant == 'pk_nc': L.append('  s_lshl_b32 s40, s%d, 1\\n  s_set_gpr_idx_idx s40\\n'%k)
        elif idx: L.append('  s_set_gpr_idx_idx s%d\\n'%k)
        L.append(mul())
    if idx or fixed: L.append('  s_set_gpr_idx_off\\n')
    L.append('  v_pk_add_f32 v[4:5], v[4:5], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\\n  v_pk_add_f32 v[6:7], v[6:7], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\\n')
    L.append('  v_pk_add_f32 v[8:9], v[8:9], v[4:5]\\n  v_pk_add_f32 v[10:11], v[10:11], v[6:7]\\n')
    L.append('  s_add_u32 s38, s38, 80\\n  s_addc_u32 s39, s39, 0\\n  s_sub_u32 s36, s36, 1\\n  s_cmp_lg_u32 s36, 0\\n  s_cbranch_scc1 1b\\n')
    for i,r in enumerate((8,9,10,11)): L.append('  v_mov_b32 %%[o%d], v%d\\n'%(i,r))
    return ''.join('        "%s"\n'%x for x in L)

def body_occ(variant):
    '''sites per lane S in (2, 1): bags at v[BASE : BASE + 32*S], product in v[4:5] (S=2) or v4 (S=1)'''
    S = 2 if variant.startswith('s2') else 1
    BASE = 64 if S == 2 else 32
    idx = variant.endswith('_idx')
    L=[]
    L.append('  s_mov_b64 s[38:39], %[tab]\\n')
    L.append('  s_mov_b32 s36, %[T]\\n')
    for r in (8,9): L.append('  v_mov_b32 v%d, 0\\n'%r)
    L.append('  .set i, 0\\n  .rept %d\\n  v_mov_b32 v3, i+1\\n  v_cvt_f32_i32 v3, v3\\n  v_fma_f32 v[%d+i], %%[seed], v3, 0.5\\n  .set i, i+1\\n  .endr\\n'%(32*S, BASE))
    L.append('1:\\n')
    L.append('  s_load_dwordx16 s[16:31], s[38:39], 0x0\\n  s_load_dwordx4 s[32:35], s[38:39], 0x40\\n')
    for r in (4,5): L.append('  v_mov_b32 v%d, 1.0\\n'%r)
    L.append('  s_waitcnt lgkmcnt(0)\\n')
    if S == 2: mul = '  v_pk_mul_f32 v[4:5], v[%d:%d], v[4:5]\\n'%(BASE, BASE+1)
    else: mul = '  v_mul_f32 v4, v%d, v4\\n'%BASE
    # the table holds 2 x index (a register pair per entry); one register per entry needs index / 2
    def setidx(k, first):
        out = ''
        src = 's%d'%k
        if S == 1:
            out += '  s_lshr_b32 s40, s%d, 1\\n'%k
            src = 's40'
        out += ('  s_set_gpr_idx_on %s, gpr_idx(SRC0)\\n' if first else '  s_set_gpr_idx_idx %s\\n')%src
        return out
    if idx: L.append(setidx(16, True))
    L.append(mul)
    for k in range(17,36):
        if idx: L.append(setidx(k, False))
        L.append(mul)
    if idx: L.append('  s_set_gpr_idx_off\\n')
    if S == 2:
        L.append('  v_pk_add_f32 v[4:5], v[4:5], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\\n  v_pk_add_f32 v[8:9], v[8:9], v[4:5]\\n')
    else:
        L.append('  v_sub_f32 v4, 1.0, v4\\n  v_add_f32 v8, v8, v4\\n')
    L.append('  s_add_u32 s38, s38, 80\\n  s_addc_u32 s39, s39, 0\\n  s_sub_u32 s36, s36, 1\\n  s_cmp_lg_u32 s36, 0\\n  s_cbranch_scc1 1b\\n')
    L.append('  v_mov_b32 %[o0], v8\\n  v_mov_b32 %[o1], v9\\n')
    return ''.join('        "%s"\n'%x for x in L)
def clob_occ(variant):
    S = 2 if variant.startswith('s2') else 1
    BASE = 64 if S == 2 else 32
    return ', '.join('"v%d"'%i for i in list(range(3,10))+list(range(BASE, BASE+32*S)))+', '+', '.join('"s%d"'%i for i in range(16,41))+', "m0", "scc", "memory"'
occ_variants=['s2_idx','s2_noidx','s1_idx','s1_noidx']
clob=', '.join('"v%d"'%i for i in list(range(3,12))+list(range(128,256)))+', '+', '.join('"s%d"'%i for i in range(16,41))+', "m0", "scc", "memory"'
src='''#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\\n", #x, hipGetErrorString(e)); return 1; } } while (0)
'''
variants=['pk','pk_noidx']
for v in variants:
    src+='''__global__ __launch_bounds__(64) void k_%s(const unsigned *tab, int T, float *out)
{
    const float seed = (float)(threadIdx.x + 1) * 1e-5f + (float)blockIdx.x * 1e-9f;
    float o0, o1, o2, o3;
    const unsigned long long tp = (unsigned long long)tab;
    asm volatile(
%s        : [o0] "=v"(o0), [o1] "=v"(o1), [o2] "=v"(o2), [o3] "=v"(o3)
        : [tab] "s"(tp), [T] "s"(T), [seed] "v"(seed)
        : %s);
    float *o = out + ((size_t)blockIdx.x * 64 + threadIdx.x) * 4;
    o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3;
}
'''%(v,body(v),clob)
for v in occ_variants:
    src+='''__global__ __launch_bounds__(64) void k_%s(const unsigned *tab, int T, float *out)
{
    const float seed = (float)(threadIdx.x + 1) * 1e-5f + (float)blockIdx.x * 1e-9f;
    float o0, o1;
    const unsigned long long tp = (unsigned long long)tab;
    asm volatile(
%s        : [o0] "=v"(o0), [o1] "=v"(o1)
        : [tab] "s"(tp), [T] "s"(T), [seed] "v"(seed)
        : %s);
    float *o = out + ((size_t)blockIdx.x * 64 + threadIdx.x) * 2;
    o[0] = o0; o[1] = o1;
}
'''%(v,body_occ(v),clob_occ(v))
src+='''int main()
{
    const int T = 1000, K = 20;
    std::vector<unsigned> tab((size_t)T * K);
    unsigned s = 12345;
    for (auto &v : tab) { s = s * 1664525u + 1013904223u; v = 2 * ((s >> 16) % 20); }
    unsigned *d_tab; float *d_out;
    const int blocks = 256 * 8 * 4;
    CHECK(hipMalloc(&d_tab, tab.size() * 4));
    CHECK(hipMalloc(&d_out, (size_t)blocks * 64 * 4 * 4 * 2));
    CHECK(hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
'''
for v in variants:
    src+='''    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_%s, dim3(blocks), dim3(64), 0, 0, d_tab, T, d_out);
        hipEventRecord(e1);
        CHECK(hipDeviceSynchronize());
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("%%-12s %%.3f ms  %%.1f T draw-sites/s  %%.2f cycles per draw (4 sites) per SIMD at 2.4 GHz\\n", "%s", ms, (double)blocks * 256 * T * K / ms / 1e9, ms * 1e-3 * 2.4e9 * 1024 / ((double)blocks * T * K));
    }
'''%(v,v)
for v in occ_variants:
    S = 2 if v.startswith('s2') else 1
    src+='''    for (int rep = 0; rep < 2; rep++) {
        const int nb = blocks * %d;                       // the same 2 M sites: fewer sites per lane, more waves
        int occ = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_%s, 64, 0);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_%s, dim3(nb), dim3(64), 0, 0, d_tab, T, d_out);
        hipEventRecord(e1);
        CHECK(hipDeviceSynchronize());
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("%%-12s %%.3f ms  %%.1f T draw-sites/s  %%.2f cycles per draw (4 sites) per SIMD at 2.4 GHz  [%%d waves per SIMD]\\n", "%s", ms, (double)nb * 64 * %d * T * K / ms / 1e9, ms * 1e-3 * 2.4e9 * 1024 / ((double)nb * %d / 4.0 * T * K), occ / 4);
    }
'''%(4//S, v, v, v, S, S)
src+='    return 0;\n}\n'
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'gpr_variants.hip'),'w').write(src)
