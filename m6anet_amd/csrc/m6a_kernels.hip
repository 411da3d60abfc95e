// m6a_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the m6A inference hot path.
//
//   enc_kernel, enc_csite_kernel
//                      read encoder: [x(9) | emb(6) | 1] -> 150 -> batch norm -> ReLU -> 32 -> ReLU
//                      -> 1 -> sigmoid, all in registers on v_mfma_f32_32x32x2_f32 (two IEEE fmas per output, k = the
//                      half-0 operand first), every sum through layer 2 in the order the reference's float32 arithmetic
//                      runs it -- enc_kernel all the way to the probability (the reference's bits); the csite variant
//                      folds the per-site constants (12 K-slots, bags >= 16 reads) and keeps a plain 32 -> 1 sum.
//   pool_scan_start_kernel + pool_scan_site_kernel, pool_scan_group_kernel
//                      site pooling, exact NumPy-stream replay, any bag sizes: a counting pass per
//                      flush group finds where each site starts in the shared MT19937 word stream
//                      (masked rejection), then one wavefront per site compacts its accepted draws
//                      through LDS and multiplies the 20-term products; or one wavefront per group.
//                      (Large ragged jobs with bags <= 4096 take pool_rtab_kernel, m6a_pool_rtab.hip,
//                      instead: per-bag-size index tables, no compaction at all.)
//   pool_table_kernel  same result when every bag has the same size n <= 32: the accepted index
//                      sequence is then identical in every flush group, so it is a precomputed table
//                      and the kernel is a pure LDS gather, 8 sites per pass.  (The default for uniform
//                      bags is pool_reg_kernel in m6a_pool_reg.hip: bags in registers, no LDS at all.)
//   sampled_noisy_or_kernel, mean_over_passes_kernel: the validation-style forward.
//   bag_noisy_or_kernel, iota_off_kernel, bag_minmax_kernel: small helpers.
//
// Reference lines each kernel restates are cited at the kernel.  Wave = 64 lanes throughout.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "m6a_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// sum over each group of 8 consecutive lanes, in NumPy's pairwise-leaf order
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)); every lane of the group gets it.  DPP only (no LDS traffic):
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], then row_half_mirror pairs the two quads.
__device__ __forceinline__ float chain8_sum(float r)
{
    r += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, r), 0xB1, 0xf, 0xf, false));
    r += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, r), 0x4E, 0xf, 0xf, false));
    r += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, r), 0x141, 0xf, 0xf, false));
    return r;
}

__device__ __forceinline__ int wave_sum_i32(int v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// Accumulator registers of layer-1 unit group m that feed layer 2: register q of lane half h holds hidden unit
// 32m + 2q + h (the host wires the units to the tile rows that way, m6a_api.hip build_fragments), so in the last group
// (units 128..159) registers 12..15 are units 152..159: padding beyond the 150 hidden units, the constant-one unit (150)
// and its zero partner (151) -- zero weights, skipped (76 instead of 80 layer-2 MFMAs per tile, same bits).
#define L2_REGS(m) ((m) == 4 ? 12 : 16)

// Layer 2's ReLU, doubled: x + |x| is exactly 2*max(x, 0) (2x for x > 0 -- doubling is exact --, x + (-x) = +0 otherwise,
// -0 + 0 = +0), and the 0.5 rides in W3 (halving a normal float is exact too), so the epilogue's fmaf chain forms the
// products it would form from max(x, 0).  Why: max is a HALF-rate VALU operation on gfx950 -- v_max_i32, v_max_f32 and
// v_med3_f32 all take 4.2-4.9 cycles per wave64 where v_add_f32 / v_mul_f32 take 2.2-2.4 (tools/valu_rate_bench,
// profiles/r04_valu_rate.json) -- on the datapath the f32 MFMAs use, and |x| is a free source modifier of v_add_f32.  (It
// was layer 1's ReLU too, 92 per tile, until the batch norm moved into the kernel: bn_relu below.  Rounds 1-3 used one
// v_max_i32 on the float's bits; fmaxf() costs two VALU ops, a canonicalising v_max first.)
// One difference at infinity: a pre-activation of -inf gives -inf + inf = NaN where max gives 0 -- it takes |x * w| beyond 3e38 to get there.
__device__ __forceinline__ float relu2(float x)
{
    return x + __builtin_fabsf(x);
}

// Eval batch norm + ReLU of a finished layer-1 unit tile, rounded as the reference rounds it: torch's batch_norm on the CPU
// forms alpha = gamma * invstd and beta = fma(-mean, alpha, bias) once and applies ONE fma(y, alpha, beta) per hidden unit
// (pinned bit for bit against torch's own tensors, tools/emulate_encoder.py).  Rounds 1-3 folded alpha into W1 on the host:
// one rounding fewer than the reference, and twice its distance from the reference's values.
//
// Batch norm AND ReLU are one VALU instruction: the host scales alpha and beta by 2^-64 (exact), so fma(y, alpha', beta') is
// exactly 2^-64 * fma(y, alpha, beta), and the instruction's clamp modifier (result to [0, 1]) is the ReLU of every
// activation below 2^64; layer 2's weights carry the 2^64 back (exact again), so its MFMAs form the very products relu(h)
// would give.  fma then x + |x| -- two instructions on the datapath the f32 MFMAs use -- cost the 12-slot kernel 3 %
// (2.202 -> 2.135 ms on the bench shape, before the loads were hidden).  What the trick costs: an activation beyond 2^64 = 1.8e19 saturates there (the
// reference carries it on towards inf; normalised signal features give |h| < 1e4, tests go to 1e7), and one below 2^-62
// loses low bits (a contribution under 1e-19).  Both kernels clear MODE.DX10_CLAMP first: with it set (the default for
// compute kernels) clamp turns NaN into 0, and a NaN feature must come out as a NaN probability, as the reference's does
// (tests/test_gpu_parity.py::test_encoder_nan_and_huge_features).
//
// The (alpha', beta') pairs sit in LDS as [unit tile][lane half][register]: 16 bytes = two hidden units per load, all 32
// lanes of a half reading the same address (a broadcast, no conflicts) -- 38 ds_read_b128 per tile, whose latency
// layer2_with_bn (below) hides behind blocks of MFMAs.
// Wave priority through a tile (round 6).  Two waves share a SIMD.  A tile is ~8 000 cycles of MFMAs with the batch-norm fmas woven in
// (the BODY) followed by the 32 -> 1 layer and the sigmoid (the EPILOGUE: ~85 mostly dependent VALU instructions, no MFMA).  What the
// kernel's own clock shows (tools/encoder_timeline.py: s_memtime stamps of every wave, pairs matched by hardware slot;
// profiles/r06_encoder_timeline.json): a VALU instruction cannot start while a 16-pass MFMA occupies the datapath, so next to a partner
// that streams MFMAs a dependent chain advances ONE instruction per MFMA (64 cycles).
//   * body 3, epilogue 0 (the round's first setting, +1 % over none): the wave that finishes its body first crawls through its epilogue
//     on the slots its partner leaves, until the partner's body ends too -- the two lock IN PHASE (partner's tile start at 0.04 of the
//     own tile), both are outside their bodies 6.8 % of the time, and the matrix pipe has nothing to run then;
//   * body 0, epilogue 3 (this): the epilogue wins every arbitration, so it ends as early as the partner's MFMA boundaries allow; the
//     pair settles half a tile apart (phase 0.5-0.6), both-outside falls to 0.5 %, and the partner's MFMAs run through:
//     enc_site16_kernel 2.229 -> 2.201 ms, enc_csite_kernel 2.070 -> 2.035, enc_kernel 2.241 -> 2.213 per 20 M reads (four interleaved
//     legs, profiles/r06_encoder_ab_phase.json; epilogue 1 or body 1: the same within noise; starting the second wave of a SIMD half a
//     tile late on top: +0.1-0.3 %, not taken; priorities inside the body -- the fmas above or below the MFMAs -- cost 1 %).
// Same instructions, same bits.  -DM6A_AB_NO_PRIO / -DM6A_AB_PRIO_BODY=.. -DM6A_AB_PRIO_EPI=.. (tools/encoder_ab.py) build the others.
#ifndef M6A_AB_PRIO_BODY
#define M6A_AB_PRIO_BODY 0
#define M6A_AB_PRIO_EPI 3
#endif
__device__ __forceinline__ void tile_body_priority()
{
#ifndef M6A_AB_NO_PRIO
    __builtin_amdgcn_s_setprio(M6A_AB_PRIO_BODY);
#endif
}
__device__ __forceinline__ void tile_epilogue_priority()
{
#ifndef M6A_AB_NO_PRIO
    __builtin_amdgcn_s_setprio(M6A_AB_PRIO_EPI);
#endif
}
#ifdef M6A_AB_PHASE
#ifndef M6A_AB_PHASE_SLEEP
#define M6A_AB_PHASE_SLEEP 127
#endif
// A/B build only: the second workgroup to arrive on a CU starts its tile loop M6A_AB_PHASE x 8 128 cycles late, so the two waves of a
// SIMD run half a tile apart (atomicInc wraps 0 -> 1 -> 0: the counter is back at 0 when both have arrived).
__device__ unsigned m6a_ab_cu_slot[2048];
__device__ __forceinline__ void phase_shift_second_workgroup()
{
#ifdef M6A_AB_PHASE_HWID
    // the partner by its hardware wave slot: the two waves of a SIMD sit in slots 0 and 1 (tools/encoder_timeline.py: every pair)
    if (__builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11)) & 1)
        for (int i = 0; i < M6A_AB_PHASE; i++) __builtin_amdgcn_s_sleep(M6A_AB_PHASE_SLEEP);
    return;
#endif
    __shared__ unsigned s_slot;
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) & 7;
        s_slot = atomicInc(&m6a_ab_cu_slot[(xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)], 1u);
    }
    __syncthreads();
    if (s_slot)
        for (int i = 0; i < M6A_AB_PHASE; i++) __builtin_amdgcn_s_sleep(M6A_AB_PHASE_SLEEP);
}
#else
__device__ __forceinline__ void phase_shift_second_workgroup() {}
#endif

__device__ __forceinline__ void clamp_keeps_nan()
{
    __builtin_amdgcn_s_setreg(1 | (8 << 6) | (0 << 11), 0);      // hwreg(HW_REG_MODE, offset 8, size 1) = DX10_CLAMP
}
__device__ __forceinline__ float bn_relu(float y, float alpha, float beta)
{
    float r;
    asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(y), "v"(alpha), "v"(beta));
    return r;
}
// Layer 2 of one unit tile, with the batch norm of its inputs woven in: blocks of BN_BLOCK hidden units -- clamp-fma them
// from the pairs in `pq`, fetch the NEXT block's pairs into `pq` (this tile's, or the first block of unit tile `m_next`),
// issue the block's MFMAs.  The loads are issued with a block of MFMAs (4 x 64 cycles of matrix pipe) in front of their
// use, so the wave never waits for LDS; 8 registers hold the pairs.  (All 16 units first and then 16 MFMAs back to back --
// how rounds 1-3 ran the plain ReLU, a VALU instruction between EVERY two dependent MFMAs costing the pipe 6 % then -- left
// the 38 loads' latency exposed: 2.135-2.15 ms for the 12-slot kernel against 2.07-2.09 this way, which is what the kernel
// took with the batch norm folded into the weights; blocks of 8 need 16 registers the 12-slot kernel does not have (spills:
// 2.25), blocks of 2 measure 2.10.  profiles/r04_encoder_order_timing.txt.)
#ifndef BN_BLOCK
#define BN_BLOCK 4
#endif
struct BnPairs { float4 v[BN_BLOCK / 2]; };
__device__ __forceinline__ void bn_pairs_load(BnPairs &pq, const float *p)
{
#pragma unroll
    for (int i = 0; i < BN_BLOCK / 2; i++) pq.v[i] = *(const float4 *)(p + 4 * i);
}
template <int M>
__device__ __forceinline__ void layer2_with_bn(f32x16 &acc2, f32x16 &cur, const float (&w2)[80], BnPairs &pq, const float *bn_half)
{
    constexpr int n = L2_REGS(M), m_next = M < 4 ? M + 1 : 0;
#pragma unroll
    for (int b = 0; b * BN_BLOCK < n; b++) {
#ifdef M6A_AB_PRIO_BLOCK_VALU
        __builtin_amdgcn_s_setprio(M6A_AB_PRIO_BLOCK_VALU);               // A/B build only: the block's fmas at another priority than its MFMAs
#endif
#pragma unroll
        for (int i = 0; i < BN_BLOCK / 2; i++) {
            const int q = b * BN_BLOCK + 2 * i;
#ifdef M6A_AB_BN_PK
            // A/B build only (tools/encoder_ab.py): two hidden units per VALU instruction -- v_pk_fma_f32 with the clamp modifier on the
            // register pair (cur[q], cur[q+1]); the kernel's preamble stores the pairs as (alpha, alpha', beta, beta').  Same fmas, same bits.
            f32x2 y = {cur[q], cur[q + 1]};
            const f32x2 al = {pq.v[i].x, pq.v[i].y}, be = {pq.v[i].z, pq.v[i].w};
            asm("v_pk_fma_f32 %0, %0, %1, %2 clamp" : "+v"(y) : "v"(al), "v"(be));
            cur[q] = y.x;
            cur[q + 1] = y.y;
#else
#ifndef M6A_AB_NO_BN                    // knock-out build only (WRONG results): layer 2 straight on layer 1's accumulators -- no fmas, no pair loads
            cur[q] = bn_relu(cur[q], pq.v[i].x, pq.v[i].y);
            cur[q + 1] = bn_relu(cur[q + 1], pq.v[i].z, pq.v[i].w);
#endif
#endif
        }
#ifndef M6A_AB_NO_BN
        bn_pairs_load(pq, (b + 1) * BN_BLOCK < n ? bn_half + M * 64 + 2 * (b + 1) * BN_BLOCK : bn_half + m_next * 64);
#endif
        __builtin_amdgcn_sched_barrier(0);
#ifdef M6A_AB_PRIO_BLOCK_VALU
        __builtin_amdgcn_s_setprio(M6A_AB_PRIO_BLOCK_MFMA);
#endif
#pragma unroll
        for (int q = b * BN_BLOCK; q < (b + 1) * BN_BLOCK; q++)
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[M * 16 + q], cur[q], acc2, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

#ifdef M6A_AB_W3
// A/B build only (tools/encoder_ab.py, -DM6A_AB_W3): THREE waves per SIMD.  Layer 2's 80 A-operand registers per lane are what stands between the kernel
// and 168 registers, so they live in LDS as [block of four hidden units][lane] float4 and come in one block ahead of their MFMAs, behind the block's first
// MFMA, into two alternating float4 buffers (19 blocks per tile: the last block refills buffer 0 for the next tile after its own MFMAs).
template <int M>
__device__ __forceinline__ void layer2_with_bn_lds(f32x16 &acc2, f32x16 &cur, float4 (&wq)[2], const float4 *w2lane, BnPairs &pq, const float *bn_half)
{
    static_assert(BN_BLOCK == 4, "one float4 of A operands per block");
    constexpr int n = L2_REGS(M), m_next = M < 4 ? M + 1 : 0;
#pragma unroll
    for (int b = 0; b * 4 < n; b++) {
        const int g = M * 4 + b;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int q = b * 4 + 2 * i;
            cur[q] = bn_relu(cur[q], pq.v[i].x, pq.v[i].y);
            cur[q + 1] = bn_relu(cur[q + 1], pq.v[i].z, pq.v[i].w);
        }
        bn_pairs_load(pq, (b + 1) * 4 < n ? bn_half + M * 64 + 2 * (b + 1) * 4 : bn_half + m_next * 64);
        __builtin_amdgcn_sched_barrier(0);
        const float4 w = wq[g & 1];
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, cur[b * 4], acc2, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (g < 18) wq[(g + 1) & 1] = w2lane[(g + 1) * 64];
        __builtin_amdgcn_sched_barrier(0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, cur[b * 4 + 1], acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, cur[b * 4 + 2], acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, cur[b * 4 + 3], acc2, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (g == 18) wq[0] = w2lane[0];
    }
}
#define M6A_SITE16_WAVES_PER_SIMD 3
#else
#define M6A_SITE16_WAVES_PER_SIMD 2
#endif

// exp as torch's vectorised sigmoid computes it -- Sleef's expf with the 1.0-ulp bound (sleefsimdsp.c `xexpf`; restated from the
// published algorithm): Cody-Waite reduction by ln 2 in two parts, a degree-6 polynomial in fma form,
// 2^q applied as two factors.  Only the 16-slot kernel uses it (its read probabilities are the reference's bits, below).
__device__ __forceinline__ float sleef_expf_u10(float d)
{
    const float qf = __builtin_rintf(d * 1.442695040888963407359924681001892137426645954152985934135449406931f);
    const int q = (int)qf;
    float s = __builtin_fmaf(qf, -0.693145751953125f, d);
    s = __builtin_fmaf(qf, -1.428606765330187045e-06f, s);
    float u = 0.000198527617612853646278381f;
    u = __builtin_fmaf(u, s, 0.00139304355252534151077271f);
    u = __builtin_fmaf(u, s, 0.00833336077630519866943359f);
    u = __builtin_fmaf(u, s, 0.0416664853692054748535156f);
    u = __builtin_fmaf(u, s, 0.166666671633720397949219f);
    u = __builtin_fmaf(u, s, 0.5f);
    u = 1.0f + __builtin_fmaf(s * s, u, s);
    const int q1 = q >> 1;
    u = u * ldexpf(1.0f, q1) * ldexpf(1.0f, q - q1);
    u = d < -104.0f ? 0.0f : u;
    return d > 104.0f ? __builtin_inff() : u;
}

// Linear(32, 1) in the order of the sgemv behind torch's addmm on the machine the reference captures were made on (MKL,
// AVX-512; rows of a batch in groups of four -- every row of a job of 20-read bags; found by probing the module with
// cancellation triples): s = x0*w0; 16 lanes of products k = 1 + l with lane 0 = fma(x1, w1, s),
// reduced by a butterfly l+8, l+4, l+2, l+1; the same for k = 17 + l (lane 15 empty) with lane 0 = fma(x17, w17, sum so far).
// The host wires layer 2's output units to the accumulator rows so that lane half 0 holds the EVEN lanes of both vectors
// (k = 1, 3, .., 15 in registers 0..7; 17, 19, .., 31 in 8..15) and half 1 the odd lanes (k = 2, 4, .., 16; 18, .., 30) plus
// k = 0 in register 15: every butterfly level but the last stays inside a half (lane l <-> register l >> 1), the last is the
// exchange between the halves.  t[q] = 2 relu(acc2[q]), w[q] = 0.5 W3[unit]: the products are relu * W3 exactly.
__device__ __forceinline__ float gemv32_as_mkl(const f32x16 &acc2, const float (&w)[16], int half)
{
    float r[16];
#pragma unroll
    for (int q = 0; q < 16; q++) r[q] = relu2(acc2[q]);
    const float k0 = r[15] * w[15];                                   // half 1: x0*w0; half 0: the product of k = 31
    // every exchange is executed by ALL lanes and selected afterwards (a shuffle under a divergent branch reads zeros from
    // the lanes that did not take it; the empty asm keeps the compiler from sinking it into the select)
    float k0_other = __shfl_xor(k0, 32, 64);
    asm volatile("" : "+v"(k0_other));
    const float s0 = half ? 0.0f : k0_other;                          // fma(x, w, +0) is the rounded product itself
    float v[8];
    v[0] = __builtin_fmaf(r[0], w[0], s0);
#pragma unroll
    for (int q = 1; q < 8; q++) v[q] = r[q] * w[q];
    float c = ((v[0] + v[4]) + (v[2] + v[6])) + ((v[1] + v[5]) + (v[3] + v[7]));
    float c_other = __shfl_xor(c, 32, 64);
    asm volatile("" : "+v"(c_other));
    const float t1 = c + c_other;
    v[0] = __builtin_fmaf(r[8], w[8], half ? 0.0f : t1);
#pragma unroll
    for (int q = 1; q < 7; q++) v[q] = r[8 + q] * w[8 + q];
    v[7] = half ? 0.0f : k0;
    c = ((v[0] + v[4]) + (v[2] + v[6])) + ((v[1] + v[5]) + (v[3] + v[7]));
    c_other = __shfl_xor(c, 32, 64);
    asm volatile("" : "+v"(c_other));
    return c + c_other;
}

// A value the program knows to be wave-uniform, made provably so: addresses built from it use
// scalar loads (s_load, counted by lgkmcnt) instead of vector loads (in-order vmcnt queue).
__device__ __forceinline__ int64_t uniform_i64(int64_t v)
{
    return ((int64_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

// Orders this wave's LDS traffic: LDS ops of one wave execute in issue order, so a compiler
// fence (no s_barrier) is all a single-wave producer/consumer needs.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// =====================================================================================
// Read encoder  (reference: m6anet/model/model_blocks/blocks.py:116-126 view, :194-205
// embedding, :55-66 concat, :257-266 Linear+BN+ReLU twice per m6anet.toml:16-28;
// pooling_blocks.py:52 Linear(32,1)+Sigmoid, as called from inference_utils.py:35-37)
//
// One wavefront owns a tile of 32 reads.  Everything is computed transposed,
//     H1^T[160 x 32] = W1aug[160 x 16] . F^T[16 x 32],   H2^T[32 x 32] = W2aug[32 x 160] . H1^T,
// so the reads sit on the MFMA N axis (lane & 31) in both layers and layer 1's accumulator
// registers ARE layer 2's B operands: an MFMA K-step takes B[k = h][n = lane&31] from register q of the
// two lane halves, and D register q of half h holds row (q&3) + 8(q>>2) + 4h of the tile.  WHICH hidden
// unit a row is, is the host's choice (the rows of W1 can be handed to the MFMA in any order): row -> unit
// 32m + 2q + h, so that layer 2's step q adds units 32m+2q and 32m+2q+1, in that order -- k = 0, 1, ..., 149,
// then b2: the order the reference's sgemm adds them in (acc = fma(h[k], W2[o][k], acc), then + b2; pinned
// against torch's tensors bit for bit, tools/emulate_encoder.py).  No LDS, no shuffles between layers.
//   F (16 features) = x0..x8, e0..e5 (three 2-float embeddings), 1.0 -- the constant adds b1 last, as the
//   reference does; batch norm + ReLU is one clamped fma per unit on the finished tile (bn_relu above);
//   hidden unit 150 is wired to the constant 1.0 (alpha 1, beta 0) and adds b2 through W2aug[:,150];
//   units 151..159 are zero padding.
// K slot 2s + h holds feature 2s + h: half 0 lanes load x0, x2, x4, x6, x8, e1, e3, e5; half 1 lanes x1, x3,
// x5, x7, e0, e2, e4, 1.  With that, layers 1 and 2 of this kernel are the reference's bits; the epilogue
// (gemv32_as_mkl, sleef_expf_u10 above) carries that on to the probability.
// =====================================================================================
// Input pipeline: the features of tile t+1 are fetched while tile t is on the matrix pipe.  The
// site lookup is a dependent chain (CSR offsets -> k-mer ids -> embedding rows); its three links
// are issued between the five 24-MFMA groups of the current tile, so no link ever waits in front
// of an MFMA.  A 32-read tile that starts in site `a` normally ends by site a+2 (bags >= 16
// reads); off[a+1..a+3] are fetched up front (wave-uniform) and lanes pick their site by
// comparison.  Smaller bags take the (rare) per-lane walk.
struct EncTile {
    float f[8];        // this lane's 8 B-operand features of the tile
    int64_t s_last;    // site of the tile's last read (wave-uniform)
};

__global__ __launch_bounds__(256, 2) void enc_kernel(EncArgs a)
{
    clamp_keeps_nan();
    m6a_clk_stamp(a.clk, 0);
    __shared__ float s_emb[132];
    __shared__ __attribute__((aligned(16))) float s_bn[M6A_BN_FLOATS];
    for (int i = threadIdx.x; i < 132; i += 256) s_emb[i] = a.emb[i];
#ifdef M6A_AB_BN_PK
    for (int i = threadIdx.x; i < M6A_BN_FLOATS; i += 256) s_bn[i] = a.bn[(i & ~3) | ((i & 1) << 1) | ((i & 2) >> 1)];   // (a, b, a', b') -> (a, a', b, b')
#else
    for (int i = threadIdx.x; i < M6A_BN_FLOATS; i += 256) s_bn[i] = a.bn[i];
#endif
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int col = lane & 31;
    const int half = lane >> 5;
    const float *bn_half = s_bn + half * 32;
    const int64_t wave = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // scalar tile loop
    const int64_t tile0 = wave * a.tiles_per_wave;
    if (tile0 >= a.n_tiles) return;
    const int64_t tile1 = (tile0 + a.tiles_per_wave < a.n_tiles) ? tile0 + a.tiles_per_wave : a.n_tiles;

    // weight fragments, lane-major on the host side: one coalesced load per register
    float w1[40], w2[80], w3[16];
#pragma unroll
    for (int i = 0; i < 40; i++) w1[i] = a.wfrag[i * 64 + lane];
#pragma unroll
    for (int i = 0; i < 80; i++) w2[i] = a.wfrag[(40 + i) * 64 + lane];
#pragma unroll
    for (int i = 0; i < 16; i++) w3[i] = a.wfrag[(120 + i) * 64 + lane];

    // site of the wave's first read: largest s with off[s] <= r (upper_bound - 1)
    int64_t s_base;
    {
        const int64_t r = tile0 * 32;
        int64_t lo = 0, hi = a.n_sites;          // invariant: off[lo] <= r < off[hi]
        while (hi - lo > 1) {
            int64_t mid = (lo + hi) >> 1;
            if (a.off[mid] <= r) lo = mid; else hi = mid;
        }
        s_base = lo;
    }
    const int64_t last_site = a.n_sites - 1;

    // ---- the three links of the input chain for one tile -------------------------------------
    // link 0: the three CSR offsets after the base site (base is wave-uniform)
    auto link0 = [&](int64_t base, int64_t (&o)[3]) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int64_t si = base + 1 + i;
            o[i] = a.off[si <= a.n_sites ? si : a.n_sites];
        }
    };
    // link 1: pick the lane's site, issue its k-mer id loads, THEN the x loads: vector loads retire
    // in order, so the k-mer ids (needed two links from now) must not queue behind the x stream
    // (needed only by the next tile)
    auto link1 = [&](int64_t tile, int64_t base, const int64_t (&o)[3], int64_t &s, int (&km)[3], float (&x)[8]) {
        const int64_t r = tile * 32 + col;
        const int64_t rc = r < a.n_reads ? r : a.n_reads - 1;
        s = base + (rc >= o[0] ? 1 : 0) + (rc >= o[1] ? 1 : 0);
        if (__any(rc >= o[2])) {                 // bags smaller than 16 reads: walk
            s = base;
            while (a.off[s + 1] <= rc) ++s;
        }
        s = s < last_site ? s : last_site;
        const uint8_t *kp = a.site_kmers + s * 3;
        km[0] = kp[0]; km[1] = kp[1]; km[2] = kp[2];
        // K slot 2i + half holds feature 2i + half (the order the reference's dot product adds them in): half 0 loads
        // x0, x2, x4, x6, x8, half 1 x1, x3, x5, x7 (and x7 again, which link 2 overwrites) -- every lane issues the same
        // five loads, no divergent branch, so nothing has to be merged and waited for here
        const float *xp = a.X + rc * 9 + half;
#pragma unroll
        for (int i = 0; i < 4; i++) x[i] = xp[2 * i];
        x[4] = xp[half ? 6 : 8];
    };
    // link 2: embedding floats from LDS: features 9..14 = e0..e5 (float c&1 of k-mer c>>1), feature 15 = the constant 1
    // that adds b1; half 0 takes the odd floats into slots 5..7, half 1 the even ones into slots 4..6
    auto link2 = [&](const int (&km)[3], float (&x)[8]) {
        const float e0 = s_emb[2 * km[0] + 1 - half], e1 = s_emb[2 * km[1] + 1 - half], e2 = s_emb[2 * km[2] + 1 - half];
        x[4] = half ? e0 : x[4];
        x[5] = half ? e1 : e0;
        x[6] = half ? e2 : e1;
        x[7] = half ? 1.0f : e2;
    };

    // prologue: first tile, unpipelined
    float f[8];
    s_base = uniform_i64(s_base);
    {
        int64_t o[3], s;
        int km[3];
        link0(s_base, o);
        link1(tile0, s_base, o, s, km, f);
        link2(km, f);
        s_base = uniform_i64(__shfl(s, 31, 64));
    }

    BnPairs bnq;
    bn_pairs_load(bnq, bn_half);
    for (int64_t tile = tile0; tile < tile1; ++tile) {
        // the chain always runs (for the last tile it refetches that tile): no guard, no merge
        const int64_t tn = tile + 1 < tile1 ? tile + 1 : tile;
        float fn[8];
        int64_t o[3], sn;
        int km[3];
        link0(s_base, o);
        tile_body_priority();

        // Layer 1 of unit-tile m+1 is issued ahead of layer 2 of unit-tile m (ping-pong
        // accumulators).  Batch norm + ReLU of the finished tile and the layer-2 MFMAs that consume it go in blocks
        // of four hidden units (layer2_with_bn): four clamped fmas, the next block's pairs fetched from LDS, four MFMAs.
        f32x16 acc2, h1a, h1b;
#pragma unroll
        for (int q = 0; q < 16; q++) { acc2[q] = 0.0f; h1a[q] = 0.0f; }
#pragma unroll
        for (int st = 0; st < 8; st++)
            h1a = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[st], f[st], h1a, 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 5; m++) {
            f32x16 &cur = (m & 1) ? h1b : h1a;
            f32x16 &nxt = (m & 1) ? h1a : h1b;
            if (m < 4) {
#pragma unroll
                for (int q = 0; q < 16; q++) nxt[q] = 0.0f;
#pragma unroll
                for (int st = 0; st < 8; st++)
                    nxt = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[(m + 1) * 8 + st], f[st], nxt, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (m == 0) link1(tn, s_base, o, sn, km, fn);
            if (m == 2) link2(km, fn);
            if (m == 0) layer2_with_bn<0>(acc2, cur, w2, bnq, bn_half);
            if (m == 1) layer2_with_bn<1>(acc2, cur, w2, bnq, bn_half);
            if (m == 2) layer2_with_bn<2>(acc2, cur, w2, bnq, bn_half);
            if (m == 3) layer2_with_bn<3>(acc2, cur, w2, bnq, bn_half);
            if (m == 4) layer2_with_bn<4>(acc2, cur, w2, bnq, bn_half);
        }
        tile_epilogue_priority();
        const float z = gemv32_as_mkl(acc2, w3, half) + a.b3;
        const float p = 1.0f / (1.0f + sleef_expf_u10(-z));
        const int64_t r = tile * 32 + col;
        if (half == 0 && r < a.n_reads) a.read_prob[r] = p;
        if (tn != tile) s_base = uniform_i64(__shfl(sn, 31, 64));
#pragma unroll
        for (int i = 0; i < 8; i++) f[i] = fn[i];
    }
    m6a_clk_stamp(a.clk, 1);
}

// =====================================================================================
// Read encoder, 16 slots, bags >= 16 reads: enc_kernel's arithmetic behind enc_csite_kernel's input chain.
//
// Every float32 operation is enc_kernel's (same fragments, same K-slot wiring, same epilogue: the reference's bits); what
// differs is how a lane finds its site and its six embedding floats.  enc_kernel walks the CSR array per lane with 64-bit
// indices (vector loads of off[], three k-mer byte loads and three LDS reads per lane, 64-bit clamps on the VALU -- the
// datapath the f32 MFMAs use); here, as in the 12-slot kernel below, a 32-read tile spans at most three sites, so
//   link0  off[a+1..a+3] come through scalar loads (constant address space), tile and site indices are 32-bit SALU values;
//   link1  the lane's site is `rel` = 0, 1, 2 relative to the wave-uniform base, by two 32-bit compares; lanes 0..17 fetch
//          ONE k-mer id byte each (float q of site a + q/6), then the x loads;
//   link2  the embedding float, one LDS read per lane;
//   link3  three ds_bpermute (LDS crossbar, not the VALU) hand every lane its site's floats: half 0 the odd ones
//          (e1, e3, e5: K slots 5..7), half 1 the even ones (e0, e2, e4: slots 4..6) and the constant 1.
// A tile that would need a fourth site raises the error flag (the host launches this kernel only when the smallest bag
// has >= 16 reads and the job fits 32-bit indices; enc_kernel stays the path for everything else).
// =====================================================================================
#ifdef M6A_AB_STAMPS
// Diagnostic build only (tools/encoder_timeline.py; never the product): every wave of enc_site16_kernel stamps s_memtime at ten points
// of each of M6A_AB_STAMP_TILES consecutive tiles from its M6A_AB_STAMP_FIRST-th on, with its hardware slot (HW_ID, XCC_ID), so the two
// waves that share a SIMD can be laid side by side.  The stamps are scalar instructions with no wait; lane 0 stores them after the
// last stamp of a tile.
#define M6A_AB_STAMP_TILES 16
#define M6A_AB_STAMP_FIRST 100
#define M6A_AB_STAMP_POINTS 10
__device__ unsigned long long m6a_ab_stamp_buf[4096][M6A_AB_STAMP_TILES][M6A_AB_STAMP_POINTS];
__device__ unsigned m6a_ab_stamp_hw[4096][2];
extern "C" int m6a_ab_read_stamps(unsigned long long *stamps, unsigned *hw)
{
    if (hipMemcpyFromSymbol(stamps, HIP_SYMBOL(m6a_ab_stamp_buf), sizeof(m6a_ab_stamp_buf)) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(hw, HIP_SYMBOL(m6a_ab_stamp_hw), sizeof(m6a_ab_stamp_hw)) != hipSuccess) return -1;
    return 0;
}
#define M6A_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); ts[k] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define M6A_STAMP(k) do { } while (0)
#endif
__global__ __launch_bounds__(256, M6A_SITE16_WAVES_PER_SIMD) void enc_site16_kernel(EncArgs a)
{
    clamp_keeps_nan();
    m6a_clk_stamp(a.clk, 0);
    __shared__ float s_emb[132];
    __shared__ __attribute__((aligned(16))) float s_bn[M6A_BN_FLOATS];
    for (int i = threadIdx.x; i < 132; i += 256) s_emb[i] = a.emb[i];
#ifdef M6A_AB_BN_PK
    for (int i = threadIdx.x; i < M6A_BN_FLOATS; i += 256) s_bn[i] = a.bn[(i & ~3) | ((i & 1) << 1) | ((i & 2) >> 1)];   // (a, b, a', b') -> (a, a', b, b')
#else
    for (int i = threadIdx.x; i < M6A_BN_FLOATS; i += 256) s_bn[i] = a.bn[i];
#endif
#ifdef M6A_AB_W3
    __shared__ float4 s_w2q[19 * 64];
    for (int idx = threadIdx.x; idx < 19 * 64; idx += 256) {
        const int g = idx >> 6, ln = idx & 63, u = (g >> 2) * 16 + (g & 3) * 4;
        s_w2q[idx] = make_float4(a.wfrag[(40 + u) * 64 + ln], a.wfrag[(41 + u) * 64 + ln], a.wfrag[(42 + u) * 64 + ln], a.wfrag[(43 + u) * 64 + ln]);
    }
#endif
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int col = lane & 31;
    const int half = lane >> 5;
    const float *bn_half = s_bn + half * 32;
    const int wave = (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n_tiles = (int)a.n_tiles, tpw = (int)a.tiles_per_wave;
    const int tile0 = wave * tpw;
    if (tile0 >= n_tiles) return;
    const int tile1 = (tile0 + tpw < n_tiles) ? tile0 + tpw : n_tiles;
    const int n_sites = (int)a.n_sites;
    const int last_lim = (int)(a.n_reads - 1 - (int64_t)(n_tiles - 1) * 32);   // last valid column of the last tile

#ifdef M6A_AB_W3
    float w1[40], w3[16];
    float4 wq[2];
    const float4 *w2lane = s_w2q + lane;
    wq[0] = w2lane[0];
#pragma unroll
    for (int i = 0; i < 40; i++) w1[i] = a.wfrag[i * 64 + lane];
#else
    float w1[40], w2[80], w3[16];
#pragma unroll
    for (int i = 0; i < 40; i++) w1[i] = a.wfrag[i * 64 + lane];
#pragma unroll
    for (int i = 0; i < 80; i++) w2[i] = a.wfrag[(40 + i) * 64 + lane];
#endif
#pragma unroll
    for (int i = 0; i < 16; i++) w3[i] = a.wfrag[(120 + i) * 64 + lane];

    int s_base;
    {
        const int64_t r = (int64_t)tile0 * 32;
        int lo = 0, hi = n_sites;                // invariant: off[lo] <= r < off[hi]
        while (hi - lo > 1) {
            const int mid = (int)(((int64_t)lo + hi) >> 1);
            if (a.off[mid] <= r) lo = mid; else hi = mid;
        }
        s_base = lo;
    }
    const int last_site = n_sites - 1;

    const __attribute__((address_space(4))) int64_t *off_c = (const __attribute__((address_space(4))) int64_t *)a.off;
    auto link0 = [&](int base, int64_t (&o)[3]) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int si = base + 1 + i;
            o[i] = off_c[si <= n_sites ? si : n_sites];
        }
    };
    const int kq = lane < 18 ? lane : 17;                    // every lane loads a valid byte: no merge
    const int kq_site = kq / 6, kq_byte = (kq % 6) / 2;
    int too_small = 0;                                       // wave-uniform
    auto link1 = [&](int tile, int base, const int64_t (&o)[3], int &rel, int &kid, float (&x)[8]) {
        const int64_t rbase = (int64_t)tile * 32;                                 // uniform
        const int lim = tile == n_tiles - 1 ? last_lim : 31;                      // last valid column of this tile
        const int crel = col < lim ? col : lim;                                   // the lane's read, clamped into the job
        const int d0 = (int)o[0] - (int)rbase, d1 = (int)o[1] - (int)rbase, d2 = (int)o[2] - (int)rbase;
        rel = (crel >= d0 ? 1 : 0) + (crel >= d1 ? 1 : 0);
        too_small |= lim >= d2 ? 1 : 0;
        const int room = last_site - base;                                        // sites after the base site
        const int ksr = kq_site < room ? kq_site : room;
#ifdef M6A_AB_ADDR64
        kid = (int)(a.site_kmers + (int64_t)base * 3)[ksr * 3 + kq_byte];
#else
        kid = (int)(a.site_kmers + (int64_t)base * 3)[__umul24((unsigned)ksr, 3u) + (unsigned)kq_byte];      // scalar base + unsigned 32-bit lane offset (see the x loads)
#endif
        // K slot 2i + half holds feature 2i + half, as in enc_kernel
#ifdef M6A_AB_ADDR64
        const float *xp = a.X + rbase * 9 + (crel * 9 + half);
#pragma unroll
        for (int i = 0; i < 4; i++) x[i] = xp[2 * i];
        x[4] = xp[half ? 6 : 8];
#else
        // the tile's base is wave-uniform and the lane's part fits 32 unsigned bits: global_load ... v_off, s[base] offset:imm -- no 64-bit
        // address arithmetic on the VALU (a signed lane offset made hipcc form every address with v_mad_u64_u32 / v_lshl_add_u64, quarter-
        // and half-rate instructions on the datapath the MFMAs use)
        const float *xt = a.X + rbase * 9;
        const float *xp = (const float *)((const char *)xt + (unsigned)(4 * (crel * 9 + half)));       // a BYTE offset of 32 bits: what the instruction takes
#pragma unroll
        for (int i = 0; i < 4; i++) x[i] = xp[2 * i];
        x[4] = *(const float *)((const char *)xt + (unsigned)(4 * (crel * 9 + (half ? 7 : 8))));
#endif
    };
    auto link2 = [&](int kid, float &ev) { ev = s_emb[2 * kid + (lane & 1)]; };
    // lanes 0..17 hold float q % 6 of site a + q / 6; a lane of half h wants floats (1 - h), (1 - h) + 2, (1 - h) + 4 of
    // site a + rel.  (A site beyond the job's last -- `room` above -- is never selected: rel counts real boundaries.)
    auto link3 = [&](float ev, int rel, float (&x)[8]) {
        const int ebits = __builtin_bit_cast(int, ev);
        const int addr = 4 * (rel * 6 + 1 - half);
        const float e0 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, ebits));
        const float e1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr + 8, ebits));
        const float e2 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr + 16, ebits));
        x[4] = half ? e0 : x[4];
        x[5] = half ? e1 : e0;
        x[6] = half ? e2 : e1;
        x[7] = half ? 1.0f : e2;
    };

    // prologue: first tile, unpipelined
    float f[8];
    s_base = __builtin_amdgcn_readfirstlane(s_base);
    {
        int64_t o[3];
        int rel, kid;
        float ev;
        link0(s_base, o);
        link1(tile0, s_base, o, rel, kid, f);
        link2(kid, ev);
        link3(ev, rel, f);
        s_base += __builtin_amdgcn_readfirstlane(__shfl(rel, 31, 64));
    }

    phase_shift_second_workgroup();
    BnPairs bnq;
    bn_pairs_load(bnq, bn_half);
    for (int tile = tile0; tile < tile1; ++tile) {
        // the chain always runs (for the last tile it refetches that tile): no guard, no merge
        const int tn = tile + 1 < tile1 ? tile + 1 : tile;
        float fn[8], evn;
        int64_t o[3];
        int reln, kidn;
#ifdef M6A_AB_STAMPS
        unsigned long long ts[M6A_AB_STAMP_POINTS];
#endif
        M6A_STAMP(0);
        link0(s_base, o);
        tile_body_priority();

        f32x16 acc2, h1a, h1b;
#pragma unroll
        for (int q = 0; q < 16; q++) { acc2[q] = 0.0f; h1a[q] = 0.0f; }
#pragma unroll
        for (int st = 0; st < 8; st++)
            h1a = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[st], f[st], h1a, 0, 0, 0);
        M6A_STAMP(1);
#pragma unroll
        for (int m = 0; m < 5; m++) {
            f32x16 &cur = (m & 1) ? h1b : h1a;
            f32x16 &nxt = (m & 1) ? h1a : h1b;
#ifndef M6A_AB_L2_FIRST
            if (m < 4) {
#pragma unroll
                for (int q = 0; q < 16; q++) nxt[q] = 0.0f;
#pragma unroll
                for (int st = 0; st < 8; st++)
                    nxt = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[(m + 1) * 8 + st], f[st], nxt, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#endif
            // where the three links of the next tile's input chain sit: behind the layer-1 MFMAs of unit tiles 2, 3 and 4 (round 6; rounds 2-6a: 1, 2 and 4 --
            // m = 0, 1, 3 here).  Measured over two runs of interleaved legs (profiles/r06_encoder_ab_schedule.json, r06_encoder_ab_links.json): -0.5 %;
            // 0/1/4, 2/3/4, 1/3/4: +-0.2 %; 0/2/3, 0/2/4, 0/3/4: +0.4 .. +1.3 %.  -DM6A_AB_LINK1_AT=.. builds the others.
#ifndef M6A_AB_LINK1_AT
#define M6A_AB_LINK1_AT 1
#define M6A_AB_LINK2_AT 2
#define M6A_AB_LINK3_AT 3
#endif
#ifdef M6A_AB_NO_LINKS                   // knock-out build only (WRONG results): no input chain -- every tile computes on the first tile's features
            if (m == 0) {
#pragma unroll
                for (int i = 0; i < 8; i++) fn[i] = f[i];
                reln = 0; kidn = 0; evn = 0.0f;
            }
#else
            if (m == M6A_AB_LINK1_AT) link1(tn, s_base, o, reln, kidn, fn);
            if (m == M6A_AB_LINK2_AT) link2(kidn, evn);
            if (m == M6A_AB_LINK3_AT) link3(evn, reln, fn);
#endif
#ifdef M6A_AB_W3
            if (m == 0) layer2_with_bn_lds<0>(acc2, cur, wq, w2lane, bnq, bn_half);
            if (m == 1) layer2_with_bn_lds<1>(acc2, cur, wq, w2lane, bnq, bn_half);
            if (m == 2) layer2_with_bn_lds<2>(acc2, cur, wq, w2lane, bnq, bn_half);
            if (m == 3) layer2_with_bn_lds<3>(acc2, cur, wq, w2lane, bnq, bn_half);
            if (m == 4) layer2_with_bn_lds<4>(acc2, cur, wq, w2lane, bnq, bn_half);
#else
            if (m == 0) layer2_with_bn<0>(acc2, cur, w2, bnq, bn_half);
            if (m == 1) layer2_with_bn<1>(acc2, cur, w2, bnq, bn_half);
            if (m == 2) layer2_with_bn<2>(acc2, cur, w2, bnq, bn_half);
            if (m == 3) layer2_with_bn<3>(acc2, cur, w2, bnq, bn_half);
            if (m == 4) layer2_with_bn<4>(acc2, cur, w2, bnq, bn_half);
#endif
#ifdef M6A_AB_L2_FIRST
            if (m < 4) {
#pragma unroll
                for (int q = 0; q < 16; q++) nxt[q] = 0.0f;
#pragma unroll
                for (int st = 0; st < 8; st++)
                    nxt = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[(m + 1) * 8 + st], f[st], nxt, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#endif
            M6A_STAMP(2 + m);
        }
        tile_epilogue_priority();
#ifdef M6A_AB_NO_EPILOGUE
        // knock-out build only (tools/encoder_ab.py): what the 32 -> 1 layer + sigmoid cost in place -- an upper bound on what
        // moving them under the next tile's MFMAs could buy.  WRONG results by construction; never the product.
        const float p = (acc2[0] + acc2[5]) + w3[0];
#else
        const float z = gemv32_as_mkl(acc2, w3, half) + a.b3;
        M6A_STAMP(7);
        const float p = 1.0f / (1.0f + sleef_expf_u10(-z));
#endif
        if (half == 0 && col <= (tile == n_tiles - 1 ? last_lim : 31)) (a.read_prob + (int64_t)tile * 32)[col] = p;
        M6A_STAMP(8);
        if (tn != tile) s_base += __builtin_amdgcn_readfirstlane(__shfl(reln, 31, 64));
#pragma unroll
        for (int i = 0; i < 8; i++) f[i] = fn[i];
#ifdef M6A_AB_STAMPS
        M6A_STAMP(9);
        {
            const int k = tile - tile0 - M6A_AB_STAMP_FIRST;
            if (k >= 0 && k < M6A_AB_STAMP_TILES && wave < 4096 && lane == 0) {
#pragma unroll
                for (int i = 0; i < M6A_AB_STAMP_POINTS; i++) m6a_ab_stamp_buf[wave][k][i] = ts[i];
                if (k == 0) {
                    m6a_ab_stamp_hw[wave][0] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));       // HW_REG_HW_ID
                    m6a_ab_stamp_hw[wave][1] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));      // HW_REG_XCC_ID
                }
            }
        }
#endif
    }
    if (too_small && lane == 0) atomicExch(a.err, 2);
    m6a_clk_stamp(a.clk, 1);
}

// =====================================================================================
// Read encoder, 12-slot variant (calls whose bags all have >= 16 reads).
//
// Six of the 16 inputs (the three k-mer embeddings) and the bias are constant per SITE, and a
// 32-read tile that starts in site `a` ends by site a+2 when bags have >= 16 reads.  Layer 1 is
// therefore run with K = 12 slots instead of 16:
//     slots 0..8   the nine signal features, against W1[:, 0..8];
//     slots 9..11  one-hot "read belongs to site a / a+1 / a+2", against the per-site vectors
//                  c_s[u] = b1[u] + sum_e W1[u][9+e] * emb_e(s)            (the same terms),
// i.e. 6 K-steps per unit tile instead of 8: 106 MFMAs per tile instead of 116.  It is the one place where this
// kernel's sums leave the reference's order: the reference adds x0..x8, then the six embedding terms, then b1, each an
// fma on the running sum; here the embedding terms and b1 are summed per site first and enter with ONE addition after x8.
// Everything after that (batch norm, layer 2) is the reference's order again.  On all 20 M reads of configs[2] x 4
// checkpoints the worst use of the reference's rtol 1e-5 bar is 0.72 here, 0.53 in enc_kernel (rounds 1-3: 1.06 / 1.05).  The c vectors
// (A operands of the indicator steps) are rebuilt for every tile on the VALU, in the shadow of the
// matrix pipe: the 18 embedding floats of the three sites are fetched by lanes 0..17, pulled into
// (x, y) pairs over the LDS crossbar with ds_bpermute (not a VALU instruction), and folded against
// W1[:, 9..14] (kept in LDS, one row per lane) with 30 v_pk_fma_f32 -- two site vectors per FMA.
// Lane halves: h=0 supplies slots x0,x2,x4,x6,x8,I(a+1); h=1 supplies x1,x3,x5,x7,I(a),I(a+2).
// Everything else (layer 2, ReLU batching, ping-pong, epilogue, input prefetch chain) is as in
// enc_kernel.  A tile that would need a fourth site raises the error flag (the host only
// launches this kernel when the smallest bag has >= 16 reads).
// =====================================================================================
__global__ __launch_bounds__(256, 2) void enc_csite_kernel(EncArgs a)
{
    clamp_keeps_nan();
    m6a_clk_stamp(a.clk, 0);
    __shared__ float s_emb[132];
    __shared__ float s_w1e[35 * 32];             // [m*7 + e][col]: W1[u][9+e] (e<6), b1[u] (e=6), u = the unit of row col of tile m
    for (int i = threadIdx.x; i < 132; i += 256) s_emb[i] = a.emb[i];
    for (int i = threadIdx.x; i < 35 * 32; i += 256) s_w1e[i] = a.w1e_tab[i];
    __shared__ __attribute__((aligned(16))) float s_bn[M6A_BN_FLOATS];
#ifdef M6A_AB_BN_PK
    for (int i = threadIdx.x; i < M6A_BN_FLOATS; i += 256) s_bn[i] = a.bn[(i & ~3) | ((i & 1) << 1) | ((i & 2) >> 1)];   // (a, b, a', b') -> (a, a', b, b')
#else
    for (int i = threadIdx.x; i < M6A_BN_FLOATS; i += 256) s_bn[i] = a.bn[i];
#endif
    // W3 in the order the layer-2 accumulator holds the 32 units: [half][q] (16 VGPRs the tile loop needs more)
    __shared__ float s_w3[32];
    if (threadIdx.x < 64) s_w3[(threadIdx.x >> 5) * 16 + (threadIdx.x & 15)] = a.wfrag[(120 + (threadIdx.x & 15)) * 64 + (threadIdx.x & 32)];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int col = lane & 31;
    const int half = lane >> 5;
    const float *bn_half = s_bn + half * 32;
    // Tile and site indices are 32-bit here (the host sends jobs beyond 2^31 sites or tiles to enc_kernel)
    // and wave-uniform: the SALU has 32-bit ordered compares but no 64-bit ones, so 64-bit indices would put
    // every clamp and loop test on the VALU -- at half rate, on the datapath the MFMAs need.
    const int wave = (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n_tiles = (int)a.n_tiles, tpw = (int)a.tiles_per_wave;
    const int tile0 = wave * tpw;
    if (tile0 >= n_tiles) return;
    const int tile1 = (tile0 + tpw < n_tiles) ? tile0 + tpw : n_tiles;
    const int n_sites = (int)a.n_sites;
    const int last_lim = (int)(a.n_reads - 1 - (int64_t)(n_tiles - 1) * 32);   // last valid column of the last tile

    // static weight fragments: w1x[m*4+st] = W1[u][2st+half], w8[m] = W1[u][8], u = the unit of row col of tile m
    float w1x[20], w8[5], w2[80];
#pragma unroll
    for (int i = 0; i < 20; i++) w1x[i] = a.wfrag2[i * 64 + lane];
#pragma unroll
    for (int i = 0; i < 5; i++) w8[i] = a.wfrag2[(20 + i) * 64 + lane];
#pragma unroll
    for (int i = 0; i < 80; i++) w2[i] = a.wfrag[(40 + i) * 64 + lane];

    int s_base;
    {
        const int64_t r = (int64_t)tile0 * 32;
        int lo = 0, hi = n_sites;                // invariant: off[lo] <= r < off[hi]
        while (hi - lo > 1) {
            const int mid = (int)(((int64_t)lo + hi) >> 1);
            if (a.off[mid] <= r) lo = mid; else hi = mid;
        }
        s_base = lo;
    }
    const int last_site = n_sites - 1;

    // ---- input chain of one tile ----------------------------------------------------------------
    // link0: the three CSR offsets after the base site -- scalar loads (base is uniform)
    // (through the constant address space: the kernel also stores to global memory, and without it the compiler issues
    // these uniform loads as vector loads -- three more VMEM instructions in the MFMA stream per tile)
    const __attribute__((address_space(4))) int64_t *off_c = (const __attribute__((address_space(4))) int64_t *)a.off;
    auto link0 = [&](int base, int64_t (&o)[3]) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int si = base + 1 + i;
            o[i] = off_c[si <= n_sites ? si : n_sites];
        }
    };
    // link1: the lane's site relative to base; lanes 0..17 fetch the k-mer id their embedding
    // float belongs to (float q of site base + q/6 is E[kmer (q%6)/2][q&1]); THEN the x loads
    // (4 per lane, + x8 on half 0) -- vector loads retire in order, the k-mer ids must not queue
    // behind the x stream
    const int kq = lane < 18 ? lane : 17;                    // every lane loads a valid byte: no merge
    const int kq_site = kq / 6, kq_byte = (kq % 6) / 2;
    int too_small = 0;                                       // wave-uniform
    auto link1 = [&](int tile, int base, const int64_t (&o)[3], int &rel, int &kid, float (&x)[5]) {
        const int64_t rbase = (int64_t)tile * 32;                                 // uniform
        const int lim = tile == n_tiles - 1 ? last_lim : 31;                      // last valid column of this tile
        const int crel = col < lim ? col : lim;                                   // the lane's read, clamped into the job
        // site boundaries relative to the tile: o[i] > rbase (base holds the tile's first read) and a bag is
        // far smaller than 2^31 reads, so the low words are enough
        const int d0 = (int)o[0] - (int)rbase, d1 = (int)o[1] - (int)rbase, d2 = (int)o[2] - (int)rbase;
        rel = (crel >= d0 ? 1 : 0) + (crel >= d1 ? 1 : 0);
        // a lane beyond the third site boundary would need a 4th site (precondition broken): the tile's last column is
        // the largest crel, so the test is scalar; the flag is raised once, after the tile loop
        too_small |= lim >= d2 ? 1 : 0;
        const int room = last_site - base;                                        // sites after the base site
        const int ksr = kq_site < room ? kq_site : room;
#ifdef M6A_AB_ADDR64
        kid = (int)(a.site_kmers + (int64_t)base * 3)[ksr * 3 + kq_byte];
#else
        kid = (int)(a.site_kmers + (int64_t)base * 3)[__umul24((unsigned)ksr, 3u) + (unsigned)kq_byte];      // scalar base + unsigned 32-bit lane offset (see the x loads)
#endif
        // K slot 2st+half holds feature 2st+half: the features enter the sum in the order the reference's dot product
        // has them.  (One 16-byte load per lane -- half 0 x0..x3, half 1 x4..x7, weights permuted to match -- is the
        // same speed and moves the summation order away from the reference's: the worst use of the rtol 1e-5 bar over
        // 10 M reads rose from 0.81 to 0.89 on the arabidopsis weights, 0.936 to 0.951 on HEK293T.)
#ifdef M6A_AB_ADDR64
        const float *xp = a.X + rbase * 9 + (crel * 9 + half);
#pragma unroll
        for (int i = 0; i < 4; i++) x[i] = xp[2 * i];
        x[4] = xp[half ? 6 : 8];                             // x8 on half 0 (half 1: a valid dummy)
#else
        const float *xt = a.X + rbase * 9;                   // scalar base + unsigned 32-bit lane offset, as in enc_site16_kernel
        const float *xp = (const float *)((const char *)xt + (unsigned)(4 * (crel * 9 + half)));
#pragma unroll
        for (int i = 0; i < 4; i++) x[i] = xp[2 * i];
        x[4] = *(const float *)((const char *)xt + (unsigned)(4 * (crel * 9 + (half ? 7 : 8))));      // x8 on half 0 (half 1: a valid dummy)
#endif
    };
    // link2: the embedding float itself
    auto link2 = [&](int kid, float &ev) { ev = s_emb[2 * kid + (lane & 1)]; };
    // link3: c vectors -> A operands of the two indicator steps, indicator B operands
    // (x8 is the value loaded for slot 4 on half 0; in place: a4/a5/x[4..5] of the running tile are
    // dead once layer 1 of its last unit tile has issued)
    auto link3 = [&](float ev, int rel, float x8, float (&x)[6], float (&a4)[5], float (&a5)[5]) {
        // lanes 0..17 hold the 18 embedding floats of sites a, a+1, a+2.  A lane needs two c vectors: that
        // of site a (its step-4 operand when half = 1) and that of site a+1 (half 0) / a+2 (half 1) for step
        // 5.  The floats come over the LDS crossbar (ds_bpermute: not a VALU instruction, and the VALU
        // shares its datapath with the f32 MFMAs) straight into (x, y) pairs, so both vectors advance with
        // one v_pk_fma_f32 per term -- 30 per tile, no register shuffling.
        const int ebits = __builtin_bit_cast(int, ev);
        const int lane_b = half ? 48 : 24;                                    // byte address of lane 12 / 6
        int lane_a;                                                           // 0, but opaque: a constant address
        asm("v_mov_b32 %0, 0" : "=v"(lane_a));                                // would become v_readlane + v_mov
        f32x2 e2[6];
#pragma unroll
        for (int q = 0; q < 6; q++) {
            e2[q].x = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(lane_a + 4 * q, ebits));
            e2[q].y = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(lane_b + 4 * q, ebits));
        }
#pragma unroll
        for (int m = 0; m < 5; m++) {
            const float *wr = s_w1e + (m * 7) * 32 + col;
#ifdef M6A_AB_CSITE_SCALAR_FMA
            // A/B build only (tools/encoder_ab.py): the same 60 fmas as 60 plain v_fma_f32 -- the MI355X guide lists
            // packed-f32 VALU beside MFMAs as an anti-lever.  asm: plain C would be SLP-packed back into v_pk_fma_f32.  Same bits.
            float cx = wr[6 * 32], cy = cx;
#pragma unroll
            for (int q = 0; q < 6; q++) {
                const float wq = wr[q * 32];
                asm("v_fma_f32 %0, %1, %2, %0" : "+v"(cx) : "v"(wq), "v"(e2[q].x));
                asm("v_fma_f32 %0, %1, %2, %0" : "+v"(cy) : "v"(wq), "v"(e2[q].y));
            }
            const f32x2 c = {cx, cy};
#else
            f32x2 c = {wr[6 * 32], wr[6 * 32]};
#pragma unroll
            for (int q = 0; q < 6; q++) {
                const float wq = wr[q * 32];
                const f32x2 w2v = {wq, wq};
                c = __builtin_elementwise_fma(w2v, e2[q], c);
            }
#endif
            a4[m] = half ? c.x : w8[m];          // step 4: h=0 x8, h=1 I(a)
            a5[m] = c.y;                         // step 5: h=0 I(a+1), h=1 I(a+2)
#ifdef M6A_AB_CSITE_PIN
            // A/B build only: hipcc sinks this loop's arithmetic (pure, used by the NEXT tile) behind the epilogue, where no MFMA
            // covers it; the empty volatile asm keeps it where it is written, between the MFMA groups.  Same bits.
            asm volatile("" : "+v"(a4[m]), "+v"(a5[m]));
#endif
        }
        x[4] = half ? (rel == 0 ? 1.0f : 0.0f) : x8;
        x[5] = (rel == (half ? 2 : 1)) ? 1.0f : 0.0f;
    };

    // prologue: first tile, unpipelined
    float f[6], a4[5], a5[5];
    s_base = __builtin_amdgcn_readfirstlane(s_base);
    {
        int64_t o[3];
        int rel, kid;
        float ev, x5[5];
        link0(s_base, o);
        link1(tile0, s_base, o, rel, kid, x5);
#pragma unroll
        for (int i = 0; i < 4; i++) f[i] = x5[i];
        link2(kid, ev);
        link3(ev, rel, x5[4], f, a4, a5);
        s_base += __builtin_amdgcn_readfirstlane(__shfl(rel, 31, 64));
    }

    BnPairs bnq;
    bn_pairs_load(bnq, bn_half);
    for (int tile = tile0; tile < tile1; ++tile) {
        // the chain always runs (for the last tile it refetches that tile): no guard, no merge
        const int tn = tile + 1 < tile1 ? tile + 1 : tile;
        float fn[5], evn;
        int64_t o[3];
        int reln, kidn;
        link0(s_base, o);
        tile_body_priority();

        auto layer1 = [&](int m, f32x16 &acc) {
#pragma unroll
            for (int q = 0; q < 16; q++) acc[q] = 0.0f;
#pragma unroll
            for (int st = 0; st < 4; st++)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1x[m * 4 + st], f[st], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[m], f[4], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a5[m], f[5], acc, 0, 0, 0);
        };

        f32x16 acc2, h1a, h1b;
#pragma unroll
        for (int q = 0; q < 16; q++) acc2[q] = 0.0f;
        layer1(0, h1a);
#pragma unroll
        for (int m = 0; m < 5; m++) {
            f32x16 &cur = (m & 1) ? h1b : h1a;
            f32x16 &nxt = (m & 1) ? h1a : h1b;
            if (m < 4) layer1(m + 1, nxt);
            __builtin_amdgcn_sched_barrier(0);
            if (m == 0) link1(tn, s_base, o, reln, kidn, fn);
            if (m == 1) link2(kidn, evn);
            if (m == 3) link3(evn, reln, fn[4], f, a4, a5);             // layer1(4) has issued: in place
            if (m == 0) layer2_with_bn<0>(acc2, cur, w2, bnq, bn_half);
            if (m == 1) layer2_with_bn<1>(acc2, cur, w2, bnq, bn_half);
            if (m == 2) layer2_with_bn<2>(acc2, cur, w2, bnq, bn_half);
            if (m == 3) layer2_with_bn<3>(acc2, cur, w2, bnq, bn_half);
            if (m == 4) layer2_with_bn<4>(acc2, cur, w2, bnq, bn_half);
        }
        tile_epilogue_priority();
        float z = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; q++) z = fmaf(relu2(acc2[q]), s_w3[half * 16 + q], z);
        z += __shfl_xor(z, 32, 64);
        z += a.b3;
        const float p = 1.0f / (1.0f + expf(-z));
        if (half == 0 && col <= (tile == n_tiles - 1 ? last_lim : 31)) (a.read_prob + (int64_t)tile * 32)[col] = p;
        if (tn != tile) s_base += __builtin_amdgcn_readfirstlane(__shfl(reln, 31, 64));
#pragma unroll
        for (int i = 0; i < 4; i++) f[i] = fn[i];
    }
    if (too_small && lane == 0) atomicExch(a.err, 2);
    m6a_clk_stamp(a.clk, 1);
}

// =====================================================================================
// Site pooling, general bags -- exact replay of
//   proba = np.random.choice(proba, n_iters*n_samples, replace=True).reshape(n_iters, n_samples)
//   (1 - np.prod(1 - proba, axis=1)).mean()               (inference_utils.py:85-86)
// for every site of every flush group, plus mod_ratio = mean(p >= thr) (inference_utils.py:53).
// RandomState.choice -> legacy randint(0,n): rng = n-1, mask = 2^ceil(log2(n))-1,
// `do v = next32() & mask; while (v > rng)`; the Pool worker of each flush group starts from
// the seeded, never-advanced state (inference_utils.py:102-104 after inference.py:86), so all
// groups read the SAME word stream `raw` (generated once on the host) from position 0, and the
// sites of a group consume it back to back.
//
// scan_site() replays one site from a given stream position.  Per step 64 consecutive words are
// masked and range-tested; accepted lanes get their rank by ballot + mbcnt, gather 1-p from the
// LDS bag and store it at buffer slot cnt+rank.  The buffer is ONE window of 32 iterations
// (CH = 32*K slots) plus the <= 255 draws that arrive before the per-block window check; a slot
// lives at dword slot + slot/4, which makes the stride between iterations K*5/4 (25 for K = 20:
// odd, so the per-iteration read-back is bank-conflict-free, and with K % 4 == 0 every read offset
// is a compile-time immediate) at no division.  When the window is full, lanes 0..31 multiply their
// K values left to right (float32, the order of np.prod), add 1-prod, and the overhang is moved
// down to slot 0.  The step that completes the site finds the lane holding the last accepted draw;
// the next site starts at the following word, exactly like the sequential NumPy loop.
//
// Where a site starts depends on how many words every earlier site of its group rejected.  Two
// drivers, chosen on the host by how much parallelism the call has:
//   pool_scan_group_kernel  one wavefront per flush group, sites in sequence (no extra pass);
//   pool_scan_start_kernel + pool_scan_site_kernel
//                           a counting-only pass per group (per-lane counters, 1024 words per
//                           step) records start_pos[site]; then one wavefront per SITE.
// =====================================================================================
__device__ __forceinline__ uint32_t pow2_mask(uint32_t rng)
{
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    return mask;
}

// dword address of ring slot `s` (see above)
__device__ __forceinline__ int ring_addr(int s) { return s + (s >> 2); }

template <int KT>
__device__ __forceinline__ float window_product(const float *ring, int wbase, int lane, int K)
{
    float prod = 1.0f;
    if (KT && (KT % 4 == 0)) {
        const float *row = ring + (wbase + lane * KT) * 5 / 4;      // wbase, lane*KT multiples of 4
#pragma unroll
        for (int k = 0; k < KT; k++) prod *= row[k + (k >> 2)];
    } else {
        const int base = wbase + lane * K;
        for (int k = 0; k < K; k++) prod *= ring[ring_addr(base + k)];
    }
    return prod;
}

// Replays one site.  `pos` = its first stream word on entry, the word after its last accepted
// draw on return.  Returns false if the stream proved too short (err flag set).
template <int KT>
__device__ __forceinline__ bool scan_site(const PoolArgs &a, int64_t s, uint32_t &pos, float *bag, float *ring,
                                          int lane, int K)
{
    float *stack = ring + (32 * K + 256) * 5 / 4 + 32;   // merge stack of the pairwise sum (lane 0)
    // the stream position is wave-uniform (every lane read the same start): say so, and the block loop, its
    // bounds checks and the stream addresses run on the SALU instead of as an exec-masked 64-bit VALU loop
    pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);
    const int bag_cap = a.bag_cap;
    const int CH = 32 * K;                       // slots per window
    const int A = a.T * K;                       // accepted draws per site
    // site-level values are wave-uniform: say so, and rng/mask/branches below become scalar
    const int64_t r0 = ((int64_t)__builtin_amdgcn_readfirstlane((int)(a.off[s] >> 32)) << 32) |
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)a.off[s]);
    const int n = __builtin_amdgcn_readfirstlane((int)(a.off[s + 1] - r0));
    int cge = 0;
    for (int i = lane; i < n; i += 64) {
        const float v = a.read_prob[r0 + i];
        cge += (v >= a.thr) ? 1 : 0;
        if (i < bag_cap) bag[i] = 1.0f - v;
    }
    cge = wave_sum_i32(cge);
    if (lane == 0) a.mod_ratio[s] = n > 0 ? (double)cge / (double)n : __builtin_nan("");
    if (n <= 0) { if (lane == 0) a.site_prob[s] = __builtin_nanf(""); return true; }
    wave_lds_fence();

    const uint32_t rng = (uint32_t)(n - 1);
    if (rng == 0) {                              // bag of one read: randint draws no words
        float prod = 1.0f;
        const float a0 = bag[0];
        for (int k = 0; k < K; k++) prod *= a0;
        // T equal values still go through the pairwise sum (its roundings are part of the result)
        const float x = 1.0f - prod;
        int sp = 0;
        for (int b = 0; b < a.n_leaves; b++) {
            const int len = a.leaf_start[b + 1] - a.leaf_start[b];
            float r = 0.0f;
            for (int i = 0; i < (len >> 3); i++) r += x;
            r = ((r + r) + (r + r)) + ((r + r) + (r + r));
            for (int i = 0; i < (len & 7); i++) r += x;
            if (lane == 0) {
                for (int m = a.merge_after[b]; m > 0; --m) r = stack[--sp] + r;
                stack[sp++] = r;
            }
        }
        if (lane == 0) a.site_prob[s] = stack[0] / (float)a.T;
        wave_lds_fence();
        return true;
    }
    const uint32_t mask = pow2_mask(rng);
    const bool in_lds = n <= bag_cap;            // wave-uniform: no per-lane test on the fast path
    int acc = 0, cnt = 0;
    // mean over iterations = NumPy's pairwise sum (see MeanPlan in m6a_api.hip).  A window hands over
    // 32 consecutive iterations in lanes 0..31; leaf starts are multiples of 8, so each group of 8 lanes is
    // one round of the current leaf's 8 accumulator chains: lanes 0..7 carry the chains (`chain`), groups are
    // shifted down to them in iteration order (DPP row_shl:8 within a 16-lane row, one ds_bpermute across
    // rows).  A finished leaf combines its chains (chain8_sum = NumPy's order), adds the n % 8 tail (last
    // leaf only) and is pushed / merged as the tree says; every lane runs the same (garbage outside 0..7).
    int it_base = 0, leaf = 0, sp = 0;
    int leaf_end = a.leaf_start[1];
    float chain = 0.0f;
    // the merge stack stays in LDS (lane 0): a register stack with wave-uniform height costs ~20 register
    // copies per pass of the main loop below (loop-carried select chains), far more than the two LDS
    // accesses per finished leaf it would save
    auto leaf_done = [&](float r) {
        // plan reads are scalar loads (uniform index): a vector load here would wait on vmcnt behind the
        // stream prefetch
        const uint32_t mw = ((const uint32_t *)a.merge_after)[leaf >> 2];
        const int merges = (int)((mw >> (8 * (leaf & 3))) & 0xffu);
        if (lane == 0) {
            int p = sp;
            for (int m = merges; m > 0; --m) r = stack[--p] + r;
            stack[p] = r;
        }
        sp = __builtin_amdgcn_readfirstlane(sp + 1 - merges);
        leaf = __builtin_amdgcn_readfirstlane(leaf + 1);
        leaf_end = leaf < a.n_leaves ? a.leaf_start[leaf + 1] : 0x7fffffff;
        chain = 0.0f;
    };
    // v: lanes 0..nvalid-1 hold iterations it_base .. it_base+nvalid-1 (nvalid = 32 except at the site's end)
    auto feed = [&](float v, int nvalid) {
        // lanes 16..31 -> 0..15.  (v_permlane16_swap would do this without the LDS crossbar, but measured
        // no faster here.)
        const float v16 = __shfl(v, (lane + 16) & 63, 64);
        const float g1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x108, 0xf, 0xf, true));
        const float g3 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v16), 0x108, 0xf, 0xf, true));
        if (nvalid == 32 && leaf_end >= it_base + 32) {          // no leaf boundary inside: four rounds
            chain = (((chain + v) + g1) + v16) + g3;
            if (leaf_end == it_base + 32) leaf_done(chain8_sum(chain));
            it_base = __builtin_amdgcn_readfirstlane(it_base + 32);
            return;
        }
#pragma unroll 1
        for (int j = 0; 8 * j < nvalid; ++j) {
            if (it_base + 8 * j == leaf_end) leaf_done(chain8_sum(chain));
            const float g = j == 0 ? v : j == 1 ? g1 : j == 2 ? v16 : g3;
            if (8 * j + 8 <= nvalid) {
                chain += g;
            } else {                             // the last leaf's tail: one by one after the chains combine
                float r = chain8_sum(chain);
                const int gbits = __builtin_bit_cast(int, g);
                for (int i = 0; i < nvalid - 8 * j; i++) r += __builtin_bit_cast(float, __builtin_amdgcn_readlane(gbits, i));
                leaf_done(r);
            }
        }
        it_base = __builtin_amdgcn_readfirstlane(it_base + nvalid);
    };
    if ((uint64_t)pos + 512 > (uint64_t)a.raw_len) { if (lane == 0) atomicExch(a.err, 1); return false; }
    uint32_t w[4], wn[4];
#pragma unroll
    for (int i = 0; i < 4; i++) w[i] = (a.raw + pos)[64 * i + lane];            // scalar base + lane offset

    // one accepted word -> window buffer: gather 1-p, store at slot cnt+rank
    auto put = [&](uint32_t v, int slot) {
        float val;
        if (in_lds) val = bag[v];
        else val = v < (uint32_t)bag_cap ? bag[v] : 1.0f - a.read_prob[r0 + v];
        ring[ring_addr(slot)] = val;
    };
    // The buffer is ONE window (CH slots) plus room for the draws that arrive before the window check
    // (<= 255 per 256-word block).  Closing the window multiplies it out and moves that overhang down to slot
    // 0 -- CH is a multiple of 4, so the overhang is one contiguous run of dwords: five reads and five writes
    // per lane, instead of a wrap-around test on every stored draw.
    auto close_window = [&]() {
        wave_lds_fence();
        float v = 0.0f;
        if (lane < 32) v = 1.0f - window_product<KT>(ring, 0, lane, K);
        const int ov_dw = ring_addr(cnt - CH);   // dwords of the overhang (<= 319)
        const float *src = ring + ring_addr(CH);
        float mv[5];
#pragma unroll
        for (int j = 0; j < 5; j++) mv[j] = src[lane + 64 * j];              // inside the buffer's slack
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < 5; j++)
            if (lane + 64 * j < ov_dw) ring[lane + 64 * j] = mv[j];
        cnt -= CH;
        wave_lds_fence();
        feed(v, 32);
    };

    // ---- whole 256-word blocks, branch-free inside: every lane gathers (a masked word is a valid bag index),
    // rejected lanes store to a trash slot (one v_cndmask instead of a branch), the slot is cnt + rank with cnt
    // folded into the mbcnt accumulator, all four gathers are in flight together, one window check per block
    // (needs CH >= 256, i.e. K >= 8).  Bags beyond the LDS bag take the tail path.
    const int trash = ring_addr(CH + 256) + 1;
    if (K >= 8 && in_lds) {
        for (;;) {
            if ((uint64_t)pos + 512 > (uint64_t)a.raw_len) { if (lane == 0) atomicExch(a.err, 1); return false; }
            uint32_t v[4];
            unsigned long long bal[4];
            int c[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                v[i] = w[i] & mask;
                bal[i] = __ballot(v[i] <= rng);
                c[i] = __popcll(bal[i]);
            }
            const int cb = c[0] + c[1] + c[2] + c[3];
            if (acc + cb >= A) break;            // the site's last draw is in this block: exact tail below
#pragma unroll
            for (int i = 0; i < 4; i++) wn[i] = (a.raw + pos)[256 + 64 * i + lane];   // next block in flight
            int base = cnt;
            float val[4];
            int addr[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                val[i] = bag[v[i]];              // v <= mask < bag_cap (a power of two >= the largest bag): in bounds unclamped
                const int slot = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal[i] >> 32),
                                 __builtin_amdgcn_mbcnt_lo((uint32_t)bal[i], (uint32_t)base));
                addr[i] = v[i] <= rng ? ring_addr(slot) : trash;
                base += c[i];
            }
#pragma unroll
            for (int i = 0; i < 4; i++) ring[addr[i]] = val[i];
            cnt += cb;
            acc += cb;
            if (cnt >= CH) close_window();
            pos += 256;
#pragma unroll
            for (int i = 0; i < 4; i++) w[i] = wn[i];
        }
    }
    // ---- exact tail (and the whole site when K < 8): 64 words at a time until the A-th draw
    bool done = false;
    while (!done) {
        if ((uint64_t)pos + 512 > (uint64_t)a.raw_len) { if (lane == 0) atomicExch(a.err, 1); return false; }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (done) break;
            const uint32_t v = w[i] & mask;
            bool ok = v <= rng;
            const unsigned long long bal = __ballot(ok);
            int c = __popcll(bal);
            const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32),
                             __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
            const int remaining = A - acc;
            if (c >= remaining) {                // this step completes the site
                ok = ok && rank < remaining;
                const unsigned long long lastb = __ballot(ok && rank == remaining - 1);
                pos += 64 * i + (uint32_t)__builtin_ctzll(lastb) + 1;
                c = remaining;
                done = true;
            }
            if (ok) put(v, cnt + rank);
            cnt += c;
            acc += c;
            while (cnt >= CH) close_window();  // K = 1: a 64-word step can fill two windows
        }
        if (!done) {
            pos += 256;
#pragma unroll
            for (int i = 0; i < 4; i++) w[i] = (a.raw + pos)[64 * i + lane];            // scalar base + lane offset
        }
    }
    wave_lds_fence();
    const int t_rem = cnt / K;                   // iterations left in the current window (< 32)
    {
        float v = 0.0f;
        if (lane < t_rem) v = 1.0f - window_product<KT>(ring, 0, lane, K);
        feed(v, t_rem);                          // it_base == T now
        if (leaf < a.n_leaves) leaf_done(chain8_sum(chain));
    }
    if (lane == 0) a.site_prob[s] = stack[0] / (float)a.T;
    wave_lds_fence();
    return true;
}

// LDS per wavefront: the bag, the window buffer (32*K slots + 256 of overhang at 5/4 dwords per slot, + trash)
// and the pairwise-sum stack
__device__ __forceinline__ int scan_wave_floats(int bag_cap, int K) { return bag_cap + (32 * K + 256) * 5 / 4 + 32 + M6A_MEAN_STACK; }

template <int KT>
__global__ __launch_bounds__(256) void pool_scan_group_kernel(PoolArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int K = KT ? KT : a.K;
    const int lane = threadIdx.x & 63, wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float *bag = smem + wib * scan_wave_floats(a.bag_cap, K);
    float *ring = bag + a.bag_cap;
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    for (int64_t g = (int64_t)blockIdx.x * 4 + wib; g < a.n_groups; g += n_waves) {
        uint32_t pos = 0;
        const int64_t s_end = a.goff[g + 1];
        for (int64_t s = a.goff[g]; s < s_end; ++s)
            if (!scan_site<KT>(a, s, pos, bag, ring, lane, K)) return;
    }
}

template <int KT>
__global__ __launch_bounds__(256) void pool_scan_site_kernel(PoolArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int K = KT ? KT : a.K;
    const int lane = threadIdx.x & 63, wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float *bag = smem + wib * scan_wave_floats(a.bag_cap, K);
    float *ring = bag + a.bag_cap;
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    for (int64_t s = (int64_t)blockIdx.x * 4 + wib; s < a.n_sites; s += n_waves) {
        uint32_t pos = a.start_pos[s];
        if (!scan_site<KT>(a, s, pos, bag, ring, lane, K)) return;
    }
}

template __global__ void pool_scan_group_kernel<20>(PoolArgs);
template __global__ void pool_scan_group_kernel<0>(PoolArgs);
template __global__ void pool_scan_site_kernel<20>(PoolArgs);
template __global__ void pool_scan_site_kernel<0>(PoolArgs);

// Counting-only pass: start_pos[s] for every site.  Per step 16 x 64 words; every lane keeps its
// own acceptance count (and / compare / add), nothing is reduced across lanes until A words have
// been scanned (a site needs at least A words), after that one wave total per step, and only the
// step in which the site's last draw falls is re-walked with ballots to get the exact word.
#define M6A_SCAN_A_LOADS 16
__global__ __launch_bounds__(256) void pool_scan_start_kernel(PoolArgs a)
{
    const int lane = threadIdx.x & 63;
    const int A = a.T * a.K;
    const uint32_t STEP = 64 * M6A_SCAN_A_LOADS;
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    // group, site, position and the running totals are wave-uniform: kept scalar so the loops and bounds checks
    // run on the SALU
    for (int64_t g = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); g < a.n_groups; g += n_waves) {
        uint32_t pos = 0;
        const int64_t s_end = a.goff[g + 1];
        for (int64_t s = a.goff[g]; s < s_end; ++s) {
            if (lane == 0) a.start_pos[s] = pos;
            const int64_t n = a.off[s + 1] - a.off[s];
            if (n <= 1) continue;                        // randint(0,1) draws no words
            const uint32_t rng = (uint32_t)(n - 1);
            const uint32_t mask = pow2_mask(rng);
            if ((uint64_t)pos + 2 * STEP > (uint64_t)a.raw_len) { if (lane == 0) atomicExch(a.err, 1); return; }
            uint32_t w[M6A_SCAN_A_LOADS], wn[M6A_SCAN_A_LOADS];
#pragma unroll
            for (int i = 0; i < M6A_SCAN_A_LOADS; i++) w[i] = (a.raw + pos)[64 * i + lane];
            int acc = 0;             // wave total of accepted words in steps already folded in
            int carry = 0;           // per-lane count of steps not folded in yet
            uint32_t scanned = 0;
            for (;;) {
                if ((uint64_t)pos + 2 * STEP > (uint64_t)a.raw_len) { if (lane == 0) atomicExch(a.err, 1); return; }
#pragma unroll
                for (int i = 0; i < M6A_SCAN_A_LOADS; i++) wn[i] = (a.raw + pos)[STEP + 64 * i + lane];
                int cl = 0;
#pragma unroll
                for (int i = 0; i < M6A_SCAN_A_LOADS; i++) cl += ((w[i] & mask) <= rng) ? 1 : 0;
                scanned += STEP;
                if (scanned < (uint32_t)A) {             // uniform: cannot be the last step yet
                    carry += cl;
                } else {
                    if (scanned - STEP < (uint32_t)A) { acc += __builtin_amdgcn_readfirstlane(wave_sum_i32(carry)); carry = 0; }   // first total
                    const int tot = __builtin_amdgcn_readfirstlane(wave_sum_i32(cl));
                    if (acc + tot >= A) {
                        // the site's last accepted draw is in this step: walk its loads in stream order
                        // (fully unrolled so w[] stays in registers; the early exit is a flag, not a break)
                        bool found = false;
#pragma unroll
                        for (int i = 0; i < M6A_SCAN_A_LOADS; i++) {
                            if (found) continue;
                            const bool ok = (w[i] & mask) <= rng;
                            const unsigned long long bal = __ballot(ok);
                            const int c = __popcll(bal);
                            if (acc + c >= A) {
                                const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32),
                                                 __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
                                const unsigned long long lastb = __ballot(ok && rank == A - acc - 1);
                                pos += 64 * i + (uint32_t)__builtin_ctzll(lastb) + 1;
                                found = true;
                            } else {
                                acc += c;
                            }
                        }
                        break;
                    }
                    acc += tot;
                }
                pos += STEP;
#pragma unroll
                for (int i = 0; i < M6A_SCAN_A_LOADS; i++) w[i] = wn[i];
            }
        }
    }
}

// =====================================================================================
// Site pooling, uniform bags (every site has the same n <= 32 reads), K = 20.
// Same arithmetic as the scan kernels; because every site consumes the stream identically,
// the accepted indices of "the j-th site of a flush group" are the same in every group:
// tab[j][round][plane][lane] packs, for iteration t = 64*round + lane, byte offsets 8*idx of
// its 20 draws (5 dwords).
//
// A workgroup is bound to one position j (blockIdx % jmax -- with the observed block -> XCD
// round-robin each XCD's L2 then serves only jmax/8 rows) and keeps a 16-round chunk of row j
// in LDS (20 KB; the whole row when T <= 1024).  Each wavefront takes position j of 8
// consecutive groups: the 8 bags sit in LDS as 4 arrays of float2 (two sites per 8-byte entry,
// 160 B per array, each array inside one 256 B bank row), so every ds_read_b64 gathers for two
// sites, conflict-free, and v_pk_mul_f32 advances both products.  Inner loop = LDS + VALU only.
// =====================================================================================
#define M6A_TAB_RC 16                       // rows per LDS chunk
#define M6A_TAB_REG_STACK 8                  // merge stacks up to this height live in registers (T <= ~16k)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void pool_table_kernel(PoolArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_idx[M6A_TAB_RC * 5 * 64];
    __shared__ __attribute__((aligned(16))) uint32_t s_meta[M6A_TAB_RC * 4];    // row_meta of the chunk
    // bag arrays: pair p of wave w lives at byte p*2112 + w*256 (+ 8*idx + 4*e).  The odd 2112 B
    // pair stride is deliberate: no difference of two pair bases fits ds_read2_b64's offset field
    // (<= 2040 B) or is a multiple of ds_read2st64_b64's 512 B unit, so the four gathers of one
    // draw stay four ds_read_b64 (256 B/clk, 64-bank addressing: the 20 entries of a bag never
    // collide).  Fused into ds_read2_b64 they run at half rate and bank modulo 32 dwords, where
    // entries i and i+16 collide -- measured 2.4 conflict cycles per gather.
    __shared__ __attribute__((aligned(256))) float s_bag[4 * 528];
    // per wave: stage[8 sites][8 leaf sums of a pass] | rem[8 sites][8 tail values] | stack[8 sites][depth]
    extern __shared__ __attribute__((aligned(16))) float s_mean[];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // scalar: site ids, offsets -> SGPRs
    float *bag = s_bag + wib * 64;             // + pair * 528 floats
    const char *bagb = (const char *)bag;
    const int depth = a.stack_depth;
    float *w_stage = s_mean + wib * (128 + 8 * depth);
    float *w_rem = w_stage + 64;
    float *w_stack = w_rem + 64 + (lane & 7) * depth;      // lane q < 8 owns site q's stack
    const int n = a.uniform_n;
    const int rows = a.n_rows;
    const int n_chunks = (rows + M6A_TAB_RC - 1) / M6A_TAB_RC;
    const int j = (int)(blockIdx.x % (unsigned)a.jmax);
    const int64_t c = blockIdx.x / (unsigned)a.jmax;
    const int64_t nbj = gridDim.x / (unsigned)a.jmax;          // workgroups per position
    const int64_t gblocks = (a.n_groups + 7) >> 3;
    const int64_t stride = nbj * 4;
    const int64_t n_iter = (gblocks + stride - 1) / stride;
    const uint32_t *row = a.tab + (int64_t)j * rows * 5 * 64;

    if (n_chunks == 1) {
        for (int i = threadIdx.x; i < rows * 5 * 64; i += 256) s_idx[i] = row[i];
        if ((int)threadIdx.x < rows * 4) s_meta[threadIdx.x] = a.row_meta[threadIdx.x];
        __syncthreads();
    }
    for (int64_t it = 0; it < n_iter; ++it) {
        const int64_t gb = c * 4 + wib + it * stride;
        const bool active = gb < gblocks;
        int64_t site[8];
        float sum[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { site[q] = -1; sum[q] = 0.0f; }
        if (active) {
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int64_t g = gb * 8 + q;
                if (g < a.n_groups) {
                    const int64_t s = a.goff[g] + j;
                    if (s < a.goff[g + 1]) site[q] = s;
                }
                float v = 0.0f;
                bool ge = false;
                if (site[q] >= 0 && lane < n) {
                    v = a.read_prob[a.off[site[q]] + lane];
                    ge = v >= a.thr;
                }
                const int cge = __popcll(__ballot(ge));
                if (lane < 32) bag[(q >> 1) * 528 + 2 * lane + (q & 1)] = 1.0f - v;
                if (lane == 0 && site[q] >= 0) a.mod_ratio[site[q]] = (double)cge / (double)n;
            }
            wave_lds_fence();
        }
        int sp = 0;                              // stack height (wave-uniform; lanes 0..7 hold the stacks)
        float st[M6A_TAB_REG_STACK];
#pragma unroll
        for (int d = 0; d < M6A_TAB_REG_STACK; d++) st[d] = 0.0f;
        for (int ch = 0; ch < n_chunks; ++ch) {
            const int rd0 = ch * M6A_TAB_RC;
            const int nr = rows - rd0 < M6A_TAB_RC ? rows - rd0 : M6A_TAB_RC;
            if (n_chunks > 1) {
                __syncthreads();
                for (int i = threadIdx.x; i < nr * 5 * 64; i += 256) s_idx[i] = row[rd0 * 5 * 64 + i];
                if ((int)threadIdx.x < nr * 4) s_meta[threadIdx.x] = a.row_meta[rd0 * 4 + threadIdx.x];
                __syncthreads();
            }
            if (!active) continue;
            for (int rd = 0; rd < nr; ++rd) {
                float2 prod[4];
#pragma unroll
                for (int pr = 0; pr < 4; pr++) prod[pr] = make_float2(1.0f, 1.0f);
                const uint32_t *ixp = s_idx + rd * 5 * 64 + lane;
                const uint4 mt = *(const uint4 *)(s_meta + rd * 4);      // consumed after the gathers
#pragma unroll 1
                for (int pl = 0; pl < 5; pl++) {
                    const uint32_t ixw = ixp[pl * 64];
                    // all 16 gathers of the plane in flight before the first multiply; pinned, because
                    // the scheduler otherwise sometimes picks a minimum-register order that waits on every
                    // single gather (measured 0.83 -> 1.26 ms for the kernel)
                    float2 g[4][4];
#pragma unroll
                    for (int bb = 0; bb < 4; bb++) {
                        const uint32_t o = (ixw >> (8 * bb)) & 0xffu;
#pragma unroll
                        for (int pr = 0; pr < 4; pr++) g[bb][pr] = *(const float2 *)(bagb + pr * 2112 + o);
                    }
#pragma unroll
                    for (int bb = 0; bb < 4; bb++)
#pragma unroll
                        for (int pr = 0; pr < 4; pr++) {
                            prod[pr].x *= g[bb][pr].x;
                            prod[pr].y *= g[bb][pr].y;
                        }
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);     // 4 address adds
                    __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);    // 16 ds_read_b64
                    __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);    // 16 v_pk_mul_f32
                }
                // which iteration this (row, lane) was: lane = accumulator chain (lane & 7) of leaf
                // 8*pass + lane/8, row = round of the pass -- or the tail row of the last leaf.  Idle
                // lanes (shorter chains, leaves beyond the last) gathered entry 0 and are masked here.
                const uint32_t flags = (uint32_t)__builtin_amdgcn_readfirstlane((int)mt.x);
                const uint64_t lmask = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)mt.z) << 32) |
                                       (uint32_t)__builtin_amdgcn_readfirstlane((int)mt.y);
                if (flags & (1u << 25)) {
                    if (lane < 8) {
#pragma unroll
                        for (int pr = 0; pr < 4; pr++) {
                            w_rem[(2 * pr) * 8 + lane] = 1.0f - prod[pr].x;
                            w_rem[(2 * pr + 1) * 8 + lane] = 1.0f - prod[pr].y;
                        }
                    }
                    continue;
                }
                const bool live = (lmask >> lane) & 1;
#pragma unroll
                for (int pr = 0; pr < 4; pr++) {
                    sum[2 * pr] += live ? 1.0f - prod[pr].x : 0.0f;
                    sum[2 * pr + 1] += live ? 1.0f - prod[pr].y : 0.0f;
                }
                if (!(flags & (1u << 24))) continue;
                // pass complete: chains -> leaf sums, then lane q pushes site q's leaves in order and
                // merges as the tree says (merge_after nibbles ride in the row's meta)
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const float sv = chain8_sum(sum[q]);
                    if ((lane & 7) == 0) w_stage[q * 8 + (lane >> 3)] = sv;
                    sum[q] = 0.0f;
                }
                wave_lds_fence();
                if (lane < 8) {
                    const uint32_t mc = (uint32_t)__builtin_amdgcn_readfirstlane((int)mt.w);
                    const int nl = (int)((flags >> 26) & 0xfu);
                    const bool final_pass = flags & (1u << 30);
                    if (depth <= M6A_TAB_REG_STACK) {
                        // stack in registers (height is wave-uniform: select chains, no LDS round trips --
                        // with the LDS pipe saturated by the other waves' gathers each dependent LDS access
                        // here would cost microseconds)
                        float xs[8], tl[7];
#pragma unroll
                        for (int bl = 0; bl < 8; bl++) xs[bl] = w_stage[lane * 8 + bl];
#pragma unroll
                        for (int i = 0; i < 7; i++) tl[i] = w_rem[lane * 8 + i];
#pragma unroll
                        for (int bl = 0; bl < 8; bl++) {
                            if (bl >= nl) break;
                            float x = xs[bl];
                            if (final_pass && bl == nl - 1) {
#pragma unroll
                                for (int i = 0; i < 7; i++)
                                    if (i < a.n_rem) x += tl[i];
                            }
                            for (int m = (int)((mc >> (4 * bl)) & 15u); m > 0; --m) {
                                --sp;
                                float top = st[0];
#pragma unroll
                                for (int d = 1; d < M6A_TAB_REG_STACK; d++) top = sp == d ? st[d] : top;
                                x = top + x;
                            }
#pragma unroll
                            for (int d = 0; d < M6A_TAB_REG_STACK; d++) st[d] = sp == d ? x : st[d];
                            ++sp;
                        }
                    } else {
                        for (int bl = 0; bl < nl; bl++) {
                            float x = w_stage[lane * 8 + bl];
                            if (final_pass && bl == nl - 1)
                                for (int i = 0; i < a.n_rem; i++) x += w_rem[lane * 8 + i];
                            for (int m = (int)((mc >> (4 * bl)) & 15u); m > 0; --m) x = w_stack[--sp] + x;
                            w_stack[sp++] = x;
                        }
                    }
                }
                wave_lds_fence();
            }
        }
        if (active) {
            float tot = 0.0f;
            if (lane < 8) tot = (depth <= M6A_TAB_REG_STACK ? st[0] : w_stack[0]) / (float)a.T;
#pragma unroll
            for (int q = 0; q < 8; q++)
                if (lane == q && site[q] >= 0) a.site_prob[site[q]] = tot;
            wave_lds_fence();
        }
    }
}

// 1 - prod_k (1 - p[b*bag + k]), float32 left to right (pooling_blocks.py:127-129)
__global__ void bag_noisy_or_kernel(const float *read_prob, int64_t n_bags, int bag, float *site_prob)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_bags) return;
    float prod = 1.0f;
    for (int k = 0; k < bag; k++) prod *= 1.0f - read_prob[b * bag + k];
    site_prob[b] = 1.0f - prod;
}

// validation-style forward: y[b] = 1 - prod_k (1 - p[gidx[b][k]]), float32 left to right
// (training_utils.py:239 -> MILModel.forward -> pooling_blocks.py:127-129 on the sampled 20-read bag).
// A thread owns a bag; for k = 20 its index row is 80 contiguous, 16-byte aligned bytes: five dwordx4 loads, all 20
// gathers in flight before the first multiply.
__global__ void sampled_noisy_or_kernel(const float *read_prob, const int32_t *gidx, int64_t n_bags, int k, float *y)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_bags) return;
    const int32_t *row = gidx + b * k;
    float prod = 1.0f;
    if (k == 20) {
        int4 q[5];
#pragma unroll
        for (int j = 0; j < 5; j++) q[j] = ((const int4 *)row)[j];
        float g[20];
#pragma unroll
        for (int j = 0; j < 5; j++) {
            g[4 * j] = read_prob[q[j].x]; g[4 * j + 1] = read_prob[q[j].y];
            g[4 * j + 2] = read_prob[q[j].z]; g[4 * j + 3] = read_prob[q[j].w];
        }
#pragma unroll
        for (int j = 0; j < 20; j++) prod *= 1.0f - g[j];
    } else {
        for (int j = 0; j < k; j++) prod *= 1.0f - read_prob[row[j]];
    }
    y[b] = 1.0f - prod;
}

// np.mean(y, axis=0) of a C-contiguous float32 [n_iters][n_sites]: pass after pass, one divide
// (training_utils.py:253)
__global__ void mean_over_passes_kernel(const float *y, int n_iters, int64_t n_sites, float *avg)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_sites) return;
    float acc = 0.0f;
    for (int t = 0; t < n_iters; t++) acc += y[(int64_t)t * n_sites + s];
    avg[s] = acc / (float)n_iters;
}

// out[i] = off[i] - off[0]: the CSR row of a chunk of sites, rebased to the chunk's first read
__global__ void rebase_off_kernel(const int64_t *off, int64_t count, int64_t *out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = off[i] - off[0];
}

__global__ void iota_off_kernel(int64_t *off, int64_t n_plus_1, int64_t step)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_plus_1) off[i] = i * step;
}

// min / max bag size over all sites -> out[0], out[1] (initialised by the host to UINT64_MAX, 0);
// out[2] = off[n_sites] (total reads); hist[n] += sites with n reads (n <= M6A_RTAB_MAX_N, the last bin takes
// the larger ones; zeroed by the host).  One atomic pair + the touched bins per workgroup.
__global__ __launch_bounds__(256) void bag_minmax_kernel(const int64_t *off, int64_t n_sites, unsigned long long *out, uint32_t *hist)
{
    __shared__ int64_t s_mn[4], s_mx[4];
    __shared__ uint32_t s_hist[M6A_HIST_BINS];
    for (int i = threadIdx.x; i < M6A_HIST_BINS; i += 256) s_hist[i] = 0;
    __syncthreads();
    int64_t mn = INT64_MAX, mx = 0;
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_sites;
         s += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = off[s + 1] - off[s];
        mn = n < mn ? n : mn;
        mx = n > mx ? n : mx;
        atomicAdd(&s_hist[n < 0 ? 0 : n > M6A_RTAB_MAX_N ? M6A_RTAB_MAX_N + 1 : n], 1u);
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const int64_t omn = __shfl_xor(mn, m, 64), omx = __shfl_xor(mx, m, 64);
        mn = omn < mn ? omn : mn;
        mx = omx > mx ? omx : mx;
    }
    if ((threadIdx.x & 63) == 0) { s_mn[threadIdx.x >> 6] = mn; s_mx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    for (int i = threadIdx.x; i < M6A_HIST_BINS; i += 256)
        if (s_hist[i]) atomicAdd(&hist[i], s_hist[i]);
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) { mn = s_mn[w] < mn ? s_mn[w] : mn; mx = s_mx[w] > mx ? s_mx[w] : mx; }
        atomicMin(&out[0], (unsigned long long)mn);
        atomicMax(&out[1], (unsigned long long)mx);
        if (blockIdx.x == 0) out[2] = (unsigned long long)off[n_sites];
    }
}

// m6a_set_host_offsets: the host took the bag statistics from its own copy of off[]; this checks them against what
// bag_minmax_kernel found in the device array (no read-back): range, total reads, and a hash of the bag-size histogram
__global__ __launch_bounds__(256) void bag_verify_kernel(const unsigned long long *got, const uint32_t *hist, unsigned long long mn,
                                                         unsigned long long mx, unsigned long long reads, unsigned long long hash, int *err)
{
    __shared__ unsigned long long s_part[256];
    unsigned long long h = 0;
    for (int i = threadIdx.x; i < M6A_HIST_BINS; i += 256) h += (unsigned long long)hist[i] * m6a_bin_weight(i);
    s_part[threadIdx.x] = h;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 256; i++) h += s_part[i];
        if (got[0] != mn || got[1] != mx || got[2] != reads || h != hash) atomicExch(err, 3);
    }
}

// An empty kernel per translation unit: HIP maps a code object on the first launch of any kernel in it (0.3-1.2 ms, measured
// in the first call's timeline, profiles/r03_first_call_timeline.txt); m6a_create's background set-up launches these instead.
__global__ void m6a_touch_kernels() {}
