// m6a_pool_rtab.hip -- site pooling for RAGGED bags through per-bag-size index tables, and the
// device-side generators they stand on (MT19937 word stream, accepted-index compaction).
//
// What the reference does per site (m6anet/utils/inference_utils.py:85-86):
//     proba = np.random.choice(proba, n_iters*n_samples, replace=True).reshape(n_iters, n_samples)
//     (1 - np.prod(1 - proba, axis=1)).mean()
// with every flush group's worker starting from the same seeded MT19937 state (:102-104), i.e. all
// groups read ONE word stream `raw` from word 0, the sites of a group back to back, each by legacy
// randint's masked rejection (v = w & mask; accept iff v < n).
//
// Whether a word is accepted depends only on (word, n).  So for every bag size n that occurs, the
// accepted draws of the WHOLE stream are one fixed sequence
//     C_n = [ w & mask_n  for w in raw  if (w & mask_n) < n ]
// and a site of size n that starts at stream word p simply uses C_n[r .. r + T*K) with
// r = rank_n(p) = #accepted words before p; it ends at the word after the (r + T*K - 1)-th accepted
// one, where the group's next site starts.  That splits the job into
//   rtab_count/scan/fill   C_n (index bytes for n <= 256, u16 byte offsets 4*v above) + RS_n (rank at every
//                          64-word block) for each new bag size -- built once per (seed, T*K), kept across calls;
//   rtab_prep_kernel       one wavefront per flush group walks its <= 32 sites: rank lookup, jump
//                          T*K ranks ahead, select -- three dependent L2 round trips per site,
//                          no stream scanning at all; the same launch orders the sites by bag size;
//                          m6a_infer runs it on a side stream under the encoder;
//   pool_rtab_kernel       one wavefront per site, sites ordered by bag size (a table of 1-3 MB
//                          stays in one XCD's L2 while its sites run): lane = iteration, a lane
//                          loads its 20 indices as one 20- or 40-byte row (aligned dwords, picked apart
//                          at the site's alignment class), gathers 1-p from the LDS bag (one ds_read_b32
//                          per draw, no compaction, no ballots) and multiplies left to right.
//                          Bound: first the index rows through the vector L1, after the byte tables the
//                          LDS gathers under random bank conflicts (profiles/r02_ragged_rows.txt).
// The mean over iterations follows NumPy's pairwise sum exactly as the other pooling kernels do
// (MeanPlan, m6a_api.hip): lane = accumulator chain (lane & 7) of leaf 8*pass + lane/8.
#include "m6a_kernels.h"

namespace {

__device__ __forceinline__ uint32_t pow2_mask_u32(uint32_t rng)
{
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    return mask;
}

__device__ __forceinline__ float chain8_sum_r(float r)
{
    r += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, r), 0xB1, 0xf, 0xf, false));
    r += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, r), 0x4E, 0xf, 0xf, false));
    r += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, r), 0x141, 0xf, 0xf, false));
    return r;
}

__device__ __forceinline__ void wave_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ int64_t uni64(int64_t v)
{
    return ((int64_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

}  // namespace

// =====================================================================================
// MT19937 raw word stream on the device: np.random.seed(int) == init_genrand == std::mt19937(seed)
// (m6anet/scripts/inference.py:86).  With x[0..623] the seeded state, the generator's words are
// temper(x[624]), temper(x[625]), ... where
//     x[j] = x[j-227] ^ T(j-624),      T(a) = twist(x[a], x[a+1]).
// As written that is only 227-wide: x[j] needs x[j-227].  It is linear over GF(2), so substituting it
// into itself twice gives
//     x[j] = x[j-681] ^ T(j-1078) ^ T(j-851) ^ T(j-624),
// whose newest operand is x[j-623]: 623 consecutive words can be computed at once from older ones.
// One workgroup of 640 threads, the last 2048 words as a ring in LDS, ONE barrier per 623 words
// (the first 454 generated words have no x[j-1078] yet and take two steps of the plain form).  The
// barrier waits for LDS traffic only (lgkmcnt); the tempered stores stay in flight.  Measured on the
// default 1.3 M words: three 227-wide phases per 624 words 1.5 ms, this form see profiles/.
// =====================================================================================
__device__ __forceinline__ uint32_t mt_twist(uint32_t a, uint32_t b)
{
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#define MT_RING 2048
// Segment blockIdx.x of the stream: output words [i*seg_words, min((i+1)*seg_words, n_words)) -- the last segment runs to
// n_words.  Segment 0 starts from the seed; segment i >= 1 from its 1078-word history x[i*G - 454 .. i*G + 623], which
// mt_jump_kernel computed from the head of the stream (hist[i-1]).  xhead (or null): the untempered words x[0 .. xhead_n) of
// segment 0, the jump's input.  A single segment with seg_words >= n_words is the plain sequential generator.
__global__ __launch_bounds__(640) void mt19937_kernel(uint32_t seed, int64_t n_words, uint32_t *raw, const uint32_t *hist, int64_t seg_words,
                                                      uint32_t *xhead, int64_t xhead_n)
{
    __shared__ uint32_t x[MT_RING];
    const int k = threadIdx.x;
    const int64_t seg = blockIdx.x;
    const int64_t w_begin = seg * seg_words;
    const int64_t w_end = (seg + 1 == (int64_t)gridDim.x) ? n_words : (n_words < w_begin + seg_words ? n_words : w_begin + seg_words);
    auto emit = [&](int64_t j, uint32_t v) {                 // x[j] -> ring, temper(x[j]) -> raw[j - 624]
        x[j & (MT_RING - 1)] = v;
        if (xhead && j < xhead_n) xhead[j] = v;
        uint32_t y = v;
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        if (j - 624 < w_end) raw[j - 624] = y;
    };
    int64_t j_first;
    if (seg == 0) {
        if (k == 0) {
            uint32_t v = seed;
            x[0] = v;
            for (uint32_t i = 1; i < 624; i++) { v = 1812433253u * (v ^ (v >> 30)) + i; x[i] = v; }
        }
        lds_barrier();
        if (xhead) for (int i = k; i < 624 && i < xhead_n; i += 640) xhead[i] = x[i];
        // x[624..850] and x[851..1077]: the plain recurrence, 227 wide
        for (int64_t j0 = 624; j0 < 1078; j0 += 227) {
            if (k < 227) {
                const int64_t j = j0 + k;
                emit(j, x[(j - 227) & (MT_RING - 1)] ^ mt_twist(x[(j - 624) & (MT_RING - 1)], x[(j - 623) & (MT_RING - 1)]));
            }
            lds_barrier();
        }
        j_first = 1078;
    } else {
        j_first = 624 + w_begin;                             // history x[j_first - 1078 .. j_first - 1] from the jump
        const uint32_t *h = hist + (seg - 1) * 1078;
        for (int m = k; m < 1078; m += 640) x[(j_first - 1078 + m) & (MT_RING - 1)] = h[m];
        lds_barrier();
    }
    for (int64_t j0 = j_first; j0 - 624 < w_end; j0 += 623) {
        if (k < 623) {
            const int64_t j = j0 + k;
            const uint32_t v = x[(j - 681) & (MT_RING - 1)] ^
                               mt_twist(x[(j - 1078) & (MT_RING - 1)], x[(j - 1077) & (MT_RING - 1)]) ^
                               mt_twist(x[(j - 851) & (MT_RING - 1)], x[(j - 850) & (MT_RING - 1)]) ^
                               mt_twist(x[(j - 624) & (MT_RING - 1)], x[(j - 623) & (MT_RING - 1)]);
            emit(j, v);
        }
        // the ring holds 2048 words: step s writes x[j0 .. j0+622], the oldest word step s+1 reads is x[j0+623-1078]
        lds_barrier();
    }
}

// Jump ahead: the history of segment seg+1, hist[seg][m] = x[(seg+1)*G - 454 + m], from the head of the stream.  MT19937's
// word sequence is linear over GF(2): x[k + D] = XOR over the set coefficients t of (t^D mod phi) of x[k + t] (phi: the
// generator's characteristic polynomial, degree 19937; tools/make_mt_jump.py derives it and writes the polynomials
// r = t^((seg+1)*G - 512) mod phi this kernel reads: hist[seg][m] = XOR_t x[58 + m + t]).  grid (17, n_seg - 1): a workgroup
// = 64 history words, its four waves take a quarter of the 19937 coefficients each; every coefficient is a predicated,
// coalesced load (half are zero: cheaper than a data-dependent loop that would expose L2 latency per term).
#define MT_JUMP_WORDS 312            // u64 per polynomial
__global__ __launch_bounds__(256) void mt_jump_kernel(const uint32_t *xhead, const uint64_t *polys, uint32_t *hist)
{
    __shared__ uint32_t part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = blockIdx.x * 64 + lane;
    const uint64_t *p = polys + (size_t)blockIdx.y * MT_JUMP_WORDS + wave * 78;
    const uint32_t *src = xhead + 58 + m + wave * 78 * 64;
    uint32_t acc = 0;
    for (int w = 0; w < 78; w++) {
        const uint64_t bits = p[w];                          // wave-uniform
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bits);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bits >> 32));
#pragma unroll
        for (int t = 0; t < 32; t++) acc ^= src[w * 64 + t] & (0u - ((lo >> t) & 1u));
#pragma unroll
        for (int t = 0; t < 32; t++) acc ^= src[w * 64 + 32 + t] & (0u - ((hi >> t) & 1u));
    }
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && m < 1078) hist[(size_t)blockIdx.y * 1078 + m] = part[0][lane] ^ part[1][lane] ^ part[2][lane] ^ part[3][lane];
}

// =====================================================================================
// Table build.  One wavefront = (one new bag size, one chunk of 16 blocks of 64 stream words); the bag size is
// the fast grid axis, so a chunk of the stream is read by all bag sizes while it sits in L2.
//   count: accepted words per block -> RS[slot][b]
//   scan:  exclusive prefix per slot, total in RS[slot][n_blk]
//   fill:  C[slot][RS[b] + rank] = w & mask as a byte (bags <= 256 reads: the slot's first half) or 4 * (w & mask) as
//          u16 (the byte offset into the float bag)
// =====================================================================================
#define RTAB_CHUNK 16

// A workgroup column builds RTAB_GROUP bag sizes at once: every 64-word block of the stream is loaded ONCE and tested against
// all of them.  With one size per column (rounds 2-3) the 5 MB stream crossed the L2 -> CU way once per size -- 2.4 GB for the
// 451 sizes of a 50..500-read job, which is what the 0.48 + 0.53 ms of the two passes were (the arithmetic is a dozen
// instructions per 64 words).
#define RTAB_GROUP 8
__global__ __launch_bounds__(256) void rtab_count_kernel(RtabBuild a)
{
    const int lane = threadIdx.x & 63;
    const int g0 = blockIdx.x * RTAB_GROUP;
    int n[RTAB_GROUP];
    int64_t slot[RTAB_GROUP];
    uint32_t mask[RTAB_GROUP];
#pragma unroll
    for (int q = 0; q < RTAB_GROUP; q++) {
        const bool has = g0 + q < a.n_build;
        n[q] = has ? a.build_n[g0 + q] : 0;                  // n = 0: nothing is accepted, nothing is written
        slot[q] = has ? a.build_slot[g0 + q] : 0;
        mask[q] = n[q] ? pow2_mask_u32((uint32_t)(n[q] - 1)) : 0u;
    }
    for (uint32_t chunk = blockIdx.y * 4 + (threadIdx.x >> 6); (uint64_t)chunk * RTAB_CHUNK < a.n_blk; chunk += gridDim.y * 4) {
        const uint32_t b0 = chunk * RTAB_CHUNK;
        uint32_t mine[RTAB_GROUP];
#pragma unroll
        for (int q = 0; q < RTAB_GROUP; q++) mine[q] = 0;
#pragma unroll
        for (int i = 0; i < RTAB_CHUNK; i++) {
            const uint32_t b = b0 + i;
            const bool in = b < a.n_blk;
            const uint32_t w = in ? a.raw[(int64_t)b * 64 + lane] : 0xffffffffu;
#pragma unroll
            for (int q = 0; q < RTAB_GROUP; q++) {
                const uint32_t c = (uint32_t)__popcll(__ballot(in && (w & mask[q]) < (uint32_t)n[q]));
                if (lane == i) mine[q] = c;
            }
        }
        if (lane < RTAB_CHUNK && b0 + lane < a.n_blk) {
#pragma unroll
            for (int q = 0; q < RTAB_GROUP; q++)
                if (n[q]) a.RS[slot[q] * ((int64_t)a.n_blk + 1) + b0 + lane] = mine[q];
        }
    }
}

__global__ __launch_bounds__(256) void rtab_scan_kernel(RtabBuild a)
{
    __shared__ uint32_t s_part[256];
    uint32_t *rs = a.RS + (int64_t)a.build_slot[blockIdx.x] * ((int64_t)a.n_blk + 1);
    const uint32_t per = (a.n_blk + 255) / 256;
    const uint32_t lo = threadIdx.x * per, hi = lo + per < a.n_blk ? lo + per : a.n_blk;
    uint32_t sum = 0;
    for (uint32_t b = lo; b < hi; b++) sum += rs[b];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int i = 0; i < 256; i++) { const uint32_t t = s_part[i]; s_part[i] = run; run += t; }
        rs[a.n_blk] = run;
    }
    __syncthreads();
    uint32_t run = s_part[threadIdx.x];
    for (uint32_t b = lo; b < hi; b++) { const uint32_t t = rs[b]; rs[b] = run; run += t; }
}

__global__ __launch_bounds__(256) void rtab_fill_kernel(RtabBuild a)
{
    const int lane = threadIdx.x & 63;
    const int g0 = blockIdx.x * RTAB_GROUP;
    int n[RTAB_GROUP];
    uint32_t mask[RTAB_GROUP];
    const uint32_t *rs[RTAB_GROUP];
    uint16_t *C[RTAB_GROUP];
#pragma unroll
    for (int q = 0; q < RTAB_GROUP; q++) {
        const bool has = g0 + q < a.n_build;
        n[q] = has ? a.build_n[g0 + q] : 0;
        const int64_t slot = has ? a.build_slot[g0 + q] : 0;
        mask[q] = n[q] ? pow2_mask_u32((uint32_t)(n[q] - 1)) : 0u;
        rs[q] = a.RS + slot * ((int64_t)a.n_blk + 1);
        C[q] = a.C + slot * a.c_stride;
    }
    for (uint32_t chunk = blockIdx.y * 4 + (threadIdx.x >> 6); (uint64_t)chunk * RTAB_CHUNK < a.n_blk; chunk += gridDim.y * 4) {
        const uint32_t b0 = chunk * RTAB_CHUNK;
        uint32_t pre[RTAB_GROUP];
#pragma unroll
        for (int q = 0; q < RTAB_GROUP; q++) pre[q] = (n[q] && lane < RTAB_CHUNK && b0 + lane < a.n_blk) ? rs[q][b0 + lane] : 0u;
#pragma unroll
        for (int i = 0; i < RTAB_CHUNK; i++) {
            const uint32_t b = b0 + i;
            if (b >= a.n_blk) break;
            const uint32_t w = a.raw[(int64_t)b * 64 + lane];
#pragma unroll
            for (int q = 0; q < RTAB_GROUP; q++) {
                const uint32_t v = w & mask[q];
                const bool ok = v < (uint32_t)n[q];
                const unsigned long long bal = __ballot(ok);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
                const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)pre[q], i);
                if (ok) {
                    if (n[q] <= M6A_RTAB_U8_MAX_N) ((uint8_t *)C[q])[(int64_t)base + rank] = (uint8_t)v;      // index
                    else C[q][(int64_t)base + rank] = (uint16_t)(4u * v);                                   // byte offset
                }
            }
        }
    }
}

// accepted-index table of pool_reg_kernel from a C_n row: idx16[j][T+8][K] 16-bit words, row j = draws [j*T*K, (j+1)*T*K) of
// the stream (site j of every flush group of uniform bags).  A word is the value M0 takes for the draw: 0x1000 (the index
// mode's SRC0 enable, M0[15:12]) | 2 x index (a register pair per bag entry, M0[7:0]) -- m6a_pool_reg.hip
__global__ __launch_bounds__(256) void rtab_to_reg_table_kernel(const uint16_t *C, int64_t A, int64_t row_words, int jmax, uint16_t *tab)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= A * jmax) return;
    const int64_t j = e / A, w = e - j * A;
    tab[j * row_words + w] = (uint16_t)(0x1000u | (2u * ((const uint8_t *)C)[e]));      // uniform bags are <= 32 reads: a byte table
}

// accepted-index table of pool_table_kernel from a C_n row: tab[j][row][plane][lane], four byte offsets (8 * index) per
// dword -- draws 4*plane .. 4*plane+3 of the iteration that (row, lane) holds in the pairwise-sum plan (-1: idle lane)
__global__ __launch_bounds__(256) void rtab_to_lds_table_kernel(const uint16_t *C, int64_t A, int K, int rows, int jmax, const int *iter_of,
                                                                 uint32_t *tab)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)jmax * rows * 5 * 64) return;
    const int lane = (int)(e & 63), plane = (int)((e >> 6) % 5), r = (int)((e / 320) % rows);
    const int64_t j = e / ((int64_t)320 * rows);
    const int t = iter_of[r * 64 + lane];
    uint32_t w = 0;
    if (t >= 0) {
        const uint8_t *row = (const uint8_t *)C + j * A + (int64_t)t * K;      // uniform bags are <= 32 reads: a byte table
        for (int q = 0; q < 4; q++) {
            const int kk = 4 * plane + q;
            if (kk < K) w |= ((uint32_t)row[kk] * 8u) << (8 * q);
        }
    }
    tab[e] = w;
}

// =====================================================================================
// Where each site starts: one wavefront per flush group, its sites in order.
//   rank r of the site's first draw in C_n  ->  rank_out[s]
//   p' = (stream word of accepted draw number r + T*K - 1) + 1   -> next site
// =====================================================================================
__device__ __forceinline__ void rtab_chain_part(const PoolArgs &a, const RtabUse &u, unsigned block, unsigned n_blocks)
{
    const int lane = threadIdx.x & 63;
    const int64_t n_waves = (int64_t)n_blocks * 4;
    const uint32_t A = (uint32_t)(a.T * a.K);
    const uint32_t n_blk = u.n_blk;
    for (int64_t g = (int64_t)block * 4 + uni((int)(threadIdx.x >> 6)); g < a.n_groups; g += n_waves) {
        uint32_t p = 0;
        const int64_t s_beg = a.goff[g], s_end = a.goff[g + 1];
        uint32_t w = a.raw[lane];                     // the stream block that holds word p
        uint32_t w_blk = 0;
        int64_t nn = s_beg < s_end ? a.off[s_beg + 1] - a.off[s_beg] : 0;
        for (int64_t s = s_beg; s < s_end; ++s) {
            const int64_t n_this = nn;
            if (s + 1 < s_end) nn = a.off[s + 2] - a.off[s + 1];         // next site's size: off the critical path
            if (n_this <= 1) { if (lane == 0) u.rank[s] = 0; continue; }  // randint(0,1) draws no words
            const uint32_t n = (uint32_t)n_this;
            const uint32_t *rs = u.RS + (int64_t)u.slot_of_n[n] * ((int64_t)n_blk + 1);
            const uint32_t mask = pow2_mask_u32(n - 1);
            const uint32_t total = rs[n_blk];
            // rank of stream word p: directory entry of its block + the accepted words of the block before p
            const uint32_t b = p >> 6;
            bool short_stream = b >= n_blk;
            uint32_t r = 0;
            if (!short_stream) {
                if (b != w_blk) { w = a.raw[(int64_t)b * 64 + lane]; w_blk = b; }
                const unsigned long long bal = __ballot((w & mask) < n);
                r = rs[b] + (uint32_t)__popcll(bal & ((1ull << (p & 63)) - 1ull));
                short_stream = (uint64_t)r + A - 1 >= total;
            }
            if (short_stream) {                                         // the stream proved too short
                if (lane == 0) { atomicExch(a.err, 1); for (int64_t q = s; q < s_end; ++q) u.rank[q] = 0xffffffffu; }
                break;
            }
            if (lane == 0) u.rank[s] = r;
            const uint32_t e = r + A - 1;                               // rank of the site's last draw
            // block b2 with RS[b2] <= e < RS[b2+1]: linear guess, then a 64-entry window of the directory
            // (tried: guessing the window from p and the table's acceptance rate BEFORE r is known, so that its load
            // travels with the two loads r needs -- two dependent round trips per site instead of three -- plus a
            // look-ahead of bag size / slot / total three sites deep: 19 us slower per 125 k sites, 0.614 vs 0.595 ms)
            uint32_t b0 = (uint32_t)(((uint64_t)e * n_blk) / total);
            b0 = b0 > 32 ? b0 - 32 : 0;
            uint32_t b2, rs_b2;
            for (;;) {
                if (b0 + 63 > n_blk) b0 = n_blk >= 63 ? n_blk - 63 : 0;
                const uint32_t idx = b0 + lane;
                const uint32_t v = idx <= n_blk ? rs[idx] : 0xffffffffu;
                const int c = __popcll(__ballot(v <= e));               // a prefix of the lanes: RS is non-decreasing
                if (c == 0) { b0 = b0 > 62 ? b0 - 62 : 0; continue; }   // b0 == 0 cannot get here: RS[0] = 0 <= e
                if (c == 64) { b0 += 62; continue; }                    // e < total guarantees the search ends
                b2 = b0 + (uint32_t)c - 1;
                rs_b2 = (uint32_t)__builtin_amdgcn_readlane((int)v, c - 1);
                break;
            }
            const uint32_t kth = e - rs_b2;                             // 0-based among block b2's accepted words
            if (b2 != w_blk) { w = a.raw[(int64_t)b2 * 64 + lane]; w_blk = b2; }
            const bool ok2 = (w & mask) < n;
            const unsigned long long bal2 = __ballot(ok2);
            const uint32_t rank2 = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal2, 0));
            const unsigned long long hit = __ballot(ok2 && rank2 == kth);
            p = (uint32_t)uni((int)(b2 * 64 + (uint32_t)__builtin_ctzll(hit) + 1));
        }
    }
}

// sites ordered by bag size: cursor[n] starts at the exclusive prefix of the bag-size histogram.  A workgroup
// counts its 256 sites in LDS first and takes ONE global slot range per bag size it holds (uniform bags would
// otherwise serialise every site on a single atomic).
__device__ __forceinline__ void rtab_order_part(const int64_t *off, int64_t n_sites, uint32_t *cursor, uint32_t *order, unsigned block)
{
    __shared__ uint32_t s_cnt[M6A_HIST_BINS], s_base[M6A_HIST_BINS];
    for (int i = threadIdx.x; i < M6A_HIST_BINS; i += 256) s_cnt[i] = 0;
    __syncthreads();
    const int64_t s = (int64_t)block * 256 + threadIdx.x;
    int64_t n = 0;
    uint32_t mine = 0;
    if (s < n_sites) {
        n = off[s + 1] - off[s];
        n = n < 0 ? 0 : n > M6A_RTAB_MAX_N ? M6A_RTAB_MAX_N + 1 : n;
        mine = atomicAdd(&s_cnt[n], 1u);
    }
    __syncthreads();
    if (s < n_sites && mine == 0) s_base[n] = atomicAdd(&cursor[n], s_cnt[n]);
    __syncthreads();
    if (s < n_sites) order[s_base[n] + mine] = (uint32_t)s;
}

// One launch for both preparations of a call -- they are independent, the chain walk is a latency chain on few
// waves and the ordering a burst of atomics, so they overlap: workgroups [0, n_order_blocks) order the sites, the
// rest walk the flush groups.
__global__ __launch_bounds__(256) void rtab_prep_kernel(PoolArgs a, RtabUse u, uint32_t *cursor, uint32_t *order, unsigned n_order_blocks)
{
    if (blockIdx.x < n_order_blocks) {
        rtab_order_part(a.off, a.n_sites, cursor, order, blockIdx.x);
    } else {
        rtab_chain_part(a, u, blockIdx.x - n_order_blocks, gridDim.x - n_order_blocks);
    }
}

// =====================================================================================
// The pooling proper: one wavefront per site.
// =====================================================================================
// The K = 20 draws of one iteration are one table row: 40 bytes of u16 byte offsets at byte 2 * (rank + 20 t), or -- bags of
// at most 256 reads -- 20 index bytes at byte rank + 20 t.  The rows are what bounds this kernel (profiles/r02_ragged_rows.txt:
// 2 bytes per draw through the vector L1), hence the byte tables; and a row that is not dword-aligned costs the address path
// twice (0.71 -> 0.58 ms per launch with every rank forced even), so a lane always loads the ALIGNED dwords around its row
// and picks its indices out of them at the site's alignment class (rank mod 2, or rank mod 4 for byte rows): wave-uniform,
// one instantiation of the loop per class, and the SDWA extract picks any half or byte for free.
//   MODE 0 / 1: u16 rows, rank even / odd;  MODE 2..5: byte rows, rank mod 4 = MODE - 2.
// 4 * byte B of w in ONE instruction (the compiler's own choice is v_bfe_u32 + v_lshl_add_u32): an SDWA shift whose source
// operand selects the byte.  SDWA takes no literal, the shift count rides in a register.
#define RTAB_BAG_LDS ((16 + M6A_MEAN_STACK) * 4)     // byte offset of the bag in the kernel's LDS block: stage[8] | tail[8] | stack | bag

template <int B>
__device__ __forceinline__ uint32_t byte_times4(uint32_t w, uint32_t two)
{
    uint32_t r;
    if (B == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(two), "v"(w));
    else if (B == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(two), "v"(w));
    else if (B == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(two), "v"(w));
    else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(two), "v"(w));
    return r;
}

template <int MODE, int J>
struct ByteGather {                                                     // draws J..19 of a byte row (compile-time byte positions)
    template <typename Row>
    static __device__ __forceinline__ void run(const Row &r, const char *bagb, uint32_t two, float (&g)[20])
    {
        constexpr int b = J + MODE - 2;
        // the bag sits at a compile-time LDS address (RTAB_BAG_LDS; the kernel checks it): as an address_space(3) access
        // the constant goes into the instruction's offset field; through the generic `bagb + x` the compiler spends a
        // v_add_u32 per draw on the (zero) base of the dynamic LDS block
        typedef const __attribute__((address_space(3))) float lds_cfloat;
        g[J] = *(lds_cfloat *)(uintptr_t)(byte_times4<(b & 3)>(r.w[b >> 2], two) + RTAB_BAG_LDS);
        ByteGather<MODE, J + 1>::run(r, bagb, two, g);
    }
};
template <int MODE>
struct ByteGather<MODE, 20> {
    template <typename Row>
    static __device__ __forceinline__ void run(const Row &, const char *, uint32_t, float (&)[20]) {}
};

template <int MODE>
__device__ __forceinline__ float rtab_product20(const char *row, const char *bagb)
{
    float g[20];
    if (MODE < 2) {
        constexpr int ND = 10 + MODE;
        struct __attribute__((aligned(4))) Row { uint32_t w[ND]; };
        const Row r = *(const Row *)(row - 2 * MODE);
#pragma unroll
        for (int j = 0; j < 20; j++) {
            const int h = j + MODE;                                     // halfword of the aligned dwords
            g[j] = *(const float *)(bagb + ((h & 1) ? r.w[h >> 1] >> 16 : r.w[h >> 1] & 0xffffu));
        }
    } else {
        constexpr int AL = MODE - 2, ND = (AL + 20 + 3) / 4;
        struct __attribute__((aligned(4))) Row { uint32_t w[ND]; };
        const Row r = *(const Row *)(row - AL);
        uint32_t two;
        asm("v_mov_b32 %0, 2" : "=v"(two));                             // opaque: a known constant would be folded back into v_bfe + shift
        ByteGather<(MODE < 2 ? 2 : MODE), 0>::run(r, bagb, two, g);     // draw j = byte j + AL of the aligned dwords
    }
    float prod = 1.0f;
#pragma unroll
    for (int k = 0; k < 20; k++) prod *= g[k];                          // left to right: np.prod's order
    return prod;
}

template <bool BYTES>
__device__ __forceinline__ float rtab_product_any(const char *row, const char *bagb, int K)
{
    float prod = 1.0f;
    for (int k = 0; k < K; k++)
        prod *= *(const float *)(bagb + (BYTES ? 4u * ((const uint8_t *)row)[k] : (uint32_t)((const uint16_t *)row)[k]));
    return prod;
}

// a lane's chain of one pass: rounds of 8 iterations apart (row stride `step` bytes); lanes whose leaf is shorter re-read
// the site's first row and add nothing
template <int KT, int MODE>
__device__ __forceinline__ float rtab_chain(const char *row, const char *tb, int64_t step, int rounds, int my_rounds, const char *bagb, int K)
{
    float sum = 0.0f;
    for (int i = 0; i < rounds; ++i) {
        const bool live = i < my_rounds;
        const char *rp = live ? row : tb;
        const float v = 1.0f - (KT == 20 ? rtab_product20<MODE>(rp, bagb) : rtab_product_any<(MODE >= 2)>(rp, bagb, K));
        sum += live ? v : 0.0f;
        row += step;
    }
    return sum;
}

template <int KT>
__device__ __forceinline__ float rtab_chain_mode(int mode, const char *row, const char *tb, int64_t step, int rounds, int my_rounds, const char *bagb, int K)
{
    // (Tried on the configs[4] shape and dropped, neither faster than the plain loop: the index row of round i+1 in flight
    // while round i runs -- by hand with inline-asm loads before the rows were aligned, 0.62 against 0.60 ms; as two row
    // buffers in a loop unrolled by two after, 0.484 against 0.467 ms -- and issuing the ten gathers of the next half row
    // before the ten multiplies of the current one, 0.63 ms.  Seven to eight resident waves per SIMD cover those latencies.)
    switch (mode) {                                                     // wave-uniform
    case 0: return rtab_chain<KT, 0>(row, tb, step, rounds, my_rounds, bagb, K);
    case 1: return rtab_chain<KT, 1>(row, tb, step, rounds, my_rounds, bagb, K);
    case 2: return rtab_chain<KT, 2>(row, tb, step, rounds, my_rounds, bagb, K);
    case 3: return rtab_chain<KT, 3>(row, tb, step, rounds, my_rounds, bagb, K);
    case 4: return rtab_chain<KT, 4>(row, tb, step, rounds, my_rounds, bagb, K);
    default: return rtab_chain<KT, 5>(row, tb, step, rounds, my_rounds, bagb, K);
    }
}

template <int KT>
__global__ __launch_bounds__(64) void pool_rtab_kernel(PoolArgs a, RtabUse u)
{
    // one wavefront = one workgroup = one site (1, 2 and 4 sites per workgroup measure the same; with one, every LDS
    // address below is a compile-time offset): stage[8] leaf sums of a pass | tail[8] | merge stack | bag
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *stage = smem, *tail = smem + 8, *stack = smem + 16, *bag = smem + 16 + M6A_MEAN_STACK;
    // the byte-row gathers address the bag by its absolute LDS offset: this kernel has no static LDS, so the dynamic
    // block starts at 0 -- checked, not assumed
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)bag != RTAB_BAG_LDS) { if (threadIdx.x == 0) atomicExch(a.err, 4); return; }
    const int K = KT ? KT : a.K;
    const int lane = threadIdx.x;
    // blockIdx -> position in the bag-size order, XCD-aware: workgroups go round-robin over the 8 XCDs, so XCD x
    // walks the contiguous eighth x of the order and the tables of "its" bag sizes stay in its L2.  One site per
    // launch slot: persistent waves were tried and are slower (a fixed stride per wave 0.73 ms, a per-XCD
    // atomic work counter 1.58 ms, against 0.60 ms) -- the hardware dispatcher balances the workgroups better.
    const uint32_t chunk = gridDim.x >> 3;                 // gridDim.x is a multiple of 8
    const int64_t si = (int64_t)(blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (si >= (int64_t)u.si_count) return;
    m6a_clk_stamp(a.clk, 0);                               // profiled launches only (a.clk null otherwise)
    const int64_t s = (int64_t)(uint32_t)uni((int)u.order[(int64_t)u.si_base + si]);
    const int64_t r0 = uni64(a.off[s]);
    const int n = uni((int)(a.off[s + 1] - r0));
    const uint32_t rank = (uint32_t)uni((int)u.rank[s]);
    // first pass of the pairwise-sum plan: which leaf and chain this lane is (its loads overlap the bag's)
    const int b0 = lane >> 3;
    const bool has0 = b0 < a.n_leaves;
    const int ls0 = has0 ? a.leaf_start[b0] : 0;
    const int le0 = has0 ? a.leaf_start[b0 + 1] : 0;
    if (n <= 0) { if (lane == 0) { a.mod_ratio[s] = __builtin_nan(""); a.site_prob[s] = __builtin_nanf(""); } return; }
    // the bag: 1-p into LDS, four loads in flight per lane (unconditional: lanes past the end re-read the last
    // read); p >= thr counted on the way (mod_ratio, inference_utils.py:53)
    int cge = 0;
    for (int i0 = 0; i0 < n; i0 += 256) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { const int i = i0 + 64 * j + lane; v[j] = a.read_prob[r0 + (i < n ? i : n - 1)]; }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int i = i0 + 64 * j + lane;
            cge += (v[j] >= a.thr && i < n) ? 1 : 0;
            if (i < n) bag[i] = 1.0f - v[j];
        }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) cge += __shfl_xor(cge, m, 64);
    if (lane == 0) a.mod_ratio[s] = (double)cge / (double)n;
    if (rank == 0xffffffffu) return;                       // stream too short: the chain kernel raised the flag
    wave_fence();
    // the site's rows: byte rows for bags <= 256 reads, u16 rows above; bags of one read draw no words: slot 0 is a
    // table of zeros (every draw is read 0), read as byte rows at rank 0
    const bool bytes = n <= M6A_RTAB_U8_MAX_N;
    const int esz = bytes ? 1 : 2;
    const char *tb = (const char *)u.C + (n >= 2 ? ((int64_t)u.slot_of_n[n] * u.c_stride) * 2 + (int64_t)rank * esz : 0);
    const int mode = n < 2 ? 2 : bytes ? 2 + (int)(rank & 3u) : (int)(rank & 1u);      // wave-uniform alignment class
    const char *bagb = (const char *)bag;
    const int T = a.T;
    const int64_t row_bytes = (int64_t)K * esz;

    // the last leaf's n % 8 tail iterations first; their values wait in LDS
    if (a.n_rem) {
        const int t = lane < a.n_rem ? T - a.n_rem + lane : 0;
        const float v = rtab_chain_mode<KT>(mode, tb + t * row_bytes, tb, 0, 1, 1, bagb, K);
        if (lane < 8) tail[lane] = v;
    }
    int sp = 0;                                            // merge-stack height (lane 0's view)
    const int n_pass = (a.n_leaves + 7) >> 3;
    for (int ps = 0; ps < n_pass; ++ps) {
        const int b = 8 * ps + (lane >> 3);
        const bool has = b < a.n_leaves;
        const int ls = ps == 0 ? ls0 : has ? a.leaf_start[b] : 0;
        const int le = ps == 0 ? le0 : has ? a.leaf_start[b + 1] : 0;
        const int my_rounds = (le - ls) >> 3;
        int rounds = my_rounds;
#pragma unroll
        for (int m = 8; m < 64; m <<= 1) { const int o = __shfl_xor(rounds, m, 64); rounds = o > rounds ? o : rounds; }
        rounds = uni(rounds);
        // merges that follow each of the pass's leaves: 8 bytes of the plan, scalar loads (uniform address)
        const uint32_t mw0 = ((const uint32_t *)a.merge_after)[2 * ps], mw1 = ((const uint32_t *)a.merge_after)[2 * ps + 1];
        const float sum = rtab_chain_mode<KT>(mode, tb + (ls + (lane & 7)) * row_bytes, tb, 8 * row_bytes, rounds, my_rounds, bagb, K);
        const float lsum = chain8_sum_r(sum);
        if ((lane & 7) == 0) stage[lane >> 3] = lsum;
        wave_fence();
        if (lane == 0) {
            const int nl = a.n_leaves - 8 * ps < 8 ? a.n_leaves - 8 * ps : 8;
            for (int bl = 0; bl < nl; bl++) {
                float x = stage[bl];
                if (8 * ps + bl == a.n_leaves - 1)
                    for (int i = 0; i < a.n_rem; i++) x += tail[i];
                const uint32_t mw = bl < 4 ? mw0 : mw1;
                for (int m = (int)((mw >> (8 * (bl & 3))) & 0xffu); m > 0; --m) x = stack[--sp] + x;
                stack[sp++] = x;
            }
        }
        wave_fence();
    }
    if (lane == 0) a.site_prob[s] = stack[0] / (float)T;
    m6a_clk_stamp(a.clk, 1);
}

template __global__ void pool_rtab_kernel<20>(PoolArgs, RtabUse);
template __global__ void pool_rtab_kernel<0>(PoolArgs, RtabUse);

// An empty kernel per translation unit: HIP maps a code object on the first launch of any kernel in it (0.3-1.2 ms, measured
// in the first call's timeline, profiles/r03_first_call_timeline.txt); m6a_create's background set-up launches these instead.
__global__ void m6a_touch_pool_rtab() {}
