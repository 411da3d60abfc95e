// m6a_kernels.h -- kernel argument blocks shared by m6a_kernels.hip and m6a_api.hip.
#ifndef M6A_KERNELS_H
#define M6A_KERNELS_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#define M6A_BAG_LDS 1024          // reads of a bag kept in LDS by pool_scan_kernel (rest: global)
#define M6A_WFRAG_FLOATS (136 * 64)   // 40 (W1) + 80 (W2) + 16 (W3) registers x 64 lanes
#define M6A_WFRAG2_FLOATS (25 * 64)   // 20 (W1 x-slots) + 5 (W1[:,8]) registers x 64 lanes
#define M6A_W1E_FLOATS (35 * 32)
#define M6A_BN_FLOATS (5 * 2 * 16 * 2)  // (alpha, beta) of eval batch norm per hidden unit, [unit tile][lane half][register]
#define M6A_MEAN_STACK 32             // pairwise-sum merge stack (tree height + 1 fits for any T*K < 2^30)
#define M6A_CSITE_MIN_BAG 16          // enc_csite_kernel: a 32-read tile must span <= 3 sites
#define M6A_TABLE_MAX_N 32        // pool_table_kernel: byte offsets 8*idx must fit a byte; pool_reg_kernel: 32 register pairs
#define M6A_REG_STACK 8           // pool_reg_kernel: merge stack entries held in registers
#define M6A_RTAB_MAX_N 4096       // pool_rtab_kernel: bag sizes with an index table (LDS bag; u16 byte offsets reach 16 383 reads)
#define M6A_RTAB_SMALL_N 1024     // ... bags above this get their own launch: their LDS bag would cost every site its occupancy
#define M6A_RTAB_U8_MAX_N 256     // ... tables of bags up to this size hold index bytes, larger ones u16 byte offsets
#define M6A_HIST_BINS (M6A_RTAB_MAX_N + 2)   // bag-size histogram: n = 0..M6A_RTAB_MAX_N, last bin = larger

// weight of histogram bin i in the hash bag_verify_kernel compares (splitmix64 finaliser: NOT linear in i -- a linear weight
// would only re-check the total number of reads)
__host__ __device__ inline unsigned long long m6a_bin_weight(int i)
{
    unsigned long long z = (unsigned long long)(i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Clock stamps (bench.py's "clock the kernel ran at"): with a non-null `clk`, lane 0 of <= 64 workgroups
// spread over the grid (an odd stride: block b runs on XCD b % 8, so the samples rotate through the XCDs) stores s_memtime
// (shader cycles) and s_memrealtime (the constant 100 MHz clock) when the workgroup starts and when it ends: clk[slot][4] =
// {cycles0, real0, cycles1, real1}; (cycles1 - cycles0) / (real1 - real0) x 100 MHz is the shader clock THAT wave lived at.
// Null in every launch that is not being profiled (m6a_profile_enable): the product path never stamps.
#define M6A_CLK_SLOTS 64
__device__ __forceinline__ void m6a_clk_stamp(unsigned long long *clk, int which)
{
    if (!clk) return;
    const unsigned stride = (gridDim.x >> 6) | 1u, b = blockIdx.x;
    if (threadIdx.x == 0 && b % stride == 0 && b / stride < M6A_CLK_SLOTS) {
        unsigned long long *p = clk + (size_t)(b / stride) * 4 + which * 2;
        p[0] = __builtin_amdgcn_s_memtime();
        p[1] = __builtin_amdgcn_s_memrealtime();
    }
}

struct EncArgs {
    const float *X;               // [R][9]
    const uint8_t *site_kmers;    // [S][3]
    const int64_t *off;           // [S+1]
    const float *wfrag;           // [136][64] lane-major MFMA weight fragments (16-slot kernel; W2/W3 shared)
    const float *wfrag2;          // [25][64]  layer-1 fragments of the 12-slot kernel
    const float *w1e_tab;         // [35][32]  W1[:, 9..14] and b1 per unit, for the per-site c vectors
    const float *bn;              // [5][2][16][2] batch-norm (alpha, beta) pairs (M6A_BN_FLOATS)
    const float *emb;             // [66][2]
    float *read_prob;             // [R]
    int *err;
    int64_t n_sites, n_reads, n_tiles, tiles_per_wave;
    float b3;
    unsigned long long *clk;      // clock stamps of a profiled launch, else null (m6a_clk_stamp)
};

struct PoolArgs {
    const float *read_prob;       // [R]
    const int64_t *off;           // [S+1]
    const int64_t *goff;          // [G+1] flush-group site offsets
    const uint32_t *raw;          // MT19937 word stream (scan) ...
    const uint32_t *tab;          // ... or accepted-index table (table)
    float *site_prob;             // [S]
    double *mod_ratio;            // [S]
    uint32_t *start_pos;          // [S] first stream word of each site (scan)
    // NumPy pairwise-sum plan for the mean over T iterations (host-built, see m6a_api.hip):
    const int *leaf_start;        // [L+1] leaf b covers iterations [leaf_start[b], leaf_start[b+1])
    const uint8_t *merge_after;   // [L]   merges to perform after pushing leaf b (post-order)
    const uint32_t *row_meta;     // [rows][4] table kernel: flags | live mask lo | hi | merge nibbles
    int n_leaves, n_rows, n_rem;  // n_rem = iterations of the last leaf beyond a multiple of 8
    int stack_depth;              // deepest the merge stack gets (<= M6A_MEAN_STACK)
    // pool_reg_kernel: one control word per round of 8 iterations (bit 0 leaf ends, bits 8.. merges)
    const uint32_t *reg_ctl;
    int reg_rounds, reg_final_merges;
    int64_t reg_items;            // pool_reg_kernel's work items: ceil(n_groups / 256) * jmax
    int *err;
    int64_t n_groups, n_sites, raw_len;
    int T, K, uniform_n, jmax, bag_cap;
    float thr;
    unsigned long long *clk;      // clock stamps of a profiled launch, else null (m6a_clk_stamp)
};

// per-bag-size index tables (m6a_pool_rtab.hip): slot k holds C (accepted draws of the whole stream as u16 byte
// offsets 4*v) and RS (number of accepted words before every 64-word block, n_blk + 1 entries)
struct RtabBuild {
    const uint32_t *raw;
    uint32_t n_blk;               // stream blocks of 64 words covered by the tables
    const int32_t *build_n;       // [n_build] bag sizes to build
    const int32_t *build_slot;    // [n_build] their slots
    int n_build;
    uint16_t *C;
    uint32_t *RS;
    int64_t c_stride;             // entries per slot in C (= 64 * n_blk)
};
struct RtabUse {
    const uint16_t *C;
    const uint32_t *RS;
    const int32_t *slot_of_n;     // [M6A_RTAB_MAX_N + 1]
    uint32_t *rank;               // [S] rank of each site's first draw in its C row (0xffffffff: stream too short)
    const uint32_t *order;        // [S] sites ordered by bag size
    int64_t c_stride;
    uint32_t n_blk;
    int bag_cap;                  // floats of LDS bag per wavefront
    uint32_t si_base, si_count;   // this launch takes positions [si_base, si_base + si_count) of `order`
};

__global__ void enc_kernel(EncArgs a);
__global__ void enc_site16_kernel(EncArgs a);
__global__ void enc_csite_kernel(EncArgs a);
__global__ void pool_scan_start_kernel(PoolArgs a);
template <int KT> __global__ void pool_scan_group_kernel(PoolArgs a);
template <int KT> __global__ void pool_scan_site_kernel(PoolArgs a);
__global__ void pool_table_kernel(PoolArgs a);
__global__ void pool_reg_kernel(PoolArgs a);
__global__ void bag_noisy_or_kernel(const float *read_prob, int64_t n_bags, int bag, float *site_prob);
__global__ void iota_off_kernel(int64_t *off, int64_t n_plus_1, int64_t step);
__global__ void rebase_off_kernel(const int64_t *off, int64_t count, int64_t *out);
__global__ void sampled_noisy_or_kernel(const float *read_prob, const int32_t *gidx, int64_t n_bags, int k, float *y);
__global__ void mean_over_passes_kernel(const float *y, int n_iters, int64_t n_sites, float *avg);
__global__ void bag_minmax_kernel(const int64_t *off, int64_t n_sites, unsigned long long *out, uint32_t *hist);
__global__ void bag_verify_kernel(const unsigned long long *got, const uint32_t *hist, unsigned long long mn, unsigned long long mx,
                                  unsigned long long reads, unsigned long long hash, int *err);
__global__ void mt19937_kernel(uint32_t seed, int64_t n_words, uint32_t *raw, const uint32_t *hist, int64_t seg_words, uint32_t *xhead, int64_t xhead_n);
__global__ void mt_jump_kernel(const uint32_t *xhead, const uint64_t *polys, uint32_t *hist);
__global__ void rtab_count_kernel(RtabBuild a);
__global__ void rtab_scan_kernel(RtabBuild a);
__global__ void rtab_fill_kernel(RtabBuild a);
__global__ void rtab_to_reg_table_kernel(const uint16_t *C, int64_t A, int64_t row_words, int jmax, uint16_t *tab);
__global__ void rtab_to_lds_table_kernel(const uint16_t *C, int64_t A, int K, int rows, int jmax, const int *iter_of, uint32_t *tab);
__global__ void rtab_prep_kernel(PoolArgs a, RtabUse u, uint32_t *cursor, uint32_t *order, unsigned n_order_blocks);
template <int KT> __global__ void pool_rtab_kernel(PoolArgs a, RtabUse u);

__global__ void m6a_touch_kernels();
__global__ void m6a_touch_pool_reg();
__global__ void m6a_touch_pool_rtab();

#endif
