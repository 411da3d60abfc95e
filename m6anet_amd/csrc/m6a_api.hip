// m6a_api.hip -- host side of libm6a_hip.so: the C ABI of include/m6a.h.
// Context/weights management, MT19937 stream + index-table preparation, launches.
#include "m6a_ctx.h"

// assets/mt19937_jump.bin inside the library (host pass only): the drop-in is ONE shared object, no file look-ups at run time
#if !defined(__HIP_DEVICE_COMPILE__)
#ifndef M6A_MT_JUMP_PATH
#error "build with -DM6A_MT_JUMP_PATH=\"<repo>/m6anet_amd/assets/mt19937_jump.bin\" (m6anet_amd/build.py does)"
#endif
asm(".section .rodata\n"
    ".global m6a_mt_jump_blob\n.global m6a_mt_jump_blob_end\n.hidden m6a_mt_jump_blob\n.hidden m6a_mt_jump_blob_end\n.balign 64\n"
    "m6a_mt_jump_blob:\n.incbin \"" M6A_MT_JUMP_PATH "\"\n"
    "m6a_mt_jump_blob_end:\n.byte 0\n.previous\n");
#endif

namespace {

thread_local std::string g_create_error;

// Contexts whose background set-up thread may still be running.  A process that exits without m6a_destroy (a Python
// exception past the engine, say) would otherwise tear the HIP runtime down under a thread that is launching kernels:
// the handler is registered after the first m6a_create has initialised HIP, so it runs BEFORE HIP's own exit handlers.
std::mutex g_live_mu;
std::vector<m6a_ctx *> g_live;
void join_background_setups()
{
    std::lock_guard<std::mutex> g(g_live_mu);
    for (m6a_ctx *c : g_live)
        if (c->warm.joinable()) c->warm.join();
    g_live.clear();
}

}  // namespace

using namespace m6a_detail;

namespace m6a_detail {


int fail(m6a_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    // the error text is shared with m6a_create's background set-up (it may fail() too and clears the text when it ends):
    // an entry point that has not waited for it yet -- the setters do not, so that a caller can configure the context while
    // the set-up runs -- waits here, on its error path only
    if (c && c->warm.joinable() && std::this_thread::get_id() != c->warm.get_id()) c->warm.join();
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}


bool is_device_ptr(const void *p)
{
    if (!p) return false;
    hipPointerAttribute_t at;
    hipError_t e = hipPointerGetAttributes(&at, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
}

// ---- weights: fold eval-mode BatchNorm into layer 1, lay out MFMA fragments ------------------
// blob offsets (floats), see include/m6a.h
enum { O_E = 0, O_W1 = 132, O_B1 = 2382, O_G = 2532, O_BE = 2682, O_MU = 2832, O_VAR = 2982,
       O_W2 = 3132, O_B2 = 7932, O_W3 = 7964, O_B3 = 7996 };

void build_fragments(const float *w, std::vector<float> &frag, std::vector<float> &frag2, std::vector<float> &w1e)
{
    // The encoder follows the ORDER OF OPERATIONS of the reference's float32 arithmetic, pinned against torch's own
    // intermediate tensors (tools/emulate_encoder.py; DESIGN.md section 2 "order of operations"):
    //   Linear 15 -> 150 : acc = 0; acc = fma(x[k], W1[u][k], acc) for k = 0..14; then acc + b1[u]
    //   BatchNorm (eval) : alpha = gamma / sqrt(var + eps) as float32, beta = fma(-mean, alpha, bias); fma(y, alpha, beta)
    //   Linear 150 -> 32 : acc = 0; acc = fma(h[k], W2[o][k], acc) for k = 0..149; then acc + b2[o]
    // A v_mfma_f32_32x32x2_f32 is two such fmas per output (k = the half-0 lane's operand first), so it is all in WHICH
    // weight sits in which fragment -- plus the batch norm as one v_fma_f32 per hidden unit in the kernel (it was folded
    // into W1 until round 4: one rounding fewer than the reference, and 2x its distance from the reference's values).
    //
    // W1aug[160][16]: columns 0..14 = W1, column 15 = b1 (the constant-one feature adds it last); row 150 = the
    // constant-one unit that carries b2 into layer 2 (its batch norm is alpha 1, beta 0); rows 151..159 = 0.
    std::vector<float> w1(160 * 16, 0.f), w2(32 * 160, 0.f), alpha(160, 0.f), beta(160, 0.f);
    for (int j = 0; j < 150; j++) {
        for (int k = 0; k < 15; k++) w1[j * 16 + k] = w[O_W1 + 15 * j + k];
        w1[j * 16 + 15] = w[O_B1 + j];
        const float invstd = 1.0f / std::sqrt(w[O_VAR + j] + 1e-5f);
        alpha[j] = w[O_G + j] * invstd;
        beta[j] = std::fmaf(-w[O_MU + j], alpha[j], w[O_BE + j]);
    }
    w1[150 * 16 + 15] = 1.0f;
    alpha[150] = 1.0f;
    for (int o = 0; o < 32; o++) {
        for (int k = 0; k < 150; k++) w2[o * 160 + k] = w[O_W2 + 150 * o + k];
        w2[o * 160 + 150] = w[O_B2 + o];
    }
    // Layer 1's ReLU is the clamp modifier of the batch-norm fma on alpha, beta scaled by 2^-64 (m6a_kernels.hip, bn_relu):
    // layer 2's weights carry the 2^64 back.  Layer 2's ReLU is x + |x| = 2 * relu(x) (one full-rate v_add_f32; max is half
    // rate): the 0.5 rides in W3.  Powers of two: every product the MFMAs and the epilogue form is the one relu(h) would give.
    const float bn_scale = 0x1p-64f;
    for (float &v : alpha) v *= bn_scale;
    for (float &v : beta) v *= bn_scale;
    for (float &v : w2) v *= 0x1p+64f;
    const float w3_scale = 0.5f;
    // Row i of a 32x32 accumulator tile lives in register (i&3) + 4(i>>3) of lane half (i>>2)&1.  Layer 1's row i of unit
    // tile m is hidden unit 32m + 2q + half, so that layer 2's MFMA q of that tile -- whose two k operands are exactly
    // register q of the two halves -- adds units 32m + 2q and 32m + 2q + 1, in that order: k = 0, 1, 2, ... 149, then the
    // constant-one unit 150 (b2, added last as the reference adds it) and the zero unit 151.
    auto unit_of_row = [](int m, int i) { return 32 * m + 2 * ((i & 3) + 4 * (i >> 3)) + ((i >> 2) & 1); };
    // Layer 2's OUTPUT units on the accumulator rows: register q of lane half h holds output unit out_unit(q, h) -- the
    // even lanes of the reference's two 16-lane gemv vectors in half 0, the odd lanes and k = 0 in half 1 (m6a_kernels.hip,
    // gemv32_as_mkl); the 12-slot kernel sums its half's 16 registers in register order, whatever units they are.
    auto out_unit = [](int q, int half) {
        if (half == 0) return q < 8 ? 2 * q + 1 : 2 * (q - 8) + 17;
        return q < 8 ? 2 * q + 2 : (q < 15 ? 2 * (q - 8) + 18 : 0);
    };
    auto out_unit_of_row = [&](int i) { return out_unit((i & 3) + 4 * (i >> 3), (i >> 2) & 1); };
    frag.assign(M6A_WFRAG_FLOATS, 0.f);
    for (int lane = 0; lane < 64; lane++) {
        const int col = lane & 31, half = lane >> 5;
        for (int m = 0; m < 5; m++) {
            for (int st = 0; st < 8; st++)      // A[i=col][k=half] of step st <-> feature 2st + half
                frag[(m * 8 + st) * 64 + lane] = w1[unit_of_row(m, col) * 16 + 2 * st + half];
            for (int q = 0; q < 16; q++)        // A[o=col][k=half] of step q <-> hidden unit 32m + 2q + half
                frag[(40 + m * 16 + q) * 64 + lane] = w2[out_unit_of_row(col) * 160 + 32 * m + 2 * q + half];
        }
        for (int q = 0; q < 16; q++)
            frag[(120 + q) * 64 + lane] = w3_scale * w[O_W3 + out_unit(q, half)];
    }
    // 12-slot kernel: x-slot fragments W1[u][2st+half] (st<4), W1[u][8]; and the per-unit rows the per-site c vectors are
    // formed from: W1[u][9..14], b1[u]
    frag2.assign(M6A_WFRAG2_FLOATS, 0.f);
    w1e.assign(M6A_W1E_FLOATS + M6A_BN_FLOATS, 0.f);
    for (int lane = 0; lane < 64; lane++) {
        const int col = lane & 31, half = lane >> 5;
        for (int m = 0; m < 5; m++) {
            for (int st = 0; st < 4; st++) frag2[(m * 4 + st) * 64 + lane] = w1[unit_of_row(m, col) * 16 + 2 * st + half];
            frag2[(20 + m) * 64 + lane] = w1[unit_of_row(m, col) * 16 + 8];
        }
    }
    for (int m = 0; m < 5; m++)
        for (int col = 0; col < 32; col++) {
            for (int e = 0; e < 6; e++) w1e[(m * 7 + e) * 32 + col] = w1[unit_of_row(m, col) * 16 + 9 + e];
            w1e[(m * 7 + 6) * 32 + col] = w1[unit_of_row(m, col) * 16 + 15];
        }
    // batch-norm pairs behind the w1e rows: [m][half][q] -> (alpha, beta) of unit 32m + 2q + half, so that a lane reads the
    // pairs of registers q, q+1 with one 16-byte LDS load
    for (int m = 0; m < 5; m++)
        for (int half = 0; half < 2; half++)
            for (int q = 0; q < 16; q++) {
                const int u = 32 * m + 2 * q + half;
                w1e[M6A_W1E_FLOATS + ((m * 2 + half) * 16 + q) * 2] = alpha[u];
                w1e[M6A_W1E_FLOATS + ((m * 2 + half) * 16 + q) * 2 + 1] = beta[u];
            }
}

// ---- flush groups (inference_utils.py:33,47) --------------------------------------------------
// `base` = index of the first site within the whole job (a multiple of bs that starts a group):
// batch indices -- and with them the flush pattern -- are global, offsets returned are local.
int64_t flush_groups(int64_t S, int64_t bs, int64_t spb, int64_t base, std::vector<int64_t> &goff)
{
    goff.clear();
    goff.push_back(0);
    if (S <= 0) return 0;
    const int64_t it0 = base / bs;
    const int64_t nb = (S + bs - 1) / bs;
    int64_t start_b = 0;
    for (int64_t b = 0; b < nb; b++) {
        if ((it0 + b + 1) % spb) {
            goff.push_back(std::min((b + 1) * bs, S));
            start_b = b + 1;
        }
    }
    if (start_b < nb) goff.push_back(S);
    return (int64_t)goff.size() - 1;
}

bool base_is_group_start(int64_t base, int64_t bs, int64_t spb)
{
    if (base < 0 || base % bs) return false;
    const int64_t it0 = base / bs;
    return it0 == 0 || (it0 % spb) != 0;     // batch it0-1 closed a group
}

int ensure_groups(m6a_ctx *c, int64_t S, int64_t bs, int64_t spb)
{
    const int64_t base = c->job_offset;
    if (c->goff_key.valid && c->goff_key.S == S && c->goff_key.bs == bs && c->goff_key.spb == spb &&
        c->goff_key.base == base) return M6A_OK;
    if (!base_is_group_start(base, bs, spb))
        return fail(c, M6A_EINVAL, "job offset %lld does not start a flush group for batch_size=%lld save_per_batch=%lld",
                    (long long)base, (long long)bs, (long long)spb);
    std::vector<int64_t> g;
    const int64_t G = flush_groups(S, bs, spb, base, g);
    int64_t gmax = 0;
    for (int64_t i = 0; i < G; i++) gmax = std::max(gmax, g[i + 1] - g[i]);
    HIPCHK(c, c->goff.ensure(g.size() * sizeof(int64_t)));
    HIPCHK(c, hipMemcpyAsync(c->goff.p, g.data(), g.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));   // g goes out of scope
    c->goff_key = {S, bs, spb, base, G, gmax, true};
    return M6A_OK;
}

// ---- NumPy legacy stream: np.random.seed(int) == init_genrand ------------------------------------------
// generated on the device (mt19937_kernel, m6a_pool_rtab.hip).  The recurrence is a chain -- 623 words per dependent step on
// one workgroup -- so streams are cut into up to 32 segments that run on 32 CUs at once: the head of the stream is generated
// first (35 steps), mt_jump_kernel derives every other segment's starting history from it through precomputed GF(2) jump
// polynomials (assets/mt19937_jump.bin, embedded below; tools/make_mt_jump.py), then all segments run.  Segment lengths come
// in three sizes (2^16, 2^20, 2^24 words); a stream beyond 32 x 2^24 words lets its last segment run on.
constexpr int kJumpPerRegime = 31, kJumpPolyWords = 312;
constexpr int64_t kXheadWords = 22016;                     // untempered x[0 .. ) the jump reads (58 + 1087 + 19967 < 22016)

extern "C" const unsigned char m6a_mt_jump_blob[];
extern "C" const unsigned char m6a_mt_jump_blob_end[];

struct JumpRegime { int64_t G; const uint64_t *polys; };   // polys: host pointer into the blob, [31][312]

// the blob's regimes, smallest segment first (nullptr if the blob is not what this build expects)
const JumpRegime *jump_regimes(int *n_out)
{
    static JumpRegime reg[8];
    static int n = 0;
    static std::once_flag once;                            // contexts may be created (and warmed) from several threads at once
    std::call_once(once, [] {
        const unsigned char *b = m6a_mt_jump_blob;
        const size_t len = (size_t)(m6a_mt_jump_blob_end - m6a_mt_jump_blob);
        uint32_t hdr[4];
        if (len >= 24 && std::memcmp(b, "M6AMTJP1", 8) == 0) {
            std::memcpy(hdr, b + 8, 16);
            const size_t per = 8 + (size_t)hdr[1] * hdr[2] * 8;
            if (hdr[0] <= 8 && hdr[1] == (uint32_t)kJumpPerRegime && hdr[2] == (uint32_t)kJumpPolyWords && hdr[3] == 512 &&
                len == 24 + hdr[0] * per)
                for (uint32_t r = 0; r < hdr[0]; r++) {
                    const unsigned char *q = b + 24 + r * per;
                    int64_t G;
                    std::memcpy(&G, q, 8);
                    reg[n++] = {G, (const uint64_t *)(q + 8)};
                }
        }
    });
    *n_out = n;
    return reg;
}

int launch_stream(m6a_ctx *c, uint32_t seed, int64_t len, uint32_t *raw)
{
    int n_reg = 0;
    const JumpRegime *reg = jump_regimes(&n_reg);
    const JumpRegime *use = nullptr;
    // worth it from two segments of the smallest size on (M6A_MT_SEGMENTS=0: always the single chain)
    static const bool allowed = !(getenv("M6A_MT_SEGMENTS") && getenv("M6A_MT_SEGMENTS")[0] == '0');
    if (allowed && n_reg > 0 && len > reg[0].G + kXheadWords) {
        use = &reg[n_reg - 1];
        for (int r = 0; r < n_reg; r++) if (len <= (int64_t)(kJumpPerRegime + 1) * reg[r].G) { use = &reg[r]; break; }
    }
    if (!use) {
        hipLaunchKernelGGL(mt19937_kernel, dim3(1), dim3(640), 0, c->stream, seed, len, raw, (const uint32_t *)nullptr, len,
                           (uint32_t *)nullptr, (int64_t)0);
        HIPCHK(c, hipGetLastError());
        return M6A_OK;
    }
    const int n_seg = (int)std::min<int64_t>((len + use->G - 1) / use->G, kJumpPerRegime + 1);
    const size_t poly_bytes = (size_t)kJumpPerRegime * kJumpPolyWords * 8;
    HIPCHK(c, c->mt_scratch.ensure((size_t)kXheadWords * 4 + (size_t)kJumpPerRegime * 1078 * 4 + poly_bytes));
    uint32_t *xhead = (uint32_t *)c->mt_scratch.p, *hist = xhead + kXheadWords;
    uint64_t *d_polys = (uint64_t *)(hist + (size_t)kJumpPerRegime * 1078);
    if (c->mt_polys_G != use->G) {                           // the blob is static host memory: the copy may complete whenever it likes
        HIPCHK(c, hipMemcpyAsync(d_polys, use->polys, poly_bytes, hipMemcpyHostToDevice, c->stream));
        c->mt_polys_G = use->G;
    }
    hipLaunchKernelGGL(mt19937_kernel, dim3(1), dim3(640), 0, c->stream, seed, kXheadWords - 624, raw, (const uint32_t *)nullptr,
                       kXheadWords, xhead, kXheadWords);
    hipLaunchKernelGGL(mt_jump_kernel, dim3(17, (unsigned)(n_seg - 1)), dim3(256), 0, c->stream, (const uint32_t *)xhead,
                       (const uint64_t *)d_polys, hist);
    hipLaunchKernelGGL(mt19937_kernel, dim3((unsigned)n_seg), dim3(640), 0, c->stream, seed, len, raw, (const uint32_t *)hist, use->G,
                       (uint32_t *)nullptr, (int64_t)0);
    HIPCHK(c, hipGetLastError());
    return M6A_OK;
}

int ensure_raw(m6a_ctx *c, uint32_t seed, int64_t len)
{
    if (c->raw_len >= len && c->raw_seed == seed) return M6A_OK;
    if (len > ((int64_t)1 << 31) - 512)
        return fail(c, M6A_ESTREAM, "a flush group would need %lld MT19937 words (cap 2^31); "
                    "reduce batch_size*save_per_batch or num_iterations", (long long)len);
    len = (len + 1023) / 1024 * 1024;
    c->raw_len = 0;
    c->rt.valid = false;                       // the index tables describe the old stream
    HIPCHK(c, c->raw.ensure((size_t)len * 4));
    int rc = launch_stream(c, seed, len, (uint32_t *)c->raw.p);
    if (rc) return rc;
    c->raw_seed = seed; c->raw_len = len;
    return M6A_OK;
}

// ---- NumPy's float32 pairwise sum, as a plan ------------------------------------------------------
// ndarray.mean() of the T per-iteration values is add.reduce's pairwise summation
// (numpy/core/src/umath/loops_utils.h.src): n <= 128 -> a LEAF: 8 interleaved accumulators
// r[k] += a[8i+k] over the first n - n%8 elements, combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)),
// then the n%8 tail added one by one (n < 8: everything is tail); n > 128 -> split at
// n2 = n/2 - (n/2)%8 and add the two halves.  The kernels make lanes the accumulator chains and
// replay the tree as "push leaf sum, then merge_after[leaf] times: pop two, push their sum".
void pairwise_rec(int lo, int n, MeanPlan &p)
{
    if (n <= 128) {
        p.leaf_start.push_back(lo);
        p.merge_after.push_back(0);
        return;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    pairwise_rec(lo, n2, p);
    pairwise_rec(lo + n2, n - n2, p);
    p.merge_after.back()++;             // the MERGE event follows the right subtree's last leaf
}

// iteration handled by (row, lane) of the table kernel, or -1: lane = accumulator chain (lane & 7) of
// leaf 8*pass + lane/8, row = round of the pass; the tail row holds the last leaf's n % 8 elements
int plan_iteration(const MeanPlan &p, int row, int lane)
{
    const int L = (int)p.merge_after.size();
    const int ps = p.row_pass[row], i = p.row_round[row];
    if (ps < 0) return lane < p.n_rem ? p.T - p.n_rem + lane : -1;
    const int b = 8 * ps + (lane >> 3);
    if (b >= L) return -1;
    const int len = p.leaf_start[b + 1] - p.leaf_start[b];
    return i < len / 8 ? p.leaf_start[b] + 8 * i + (lane & 7) : -1;
}

void build_mean_plan(int T, MeanPlan &p)
{
    p = MeanPlan();
    p.T = T;
    pairwise_rec(0, T, p);
    p.leaf_start.push_back(T);
    const int L = (int)p.merge_after.size();
    const int last_len = p.leaf_start[L] - p.leaf_start[L - 1];
    p.n_rem = last_len % 8;             // only the rightmost leaf can have a tail (every n2 is a multiple of 8)
    // table-kernel rows: the tail row first (its values wait in LDS for the last leaf), then per pass of 8
    // leaves as many rounds as its longest chain; the last row of a pass carries the flush
    if (p.n_rem) { p.row_pass.push_back(-1); p.row_round.push_back(0); }
    const int P = (L + 7) / 8;
    for (int ps = 0; ps < P; ps++) {
        int rounds = 0;
        for (int b = 8 * ps; b < std::min(L, 8 * ps + 8); b++)
            rounds = std::max(rounds, (p.leaf_start[b + 1] - p.leaf_start[b]) / 8);
        rounds = std::max(rounds, 1);   // T < 8: no chains at all, the row only carries the flush
        for (int i = 0; i < rounds; i++) { p.row_pass.push_back(ps); p.row_round.push_back(i); }
    }
    int d = 0;
    for (int b = 0; b < L; b++) {
        d++; p.depth = std::max(p.depth, d); d -= p.merge_after[b];
        p.max_merge = std::max<int>(p.max_merge, p.merge_after[b]);
    }
    // register kernel: iterations in order, a leaf is a whole number of rounds; the last leaf is finished
    // after the loop (its tail and merges), so it carries no flag
    p.reg_ctl.assign((size_t)std::max(1, (T - p.n_rem) / 8), 0u);
    for (int b = 0; b + 1 < L; b++) p.reg_ctl[(size_t)p.leaf_start[b + 1] / 8 - 1] = 1u | ((uint32_t)p.merge_after[b] << 8);
    const int rows = (int)p.row_pass.size();
    p.row_meta.assign((size_t)rows * 4, 0u);
    for (int r = 0; r < rows; r++) {
        uint32_t *m = &p.row_meta[(size_t)r * 4];
        uint64_t live = 0;
        for (int l = 0; l < 64; l++) live |= (uint64_t)(plan_iteration(p, r, l) >= 0) << l;
        m[1] = (uint32_t)live; m[2] = (uint32_t)(live >> 32);
        const int ps = p.row_pass[r];
        if (ps < 0) { m[0] = 1u << 25; continue; }
        if (r + 1 == rows || p.row_pass[r + 1] != ps) {
            const int nl = std::min(8, L - 8 * ps);
            m[0] = (1u << 24) | ((uint32_t)nl << 26) | (ps == P - 1 ? 1u << 30 : 0u);
            for (int bl = 0; bl < nl; bl++) m[3] |= (uint32_t)(p.merge_after[8 * ps + bl] & 15) << (4 * bl);
        }
    }
}

int ensure_mean_plan(m6a_ctx *c, int T)
{
    if (c->plan.T == T && c->plan_dev.p) return M6A_OK;
    build_mean_plan(T, c->plan);
    const MeanPlan &p = c->plan;
    if (p.depth > M6A_MEAN_STACK) return fail(c, M6A_EUNSUPPORTED, "n_iters %d: pairwise-sum tree deeper than %d", T, M6A_MEAN_STACK);
    const size_t b0 = p.leaf_start.size() * 4, b1 = (p.merge_after.size() + 3) / 4 * 4, b2 = std::max<size_t>(p.row_meta.size(), 1) * 4;
    const size_t b3 = p.reg_ctl.size() * 4;
    HIPCHK(c, c->plan_dev.ensure(b0 + b1 + b2 + b3));
    char *d = (char *)c->plan_dev.p;
    HIPCHK(c, hipMemcpyAsync(d, p.leaf_start.data(), b0, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d + b0, p.merge_after.data(), p.merge_after.size(), hipMemcpyHostToDevice, c->stream));
    if (!p.row_meta.empty())
        HIPCHK(c, hipMemcpyAsync(d + b0 + b1, p.row_meta.data(), p.row_meta.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d + b0 + b1 + b2, p.reg_ctl.data(), b3, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->plan_off[0] = 0; c->plan_off[1] = b0; c->plan_off[2] = b0 + b1; c->plan_off[3] = b0 + b1 + b2;
    c->tab_key.valid = false;           // the table layout follows the plan
    return M6A_OK;
}

void plan_args(m6a_ctx *c, PoolArgs &a)
{
    const char *d = (const char *)c->plan_dev.p;
    a.leaf_start = (const int *)(d + c->plan_off[0]);
    a.merge_after = (const uint8_t *)(d + c->plan_off[1]);
    a.row_meta = (const uint32_t *)(d + c->plan_off[2]);
    a.n_leaves = (int)c->plan.merge_after.size();
    a.n_rows = (int)c->plan.row_pass.size();
    a.n_rem = c->plan.n_rem;
    a.stack_depth = c->plan.depth;
    a.reg_ctl = (const uint32_t *)(d + c->plan_off[3]);
    a.reg_rounds = (a.T - c->plan.n_rem) / 8;
    a.reg_final_merges = c->plan.merge_after.back();
}

// ---- per-bag-size index tables (m6a_pool_rtab.hip) ------------------------------------------------
// words of stream a flush group of gmax sites can consume: expected <= 2 per accepted draw, slack for the spread
int64_t stream_need(int64_t gmax, int T, int K)
{
    const int64_t A = (int64_t)T * K;
    return gmax * (2 * A + A / 16) + 8192;        // incl. the scan kernels' 2 x 1024-word read-ahead
}

uint32_t *ctl_cursor(m6a_ctx *c) { return c->h_ctl; }
int32_t *ctl_slot(m6a_ctx *c) { return (int32_t *)(c->h_ctl + M6A_HIST_BINS); }
int32_t *ctl_build_n(m6a_ctx *c) { return ctl_slot(c) + M6A_RTAB_MAX_N + 1; }
int32_t *ctl_build_slot(m6a_ctx *c) { return ctl_build_n(c) + M6A_RTAB_MAX_N; }
constexpr size_t kCtlWords = M6A_HIST_BINS + (M6A_RTAB_MAX_N + 1) + 2 * M6A_RTAB_MAX_N;

// bag sizes (2..M6A_RTAB_MAX_N) of `hist` that have no table yet for (seed, T*K, a stream of >= need words)
int rtab_missing(const m6a_ctx *c, uint32_t seed, int T, int K, int64_t need, const uint32_t *hist, int *n_distinct)
{
    const bool valid = c->rt.valid && c->rt.seed == seed && c->rt.A == (int64_t)T * K && c->raw_seed == seed &&
                       c->rt.n_blk * 64 >= need && c->rt.n_blk == c->raw_len / 64;
    int missing = 0, distinct = 0;
    for (int n = 2; n <= M6A_RTAB_MAX_N; n++)
        if (hist[n]) { distinct++; if (!valid || c->rt.slot_of_n[n] < 0) missing++; }
    if (n_distinct) *n_distinct = distinct;
    return missing;
}

// Builds the tables of every bag size in `hist` that lacks one.  *usable = false (and M6A_OK) when the tables would
// not fit the memory budget: the caller then takes the scan kernels.
int ensure_rtab(m6a_ctx *c, uint32_t seed, int T, int K, int64_t gmax, const uint32_t *hist, bool *usable)
{
    *usable = false;
    const int64_t need = stream_need(gmax, T, K);
    int rc = ensure_raw(c, seed, need);
    if (rc) return rc;
    auto &rt = c->rt;
    const int64_t n_blk = c->raw_len / 64;
    if (!(rt.valid && rt.seed == seed && rt.A == (int64_t)T * K && rt.n_blk == n_blk)) {
        rt.valid = false; rt.used = 0;
        for (int n = 0; n <= M6A_RTAB_MAX_N; n++) rt.slot_of_n[n] = -1;
    }
    std::vector<int> todo;
    for (int n = 2; n <= M6A_RTAB_MAX_N; n++)
        if (hist[n] && rt.slot_of_n[n] < 0) todo.push_back(n);
    const int64_t c_stride = n_blk * 64;
    const size_t slot_bytes = (size_t)c_stride * 2 + (size_t)(n_blk + 1) * 4;
    const bool fresh = !rt.valid || rt.n_blk != n_blk;
    const int want = (fresh ? 1 : rt.used) + (int)todo.size();
    if (fresh || want > rt.cap) {
        int new_cap = std::max(want, fresh ? 0 : rt.cap * 2);
        new_cap = std::min(std::max(new_cap, std::max(32, c->rt_presize)), M6A_RTAB_MAX_N + 1);
        size_t free_b = 0, total_b = 0;
        HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
        const size_t held = fresh ? 0 : (size_t)rt.cap * slot_bytes;
        const size_t budget = std::min<size_t>((size_t)16 << 30, (free_b + held) / 2);
        if ((size_t)new_cap * slot_bytes > budget) new_cap = want;
        if ((size_t)new_cap * slot_bytes > budget) return M6A_OK;          // not usable: scan kernels instead
        uint16_t *nC = nullptr; uint32_t *nRS = nullptr;
        if (fresh) {                                                        // old tables are void: free first
            if (rt.C) (void)hipFree(rt.C);
            if (rt.RS) (void)hipFree(rt.RS);
            rt.C = nullptr; rt.RS = nullptr; rt.cap = 0;
        }
        // an allocation that fails is not an error of the call: the scan kernels need no tables
        // + 64 bytes: a site of odd rank reads its rows as 11 aligned dwords, one halfword past the row on either side
        if (hipMalloc((void **)&nC, (size_t)new_cap * c_stride * 2 + 64) != hipSuccess) { (void)hipGetLastError(); return M6A_OK; }
        if (hipMalloc((void **)&nRS, (size_t)new_cap * (n_blk + 1) * 4) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(nC); return M6A_OK; }
        if (!fresh && rt.used) {
            HIPCHK(c, hipMemcpyAsync(nC, rt.C, (size_t)rt.used * c_stride * 2, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(nRS, rt.RS, (size_t)rt.used * (n_blk + 1) * 4, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            // hipFree waits for the WHOLE device: behind m6a_infer's running encoder that was 3.3 ms of the first ragged call
            // (profiles/r04_first_call_ragged.txt), and it kept the build kernels below from being queued.  The outgrown arena
            // is dropped at the next point where the context is idle anyway (m6a_sync / m6a_destroy).
            c->graveyard.push_back(rt.C); c->graveyard.push_back(rt.RS);
        }
        rt.C = nC; rt.RS = nRS; rt.cap = new_cap;
        if (fresh) {
            // slot 0: a table of zeros for bags of one read (randint(0,1) draws no words, every draw is read 0)
            HIPCHK(c, hipMemsetAsync(rt.C, 0, (size_t)c_stride * 2, c->stream));
            HIPCHK(c, hipMemsetAsync(rt.RS, 0, (size_t)(n_blk + 1) * 4, c->stream));
            rt.used = 1;
        }
        rt.valid = true; rt.seed = seed; rt.A = (int64_t)T * K; rt.n_blk = n_blk;
    }
    if (!todo.empty()) {
        int32_t *bn = ctl_build_n(c), *bs = ctl_build_slot(c);
        for (size_t i = 0; i < todo.size(); i++) { bn[i] = todo[i]; bs[i] = rt.used; rt.slot_of_n[todo[i]] = rt.used++; }
        HIPCHK(c, c->ctl_dev.ensure(kCtlWords * 4));
        int32_t *d_bn = (int32_t *)c->ctl_dev.p + (ctl_build_n(c) - (int32_t *)c->h_ctl);
        HIPCHK(c, hipMemcpyAsync(d_bn, bn, (size_t)2 * M6A_RTAB_MAX_N * 4, hipMemcpyHostToDevice, c->stream));
        RtabBuild b;
        b.raw = (const uint32_t *)c->raw.p; b.n_blk = (uint32_t)n_blk; b.build_n = d_bn; b.build_slot = d_bn + M6A_RTAB_MAX_N;
        b.C = rt.C; b.RS = rt.RS; b.c_stride = c_stride; b.n_build = (int)todo.size();
        const unsigned gx = (unsigned)std::min<int64_t>((n_blk + 63) / 64, 65535);   // 4 waves x 16 blocks per workgroup
        const unsigned gn = (unsigned)((todo.size() + 7) / 8);                        // RTAB_GROUP = 8 bag sizes share every stream read
        hipLaunchKernelGGL(rtab_count_kernel, dim3(gn, gx), dim3(256), 0, c->stream, b);
        hipLaunchKernelGGL(rtab_scan_kernel, dim3((unsigned)todo.size()), dim3(256), 0, c->stream, b);
        hipLaunchKernelGGL(rtab_fill_kernel, dim3(gn, gx), dim3(256), 0, c->stream, b);
        HIPCHK(c, hipGetLastError());
        // the pinned build list is reused by the next call: it must have been consumed
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    *usable = true;
    return M6A_OK;
}

// Uniform bags of n reads: every flush group consumes the stream identically, so "site j of a group" uses draws
// [j*T*K, (j+1)*T*K) of the accepted sequence C_n -- a slice of the GPU-built index table of bag size n (both uniform
// pooling kernels cut their tables out of it; nothing on an inference path runs a host generator).  Builds C_n if it is
// missing and checks that the stream holds jmax sites' worth of accepted draws.  *C = nullptr for n = 1 (no draws).
int uniform_slice(m6a_ctx *c, uint32_t seed, int n, int T, int K, int jmax, const uint16_t **C)
{
    *C = nullptr;
    if (n < 2) return M6A_OK;
    std::vector<uint32_t> hist(M6A_HIST_BINS, 0u);
    hist[n] = 1;
    bool usable = false;
    int rc = ensure_rtab(c, seed, T, K, jmax, hist.data(), &usable);
    if (rc) return rc;
    if (!usable) return fail(c, M6A_ENOMEM, "no device memory for the index table of bag size %d", n);
    const int64_t slot = c->rt.slot_of_n[n], A = (int64_t)T * K;
    uint32_t total = 0;
    HIPCHK(c, hipMemcpyAsync(&total, c->rt.RS + slot * (c->rt.n_blk + 1) + c->rt.n_blk, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if ((int64_t)total < A * jmax) return fail(c, M6A_ESTREAM, "MT19937 stream too short for a flush group");
    *C = c->rt.C + slot * c->rt.n_blk * 64;
    return M6A_OK;
}

// accepted indices for pool_reg_kernel: idx16[j][T + 8][K] 16-bit words (0x1000 | 2 x index: the draw's M0, a register pair
// per bag entry), iterations in order, one round of zero padding for the prefetch past the end.  Bags of one read: all zero
// (no operand indexed: every draw reads entry 0).
int ensure_table_reg(m6a_ctx *c, uint32_t seed, int n, int T, int K, int jmax)
{
    auto &k = c->tab_reg_key;
    if (k.valid && k.seed == seed && k.n == n && k.T == T && k.K == K && k.jmax >= jmax) return M6A_OK;
    k.valid = false;                                       // the table is rewritten below: a failure must not leave the old key standing
    const size_t per_j = (size_t)(T + 8) * K;              // 16-bit words per position; K = 20: 40 bytes per iteration
    const size_t bytes = (size_t)jmax * per_j * 2 + 4096;  // the kernel's look-ahead touches up to 2 KB past the last row
    HIPCHK(c, c->tab_reg.ensure(bytes));
    HIPCHK(c, hipMemsetAsync(c->tab_reg.p, 0, bytes, c->stream));
    const uint16_t *C = nullptr;
    int rc = uniform_slice(c, seed, n, T, K, jmax, &C);
    if (rc) return rc;
    if (C) {
        const int64_t A = (int64_t)T * K;
        hipLaunchKernelGGL(rtab_to_reg_table_kernel, dim3((unsigned)((A * jmax + 255) / 256)), dim3(256), 0, c->stream,
                           C, A, (int64_t)per_j, jmax, (uint16_t *)c->tab_reg.p);
        HIPCHK(c, hipGetLastError());
    }
    k = {seed, n, T, K, jmax, true};
    return M6A_OK;
}

// accepted-index table of pool_table_kernel (the LDS-gather fallback for uniform bags): tab[j][row][plane][lane],
// 4 byte offsets (8 * index) per dword, lane = accumulator chain of the pairwise sum (plan_iteration)
int ensure_table(m6a_ctx *c, uint32_t seed, int n, int T, int K, int jmax)
{
    int rc = ensure_mean_plan(c, T);    // invalidates the key when the plan changes
    if (rc) return rc;
    auto &k = c->tab_key;
    if (k.valid && k.seed == seed && k.n == n && k.T == T && k.K == K && k.jmax >= jmax) return M6A_OK;
    k.valid = false;
    const MeanPlan &p = c->plan;
    const int rows = (int)p.row_pass.size();
    // which iteration each (row, lane) holds: lanes are the accumulator chains of the pairwise sum
    c->iter_of.resize((size_t)rows * 64);
    for (int r = 0; r < rows; r++)
        for (int l = 0; l < 64; l++) c->iter_of[(size_t)r * 64 + l] = plan_iteration(p, r, l);
    const size_t words = (size_t)jmax * rows * 5 * 64;
    HIPCHK(c, c->tab.ensure(words * 4 + c->iter_of.size() * 4));
    int *d_iter = (int *)((uint32_t *)c->tab.p + words);
    HIPCHK(c, hipMemcpyAsync(d_iter, c->iter_of.data(), c->iter_of.size() * 4, hipMemcpyHostToDevice, c->stream));
    const uint16_t *C = nullptr;
    rc = uniform_slice(c, seed, n, T, K, jmax, &C);
    if (rc) return rc;
    if (C) {
        hipLaunchKernelGGL(rtab_to_lds_table_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, c->stream,
                           C, (int64_t)T * K, K, rows, jmax, (const int *)d_iter, (uint32_t *)c->tab.p);
        HIPCHK(c, hipGetLastError());
    } else {
        HIPCHK(c, hipMemsetAsync(c->tab.p, 0, words * 4, c->stream));   // bags of one read: every draw is read 0
    }
    k = {seed, n, T, K, jmax, true};
    return M6A_OK;
}

// ---- profiling ---------------------------------------------------------------------------------
void prof_begin(m6a_ctx *c, int kind)
{
    Profiler &p = c->prof;
    if (!p.on || !(p.mask >> kind & 1)) return;
    if (p.used[kind] >= kMaxProfiled) { p.dropped[kind]++; return; }
    if ((int)p.start[kind].size() <= p.used[kind]) {
        hipEvent_t a, b;
        (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        p.start[kind].push_back(a); p.stop[kind].push_back(b);
    }
    (void)hipEventRecord(p.start[kind][p.used[kind]], c->stream);
}
void prof_end(m6a_ctx *c, int kind)
{
    Profiler &p = c->prof;
    if (!p.on || !(p.mask >> kind & 1) || p.used[kind] >= kMaxProfiled) return;
    (void)hipEventRecord(p.stop[kind][p.used[kind]], c->stream);
    p.used[kind]++;
}

void host_bag_range(m6a_ctx *c, const int64_t *off, int64_t S);
int sync_and_check(m6a_ctx *c);

// bag-size range and histogram (they decide the pooling kernel) and total reads: one 4 KB read-back,
// which blocks on the stream
int query_bags(m6a_ctx *c, const int64_t *d_off, int64_t S)
{
    if (c->side_work) { HIPCHK(c, hipStreamSynchronize(c->s_prep)); c->side_work = false; }   // a pending check uses the same scratch
    c->h_minmax[0] = ~0ull; c->h_minmax[1] = 0ull; c->h_minmax[2] = 0ull;
    HIPCHK(c, hipMemcpyAsync(c->d_minmax, c->h_minmax, 24, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_hist, 0, M6A_HIST_BINS * 4, c->stream));
    hipLaunchKernelGGL(bag_minmax_kernel, dim3((unsigned)std::min<int64_t>((S + 255) / 256, 512)), dim3(256), 0,
                       c->stream, d_off, S, c->d_minmax, c->d_hist);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_minmax, c->d_minmax, 24, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_hist, c->d_hist, M6A_HIST_BINS * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->bag_min = (int64_t)c->h_minmax[0];
    c->bag_max = (int64_t)c->h_minmax[1];
    c->n_reads = (int64_t)c->h_minmax[2];
    return M6A_OK;
}

// Device-pointer calls: bag statistics of d_off.  Default: query_bags (a read-back, blocks on the stream).  After
// m6a_set_host_offsets the statistics come from the caller's host copy instead -- the loader that built the CSR array
// has it -- and the device array is only CHECKED against them, asynchronously: the call does not block, consecutive
// calls queue back to back, a mismatch surfaces as a deferred M6A_EINVAL at the next m6a_sync.
int bag_stats(m6a_ctx *c, const int64_t *d_off, int64_t S)
{
    const int64_t *h = c->hint_off;
    c->hint_off = nullptr;                                     // one call
    if (!h) return query_bags(c, d_off, S);
    if (h[0] != 0) return fail(c, M6A_EINVAL, "off[0] must be 0");
    host_bag_range(c, h, S);
    if (c->bag_min < 0) return fail(c, M6A_EINVAL, "off[] must be non-decreasing");
    unsigned long long hash = 0;
    for (int i = 0; i < M6A_HIST_BINS; i++) hash += (unsigned long long)c->h_hist[i] * m6a_bin_weight(i);
    // the check runs on the side stream, next to the kernels of the call (it orders itself behind everything queued so far)
    hipStream_t st = c->s_prep ? c->s_prep : c->stream;
    if (st != c->stream) {
        HIPCHK(c, hipEventRecord(c->ev_main, c->stream));
        HIPCHK(c, hipStreamWaitEvent(st, c->ev_main, 0));
    }
    HIPCHK(c, hipMemsetAsync(c->d_minmax, 0xff, 8, st));
    HIPCHK(c, hipMemsetAsync(c->d_minmax + 1, 0, 16, st));
    HIPCHK(c, hipMemsetAsync(c->d_hist, 0, M6A_HIST_BINS * 4, st));
    hipLaunchKernelGGL(bag_minmax_kernel, dim3((unsigned)std::min<int64_t>((S + 255) / 256, 512)), dim3(256), 0,
                       st, d_off, S, c->d_minmax, c->d_hist);
    hipLaunchKernelGGL(bag_verify_kernel, dim3(1), dim3(256), 0, st, c->d_minmax, c->d_hist, (unsigned long long)c->bag_min,
                       (unsigned long long)c->bag_max, (unsigned long long)c->n_reads, hash, c->d_err);
    HIPCHK(c, hipGetLastError());
    c->side_work = true;                                       // m6a_sync also waits for the side stream
    return M6A_OK;
}

// ---- launches (all pointers are device pointers here) ------------------------------------------
// c->bag_min must describe `off` (query_bags / host_bag_range ran for this call)
int launch_encode(m6a_ctx *c, const float *X, const uint8_t *km, const int64_t *off, int64_t S,
                  int64_t R, float *rp)
{
    if (R <= 0 || S <= 0) return M6A_OK;
    EncArgs a;
    a.X = X; a.site_kmers = km; a.off = off; a.wfrag = c->d_wfrag; a.emb = c->d_emb; a.read_prob = rp;
    a.wfrag2 = c->d_wfrag2; a.w1e_tab = c->d_w1e; a.bn = c->d_w1e + M6A_W1E_FLOATS; a.err = c->d_err;
    a.n_sites = S; a.n_reads = R; a.n_tiles = (R + 31) / 32; a.b3 = c->b3;
    a.clk = c->prof.clk_for(0);
#ifdef M6A_AB_W3
    // A/B build only: enc_site16_kernel at three waves per SIMD (the grid is sized before the kernel is chosen: every encoder gets 12 waves per CU's worth of
    // work items in this build, the two-wave kernels simply run their blocks in two rounds -- only enc_site16_kernel's time means anything here)
    const int64_t max_waves = (int64_t)c->n_cu * 12;
#else
    const int64_t max_waves = (int64_t)c->n_cu * 8;        // 2 blocks/CU x 4 waves
#endif
    a.tiles_per_wave = (a.n_tiles + max_waves - 1) / max_waves;
    const int64_t waves = (a.n_tiles + a.tiles_per_wave - 1) / a.tiles_per_wave;
    const unsigned blocks = (unsigned)((waves + 3) / 4);
    // every bag >= 16 reads: a 32-read tile spans <= 3 sites, and the site lookup is the scalar 32-bit chain of
    // enc_site16_kernel (the reference's 16 slots, 116 MFMAs per tile: auto) / enc_csite_kernel (12-slot layer 1, 106: opt-in);
    // otherwise enc_kernel: the same 16-slot arithmetic behind a per-lane 64-bit walk of off[].
    // auto (0) and mode 1 are the reference's float32 operations in the reference's order on EVERY input; the 12-slot kernel runs
    // only for callers that asked for it: mode 2 (strict: every bag must have >= 16 reads) or mode 4 ("fast": where it applies).
    const bool fits32 = S < 0x7ffffff0LL && a.n_tiles < 0x7ffffff0LL;
    if (c->enc_variant == 2 && !fits32) return fail(c, M6A_EUNSUPPORTED, "12-slot encoder: more than 2^31 sites or tiles");
    const bool scalar_chain = c->bag_min >= M6A_CSITE_MIN_BAG && fits32;
    const bool csite = c->enc_variant == 2 || (c->enc_variant == 4 && scalar_chain);
    const bool site16 = !csite && c->enc_variant != 3 && scalar_chain;
    c->enc_variant_used = csite ? "csite12" : "general16";
    c->enc_kernel_used = csite ? "enc_csite_kernel" : site16 ? "enc_site16_kernel" : "enc_kernel";
    prof_begin(c, 0);
    if (csite) hipLaunchKernelGGL(enc_csite_kernel, dim3(blocks), dim3(256), 0, c->stream, a);
    else if (site16) hipLaunchKernelGGL(enc_site16_kernel, dim3(blocks), dim3(256), 0, c->stream, a);
    else hipLaunchKernelGGL(enc_kernel, dim3(blocks), dim3(256), 0, c->stream, a);
    prof_end(c, 0);
    HIPCHK(c, hipGetLastError());
    return M6A_OK;
}

RtabUse rtab_use(m6a_ctx *c, int64_t nmax, uint32_t si_base = 0, uint32_t si_count = 0)
{
    RtabUse u;
    u.C = c->rt.C; u.RS = c->rt.RS; u.slot_of_n = (const int32_t *)c->ctl_dev.p + M6A_HIST_BINS;
    u.rank = (uint32_t *)c->rt_rank.p; u.order = (const uint32_t *)c->rt_order.p;
    u.c_stride = c->rt.n_blk * 64; u.n_blk = (uint32_t)c->rt.n_blk;
    u.bag_cap = (int)std::max<int64_t>(64, (nmax + 63) / 64 * 64);
    u.si_base = si_base; u.si_count = si_count;
    return u;
}

// Ragged bags, first half: decide whether this call pools through the per-bag-size index tables, build the ones
// that are missing and launch the preparation -- every site's rank in its table and the bag-size order -- all on
// the context's current stream.  Needs only off[] and the histogram of query_bags / host_bag_range, not the read
// probabilities, so m6a_infer runs it on the side stream next to the encoder (pool_setup_aside; a latency chain on a
// few waves: 0.1 ms that would otherwise sit between the two big kernels).  a: off, goff, n_groups, n_sites, T, K, err.
int rtab_prepare(m6a_ctx *c, PoolArgs a, int64_t nmax, int64_t gmax, uint32_t seed, bool *use)
{
    *use = false;
    const int T = a.T, K = a.K;
    const int64_t S = a.n_sites;
    const int64_t need = stream_need(gmax, T, K);
    // Default: per-bag-size index tables (pool_rtab_kernel) when every bag fits one (n <= 4096)
    // and the work seen so far pays for the tables still lacking (a table = one pass over the stream, about
    // what 50 sites cost the scan kernels); otherwise the scan kernels replay the stream per site.
    if (nmax > M6A_RTAB_MAX_N || c->scan_driver == 1 || c->scan_driver == 2) {
        if (c->scan_driver == 3)
            return fail(c, M6A_EUNSUPPORTED, "index-table pooling needs every bag <= %d reads (largest: %lld)", M6A_RTAB_MAX_N, (long long)nmax);
        return M6A_OK;
    }
    int distinct = 0;
    const int missing = rtab_missing(c, seed, T, K, need, c->h_hist, &distinct);
    // Tables that exist are always worth using (a 32-site call: 125 us against 485 us on the scan kernels).
    // Missing ones are built once the sites pooled for this (seed, T*K) -- this call's plus those of earlier
    // calls that went to the scan kernels, e.g. a caller that hands over one flush group at a time -- would
    // have paid for them.
    if (c->rt_credit_seed != seed || c->rt_credit_A != (int64_t)T * K) { c->rt_credit_seed = seed; c->rt_credit_A = (int64_t)T * K; c->rt_credit = 0; }
    bool use_rtab = c->scan_driver == 3 || missing == 0 || c->rt_credit + S >= (int64_t)16 * missing;
    if (use_rtab) c->rt_credit = 0; else c->rt_credit += S;
    if (!use_rtab) return M6A_OK;
    int rc = ensure_rtab(c, seed, T, K, gmax, c->h_hist, &use_rtab);      // false: over the memory budget
    if (rc) return rc;
    if (!use_rtab) return M6A_OK;
    a.raw = (const uint32_t *)c->raw.p; a.raw_len = c->raw_len;
    HIPCHK(c, c->rt_rank.ensure((size_t)S * 4));
    HIPCHK(c, c->rt_order.ensure((size_t)S * 4));
    HIPCHK(c, c->ctl_dev.ensure(kCtlWords * 4));
    // sites grouped by bag size: cursor[n] = first position of size n (the histogram came with the bag range).
    // The kernel gives XCD x the x-th eighth of this order, and a site's cost grows with its bag size:
    // sizes are dealt to the eighths by n mod 8, so every XCD gets the whole range of sizes and still owns the
    // tables of "its" sizes; largest first inside an eighth, so the longest sites do not start last.
    HIPCHK(c, hipEventSynchronize(c->ev_ctl));                // the previous call's upload from h_ctl (calls need not block any more)
    // Bags above M6A_RTAB_SMALL_N reads come first, as a block of their own: launch_pool gives them a launch with the
    // LDS bag they need (16 KB at 4 096 reads, nine sites per CU) and everybody else one with a small bag.
    uint32_t *cur = ctl_cursor(c);
    uint32_t run = 0;
    for (int n = M6A_HIST_BINS - 1; n > M6A_RTAB_SMALL_N; n--) { cur[n] = run; run += c->h_hist[n]; }
    for (int x = 0; x < 8; x++)
        for (int n = M6A_RTAB_SMALL_N - ((M6A_RTAB_SMALL_N - x) & 7); n >= 0; n -= 8) { cur[n] = run; run += c->h_hist[n]; }
    std::memcpy(ctl_slot(c), c->rt.slot_of_n, sizeof c->rt.slot_of_n);
    HIPCHK(c, hipMemcpyAsync(c->ctl_dev.p, c->h_ctl, (size_t)(M6A_HIST_BINS + M6A_RTAB_MAX_N + 1) * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_ctl, c->stream));
    const RtabUse u = rtab_use(c, nmax);
    const unsigned n_order_blocks = (unsigned)((S + 255) / 256);
    const unsigned n_chain_blocks = (unsigned)std::min<int64_t>((a.n_groups + 3) / 4, (int64_t)c->n_cu * 16);
    hipLaunchKernelGGL(rtab_prep_kernel, dim3(n_order_blocks + n_chain_blocks), dim3(256), 0, c->stream, a, u,
                       (uint32_t *)c->ctl_dev.p, (uint32_t *)c->rt_order.p, n_order_blocks);
    HIPCHK(c, hipGetLastError());
    *use = true;
    return M6A_OK;
}

// dry: do everything the pooling of this call needs EXCEPT the pooling kernels -- flush-group offsets, the mean plan,
// the MT19937 stream, index tables, and for ragged bags the rank / order preparation -- on the context's current stream.
// m6a_infer runs it on the side stream while the encoder is busy (pool_setup_aside), then calls again for the kernels.
int launch_pool(m6a_ctx *c, const float *rp, const int64_t *off, int64_t S, int T, int K, float thr,
                uint32_t seed, int64_t bs, int64_t spb, float *site, double *mod, bool dry)
{
    // did a dry run already do the ragged preparation of exactly this call?  One-shot.
    const bool prepared = c->prep.ready && c->prep.off == off && c->prep.S == S && c->prep.T == T && c->prep.K == K &&
                          c->prep.seed == seed && c->prep.bs == bs && c->prep.spb == spb;
    c->prep.ready = false;
    if (S <= 0) return M6A_OK;
    int rc = ensure_groups(c, S, bs, spb);
    if (rc) return rc;
    const int64_t nmin = c->bag_min, nmax = c->bag_max;      // from query_bags()
    if (nmin < 0 || nmax > 0x7fffffff) return fail(c, M6A_EINVAL, "off[] is not a non-decreasing CSR array");

    PoolArgs a;
    memset(&a, 0, sizeof a);
    a.read_prob = rp; a.off = off; a.goff = (const int64_t *)c->goff.p; a.site_prob = site; a.mod_ratio = mod;
    a.err = c->d_err; a.n_groups = c->goff_key.G; a.n_sites = S; a.T = T; a.K = K; a.thr = thr;
    a.clk = dry ? nullptr : c->prof.clk_for(1);
    const int64_t gmax = c->goff_key.gmax;

    rc = ensure_mean_plan(c, T);
    if (rc) return rc;
    const bool uniform = nmin == nmax && nmin >= 1 && nmin <= M6A_TABLE_MAX_N && K == 20 && gmax <= 4096;
    // register kernel: stack in 8 register quads, 32-bit byte offsets into read_prob, table <= 256 MB
    const bool reg_ok = uniform && c->plan.depth <= M6A_REG_STACK && c->n_reads < (int64_t)1 << 30 &&
                        (int64_t)gmax * (T + 8) * K * 2 <= (int64_t)256 << 20;
    if (uniform && reg_ok && c->table_variant != 1) {
        rc = ensure_table_reg(c, seed, (int)nmin, T, K, (int)gmax);
        if (rc) return rc;
        plan_args(c, a);
        a.tab = (const uint32_t *)c->tab_reg.p; a.uniform_n = (int)nmin; a.jmax = (int)gmax;
        if (dry) return M6A_OK;
        c->pool_variant = "table-reg";
        prof_begin(c, 1);
        // one wavefront = position j of 256 flush groups (4 sites per lane); the grid is 8 x ceil(items / 8) so that every
        // XCD takes a contiguous run of items (m6a_pool_reg.hip: the waves that share cache lines of read_prob share an L2)
        const int64_t wpj = (a.n_groups + 255) / 256;
        a.reg_items = wpj * a.jmax;
        hipLaunchKernelGGL(pool_reg_kernel, dim3((unsigned)((a.reg_items + 7) / 8 * 8)), dim3(64), 0, c->stream, a);
        prof_end(c, 1);
    } else if (uniform && c->plan.max_merge <= 15) {
        rc = ensure_table(c, seed, (int)nmin, T, K, (int)gmax);
        if (rc) return rc;
        if (dry) return M6A_OK;
        plan_args(c, a);
        a.tab = (const uint32_t *)c->tab.p; a.uniform_n = (int)nmin; a.jmax = (int)gmax;
        // workgroups are bound to a position j: jmax x nbj of them, 5 resident per CU (LDS)
        const int64_t gblocks = (a.n_groups + 7) / 8;
        int64_t nbj = std::max<int64_t>(1, ((int64_t)c->n_cu * 5 + a.jmax - 1) / a.jmax);
        nbj = std::min<int64_t>(nbj, (gblocks + 3) / 4);
        const unsigned blocks = (unsigned)(nbj * a.jmax);
        c->pool_variant = "table";
        prof_begin(c, 1);
        const size_t mean_lds = (size_t)4 * (128 + 8 * a.stack_depth) * sizeof(float);
        hipLaunchKernelGGL(pool_table_kernel, dim3(blocks), dim3(256), mean_lds, c->stream, a);
        prof_end(c, 1);
    } else {
        // Ragged bags.  Default: per-bag-size index tables (pool_rtab_kernel) when every bag fits one (n <= 4096)
        // and the work seen so far pays for the tables still lacking (a table = one pass over the stream, about
        // what 50 sites cost the scan kernels); otherwise the scan kernels replay the stream per site.
        const int64_t need = stream_need(gmax, T, K);
        bool use_rtab = false;
        if (prepared && !dry) {
            use_rtab = c->prep.use;                           // the dry run decided (and, if so, built and launched)
        } else {
            rc = rtab_prepare(c, a, nmax, gmax, seed, &use_rtab);
            if (rc) return rc;
        }
        if (dry) {
            c->prep.ready = true; c->prep.use = use_rtab;
            c->prep.off = off; c->prep.S = S; c->prep.bs = bs; c->prep.spb = spb; c->prep.T = T; c->prep.K = K; c->prep.seed = seed;
            if (!use_rtab) {                                  // the scan kernels' share of the set-up
                rc = ensure_raw(c, seed, need);
                if (rc) return rc;
                HIPCHK(c, c->start_pos.ensure((size_t)S * sizeof(uint32_t)));
            }
            return M6A_OK;
        }
        if (use_rtab) {
            plan_args(c, a);
            a.raw = (const uint32_t *)c->raw.p; a.raw_len = c->raw_len;
            c->pool_variant = "ragged-table";
            // positions [0, n_big) of the bag-size order hold the bags above M6A_RTAB_SMALL_N reads (rtab_prepare)
            int64_t n_big = 0;
            for (int n = M6A_RTAB_SMALL_N + 1; n < M6A_HIST_BINS; n++) n_big += c->h_hist[n];
            prof_begin(c, 1);
            auto launch = [&](int64_t base, int64_t count, int64_t cap_n) {
                if (count <= 0) return;
                const RtabUse u = rtab_use(c, cap_n, (uint32_t)base, (uint32_t)count);
                const size_t lds = (size_t)(u.bag_cap + 16 + M6A_MEAN_STACK) * sizeof(float);
                const unsigned blocks = (unsigned)((count + 7) / 8 * 8);    // one wavefront (workgroup) per site
                if (K == 20) hipLaunchKernelGGL(pool_rtab_kernel<20>, dim3(blocks), dim3(64), lds, c->stream, a, u);
                else hipLaunchKernelGGL(pool_rtab_kernel<0>, dim3(blocks), dim3(64), lds, c->stream, a, u);
            };
            launch(0, n_big, nmax);
            launch(n_big, S - n_big, std::min<int64_t>(nmax, M6A_RTAB_SMALL_N));
            prof_end(c, 1);
            HIPCHK(c, hipGetLastError());
            return M6A_OK;
        }
        rc = ensure_raw(c, seed, need);
        if (rc) return rc;
        plan_args(c, a);
        a.raw = (const uint32_t *)c->raw.p; a.raw_len = c->raw_len;
        HIPCHK(c, c->start_pos.ensure((size_t)S * sizeof(uint32_t)));
        a.start_pos = (uint32_t *)c->start_pos.p;
        // LDS bag sized to the largest bag of this call (bags beyond M6A_BAG_LDS gather from global)
        // a power of two: a masked stream word (< 2^ceil(log2 n)) is then always a valid LDS bag index, so the
        // fast path gathers without clamping
        int64_t cap = 64;
        while (cap < nmax && cap < M6A_BAG_LDS) cap *= 2;
        a.bag_cap = (int)cap;
        const size_t lds = (size_t)4 * (a.bag_cap + (32 * K + 256) * 5 / 4 + 32 + M6A_MEAN_STACK) * sizeof(float);
        // resident workgroups per CU: what LDS allows, but with big bags (random gathers over >= 1 KB per wave) five
        // measured best -- 4.19 ms against 4.6 ms at six on the 50..500-read shape; small bags keep gaining up to 6-8
        const int64_t wg_per_cu = std::max<int64_t>(1, std::min<int64_t>(a.bag_cap >= 256 ? 5 : 8, (160 * 1024) / (int64_t)lds));
        const int64_t wave_slots = (int64_t)c->n_cu * wg_per_cu * 4;
        // enough flush groups to fill most of the chip: walk each group's sites in sequence (the stream is
        // scanned once); otherwise buy 32x the parallelism with a counting pass that finds every site's start
        // position first.  Measured on 50..500-read bags: at 0.3 x the wave slots the per-site driver wins
        // (1.3 vs 1.9 ms), at 0.6 x they tie, at 1.2 x the per-group driver is 11 % ahead, at 5 x 22 %.
        const bool by_group = c->scan_driver ? c->scan_driver == 1 : a.n_groups * 5 >= wave_slots * 3;
        c->pool_variant = by_group ? "scan-group" : "scan-site";
        prof_begin(c, 1);
        if (by_group) {
            const unsigned blocks = (unsigned)std::min<int64_t>((a.n_groups + 3) / 4, wave_slots / 4);
            if (K == 20) hipLaunchKernelGGL(pool_scan_group_kernel<20>, dim3(blocks), dim3(256), lds, c->stream, a);
            else hipLaunchKernelGGL(pool_scan_group_kernel<0>, dim3(blocks), dim3(256), lds, c->stream, a);
        } else {
            hipLaunchKernelGGL(pool_scan_start_kernel, dim3((unsigned)std::min<int64_t>((a.n_groups + 3) / 4, (int64_t)c->n_cu * 8)),
                               dim3(256), 0, c->stream, a);
            const unsigned blocks = (unsigned)std::min<int64_t>((S + 3) / 4, wave_slots / 4);
            if (K == 20) hipLaunchKernelGGL(pool_scan_site_kernel<20>, dim3(blocks), dim3(256), lds, c->stream, a);
            else hipLaunchKernelGGL(pool_scan_site_kernel<0>, dim3(blocks), dim3(256), lds, c->stream, a);
        }
        prof_end(c, 1);
    }
    HIPCHK(c, hipGetLastError());
    return M6A_OK;
}

// m6a_infer, device pointers, after the encoder has been launched: the pooling's set-up (a dry launch_pool) runs on the
// side stream next to it -- in the steady state that is the ragged rank / order kernel (0.1 ms), in the first call the
// MT19937 stream and the index tables as well (milliseconds, incl. host syncs that now wait for the side stream only).
// The caller recorded ev_main on the context's stream BEFORE the encoder: the set-up orders itself behind everything
// queued up to there (the previous call's pooling still reads what it rebuilds), not behind the encoder.
int pool_setup_aside(m6a_ctx *c, const int64_t *off, int64_t S, int T, int K, uint32_t seed, int64_t bs, int64_t spb)
{
    c->prep.ready = false;
    if (S <= 0 || !c->s_prep) return M6A_OK;
    HIPCHK(c, hipStreamWaitEvent(c->s_prep, c->ev_main, 0));
    hipStream_t main_stream = c->stream;
    c->stream = c->s_prep;
    const int rc = launch_pool(c, nullptr, off, S, T, K, 0.0f, seed, bs, spb, nullptr, nullptr, true);
    c->stream = main_stream;
    if (rc) { c->prep.ready = false; return rc; }
    HIPCHK(c, hipEventRecord(c->ev_prep, c->s_prep));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_prep, 0));
    return M6A_OK;
}

// smallest and largest bag of a CSR array.  The baseline x86-64 target has no 64-bit vector compare, so the plain loop stays
// scalar (0.35 ms per 1 M sites, in front of every call's first launch); every host that carries an MI355X has AVX2.
template <int>
static inline void bag_range_loop(const int64_t *off, int64_t S, int64_t *mn_out, int64_t *mx_out)
{
    int64_t mn = INT64_MAX, mx = 0;
    for (int64_t s = 0; s < S; s++) {
        const int64_t n = off[s + 1] - off[s];
        mn = n < mn ? n : mn;
        mx = n > mx ? n : mx;
    }
    *mn_out = mn; *mx_out = mx;
}
__attribute__((target("avx2"))) static void bag_range_avx2(const int64_t *off, int64_t S, int64_t *mn, int64_t *mx) { bag_range_loop<1>(off, S, mn, mx); }
static void bag_range_base(const int64_t *off, int64_t S, int64_t *mn, int64_t *mx) { bag_range_loop<0>(off, S, mn, mx); }

void host_bag_range(m6a_ctx *c, const int64_t *off, int64_t S)
{
    // pass 1, range only: vectorised where the host has AVX2, memory-bound then (0.15-0.2 ms per 1 M sites)
    int64_t mn = INT64_MAX, mx = 0;
    static const bool avx2 = __builtin_cpu_supports("avx2");
    (avx2 ? bag_range_avx2 : bag_range_base)(off, S, &mn, &mx);
    auto bin = [](int64_t n) { return n < 0 ? 0 : n > M6A_RTAB_MAX_N ? M6A_RTAB_MAX_N + 1 : n; };
    std::memset(c->h_hist, 0, M6A_HIST_BINS * 4);
    if (mn == mx || S == 0) {
        if (S > 0) c->h_hist[bin(mn)] = (uint32_t)S;           // uniform bags: nothing to count
    } else {
        // pass 2: eight interleaved histograms, so that runs of equal bag sizes do not serialise on one counter
        c->hist_part.assign((size_t)8 * M6A_HIST_BINS, 0u);
        uint32_t (*part)[M6A_HIST_BINS] = (uint32_t (*)[M6A_HIST_BINS])c->hist_part.data();
        int64_t s = 0;
        for (; s + 8 <= S; s += 8)
            for (int k = 0; k < 8; k++) part[k][bin(off[s + k + 1] - off[s + k])]++;
        for (; s < S; s++) part[0][bin(off[s + 1] - off[s])]++;
        for (int i = 0; i < M6A_HIST_BINS; i++) {
            uint32_t t = 0;
            for (int k = 0; k < 8; k++) t += part[k][i];
            c->h_hist[i] = t;
        }
    }
    c->bag_min = S > 0 ? mn : 0; c->bag_max = mx; c->n_reads = off[S];
}

int job_busy(m6a_ctx *c)
{
    if (c->job.open) return fail(c, M6A_EINVAL, "a streaming job is open on this context (m6a_job_end or m6a_job_abort first)");
    return M6A_OK;
}

int check_pool_args(m6a_ctx *c, int64_t S, int T, int K, int rng_mode, int64_t bs, int64_t spb)
{
    if (!c) return M6A_EINVAL;
    if (S < 0) return fail(c, M6A_EINVAL, "n_sites < 0");
    if (T < 1) return fail(c, M6A_EINVAL, "n_iters must be >= 1");
    if (K < 1 || K > M6A_MAX_SAMPLES) return fail(c, M6A_EINVAL, "n_samples must be in 1..%d", M6A_MAX_SAMPLES);
    if ((int64_t)T * K > 0x3fffffff) return fail(c, M6A_EINVAL, "n_iters*n_samples too large");
    if (rng_mode != M6A_RNG_NUMPY) return fail(c, M6A_EINVAL, "rng_mode %d: M6A_RNG_NUMPY (0) is the only value (include/m6a.h)", rng_mode);
    if (bs < 1 || spb < 1) return fail(c, M6A_EINVAL, "batch_size and save_per_batch must be >= 1");
    return M6A_OK;
}

int deferred_error(m6a_ctx *c)
{
    // stream is idle here
    if (*c->h_err) {
        const int e = *c->h_err;
        *c->h_err = 0;
        if (e == 2) return fail(c, M6A_EINVAL, "encoder: a 32-read tile spans more than 3 sites (bag < 16 reads) in the 12-slot kernel");
        if (e == 4) return fail(c, M6A_EHIP, "pool_rtab_kernel: the dynamic LDS block does not start at offset 0");
        if (e == 3) return fail(c, M6A_EINVAL, "the host offsets given to m6a_set_host_offsets differ from the device off[] of the call");
        return fail(c, M6A_ESTREAM, "MT19937 stream too short for a flush group");
    }
    return M6A_OK;
}

int sync_and_check(m6a_ctx *c)
{
    if (c->side_work) { HIPCHK(c, hipStreamSynchronize(c->s_prep)); c->side_work = false; }
    HIPCHK(c, hipMemcpyAsync(c->h_err, c->d_err, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (void *p : c->graveyard) (void)hipFree(p);           // the context is idle: outgrown arenas go now (ensure_rtab)
    c->graveyard.clear();
    if (*c->h_err) {
        HIPCHK(c, hipMemsetAsync(c->d_err, 0, sizeof(int), c->stream));
        return deferred_error(c);
    }
    return M6A_OK;
}

// Background half of m6a_create: what a first call with the reference's DEFAULT job parameters would otherwise
// build inside the call -- seed 0 (scripts/inference.py:60), num_iterations 1000 (:56), 20 samples
// (inference_utils.py:54), batch_size 16 x save_per_batch 2 (:46-50) -> flush groups of <= 32 sites: the pairwise-sum
// plan, the MT19937 stream, the register kernel's table of the smallest legal bag (min_reads = 20, constants.py:14),
// and -- only when M6A_WARM_SLOTS asks -- the index tables of bag sizes 2 .. M6A_WARM_SLOTS - 1.  Runs on the side stream from its
// own thread, so neither m6a_create nor the caller's loader waits for it; a first call with other parameters simply
// rebuilds what differs, exactly as before.  M6A_WARMUP=0 turns it off.
void warm_default(m6a_ctx *c)
{
    if (hipSetDevice(c->device) != hipSuccess) { (void)hipGetLastError(); return; }
    const uint32_t seed = 0;
    const int T = 1000, K = 20, n = 20;
    const int64_t gmax = 32;
    hipStream_t main_stream = c->stream;
    c->stream = c->s_prep;
    // map the code object of every translation unit now: the first launch of a kernel from each costs 0.3-1.2 ms otherwise
    hipLaunchKernelGGL(m6a_touch_kernels, dim3(1), dim3(1), 0, c->s_prep);
    hipLaunchKernelGGL(m6a_touch_pool_reg, dim3(1), dim3(1), 0, c->s_prep);
    hipLaunchKernelGGL(m6a_touch_pool_rtab, dim3(1), dim3(1), 0, c->s_prep);
    (void)hipGetLastError();
    int rc = ensure_mean_plan(c, T);
    if (!rc) rc = ensure_raw(c, seed, stream_need(gmax, T, K));
    if (!rc) {
        // The ragged kernels' per-bag-size index tables are NOT built on speculation (rounds 2-3 built sizes 2..511 here: 1.4 GB
        // and a 1.2 ms pass that a uniform-bag or small caller never uses); a first ragged call builds the sizes it meets inside
        // the call.  M6A_WARM_SLOTS=N asks for sizes 2..N-1 up front (a service that knows ragged jobs are coming: 512 covers
        // the reference's default read cap); the memory is then taken at the caller's word.
        const char *e = getenv("M6A_WARM_SLOTS");
        c->rt_presize = e && atoi(e) > 0 ? std::min(atoi(e), M6A_RTAB_MAX_N + 1) : 0;
        rc = ensure_table_reg(c, seed, n, T, K, (int)gmax);
        if (!rc && c->rt_presize > 2) {
            // ... and the index table of every bag size the arena was sized for: one pass over the stream for all of them
            // (1.2 ms of an idle GPU); a first call then only builds tables for bags beyond that
            std::vector<uint32_t> hist(M6A_HIST_BINS, 0u);
            for (int m = 2; m < c->rt_presize && m <= M6A_RTAB_MAX_N; m++) hist[m] = 1;
            bool usable = false;
            rc = ensure_rtab(c, seed, T, K, gmax, hist.data(), &usable);
        }
        c->rt_presize = 0;
    }
    (void)hipStreamSynchronize(c->s_prep);
    c->stream = main_stream;
    if (rc) { c->tab_reg_key.valid = false; c->err.clear(); }   // not an error of anybody's call: the first call builds what it needs
}

}  // namespace m6a_detail


// =================================================================================================
extern "C" {

const char *m6a_version(void) { return "m6a_hip 0.1 (gfx950)"; }

const char *m6a_last_error(const m6a_ctx *ctx)
{
    settle(const_cast<m6a_ctx *>(ctx));                    // the background set-up writes the same string
    return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int m6a_create(m6a_ctx **out, const float *weights, size_t n_floats, int device_id)
{
    if (!out) return M6A_EINVAL;
    *out = nullptr;
    if (!weights || n_floats != M6A_N_WEIGHTS)
        return fail(nullptr, M6A_EINVAL, "weights must be %d floats (got %zu)", M6A_N_WEIGHTS, n_floats);
    // the encoder carries layer 2's weights scaled by 2^64 (build_fragments): a finite weight that would overflow there is
    // refused here, loudly, not turned into inf inside the kernel (the bundled checkpoints' largest is below 1)
    for (int i = O_W2; i < O_W3; i++)
        if (std::isfinite(weights[i]) && std::fabs(weights[i]) >= 0x1p+63f)
            return fail(nullptr, M6A_EUNSUPPORTED, "layer-2 weight %d is %g: beyond 2^63 in magnitude", i - O_W2, (double)weights[i]);
    // M6A_ENCODER=general16 | csite12 | walk16 | fast: the context starts with that encoder selected (m6a_set_encoder_variant
    // 1 / 2 / 3 / 4) -- for callers that cannot be changed.  A value that is none of these is refused, not ignored: a typo must
    // not silently select another kernel.
    int env_enc = 0;
    if (const char *ev = getenv("M6A_ENCODER")) {
        if (!*ev || !strcmp(ev, "auto") || !strcmp(ev, "reference")) env_enc = 0;
        else if (!strcmp(ev, "general16")) env_enc = 1;
        else if (!strcmp(ev, "csite12")) env_enc = 2;
        else if (!strcmp(ev, "walk16")) env_enc = 3;
        else if (!strcmp(ev, "fast")) env_enc = 4;
        else return fail(nullptr, M6A_EINVAL, "M6A_ENCODER=%s: must be auto, reference, general16, csite12, walk16 or fast", ev);
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return fail(nullptr, M6A_ENODEV, "no HIP device visible");
    }
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, M6A_EINVAL, "device_id %d out of range (%d devices)", device_id, ndev);
    m6a_ctx *c = new (std::nothrow) m6a_ctx;
    if (!c) return fail(nullptr, M6A_ENOMEM, "out of host memory");
    c->device = device_id;
#define CRCHK(expr)                                                                                  \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            (void)hipGetLastError();                                                                 \
            fail(nullptr, M6A_EHIP, "%s: %s", #expr, hipGetErrorString(e_));                         \
            m6a_destroy(c);                                                                          \
            return M6A_EHIP;                                                                         \
        }                                                                                            \
    } while (0)
    CRCHK(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    CRCHK(hipGetDeviceProperties(&prop, device_id));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fail(nullptr, M6A_ENODEV, "device %d is %s; this library is built for gfx950 only", device_id, prop.gcnArchName);
        m6a_destroy(c);
        return M6A_ENODEV;
    }
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    CRCHK(hipStreamCreate(&c->own_stream));
    CRCHK(hipStreamCreateWithFlags(&c->s_prep, hipStreamNonBlocking));
    CRCHK(hipEventCreateWithFlags(&c->ev_main, hipEventDisableTiming));
    CRCHK(hipEventCreateWithFlags(&c->ev_prep, hipEventDisableTiming));
    CRCHK(hipEventCreateWithFlags(&c->ev_ctl, hipEventDisableTiming));
    c->stream = c->own_stream;
    std::vector<float> frag, frag2, w1e;
    build_fragments(weights, frag, frag2, w1e);
    CRCHK(hipMalloc((void **)&c->d_wfrag, frag.size() * sizeof(float)));
    CRCHK(hipMalloc((void **)&c->d_wfrag2, frag2.size() * sizeof(float)));
    CRCHK(hipMalloc((void **)&c->d_w1e, w1e.size() * sizeof(float)));
    CRCHK(hipMemcpy(c->d_wfrag2, frag2.data(), frag2.size() * sizeof(float), hipMemcpyHostToDevice));
    CRCHK(hipMemcpy(c->d_w1e, w1e.data(), w1e.size() * sizeof(float), hipMemcpyHostToDevice));
    CRCHK(hipMalloc((void **)&c->d_emb, 132 * sizeof(float)));
    CRCHK(hipMemcpy(c->d_wfrag, frag.data(), frag.size() * sizeof(float), hipMemcpyHostToDevice));
    CRCHK(hipMemcpy(c->d_emb, weights + O_E, 132 * sizeof(float), hipMemcpyHostToDevice));
    c->b3 = weights[O_B3];
    CRCHK(hipMalloc((void **)&c->d_err, sizeof(int)));
    CRCHK(hipMemset(c->d_err, 0, sizeof(int)));
    CRCHK(hipMalloc((void **)&c->d_minmax, 24));
    CRCHK(hipHostMalloc((void **)&c->h_minmax, 24, hipHostMallocDefault));
    CRCHK(hipHostMalloc((void **)&c->h_err, sizeof(int), hipHostMallocDefault));
    *c->h_err = 0;
    CRCHK(hipMalloc((void **)&c->d_hist, M6A_HIST_BINS * 4));
    CRCHK(hipHostMalloc((void **)&c->h_hist, M6A_HIST_BINS * 4, hipHostMallocDefault));
    CRCHK(hipHostMalloc((void **)&c->h_ctl, kCtlWords * 4, hipHostMallocDefault));
    for (int n = 0; n <= M6A_RTAB_MAX_N; n++) c->rt.slot_of_n[n] = -1;
#undef CRCHK
    c->enc_variant = env_enc;
    const char *w = getenv("M6A_WARMUP");
    if (!(w && w[0] == '0')) {
        try { c->warm = std::thread(warm_default, c); } catch (...) { /* no thread: the first call sets up what it needs */ }
        if (c->warm.joinable()) {
            static std::once_flag once;
            std::call_once(once, [] { atexit(join_background_setups); });
            std::lock_guard<std::mutex> g(g_live_mu);
            g_live.push_back(c);
        }
    }
    *out = c;
    return M6A_OK;
}

void m6a_destroy(m6a_ctx *c)
{
    if (!c) return;
    settle(c);
    {
        std::lock_guard<std::mutex> g(g_live_mu);
        g_live.erase(std::remove(g_live.begin(), g_live.end(), c), g_live.end());
    }
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int k = 0; k < 2; k++) {
        for (auto e : c->prof.start[k]) (void)hipEventDestroy(e);
        for (auto e : c->prof.stop[k]) (void)hipEventDestroy(e);
    }
    for (DevBuf *b : {&c->raw, &c->tab, &c->goff, &c->rp_scratch, &c->off_scratch, &c->start_pos, &c->plan_dev, &c->tab_reg, &c->val_idx, &c->val_y, &c->val_avg, &c->sX, &c->sK, &c->sOff,
                      &c->sP, &c->sSite, &c->sMod}) b->release();
    if (c->prof.d_clk) (void)hipFree(c->prof.d_clk);
    if (c->d_wfrag) (void)hipFree(c->d_wfrag);
    if (c->d_wfrag2) (void)hipFree(c->d_wfrag2);
    if (c->d_w1e) (void)hipFree(c->d_w1e);
    if (c->d_emb) (void)hipFree(c->d_emb);
    if (c->d_err) (void)hipFree(c->d_err);
    if (c->d_minmax) (void)hipFree(c->d_minmax);
    if (c->h_minmax) (void)hipHostFree(c->h_minmax);
    if (c->h_err) (void)hipHostFree(c->h_err);
    if (c->d_hist) (void)hipFree(c->d_hist);
    if (c->h_hist) (void)hipHostFree(c->h_hist);
    if (c->h_ctl) (void)hipHostFree(c->h_ctl);
    for (void *p : c->graveyard) (void)hipFree(p);
    if (c->rt.C) (void)hipFree(c->rt.C);
    if (c->rt.RS) (void)hipFree(c->rt.RS);
    for (DevBuf *b : {&c->ctl_dev, &c->rt_rank, &c->rt_order, &c->sOffChunk, &c->jX, &c->jP, &c->jOff, &c->gSite, &c->gMod, &c->gP, &c->mt_scratch}) b->release();
    for (auto e : c->job.ev_h2d) (void)hipEventDestroy(e);
    for (auto e : c->job.ev_enc) (void)hipEventDestroy(e);
    release_staging(c);
    comm_release(c);
    if (c->s_prep) (void)hipStreamDestroy(c->s_prep);
    if (c->ev_main) (void)hipEventDestroy(c->ev_main);
    if (c->ev_prep) (void)hipEventDestroy(c->ev_prep);
    if (c->ev_ctl) (void)hipEventDestroy(c->ev_ctl);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int m6a_set_stream(m6a_ctx *c, void *hip_stream)
{
    settle(c);
    if (!c) return M6A_EINVAL;
    // an open job's encoders are queued on the stream it was begun on: m6a_job_end would pool on a stream that never waited for them
    if (c->job.open) return job_busy(c);
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return M6A_OK;
}

int m6a_set_host_offsets(m6a_ctx *c, const int64_t *off_host)
{
    if (!c) return M6A_EINVAL;
    if (off_host && is_device_ptr(off_host)) return fail(c, M6A_EINVAL, "m6a_set_host_offsets takes a host pointer");
    c->hint_off = off_host;
    return M6A_OK;
}

int m6a_set_job_offset(m6a_ctx *c, int64_t first_site)
{
    if (!c) return M6A_EINVAL;
    if (c->job.open) return job_busy(c);                   // m6a_job_begin validated the offset the job runs with
    if (first_site < 0) return fail(c, M6A_EINVAL, "job offset must be >= 0");
    c->job_offset = first_site;
    return M6A_OK;
}

int m6a_set_encoder_variant(m6a_ctx *c, int mode)
{
    if (!c) return M6A_EINVAL;
    if (mode < 0 || mode > 4)
        return fail(c, M6A_EINVAL, "encoder variant must be 0 (auto: 16-slot), 1 (16-slot), 2 (12-slot), 3 (16-slot, per-lane walk) or 4 (fast: 12-slot where every bag has >= 16 reads)");
    c->enc_variant = mode;
    return M6A_OK;
}

const char *m6a_last_encoder_variant(const m6a_ctx *c) { return c ? c->enc_variant_used : "none"; }
const char *m6a_last_encoder_kernel(const m6a_ctx *c) { return c ? c->enc_kernel_used : "none"; }

int m6a_set_scan_driver(m6a_ctx *c, int mode)
{
    if (!c) return M6A_EINVAL;
    if (mode < 0 || mode > 3) return fail(c, M6A_EINVAL, "scan driver must be 0 (auto), 1 (group), 2 (site) or 3 (index tables)");
    c->scan_driver = mode;
    return M6A_OK;
}

int m6a_set_table_variant(m6a_ctx *c, int mode)
{
    if (!c) return M6A_EINVAL;
    if (mode < 0 || mode > 2) return fail(c, M6A_EINVAL, "table variant must be 0 (auto), 1 (LDS) or 2 (registers)");
    c->table_variant = mode;
    return M6A_OK;
}

int m6a_sync(m6a_ctx *c)
{
    settle(c);
    if (!c) return M6A_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    return sync_and_check(c);
}

int m6a_prepare_host_io(m6a_ctx *c)
{
    settle(c);
    if (!c) return M6A_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    return ensure_staging(c);
}

// Page-locked host memory for callers that have no other way to get it (plain C, Python without torch): buffers allocated here
// are the ones the host-pointer calls DMA in place (m6a_host_ring.hip is_pinned_host).
int m6a_host_alloc(size_t bytes, void **out)
{
    if (!out) return M6A_EINVAL;
    *out = nullptr;
    if (bytes == 0) return M6A_OK;
    const hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return e == hipErrorOutOfMemory ? M6A_ENOMEM : M6A_EHIP; }
    return M6A_OK;
}

int m6a_host_free(void *p)
{
    if (!p) return M6A_OK;
    if (hipHostFree(p) != hipSuccess) { (void)hipGetLastError(); return M6A_EHIP; }
    return M6A_OK;
}

int m6a_encode_reads(m6a_ctx *c, const float *X, const uint8_t *km, const int64_t *off, int64_t S, float *rp)
{
    settle(c);
    HintScope hint_scope(c);
    if (c && c->job.open) return job_busy(c);
    if (!c) return M6A_EINVAL;
    if (S < 0) return fail(c, M6A_EINVAL, "n_sites < 0");
    if (S == 0) return M6A_OK;
    if (!X || !km || !off || !rp) return fail(c, M6A_EINVAL, "null pointer argument");
    HIPCHK(c, hipSetDevice(c->device));
    const bool dev = is_device_ptr(X);
    if (dev != is_device_ptr(km) || dev != is_device_ptr(off) || dev != is_device_ptr(rp))
        return fail(c, M6A_EINVAL, "X, site_kmers, off, read_prob must be all host or all device pointers");
    if (!dev) {
        if (off[0] != 0) return fail(c, M6A_EINVAL, "off[0] must be 0");
        for (int64_t s = 0; s < S; s++) if (off[s + 1] < off[s]) return fail(c, M6A_EINVAL, "off[] must be non-decreasing");
        const int64_t R = off[S];
        if (R == 0) return M6A_OK;
        host_bag_range(c, off, S);
        Prefault pf_rp;
        pf_rp.start(rp, (size_t)R * 4, 2);
        int rc = staged_encode(c, X, km, off, S, R, rp);
        if (rc) return rc;
        return sync_and_check(c);
    }
    // device pointers: total reads (grid size) and smallest bag (kernel choice): one read-back
    int rc = bag_stats(c, off, S);
    if (rc) return rc;
    return launch_encode(c, X, km, off, S, c->n_reads, rp);
}

int m6a_site_pool(m6a_ctx *c, const float *rp, const int64_t *off, int64_t S, int T, int K, float thr,
                  uint32_t seed, int rng_mode, int64_t bs, int64_t spb, float *site, double *mod)
{
    settle(c);
    HintScope hint_scope(c);
    if (c && c->job.open) return job_busy(c);
    int rc = check_pool_args(c, S, T, K, rng_mode, bs, spb);
    if (rc) return rc;
    if (S == 0) return M6A_OK;
    if (!rp || !off || !site || !mod) return fail(c, M6A_EINVAL, "null pointer argument");
    HIPCHK(c, hipSetDevice(c->device));
    const bool dev = is_device_ptr(rp);
    if (dev != is_device_ptr(off) || dev != is_device_ptr(site) || dev != is_device_ptr(mod))
        return fail(c, M6A_EINVAL, "read_prob, off, site_prob, mod_ratio must be all host or all device pointers");
    if (dev) {
        rc = bag_stats(c, off, S);
        if (rc) return rc;
        return launch_pool(c, rp, off, S, T, K, thr, seed, bs, spb, site, mod);
    }
    if (off[0] != 0) return fail(c, M6A_EINVAL, "off[0] must be 0");
    for (int64_t s = 0; s < S; s++) if (off[s + 1] < off[s]) return fail(c, M6A_EINVAL, "off[] must be non-decreasing");
    const int64_t R = off[S];
    HIPCHK(c, c->sP.ensure((size_t)std::max<int64_t>(R, 1) * 4));
    HIPCHK(c, c->sOff.ensure((size_t)(S + 1) * 8));
    HIPCHK(c, c->sSite.ensure((size_t)S * 4));
    HIPCHK(c, c->sMod.ensure((size_t)S * 8));
    HIPCHK(c, hipMemcpyAsync(c->sP.p, rp, (size_t)R * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->sOff.p, off, (size_t)(S + 1) * 8, hipMemcpyHostToDevice, c->stream));
    host_bag_range(c, off, S);
    rc = launch_pool(c, (const float *)c->sP.p, (const int64_t *)c->sOff.p, S, T, K, thr, seed, bs, spb,
                     (float *)c->sSite.p, (double *)c->sMod.p);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(site, c->sSite.p, (size_t)S * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(mod, c->sMod.p, (size_t)S * 8, hipMemcpyDeviceToHost, c->stream));
    return sync_and_check(c);
}

int m6a_infer(m6a_ctx *c, const float *X, const uint8_t *km, const int64_t *off, int64_t S, int T, int K,
              float thr, uint32_t seed, int rng_mode, int64_t bs, int64_t spb, float *rp, float *site, double *mod)
{
    settle(c);
    HintScope hint_scope(c);
    if (c && c->job.open) return job_busy(c);
    int rc = check_pool_args(c, S, T, K, rng_mode, bs, spb);
    if (rc) return rc;
    if (S == 0) return M6A_OK;
    if (!X || !km || !off || !site || !mod) return fail(c, M6A_EINVAL, "null pointer argument");
    HIPCHK(c, hipSetDevice(c->device));
    const bool dev = is_device_ptr(X);
    if (dev != is_device_ptr(km) || dev != is_device_ptr(off) || dev != is_device_ptr(site) ||
        dev != is_device_ptr(mod) || (rp && dev != is_device_ptr(rp)))
        return fail(c, M6A_EINVAL, "all data pointers must be host pointers or all device pointers");
    if (dev) {
        rc = bag_stats(c, off, S);
        if (rc) return rc;
        const int64_t R = c->n_reads;
        float *p = rp;
        if (!p) { HIPCHK(c, c->rp_scratch.ensure((size_t)std::max<int64_t>(R, 1) * 4)); p = (float *)c->rp_scratch.p; }
        HIPCHK(c, hipEventRecord(c->ev_main, c->stream));
        rc = launch_encode(c, X, km, off, S, R, p);
        if (rc) return rc;
        rc = pool_setup_aside(c, off, S, T, K, seed, bs, spb);
        if (rc) return rc;
        return launch_pool(c, p, off, S, T, K, thr, seed, bs, spb, site, mod);
    }
    if (off[0] != 0) return fail(c, M6A_EINVAL, "off[0] must be 0");
    for (int64_t s = 0; s < S; s++) if (off[s + 1] < off[s]) return fail(c, M6A_EINVAL, "off[] must be non-decreasing");
    const int64_t R = off[S];
    HIPCHK(c, c->sSite.ensure((size_t)S * 4));
    HIPCHK(c, c->sMod.ensure((size_t)S * 8));
    host_bag_range(c, off, S);
    Prefault pf_rp, pf_out;                      // joined on every return path
    if (rp) pf_rp.start(rp, (size_t)R * 4, 2);   // measured: 2-3 threads 60-61 M sites/s, 4 and more 52 M (they contend with the copy threads)
    pf_out.start(mod, (size_t)S * 8, 1);
    // chunks of X cross PCIe while earlier chunks are being encoded; read probabilities stream back the same way
    rc = staged_encode(c, X, km, off, S, R, rp);
    if (rc) return rc;
    rc = launch_pool(c, (const float *)c->sP.p, (const int64_t *)c->sOff.p, S, T, K, thr, seed, bs, spb,
                     (float *)c->sSite.p, (double *)c->sMod.p);
    if (rc) return rc;
    return staged_outputs(c, S, site, mod);
}

int m6a_bag_forward(m6a_ctx *c, const float *X, const uint8_t *km, int64_t B, int bag, float *site)
{
    settle(c);
    HintScope hint_scope(c);
    if (c && c->job.open) return job_busy(c);
    if (!c) return M6A_EINVAL;
    if (B < 0 || bag < 1) return fail(c, M6A_EINVAL, "n_bags must be >= 0 and bag >= 1");
    if (B == 0) return M6A_OK;
    if (!X || !km || !site) return fail(c, M6A_EINVAL, "null pointer argument");
    HIPCHK(c, hipSetDevice(c->device));
    const bool dev = is_device_ptr(X);
    if (dev != is_device_ptr(km) || dev != is_device_ptr(site))
        return fail(c, M6A_EINVAL, "X, site_kmers, site_prob must be all host or all device pointers");
    const int64_t R = B * bag;
    HIPCHK(c, c->off_scratch.ensure((size_t)(B + 1) * 8));
    HIPCHK(c, c->rp_scratch.ensure((size_t)R * 4));
    int64_t *d_off = (int64_t *)c->off_scratch.p;
    float *d_p = (float *)c->rp_scratch.p;
    hipLaunchKernelGGL(iota_off_kernel, dim3((unsigned)((B + 1 + 255) / 256)), dim3(256), 0, c->stream, d_off, B + 1, (int64_t)bag);
    HIPCHK(c, hipGetLastError());
    const float *dX = X; const uint8_t *dK = km; float *dS = site;
    if (!dev) {
        HIPCHK(c, c->sX.ensure((size_t)R * 9 * 4));
        HIPCHK(c, c->sK.ensure((size_t)B * 3));
        HIPCHK(c, c->sSite.ensure((size_t)B * 4));
        HIPCHK(c, hipMemcpyAsync(c->sX.p, X, (size_t)R * 9 * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->sK.p, km, (size_t)B * 3, hipMemcpyHostToDevice, c->stream));
        dX = (const float *)c->sX.p; dK = (const uint8_t *)c->sK.p; dS = (float *)c->sSite.p;
    }
    c->bag_min = c->bag_max = bag; c->n_reads = R;
    int rc = launch_encode(c, dX, dK, d_off, B, R, d_p);
    if (rc) return rc;
    hipLaunchKernelGGL(bag_noisy_or_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, c->stream, d_p, B, bag, dS);
    HIPCHK(c, hipGetLastError());
    if (!dev) {
        HIPCHK(c, hipMemcpyAsync(site, dS, (size_t)B * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return M6A_OK;
}

int64_t m6a_flush_groups(int64_t S, int64_t bs, int64_t spb, int64_t *group_off, int64_t cap)
{
    if (S < 0 || bs < 1 || spb < 1 || !group_off) return M6A_EINVAL;
    std::vector<int64_t> g;
    const int64_t G = flush_groups(S, bs, spb, 0, g);
    if (cap < G + 1) return M6A_EINVAL;
    std::copy(g.begin(), g.end(), group_off);
    return G;
}

int64_t m6a_reference_written_sites(int64_t S, int64_t bs, int64_t spb)
{
    if (S < 0 || bs < 1 || spb < 1) return M6A_EINVAL;
    const int64_t nb = (S + bs - 1) / bs;
    for (int64_t b = nb - 1; b >= 0; b--)
        if ((b + 1) % spb) return std::min((b + 1) * bs, S);     // the last batch that flushes (inference_utils.py:47)
    return 0;
}

int m6a_shard_plan(const int64_t *off, int64_t S, int64_t bs, int64_t spb, int n_shards, int64_t *shard_off)
{
    if (!off || !shard_off || S < 0 || n_shards < 1 || bs < 1 || spb < 1) return M6A_EINVAL;
    std::vector<int64_t> g;
    const int64_t G = flush_groups(S, bs, spb, 0, g);
    const int64_t R = S > 0 ? off[S] : 0;
    shard_off[0] = 0;
    int64_t gi = 0;
    for (int k = 1; k <= n_shards; k++) {
        if (k == n_shards) { shard_off[k] = S; break; }
        // smallest group boundary whose read prefix reaches k/n of the reads
        const double target = (double)R * k / n_shards;
        while (gi < G && (double)off[g[gi]] < target) gi++;
        // pick the closer of the two neighbouring boundaries
        if (gi > 0 && gi <= G) {
            const double hi = (double)off[g[std::min(gi, G)]], lo = (double)off[g[gi - 1]];
            if (target - lo < hi - target && g[gi - 1] >= shard_off[k - 1]) gi--;
        }
        shard_off[k] = std::max(g[std::min(gi, G)], shard_off[k - 1]);
    }
    return M6A_OK;
}

int m6a_random_stream(m6a_ctx *c, uint32_t seed, int64_t n_words, uint32_t *words)
{
    settle(c);
    if (!c) return M6A_EINVAL;
    if (n_words < 0 || n_words > ((int64_t)1 << 31) - 2048) return fail(c, M6A_EINVAL, "n_words out of range");
    if (n_words == 0) return M6A_OK;
    if (!words) return fail(c, M6A_EINVAL, "null pointer argument");
    HIPCHK(c, hipSetDevice(c->device));
    if (is_device_ptr(words)) return launch_stream(c, seed, n_words, words);
    HIPCHK(c, c->val_idx.ensure((size_t)n_words * 4));
    int rc = launch_stream(c, seed, n_words, (uint32_t *)c->val_idx.p);
    if (rc) return rc;
    rc = d2h_through_ring(c, words, c->val_idx.p, (size_t)n_words * 4);
    if (rc) return rc;
    return sync_and_check(c);
}

int m6a_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int m6a_profile_enable(m6a_ctx *c, int on)
{
    if (!c) return M6A_EINVAL;
    if (on && !c->prof.d_clk) {
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipMalloc((void **)&c->prof.d_clk, (size_t)2 * M6A_CLK_SLOTS * 4 * sizeof(unsigned long long)));
    }
    if (on) HIPCHK(c, hipMemsetAsync(c->prof.d_clk, 0, (size_t)2 * M6A_CLK_SLOTS * 4 * sizeof(unsigned long long), c->stream));
    c->prof.on = on != 0;
    c->prof.mask = on == 2 ? 1 : on == 3 ? 2 : 3;
    c->prof.used[0] = c->prof.used[1] = 0;
    c->prof.dropped[0] = c->prof.dropped[1] = 0;
    return M6A_OK;
}

int m6a_profile_read(m6a_ctx *c, int kind, double *total_ms, int64_t *n_launches)
{
    settle(c);
    if (!c || kind < 0 || kind > 1 || !total_ms || !n_launches) return M6A_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double tot = 0;
    for (int i = 0; i < c->prof.used[kind]; i++) {
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, c->prof.start[kind][i], c->prof.stop[kind][i]));
        tot += ms;
    }
    *total_ms = tot;
    *n_launches = c->prof.used[kind];
    return M6A_OK;
}

int m6a_profile_clock(m6a_ctx *c, int kind, double *ghz_median, double *ghz_min, double *ghz_max, double *span_ms, int *n_waves)
{
    settle(c);
    if (!c || kind < 0 || kind > 1) return M6A_EINVAL;
    if (!c->prof.d_clk) return fail(c, M6A_EINVAL, "m6a_profile_clock: profiling was never enabled on this context");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    unsigned long long h[M6A_CLK_SLOTS * 4];
    HIPCHK(c, hipMemcpy(h, c->prof.d_clk + (size_t)kind * M6A_CLK_SLOTS * 4, sizeof h, hipMemcpyDeviceToHost));
    std::vector<double> ghz;
    unsigned long long r_first = ~0ull, r_last = 0;
    for (int i = 0; i < M6A_CLK_SLOTS; i++) {
        const unsigned long long c0 = h[4 * i], r0 = h[4 * i + 1], c1 = h[4 * i + 2], r1 = h[4 * i + 3];
        if (!r0 || r1 <= r0 || c1 <= c0) continue;                // slot unused, or its end stamp is an older launch's
        ghz.push_back((double)(c1 - c0) / (double)(r1 - r0) * 0.1);   // cycles per 10 ns tick
        r_first = std::min(r_first, r0);
        r_last = std::max(r_last, r1);
    }
    std::sort(ghz.begin(), ghz.end());
    const bool any = !ghz.empty();
    if (ghz_median) *ghz_median = any ? ghz[ghz.size() / 2] : 0.0;
    if (ghz_min) *ghz_min = any ? ghz.front() : 0.0;
    if (ghz_max) *ghz_max = any ? ghz.back() : 0.0;
    if (span_ms) *span_ms = any ? (double)(r_last - r_first) * 1e-5 : 0.0;
    if (n_waves) *n_waves = (int)ghz.size();
    return M6A_OK;
}

const char *m6a_last_pool_variant(const m6a_ctx *c) { return c ? c->pool_variant : "none"; }

}  // extern "C"
