/*
 * m6a_oracle.c -- CPU restatement of m6anet's inference hot path (plain C, float32).
 * TEST INFRASTRUCTURE ONLY -- see m6a_oracle.h.  Parity status: PINNED by
 * tests/test_oracle_golden.py against vectors captured from the imported reference.
 *
 * Third-party arithmetic the reference delegates to, restated here from the published
 * algorithms (the reference pins torch==1.6.0 / numpy>=1.18.0 in setup.py:34,41):
 *   - MT19937 (Matsumoto & Nishimura 1998) with init_genrand seeding, which is what
 *     np.random.seed(int) does (called at m6anet/scripts/inference.py:86);
 *   - legacy RandomState.randint/choice bounded draw: masked rejection on 32-bit words;
 *   - NumPy's float32 pairwise summation (block 128, 8 accumulators) behind ndarray.mean();
 *   - torch's float32 Linear and eval-mode BatchNorm1d on the CPU, operation by operation (the reference runs them at
 *     m6anet/model/model_blocks/blocks.py:249-254 through torch.nn): pinned in the build container against torch's own
 *     intermediate tensors, every candidate order compared bit for bit (tools/emulate_encoder.py):
 *       Linear      : acc = 0; acc = fma(x[k], W[j][k], acc) for k = 0, 1, 2, ...; then acc + b[j]   (MKL sgemm, any batch >= 7 rows)
 *       BatchNorm1d : alpha = gamma * (1 / sqrt(var + eps)); beta = fma(-mean, alpha, bias); y -> fma(y, alpha, beta)
 *     Layers 1 and 2 below reproduce torch's values bit for bit; the 32 -> 1 layer follows the AVX-512 sgemv of the
 *     machine the captures were made on (gemv32 below), and the sigmoid's exp is Sleef's, restated (sleef_expf_u10 below).
 */
#include "m6a_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ MT19937 ------------ */
void m6a_or_mt_seed(m6a_or_mt *st, uint32_t seed)
{
    st->mt[0] = seed;
    for (int i = 1; i < 624; i++)
        st->mt[i] = 1812433253u * (st->mt[i - 1] ^ (st->mt[i - 1] >> 30)) + (uint32_t)i;
    st->pos = 624;
}

static void mt_regen(m6a_or_mt *st)
{
    uint32_t *mt = st->mt;
    for (int k = 0; k < 624; k++) {
        uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
        mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    st->pos = 0;
}

uint32_t m6a_or_mt_next(m6a_or_mt *st)
{
    if (st->pos >= 624) mt_regen(st);
    uint32_t y = st->mt[st->pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

void m6a_or_mt_fill(uint32_t seed, int64_t count, uint32_t *out)
{
    m6a_or_mt st;
    m6a_or_mt_seed(&st, seed);
    for (int64_t i = 0; i < count; i++) out[i] = m6a_or_mt_next(&st);
}

/* ------------------------------------------------- choice(n, count, replace=True) ------ */
/* m6anet/utils/inference_utils.py:85 -> RandomState.choice -> randint(0, n): for
 * rng = n-1: rng == 0 draws nothing; otherwise mask = smallest 2^k-1 >= rng and
 * `do v = next32() & mask; while (v > rng)`. */
void m6a_or_choice(m6a_or_mt *st, int64_t n, int64_t count, int32_t *out_idx)
{
    uint32_t rng = (uint32_t)(n - 1);
    if (rng == 0) { memset(out_idx, 0, (size_t)count * sizeof(int32_t)); return; }
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    for (int64_t i = 0; i < count; i++) {
        uint32_t v;
        do { v = m6a_or_mt_next(st) & mask; } while (v > rng);
        out_idx[i] = (int32_t)v;
    }
}

/* ------------------------------------------------------ NumPy pairwise float32 sum ----- */
float m6a_or_pairwise_sum_f32(const float *a, int64_t n)
{
    if (n < 8) {
        float res = 0.0f;
        for (int64_t i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        float r[8], res;
        int64_t i;
        for (i = 0; i < 8; i++) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += a[i + k];
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return m6a_or_pairwise_sum_f32(a, n2) + m6a_or_pairwise_sum_f32(a + n2, n - n2);
    }
}

/* --------------------------------------------------------------- read encoder ---------- */
/* weights blob layout (floats): E[66][2] | W1[150][15] | b1 | gamma | beta | mu | var (150 each)
 *                               | W2[32][150] | b2[32] | W3[32] | b3[1] */
enum { O_E = 0, O_W1 = 132, O_B1 = 2382, O_G = 2532, O_BE = 2682, O_MU = 2832, O_VAR = 2982,
       O_W2 = 3132, O_B2 = 7932, O_W3 = 7964, O_B3 = 7996 };

/* a1-a3: [X | emb(k0) emb(k1) emb(k2)] (m6anet/model/model_blocks/blocks.py:126,205,65);
 * a4: relu(BN_eval(x W1^T + b1)) (blocks.py:249-254, m6anet.toml:16-21);
 * a5: relu(h W2^T + b2) (m6anet.toml:23-28);
 * a6: sigmoid(h W3^T + b3) (pooling_blocks.py:52, called at inference_utils.py:37). */
typedef struct {
    float w1t[15][152];   /* W1 transposed: [k][j], so the j loop vectorises while every */
    float w2t[150][32];   /* output still sums its k terms strictly left to right        */
    float b1[152], alpha[152], shift[152], b2[32], w3[32], b3;
    const float *emb;
} enc_tables;

static void enc_prepare(const float *w, enc_tables *t)
{
    memset(t, 0, sizeof(*t));
    for (int j = 0; j < 150; j++) {
        for (int k = 0; k < 15; k++) t->w1t[k][j] = w[O_W1 + 15 * j + k];
        float invstd = 1.0f / sqrtf(w[O_VAR + j] + 1e-5f);
        t->alpha[j] = w[O_G + j] * invstd;
        t->shift[j] = fmaf(-w[O_MU + j], t->alpha[j], w[O_BE + j]);
        t->b1[j] = w[O_B1 + j];
    }
    for (int j = 0; j < 32; j++) {
        for (int k = 0; k < 150; k++) t->w2t[k][j] = w[O_W2 + 150 * j + k];
        t->b2[j] = w[O_B2 + j];
        t->w3[j] = w[O_W3 + j];
    }
    t->b3 = w[O_B3];
    t->emb = w + O_E;
}

/* Linear(32, 1) as the sgemv behind torch's addmm computes it in the build container (MKL 2024.2, AVX-512; rows in groups of
 * four -- every row of a batch of 20-read bags, all but the last 0..3 rows of any other batch, which take a path this does not
 * restate): found by probing the module itself with cancellation triples (2^24, 1, -2^24 on every triple of inputs) and then
 * checked bit for bit against the reference's logits (tests/golden/reference_layers.npz):
 *     s = x0*w0;   16 lanes of products k = 1 + l, lane 0 = fma(x1, w1, s);   butterfly l+8, l+4, l+2, l+1;
 *     the same for k = 17 + l (lane 15 empty), lane 0 = fma(x17, w17, sum so far).
 * This is one MKL kernel on one ISA, not m6anet's algorithm: the HIP kernels do NOT follow it.  The
 * oracle does, because it is pinned to captures made on that machine and stands in for the reference on the GPU box -- with
 * it (and Sleef's exp), the oracle's read probabilities ARE the capture's, bit for bit, on every read of a 20-read-bag job. */
static inline float butterfly16(float *v)
{
    for (int step = 8; step >= 1; step >>= 1)
        for (int l = 0; l < step; l++) v[l] = v[l] + v[l + step];
    return v[0];
}

static inline float gemv32(const float *x, const float *w)
{
    float v[16];
    const float s0 = x[0] * w[0];
    for (int l = 0; l < 16; l++) v[l] = x[1 + l] * w[1 + l];
    v[0] = fmaf(x[1], w[1], s0);
    const float s1 = butterfly16(v);
    for (int l = 0; l < 15; l++) v[l] = x[17 + l] * w[17 + l];
    v[15] = 0.0f;
    v[0] = fmaf(x[17], w[17], s1);
    return butterfly16(v);
}

/* exp as torch's vectorised sigmoid computes it: Sleef's expf with 1.0-ulp bound (Sleef_expf16_u10 / Sleef_expf8_u10,
 * sleefsimdsp.c `xexpf`, Boost licence; restated from the published algorithm: Cody-Waite reduction by ln 2 in two parts, a
 * degree-6 polynomial in fma form, scaling by 2^q in two factors).  Checked in the build container against torch.sigmoid on
 * 2 M random floats in [-40, 40]: 99.9997 % identical. */
static inline float sleef_expf_u10(float d)
{
    const float R_LN2f = 1.442695040888963407359924681001892137426645954152985934135449406931f;
    const float L2Uf = 0.693145751953125f, L2Lf = 1.428606765330187045e-06f;
    if (d != d) return d;                                    /* NaN in, NaN out (and no float -> int conversion of a NaN) */
    if (d < -104.0f) return 0.0f;                            /* Sleef's own clips, taken first: they also bound q */
    if (d > 104.0f) return INFINITY;
    const int q = (int)rintf(d * R_LN2f);                    /* |d| is clipped below: q stays far inside int */
    float s = fmaf((float)q, -L2Uf, d);
    s = fmaf((float)q, -L2Lf, s);
    float u = 0.000198527617612853646278381f;
    u = fmaf(u, s, 0.00139304355252534151077271f);
    u = fmaf(u, s, 0.00833336077630519866943359f);
    u = fmaf(u, s, 0.0416664853692054748535156f);
    u = fmaf(u, s, 0.166666671633720397949219f);
    u = fmaf(u, s, 0.5f);
    u = 1.0f + fmaf(s * s, u, s);
    const int q1 = q >> 1, q2 = q - q1;
    u = u * ldexpf(1.0f, q1) * ldexpf(1.0f, q2);
    return u;
}

/* fmaf is one instruction in the first two clones (any x86 with FMA3: the GPU boxes and the build container have it) and
 * libm's correctly rounded software fma in the last: same bits, ~35x slower */
__attribute__((target_clones("avx512f", "fma", "default")))
static void enc_range(const enc_tables *t, const float *X, const uint8_t *site_kmers,
                      const int64_t *off, int64_t s0, int64_t s1, float *read_prob, float *hidden, float *logit)
{
    for (int64_t s = s0; s < s1; s++) {
        float in[15], h1[152], h2[32];
        for (int c = 0; c < 3; c++) {
            int id = site_kmers[3 * s + c];
            in[9 + 2 * c] = t->emb[2 * id];
            in[10 + 2 * c] = t->emb[2 * id + 1];
        }
        for (int64_t r = off[s]; r < off[s + 1]; r++) {
            memcpy(in, X + 9 * r, 9 * sizeof(float));
            for (int j = 0; j < 152; j++) h1[j] = 0.0f;
            for (int k = 0; k < 15; k++) {
                const float xk = in[k];
                for (int j = 0; j < 152; j++) h1[j] = fmaf(xk, t->w1t[k][j], h1[j]);
            }
            for (int j = 0; j < 152; j++) {
                float a = fmaf(h1[j] + t->b1[j], t->alpha[j], t->shift[j]);
                h1[j] = a < 0.0f ? 0.0f : a;                 /* torch's relu keeps NaN (a > 0 ? a : 0 would not) */
            }
            for (int j = 0; j < 32; j++) h2[j] = 0.0f;
            for (int k = 0; k < 150; k++) {
                const float hk = h1[k];
                for (int j = 0; j < 32; j++) h2[j] = fmaf(hk, t->w2t[k][j], h2[j]);
            }
            for (int j = 0; j < 32; j++) {
                float a = h2[j] + t->b2[j];
                h2[j] = a < 0.0f ? 0.0f : a;
            }
            const float z = gemv32(h2, t->w3) + t->b3;
            if (hidden) memcpy(hidden + 32 * r, h2, 32 * sizeof(float));
            if (logit) logit[r] = z;
            read_prob[r] = 1.0f / (1.0f + sleef_expf_u10(-z));
        }
    }
}

typedef struct {
    const enc_tables *t; const float *X; const uint8_t *km; const int64_t *off;
    int64_t s0, s1; float *out;
} enc_job;

static void *enc_worker(void *arg)
{
    enc_job *j = (enc_job *)arg;
    enc_range(j->t, j->X, j->km, j->off, j->s0, j->s1, j->out, NULL, NULL);
    return NULL;
}

void m6a_or_encode_reads_mt(const float *w, const float *X, const uint8_t *site_kmers,
                            const int64_t *off, int64_t n_sites, int n_threads, float *read_prob)
{
    enc_tables *t = (enc_tables *)malloc(sizeof(enc_tables));
    enc_prepare(w, t);
    if (n_threads <= 1 || n_sites < 2 * n_threads) {
        enc_range(t, X, site_kmers, off, 0, n_sites, read_prob, NULL, NULL);
    } else {
        pthread_t *th = (pthread_t *)malloc((size_t)n_threads * sizeof(pthread_t));
        enc_job *jobs = (enc_job *)malloc((size_t)n_threads * sizeof(enc_job));
        /* contiguous site ranges balanced by read count */
        int64_t R = off[n_sites], s = 0;
        for (int i = 0; i < n_threads; i++) {
            int64_t target = R * (i + 1) / n_threads, e = s;
            while (e < n_sites && off[e + 1] <= target) e++;
            if (i == n_threads - 1) e = n_sites;
            jobs[i] = (enc_job){ t, X, site_kmers, off, s, e, read_prob };
            pthread_create(&th[i], NULL, enc_worker, &jobs[i]);
            s = e;
        }
        for (int i = 0; i < n_threads; i++) pthread_join(th[i], NULL);
        free(th); free(jobs);
    }
    free(t);
}

/* The same pass, also handing out what the reference's own tensors can be compared with: the read representation
 * (layer 2 after its ReLU; m6anet/model/model.py get_read_representation) and the logit in front of the Sigmoid
 * (pooling_blocks.py:52).  tests/test_oracle_golden.py holds the first to the reference's bits. */
void m6a_or_encode_layers(const float *w, const float *X, const uint8_t *site_kmers, const int64_t *off,
                          int64_t n_sites, float *read_prob, float *hidden, float *logit)
{
    enc_tables *t = (enc_tables *)malloc(sizeof(enc_tables));
    enc_prepare(w, t);
    enc_range(t, X, site_kmers, off, 0, n_sites, read_prob, hidden, logit);
    free(t);
}

void m6a_or_encode_reads(const float *w, const float *X, const uint8_t *site_kmers,
                         const int64_t *off, int64_t n_sites, float *read_prob)
{
    m6a_or_encode_reads_mt(w, X, site_kmers, off, n_sites, 1, read_prob);
}

/* --------------------------------------------------------------- site sampling --------- */
/* inference_utils.py:85-86:  idx = choice(n, T*K); reshape (T,K);
 * (1 - prod(1 - p[idx], axis=1)).mean()  -- all float32: sequential product left to right,
 * pairwise sum, then one float32 divide by T. */
float m6a_or_site_proba(m6a_or_mt *st, const float *p, int64_t n, int n_iters, int n_samples,
                        int32_t *idx, float *vals)
{
    /* a site without reads: np.random.choice raises ("a must be non-empty") and the reference never has one
     * (data_utils.py:129 keeps sites with >= 20 reads); the HIP path answers NaN and consumes no words */
    if (n <= 0) return NAN;
    m6a_or_choice(st, n, (int64_t)n_iters * n_samples, idx);
    for (int t = 0; t < n_iters; t++) {
        float prod = 1.0f;
        const int32_t *row = idx + (int64_t)t * n_samples;
        for (int k = 0; k < n_samples; k++) prod *= (1.0f - p[row[k]]);
        vals[t] = 1.0f - prod;
    }
    return m6a_or_pairwise_sum_f32(vals, n_iters) / (float)n_iters;
}

/* inference_utils.py:33,47: batches of `batch_size` sites; batch `it` closes a flush group
 * when (it+1) % save_per_batch != 0.  Batches after the last flush are never written by
 * the reference; here they form a final group so every site gets a value. */
int64_t m6a_or_flush_groups_at(int64_t n_sites, int64_t batch_size, int64_t save_per_batch,
                               int64_t first_site, int64_t *group_off)
{
    /* the sites are sites [first_site, first_site+n_sites) of a larger job: batch indices are
     * global (first_site is a multiple of batch_size that starts a group), offsets local */
    int64_t n_batches = (n_sites + batch_size - 1) / batch_size, g = 0, start_b = 0;
    const int64_t it0 = first_site / batch_size;
    group_off[0] = 0;
    for (int64_t it = 0; it < n_batches; it++) {
        if ((it0 + it + 1) % save_per_batch) {
            int64_t end = (it + 1) * batch_size;
            group_off[++g] = end < n_sites ? end : n_sites;
            start_b = it + 1;
        }
    }
    if (start_b < n_batches) group_off[++g] = n_sites;
    return g;
}

int64_t m6a_or_flush_groups(int64_t n_sites, int64_t batch_size, int64_t save_per_batch,
                            int64_t *group_off)
{
    return m6a_or_flush_groups_at(n_sites, batch_size, save_per_batch, 0, group_off);
}

/* Threads: flush groups are independent (each reseeds), so every thread owns ONE contiguous range of groups, cut once by
 * site count (a site costs n_iters * n_samples accepted draws whatever its bag size) -- no shared cursor, no lock; the
 * scratch of all threads is one block the caller allocated; each thread also counts its own sites' mod_ratio. */
typedef struct {
    const float *p; const int64_t *off; const int64_t *goff; int64_t g0, g1;
    int n_iters, n_samples; uint32_t seed; float thr; float *site; double *mod;
    int32_t *idx; float *vals;
} pool_job;

static void *pool_worker(void *arg)
{
    pool_job *j = (pool_job *)arg;
    for (int64_t g = j->g0; g < j->g1; g++) {
        /* every flush group's Pool worker starts from the parent's never-advanced state
         * (inference_utils.py:102-104 forks after inference.py:86 seeded): reseed. */
        m6a_or_mt st;
        m6a_or_mt_seed(&st, j->seed);
        for (int64_t s = j->goff[g]; s < j->goff[g + 1]; s++) {
            const int64_t n = j->off[s + 1] - j->off[s];
            j->site[s] = m6a_or_site_proba(&st, j->p + j->off[s], n, j->n_iters, j->n_samples, j->idx, j->vals);
            /* inference_utils.py:53: np.mean(x >= thr): float32 compare, float64 mean */
            int64_t c = 0;
            for (int64_t r = j->off[s]; r < j->off[s + 1]; r++) c += j->p[r] >= j->thr;
            j->mod[s] = n ? (double)c / (double)n : NAN;
        }
    }
    return NULL;
}

int m6a_or_site_pool(const float *read_prob, const int64_t *off, int64_t n_sites, int n_iters,
                     int n_samples, float thr, uint32_t seed, int64_t batch_size,
                     int64_t save_per_batch, int n_threads, float *site_prob, double *mod_ratio)
{
    return m6a_or_site_pool_at(read_prob, off, n_sites, n_iters, n_samples, thr, seed, batch_size,
                               save_per_batch, 0, n_threads, site_prob, mod_ratio);
}

int m6a_or_site_pool_at(const float *read_prob, const int64_t *off, int64_t n_sites, int n_iters,
                        int n_samples, float thr, uint32_t seed, int64_t batch_size,
                        int64_t save_per_batch, int64_t first_site, int n_threads,
                        float *site_prob, double *mod_ratio)
{
    if (n_sites <= 0) return 0;
    int64_t n_batches = (n_sites + batch_size - 1) / batch_size;
    int64_t *goff = (int64_t *)malloc((size_t)(n_batches + 2) * sizeof(int64_t));
    if (!goff) return -1;
    const int64_t G = m6a_or_flush_groups_at(n_sites, batch_size, save_per_batch, first_site, goff);
    int W = n_threads > 1 ? n_threads : 1;
    if ((int64_t)W > G) W = (int)G;
    const size_t n_idx = (size_t)n_iters * (size_t)n_samples, n_val = (size_t)n_iters;
    /* per-thread scratch, 64-byte strides so two threads never share a line */
    const size_t stride = ((n_idx * sizeof(int32_t) + n_val * sizeof(float) + 63) / 64 + 1) * 64;
    char *scratch = (char *)malloc(stride * (size_t)W + 64);
    pool_job *jobs = (pool_job *)malloc((size_t)W * sizeof(pool_job));
    pthread_t *th = (pthread_t *)malloc((size_t)W * sizeof(pthread_t));
    if (!scratch || !jobs || !th) { free(goff); free(scratch); free(jobs); free(th); return -1; }
    char *base = (char *)(((uintptr_t)scratch + 63) & ~(uintptr_t)63);
    int64_t g = 0;
    for (int i = 0; i < W; i++) {
        /* groups [g, e): up to the group whose last site reaches this thread's share of the sites */
        const int64_t target = n_sites * (i + 1) / W;
        int64_t e = g;
        while (e < G && goff[e + 1] <= target) e++;
        if (i == W - 1) e = G;
        char *b = base + stride * (size_t)i;
        jobs[i] = (pool_job){ read_prob, off, goff, g, e, n_iters, n_samples, seed, thr, site_prob, mod_ratio,
                              (int32_t *)b, (float *)(b + n_idx * sizeof(int32_t)) };
        g = e;
    }
    if (W == 1) {
        pool_worker(&jobs[0]);
    } else {
        for (int i = 0; i < W; i++) pthread_create(&th[i], NULL, pool_worker, &jobs[i]);
        for (int i = 0; i < W; i++) pthread_join(th[i], NULL);
    }
    free(th); free(jobs); free(scratch); free(goff);
    return 0;
}

void m6a_or_bag_noisy_or(const float *read_prob, int64_t n_bags, int bag, float *site_prob)
{
    for (int64_t b = 0; b < n_bags; b++) {
        float prod = 1.0f;
        for (int k = 0; k < bag; k++) prod *= (1.0f - read_prob[b * bag + k]);
        site_prob[b] = 1.0f - prod;
    }
}

/* ------------------------------------------- validation-style forward ----- */
/* data_utils.py:213-214 -> RandomState.choice(n, k, replace=False) -> permutation(n)[:k] -> legacy
 * shuffle: for i in reversed(range(1, n)): j = rk_interval(i); arr[i], arr[j] = arr[j], arr[i] */
int m6a_or_choice_noreplace(m6a_or_mt *st, int64_t n, int k, int32_t *perm, int32_t *out_idx)
{
    if (k > n) return -1;
    for (int64_t i = 0; i < n; i++) perm[i] = (int32_t)i;
    for (int64_t i = n - 1; i >= 1; i--) {
        uint32_t mask = (uint32_t)i, v;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        do { v = m6a_or_mt_next(st) & mask; } while (v > (uint32_t)i);
        const int32_t tmp = perm[i]; perm[i] = perm[v]; perm[v] = tmp;
    }
    for (int j = 0; j < k; j++) out_idx[j] = perm[j];
    return 0;
}

int m6a_or_validation_indices(uint32_t seed, const int64_t *off, int64_t n_sites, int n_iters, int k,
                              int32_t *idx)
{
    int64_t nmax = 0;
    for (int64_t s = 0; s < n_sites; s++) if (off[s + 1] - off[s] > nmax) nmax = off[s + 1] - off[s];
    int32_t *perm = (int32_t *)malloc((size_t)(nmax > 0 ? nmax : 1) * sizeof(int32_t));
    if (!perm) return -2;
    m6a_or_mt st;
    m6a_or_mt_seed(&st, seed);
    int rc = 0;
    for (int t = 0; t < n_iters && !rc; t++)
        for (int64_t s = 0; s < n_sites && !rc; s++)
            rc = m6a_or_choice_noreplace(&st, off[s + 1] - off[s], k, perm, idx + ((int64_t)t * n_sites + s) * k);
    free(perm);
    return rc;
}

int m6a_or_validate(const float *read_prob, const int64_t *off, int64_t n_sites, int n_iters, int k,
                    uint32_t seed, float *y_pred, float *y_pred_avg)
{
    int32_t *idx = (int32_t *)malloc((size_t)n_iters * (size_t)n_sites * (size_t)k * sizeof(int32_t) + 4);
    if (!idx) return -2;
    int rc = m6a_or_validation_indices(seed, off, n_sites, n_iters, k, idx);
    if (!rc) {
        for (int t = 0; t < n_iters; t++)
            for (int64_t s = 0; s < n_sites; s++) {
                const int32_t *row = idx + ((int64_t)t * n_sites + s) * k;
                float prod = 1.0f;
                for (int j = 0; j < k; j++) prod *= (1.0f - read_prob[off[s] + row[j]]);
                y_pred[(int64_t)t * n_sites + s] = 1.0f - prod;
            }
        if (y_pred_avg)
            for (int64_t s = 0; s < n_sites; s++) {
                float acc = 0.0f;                 /* np.mean(axis=0) of a C-contiguous (T,S) array: row by row */
                for (int t = 0; t < n_iters; t++) acc += y_pred[(int64_t)t * n_sites + s];
                y_pred_avg[s] = acc / (float)n_iters;
            }
    }
    free(idx);
    return rc;
}
