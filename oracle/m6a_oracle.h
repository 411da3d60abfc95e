/*
 * m6a_oracle.h -- CPU restatement of m6anet's inference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the checker for the HIP path, never the product: only tests/,
 * __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load it.  The shipped
 * path (m6anet_amd/) never links, imports or falls back to anything in oracle/.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against
 * vectors captured from the imported reference (tests/golden/make_golden.py): read
 * probabilities for the 4 pretrained models, site probabilities of full `m6anet inference`
 * runs at n_processes=1, and NumPy's legacy MT19937 / RandomState.choice stream.
 *
 * Each function cites the reference lines (relative to the reference checkout) it restates.
 */
#ifndef M6A_ORACLE_H
#define M6A_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define M6A_OR_N_WEIGHTS 7997

/* MT19937 as NumPy's legacy RandomState uses it (np.random.seed(int) == init_genrand). */
typedef struct { uint32_t mt[624]; int pos; } m6a_or_mt;
void     m6a_or_mt_seed(m6a_or_mt *st, uint32_t seed);
uint32_t m6a_or_mt_next(m6a_or_mt *st);
/* fill `out[count]` with raw 32-bit outputs */
void     m6a_or_mt_fill(uint32_t seed, int64_t count, uint32_t *out);

/* RandomState.choice(n, count, replace=True) == legacy randint(0, n): masked rejection
 * over raw 32-bit words (m6anet/utils/inference_utils.py:85). */
void m6a_or_choice(m6a_or_mt *st, int64_t n, int64_t count, int32_t *out_idx);

/* NumPy float32 pairwise add.reduce (what ndarray.mean() uses). */
float m6a_or_pairwise_sum_f32(const float *a, int64_t n);

/* a1-a7, a6: read encoder.  weights = flat blob (order in include/m6a.h). */
void m6a_or_encode_reads(const float *weights, const float *X, const uint8_t *site_kmers,
                         const int64_t *off, int64_t n_sites, float *read_prob);

/* same, site ranges spread over n_threads host threads (results identical) */
void m6a_or_encode_reads_mt(const float *weights, const float *X, const uint8_t *site_kmers,
                            const int64_t *off, int64_t n_sites, int n_threads, float *read_prob);
/* as m6a_or_encode_reads, plus hidden[R][32] (the read representation: layer 2 after ReLU) and logit[R] */
void m6a_or_encode_layers(const float *weights, const float *X, const uint8_t *site_kmers, const int64_t *off,
                          int64_t n_sites, float *read_prob, float *hidden, float *logit);

/* a10: _calculate_site_proba for one site, consuming `st` (inference_utils.py:74-87). */
float m6a_or_site_proba(m6a_or_mt *st, const float *p, int64_t n, int n_iters, int n_samples,
                        int32_t *scratch_idx /* [n_iters*n_samples] */, float *scratch_f /* [n_iters] */);

/* flush-group boundaries (inference_utils.py:47, the inverted modulo); returns #groups,
 * writes group_off[0..G] (site indices).  group_off must hold n_batches+2 entries. */
int64_t m6a_or_flush_groups(int64_t n_sites, int64_t batch_size, int64_t save_per_batch,
                            int64_t *group_off);

/* same for a shard of a larger job: the sites are [first_site, first_site+n_sites) of the job;
 * first_site must be a multiple of batch_size that starts a flush group */
int64_t m6a_or_flush_groups_at(int64_t n_sites, int64_t batch_size, int64_t save_per_batch,
                               int64_t first_site, int64_t *group_off);

/* a10-a13 numerics for a whole job: reseed per flush group, sites sequential inside a group
 * (inference_utils.py:47-54,90-104 at n_processes=1), mod_ratio (inference_utils.py:53).
 * n_threads>1 runs flush groups concurrently (they are independent); results identical. */
int m6a_or_site_pool(const float *read_prob, const int64_t *off, int64_t n_sites, int n_iters,
                     int n_samples, float thr, uint32_t seed, int64_t batch_size,
                     int64_t save_per_batch, int n_threads, float *site_prob, double *mod_ratio);

int m6a_or_site_pool_at(const float *read_prob, const int64_t *off, int64_t n_sites, int n_iters,
                        int n_samples, float thr, uint32_t seed, int64_t batch_size,
                        int64_t save_per_batch, int64_t first_site, int n_threads,
                        float *site_prob, double *mod_ratio);

/* a8: SigmoidProdPooling.forward on fixed-size bags: 1 - prod(1-p) per bag of `bag` reads
 * (m6anet/model/model_blocks/pooling_blocks.py:127-129). */
void m6a_or_bag_noisy_or(const float *read_prob, int64_t n_bags, int bag, float *site_prob);

/* RandomState.choice(n, k, replace=False) == permutation(n)[:k]: legacy shuffle of arange(n), for
 * i = n-1 .. 1: j = rk_interval(i) (masked rejection over 32-bit words, mask = smallest 2^b-1 >= i),
 * swap(i, j) -- the training-mode read sampler, m6anet/utils/data_utils.py:213-214.
 * perm is scratch of n entries; returns -1 when k > n (NumPy raises ValueError). */
int m6a_or_choice_noreplace(m6a_or_mt *st, int64_t n, int k, int32_t *perm, int32_t *out_idx);

/* The sampler of a whole validation run at DataLoader num_workers=0, shuffle=False
 * (m6anet/utils/training_utils.py:235-240 over data_utils.py:213-214): the stream is seeded once,
 * pass after pass, site after site.  idx [n_iters][n_sites][k] = read index inside the site's bag. */
int m6a_or_validation_indices(uint32_t seed, const int64_t *off, int64_t n_sites, int n_iters, int k,
                              int32_t *idx);

/* validate()'s predictions (training_utils.py:233-250): y_pred[t][s] = 1 - prod_k (1 - p[off[s]+idx[t][s][k]])
 * (float32, left to right: MILModel.forward -> SigmoidProdPooling, pooling_blocks.py:127-129), and
 * y_pred_avg = np.mean(y_pred, axis=0): float32, pass after pass, then one divide. */
int m6a_or_validate(const float *read_prob, const int64_t *off, int64_t n_sites, int n_iters, int k,
                    uint32_t seed, float *y_pred, float *y_pred_avg);

#ifdef __cplusplus
}
#endif
#endif
