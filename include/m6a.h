/*
 * m6a.h -- C ABI of libm6a_hip.so: m6anet's inference hot path on MI355X (gfx950).
 *
 * The reference (GoekeLab/m6anet) is pure Python and has no FFI; its seams for this path are
 * Python call sites.  Each entry point below names the reference interface it replaces
 * (paths relative to the reference checkout).  INTEGRATION.md shows the ctypes binding a
 * maintainer would add on the reference side.
 *
 * Conventions
 *   - every function returns M6A_OK (0) or a negative M6A_E* code; m6a_last_error(ctx) gives text;
 *   - the caller owns every buffer; the library keeps no input pointer after a call returns (one exception, stated
 *     there: DEVICE batches of a streaming job are read in place until m6a_job_end / m6a_job_abort);
 *   - data pointers may be HOST or DEVICE pointers (detected with hipPointerGetAttributes):
 *       all-host   -> the library stages through its own device buffers and the call is
 *                     synchronous (results are in the host buffers on return);
 *       all-device -> kernels are enqueued on the context's stream (m6a_set_stream); the
 *                     caller synchronises (m6a_sync or its own stream sync).  Entry points
 *                     read back a few bytes first (off[S]; min/max bag size to choose the
 *                     sampling kernel) and so block on the stream once or twice per call;
 *   - one ctx per (process, GPU, stream); calls on one ctx are not thread-safe;
 *   - no exceptions, no callbacks, no torch types cross this boundary.
 */
#ifndef M6A_H
#define M6A_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct m6a_ctx m6a_ctx;

enum {
    M6A_OK = 0,
    M6A_EINVAL = -1,     /* bad argument */
    M6A_ENOMEM = -2,     /* host or device allocation failed */
    M6A_EHIP = -3,       /* HIP runtime error (text in m6a_last_error) */
    M6A_ESTREAM = -4,    /* a flush group needs more MT19937 words than the stream cap allows */
    M6A_ENODEV = -5,     /* no usable gfx950 device */
    M6A_EUNSUPPORTED = -6
};

/* rng_mode of m6a_site_pool / m6a_infer / m6a_job_begin.  M6A_RNG_NUMPY is the ONLY value: exact replay of the reference's
 * NumPy MT19937 stream -- the only sampler the reference has (np.random.choice at m6anet/utils/inference_utils.py:85, seeded at
 * scripts/inference.py:86).  The parameter stays in the three signatures because SURVEY.md section 8(b) fixed them; any other
 * value is M6A_EINVAL (a bad argument, not a missing feature: no counter-based mode is planned -- it could not meet the 1e-5
 * bar against the reference's draws, and indices are table reads in both pooling kernels, so it would not be faster either;
 * DESIGN.md section 8). */
enum { M6A_RNG_NUMPY = 0 };

#define M6A_N_WEIGHTS 7997
#define M6A_N_FEATURES 9     /* [dwell, std, mean] x {-1,0,+1}: DeaggregateNanopolish, blocks.py:113 */
#define M6A_MAX_SAMPLES 64
#define M6A_N_KMERS 66       /* vocabulary of 5-mers the DRACH 7-mers are made of (constants.py:29-36): k-mer ids are 0..65 */

/* Flat float32 weight blob, in this order (state-dict keys of m6anet/model/model.py:40-69,
 * files under m6anet/model/model_states/):
 *   E[66][2] | W1[150][15] | b1[150] | bn_gamma[150] | bn_beta[150] | bn_mean[150] | bn_var[150]
 *   | W2[32][150] | b2[32] | W3[32] | b3[1]                                     = 7997 floats.
 * Replaces: MILModel(...).load_state_dict(torch.load(...)) at m6anet/scripts/inference.py:88-90.
 * `weights` is a host pointer.  The weights are laid out for the kernels here; BatchNorm (eval, eps 1e-5) becomes the
 * (alpha, beta) = (gamma / sqrt(var + eps), fma(-mean, alpha, bias)) pair torch forms, applied per read in the kernel as
 * torch applies it -- nothing is folded into W1: the encoder's float32 sums run in the reference's order.
 * M6A_EUNSUPPORTED for a layer-2 weight or bias of magnitude >= 2^63 (the kernels carry them scaled by 2^64). */
int m6a_create(m6a_ctx **out, const float *weights, size_t n_floats, int device_id);
void m6a_destroy(m6a_ctx *ctx);
const char *m6a_last_error(const m6a_ctx *ctx);   /* ctx may be NULL: last create error */

/* Use the caller's HIP stream (a hipStream_t, e.g. torch.cuda.current_stream().cuda_stream).
 * NULL selects the context's own stream. */
int m6a_set_stream(m6a_ctx *ctx, void *hip_stream);
/* Multi-GPU sharding: the sites passed to m6a_site_pool / m6a_infer on this context are sites
 * [first_site, first_site + n_sites) of a larger job.  Batch indices, hence flush groups and RNG
 * restarts (inference_utils.py:33,47), are counted from the job's first site, so results do not
 * depend on how the job is cut.  first_site must start a flush group (m6a_shard_plan returns
 * such cuts).  Default 0. */
int m6a_set_job_offset(m6a_ctx *ctx, int64_t first_site);
int m6a_sync(m6a_ctx *ctx);    /* waits for the stream; returns a deferred kernel-side error if any */
/* Optional, for calls with DEVICE pointers: hand over the host copy of the NEXT call's off[] (S+1 values; the loader that
 * built the CSR array has it -- the reference's collate keeps n_reads on the host, data_utils.py:498-506).  The bag
 * statistics that choose the kernels then come from this copy and the call neither reads anything back nor blocks on the
 * stream: consecutive calls queue back to back.  The device array is still checked against the copy, on the device; a
 * mismatch makes the results of that call undefined and is reported as a deferred M6A_EINVAL by the next m6a_sync.
 * One-shot: consumed by the next m6a_encode_reads / m6a_site_pool / m6a_infer / m6a_validate call on this context; the
 * pointer is read inside that call only.  NULL withdraws it. */
int m6a_set_host_offsets(m6a_ctx *ctx, const int64_t *off_host);
/* Optional: set up the host-pointer path now (pinned staging ring + copy threads; tens of milliseconds of page
 * pinning) instead of inside the first call that passes host buffers -- e.g. while the caller is still parsing
 * its input.  Host-buffer calls to m6a_infer / m6a_encode_reads cut the job into chunks and overlap the PCIe
 * transfer of chunk k+1 with the encoder of chunk k (the shape of the reference's batch loop,
 * m6anet/utils/inference_utils.py:33-41, without its per-batch synchronisation).  Pageable buffers pass through the ring
 * (copy threads + DMA); buffers the caller has page-locked (hipHostMalloc / hipHostRegister, torch's pin_memory()) are recognised
 * with hipPointerGetAttributes and DMA'd in place, chunk by chunk, in both directions -- same chunks, same kernels, same bits. */
int m6a_prepare_host_io(m6a_ctx *ctx);
/* Page-locked host memory (hipHostMalloc / hipHostFree) for callers with no other way to get it -- plain C, Python without
 * torch: buffers from m6a_host_alloc are DMA'd in place by the host-pointer calls above.  No context needed (the HIP runtime
 * must see a device).  Returns M6A_OK, M6A_ENOMEM or M6A_EHIP.  The reference's counterpart: DataLoader(pin_memory=...) is not
 * used by m6anet/utils/data_utils.py -- its batches are pageable. */
int m6a_host_alloc(size_t bytes, void **out);
int m6a_host_free(void *p);
/* 1 if the WHOLE range [p, p + bytes) lies inside ONE page-locked host allocation the DMA engines can address (hipHostMalloc,
 * hipHostRegister, torch's pin_memory(), m6a_host_alloc) -- exactly the test the host-pointer calls apply to decide between
 * in-place DMA and the staging ring; 0 otherwise (pageable memory, device memory, a range that leaves its allocation or spans
 * two registrations, no HIP runtime).  Never an error: a buffer that is not recognised simply goes through the ring. */
int m6a_host_is_pinned(const void *p, size_t bytes);

/* Read encoder.  Replaces, for one batch of sites,
 *     model.get_read_representation({'X','kmer'}) + model.pooling_filter.probability_layer(.)
 * (m6anet/utils/inference_utils.py:35-37 -> m6anet/model/model.py:85-97,
 *  m6anet/model/model_blocks/blocks.py:116-126,194-205,55-66,257-266, pooling_blocks.py:52).
 *   X          [R][9] float32, already z-normalised (data_utils.py:216-218)
 *   site_kmers [S][3] uint8 vocabulary ids 0..65 of the three 5-mers (stored once per site;
 *              the reference repeats them per read as int64, data_utils.py:223-224)
 *   off        [S+1] int64 CSR read offsets, off[0]=0, off[S]=R (the reference's n_reads vector,
 *              data_utils.py:499)
 *   read_prob  [R] float32 out */
int m6a_encode_reads(m6a_ctx *ctx, const float *X, const uint8_t *site_kmers, const int64_t *off,
                     int64_t n_sites, float *read_prob);

/* Site pooling.  Replaces calculate_site_proba(read_probs, n_iters, n_samples, n_processes) +
 * the mod_ratio line (m6anet/utils/inference_utils.py:53-54,74-104) for ALL flush groups of a
 * job at once, with the reference's n_processes=1 semantics: the NumPy MT19937 stream restarts
 * from `seed` at every flush group (groups follow batch_size/save_per_batch exactly as the
 * loop at inference_utils.py:33,47 forms them -- see m6a_flush_groups), sites inside a group
 * consume it sequentially, indices come from masked rejection, the 20-term product is float32
 * left-to-right, the mean over iterations is NumPy's float32 pairwise sum and one divide -- given the
 * same read probabilities the site probabilities are bit-identical to the reference's.
 *   site_prob [S] float32 out;  mod_ratio [S] float64 out = mean(read_prob >= thr)
 *   n_samples <= M6A_MAX_SAMPLES (the reference passes 20, inference_utils.py:54). */
int m6a_site_pool(m6a_ctx *ctx, const float *read_prob, const int64_t *off, int64_t n_sites,
                  int n_iters, int n_samples, float read_proba_threshold, uint32_t seed,
                  int rng_mode, int64_t batch_size, int64_t save_per_batch,
                  float *site_prob, double *mod_ratio);

/* Fused encode + pool (what run_inference does per job, inference_utils.py:14-71, minus text
 * I/O).  read_prob may be NULL when per-read output (data.indiv_proba.csv) is not wanted. */
int m6a_infer(m6a_ctx *ctx, const float *X, const uint8_t *site_kmers, const int64_t *off,
              int64_t n_sites, int n_iters, int n_samples, float read_proba_threshold,
              uint32_t seed, int rng_mode, int64_t batch_size, int64_t save_per_batch,
              float *read_prob, float *site_prob, double *mod_ratio);

/* Streaming form of m6a_infer: the batch loop of run_inference (m6anet/utils/inference_utils.py:33-54) fed as the
 * DataLoader produces it, so a reference-side binding keeps its loader and its loop.
 *   m6a_job_begin   the arguments calculate_site_proba / the flush test get per job (inference_utils.py:47,54);
 *                   expect_sites / expect_reads size the device arrays up front (0 = unknown: they grow);
 *                   m6a_set_job_offset applies as in m6a_infer.
 *   m6a_job_feed    one batch = the arguments of m6a_encode_reads (inference_utils.py:35-37): X [r][9], site_kmers [n][3],
 *                   off [n+1] batch-local CSR offsets (off[0] = 0).  ASYNCHRONOUS: host rows are copied into a pinned
 *                   ring (the buffers are the caller's again on return), cross PCIe while earlier chunks are being
 *                   encoded, and the encoder is queued per chunk of <= ~130 k reads.  X / site_kmers may be host or
 *                   device pointers (both of a kind).  DEVICE rows are NOT copied: the encoder queued on the context's
 *                   stream reads them in place, possibly after the call has returned -- they must stay allocated and
 *                   unmodified until m6a_job_end / m6a_job_abort returns, and whatever produced them must be ordered
 *                   before the context's stream (same stream, or completed).  off is ALWAYS a host pointer -- it is
 *                   the loader's n_reads vector (data_utils.py:499).  Batches of any size, in job order.
 *   m6a_job_end     pools all sites fed so far exactly as m6a_infer would (same flush groups, same random stream) and
 *                   delivers read_prob [R] (or NULL), site_prob [S], mod_ratio [S] -- all host (synchronous) or all
 *                   device (the call still synchronises).  Closes the job, also on error.  Results are bit-identical
 *                   to one m6a_infer over the concatenated batches whenever both pick the same encoder kernel:
 *                   always when every bag has >= 16 reads (datasets are filtered to >= 20, data_utils.py:129) or when
 *                   m6a_set_encoder_variant pins it (the choice is made per chunk here, per job there; the two
 *                   kernels agree to float32 rounding); site_prob / mod_ratio are in every case exactly
 *                   m6a_site_pool of the job's read probabilities.
 *   m6a_job_size    sites / reads fed so far;  m6a_job_abort drops an open job.
 * A feed that fails voids the job: later feeds and m6a_job_end return its code.  While a job is open the other
 * compute entry points of the context return M6A_EINVAL. */
int m6a_job_begin(m6a_ctx *ctx, int n_iters, int n_samples, float read_proba_threshold, uint32_t seed, int rng_mode,
                  int64_t batch_size, int64_t save_per_batch, int64_t expect_sites, int64_t expect_reads);
int m6a_job_feed(m6a_ctx *ctx, const float *X, const uint8_t *site_kmers, const int64_t *off, int64_t n_sites);
/* The same batch in the layout the reference's loader hands its loop -- inference_collate's first three tensors
 * (m6anet/utils/data_utils.py:498-506): features f32 [r][9], kmers int64 [r][3] (every read repeats its site's three
 * vocabulary ids, :221-224), n_reads int64 [n] -- so a reference-side binding passes three data_ptr()s and converts
 * nothing.  HOST pointers only; ids outside 0..65 are M6A_EINVAL. */
int m6a_job_feed_collated(m6a_ctx *ctx, const float *features, const int64_t *kmers, const int64_t *n_reads, int64_t n_sites);
int m6a_job_size(const m6a_ctx *ctx, int64_t *n_sites, int64_t *n_reads);
int m6a_job_end(m6a_ctx *ctx, float *read_prob, float *site_prob, double *mod_ratio);
int m6a_job_abort(m6a_ctx *ctx);

/* MILModel.forward on fixed-size bags (m6anet/model/model.py:155-164 ->
 * SigmoidProdPooling.forward, pooling_blocks.py:127-129): X [B*bag][9], site_kmers [B][3],
 * site_prob[b] = 1 - prod_k (1 - p[b*bag+k]) in float32, left to right. */
int m6a_bag_forward(m6a_ctx *ctx, const float *X, const uint8_t *site_kmers, int64_t n_bags,
                    int bag, float *site_prob);

/* validate() at DataLoader num_workers=0, shuffle=False (m6anet/utils/training_utils.py:213-268):
 * n_iters passes over the sites in order; each pass draws n_samples reads of every site WITHOUT
 * replacement -- np.random.choice(n, n_samples, replace=False), m6anet/utils/data_utils.py:213-214,
 * i.e. the legacy shuffle of arange(n), the stream seeded once with `seed` -- and predicts
 * y_pred[t][s] = 1 - prod_k (1 - p[read k of the sample]) (MILModel.forward, model.py:155-164).
 * y_pred [n_iters][n_sites]; y_pred_avg [n_sites] = np.mean(y_pred, axis=0) (training_utils.py:253)
 * or NULL.  Every bag needs >= n_samples reads (NumPy raises otherwise: M6A_EINVAL).  The sampler
 * is a sequential walk over one random stream and runs on the host; gathers, products and the
 * mean run on the GPU.  m6a_validate_pool starts from read probabilities, m6a_validate encodes
 * first (read_prob [R] or NULL).  Pointers: all host or all device, as everywhere. */
int m6a_validate_pool(m6a_ctx *ctx, const float *read_prob, const int64_t *off, int64_t n_sites,
                      int n_iters, int n_samples, uint32_t seed, float *y_pred, float *y_pred_avg);
int m6a_validate(m6a_ctx *ctx, const float *X, const uint8_t *site_kmers, const int64_t *off,
                 int64_t n_sites, int n_iters, int n_samples, uint32_t seed, float *read_prob,
                 float *y_pred, float *y_pred_avg);

/* Host-only helpers (no GPU, usable with ctx == NULL semantics: they take no ctx). */

/* Flush groups of the reference's loop (inference_utils.py:33,47): batches of `batch_size`
 * sites; batch `it` closes a group when (it+1) % save_per_batch != 0 (the reference's inverted
 * test).  Batches after the last flush -- which the reference silently never writes -- form a
 * final group here.  Writes group_off[0..G] and returns G, or M6A_EINVAL if cap < G+1. */
int64_t m6a_flush_groups(int64_t n_sites, int64_t batch_size, int64_t save_per_batch,
                         int64_t *group_off, int64_t cap);

/* Sites the REFERENCE writes for this geometry: everything up to its last flush.  Batches after it are computed
 * by nobody and never reach the CSVs there (an even batch count always loses the last batch at the default
 * save_per_batch = 2); the reference-compatible CLI mode (--drop_unflushed_tail) cuts its output here. */
int64_t m6a_reference_written_sites(int64_t n_sites, int64_t batch_size, int64_t save_per_batch);

/* Contiguous, flush-group-aligned site shards balanced by read count for n_shards GPUs
 * (sites are independent; aligning to groups keeps results independent of the GPU count).
 * off is a HOST pointer.  Writes shard_site_off[0..n_shards]. */
int m6a_shard_plan(const int64_t *off, int64_t n_sites, int64_t batch_size, int64_t save_per_batch,
                   int n_shards, int64_t *shard_site_off);

/* The job's one exchange: every rank's site_prob / mod_ratio to rank `dst` over RCCL
 * (xGMI inside a node) -- the counterpart of nothing in the reference, which has no multi-device path; it is what
 * a launcher calls after each rank ran m6a_infer on its shard (m6a_shard_plan, m6a_set_job_offset).
 *   m6a_comm_unique_id   rank 0 makes the 128-byte RCCL id (ncclGetUniqueId); the launcher hands it to every rank
 *                        by whatever transport it has (its process group, MPI, a file);
 *   m6a_comm_init        ncclCommInitRank on the context's device -- one communicator per context;
 *   m6a_gather           shard_site_off [world+1] (always a HOST array) are the cuts of m6a_shard_plan;
 *                        site_prob / mod_ratio hold this rank's shard_site_off[rank+1] - shard_site_off[rank] sites;
 *                        site_all / mod_all [shard_site_off[world]] are written on rank dst only (may be NULL
 *                        elsewhere).  ONE grouped send/recv exchange on the context's stream, ragged shards land at
 *                        their offsets without padding.  Data pointers all device (stream-ordered, the caller
 *                        synchronises with m6a_sync) or all host (staged through the context, synchronous), as everywhere;
 *   m6a_gather_reads     the same exchange for the per-read output (data.indiv_proba.csv needs it on the rank that
 *                        writes): read_prob holds this rank's shard_read_off[rank+1] - shard_read_off[rank] reads
 *                        (shard_read_off[r] = off[shard_site_off[r]] of the job's CSR offsets), read_all on dst;
 * RCCL is bound at run time (dlopen of librccl; M6A_RCCL_LIB names a specific copy): a process that never calls
 * these needs no RCCL.  M6A_EUNSUPPORTED if the library cannot be found. */
#define M6A_COMM_ID_BYTES 128
int m6a_comm_unique_id(void *id_out);
int m6a_comm_init(m6a_ctx *ctx, const void *unique_id, int rank, int world_size);
int m6a_gather(m6a_ctx *ctx, const float *site_prob, const double *mod_ratio, const int64_t *shard_site_off,
               int dst, float *site_all, double *mod_all);
int m6a_gather_reads(m6a_ctx *ctx, const float *read_prob, const int64_t *shard_read_off, int dst, float *read_all);
int m6a_comm_destroy(m6a_ctx *ctx);
/* What the communicator ITSELF reports, for a launcher or a benchmark line that has to certify an N-rank run instead of
 * trusting its own arguments: m6a_comm_count = ncclCommCount (the ranks RCCL formed the communicator with);
 * m6a_comm_info = ncclCommUserRank, ncclCommCuDevice (the HIP device the communicator is bound to) and ncclGetVersion
 * (e.g. 22707); any of the three pointers may be NULL.  m6a_device_link = how two visible HIP devices are connected:
 * link_type as hipExtGetLinkTypeAndHopCount reports it (hsa_amd_link_info_type_t: 2 = PCIe, 4 = xGMI), hop count,
 * and whether dev_a can map dev_b's memory (hipDeviceCanAccessPeer) -- what decides whether the gather's direct
 * send/recv writes travel over xGMI.  No reference counterpart (it has no multi-device path). */
int m6a_comm_count(m6a_ctx *ctx, int *ranks_seen);
int m6a_comm_info(m6a_ctx *ctx, int *rank, int *device, int *rccl_version);
int m6a_device_link(int dev_a, int dev_b, int *link_type, int *hops, int *peer_access);
/* HIP devices visible to this process (0 if none / no runtime): what a launcher sizes `world` against. */
int m6a_device_count(void);

/* The random stream the site sampling replays, by itself: the first n_words 32-bit outputs of NumPy's legacy generator
 * after np.random.seed(seed) (m6anet/scripts/inference.py:86) -- MT19937 seeded by init_genrand, the words
 * RandomState.bytes / randint consume (m6anet/utils/inference_utils.py:85 draws from it through np.random.choice).
 * Generated on the GPU in up to 32 parallel segments (GF(2) jump-ahead).  words: host or device pointer; a device-pointer
 * call is stream-ordered (m6a_sync).  Known-answer vectors: tests/golden/mt19937_seed*_first4096.u32. */
int m6a_random_stream(m6a_ctx *ctx, uint32_t seed, int64_t n_words, uint32_t *words);

/* Per-kernel timing with HIP events on the context's stream (bench.py's live roofline).
 * kind: 0 = read encoder, 1 = site pooling.  on: 0 off, 1 both kinds, 2 the encoder only, 3 the pooling only (two events per
 * timed launch cost ~5 us each on the stream).  m6a_profile_read synchronises the stream. */
int m6a_profile_enable(m6a_ctx *ctx, int on);
int m6a_profile_read(m6a_ctx *ctx, int kind, double *total_ms, int64_t *n_launches);
/* The shader clock the LAST profiled launch of `kind` ran at, from inside the kernel: while profiling is on, lane 0 of up to
 * 64 workgroups spread over the grid (and over the 8 XCDs) stamps s_memtime (shader cycles) and s_memrealtime (constant
 * 100 MHz) at its start and end; GHz = cycles / real time of each such wave.  bench.py prices the rooflines at this clock
 * beside the nominal 2.4 GHz (a kernel that makes the part clock down is not an issue-limited kernel).  span_ms = first start
 * to last end among the stamped waves (a cross-check of the HIP-event duration).  Any out pointer may be NULL; n_waves = 0
 * when nothing was stamped (no profiled launch yet; kernels without stamps: the scan and LDS-table pooling fallbacks).
 * Launches outside profiling carry a null stamp pointer. */
int m6a_profile_clock(m6a_ctx *ctx, int kind, double *ghz_median, double *ghz_min, double *ghz_max, double *span_ms, int *n_waves);
/* Which read-encoder kernel runs.  0 = auto (default) = the REFERENCE'S BITS on every input: the 16-slot kernels perform the
 * reference's float32 operations in the reference's order all the way (layer sums as fma chains over k then the bias, batch
 * norm as one fma, the 32 -> 1 layer in the capture machine's sgemv order, Sleef's expf: blocks.py:249-254, pooling_blocks.py:52
 * as torch executes them on an AVX-512 host) -- enc_site16_kernel when every bag has >= 16 reads and the job fits 32-bit
 * indices, enc_kernel otherwise; same operations, same bits.  This is what the CLI (`--encoder reference`, its default),
 * INTEGRATION.md's stub, bench.py's headline and every caller that sets nothing get: ONE kernel family for the product and
 * the quoted number.
 *   1 = the same, said explicitly;   3 = the 16-slot arithmetic behind enc_kernel's per-lane walk even where the scalar site
 *   chain would do (A/B and tests);
 *   2 = the 12-slot kernel, strictly (every bag must have >= 16 reads: a call that violates this reports M6A_EINVAL at the
 *   next sync);   4 = "fast": the 12-slot kernel where every bag has >= 16 reads, the 16-slot kernels elsewhere.
 * The 12-slot kernel (enc_csite_kernel) is OPT-IN: it issues 106 MFMAs per 32-read tile instead of 116 (2.03 vs 2.16 ms per
 * 20 M reads, 4-5 % per step) by adding a site's six embedding terms and b1 pre-summed and summing the 32 -> 1 layer in
 * register order -- two departures from the reference's order.  Its read probabilities are within the reference test's
 * rtol 1e-5 / atol 1e-8 (m6anet/tests/test_inference.py:32) on every fixture and on 218 M reads of the full-size configs; a
 * random fuzz finds a few reads in 10^10 beyond that bar -- 4 of 20 G at up to 1.005 x in round 5, none of 20 G and 4 of 14 G at up to
 * 1.08 x in round 6 -- all far inside north_star's absolute 1e-5.
 * The environment variable M6A_ENCODER=auto|reference|general16|csite12|walk16|fast preselects 0 / 0 / 1 / 2 / 3 / 4 in every
 * context the process creates; any other value makes m6a_create fail with M6A_EINVAL. */
int m6a_set_encoder_variant(m6a_ctx *ctx, int mode);
const char *m6a_last_encoder_variant(const m6a_ctx *ctx);   /* "general16" (auto, 1, 3) | "csite12" (2, 4 where it applies) */
/* The __global__ function the last encode launched: "enc_site16_kernel" (16 slots, scalar 32-bit site chain: every bag
 * >= 16 reads), "enc_kernel" (16 slots, per-lane 64-bit walk: any bags; mode 3 forces it), "enc_csite_kernel" (12 slots).
 * The two 16-slot kernels perform the same float32 operations: same bits. */
const char *m6a_last_encoder_kernel(const m6a_ctx *ctx);
/* Tuning knob for ragged bags: 0 = auto (default: per-bag-size index tables once the work seen pays for them, else
 * the stream-replaying scan kernels, their driver chosen by the parallelism on offer), 1 = scan, one wavefront per
 * flush group, 2 = scan, counting pass + one wavefront per site, 3 = index tables always (M6A_EUNSUPPORTED if a bag
 * exceeds 4096 reads).  Results are identical. */
int m6a_set_scan_driver(m6a_ctx *ctx, int mode);
/* Tuning knob for uniform bags (n <= 32, n_samples = 20): 0 = auto (default: the register kernel
 * where it applies), 1 = LDS-gather kernel, 2 = register kernel (bags in VGPRs, draws through the
 * VGPR index mode).  Results are identical. */
int m6a_set_table_variant(m6a_ctx *ctx, int mode);
/* pooling kernel variant used by the last pool/infer call:
 * "table-reg" | "table" (uniform bags) | "ragged-table" | "scan-group" | "scan-site" (ragged bags) */
const char *m6a_last_pool_variant(const m6a_ctx *ctx);

const char *m6a_version(void);

#ifdef __cplusplus
}
#endif
#endif
