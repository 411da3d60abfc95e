/*
 * m6a_io.h -- C ABI of libm6a_io.so: the data formats either side of the hot path
 * (SURVEY.md section 8(f) ranks 1-2).  Host-only C++ (no HIP): usable without a GPU.
 *
 *   m6a_io_load_sites   replaces NanopolishDS / NanopolishReplicateDS.__getitem__ + inference_collate
 *                       for a whole job (m6anet/utils/data_utils.py:118-129,152-231,341-427,498-506):
 *                       data.info + data.json -> the flat arrays include/m6a.h takes.
 *   m6a_io_write_csv    replaces the row formatting of run_inference
 *                       (m6anet/utils/inference_utils.py:59-67): data.site_proba.csv /
 *                       data.indiv_proba.csv, byte-identical to the Python '%' formatting.
 *
 * Every function returns 0 or a negative code; m6a_io_last_error() has the text (thread-local).
 */
#ifndef M6A_IO_H
#define M6A_IO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct m6a_sites m6a_sites;

enum { M6A_IO_OK = 0, M6A_IO_EINVAL = -1, M6A_IO_ENOMEM = -2, M6A_IO_EIO = -3, M6A_IO_EFORMAT = -4 };

const char *m6a_io_last_error(void);

/* input_dirs: n_dirs directories, each holding data.info + data.json (one = single sample, several
 * = replicates: union of sites in order of first appearance, read counts summed, reads concatenated
 * in directory order, read ids printed "<id>_<replicate>").  Sites with fewer than min_reads reads are
 * dropped (data_utils.py:129).  Normalisation factors (data_utils.py:233-248): n_norm 5-mers
 * (norm_kmers = n_norm x 5 chars, no terminators) with mean/std [n_norm][3] in float64; features are
 * (x - mean) / std in float64, then float32.  n_norm = 0 disables normalisation.  n_threads <= 0:
 * all hardware threads. */
int m6a_io_load_sites(const char *const *input_dirs, int n_dirs, int min_reads,
                      const char *norm_kmers, const double *norm_mean, const double *norm_std, int n_norm,
                      int n_threads, m6a_sites **out);
void m6a_io_free(m6a_sites *s);

int64_t m6a_io_n_sites(const m6a_sites *s);
int64_t m6a_io_n_reads(const m6a_sites *s);
int m6a_io_n_replicates(const m6a_sites *s);
const float *m6a_io_X(const m6a_sites *s);                 /* [R][9] */
const uint8_t *m6a_io_site_kmers(const m6a_sites *s);      /* [S][3] vocabulary ids */
const int64_t *m6a_io_off(const m6a_sites *s);             /* [S+1] */
const int64_t *m6a_io_tx_pos(const m6a_sites *s);          /* [S] */
const double *m6a_io_read_ids(const m6a_sites *s);         /* [R] numeric read index */
const int32_t *m6a_io_read_rep(const m6a_sites *s);        /* [R] replicate of each read */
const char *m6a_io_tx_id(const m6a_sites *s, int64_t site);    /* NUL-terminated */
const char *m6a_io_kmer5(const m6a_sites *s, int64_t site);    /* centre 5-mer, NUL-terminated */

/* Binary site store: everything m6a_io_load_sites produces -- normalised features,
 * k-mer ids, CSR offsets, ids -- in one file, so a dataset is parsed from data.json ONCE and every later run maps it
 * (replaces the per-item seek + json.loads + normalise of NanopolishDS.__getitem__, m6anet/utils/data_utils.py:
 * 152-231, written by m6anet/utils/dataprep_utils.py:473-485).  m6a_io_open_store maps the file read-only: the
 * arrays are views into the page cache, nothing is copied or parsed; the host-pointer path of m6a_infer streams X
 * from the mapping through its pinned ring.  `tag` (<= 63 chars) records what the features were normalised with;
 * m6a_io_store_tag returns it ("" for sites that came from m6a_io_load_sites). */
int m6a_io_save_store(const m6a_sites *s, const char *path, const char *tag);
int m6a_io_open_store(const char *path, m6a_sites **out);
const char *m6a_io_store_tag(const m6a_sites *s);

/* Appends the rows of all sites to <out_dir>/data.site_proba.csv and data.indiv_proba.csv
 * (headers are written when write_header != 0, truncating the files like
 * m6anet/scripts/inference.py:94-97).  read_prob [R], site_prob [S], mod_ratio [S]. */
int m6a_io_write_csv(const m6a_sites *s, const char *out_dir, const float *read_prob,
                     const float *site_prob, const double *mod_ratio, int write_header, int n_threads);
/* same, but only the first n_sites_limit sites (< 0: all): with m6a_reference_written_sites() (m6a.h) the files
 * hold exactly the rows the reference writes when its flush test leaves the last batches unwritten
 * (m6anet/utils/inference_utils.py:47) */
int m6a_io_write_csv_n(const m6a_sites *s, const char *out_dir, const float *read_prob,
                       const float *site_prob, const double *mod_ratio, int write_header, int n_threads,
                       int64_t n_sites_limit);

/* The same rows written by SEVERAL processes at once (`inference --gpus N`: every rank maps the same store and holds the
 * results of its own shard, so nobody has to collect the job's read probabilities -- 4 B per read -- just to print them):
 *   m6a_io_csv_shard_size    bytes the rows of sites [site_begin, site_end) take in each file (they are formatted and
 *                            counted; up to M6A_IO_CSV_KEEP_MB = 1024 MB of the text stays with the handle, and the
 *                            m6a_io_csv_shard_write that follows with the same range and arrays writes it instead of
 *                            formatting again); read_prob / site_prob / mod_ratio hold THAT range's values only;
 *   m6a_io_csv_shard_write   pwrite()s them at site_offset / indiv_offset (= header + the sizes of
 *                            all earlier shards, exchanged by the caller); the one rank with write_header != 0 also writes
 *                            both header lines and sets the files to their final sizes site_total / indiv_total
 *                            (< 0: leaves the size alone), which cuts whatever an earlier run left there;
 *   m6a_io_csv_header_bytes  length of the header line of data.site_proba.csv (which = 0) / data.indiv_proba.csv (1).
 * The bytes are those of m6a_io_write_csv over the whole job, whatever the cut (tests/test_host_io.py). */
int m6a_io_csv_shard_size(const m6a_sites *s, const float *read_prob, const float *site_prob, const double *mod_ratio,
                          int64_t site_begin, int64_t site_end, int n_threads, int64_t *site_bytes, int64_t *indiv_bytes);
int m6a_io_csv_shard_write(const m6a_sites *s, const char *out_dir, const float *read_prob, const float *site_prob,
                           const double *mod_ratio, int64_t site_begin, int64_t site_end, int n_threads,
                           int64_t site_offset, int64_t indiv_offset, int write_header, int64_t site_total, int64_t indiv_total);
int64_t m6a_io_csv_header_bytes(int which);

/* The writers' '%.16f' (inference_utils.py:62,66) without printf: same characters as snprintf("%.16f", v) for every
 * double (exact 128-bit arithmetic for 0 <= v < 2, snprintf itself otherwise); buf336 holds >= 336 bytes, NUL-terminated;
 * returns the length.  Exported so the tests can pin it against printf. */
int m6a_io_format_f16(double v, char *buf336);

/* How dataprep prints a float into data.json: Python's repr(float) -- what ujson / json.dump write in the reference
 * (dataprep_utils.py:473-480) -- shortest digits that round-trip, positional for 1e-4 <= |v| < 1e16, exponent form
 * otherwise.  buf40 holds >= 40 bytes, NUL-terminated; returns the length.  Exported so the tests can pin it against
 * Python's own repr. */
int m6a_io_py_repr(double v, char *buf40);
/* The writer's fast path for values np.round produced -- the mean (one decimal, dataprep_utils.py:293) and, with --compress,
 * every feature (three decimals): k / 10^digits written positionally with trailing zeros dropped, which IS repr() of such a
 * double for 0 < |v| < 1e9; returns the length, or -1 where the fast path declines (zero, huge, not the double nearest to
 * k / 10^digits, digits other than 1 or 3) and m6a_io_py_repr's general algorithm is used.  Exported for the tests. */
int m6a_io_repr_rounded(double v, int digits, char *buf40);

/* `m6anet dataprep` (m6anet/scripts/dataprep.py:54-70 -> m6anet/utils/dataprep_utils.py):
 * eventalign.txt -> <out_dir>/eventalign.index (parallel_index, :187-266), data.json + data.info +
 * data.log (combine :269-325, filter_events :19-168, preprocess_tx :399-488).  Same arithmetic as
 * the reference on pandas >= 1.3 (Kahan-compensated group sums, np.round half-to-even, floats
 * printed as Python repr) and the reference's n_processes = 1 record order (transcripts in index
 * order, positions ascending); inside a position reads stay in index order (the reference's order
 * there comes from an unstable argsort and is machine-dependent).  n_neighbors = 1..16 flanking positions either side
 * (m6anet/scripts/dataprep.py:45-47; roll / partition_into_continuous_positions, dataprep_utils.py:51-67,117-147): rows
 * of 3 (2 n + 1) features and a (5 + 2 n)-mer; the shipped models take n_neighbors = 1.
 * skip_index != 0 reads an existing eventalign.index instead of rebuilding it.
 * Built for files of tens to hundreds of GB: the index is computed over byte ranges on all threads and stitched where a
 * read crosses a range boundary (one index row = one contiguous (contig, read_index) run; the reference sums the line
 * lengths of all rows of a key per chunk instead, which only differs for reads whose lines are not contiguous --
 * tests/golden/dataprep_noncontiguous pins that divergence); files over 2 GB (M6A_IO_POPULATE_MAX_MB) are mapped
 * lazily; a transcript's records are written as soon as every earlier transcript's are, so memory holds the index
 * (32 B per read) and at most M6A_IO_PENDING_MB (256) of finished-but-unwritten records, never the whole data.json.  M6A_IO_TRACE=1 prints
 * the phases. */
int m6a_io_dataprep(const char *eventalign_path, const char *out_dir, int n_threads,
                    int readcount_min, int readcount_max, int min_segment_count, int n_neighbors,
                    int compress, int skip_index);

#ifdef __cplusplus
}
#endif
#endif
