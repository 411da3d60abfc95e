// m6a_job.hip -- the streaming job (split out of m6a_api.hip; internal declarations: m6a_ctx.h)
#include "m6a_ctx.h"

using namespace m6a_detail;

namespace m6a_detail {

// ---- streaming job: the reference's batch loop (inference_utils.py:33-54) fed as the loader produces it ----------
// A device buffer that grows and KEEPS its contents (the job's read probabilities and CSR offsets: their final size
// is not known while batches arrive).  Growth is geometric, so a job pays for it O(log) times; every stream that may
// still be writing the old block is drained first.
int grow_keep(m6a_ctx *c, DevBuf &b, size_t used_bytes, size_t need_bytes)
{
    if (need_bytes <= b.cap) return M6A_OK;
    const size_t want = std::max(need_bytes + need_bytes / 8 + 256, b.cap * 2);
    void *np_ = nullptr;
    HIPCHK(c, hipMalloc(&np_, want));
    if (b.p && used_bytes) {
        if (c->stg.s_h2d) HIPCHK(c, hipStreamSynchronize(c->stg.s_h2d));
        HIPCHK(c, hipMemcpyAsync(np_, b.p, used_bytes, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (b.p) (void)hipFree(b.p);
    b.p = np_; b.cap = want;
    return M6A_OK;
}

int job_setup_ring(m6a_ctx *c)
{
    auto &j = c->job;
    int rc = ensure_staging(c);
    if (rc) return rc;
    if (j.n_sub) return M6A_OK;
    // (a previous attempt may have failed half way: start from nothing)
    for (auto e : j.ev_h2d) (void)hipEventDestroy(e);
    for (auto e : j.ev_enc) (void)hipEventDestroy(e);
    j.ev_h2d.clear(); j.ev_enc.clear(); j.pin.clear();
    Staging &g = c->stg;
    const size_t slot_bytes = (size_t)g.chunk_reads * M6A_N_FEATURES * 4;
    const int per_slot = slot_bytes >= ((size_t)20 << 20) ? 5 : slot_bytes >= ((size_t)8 << 20) ? 2 : 1;
    j.sub_bytes = (slot_bytes / (size_t)per_slot) & ~(size_t)4095;
    j.cap_sites = 4096;
    j.o_goff = (size_t)(j.cap_sites + 1) * 8;
    j.o_km = 2 * j.o_goff;
    j.o_x = (j.o_km + (size_t)j.cap_sites * 3 + 255) & ~(size_t)255;
    if (j.sub_bytes < j.o_x + ((size_t)1 << 16)) return fail(c, M6A_EINVAL, "M6A_STAGE_MB too small for the streaming ring");
    j.cap_reads = (int64_t)((j.sub_bytes - j.o_x) / (M6A_N_FEATURES * 4));
    for (int i = 0; i < kStageSlots; i++)
        for (int k = 0; k < per_slot; k++) j.pin.push_back(g.pin_in[i] + (size_t)k * j.sub_bytes);
    const int n = (int)j.pin.size();
    HIPCHK(c, c->jX.ensure((size_t)n * j.sub_bytes));
    for (int i = 0; i < n; i++) {
        hipEvent_t a = nullptr, b = nullptr;
        HIPCHK(c, hipEventCreateWithFlags(&a, hipEventDisableTiming));
        j.ev_h2d.push_back(a);
        HIPCHK(c, hipEventCreateWithFlags(&b, hipEventDisableTiming));
        j.ev_enc.push_back(b);
    }
    j.used.assign((size_t)n, 0);
    j.n_sub = n;
    return M6A_OK;
}

// the pinned sub-slot the next rows go into; its previous DMA (n_sub chunks ago) must have left it
int job_acquire(m6a_ctx *c, char **pin)
{
    auto &j = c->job;
    const int sub = (int)(j.item % j.n_sub);
    if (!j.cur_ready) {
        if (j.used[(size_t)sub]) HIPCHK(c, hipEventSynchronize(j.ev_h2d[(size_t)sub]));
        j.cur_ready = true;
    }
    *pin = j.pin[(size_t)sub];
    return M6A_OK;
}

// Sends the chunk being filled: offsets, k-mer ids and features cross PCIe on the copy stream while earlier chunks are
// being encoded; the encoder of this chunk is queued on the context's stream behind the copy.  dX / dK non-null: the
// chunk's features and k-mer ids are already on the device (a device-pointer feed), only the offsets travel.
int job_flush(m6a_ctx *c, const float *dX = nullptr, const uint8_t *dK = nullptr)
{
    auto &j = c->job;
    Staging &g = c->stg;
    if (j.fill_sites == 0) return M6A_OK;
    const int sub = (int)(j.item % j.n_sub);
    char *pin = j.pin[(size_t)sub];
    char *dev = (char *)c->jX.p + (size_t)sub * j.sub_bytes;
    const int64_t ns = j.fill_sites, nr = j.fill_reads, s0 = j.S - ns, r0 = j.R - nr;
    int rc = grow_keep(c, c->jP, (size_t)r0 * 4, (size_t)std::max<int64_t>(j.R, 1) * 4);
    if (rc) return rc;
    rc = grow_keep(c, c->jOff, (size_t)(s0 + 1) * 8, (size_t)(j.S + 1) * 8);
    if (rc) return rc;
    if (j.used[(size_t)sub]) HIPCHK(c, hipStreamWaitEvent(g.s_h2d, j.ev_enc[(size_t)sub], 0));   // the encoder that read this device sub-slot
    HIPCHK(c, hipMemcpyAsync(dev, pin, (size_t)(ns + 1) * 8, hipMemcpyHostToDevice, g.s_h2d));
    HIPCHK(c, hipMemcpyAsync((int64_t *)c->jOff.p + s0, pin + j.o_goff, (size_t)(ns + 1) * 8, hipMemcpyHostToDevice, g.s_h2d));
    if (!dX) {
        HIPCHK(c, hipMemcpyAsync(dev + j.o_km, pin + j.o_km, (size_t)ns * 3, hipMemcpyHostToDevice, g.s_h2d));
        if (nr) HIPCHK(c, hipMemcpyAsync(dev + j.o_x, pin + j.o_x, (size_t)nr * M6A_N_FEATURES * 4, hipMemcpyHostToDevice, g.s_h2d));
    }
    HIPCHK(c, hipEventRecord(j.ev_h2d[(size_t)sub], g.s_h2d));
    HIPCHK(c, hipStreamWaitEvent(c->stream, j.ev_h2d[(size_t)sub], 0));
    if (nr) {
        c->bag_min = j.fill_min; c->n_reads = nr;             // what launch_encode looks at (kernel choice)
        rc = launch_encode(c, dX ? dX : (const float *)(dev + j.o_x), dK ? dK : (const uint8_t *)(dev + j.o_km), (const int64_t *)dev, ns, nr,
                           (float *)c->jP.p + r0);
        if (rc) return rc;
    }
    HIPCHK(c, hipEventRecord(j.ev_enc[(size_t)sub], c->stream));
    j.used[(size_t)sub] = 1;
    j.item++; j.chunks++;
    j.fill_sites = 0; j.fill_reads = 0; j.fill_min = INT64_MAX; j.cur_ready = false;
    return M6A_OK;
}

// rows [i, i+k) of a batch join the chunk being filled: CSR offsets (chunk-local for the encoder, job-global for the
// pooling) are written into the pinned sub-slot, the job's host copy of off[] grows
void job_append_offsets(m6a_ctx *c, char *pin, const int64_t *off, int64_t i, int64_t k)
{
    auto &j = c->job;
    int64_t *ol = (int64_t *)pin, *og = (int64_t *)(pin + j.o_goff);
    if (j.fill_sites == 0) { ol[0] = 0; og[0] = j.R; }
    const int64_t base = off[i], lbase = j.fill_reads, gbase = j.R;
    int64_t mn = j.fill_min;
    for (int64_t t = 0; t < k; t++) {
        const int64_t e = off[i + t + 1] - base, n = off[i + t + 1] - off[i + t];
        ol[j.fill_sites + t + 1] = lbase + e;
        og[j.fill_sites + t + 1] = gbase + e;
        j.off.push_back(gbase + e);
        mn = n < mn ? n : mn;
    }
    j.fill_min = mn;
    const int64_t nr = off[i + k] - base;
    j.fill_sites += k; j.fill_reads += nr; j.S += k; j.R += nr;
}

int job_feed_impl(m6a_ctx *c, const float *X, const uint8_t *km, const int64_t *off, int64_t n, bool dev)
{
    auto &j = c->job;
    if (off[0] != 0) return fail(c, M6A_EINVAL, "off[0] must be 0");
    for (int64_t s = 0; s < n; s++) if (off[s + 1] < off[s]) return fail(c, M6A_EINVAL, "off[] must be non-decreasing");
    int rc;
    if (dev) {
        // features already on the device: every piece of <= cap_sites sites is a chunk of its own, read in place
        rc = job_flush(c);
        if (rc) return rc;
        for (int64_t i = 0; i < n;) {
            char *pin;
            rc = job_acquire(c, &pin);
            if (rc) return rc;
            const int64_t k = std::min<int64_t>(n - i, j.cap_sites);
            const int64_t r0 = off[i];
            job_append_offsets(c, pin, off, i, k);
            rc = job_flush(c, X + r0 * M6A_N_FEATURES, km + i * 3);
            if (rc) return rc;
            i += k;
        }
        return M6A_OK;
    }
    for (int64_t i = 0; i < n;) {
        char *pin;
        rc = job_acquire(c, &pin);
        if (rc) return rc;
        const int64_t room_s = j.cap_sites - j.fill_sites, room_r = j.cap_reads - j.fill_reads;
        const int64_t lim = std::min<int64_t>(n, i + room_s);
        // the most sites of the batch that still fit this chunk
        const int64_t k = (std::upper_bound(off + i, off + lim + 1, off[i] + room_r) - (off + i)) - 1;
        if (k <= 0) {
            if (j.fill_sites == 0)
                return fail(c, M6A_EUNSUPPORTED, "a site of %lld reads does not fit a streaming chunk (%lld reads): use m6a_infer",
                            (long long)(off[i + 1] - off[i]), (long long)j.cap_reads);
            rc = job_flush(c);
            if (rc) return rc;
            continue;
        }
        const int64_t r0 = off[i], nr = off[i + k] - r0;
        // a DataLoader-sized batch is one memcpy on the caller's thread (16 sites ~ 30 KB); megabyte batches use the copy threads
        c->stg.pool->copy(pin + j.o_x + (size_t)j.fill_reads * M6A_N_FEATURES * 4, X + r0 * M6A_N_FEATURES, (size_t)nr * M6A_N_FEATURES * 4);
        std::memcpy(pin + j.o_km + (size_t)j.fill_sites * 3, km + i * 3, (size_t)k * 3);
        job_append_offsets(c, pin, off, i, k);
        i += k;
        if (j.fill_sites == j.cap_sites || j.fill_reads == j.cap_reads) {
            rc = job_flush(c);
            if (rc) return rc;
        }
    }
    return M6A_OK;
}


}  // namespace m6a_detail

extern "C" {

int m6a_job_begin(m6a_ctx *c, int T, int K, float thr, uint32_t seed, int rng_mode, int64_t bs, int64_t spb,
                  int64_t expect_sites, int64_t expect_reads)
{
    settle(c);
    int rc = check_pool_args(c, 0, T, K, rng_mode, bs, spb);
    if (rc) return rc;
    HintScope hint_scope(c);
    rc = job_busy(c);
    if (rc) return rc;
    if (expect_sites < 0 || expect_reads < 0) return fail(c, M6A_EINVAL, "expected sizes must be >= 0 (0 = unknown)");
    if (!base_is_group_start(c->job_offset, bs, spb))
        return fail(c, M6A_EINVAL, "job offset %lld does not start a flush group for batch_size=%lld save_per_batch=%lld",
                    (long long)c->job_offset, (long long)bs, (long long)spb);
    HIPCHK(c, hipSetDevice(c->device));
    rc = job_setup_ring(c);
    if (rc) return rc;
    auto &j = c->job;
    j.T = T; j.K = K; j.thr = thr; j.seed = seed; j.bs = bs; j.spb = spb;
    try {
        j.off.clear();
        j.off.reserve((size_t)std::max<int64_t>(expect_sites, 1 << 16) + 1);
        j.off.push_back(0);
    } catch (const std::bad_alloc &) {
        return fail(c, M6A_ENOMEM, "out of host memory");
    }
    j.S = j.R = 0; j.item = 0; j.chunks = 0; j.failed = 0; j.failed_msg.clear();
    j.fill_sites = j.fill_reads = 0; j.fill_min = INT64_MAX; j.cur_ready = false;
    std::fill(j.used.begin(), j.used.end(), 0);
    // the ring may still carry DMAs of an earlier host-pointer call or job on other streams: start from idle
    HIPCHK(c, hipStreamSynchronize(c->stg.s_h2d));
    if (expect_reads) { rc = grow_keep(c, c->jP, 0, (size_t)expect_reads * 4); if (rc) return rc; }
    if (expect_sites) { rc = grow_keep(c, c->jOff, 0, (size_t)(expect_sites + 1) * 8); if (rc) return rc; }
    j.open = true;
    return M6A_OK;
}

int m6a_job_feed(m6a_ctx *c, const float *X, const uint8_t *km, const int64_t *off, int64_t n_sites)
{
    settle(c);
    if (!c) return M6A_EINVAL;
    auto &j = c->job;
    if (!j.open) return fail(c, M6A_EINVAL, "no streaming job is open (m6a_job_begin)");
    if (j.failed) { c->err = j.failed_msg; return j.failed; }        // the first failure, with its own text
    if (n_sites < 0) return fail(c, M6A_EINVAL, "n_sites < 0");
    if (n_sites == 0) return M6A_OK;
    if (!km || !off) return fail(c, M6A_EINVAL, "null pointer argument");
    if (is_device_ptr(off)) return fail(c, M6A_EINVAL, "m6a_job_feed takes off[] as a HOST pointer");
    if (!X && off[n_sites] != 0) return fail(c, M6A_EINVAL, "null pointer argument");
    const bool dev = is_device_ptr(km);
    // a batch without reads has no X to speak of (an empty tensor's pointer may be anything): the k-mer ids decide
    if (off[n_sites] != 0 && dev != is_device_ptr(X)) return fail(c, M6A_EINVAL, "X and site_kmers must be both host or both device pointers");
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    try {
        rc = job_feed_impl(c, X, km, off, n_sites, dev);
    } catch (const std::bad_alloc &) {                       // the job's host copy of off[] grows with every batch
        rc = fail(c, M6A_ENOMEM, "out of host memory");
    }
    if (rc) { j.failed = rc; j.failed_msg = c->err; }
    return rc;
}

int m6a_job_feed_collated(m6a_ctx *c, const float *features, const int64_t *kmers, const int64_t *n_reads, int64_t n_sites)
{
    settle(c);
    if (!c) return M6A_EINVAL;
    auto &j = c->job;
    if (!j.open) return fail(c, M6A_EINVAL, "no streaming job is open (m6a_job_begin)");
    if (j.failed) { c->err = j.failed_msg; return j.failed; }
    if (n_sites < 0) return fail(c, M6A_EINVAL, "n_sites < 0");
    if (n_sites == 0) return M6A_OK;
    if (!n_reads || !kmers) return fail(c, M6A_EINVAL, "null pointer argument");
    int rc;
    try {
        // inference_collate's layout -> the library's: CSR offsets from n_reads, one k-mer row per SITE from the per-read rows
        // (every read of a site repeats its site's three ids, data_utils.py:221-224: the first read's row is the site's)
        static thread_local std::vector<int64_t> off;
        static thread_local std::vector<uint8_t> km;
        off.resize((size_t)n_sites + 1);
        km.resize((size_t)n_sites * 3);
        off[0] = 0;
        for (int64_t s = 0; s < n_sites; s++) {
            if (n_reads[s] < 0) return fail(c, M6A_EINVAL, "n_reads[%lld] < 0", (long long)s);
            off[(size_t)s + 1] = off[(size_t)s] + n_reads[s];
        }
        const int64_t R = off[(size_t)n_sites];
        if (R > 0 && !features) return fail(c, M6A_EINVAL, "null pointer argument");
        if (R > 0 && is_device_ptr(features)) return fail(c, M6A_EINVAL, "m6a_job_feed_collated takes the collate's HOST tensors");
        for (int64_t s = 0; s < n_sites; s++)
            for (int q = 0; q < 3; q++) {
                // a site without reads has no row of its own in `kmers`: its ids are never used (id 0 stands in)
                const int64_t v = n_reads[s] > 0 ? kmers[off[(size_t)s] * 3 + q] : 0;
                if (v < 0 || v >= M6A_N_KMERS) return fail(c, M6A_EINVAL, "k-mer id %lld outside the %d-word vocabulary", (long long)v, M6A_N_KMERS);
                km[(size_t)s * 3 + (size_t)q] = (uint8_t)v;
            }
        HIPCHK(c, hipSetDevice(c->device));
        rc = job_feed_impl(c, features, km.data(), off.data(), n_sites, false);
    } catch (const std::bad_alloc &) {
        rc = fail(c, M6A_ENOMEM, "out of host memory");
    }
    if (rc) { j.failed = rc; j.failed_msg = c->err; }
    return rc;
}

int m6a_job_size(const m6a_ctx *c, int64_t *n_sites, int64_t *n_reads)
{
    if (!c) return M6A_EINVAL;
    if (n_sites) *n_sites = c->job.open ? c->job.S : 0;
    if (n_reads) *n_reads = c->job.open ? c->job.R : 0;
    return M6A_OK;
}

int m6a_job_abort(m6a_ctx *c)
{
    settle(c);
    if (!c) return M6A_EINVAL;
    if (!c->job.open) return M6A_OK;
    c->job.open = false;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->stg.s_h2d) HIPCHK(c, hipStreamSynchronize(c->stg.s_h2d));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return M6A_OK;
}

int m6a_job_end(m6a_ctx *c, float *rp, float *site, double *mod)
{
    settle(c);
    if (!c) return M6A_EINVAL;
    auto &j = c->job;
    if (!j.open) return fail(c, M6A_EINVAL, "no streaming job is open (m6a_job_begin)");
    HintScope hint_scope(c);
    HIPCHK(c, hipSetDevice(c->device));
    struct Closer { m6a_ctx *c; ~Closer() { (void)m6a_job_abort(c); } } closer{c};     // whatever happens, the job ends here
    if (j.failed) {
        const int rc_failed = j.failed;
        (void)m6a_job_abort(c);                                      // (may touch the error text)
        c->err = j.failed_msg;
        return rc_failed;
    }
    int rc = job_flush(c);
    if (rc) return rc;
    const int64_t S = j.S, R = j.R;
    if (S == 0) return M6A_OK;
    if (!site || !mod) return fail(c, M6A_EINVAL, "null pointer argument");
    const bool dev = is_device_ptr(site);
    if (dev != is_device_ptr(mod) || (rp && R > 0 && dev != is_device_ptr(rp)))
        return fail(c, M6A_EINVAL, "read_prob, site_prob, mod_ratio must be all host or all device pointers");
    host_bag_range(c, j.off.data(), S);
    float *d_site = site; double *d_mod = mod;
    if (!dev) {
        HIPCHK(c, c->sSite.ensure((size_t)S * 4));
        HIPCHK(c, c->sMod.ensure((size_t)S * 8));
        d_site = (float *)c->sSite.p; d_mod = (double *)c->sMod.p;
    }
    Prefault pf_rp, pf_out;
    if (!dev && rp) pf_rp.start(rp, (size_t)R * 4, 2);
    if (!dev) pf_out.start(mod, (size_t)S * 8, 1);
    rc = launch_pool(c, (const float *)c->jP.p, (const int64_t *)c->jOff.p, S, j.T, j.K, j.thr, j.seed, j.bs, j.spb, d_site, d_mod);
    if (rc) return rc;
    if (dev) {
        if (rp && R) HIPCHK(c, hipMemcpyAsync(rp, c->jP.p, (size_t)R * 4, hipMemcpyDeviceToDevice, c->stream));
        return sync_and_check(c);
    }
    if (rp && R) { rc = d2h_through_ring(c, rp, c->jP.p, (size_t)R * 4); if (rc) return rc; }
    return staged_outputs(c, S, site, mod);
}


}  // extern "C"
