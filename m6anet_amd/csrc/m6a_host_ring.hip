// m6a_host_ring.hip -- host-pointer calls: the pinned staging ring (split out of m6a_api.hip; internal declarations: m6a_ctx.h)
#include "m6a_ctx.h"

using namespace m6a_detail;

namespace m6a_detail {

// ---- host-pointer path: pinned staging ring, H2D of chunk k+1 under the encoder of chunk k ---------------------
void release_staging(m6a_ctx *c);

int ensure_staging(m6a_ctx *c)
{
    Staging &g = c->stg;
    if (g.ready) return M6A_OK;
    release_staging(c);                                       // whatever a failed earlier attempt left behind
    const char *env = getenv("M6A_STAGE_MB");
    const size_t slot_mb = env && atoi(env) > 0 ? (size_t)atoi(env) : 24;
    g.chunk_reads = (int64_t)(slot_mb << 20) / (M6A_N_FEATURES * 4);
    HIPCHK(c, hipStreamCreateWithFlags(&g.s_h2d, hipStreamNonBlocking));
    HIPCHK(c, hipStreamCreateWithFlags(&g.s_d2h, hipStreamNonBlocking));
    for (int i = 0; i < kStageSlots; i++) {
        HIPCHK(c, hipHostMalloc((void **)&g.pin_in[i], (size_t)g.chunk_reads * M6A_N_FEATURES * 4, hipHostMallocDefault));
        HIPCHK(c, hipHostMalloc((void **)&g.pin_out[i], (size_t)g.chunk_reads * 4, hipHostMallocDefault));
        HIPCHK(c, hipEventCreateWithFlags(&g.ev_h2d[i], hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&g.ev_enc[i], hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&g.ev_d2h[i], hipEventDisableTiming));
    }
    const char *et = getenv("M6A_COPY_THREADS");
    int nt = et && atoi(et) > 0 ? atoi(et) : std::min(16, std::max(2, m6a_usable_cpus()));
    g.pool.reset(new (std::nothrow) CopyPool(nt - 1));
    if (!g.pool) return fail(c, M6A_ENOMEM, "out of host memory");
    g.ready = true;
    return M6A_OK;
}

void release_staging(m6a_ctx *c)
{
    Staging &g = c->stg;
    g.pool.reset();
    for (int i = 0; i < kStageSlots; i++) {
        if (g.pin_in[i]) (void)hipHostFree(g.pin_in[i]);
        if (g.pin_out[i]) (void)hipHostFree(g.pin_out[i]);
        if (g.ev_h2d[i]) (void)hipEventDestroy(g.ev_h2d[i]);
        if (g.ev_enc[i]) (void)hipEventDestroy(g.ev_enc[i]);
        if (g.ev_d2h[i]) (void)hipEventDestroy(g.ev_d2h[i]);
    }
    if (g.s_h2d) (void)hipStreamDestroy(g.s_h2d);
    if (g.s_d2h) (void)hipStreamDestroy(g.s_d2h);
    g = Staging();
}

// Caller memory the DMA engines can address directly: page-locked by hipHostMalloc / hipHostRegister (torch's pin_memory()).
// Such buffers skip the staging copy in both directions -- the chunks are still cut and overlapped the same way.
// The WHOLE range [p, p + bytes) must lie inside ONE page-locked allocation (probing the first and the last byte
// accepts a buffer that starts in one registration and ends in another with pageable pages between, and the in-place DMA
// would then run over pageable memory): the runtime is asked for the allocation's base and size.  If this runtime cannot
// say (the range attributes fail), the buffer is treated as pageable -- the ring path is always correct.
bool is_pinned_host(const void *p, size_t bytes)
{
    if (!p || !bytes) return false;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (at.type != hipMemoryTypeHost) return false;
    void *base = nullptr;
    size_t size = 0;
    if (hipPointerGetAttribute(&base, HIP_POINTER_ATTRIBUTE_RANGE_START_ADDR, (hipDeviceptr_t)p) != hipSuccess ||
        hipPointerGetAttribute(&size, HIP_POINTER_ATTRIBUTE_RANGE_SIZE, (hipDeviceptr_t)p) != hipSuccess || !base || !size) {
        (void)hipGetLastError();
        return false;
    }
    // hipHostRegister'ed memory may report the DEVICE-side alias of the range: compare offsets inside the allocation, not addresses
    const char *host_base = (const char *)base;
    if ((const char *)p < host_base || (const char *)p >= host_base + size) {
        if (!at.devicePointer || !at.hostPointer) return false;
        host_base = (const char *)base - ((const char *)at.devicePointer - (const char *)at.hostPointer);
        if ((const char *)p < host_base || (const char *)p >= host_base + size) return false;
    }
    return (const char *)p + bytes <= host_base + size;
}

// An error exit of a host-pointer call that DMAs caller memory in place must not leave copies in flight: they read X and
// write rp_host, and a caller that recycles those buffers after the error (torch's pinned caching allocator does exactly
// that) would see stale D2H writes land in reused memory.  The pageable path never exposes caller memory to
// asynchronous DMA -- the ring's slots are the library's own.
struct DrainOnError {
    m6a_ctx *c;
    bool armed;
    ~DrainOnError()
    {
        if (!armed) return;
        Staging &g = c->stg;
        if (g.s_h2d) (void)hipStreamSynchronize(g.s_h2d);
        if (g.s_d2h) (void)hipStreamSynchronize(g.s_d2h);
        (void)hipStreamSynchronize(c->stream);
        (void)hipGetLastError();
    }
};

// Encodes a job whose X / site_kmers / off live in HOST memory: the job is cut at site boundaries into chunks of
// <= chunk_reads reads; chunk k is copied by the host threads into a pinned slot, DMA'd on its own stream and
// encoded on the context's stream while chunk k+1 is being copied; read probabilities flow back the same way
// (rp_host may be null).  On return every kernel is enqueued, sX/sK/sOff/sP hold the job on the device, and --
// if rp_host -- all read probabilities are in rp_host.  c->bag_min etc. describe `off` (host_bag_range ran).
int staged_encode(m6a_ctx *c, const float *X, const uint8_t *km, const int64_t *off, int64_t S, int64_t R, float *rp_host)
{
    Staging &g = c->stg;
    HIPCHK(c, c->sX.ensure((size_t)std::max<int64_t>(R, 1) * 9 * 4));
    HIPCHK(c, c->sK.ensure((size_t)S * 3));
    HIPCHK(c, c->sOff.ensure((size_t)(S + 1) * 8));
    HIPCHK(c, c->sP.ensure((size_t)std::max<int64_t>(R, 1) * 4));
    // Jobs under ~200 MB of features are not worth SETTING UP the pinned ring (pinning its 80 MB costs 15-40 ms once
    // per context, a 72 MB job copies in 2.4 ms without it): plain copies, unless the ring already exists
    // (m6a_prepare_host_io, or an earlier large call).  Single bags larger than a slot take the plain path too.
    const bool small = !g.ready && (size_t)R * 9 * 4 < ((size_t)192 << 20);
    int rc = small ? M6A_OK : ensure_staging(c);
    if (rc) return rc;
    if (small || R == 0 || c->bag_max > g.chunk_reads) {
        HIPCHK(c, hipMemcpyAsync(c->sK.p, km, (size_t)S * 3, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->sOff.p, off, (size_t)(S + 1) * 8, hipMemcpyHostToDevice, c->stream));
        if (R == 0) return M6A_OK;
        HIPCHK(c, hipMemcpyAsync(c->sX.p, X, (size_t)R * 9 * 4, hipMemcpyHostToDevice, c->stream));
        rc = launch_encode(c, (const float *)c->sX.p, (const uint8_t *)c->sK.p, (const int64_t *)c->sOff.p, S, R, (float *)c->sP.p);
        if (rc) return rc;
        if (rp_host) HIPCHK(c, hipMemcpyAsync(rp_host, c->sP.p, (size_t)R * 4, hipMemcpyDeviceToHost, c->stream));
        return M6A_OK;
    }
    const size_t slot_bytes = (size_t)g.chunk_reads * 9 * 4;
    // pinned caller buffers (the whole range inside one page-locked allocation) go straight onto the link
    const bool x_pinned = is_pinned_host(X, (size_t)R * 9 * 4);
    const bool rp_pinned = rp_host && is_pinned_host(rp_host, (size_t)R * 4);
    DrainOnError drain{c, x_pinned || rp_pinned};             // every `return rc` / HIPCHK exit below waits for the in-place DMA first
    // ring item 0: the CSR offsets and the k-mer ids, through a pinned slot like everything else
    const size_t off_bytes = (size_t)(S + 1) * 8, km_bytes = (size_t)S * 3;
    int item = 0;
    if (off_bytes + km_bytes <= slot_bytes) {
        g.pool->copy(g.pin_in[0], off, off_bytes);
        std::memcpy(g.pin_in[0] + off_bytes, km, km_bytes);
        HIPCHK(c, hipMemcpyAsync(c->sOff.p, g.pin_in[0], off_bytes, hipMemcpyHostToDevice, g.s_h2d));
        HIPCHK(c, hipMemcpyAsync(c->sK.p, g.pin_in[0] + off_bytes, km_bytes, hipMemcpyHostToDevice, g.s_h2d));
        HIPCHK(c, hipEventRecord(g.ev_h2d[0], g.s_h2d));
        HIPCHK(c, hipStreamWaitEvent(c->stream, g.ev_h2d[0], 0));
        item = 1;
    } else {
        HIPCHK(c, hipMemcpyAsync(c->sK.p, km, km_bytes, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->sOff.p, off, off_bytes, hipMemcpyHostToDevice, c->stream));
    }
    const int item0 = item;
    // chunk table: sites [cs[k], cs[k+1])
    std::vector<int64_t> cs{0};
    while (cs.back() < S) {
        const int64_t s0 = cs.back();
        int64_t s1 = std::upper_bound(off + s0, off + S + 1, off[s0] + g.chunk_reads) - off - 1;
        s1 = std::min<int64_t>(S, std::max<int64_t>(s1, s0 + 1));
        cs.push_back(s1);
    }
    const int64_t nchunk = (int64_t)cs.size() - 1;
    HIPCHK(c, c->sOffChunk.ensure((size_t)(S + nchunk) * 8));
    std::vector<char> out_pending((size_t)nchunk, 0);
    auto drain_out = [&](int64_t k) -> int {     // read probabilities of chunk k: pinned slot -> caller memory
        if (!out_pending[(size_t)k]) return M6A_OK;
        const int slot = (int)((k + item0) % kStageSlots);
        HIPCHK(c, hipEventSynchronize(g.ev_d2h[slot]));
        g.pool->copy(rp_host + off[cs[k]], g.pin_out[slot], (size_t)(off[cs[k + 1]] - off[cs[k]]) * 4);
        out_pending[(size_t)k] = 0;
        return M6A_OK;
    };
    for (int64_t k = 0; k < nchunk; k++, item++) {
        const int slot = item % kStageSlots;
        const int64_t s0 = cs[k], s1 = cs[k + 1], r0 = off[s0], nr = off[s1] - r0;
        if (nr == 0) continue;
        if (item >= kStageSlots) {
            if (!x_pinned) HIPCHK(c, hipEventSynchronize(g.ev_h2d[slot]));   // the slot's previous DMA has left it
            if (k >= kStageSlots) { rc = drain_out(k - kStageSlots); if (rc) return rc; }
        }
        const float *src = X + r0 * 9;
        if (!x_pinned) { g.pool->copy(g.pin_in[slot], src, (size_t)nr * 9 * 4); src = (const float *)g.pin_in[slot]; }
        HIPCHK(c, hipMemcpyAsync((float *)c->sX.p + r0 * 9, src, (size_t)nr * 9 * 4, hipMemcpyHostToDevice, g.s_h2d));
        HIPCHK(c, hipEventRecord(g.ev_h2d[slot], g.s_h2d));
        HIPCHK(c, hipStreamWaitEvent(c->stream, g.ev_h2d[slot], 0));
        // the encoder wants offsets that start at 0: the chunk's own CSR row
        int64_t *d_off = (int64_t *)c->sOffChunk.p + s0 + k;
        hipLaunchKernelGGL(rebase_off_kernel, dim3((unsigned)((s1 - s0 + 1 + 255) / 256)), dim3(256), 0, c->stream,
                           (const int64_t *)c->sOff.p + s0, s1 - s0 + 1, d_off);
        rc = launch_encode(c, (const float *)c->sX.p + r0 * 9, (const uint8_t *)c->sK.p + s0 * 3, d_off, s1 - s0, nr, (float *)c->sP.p + r0);
        if (rc) return rc;
        if (rp_host) {
            HIPCHK(c, hipEventRecord(g.ev_enc[slot], c->stream));
            HIPCHK(c, hipStreamWaitEvent(g.s_d2h, g.ev_enc[slot], 0));
            HIPCHK(c, hipMemcpyAsync(rp_pinned ? rp_host + r0 : (float *)g.pin_out[slot], (const float *)c->sP.p + r0, (size_t)nr * 4,
                                     hipMemcpyDeviceToHost, g.s_d2h));
            HIPCHK(c, hipEventRecord(g.ev_d2h[slot], g.s_d2h));
            out_pending[(size_t)k] = rp_pinned ? 0 : 1;
        }
    }
    for (int64_t k = 0; k < nchunk; k++) { rc = drain_out(k); if (rc) return rc; }
    if (rp_pinned) HIPCHK(c, hipStreamSynchronize(g.s_d2h));         // the contract: the read probabilities are in rp_host on return
    // the caller may reuse X as soon as the call returns: its last chunk must have left it
    if (x_pinned) HIPCHK(c, hipStreamSynchronize(g.s_h2d));
    drain.armed = false;                                      // success: the two waits above are the contract, the encoders stay queued
    return M6A_OK;
}

// site_prob / mod_ratio of a host-pointer call: through the (now idle) pinned slots when they fit, so the caller's
// pageable arrays are filled by the copy threads instead of a staged synchronous hipMemcpy.  Synchronises the stream.
int staged_outputs(m6a_ctx *c, int64_t S, float *site, double *mod)
{
    Staging &g = c->stg;
    const size_t slot_bytes = g.ready ? (size_t)g.chunk_reads * 9 * 4 : 0;
    const bool direct = is_pinned_host(site, (size_t)S * 4) && is_pinned_host(mod, (size_t)S * 8);
    if (direct || (size_t)S * 8 > slot_bytes) {
        HIPCHK(c, hipMemcpyAsync(site, c->sSite.p, (size_t)S * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(mod, c->sMod.p, (size_t)S * 8, hipMemcpyDeviceToHost, c->stream));
        return sync_and_check(c);
    }
    HIPCHK(c, hipMemcpyAsync(g.pin_in[0], c->sSite.p, (size_t)S * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(g.pin_in[1], c->sMod.p, (size_t)S * 8, hipMemcpyDeviceToHost, c->stream));
    const int rc = sync_and_check(c);
    if (rc) return rc;
    g.pool->copy(site, g.pin_in[0], (size_t)S * 4);
    g.pool->copy(mod, g.pin_in[1], (size_t)S * 8);
    return M6A_OK;
}

// A large device array into the caller's pageable memory: DMA of piece k+1 into a pinned slot while the copy threads
// deliver piece k (a plain hipMemcpy from device to pageable memory runs at a third of the link).  Orders itself
// behind everything queued on the context's stream; returns when the data is in `host`.
int d2h_through_ring(m6a_ctx *c, void *host, const void *dev, size_t bytes)
{
    if (!bytes) return M6A_OK;
    Staging &g = c->stg;
    if (!g.ready || bytes < ((size_t)1 << 20)) {
        HIPCHK(c, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return M6A_OK;
    }
    const size_t piece = (size_t)g.chunk_reads * 4;              // bytes per pin_out slot
    HIPCHK(c, hipEventRecord(c->ev_main, c->stream));
    HIPCHK(c, hipStreamWaitEvent(g.s_d2h, c->ev_main, 0));
    const size_t np_ = (bytes + piece - 1) / piece;
    for (size_t k = 0; k < np_ + 1; k++) {                       // piece k-1 is delivered while piece k is on the link; slot k % 3 was piece k-3's
        if (k < np_) {
            const int slot = (int)(k % kStageSlots);
            const size_t a = k * piece, n = std::min(piece, bytes - a);
            HIPCHK(c, hipMemcpyAsync(g.pin_out[slot], (const char *)dev + a, n, hipMemcpyDeviceToHost, g.s_d2h));
            HIPCHK(c, hipEventRecord(g.ev_d2h[slot], g.s_d2h));
        }
        if (k >= 1) {
            const size_t q = k - 1;
            const int slot = (int)(q % kStageSlots);
            const size_t a = q * piece, n = std::min(piece, bytes - a);
            HIPCHK(c, hipEventSynchronize(g.ev_d2h[slot]));
            g.pool->copy((char *)host + a, g.pin_out[slot], n);
        }
    }
    return M6A_OK;
}


}  // namespace m6a_detail

extern "C" int m6a_host_is_pinned(const void *p, size_t bytes)
{
    return m6a_detail::is_pinned_host(p, bytes) ? 1 : 0;
}
