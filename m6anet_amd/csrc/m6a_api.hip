// m6a_api.hip -- host side of libm6a_hip.so: the C ABI of include/m6a.h.
// Context/weights management, MT19937 stream + index-table preparation, launches.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "m6a.h"
#include "m6a_kernels.h"
#include "m6a_host_cpus.h"

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

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

constexpr int kMaxProfiled = 8192;

struct Profiler {
    bool on = false;
    int mask = 3;                             // bit 0: time the encoder launches, bit 1: the pooling launches
    std::vector<hipEvent_t> start[2], stop[2];
    int used[2] = {0, 0};
    int64_t dropped[2] = {0, 0};
};

}  // namespace

// Host threads that move caller memory into / out of the pinned staging slots of the host-pointer path: one
// thread cannot feed PCIe (memcpy of pageable memory runs at 10-15 GB/s per thread, the link takes ~45 GB/s).
class CopyPool {
public:
    explicit CopyPool(int n_threads)
    {
        for (int i = 0; i < n_threads; i++) workers_.emplace_back([this] { loop(); });
    }
    ~CopyPool()
    {
        { std::lock_guard<std::mutex> g(mu_); stop_ = true; }
        cv_job_.notify_all();
        for (auto &t : workers_) t.join();
    }
    // memcpy(dst, src, n) split over the workers and the calling thread; returns when all of it is done
    void copy(void *dst, const void *src, size_t n)
    {
        const size_t parts = std::max<size_t>(1, std::min<size_t>(workers_.size() + 1, n / ((size_t)256 << 10)));
        if (parts == 1) { std::memcpy(dst, src, n); return; }
        const size_t slice = ((n + parts - 1) / parts + 4095) & ~(size_t)4095;
        {
            std::lock_guard<std::mutex> g(mu_);
            dst_ = (char *)dst; src_ = (const char *)src; n_ = n; slice_ = slice; next_ = 0; pending_ = (n + slice - 1) / slice;
        }
        cv_job_.notify_all();
        work();
        std::unique_lock<std::mutex> g(mu_);
        cv_done_.wait(g, [this] { return pending_ == 0; });
    }

private:
    bool work()                     // take slices until none is left; true if any was taken
    {
        bool any = false;
        for (;;) {
            size_t off;
            {
                std::lock_guard<std::mutex> g(mu_);
                if (next_ * slice_ >= n_) return any;
                off = next_++ * slice_;
            }
            std::memcpy(dst_ + off, src_ + off, std::min(slice_, n_ - off));
            any = true;
            std::lock_guard<std::mutex> g(mu_);
            if (--pending_ == 0) cv_done_.notify_all();
        }
    }
    void loop()
    {
        for (;;) {
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_job_.wait(g, [this] { return stop_ || next_ * slice_ < n_; });
                if (stop_) return;
            }
            work();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_job_, cv_done_;
    char *dst_ = nullptr; const char *src_ = nullptr;
    size_t n_ = 0, slice_ = 1, next_ = 0, pending_ = 0;
    bool stop_ = false;
};

// First touch of a caller's fresh output array, off the critical path: a call that returns 80 MB of read
// probabilities into a just-allocated buffer otherwise pays ~20 000 page faults (zeroing included) inside the
// copies that deliver the results.  A few threads walk the pages front to back while the first chunks are still
// crossing PCIe; the touch is an atomic add of zero, so a page that already holds results is left as it is.
class Prefault {
public:
    Prefault() = default;
    void start(void *p, size_t bytes, int n_threads)
    {
        if (!p || bytes < ((size_t)4 << 20)) return;
        char *b = (char *)p;
        const size_t page = 4096;
        const size_t first = (page - ((uintptr_t)b & (page - 1))) & (page - 1);
        if (first >= bytes) return;
        const size_t n_pages = (bytes - first) / page;
        for (int t = 0; t < n_threads; t++)
            th_.emplace_back([=] {
                // thread t takes the t-th contiguous part (the results arrive front to back, part 0 is needed first);
                // one madvise(MADV_POPULATE_WRITE) per 2 MB where the kernel has it (Linux >= 5.14), page touches otherwise
                const size_t p0 = n_pages * (size_t)t / (size_t)n_threads, p1 = n_pages * (size_t)(t + 1) / (size_t)n_threads;
                char *lo = b + first + p0 * page, *hi = b + first + p1 * page;
                bool populate = true;
                for (char *q = lo; q < hi;) {
                    const size_t len = std::min<size_t>((size_t)2 << 20, (size_t)(hi - q));
                    if (populate && madvise(q, len, 23 /* MADV_POPULATE_WRITE */) != 0) populate = false;
                    if (!populate)
                        for (size_t o = 0; o < len; o += page) __atomic_fetch_add(q + o, (char)0, __ATOMIC_RELAXED);
                    q += len;
                }
            });
    }
    void join() { for (auto &t : th_) t.join(); th_.clear(); }
    ~Prefault() { join(); }
private:
    std::vector<std::thread> th_;
};

// pinned staging ring of the host-pointer path (m6a_infer / m6a_encode_reads with host buffers)
constexpr int kStageSlots = 3;
struct Staging {
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    char *pin_in[kStageSlots] = {nullptr, nullptr, nullptr};
    char *pin_out[kStageSlots] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_h2d[kStageSlots] = {nullptr, nullptr, nullptr}, ev_enc[kStageSlots] = {nullptr, nullptr, nullptr},
               ev_d2h[kStageSlots] = {nullptr, nullptr, nullptr};
    int64_t chunk_reads = 0;
    std::unique_ptr<CopyPool> pool;
    bool ready = false;
};

// NumPy's float32 pairwise sum as a plan (built by build_mean_plan below)
struct MeanPlan {
    int T = 0;
    std::vector<int> leaf_start;        // [L+1]
    std::vector<uint8_t> merge_after;   // [L]
    // table kernel rows, 4 dwords each: flags (bit 24 pass ends here, bit 25 tail row, bits 26-29 leaves
    // in the pass, bit 30 last pass) | live-lane mask lo | hi | merge_after of the pass's leaves (nibbles)
    std::vector<uint32_t> row_meta;
    std::vector<int> row_pass, row_round;   // host side of the same rows (pass -1 = tail row)
    std::vector<uint32_t> reg_ctl;          // pool_reg_kernel: per round of 8 iterations, bit 0 = a leaf ends, bits 8.. = its merges
    int n_rem = 0;
    int depth = 1;                      // deepest the merge stack gets
    int max_merge = 0;                  // largest merge_after[]
};

struct m6a_ctx {
    int device = 0;
    int n_cu = 256;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    // model
    float *d_wfrag = nullptr, *d_wfrag2 = nullptr, *d_w1e = nullptr, *d_emb = nullptr;
    float b3 = 0.f;
    // sampling state (device) + what it was built for
    DevBuf raw, tab, tab_reg, goff, rp_scratch, off_scratch, start_pos, plan_dev;
    MeanPlan plan; size_t plan_off[4] = {0, 0, 0, 0};
    uint32_t raw_seed = 0; int64_t raw_len = 0;
    struct { uint32_t seed; int n, T, K, jmax; bool valid; } tab_key = {0, 0, 0, 0, 0, false}, tab_reg_key = {0, 0, 0, 0, 0, false};
    int table_variant = 0;                            // 0 auto, 1 LDS gather kernel, 2 register kernel
    struct { int64_t S, bs, spb, base, G, gmax; bool valid; } goff_key = {0, 0, 0, 0, 0, 0, false};
    int64_t job_offset = 0;
    int64_t bag_min = 0, bag_max = 0, n_reads = 0;   // last query_bags()
    int enc_variant = 0;                              // 0 auto, 1 general 16-slot, 2 12-slot (bags >= 16)
    const char *enc_variant_used = "none";
    int scan_driver = 0;                              // 0 auto, 1 per group, 2 counting pass + per site
    int *d_err = nullptr;
    unsigned long long *d_minmax = nullptr;
    unsigned long long *h_minmax = nullptr;   // pinned
    int *h_err = nullptr;                     // pinned
    // bag-size histogram of the last query_bags()/host_bag_range() (pinned; bins 0..M6A_RTAB_MAX_N, last = larger) and the
    // pinned staging of the small control arrays of the index-table path
    uint32_t *h_hist = nullptr, *d_hist = nullptr;
    std::vector<uint32_t> hist_part;         // host_bag_range: eight interleaved histograms
    uint32_t *h_ctl = nullptr;                // [cursor HIST_BINS | slot_of_n MAX_N+1 | build_n MAX_N | build_slot MAX_N]
    DevBuf ctl_dev, rt_rank, rt_order, reg_out;
    // m6a_infer runs the pooling's set-up on a side stream next to the encoder (pool_setup_aside)
    bool side_work = false;                   // something is queued on s_prep that the main stream does not wait for
    const int64_t *hint_off = nullptr;        // m6a_set_host_offsets: host copy of the next device call's off[]
    hipEvent_t ev_ctl = nullptr;              // the last upload from h_ctl (the host rewrites it per call)
    hipStream_t s_prep = nullptr;
    hipEvent_t ev_main = nullptr, ev_prep = nullptr;
    struct { bool ready = false, use = false; const int64_t *off = nullptr; int64_t S = 0, bs = 0, spb = 0; int T = 0, K = 0; uint32_t seed = 0; } prep;
    // per-bag-size index tables (m6a_pool_rtab.hip), valid for (seed, T*K, stream length)
    struct {
        bool valid = false; uint32_t seed = 0; int64_t A = 0, n_blk = 0;
        int cap = 0, used = 0;
        uint16_t *C = nullptr; uint32_t *RS = nullptr;
        int32_t slot_of_n[M6A_RTAB_MAX_N + 1];
    } rt;
    // host-pointer staging
    DevBuf sX, sK, sOff, sP, sSite, sMod, val_idx, val_y, val_avg, sOffChunk;
    Staging stg;
    int rt_presize = 0;                       // warm_default: allocate the index-table arena for this many bag sizes
    uint32_t rt_credit_seed = 0; int64_t rt_credit_A = 0, rt_credit = 0;   // sites pooled on the scan kernels while tables were missing
    // streaming job (m6a_job_begin / m6a_job_feed / m6a_job_end): the reference's batch loop fed as it is produced
    struct Job {
        bool open = false;
        int failed = 0;                       // first error of a feed: the job is void, m6a_job_end reports it
        std::string failed_msg;               // ... with the text it had (other calls may have overwritten the context's since)
        int T = 0, K = 0; float thr = 0.f; uint32_t seed = 0; int64_t bs = 1, spb = 1;
        std::vector<int64_t> off;             // the job's CSR offsets so far, host [S+1]
        int64_t S = 0, R = 0;                 // sites / reads fed so far (R == off[S])
        // ring of sub-slots carved out of the pinned staging ring, each mirrored by a device sub-slot:
        // [off_local i64 (cap_sites+1) | off_global i64 (cap_sites+1) | site_kmers u8 cap_sites*3 | X f32 cap_reads*9]
        int n_sub = 0; size_t sub_bytes = 0, o_goff = 0, o_km = 0, o_x = 0;
        int64_t cap_sites = 0, cap_reads = 0;
        std::vector<char *> pin;
        std::vector<hipEvent_t> ev_h2d, ev_enc;
        std::vector<char> used;               // sub-slot has carried a chunk of this job (its events are live)
        int64_t item = 0;                     // chunks flushed so far: chunk k uses sub-slot k % n_sub
        int64_t fill_sites = 0, fill_reads = 0, fill_min = INT64_MAX;   // the chunk being filled
        bool cur_ready = false;               // the current sub-slot's previous DMA has been waited for
        int64_t chunks = 0;
    } job;
    DevBuf gSite, gMod, gP;                   // host-pointer m6a_gather / m6a_gather_reads: what rank dst receives
    DevBuf jX, jP, jOff;                      // device sub-slots; read probabilities [R] and CSR offsets [S+1] of the job
    DevBuf mt_scratch; int64_t mt_polys_G = 0;   // segmented stream generator: head words, segment histories, jump polynomials
    std::vector<int> iter_of;                 // ensure_table: iteration held by (row, lane) of the LDS table kernel
    std::thread warm;                         // m6a_create's background set-up for the default job parameters (settle() joins it)
    void *comm = nullptr;                     // ncclComm_t
    int comm_rank = 0, comm_world = 0;
    Profiler prof;
    const char *pool_variant = "none";
};

namespace {

int fail(m6a_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(c, expr)                                                                    \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            (void)hipGetLastError();                                                       \
            return fail((c), e_ == hipErrorOutOfMemory ? M6A_ENOMEM : M6A_EHIP, "%s: %s (%s:%d)", \
                        #expr, hipGetErrorString(e_), __FILE__, __LINE__);                 \
        }                                                                                  \
    } while (0)

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
    // W1aug[160][16]: columns 0..14 = alpha*W1, column 15 = alpha*b1 + (beta - mean*alpha)
    // (torch eval BatchNorm1d: y*alpha + beta - mean*alpha, alpha = gamma/sqrt(var+eps), blocks.py:250);
    // row 150 = constant-one unit (carries b2), rows 151..159 = 0.
    std::vector<float> w1(160 * 16, 0.f), w2(32 * 160, 0.f);
    for (int j = 0; j < 150; j++) {
        const float invstd = 1.0f / std::sqrt(w[O_VAR + j] + 1e-5f);
        const float alpha = w[O_G + j] * invstd;
        const float shift = w[O_BE + j] - w[O_MU + j] * alpha;
        for (int k = 0; k < 15; k++) w1[j * 16 + k] = alpha * w[O_W1 + 15 * j + k];
        w1[j * 16 + 15] = alpha * w[O_B1 + j] + shift;
    }
    w1[150 * 16 + 15] = 1.0f;
    for (int o = 0; o < 32; o++) {
        for (int k = 0; k < 150; k++) w2[o * 160 + k] = w[O_W2 + 150 * o + k];
        w2[o * 160 + 150] = w[O_B2 + o];
    }
    // the kernels' ReLU is x + |x| = 2 * relu(x) (one full-rate v_add_f32; max is half rate, m6a_kernels.hip): the 0.5 rides in
    // the weights that consume the activations -- exact for every normal float (a weight below 2^-125 would lose its last bit:
    // a contribution under 1e-37 per unit of activation)
    for (float &v : w2) v *= 0.5f;
    const float w3_scale = 0.5f;
    frag.assign(M6A_WFRAG_FLOATS, 0.f);
    for (int lane = 0; lane < 64; lane++) {
        const int col = lane & 31, half = lane >> 5;
        for (int m = 0; m < 5; m++) {
            for (int st = 0; st < 8; st++)      // A[i=col][k=2st+half] <-> feature st + 8*half
                frag[(m * 8 + st) * 64 + lane] = w1[(32 * m + col) * 16 + st + 8 * half];
            for (int q = 0; q < 16; q++) {      // K index = hidden unit held by acc register q
                const int unit = 32 * m + (q & 3) + 8 * (q >> 2) + 4 * half;
                frag[(40 + m * 16 + q) * 64 + lane] = w2[col * 160 + unit];
            }
        }
        for (int q = 0; q < 16; q++)
            frag[(120 + q) * 64 + lane] = w3_scale * w[O_W3 + (q & 3) + 8 * (q >> 2) + 4 * half];
    }
    // 12-slot kernel: x-slot fragments W1'[u][2st+half] (st<4), W1'[u][8]; and the per-unit rows the
    // per-site c vectors are folded from: W1'[u][9..14], b1'[u]  (w1 column 15 is the folded bias)
    frag2.assign(M6A_WFRAG2_FLOATS, 0.f);
    w1e.assign(M6A_W1E_FLOATS, 0.f);
    for (int lane = 0; lane < 64; lane++) {
        const int col = lane & 31, half = lane >> 5;
        for (int m = 0; m < 5; m++) {
            for (int st = 0; st < 4; st++) frag2[(m * 4 + st) * 64 + lane] = w1[(32 * m + col) * 16 + 2 * st + half];
            frag2[(20 + m) * 64 + lane] = w1[(32 * m + col) * 16 + 8];
        }
    }
    for (int m = 0; m < 5; m++)
        for (int col = 0; col < 32; col++) {
            for (int e = 0; e < 6; e++) w1e[(m * 7 + e) * 32 + col] = w1[(32 * m + col) * 16 + 9 + e];
            w1e[(m * 7 + 6) * 32 + col] = w1[(32 * m + col) * 16 + 15];
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
            (void)hipFree(rt.C); (void)hipFree(rt.RS);
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
        b.C = rt.C; b.RS = rt.RS; b.c_stride = c_stride;
        const unsigned gx = (unsigned)std::min<int64_t>((n_blk + 63) / 64, 65535);   // 4 waves x 16 blocks per workgroup
        hipLaunchKernelGGL(rtab_count_kernel, dim3((unsigned)todo.size(), gx), dim3(256), 0, c->stream, b);
        hipLaunchKernelGGL(rtab_scan_kernel, dim3((unsigned)todo.size()), dim3(256), 0, c->stream, b);
        hipLaunchKernelGGL(rtab_fill_kernel, dim3((unsigned)todo.size(), gx), dim3(256), 0, c->stream, b);
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

// accepted indices for pool_reg_kernel: idx2[j][T + 8][K] bytes, 2 x index (a register pair per bag entry),
// iterations in order, one round of zero padding for the prefetch past the end.
int ensure_table_reg(m6a_ctx *c, uint32_t seed, int n, int T, int K, int jmax)
{
    auto &k = c->tab_reg_key;
    if (k.valid && k.seed == seed && k.n == n && k.T == T && k.K == K && k.jmax >= jmax) return M6A_OK;
    k.valid = false;                                       // the table is rewritten below: a failure must not leave the old key standing
    const size_t per_j = (size_t)(T + 8) * K;              // K = 20: a multiple of 4 bytes
    const size_t bytes = (size_t)jmax * per_j + 4096;      // the kernel's look-ahead touches up to 2 KB past the last row
    HIPCHK(c, c->tab_reg.ensure(bytes));
    HIPCHK(c, hipMemsetAsync(c->tab_reg.p, 0, bytes, c->stream));
    const uint16_t *C = nullptr;
    int rc = uniform_slice(c, seed, n, T, K, jmax, &C);
    if (rc) return rc;
    if (C) {
        const int64_t A = (int64_t)T * K;
        hipLaunchKernelGGL(rtab_to_reg_table_kernel, dim3((unsigned)((A * jmax + 255) / 256)), dim3(256), 0, c->stream,
                           C, A, (int64_t)per_j, jmax, (uint8_t *)c->tab_reg.p);
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
    a.wfrag2 = c->d_wfrag2; a.w1e_tab = c->d_w1e; a.err = c->d_err;
    a.n_sites = S; a.n_reads = R; a.n_tiles = (R + 31) / 32; a.b3 = c->b3;
    const int64_t max_waves = (int64_t)c->n_cu * 8;        // 2 blocks/CU x 4 waves
    a.tiles_per_wave = (a.n_tiles + max_waves - 1) / max_waves;
    const int64_t waves = (a.n_tiles + a.tiles_per_wave - 1) / a.tiles_per_wave;
    const unsigned blocks = (unsigned)((waves + 3) / 4);
    // every bag >= 16 reads: a 32-read tile spans <= 3 sites -> 12-slot layer 1 (106 MFMAs per tile);
    // otherwise the general 16-slot kernel (116)
    // the 12-slot kernel keeps tile and site indices in 32 bits (wave-uniform SALU arithmetic)
    const bool fits32 = S < 0x7ffffff0LL && a.n_tiles < 0x7ffffff0LL;
    if (c->enc_variant == 2 && !fits32) return fail(c, M6A_EUNSUPPORTED, "12-slot encoder: more than 2^31 sites or tiles");
    const bool csite = c->enc_variant ? c->enc_variant == 2 : (c->bag_min >= M6A_CSITE_MIN_BAG && fits32);
    c->enc_variant_used = csite ? "csite12" : "general16";
    prof_begin(c, 0);
    if (csite) hipLaunchKernelGGL(enc_csite_kernel, dim3(blocks), dim3(256), 0, c->stream, a);
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
                uint32_t seed, int64_t bs, int64_t spb, float *site, double *mod, bool dry = false)
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
    const int64_t gmax = c->goff_key.gmax;

    rc = ensure_mean_plan(c, T);
    if (rc) return rc;
    const bool uniform = nmin == nmax && nmin >= 1 && nmin <= M6A_TABLE_MAX_N && K == 20 && gmax <= 4096;
    // register kernel: stack in 8 register quads, 32-bit byte offsets into read_prob, table <= 256 MB
    const bool reg_ok = uniform && c->plan.depth <= M6A_REG_STACK && c->n_reads < (int64_t)1 << 30 &&
                        (int64_t)gmax * (T + 8) * K <= (int64_t)256 << 20;
    if (uniform && reg_ok && c->table_variant != 1) {
        rc = ensure_table_reg(c, seed, (int)nmin, T, K, (int)gmax);
        if (rc) return rc;
        plan_args(c, a);
        a.tab = (const uint32_t *)c->tab_reg.p; a.uniform_n = (int)nmin; a.jmax = (int)gmax;
        a.reg_gpad = (a.n_groups + 63) / 64 * 64;
        HIPCHK(c, c->reg_out.ensure((size_t)a.jmax * a.reg_gpad * 5));
        if (dry) return M6A_OK;
        c->pool_variant = "table-reg";
        prof_begin(c, 1);
        // one wavefront = position j of 256 flush groups (4 sites per lane); blockIdx % jmax = j keeps a
        // position's index rows in one XCD's L2
        const int64_t wpj = (a.n_groups + 255) / 256;
        a.reg_site = (float *)c->reg_out.p;
        a.reg_cnt = (uint8_t *)c->reg_out.p + (size_t)a.jmax * a.reg_gpad * 4;
        hipLaunchKernelGGL(pool_reg_kernel, dim3((unsigned)(wpj * a.jmax)), dim3(64), 0, c->stream, a);
        hipLaunchKernelGGL(pool_reg_finish_kernel, dim3((unsigned)((a.n_groups + 31) / 32), (unsigned)((a.jmax + 31) / 32)), dim3(256), 0, c->stream, a);
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
            HIPCHK(c, hipEventSynchronize(g.ev_h2d[slot]));            // the slot's previous DMA has left it
            if (k >= kStageSlots) { rc = drain_out(k - kStageSlots); if (rc) return rc; }
        }
        g.pool->copy(g.pin_in[slot], X + r0 * 9, (size_t)nr * 9 * 4);
        HIPCHK(c, hipMemcpyAsync((float *)c->sX.p + r0 * 9, g.pin_in[slot], (size_t)nr * 9 * 4, hipMemcpyHostToDevice, g.s_h2d));
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
            HIPCHK(c, hipMemcpyAsync(g.pin_out[slot], (const float *)c->sP.p + r0, (size_t)nr * 4, hipMemcpyDeviceToHost, g.s_d2h));
            HIPCHK(c, hipEventRecord(g.ev_d2h[slot], g.s_d2h));
            out_pending[(size_t)k] = 1;
        }
    }
    for (int64_t k = 0; k < nchunk; k++) { rc = drain_out(k); if (rc) return rc; }
    return M6A_OK;
}

// site_prob / mod_ratio of a host-pointer call: through the (now idle) pinned slots when they fit, so the caller's
// pageable arrays are filled by the copy threads instead of a staged synchronous hipMemcpy.  Synchronises the stream.
int staged_outputs(m6a_ctx *c, int64_t S, float *site, double *mod)
{
    Staging &g = c->stg;
    const size_t slot_bytes = g.ready ? (size_t)g.chunk_reads * 9 * 4 : 0;
    if ((size_t)S * 8 > slot_bytes) {
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

// ---- RCCL, bound at run time ---------------------------------------------------------------------------------
// (types restated from rccl.h so the library builds and loads without RCCL: NCCL_UNIQUE_ID_BYTES = 128,
// ncclFloat32 = 7, ncclFloat64 = 8, ncclSuccess = 0)
struct RcclId { char internal[M6A_COMM_ID_BYTES]; };
struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(RcclId *) = nullptr;
    int (*CommInitRank)(void **, int, RcclId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*CommCount)(void *, int *) = nullptr;          // the four below are optional: a copy without them still gathers
    int (*CommUserRank)(void *, int *) = nullptr;
    int (*CommCuDevice)(void *, int *) = nullptr;
    int (*GetVersion)(int *) = nullptr;
    std::string err;
};

Rccl *rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // M6A_RCCL_LIB names THE copy to use (nothing else is tried); otherwise the usual names
        std::vector<std::string> names;
        const char *e = getenv("M6A_RCCL_LIB");
        if (e && *e) names.push_back(e);
        else {
            for (const char *n : {"librccl.so.1", "librccl.so"}) names.push_back(n);
            names.push_back("/opt/rocm/lib/librccl.so.1");
        }
        for (size_t i = 0; i < names.size() && !r.h; i++) {
            // a copy that is already mapped (e.g. PyTorch's) wins: it is bound to the process's HIP runtime
            r.h = dlopen(names[i].c_str(), RTLD_NOW | RTLD_NOLOAD);
        }
        for (size_t i = 0; i < names.size() && !r.h; i++) r.h = dlopen(names[i].c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!r.h) { r.err = e && *e ? std::string("cannot load M6A_RCCL_LIB=") + e : std::string("librccl not found (set M6A_RCCL_LIB)"); return; }
        auto sym = [&](const char *n) { void *p = dlsym(r.h, n); if (!p && r.err.empty()) r.err = std::string("librccl lacks ") + n; return p; };
        r.GetUniqueId = (int (*)(RcclId *))sym("ncclGetUniqueId");
        r.CommInitRank = (int (*)(void **, int, RcclId, int))sym("ncclCommInitRank");
        r.CommDestroy = (int (*)(void *))sym("ncclCommDestroy");
        r.GroupStart = (int (*)())sym("ncclGroupStart");
        r.GroupEnd = (int (*)())sym("ncclGroupEnd");
        r.Send = (int (*)(const void *, size_t, int, int, void *, hipStream_t))sym("ncclSend");
        r.Recv = (int (*)(void *, size_t, int, int, void *, hipStream_t))sym("ncclRecv");
        r.GetErrorString = (const char *(*)(int))sym("ncclGetErrorString");
        r.CommCount = (int (*)(void *, int *))dlsym(r.h, "ncclCommCount");
        r.CommUserRank = (int (*)(void *, int *))dlsym(r.h, "ncclCommUserRank");
        r.CommCuDevice = (int (*)(void *, int *))dlsym(r.h, "ncclCommCuDevice");
        r.GetVersion = (int (*)(int *))dlsym(r.h, "ncclGetVersion");
    });
    return &r;
}

#define RCCLCHK(c, R, expr)                                                                             \
    do {                                                                                                \
        const int e_ = (expr);                                                                          \
        if (e_ != 0) return fail((c), M6A_EHIP, "%s: %s", #expr, (R)->GetErrorString ? (R)->GetErrorString(e_) : "RCCL error"); \
    } while (0)

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

int job_busy(m6a_ctx *c)
{
    if (c->job.open) return fail(c, M6A_EINVAL, "a streaming job is open on this context (m6a_job_end or m6a_job_abort first)");
    return M6A_OK;
}

// m6a_set_host_offsets is one-shot: whichever entry point runs next consumes the hint -- on its device-pointer branch
// through bag_stats, on every other path (host pointers, argument errors) by leaving this scope.  A pointer that stayed
// armed would describe some later call's off[] wrongly (or point at memory the caller has freed by then).
struct HintScope {
    m6a_ctx *c;
    explicit HintScope(m6a_ctx *ctx) : c(ctx) {}
    ~HintScope() { if (c) c->hint_off = nullptr; }
};

int check_pool_args(m6a_ctx *c, int64_t S, int T, int K, int rng_mode, int64_t bs, int64_t spb)
{
    if (!c) return M6A_EINVAL;
    if (S < 0) return fail(c, M6A_EINVAL, "n_sites < 0");
    if (T < 1) return fail(c, M6A_EINVAL, "n_iters must be >= 1");
    if (K < 1 || K > M6A_MAX_SAMPLES) return fail(c, M6A_EINVAL, "n_samples must be in 1..%d", M6A_MAX_SAMPLES);
    if ((int64_t)T * K > 0x3fffffff) return fail(c, M6A_EINVAL, "n_iters*n_samples too large");
    if (rng_mode != M6A_RNG_NUMPY) return fail(c, M6A_EUNSUPPORTED, "rng_mode %d not supported", rng_mode);
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
    if (*c->h_err) {
        HIPCHK(c, hipMemsetAsync(c->d_err, 0, sizeof(int), c->stream));
        return deferred_error(c);
    }
    return M6A_OK;
}

// Every entry point that touches the stream or the sampling state waits for m6a_create's background set-up first.
inline void settle(m6a_ctx *c) { if (c && c->warm.joinable()) c->warm.join(); }

// Background half of m6a_create: everything a first call with the reference's DEFAULT job parameters would otherwise
// build inside the call -- seed 0 (scripts/inference.py:60), num_iterations 1000 (:56), 20 samples
// (inference_utils.py:54), batch_size 16 x save_per_batch 2 (:46-50) -> flush groups of <= 32 sites: the pairwise-sum
// plan, the MT19937 stream, the index tables of bag sizes 2 .. M6A_WARM_SLOTS - 1 (default 512: 1.4 GB of a 288 GB part,
// one 1.2 ms pass of an idle GPU) and the register kernel's table of the smallest legal bag (min_reads = 20,
// constants.py:14).  Runs on the side stream from its
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
        const char *e = getenv("M6A_WARM_SLOTS");
        c->rt_presize = e && atoi(e) > 0 ? std::min(atoi(e), M6A_RTAB_MAX_N + 1) : 512;
        // speculative memory stays a small share of what is free (a shared or nearly full GPU gets the 32-slot arena a first
        // call would make anyway); an explicit M6A_WARM_SLOTS is taken at its word
        size_t free_b = 0, total_b = 0;
        if (!e && hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const size_t slot_bytes = (size_t)(c->raw_len) * 2 + (size_t)(c->raw_len / 64 + 1) * 4;
            if ((size_t)c->rt_presize * slot_bytes > free_b / 20) c->rt_presize = 32;
        }
        rc = ensure_table_reg(c, seed, n, T, K, (int)gmax);
        if (!rc) {
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

}  // namespace

// =================================================================================================
extern "C" {

const char *m6a_version(void) { return "m6a_hip 0.1 (gfx950)"; }

const char *m6a_last_error(const m6a_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int m6a_create(m6a_ctx **out, const float *weights, size_t n_floats, int device_id)
{
    if (!out) return M6A_EINVAL;
    *out = nullptr;
    if (!weights || n_floats != M6A_N_WEIGHTS)
        return fail(nullptr, M6A_EINVAL, "weights must be %d floats (got %zu)", M6A_N_WEIGHTS, n_floats);
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
    const char *w = getenv("M6A_WARMUP");
    if (!(w && w[0] == '0')) {
        try { c->warm = std::thread(warm_default, c); } catch (...) { /* no thread: the first call sets up what it needs */ }
    }
    *out = c;
    return M6A_OK;
}

void m6a_destroy(m6a_ctx *c)
{
    if (!c) return;
    settle(c);
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int k = 0; k < 2; k++) {
        for (auto e : c->prof.start[k]) (void)hipEventDestroy(e);
        for (auto e : c->prof.stop[k]) (void)hipEventDestroy(e);
    }
    for (DevBuf *b : {&c->raw, &c->tab, &c->goff, &c->rp_scratch, &c->off_scratch, &c->start_pos, &c->plan_dev, &c->tab_reg, &c->val_idx, &c->val_y, &c->val_avg, &c->sX, &c->sK, &c->sOff,
                      &c->sP, &c->sSite, &c->sMod}) b->release();
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
    if (c->rt.C) (void)hipFree(c->rt.C);
    if (c->rt.RS) (void)hipFree(c->rt.RS);
    for (DevBuf *b : {&c->ctl_dev, &c->rt_rank, &c->rt_order, &c->sOffChunk, &c->reg_out, &c->jX, &c->jP, &c->jOff, &c->gSite, &c->gMod, &c->gP, &c->mt_scratch}) b->release();
    for (auto e : c->job.ev_h2d) (void)hipEventDestroy(e);
    for (auto e : c->job.ev_enc) (void)hipEventDestroy(e);
    release_staging(c);
    if (c->comm) { Rccl *R = rccl(); if (R->CommDestroy) (void)R->CommDestroy(c->comm); c->comm = nullptr; }
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
    if (first_site < 0) return fail(c, M6A_EINVAL, "job offset must be >= 0");
    c->job_offset = first_site;
    return M6A_OK;
}

int m6a_set_encoder_variant(m6a_ctx *c, int mode)
{
    if (!c) return M6A_EINVAL;
    if (mode < 0 || mode > 2) return fail(c, M6A_EINVAL, "encoder variant must be 0 (auto), 1 (16-slot) or 2 (12-slot)");
    c->enc_variant = mode;
    return M6A_OK;
}

const char *m6a_last_encoder_variant(const m6a_ctx *c) { return c ? c->enc_variant_used : "none"; }

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

namespace {

// MT19937 with the 624-word state refilled in bulk: the three recurrence loops have dependence distances of
// 227 and more, so the compiler vectorises them; std::mt19937's per-call path was a third of the sampler's time.
struct MtBulk {
    uint32_t s[624], out[624];
    int pos = 624;
    explicit MtBulk(uint32_t seed)
    {
        uint32_t x = seed;
        s[0] = x;
        for (uint32_t i = 1; i < 624; i++) { x = 1812433253u * (x ^ (x >> 30)) + i; s[i] = x; }
    }
    static inline uint32_t tw(uint32_t a, uint32_t b)
    {
        const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
        return (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
    }
    __attribute__((always_inline)) inline void refill_body()
    {
        for (int k = 0; k < 227; k++) s[k] = s[k + 397] ^ tw(s[k], s[k + 1]);
        for (int k = 227; k < 623; k++) s[k] = s[k - 227] ^ tw(s[k], s[k + 1]);
        s[623] = s[396] ^ tw(s[623], s[0]);
        for (int k = 0; k < 624; k++) {
            uint32_t y = s[k];
            y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
            out[k] = y;
        }
        pos = 0;
    }
    __attribute__((target("avx2"))) void refill_avx2() { refill_body(); }     // 8 lanes per step: 0.57 -> 0.35 ns per word
    void refill_base() { refill_body(); }
    void refill()
    {
        static const bool avx2 = __builtin_cpu_supports("avx2");
        if (avx2) refill_avx2(); else refill_base();
    }
    inline uint32_t next() { if (pos == 624) refill(); return out[pos++]; }
};

// The training-mode sampler of a whole validation run (data_utils.py:213-214 under training_utils.py:235-240,
// num_workers=0): RandomState.choice(n, K, replace=False) = permutation(n)[:K] = the legacy shuffle of arange(n): for
// i = n-1..1: j = rk_interval(i) (masked rejection over 32-bit words), swap.  ONE stream, seeded once, pass after pass,
// site after site: where an item (pass, site) starts depends on how many words every earlier shuffle rejected, so the walk
// over the stream is a chain.  It is split in two:
//   * the WALK (this thread): per item only COUNT -- i steps down on every accepted word, the mask changes when i crosses a
//     power of two -- no permutation, no memory traffic, ~1 ns per word; it hands out blocks of items together with the
//     stream words they consume and every item's offset into them;
//   * the SHUFFLES (worker threads): every item replayed from its offset, independently of all others.
// Both loops are branch-free per word (the accept/reject branch of the textbook loop mispredicts every third word): inside
// a phase -- i in [2^b, 2^(b+1)) -- the mask is fixed, acceptance is one compare, the swap is two unconditional stores of
// selected values.  gidx gets GLOBAL read indices [T][S][K].
struct ValBlock {
    int64_t k0 = 0, k1 = 0;                  // items [k0, k1) of the run, item k = (pass k / S, site k % S)
    std::vector<uint32_t> words;             // the stream words these items consume, in order
    std::vector<uint32_t> start;             // offset of every item's first word in `words`
};

// words the shuffle of n entries consumes from w[] (w holds at least `avail` words; returns ~0 if they run out)
inline size_t shuffle_count(const uint32_t *w, size_t avail, uint32_t n, uint32_t *i_io)
{
    uint32_t i = *i_io;
    size_t q = 0;
    while (i) {
        const uint32_t mask = 0xffffffffu >> __builtin_clz(i), lo = (mask >> 1) + 1;      // this phase: i in [lo, mask]
        for (;;) {
            if (q == avail) { *i_io = i; return q; }
            i -= ((w[q++] & mask) <= i);
            if (i < lo) break;
        }
    }
    (void)n;
    *i_io = 0;
    return q;
}

// the shuffle itself: p holds >= 2 * n + 2 entries (a rejected word indexes up to the mask, its slot is rewritten unchanged)
inline void shuffle_item(const uint32_t *w, uint32_t n, int32_t *p, int K, int32_t base, int32_t *out)
{
    for (uint32_t i = 0; i < n; i++) p[i] = (int32_t)i;
    uint32_t i = n - 1;
    int32_t a = p[i];                         // perm[i] rides in a register while i stands still
    while (i) {
        const uint32_t mask = 0xffffffffu >> __builtin_clz(i), lo = (mask >> 1) + 1;
        for (;;) {
            const uint32_t v = *w++ & mask;
            const bool acc = v <= i;
            const int32_t b = p[v];
            p[i] = acc ? b : a;
            p[v] = acc ? a : b;
            i -= acc;
            a = p[i];
            if (i < lo) break;
        }
    }
    for (int k = 0; k < K; k++) out[k] = base + p[k];
}

// The stream, produced one batch ahead of the walk on a thread of its own (the generator is a third of the walk's time
// otherwise): a ring of batches of 32 refills, handed over through two counters.
class MtProducer {
public:
    static constexpr int64_t kBatch = 624 * 32;
    explicit MtProducer(uint32_t seed) : gen_(seed), buf_((size_t)kBatch * kRing) { th_ = std::thread([this] { run(); }); }
    ~MtProducer() { stop_.store(true); th_.join(); }
    const uint32_t *get(int64_t b)                           // batch b (blocks until it exists); valid until release(b)
    {
        while (produced_.load(std::memory_order_acquire) <= b) __builtin_ia32_pause();
        return &buf_[(size_t)(b % kRing) * kBatch];
    }
    void release(int64_t b) { consumed_.store(b + 1, std::memory_order_release); }
private:
    static constexpr int kRing = 8;
    void run()
    {
        for (int64_t b = 0;; b++) {
            while (b - consumed_.load(std::memory_order_acquire) >= kRing) {
                if (stop_.load()) return;
                __builtin_ia32_pause();
            }
            if (stop_.load()) return;
            uint32_t *dst = &buf_[(size_t)(b % kRing) * kBatch];
            for (int r = 0; r < 32; r++) { gen_.refill(); std::memcpy(dst + r * 624, gen_.out, sizeof gen_.out); }
            produced_.store(b + 1, std::memory_order_release);
        }
    }
    MtBulk gen_;
    std::vector<uint32_t> buf_;
    std::atomic<int64_t> produced_{0}, consumed_{0};
    std::atomic<bool> stop_{false};
    std::thread th_;
};

int validation_indices(m6a_ctx *c, const int64_t *h_off, int64_t S, int T, int K, uint32_t seed, std::vector<int32_t> &gidx)
{
    int64_t nmax = 0;
    for (int64_t s = 0; s < S; s++) {
        const int64_t n = h_off[s + 1] - h_off[s];
        if (n < K) return fail(c, M6A_EINVAL, "site %lld has %lld reads, fewer than n_samples = %d (sampling without replacement)",
                               (long long)s, (long long)n, K);
        nmax = std::max(nmax, n);
    }
    if (h_off[S] > 0x7fffffff) return fail(c, M6A_EUNSUPPORTED, "more than 2^31 reads");
    gidx.resize((size_t)T * S * K);
    const int64_t n_items = (int64_t)T * S;
    const int64_t block_items = 2048;
    const char *env = getenv("M6A_VALIDATE_THREADS");
    int n_workers = env ? atoi(env) : std::min(32, std::max(1, m6a_usable_cpus() - 2));
    if (n_items < 4 * block_items || n_workers < 1) n_workers = 0;          // small runs: walk and shuffle on this thread

    std::mutex mu;
    std::condition_variable cv_put, cv_get;
    std::vector<std::unique_ptr<ValBlock>> queue;
    bool done = false;
    auto shuffle_block = [&](const ValBlock &b, std::vector<int32_t> &perm) {
        for (int64_t k = b.k0; k < b.k1; k++) {
            const int64_t s = k % S;
            shuffle_item(b.words.data() + b.start[(size_t)(k - b.k0)], (uint32_t)(h_off[s + 1] - h_off[s]), perm.data(), K, (int32_t)h_off[s],
                         gidx.data() + (size_t)k * K);
        }
    };
    std::vector<std::thread> workers;
    for (int t = 0; t < n_workers; t++)
        workers.emplace_back([&] {
            std::vector<int32_t> perm((size_t)2 * nmax + 2);
            for (;;) {
                std::unique_ptr<ValBlock> b;
                {
                    std::unique_lock<std::mutex> g(mu);
                    cv_get.wait(g, [&] { return done || !queue.empty(); });
                    if (queue.empty()) return;
                    b = std::move(queue.back());
                    queue.pop_back();
                }
                cv_put.notify_one();
                shuffle_block(*b, perm);
            }
        });

    {
        MtProducer src(seed);
        int64_t batch = 0;                                    // the batch the walk is in, and how far
        const uint32_t *bw = src.get(0);
        size_t bpos = 0;
        std::vector<int32_t> perm0;
        if (!n_workers) perm0.resize((size_t)2 * nmax + 2);
        for (int64_t k0 = 0; k0 < n_items; k0 += block_items) {
            std::unique_ptr<ValBlock> b(new ValBlock);
            b->k0 = k0; b->k1 = std::min(n_items, k0 + block_items);
            b->start.resize((size_t)(b->k1 - b->k0));
            b->words.reserve((size_t)(b->k1 - b->k0) * (size_t)nmax * 3 / 2 + (size_t)MtProducer::kBatch);
            b->words.assign(bw + bpos, bw + MtProducer::kBatch);               // what is left of the current batch
            size_t cur = 0;
            for (int64_t k = b->k0; k < b->k1; k++) {
                const int64_t s = k % S;
                const uint32_t n = (uint32_t)(h_off[s + 1] - h_off[s]);
                b->start[(size_t)(k - b->k0)] = (uint32_t)cur;
                uint32_t i = n - 1;
                while (i) {
                    cur += shuffle_count(b->words.data() + cur, b->words.size() - cur, n, &i);
                    if (i) {                                                     // the words ran out: the next batch joins them
                        src.release(batch++);
                        bw = src.get(batch);
                        b->words.insert(b->words.end(), bw, bw + MtProducer::kBatch);
                    }
                }
            }
            bpos = (size_t)MtProducer::kBatch - (b->words.size() - cur);        // the next block starts inside this batch
            if (!n_workers) { shuffle_block(*b, perm0); continue; }
            {
                std::unique_lock<std::mutex> g(mu);
                cv_put.wait(g, [&] { return queue.size() < 4 * (size_t)n_workers; });
                queue.push_back(std::move(b));
            }
            cv_get.notify_one();
        }
    }
    { std::lock_guard<std::mutex> g(mu); done = true; }
    cv_get.notify_all();
    for (auto &w : workers) w.join();
    return M6A_OK;
}

// d_rp: device read probabilities; d_y [T][S] and d_avg [S] (or null): device
int launch_validate_pool(m6a_ctx *c, const float *d_rp, const int64_t *h_off, int64_t S, int T, int K, uint32_t seed,
                         float *d_y, float *d_avg)
{
    std::vector<int32_t> gidx;
    int rc;
    try {
        rc = validation_indices(c, h_off, S, T, K, seed, gidx);
    } catch (const std::bad_alloc &) {
        rc = fail(c, M6A_ENOMEM, "out of host memory");
    }
    if (rc) return rc;
    HIPCHK(c, c->val_idx.ensure(gidx.size() * 4));
    HIPCHK(c, hipMemcpyAsync(c->val_idx.p, gidx.data(), gidx.size() * 4, hipMemcpyHostToDevice, c->stream));
    const int64_t nb = (int64_t)T * S;
    prof_begin(c, 1);
    hipLaunchKernelGGL(sampled_noisy_or_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, c->stream,
                       d_rp, (const int32_t *)c->val_idx.p, nb, K, d_y);
    if (d_avg)
        hipLaunchKernelGGL(mean_over_passes_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, c->stream,
                           (const float *)d_y, T, S, d_avg);
    prof_end(c, 1);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));      // gidx (pageable) must outlive the copy
    return M6A_OK;
}

int check_validate_args(m6a_ctx *c, int64_t S, int T, int K)
{
    if (!c) return M6A_EINVAL;
    if (S < 0) return fail(c, M6A_EINVAL, "n_sites < 0");
    if (T < 1) return fail(c, M6A_EINVAL, "n_iters must be >= 1");
    if (K < 1 || K > M6A_MAX_SAMPLES) return fail(c, M6A_EINVAL, "n_samples must be in 1..%d", M6A_MAX_SAMPLES);
    if ((double)T * (double)S * K > 2.0e9) return fail(c, M6A_EUNSUPPORTED, "n_iters * n_sites * n_samples too large for one call");
    return M6A_OK;
}

}  // namespace

int m6a_validate_pool(m6a_ctx *c, const float *rp, const int64_t *off, int64_t S, int T, int K, uint32_t seed,
                      float *y, float *avg)
{
    settle(c);
    HintScope hint_scope(c);
    if (c && c->job.open) return job_busy(c);
    int rc = check_validate_args(c, S, T, K);
    if (rc) return rc;
    if (S == 0) return M6A_OK;
    if (!rp || !off || !y) return fail(c, M6A_EINVAL, "null pointer argument");
    HIPCHK(c, hipSetDevice(c->device));
    const bool dev = is_device_ptr(rp);
    if (dev != is_device_ptr(off) || dev != is_device_ptr(y) || (avg && dev != is_device_ptr(avg)))
        return fail(c, M6A_EINVAL, "read_prob, off, y_pred, y_pred_avg must be all host or all device pointers");
    std::vector<int64_t> h_off;
    const int64_t *ho = off;
    if (dev) {
        h_off.resize((size_t)S + 1);
        HIPCHK(c, hipMemcpyAsync(h_off.data(), off, (size_t)(S + 1) * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        ho = h_off.data();
    }
    if (ho[0] != 0) return fail(c, M6A_EINVAL, "off[0] must be 0");
    for (int64_t s = 0; s < S; s++) if (ho[s + 1] < ho[s]) return fail(c, M6A_EINVAL, "off[] must be non-decreasing");
    if (dev) return launch_validate_pool(c, rp, ho, S, T, K, seed, y, avg);
    const int64_t R = ho[S];
    HIPCHK(c, c->sP.ensure((size_t)std::max<int64_t>(R, 1) * 4));
    HIPCHK(c, c->val_y.ensure((size_t)T * S * 4));
    HIPCHK(c, c->val_avg.ensure((size_t)S * 4));
    HIPCHK(c, hipMemcpyAsync(c->sP.p, rp, (size_t)R * 4, hipMemcpyHostToDevice, c->stream));
    rc = launch_validate_pool(c, (const float *)c->sP.p, ho, S, T, K, seed, (float *)c->val_y.p, avg ? (float *)c->val_avg.p : nullptr);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(y, c->val_y.p, (size_t)T * S * 4, hipMemcpyDeviceToHost, c->stream));
    if (avg) HIPCHK(c, hipMemcpyAsync(avg, c->val_avg.p, (size_t)S * 4, hipMemcpyDeviceToHost, c->stream));
    return sync_and_check(c);
}

int m6a_validate(m6a_ctx *c, const float *X, const uint8_t *km, const int64_t *off, int64_t S, int T, int K,
                 uint32_t seed, float *rp, float *y, float *avg)
{
    settle(c);
    HintScope hint_scope(c);
    if (c && c->job.open) return job_busy(c);
    int rc = check_validate_args(c, S, T, K);
    if (rc) return rc;
    if (S == 0) return M6A_OK;
    if (!X || !km || !off || !y) return fail(c, M6A_EINVAL, "null pointer argument");
    HIPCHK(c, hipSetDevice(c->device));
    const bool dev = is_device_ptr(X);
    if (dev != is_device_ptr(km) || dev != is_device_ptr(off) || dev != is_device_ptr(y) || (rp && dev != is_device_ptr(rp)) ||
        (avg && dev != is_device_ptr(avg)))
        return fail(c, M6A_EINVAL, "X, site_kmers, off and the outputs must be all host or all device pointers");
    if (dev) {
        float *d_rp = rp;
        rc = bag_stats(c, off, S);
        if (rc) return rc;
        if (!d_rp) { HIPCHK(c, c->rp_scratch.ensure((size_t)std::max<int64_t>(c->n_reads, 1) * 4)); d_rp = (float *)c->rp_scratch.p; }
        rc = launch_encode(c, X, km, off, S, c->n_reads, d_rp);
        if (rc) return rc;
        return m6a_validate_pool(c, d_rp, off, S, T, K, seed, y, avg);
    }
    // host pointers: encode through the staging buffers, pool from the staged read probabilities
    if (off[0] != 0) return fail(c, M6A_EINVAL, "off[0] must be 0");
    for (int64_t s = 0; s < S; s++) if (off[s + 1] < off[s]) return fail(c, M6A_EINVAL, "off[] must be non-decreasing");
    const int64_t R = off[S];
    if (R == 0) return fail(c, M6A_EINVAL, "no reads");
    HIPCHK(c, c->sX.ensure((size_t)R * 9 * 4));
    HIPCHK(c, c->sK.ensure((size_t)S * 3));
    HIPCHK(c, c->sOff.ensure((size_t)(S + 1) * 8));
    HIPCHK(c, c->sP.ensure((size_t)R * 4));
    HIPCHK(c, c->val_y.ensure((size_t)T * S * 4));
    HIPCHK(c, c->val_avg.ensure((size_t)S * 4));
    HIPCHK(c, hipMemcpyAsync(c->sX.p, X, (size_t)R * 9 * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->sK.p, km, (size_t)S * 3, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->sOff.p, off, (size_t)(S + 1) * 8, hipMemcpyHostToDevice, c->stream));
    host_bag_range(c, off, S);
    rc = launch_encode(c, (const float *)c->sX.p, (const uint8_t *)c->sK.p, (const int64_t *)c->sOff.p, S, R, (float *)c->sP.p);
    if (rc) return rc;
    rc = launch_validate_pool(c, (const float *)c->sP.p, off, S, T, K, seed, (float *)c->val_y.p, avg ? (float *)c->val_avg.p : nullptr);
    if (rc) return rc;
    if (rp) HIPCHK(c, hipMemcpyAsync(rp, c->sP.p, (size_t)R * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(y, c->val_y.p, (size_t)T * S * 4, hipMemcpyDeviceToHost, c->stream));
    if (avg) HIPCHK(c, hipMemcpyAsync(avg, c->val_avg.p, (size_t)S * 4, hipMemcpyDeviceToHost, c->stream));
    return sync_and_check(c);
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

int m6a_comm_unique_id(void *id_out)
{
    if (!id_out) return M6A_EINVAL;
    Rccl *R = rccl();
    if (!R->err.empty()) return fail(nullptr, M6A_EUNSUPPORTED, "%s", R->err.c_str());
    RcclId id;
    const int e = R->GetUniqueId(&id);
    if (e != 0) return fail(nullptr, M6A_EHIP, "ncclGetUniqueId: %s", R->GetErrorString(e));
    std::memcpy(id_out, id.internal, M6A_COMM_ID_BYTES);
    return M6A_OK;
}

int m6a_comm_init(m6a_ctx *c, const void *unique_id, int rank, int world)
{
    settle(c);
    if (!c) return M6A_EINVAL;
    if (!unique_id || world < 1 || rank < 0 || rank >= world) return fail(c, M6A_EINVAL, "bad communicator arguments");
    if (c->comm) return fail(c, M6A_EINVAL, "the context already has a communicator");
    Rccl *R = rccl();
    if (!R->err.empty()) return fail(c, M6A_EUNSUPPORTED, "%s", R->err.c_str());
    HIPCHK(c, hipSetDevice(c->device));
    RcclId id;
    std::memcpy(id.internal, unique_id, M6A_COMM_ID_BYTES);
    RCCLCHK(c, R, R->CommInitRank(&c->comm, world, id, rank));
    c->comm_rank = rank; c->comm_world = world;
    return M6A_OK;
}

int m6a_comm_destroy(m6a_ctx *c)
{
    settle(c);
    if (!c) return M6A_EINVAL;
    if (!c->comm) return M6A_OK;
    Rccl *R = rccl();
    HIPCHK(c, hipSetDevice(c->device));
    (void)hipStreamSynchronize(c->stream);
    void *comm = c->comm;
    c->comm = nullptr; c->comm_world = 0;                  // whatever CommDestroy says: m6a_destroy must not destroy it again
    RCCLCHK(c, R, R->CommDestroy(comm));
    return M6A_OK;
}

// What the communicator itself says (not what the launcher asked for): how a bench line or a launcher certifies that RCCL
// really formed an N-rank communicator on the devices it meant.
int m6a_comm_count(m6a_ctx *c, int *ranks_seen)
{
    settle(c);
    if (!c) return M6A_EINVAL;
    if (!ranks_seen) return fail(c, M6A_EINVAL, "null pointer argument");
    if (!c->comm) return fail(c, M6A_EINVAL, "m6a_comm_init has not run on this context");
    Rccl *R = rccl();
    if (!R->CommCount) return fail(c, M6A_EUNSUPPORTED, "librccl lacks ncclCommCount");
    RCCLCHK(c, R, R->CommCount(c->comm, ranks_seen));
    return M6A_OK;
}

int m6a_comm_info(m6a_ctx *c, int *rank, int *device, int *rccl_version)
{
    settle(c);
    if (!c) return M6A_EINVAL;
    if (!c->comm) return fail(c, M6A_EINVAL, "m6a_comm_init has not run on this context");
    Rccl *R = rccl();
    if (rank) { *rank = -1; if (R->CommUserRank) RCCLCHK(c, R, R->CommUserRank(c->comm, rank)); }
    if (device) { *device = -1; if (R->CommCuDevice) RCCLCHK(c, R, R->CommCuDevice(c->comm, device)); }
    if (rccl_version) { *rccl_version = 0; if (R->GetVersion) RCCLCHK(c, R, R->GetVersion(rccl_version)); }
    return M6A_OK;
}

int m6a_device_link(int dev_a, int dev_b, int *link_type, int *hops, int *peer_access)
{
    if (link_type) *link_type = -1;
    if (hops) *hops = -1;
    if (peer_access) *peer_access = 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return M6A_ENODEV; }
    if (dev_a < 0 || dev_b < 0 || dev_a >= n || dev_b >= n) return M6A_EINVAL;
    if (dev_a == dev_b) { if (hops) *hops = 0; if (peer_access) *peer_access = 1; return M6A_OK; }
    uint32_t lt = 0, hc = 0;
    if (hipExtGetLinkTypeAndHopCount(dev_a, dev_b, &lt, &hc) != hipSuccess) { (void)hipGetLastError(); return M6A_EHIP; }
    if (link_type) *link_type = (int)lt;
    if (hops) *hops = (int)hc;
    int pa = 0;
    if (hipDeviceCanAccessPeer(&pa, dev_a, dev_b) != hipSuccess) { (void)hipGetLastError(); pa = 0; }
    if (peer_access) *peer_access = pa;
    return M6A_OK;
}

namespace {

struct GatherArray { const void *src; void *out; int dtype; size_t esz; const char *name; };

// ONE grouped exchange on the context's stream: every rank (dst included) sends its slice of each array, dst posts the
// matching receives at the shards' offsets -- direct peer-to-peer writes over xGMI, no ring, no padding.  A failing
// Send/Recv must not leave the thread's RCCL group open (every later RCCL call of the thread would queue into it):
// remember the first error, always close the group.  All pointers are device pointers.
int gather_group(m6a_ctx *c, const GatherArray *arr, int n_arr, const int64_t *cuts, int dst)
{
    Rccl *R = rccl();
    const int W = c->comm_world, me = c->comm_rank;
    const int64_t mine = cuts[me + 1] - cuts[me];
    RCCLCHK(c, R, R->GroupStart());
    int first = 0;
    const char *what = "";
    auto op = [&](int e, const char *w) { if (e != 0 && first == 0) { first = e; what = w; } return first == 0; };
    if (mine > 0)
        for (int a = 0; a < n_arr && first == 0; a++)
            op(R->Send(arr[a].src, (size_t)mine, arr[a].dtype, dst, c->comm, c->stream), arr[a].name);
    if (me == dst)
        for (int r = 0; r < W && first == 0; r++) {
            const int64_t n = cuts[r + 1] - cuts[r];
            if (n <= 0) continue;
            for (int a = 0; a < n_arr && first == 0; a++)
                op(R->Recv((char *)arr[a].out + (size_t)(cuts[r] - cuts[0]) * arr[a].esz, (size_t)n, arr[a].dtype, r, c->comm, c->stream), arr[a].name);
        }
    const int e_end = R->GroupEnd();
    if (first != 0) return fail(c, M6A_EHIP, "RCCL send/recv of %s: %s", what, R->GetErrorString ? R->GetErrorString(first) : "RCCL error");
    if (e_end != 0) return fail(c, M6A_EHIP, "ncclGroupEnd: %s", R->GetErrorString ? R->GetErrorString(e_end) : "RCCL error");
    return M6A_OK;
}

int check_gather_args(m6a_ctx *c, const int64_t *cuts, int dst)
{
    if (!c->comm) return fail(c, M6A_EINVAL, "m6a_comm_init has not run on this context");
    const int W = c->comm_world;
    if (!cuts || dst < 0 || dst >= W) return fail(c, M6A_EINVAL, "bad gather arguments");
    if (is_device_ptr(cuts)) return fail(c, M6A_EINVAL, "shard offsets are a HOST array");
    for (int r = 0; r < W; r++) if (cuts[r + 1] < cuts[r]) return fail(c, M6A_EINVAL, "shard offsets must be non-decreasing");
    return M6A_OK;
}

}  // namespace

int m6a_gather(m6a_ctx *c, const float *site, const double *mod, const int64_t *cuts, int dst, float *site_all, double *mod_all)
{
    settle(c);
    if (!c) return M6A_EINVAL;
    int rc = check_gather_args(c, cuts, dst);
    if (rc) return rc;
    if (c->job.open) return job_busy(c);
    const int W = c->comm_world, me = c->comm_rank;
    const int64_t mine = cuts[me + 1] - cuts[me], total = cuts[W] - cuts[0];
    if (mine > 0 && (!site || !mod)) return fail(c, M6A_EINVAL, "null pointer argument");
    if (me == dst && total > 0 && (!site_all || !mod_all)) return fail(c, M6A_EINVAL, "rank dst needs site_all and mod_all");
    HIPCHK(c, hipSetDevice(c->device));
    const bool recv = me == dst && total > 0;
    const bool dev = mine > 0 ? is_device_ptr(site) : recv ? is_device_ptr(site_all) : true;
    if ((mine > 0 && dev != is_device_ptr(mod)) || (recv && (dev != is_device_ptr(site_all) || dev != is_device_ptr(mod_all))))
        return fail(c, M6A_EINVAL, "site_prob, mod_ratio, site_all, mod_all must be all host or all device pointers");
    GatherArray arr[2] = {{site, site_all, 7 /* ncclFloat32 */, 4, "site_prob"}, {mod, mod_all, 8 /* ncclFloat64 */, 8, "mod_ratio"}};
    if (dev) return gather_group(c, arr, 2, cuts, dst);
    // host arrays: staged through the context's device buffers, synchronous
    if (mine > 0) {
        HIPCHK(c, c->sSite.ensure((size_t)mine * 4));
        HIPCHK(c, c->sMod.ensure((size_t)mine * 8));
        HIPCHK(c, hipMemcpyAsync(c->sSite.p, site, (size_t)mine * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->sMod.p, mod, (size_t)mine * 8, hipMemcpyHostToDevice, c->stream));
        arr[0].src = c->sSite.p; arr[1].src = c->sMod.p;
    }
    if (recv) {
        HIPCHK(c, c->gSite.ensure((size_t)total * 4));
        HIPCHK(c, c->gMod.ensure((size_t)total * 8));
        arr[0].out = c->gSite.p; arr[1].out = c->gMod.p;
    }
    rc = gather_group(c, arr, 2, cuts, dst);
    if (rc) return rc;
    if (recv) {
        rc = d2h_through_ring(c, site_all, c->gSite.p, (size_t)total * 4);
        if (rc) return rc;
        rc = d2h_through_ring(c, mod_all, c->gMod.p, (size_t)total * 8);
        if (rc) return rc;
    }
    return sync_and_check(c);
}

int m6a_gather_reads(m6a_ctx *c, const float *rp, const int64_t *cuts, int dst, float *rp_all)
{
    settle(c);
    if (!c) return M6A_EINVAL;
    int rc = check_gather_args(c, cuts, dst);
    if (rc) return rc;
    if (c->job.open) return job_busy(c);
    const int W = c->comm_world, me = c->comm_rank;
    const int64_t mine = cuts[me + 1] - cuts[me], total = cuts[W] - cuts[0];
    if (mine > 0 && !rp) return fail(c, M6A_EINVAL, "null pointer argument");
    if (me == dst && total > 0 && !rp_all) return fail(c, M6A_EINVAL, "rank dst needs read_all");
    HIPCHK(c, hipSetDevice(c->device));
    const bool recv = me == dst && total > 0;
    const bool dev = mine > 0 ? is_device_ptr(rp) : recv ? is_device_ptr(rp_all) : true;
    if (mine > 0 && recv && dev != is_device_ptr(rp_all)) return fail(c, M6A_EINVAL, "read_prob and read_all must be both host or both device pointers");
    GatherArray arr[1] = {{rp, rp_all, 7 /* ncclFloat32 */, 4, "read_prob"}};
    if (dev) return gather_group(c, arr, 1, cuts, dst);
    if (mine > 0) {
        HIPCHK(c, c->sP.ensure((size_t)mine * 4));
        HIPCHK(c, hipMemcpyAsync(c->sP.p, rp, (size_t)mine * 4, hipMemcpyHostToDevice, c->stream));
        arr[0].src = c->sP.p;
    }
    if (recv) { HIPCHK(c, c->gP.ensure((size_t)total * 4)); arr[0].out = c->gP.p; }
    rc = gather_group(c, arr, 1, cuts, dst);
    if (rc) return rc;
    if (recv) { rc = d2h_through_ring(c, rp_all, c->gP.p, (size_t)total * 4); if (rc) return rc; }
    return sync_and_check(c);
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

const char *m6a_last_pool_variant(const m6a_ctx *c) { return c ? c->pool_variant : "none"; }

}  // extern "C"
