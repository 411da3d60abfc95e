// m6a_ctx.h -- internal to libm6a_hip.so: the context behind include/m6a.h and what its translation units share.
//   m6a_api.hip        the C ABI proper: context, weights, random stream, pairwise-sum plan, index tables, launches, encode / pool / infer
//   m6a_host_ring.hip  host-pointer calls: pinned staging ring, copy threads, H2D || encoder || D2H
//   m6a_job.hip        the streaming job (m6a_job_begin / feed / end): the reference's batch loop fed as the loader produces it
//   m6a_comm.hip       RCCL bound at run time, m6a_comm_*, m6a_gather, m6a_gather_reads, m6a_device_link
//   m6a_validate.hip   validation forward: the without-replacement sampler on host threads, m6a_validate / m6a_validate_pool
#pragma once
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
    unsigned long long *d_clk = nullptr;      // [2 kinds][M6A_CLK_SLOTS][4] clock stamps of the last profiled launch of each kind
    unsigned long long *clk_for(int kind) const { return on && (mask >> kind & 1) ? d_clk + (size_t)kind * M6A_CLK_SLOTS * 4 : nullptr; }
    bool on = false;
    int mask = 3;                             // bit 0: time the encoder launches, bit 1: the pooling launches
    std::vector<hipEvent_t> start[2], stop[2];
    int used[2] = {0, 0};
    int64_t dropped[2] = {0, 0};
};


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
    int enc_variant = 0;                              // 0 auto (= the 16-slot kernels), 1 16-slot, 2 12-slot (bags >= 16), 3 16-slot per-lane walk, 4 fast (12-slot where it applies)
    const char *enc_variant_used = "none";
    const char *enc_kernel_used = "none";             // the __global__ function the last encode launched
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
    DevBuf ctl_dev, rt_rank, rt_order;
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
    std::vector<void *> graveyard;            // outgrown device blocks, freed when the context is idle (hipFree stalls behind running kernels)
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

namespace m6a_detail {

int fail(m6a_ctx *c, int code, const char *fmt, ...);

#define HIPCHK(c, expr)                                                                    \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            (void)hipGetLastError();                                                       \
            return fail((c), e_ == hipErrorOutOfMemory ? M6A_ENOMEM : M6A_EHIP, "%s: %s (%s:%d)", \
                        #expr, hipGetErrorString(e_), __FILE__, __LINE__);                 \
        }                                                                                  \
    } while (0)

// ---- shared between the translation units (definitions: see the list at the top) ----------------------------------
bool is_device_ptr(const void *p);
// Every entry point that touches the stream, the sampling state or the error text waits for m6a_create's background set-up first.
inline void settle(m6a_ctx *c) { if (c && c->warm.joinable()) c->warm.join(); }
int job_busy(m6a_ctx *c);
int check_pool_args(m6a_ctx *c, int64_t S, int T, int K, int rng_mode, int64_t bs, int64_t spb);
int deferred_error(m6a_ctx *c);
int sync_and_check(m6a_ctx *c);
// m6a_set_host_offsets is one-shot: whichever entry point runs next consumes the hint -- on its device-pointer branch
// through bag_stats, on every other path (host pointers, argument errors) by leaving this scope.  A pointer that stayed
// armed would describe some later call's off[] wrongly (or point at memory the caller has freed by then).
struct HintScope {
    m6a_ctx *c;
    explicit HintScope(m6a_ctx *ctx) : c(ctx) {}
    ~HintScope() { if (c) c->hint_off = nullptr; }
};
// m6a_api.hip
int ensure_raw(m6a_ctx *c, uint32_t seed, int64_t len);
int launch_stream(m6a_ctx *c, uint32_t seed, int64_t len, uint32_t *raw);
int launch_encode(m6a_ctx *c, const float *X, const uint8_t *km, const int64_t *off, int64_t S, int64_t R, float *rp);
int launch_pool(m6a_ctx *c, const float *rp, const int64_t *off, int64_t S, int T, int K, float thr, uint32_t seed, int64_t bs, int64_t spb,
                float *site, double *mod, bool dry = false);
int bag_stats(m6a_ctx *c, const int64_t *d_off, int64_t S);
void host_bag_range(m6a_ctx *c, const int64_t *off, int64_t S);
void prof_begin(m6a_ctx *c, int kind);
void prof_end(m6a_ctx *c, int kind);
// m6a_host_ring.hip
int ensure_staging(m6a_ctx *c);
void release_staging(m6a_ctx *c);
int staged_encode(m6a_ctx *c, const float *X, const uint8_t *km, const int64_t *off, int64_t S, int64_t R, float *rp_host);
int staged_outputs(m6a_ctx *c, int64_t S, float *site, double *mod);
int d2h_through_ring(m6a_ctx *c, void *host, const void *dev, size_t bytes);
bool base_is_group_start(int64_t base, int64_t bs, int64_t spb);
// m6a_comm.hip
void comm_release(m6a_ctx *c);                 // m6a_destroy: drops the communicator without touching the error text

}  // namespace m6a_detail

