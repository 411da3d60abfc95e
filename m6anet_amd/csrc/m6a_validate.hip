// m6a_validate.hip -- validation-style forward (split out of m6a_api.hip; internal declarations: m6a_ctx.h)
#include "m6a_ctx.h"

using namespace m6a_detail;

namespace m6a_detail {


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
    // The workers are joined on EVERY way out of this function -- a bad_alloc on the walk thread must not unwind past
    // joinable threads (std::terminate) -- and an exception inside a worker is recorded, not thrown across the thread.
    bool worker_failed = false;
    struct Joiner {
        std::vector<std::thread> th;
        std::mutex &mu; std::condition_variable &cv_get, &cv_put; bool &done;
        ~Joiner()
        {
            { std::lock_guard<std::mutex> g(mu); done = true; }
            cv_get.notify_all();
            cv_put.notify_all();
            for (auto &t : th) if (t.joinable()) t.join();
        }
    } workers{{}, mu, cv_get, cv_put, done};
    for (int t = 0; t < n_workers; t++)
        workers.th.emplace_back([&] {
            try {
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
            } catch (...) {                                   // bad_alloc of the permutation buffer: the caller reports M6A_ENOMEM
                std::lock_guard<std::mutex> g(mu);
                worker_failed = true;
                cv_put.notify_all();
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
            {
                // what the block's shuffles will consume, from ITS bags (n - 1 draws each, ~1.4 words per draw under the masked
                // rejection) -- not from the largest bag of the job, which asked for hundreds of MB per block once a single
                // site had tens of thousands of reads
                size_t draws = 0;
                for (int64_t k = b->k0; k < b->k1; k++) { const int64_t s = k % S; draws += (size_t)(h_off[s + 1] - h_off[s]); }
                b->words.reserve(draws * 3 / 2 + (size_t)MtProducer::kBatch);
            }
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
                cv_put.wait(g, [&] { return worker_failed || queue.size() < 4 * (size_t)n_workers; });
                if (worker_failed) throw std::bad_alloc();
                queue.push_back(std::move(b));
            }
            cv_get.notify_one();
        }
    }
    {
        // drain: the workers finish what is queued, then leave
        std::lock_guard<std::mutex> g(mu);
        done = true;
    }
    cv_get.notify_all();
    for (auto &w : workers.th) w.join();
    if (worker_failed) throw std::bad_alloc();
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


}  // namespace m6a_detail

extern "C" {


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


}  // extern "C"
