// walker_probe.hip -- VERDICT r2 "Next #9": the validation sampler's sequential walk as a single-wavefront GPU kernel, MEASURED.
//
// The training-mode sampler of m6anet's validate() (m6anet/utils/data_utils.py:213-214 under training_utils.py:235-240) is ONE
// chain over one MT19937 stream: item (pass, site) shuffles arange(n) -- for i = n-1..1 draw words until (w & mask(i)) <= i --
// and where an item starts depends on every rejection before it.  The library splits it into a counting walk (where does each
// item start?) and independent per-item shuffles; the walk is the serial part.  This probe runs that walk on the GPU the only
// way a chain can run there -- one wavefront, 64 stream words per step, the accept/reject decisions of the 64 lanes resolved by
// fixed-point iteration (lane l tests its word against draw slot G0 + a_l, a_l = accepted lanes before it; iterate until the
// ballot stops changing) -- checks every item's start position against the host walk, and times both.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/walker_probe.hip -o tools/walker_probe
//   run:   tools/walker_probe [sites=200000] [passes=5] [lo=50] [hi=500]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Mt {
    uint32_t s[624], out[624];
    explicit Mt(uint32_t seed) { uint32_t x = seed; s[0] = x; for (uint32_t i = 1; i < 624; i++) { x = 1812433253u * (x ^ (x >> 30)) + i; s[i] = x; } }
    static uint32_t tw(uint32_t a, uint32_t b) { const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu); return (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu); }
    void refill()
    {
        for (int k = 0; k < 227; k++) s[k] = s[k + 397] ^ tw(s[k], s[k + 1]);
        for (int k = 227; k < 623; k++) s[k] = s[k - 227] ^ tw(s[k], s[k + 1]);
        s[623] = s[396] ^ tw(s[623], s[0]);
        for (int k = 0; k < 624; k++) { uint32_t y = s[k]; y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18; out[k] = y; }
    }
};

// D[k] = number of draw slots before item k (item k has n_k - 1 slots, i = n_k - 1 .. 1); D has n_items + 9 entries, the
// tail repeated so that the window loads below never leave the array
__global__ __launch_bounds__(64) void walk_kernel(const uint32_t *stream, const uint32_t *D, uint32_t n_items, uint32_t *start,
                                                  unsigned long long *iters_out)
{
    const int lane = threadIdx.x;
    uint32_t p = 0, G0 = 0, k = 0;
    unsigned long long iters = 0;
    if (lane == 0) start[0] = 0;
    while (k < n_items) {
        const uint32_t d1 = D[k + 1], d2 = D[k + 2], d3 = D[k + 3], d4 = D[k + 4], d5 = D[k + 5];   // wave-uniform (scalar loads)
        const uint32_t w = stream[p + lane];
        unsigned long long acc = ~0ull, prev;
        uint32_t g;
        do {                                                   // fixed point: the ballot that reproduces itself
            prev = acc;
            const uint32_t a = __builtin_amdgcn_mbcnt_hi((uint32_t)(prev >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)prev, 0u));
            g = G0 + a;
            const uint32_t end = g < d1 ? d1 : g < d2 ? d2 : g < d3 ? d3 : g < d4 ? d4 : d5;
            const uint32_t i = end - g;                        // the slot's i (>= 1 while slots remain)
            const uint32_t mask = 0xffffffffu >> __builtin_clz(i | 1u);
            acc = __ballot((w & mask) <= i && g < D[n_items]);
            iters++;
        } while (acc != prev);
        // an accepted word that fills the last slot of an item: the next item starts right after it
        if ((acc >> lane) & 1ull) {
            const uint32_t g1 = g + 1;
            if (g1 == d1) start[k + 1] = p + lane + 1;
            else if (g1 == d2) start[k + 2] = p + lane + 1;
            else if (g1 == d3) start[k + 3] = p + lane + 1;
            else if (g1 == d4) start[k + 4] = p + lane + 1;
        }
        G0 += (uint32_t)__popcll(acc);
        p += 64;
        while (k < n_items && G0 >= D[k + 1]) k++;
    }
    if (lane == 0) *iters_out = iters;
}

int main(int argc, char **argv)
{
    const int64_t S = argc > 1 ? atoll(argv[1]) : 200000;
    const int T = argc > 2 ? atoi(argv[2]) : 5;
    const int lo = argc > 3 ? atoi(argv[3]) : 50, hi = argc > 4 ? atoi(argv[4]) : 500;
    std::vector<uint32_t> n((size_t)S);
    uint64_t r = 12345;
    for (auto &v : n) { r = r * 6364136223846793005ull + 1442695040888963407ull; v = (uint32_t)(lo + (r >> 33) % (uint64_t)(hi - lo + 1)); }
    const uint32_t n_items = (uint32_t)(S * T);
    std::vector<uint32_t> D((size_t)n_items + 9);
    uint64_t tot = 0;
    for (uint32_t k = 0; k < n_items; k++) { D[k] = (uint32_t)tot; tot += n[k % S] - 1; }
    for (size_t k = n_items; k < D.size(); k++) D[k] = (uint32_t)tot;
    if (tot >= 0xffffff00ull) { printf("too many draws for the probe's 32-bit counters\n"); return 1; }
    // the stream: twice the draws is plenty (acceptance >= 1/2); host walk = the library's counting loop
    const size_t n_words = (size_t)(tot * 2 + 4096) / 624 * 624 + 624;
    std::vector<uint32_t> stream(n_words);
    {
        Mt g(7);
        for (size_t q = 0; q < n_words; q += 624) { g.refill(); std::copy(g.out, g.out + 624, stream.begin() + q); }
    }
    std::vector<uint32_t> want(n_items);
    auto t0 = std::chrono::steady_clock::now();
    {
        size_t p = 0;
        for (uint32_t k = 0; k < n_items; k++) {
            want[k] = (uint32_t)p;
            uint32_t i = n[k % S] - 1;
            while (i) {
                const uint32_t mask = 0xffffffffu >> __builtin_clz(i), lo_i = (mask >> 1) + 1;
                for (;;) { i -= ((stream[p++] & mask) <= i); if (i < lo_i) break; }
            }
        }
        printf("host walk: %zu words consumed, %.3f s (one thread, stream already in memory)\n", p,
               std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    uint32_t *d_stream, *d_D, *d_start;
    unsigned long long *d_iters, iters = 0;
    CHK(hipMalloc(&d_stream, n_words * 4 + 256));
    CHK(hipMalloc(&d_D, D.size() * 4));
    CHK(hipMalloc(&d_start, (size_t)(n_items + 8) * 4));
    CHK(hipMalloc(&d_iters, 8));
    CHK(hipMemcpy(d_stream, stream.data(), n_words * 4, hipMemcpyHostToDevice));
    CHK(hipMemcpy(d_D, D.data(), D.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(walk_kernel, dim3(1), dim3(64), 0, 0, d_stream, d_D, n_items, d_start, d_iters);
        hipEventRecord(e1);
        CHK(hipDeviceSynchronize());
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<uint32_t> got(n_items);
    CHK(hipMemcpy(got.data(), d_start, (size_t)n_items * 4, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(&iters, d_iters, 8, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (uint32_t k = 0; k < n_items; k++) bad += got[k] != want[k];
    const double steps = (double)want[n_items - 1] / 64.0;
    printf("GPU walk (one wavefront, 64 words per step, fixed-point ballots): %.3f s, %.2f ballot rounds per step, %.0f cycles per step "
           "at 2.4 GHz; start positions of %u items %s the host walk's\n", ms * 1e-3, (double)iters / steps, ms * 1e-3 * 2.4e9 / steps, n_items,
           bad ? "DIFFER from" : "equal");
    return bad ? 3 : 0;
}
