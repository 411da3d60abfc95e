/* feed_probe.c -- the streaming job API of include/m6a.h driven from plain C, the way a non-Python host would: dlopen the
 * library, m6a_create, then the reference's batch loop (16-site batches) through m6a_job_begin / m6a_job_feed / m6a_job_end,
 * and the same job as ONE m6a_infer for comparison (results must be bit-identical).  Prints one JSON line.
 *   build: gcc -O2 -Iinclude tools/feed_probe.c -ldl -o tools/feed_probe
 *   run:   tools/feed_probe [n_sites=20000] [batch=16] [lo=20] [hi=90] [n_iters=1000]        (on an MI355X)            */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "m6a.h"

static double now(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

#define SYM(name) __typeof__(&name) p_##name = (__typeof__(&name))dlsym(lib, #name); if (!p_##name) { fprintf(stderr, "missing %s\n", #name); return 2; }

int main(int argc, char **argv)
{
    const int64_t S = argc > 1 ? atoll(argv[1]) : 20000;
    const int64_t batch = argc > 2 ? atoll(argv[2]) : 16;
    const int lo = argc > 3 ? atoi(argv[3]) : 20, hi = argc > 4 ? atoi(argv[4]) : 90;
    const int T = argc > 5 ? atoi(argv[5]) : 1000;
    void *lib = dlopen("m6anet_amd/libm6a_hip.so", RTLD_NOW);
    if (!lib) { fprintf(stderr, "%s\n", dlerror()); return 2; }
    SYM(m6a_create) SYM(m6a_destroy) SYM(m6a_last_error) SYM(m6a_infer) SYM(m6a_job_begin) SYM(m6a_job_feed) SYM(m6a_job_end)
    SYM(m6a_prepare_host_io)
    static float w[M6A_N_WEIGHTS];
    FILE *f = fopen("m6anet_amd/assets/weights_hct116.bin", "rb");
    if (!f || fread(w, 4, M6A_N_WEIGHTS, f) != M6A_N_WEIGHTS) { fprintf(stderr, "cannot read the weights\n"); return 2; }
    fclose(f);
    int64_t *off = malloc((size_t)(S + 1) * 8);
    off[0] = 0;
    for (int64_t s = 0; s < S; s++) off[s + 1] = off[s] + lo + (int64_t)(rnd() % (uint64_t)(hi - lo + 1));
    const int64_t R = off[S];
    float *X = malloc((size_t)R * 9 * 4), *rp = malloc((size_t)R * 4), *rp2 = malloc((size_t)R * 4);
    uint8_t *km = malloc((size_t)S * 3);
    float *site = malloc((size_t)S * 4), *site2 = malloc((size_t)S * 4);
    double *mod = malloc((size_t)S * 8), *mod2 = malloc((size_t)S * 8);
    int64_t *boff = malloc((size_t)(batch + 1) * 8);
    for (int64_t i = 0; i < R * 9; i++) X[i] = (float)((double)(rnd() >> 11) / 9007199254740992.0 * 4.0 - 2.0);
    for (int64_t i = 0; i < S * 3; i++) km[i] = (uint8_t)(rnd() % 66);
    m6a_ctx *h = NULL;
    if (p_m6a_create(&h, w, M6A_N_WEIGHTS, 0)) { fprintf(stderr, "m6a_create: %s\n", p_m6a_last_error(NULL)); return 1; }
    p_m6a_prepare_host_io(h);
    const float thr = 0.033379376f;
    double best_feed = 1e9, best_total = 1e9, best_infer = 1e9;
    for (int rep = 0; rep < 4; rep++) {
        double t0 = now();
        if (p_m6a_job_begin(h, T, 20, thr, 0, 0, 16, 2, 0, 0)) { fprintf(stderr, "begin: %s\n", p_m6a_last_error(h)); return 1; }
        for (int64_t s0 = 0; s0 < S; s0 += batch) {
            const int64_t s1 = s0 + batch < S ? s0 + batch : S;
            for (int64_t i = 0; i <= s1 - s0; i++) boff[i] = off[s0 + i] - off[s0];
            if (p_m6a_job_feed(h, X + off[s0] * 9, km + s0 * 3, boff, s1 - s0)) { fprintf(stderr, "feed: %s\n", p_m6a_last_error(h)); return 1; }
        }
        double t1 = now();
        if (p_m6a_job_end(h, rp, site, mod)) { fprintf(stderr, "end: %s\n", p_m6a_last_error(h)); return 1; }
        double t2 = now();
        if (rep && t1 - t0 < best_feed) best_feed = t1 - t0;
        if (rep && t2 - t0 < best_total) best_total = t2 - t0;
        t0 = now();
        if (p_m6a_infer(h, X, km, off, S, T, 20, thr, 0, 0, 16, 2, rp2, site2, mod2)) { fprintf(stderr, "infer: %s\n", p_m6a_last_error(h)); return 1; }
        t1 = now();
        if (rep && t1 - t0 < best_infer) best_infer = t1 - t0;
    }
    const int same = !memcmp(rp, rp2, (size_t)R * 4) && !memcmp(site, site2, (size_t)S * 4) && !memcmp(mod, mod2, (size_t)S * 8);
    printf("{\"sites\": %lld, \"reads\": %lld, \"batch_sites\": %lld, \"n_iters\": %d, \"feed_calls\": %lld, \"feed_loop_s\": %.6f, "
           "\"us_per_feed_call\": %.3f, \"begin_to_end_s\": %.6f, \"streamed_sites_per_s\": %.0f, \"one_m6a_infer_s\": %.6f, "
           "\"one_m6a_infer_sites_per_s\": %.0f, \"bit_identical_to_m6a_infer\": %s, \"host\": \"plain C through dlopen, best of 3\"}\n",
           (long long)S, (long long)R, (long long)batch, T, (long long)((S + batch - 1) / batch), best_feed,
           best_feed / (double)((S + batch - 1) / batch) * 1e6, best_total, (double)S / best_total, best_infer, (double)S / best_infer,
           same ? "true" : "false");
    p_m6a_destroy(h);
    return same ? 0 : 3;
}
