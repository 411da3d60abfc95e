/* host_ceilings.c -- what the HOST can do at best for the byte work either side of the hot path,
 * measured on the box the numbers are quoted from: tools/host_rooflines.py prices dataprep, the loader and the CSV writers
 * against these.  Plain C + pthreads, no dependency on the product.
 *
 *   host_ceilings scan <file> <threads>          mmap the file (page cache warm after the first pass), every thread takes a
 *                                                contiguous range: (a) memchr('\n') over it -- the least a line-oriented parser
 *                                                must do; (b) a 64-bit sum of every 8 bytes -- the memory system alone
 *   host_ceilings memcpy <MB> <threads>          private src -> dst copies, all threads at once
 *   host_ceilings pwrite <file> <MB> <threads>   every thread pwrite()s its range of a new file from a warm buffer
 * Output: one JSON object per run.
 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

static double now(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}

typedef struct { const char *p; size_t n; int mode; uint64_t out; char *dst; int fd; off_t off; } Job;

static void *work(void *v)
{
    Job *j = (Job *)v;
    if (j->mode == 0) {                      /* newline count */
        const char *p = j->p, *e = j->p + j->n;
        uint64_t c = 0;
        while (p < e) {
            const char *q = memchr(p, '\n', (size_t)(e - p));
            if (!q) break;
            c++; p = q + 1;
        }
        j->out = c;
    } else if (j->mode == 1) {               /* touch: sum of 8-byte words */
        const uint64_t *w = (const uint64_t *)(((uintptr_t)j->p + 7) & ~(uintptr_t)7);
        const size_t n = (size_t)((j->p + j->n) - (const char *)w) / 8;
        uint64_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        size_t i = 0;
        for (; i + 4 <= n; i += 4) { s0 += w[i]; s1 += w[i + 1]; s2 += w[i + 2]; s3 += w[i + 3]; }
        for (; i < n; i++) s0 += w[i];
        j->out = s0 + s1 + s2 + s3;
    } else if (j->mode == 2) {               /* memcpy */
        memcpy(j->dst, j->p, j->n);
        j->out = (uint64_t)(unsigned char)j->dst[j->n / 2];
    } else {                                 /* pwrite */
        size_t left = j->n;
        const char *p = j->p;
        off_t o = j->off;
        while (left) {
            ssize_t w = pwrite(j->fd, p, left > (64u << 20) ? (64u << 20) : left, o);
            if (w <= 0) { j->out = 1; return 0; }
            p += w; o += w; left -= (size_t)w;
        }
    }
    return 0;
}

static double run(Job *jobs, int T)
{
    pthread_t th[256];
    const double t0 = now();
    for (int i = 1; i < T; i++) pthread_create(&th[i], 0, work, &jobs[i]);
    work(&jobs[0]);
    for (int i = 1; i < T; i++) pthread_join(th[i], 0);
    return now() - t0;
}

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: see the header of host_ceilings.c\n"); return 2; }
    Job jobs[256];
    memset(jobs, 0, sizeof jobs);
    if (!strcmp(argv[1], "scan")) {
        const int T = atoi(argv[3]) > 0 && atoi(argv[3]) <= 256 ? atoi(argv[3]) : 1;
        const int fd = open(argv[2], O_RDONLY);
        struct stat st;
        if (fd < 0 || fstat(fd, &st) != 0) { perror(argv[2]); return 1; }
        const size_t n = (size_t)st.st_size;
        const char *p = mmap(0, n, PROT_READ, MAP_SHARED | MAP_POPULATE, fd, 0);
        if (p == MAP_FAILED) { perror("mmap"); return 1; }
        double best[2] = {1e30, 1e30};
        uint64_t lines = 0;
        for (int rep = 0; rep < 3; rep++)
            for (int mode = 0; mode < 2; mode++) {
                for (int i = 0; i < T; i++) { jobs[i].p = p + n * (size_t)i / T; jobs[i].n = n * (size_t)(i + 1) / T - n * (size_t)i / T; jobs[i].mode = mode; }
                const double dt = run(jobs, T);
                if (dt < best[mode]) best[mode] = dt;
                if (mode == 0) { lines = 0; for (int i = 0; i < T; i++) lines += jobs[i].out; }
            }
        printf("{\"what\": \"scan\", \"bytes\": %zu, \"threads\": %d, \"lines\": %llu, \"memchr_newline_GBps\": %.3f, \"sum_words_GBps\": %.3f}\n",
               n, T, (unsigned long long)lines, n / best[0] / 1e9, n / best[1] / 1e9);
    } else if (!strcmp(argv[1], "memcpy")) {
        const size_t mb = (size_t)atol(argv[2]);
        const int T = atoi(argv[3]) > 0 && atoi(argv[3]) <= 256 ? atoi(argv[3]) : 1;
        const size_t per = (mb << 20) / (size_t)T;
        for (int i = 0; i < T; i++) {
            char *s = malloc(per), *d = malloc(per);
            if (!s || !d) { fprintf(stderr, "out of memory\n"); return 1; }
            memset(s, i + 1, per); memset(d, 0, per);
            jobs[i].p = s; jobs[i].dst = d; jobs[i].n = per; jobs[i].mode = 2;
        }
        double best = 1e30;
        for (int rep = 0; rep < 5; rep++) { const double dt = run(jobs, T); if (dt < best) best = dt; }
        printf("{\"what\": \"memcpy\", \"bytes\": %zu, \"threads\": %d, \"GBps_copied\": %.3f}\n", per * (size_t)T, T, per * (size_t)T / best / 1e9);
    } else if (!strcmp(argv[1], "pwrite")) {
        if (argc < 5) return 2;
        const size_t mb = (size_t)atol(argv[3]);
        const int T = atoi(argv[4]) > 0 && atoi(argv[4]) <= 256 ? atoi(argv[4]) : 1;
        const size_t per = (mb << 20) / (size_t)T;
        char *buf = malloc(per);
        if (!buf) return 1;
        for (size_t i = 0; i < per; i++) buf[i] = (char)('0' + i % 10);
        double best = 1e30;
        for (int rep = 0; rep < 3; rep++) {
            const int fd = open(argv[2], O_WRONLY | O_CREAT | O_TRUNC, 0644);
            if (fd < 0) { perror(argv[2]); return 1; }
            for (int i = 0; i < T; i++) { jobs[i].p = buf; jobs[i].n = per; jobs[i].mode = 3; jobs[i].fd = fd; jobs[i].off = (off_t)(per * (size_t)i); jobs[i].out = 0; }
            const double dt = run(jobs, T);
            close(fd);
            if (dt < best) best = dt;
        }
        unlink(argv[2]);
        printf("{\"what\": \"pwrite\", \"bytes\": %zu, \"threads\": %d, \"GBps_written\": %.3f}\n", per * (size_t)T, T, per * (size_t)T / best / 1e9);
    } else return 2;
    return 0;
}
