#include "m6a_io.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char **argv)
{
    const char *dirs[1] = {argv[1]};
    m6a_sites *s = nullptr;
    int rc = m6a_io_load_sites(dirs, 1, 20, nullptr, nullptr, nullptr, 0, 8, &s);
    printf("load rc %d %s\n", rc, rc ? m6a_io_last_error() : "");
    if (rc) return 1;
    const int64_t S = m6a_io_n_sites(s), R = m6a_io_n_reads(s);
    std::vector<float> rp(R, 0.25f), sp(S, 0.5f);
    std::vector<double> mr(S, 0.125);
    rc = m6a_io_write_csv(s, argv[2], rp.data(), sp.data(), mr.data(), 1, 8);
    int64_t a = 0, b = 0;
    const int64_t off0 = m6a_io_off(s)[0], cut = S / 2;
    rc |= m6a_io_csv_shard_size(s, rp.data(), sp.data(), mr.data(), 0, cut, 8, &a, &b);
    rc |= m6a_io_csv_shard_write(s, argv[2], rp.data(), sp.data(), mr.data(), 0, cut, 8, m6a_io_csv_header_bytes(0), m6a_io_csv_header_bytes(1), 1, -1, -1);
    printf("rc %d sites %lld reads %lld shard bytes %lld %lld (off0 %lld)\n", rc, (long long)S, (long long)R, (long long)a, (long long)b, (long long)off0);
    m6a_io_free(s);
    return rc;
}
