// m6a_io.cpp -- native loader for dataprep output and CSV writers (include/m6a_io.h).
// Host-only C++17; parsing and formatting are spread over std::threads by site range.
#include "m6a_io.h"
#include "m6a_host_cpus.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <charconv>
#include <chrono>
#include <clocale>
#include <cstdlib>
#include <locale.h>
#include <cerrno>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <array>
#include <atomic>
#include <condition_variable>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <string_view>
#include <limits>
#include <memory>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[768];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

struct Mapped {
    const char *p = nullptr;
    size_t n = 0;
    int fd = -1;
    // populate_max: files up to this size are mapped with MAP_POPULATE (one pass over the page tables instead of a fault
    // per page per thread); larger ones -- an eventalign.txt is tens to hundreds of GB -- are mapped lazily and read ahead
    // sequentially, so neither the address space nor the page cache has to hold the file at once
    int open(const std::string &path, size_t populate_max = ~(size_t)0)
    {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return fail(M6A_IO_EIO, "cannot open %s", path.c_str());
        struct stat st;
        if (fstat(fd, &st) != 0) return fail(M6A_IO_EIO, "cannot stat %s", path.c_str());
        n = (size_t)st.st_size;
        if (n) {
            const bool populate = n <= populate_max;
            void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE | (populate ? MAP_POPULATE : 0), fd, 0);
            if (m == MAP_FAILED) return fail(M6A_IO_EIO, "cannot mmap %s", path.c_str());
            p = (const char *)m;
            if (!populate) (void)madvise(m, n, MADV_SEQUENTIAL);
        }
        return 0;
    }
    ~Mapped()
    {
        if (p) munmap((void *)p, n);
        if (fd >= 0) ::close(fd);
    }
};

// the 66-word vocabulary: sorted unique 5-mers of all N-DRACH-N 7-mers
// (m6anet/utils/constants.py:29-36, m6anet/utils/data_utils.py:89-96)
const std::map<std::string, int> &vocab()
{
    static const std::map<std::string, int> v = [] {
        std::set<std::string> s;
        const std::string N = "ACGT", D = "AGT", R = "GA", H = "ACT";
        for (char a : N) for (char d : D) for (char r : R) for (char h : H) for (char b : N) {
            const std::string k7 = {a, d, r, 'A', 'C', h, b};
            for (int i = 0; i < 3; i++) s.insert(k7.substr(i, 5));
        }
        std::map<std::string, int> m;
        int i = 0;
        for (const auto &k : s) m[k] = i++;
        return m;
    }();
    return v;
}

struct Part { int rep; int64_t start, end; };
// one row of data.info.  `tx` points into the file's text (kept alive by the caller), and a site of a single input
// directory keeps its one byte range inline: no heap allocation per row (three of them made the parse 0.4 us per row)
struct SiteRef {
    std::string_view tx;
    int64_t pos;
    int64_t n_reads = 0;
    Part one{-1, 0, 0};
    std::vector<Part> more;              // replicates: one range per directory that has the site
    const Part *parts_begin() const { return more.empty() ? &one : more.data(); }
    size_t n_parts() const { return more.empty() ? (one.rep >= 0 ? 1 : 0) : more.size(); }
};

// `index` = nullptr for a single directory: every row is a new site, no union over replicates to build
int parse_info(const std::string &dir, int rep, std::vector<SiteRef> &sites,
               std::unordered_map<std::string, size_t> *index, std::string &text)
{
    const std::string path = dir + "/data.info";
    // the whole file in one read, rows split with memchr, integers with from_chars: fgets + sscanf took 0.1 s for the
    // 141 k rows of a 900 MB dataset -- most of what the loader did NOT do in parallel
    text.clear();                                           // the SiteRefs keep views into it
    {
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) return fail(M6A_IO_EIO, "cannot open %s", path.c_str());
        char buf[1 << 16];
        size_t k;
        while ((k = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, k);
        const bool bad = ferror(f) != 0;
        fclose(f);
        if (bad) return fail(M6A_IO_EIO, "cannot read %s", path.c_str());
    }
    const char *p = text.data(), *const end = p + text.size();
    bool first = true;
    sites.reserve(sites.size() + (size_t)std::count(p, end, '\n'));
    while (p < end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;                         // line = [p, le)
        const char *next = nl ? nl + 1 : end;
        if (first) {          // header: transcript_id,transcript_position,start,end,n_reads
            first = false;
            if (le - p < 51 || strncmp(p, "transcript_id,transcript_position,start,end,n_reads", 51) != 0)
                return fail(M6A_IO_EFORMAT, "%s: unexpected header", path.c_str());
            p = next;
            continue;
        }
        const char *c1 = (const char *)memchr(p, ',', (size_t)(le - p));
        if (!c1) { p = next; continue; }
        long long v[4];
        const char *q = c1 + 1;
        bool ok = true;
        for (int i = 0; i < 4 && ok; i++) {
            const auto r = std::from_chars(q, le, v[i]);
            ok = r.ec == std::errc() && (i == 3 || (r.ptr < le && *r.ptr == ','));
            q = r.ptr + 1;
        }
        if (!ok) return fail(M6A_IO_EFORMAT, "%s: bad row '%.*s'", path.c_str(), (int)std::min<ptrdiff_t>(le - p, 200), p);
        const long long pos = v[0], start = v[1], end_b = v[2], n = v[3];
        size_t i;
        const std::string_view tx(p, (size_t)(c1 - p));
        if (!index) {
            i = sites.size();
            sites.push_back(SiteRef{tx, pos, 0, Part{rep, start, end_b}, {}});
        } else {
            const std::string key = std::string(tx) + ":" + std::to_string(pos);
            auto it = index->find(key);
            if (it == index->end()) {
                i = sites.size();
                index->emplace(key, i);
                sites.push_back(SiteRef{tx, pos, 0, Part{-1, 0, 0}, {}});
            } else {
                i = it->second;
            }
            sites[i].more.push_back(Part{rep, start, end_b});
        }
        sites[i].n_reads += n;
        p = next;
    }
    return 0;
}

// --- minimal JSON walker for one dataprep record: {"tx":{"pos":{"KMER":[[n,...],[n,...]]}}} -------
struct Cursor {
    const char *p, *e;
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p; }
    bool eat(char c) { ws(); if (p < e && *p == c) { ++p; return true; } return false; }
    bool str(std::string &out)
    {
        ws();
        if (p >= e || *p != '"') return false;
        const char *q = ++p;
        while (p < e && *p != '"') { if (*p == '\\') ++p; ++p; }
        if (p >= e) return false;
        out.assign(q, p - q);
        ++p;
        return true;
    }
    // Correctly rounded decimal -> double, like Python's float().  (libstdc++ 11's
    // std::from_chars(double) switches locale under a global lock -- it serialises the worker
    // threads -- so: Clinger's exact fast path for <= 15 significant digits and |exp10| <= 22,
    // glibc strtod_l on a cached "C" locale for everything else -- on a NUL-terminated copy of the
    // token: the mapped file is not NUL-terminated, and only [0-9+-.eE] is a JSON / eventalign number
    // (strtod alone would also take "inf", "nan" and hex floats); NaN / Infinity / -Infinity, which Python's
    // json module reads and writes, are recognised by name.)
    bool num(double &v)
    {
        ws();
        const char *q = p;
        bool neg = false;
        if (q < e && (*q == '-' || *q == '+')) { neg = *q == '-'; ++q; }
        // the three non-finite literals Python's json.loads takes (and json.dumps writes): NaN, Infinity, -Infinity
        if (q < e && (*q == 'N' || *q == 'I')) {
            if (e - q >= 3 && std::memcmp(q, "NaN", 3) == 0 && q == p) { v = std::numeric_limits<double>::quiet_NaN(); p = q + 3; return true; }
            if (e - q >= 8 && std::memcmp(q, "Infinity", 8) == 0 && (q == p || neg)) {
                v = neg ? -std::numeric_limits<double>::infinity() : std::numeric_limits<double>::infinity();
                p = q + 8;
                return true;
            }
            return false;
        }
        uint64_t mant = 0;
        int nd = 0, frac = 0;
        bool dot = false, any = false;
        for (; q < e; ++q) {
            const char ch = *q;
            if (ch >= '0' && ch <= '9') {
                any = true;
                if (mant || ch != '0') {
                    if (++nd > 15) break;                 // too many significant digits: slow path
                    mant = mant * 10 + (uint64_t)(ch - '0');
                }
                if (dot) ++frac;
            } else if (ch == '.' && !dot) {
                dot = true;
            } else {
                break;
            }
        }
        const bool simple_end = q >= e || (*q != 'e' && *q != 'E' && !(*q >= '0' && *q <= '9') && *q != '.');
        if (any && nd <= 15 && simple_end && frac <= 22) {
            static const double p10[] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14,
                                         1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
            const double r = (double)mant / p10[frac];    // both exact doubles: one correctly rounded division
            v = neg ? -r : r;
            p = q;
            return true;
        }
        static const locale_t c_loc = newlocale(LC_ALL_MASK, "C", (locale_t)0);
        char tok[96];
        size_t len = 0;
        for (const char *t = p; t < e && len + 1 < sizeof tok; ++t, ++len) {
            const char ch = *t;
            if (!((ch >= '0' && ch <= '9') || ch == '+' || ch == '-' || ch == '.' || ch == 'e' || ch == 'E')) break;
            tok[len] = ch;
        }
        tok[len] = 0;
        if (len == 0 || len + 1 >= sizeof tok) return false;
        char *end = nullptr;
        v = strtod_l(tok, &end, c_loc);
        if (end == tok) return false;
        p += end - tok;
        return true;
    }
};

}  // namespace

// a big per-read array, allocated WITHOUT a zero-filling pass: the loader's worker threads write every
// element, so the first touch (page faults included) happens in parallel
template <class T>
struct RawBuf {
    std::unique_ptr<T[]> p;
    size_t n = 0;
    void resize(size_t k) { p.reset(new T[k]); n = k; }
    T *data() const { return p.get(); }
    size_t size() const { return n; }
    T &operator[](size_t i) const { return p[i]; }
};

struct m6a_sites {
    int n_rep = 1;
    RawBuf<float> X;
    std::vector<uint8_t> site_kmers;
    std::vector<int64_t> off, tx_pos;
    RawBuf<double> read_ids;
    RawBuf<int32_t> read_rep;
    std::vector<std::string> tx_ids, kmer5;
    // what the accessors and the writers read: the owned buffers above after m6a_io_load_sites, or the file
    // mapping of a binary site store (m6a_io_open_store) -- zero-copy, the kernel pages it in on first touch
    const float *vX = nullptr;
    const uint8_t *vK = nullptr;
    const int64_t *vOff = nullptr, *vPos = nullptr;
    const double *vIds = nullptr;
    const int32_t *vRep = nullptr;
    int64_t nS = 0, nR = 0;
    void *map = nullptr;
    size_t map_len = 0;
    std::string tag;
    // m6a_io_csv_shard_size keeps the text it formatted (up to M6A_IO_CSV_KEEP_MB, default 1024) so that the
    // m6a_io_csv_shard_write that follows for the same range and the same arrays writes it instead of formatting again
    struct CsvKeep {
        int64_t a = -1, b = -1;
        const void *rp = nullptr, *sp = nullptr, *mr = nullptr;
        // the kept text belongs to VALUES, not to addresses: a caller's temporary arrays can be freed after the size call and
        // other values allocated at the same addresses before the write -- a checksum of what was formatted is compared too
        uint64_t sum = 0;
        std::vector<std::string> site, indiv;
        void clear() { a = b = -1; sum = 0; std::vector<std::string>().swap(site); std::vector<std::string>().swap(indiv); }
    } csv_keep;
    void view_owned()
    {
        vX = X.data(); vK = site_kmers.data(); vOff = off.data(); vPos = tx_pos.data();
        vIds = read_ids.data(); vRep = read_rep.data();
        nS = (int64_t)tx_pos.size(); nR = off.empty() ? 0 : off.back();
    }
    ~m6a_sites() { if (map) munmap(map, map_len); }
};

namespace {

// parses one record into rows of `ncol` numbers appended to `vals`; returns the sequence key
int parse_record(const char *p, const char *e, const std::string &tx, int64_t pos, std::string &kmer,
                 std::vector<double> &vals, int &ncol)
{
    Cursor c{p, e};
    std::string key;
    if (!c.eat('{') || !c.str(key) || !c.eat(':')) return fail(M6A_IO_EFORMAT, "record of %s:%lld: bad JSON", tx.c_str(), (long long)pos);
    if (key != tx) return fail(M6A_IO_EFORMAT, "record at offset of %s:%lld is for transcript %s", tx.c_str(), (long long)pos, key.c_str());
    if (!c.eat('{') || !c.str(key) || !c.eat(':')) return fail(M6A_IO_EFORMAT, "record of %s:%lld: bad JSON", tx.c_str(), (long long)pos);
    if (key != std::to_string(pos)) return fail(M6A_IO_EFORMAT, "record of %s:%lld holds position %s", tx.c_str(), (long long)pos, key.c_str());
    if (!c.eat('{') || !c.str(kmer) || !c.eat(':') || !c.eat('['))
        return fail(M6A_IO_EFORMAT, "record of %s:%lld: bad JSON", tx.c_str(), (long long)pos);
    ncol = -1;
    if (!c.eat(']')) {
        do {
            if (!c.eat('[')) return fail(M6A_IO_EFORMAT, "record of %s:%lld: expected a row", tx.c_str(), (long long)pos);
            int n = 0;
            do {
                double v;
                if (!c.num(v)) return fail(M6A_IO_EFORMAT, "record of %s:%lld: bad number", tx.c_str(), (long long)pos);
                vals.push_back(v);
                ++n;
            } while (c.eat(','));
            if (!c.eat(']')) return fail(M6A_IO_EFORMAT, "record of %s:%lld: unterminated row", tx.c_str(), (long long)pos);
            if (ncol < 0) ncol = n;
            else if (ncol != n) return fail(M6A_IO_EFORMAT, "record of %s:%lld: ragged rows", tx.c_str(), (long long)pos);
        } while (c.eat(','));
        if (!c.eat(']')) return fail(M6A_IO_EFORMAT, "record of %s:%lld: unterminated array", tx.c_str(), (long long)pos);
    }
    if (c.eat(',')) return fail(M6A_IO_EFORMAT, "site %s:%lld has more than one sequence key", tx.c_str(), (long long)pos);
    if (!c.eat('}') || !c.eat('}') || !c.eat('}')) return fail(M6A_IO_EFORMAT, "record of %s:%lld: bad JSON tail", tx.c_str(), (long long)pos);
    return 0;
}

// str(numpy.float64): shortest repr that round-trips; read indices are integral, so "<int>.0"
void format_py_float(double v, std::string &out)
{
    char buf[40];
    if (std::isfinite(v) && v == std::floor(v) && std::fabs(v) < 1e16) {
        snprintf(buf, sizeof buf, "%.1f", v);
        out += buf;
        return;
    }
    for (int prec = 1; prec <= 17; prec++) {
        snprintf(buf, sizeof buf, "%.*g", prec, v);
        if (strtod(buf, nullptr) == v) break;
    }
    out += buf;
    if (!strpbrk(buf, ".en")) out += ".0";
}

// ---- "%.16f" without printf ---------------------------------------------------------------------------
// The CSV rows print probabilities with '%.16f' (inference_utils.py:62,66): glibc formats a double exactly
// (multi-precision) and costs ~1 us per number; for 0 <= v < 2 the same digits come from one 128-bit product:
// v = m * 2^e exactly (m < 2^53), so round_half_even(v * 10^16) = (m * 10^16) >> -e with the dropped bits deciding
// the rounding -- m * 10^16 < 2^107 fits an unsigned __int128.  Anything else (v >= 2, negative, nan, inf) goes
// through snprintf.  Returns the number of characters written (no terminator); buf holds >= 336 (the longest
// "%.16f" of a double: sign, 309 digits, point, 16 digits, NUL).
const char kDigitPairs[201] =
    "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960"
    "616263646566676869707172737475767778798081828384858687888990919293949596979899";

inline void put8(char *o, uint32_t v)       // exactly 8 digits, zero padded
{
    for (int i = 3; i >= 0; i--) { const uint32_t q = v / 100, r = v - q * 100; memcpy(o + 2 * i, kDigitPairs + 2 * r, 2); v = q; }
}

int format_f16(double v, char *buf)
{
    uint64_t bits;
    memcpy(&bits, &v, 8);
    const int be = (int)(bits >> 52) & 0x7ff;
    if ((bits >> 63) || be >= 1024) return snprintf(buf, 336, "%.16f", v);      // negative (incl. -0.0), >= 2, nan, inf
    uint64_t m = bits & ((1ull << 52) - 1);
    int e;                                                  // v = m * 2^e
    if (be == 0) e = -1074; else { m |= 1ull << 52; e = be - 1075; }
    const unsigned __int128 prod = (unsigned __int128)m * 10000000000000000ull;
    const int sh = -e;                                      // >= 52 here (v < 2)
    uint64_t n;                                             // round_half_even(v * 1e16) <= 2e16
    if (sh >= 108) n = 0;                                   // prod < 2^107: below one half (a tie needs prod = 2^(sh-1), and 5^16 | prod)
    else {
        n = (uint64_t)(prod >> sh);
        const unsigned __int128 rem = prod & (((unsigned __int128)1 << sh) - 1), half = (unsigned __int128)1 << (sh - 1);
        if (rem > half || (rem == half && (n & 1))) n++;
    }
    const uint64_t ip = n / 10000000000000000ull, fp = n - ip * 10000000000000000ull;
    buf[0] = (char)('0' + ip);
    buf[1] = '.';
    put8(buf + 2, (uint32_t)(fp / 100000000ull));
    put8(buf + 10, (uint32_t)(fp % 100000000ull));
    return 18;
}

inline int format_i64(long long v, char *buf)
{
    return (int)(std::to_chars(buf, buf + 24, v).ptr - buf);
}

// M6A_IO_TRACE=1: phase times of m6a_io_load_sites on stderr
struct PhaseTrace {
    bool on = getenv("M6A_IO_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void mark(const char *what)
    {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "m6a_io: %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

int n_workers(int n_threads, int64_t items)
{
    int n = n_threads > 0 ? n_threads : m6a_usable_cpus();
    n = std::max(1, std::min<int>(n, 64));
    return (int)std::max<int64_t>(1, std::min<int64_t>(n, items));
}

}  // namespace

extern "C" {

const char *m6a_io_last_error(void) { return g_err.c_str(); }

int m6a_io_load_sites(const char *const *input_dirs, int n_dirs, int min_reads, const char *norm_kmers,
                      const double *norm_mean, const double *norm_std, int n_norm, int n_threads, m6a_sites **out)
{
    if (!out) return M6A_IO_EINVAL;
    *out = nullptr;
    if (!input_dirs || n_dirs < 1) return fail(M6A_IO_EINVAL, "no input directory");
    if (n_norm < 0 || (n_norm > 0 && (!norm_kmers || !norm_mean || !norm_std))) return fail(M6A_IO_EINVAL, "bad normalisation arguments");

    PhaseTrace trace;
    std::vector<std::string> info_text((size_t)n_dirs);       // the data.info files: the SiteRefs point into them
    std::vector<SiteRef> all;
    std::unordered_map<std::string, size_t> index;
    std::vector<Mapped> json((size_t)n_dirs);
    {
        // map (and pre-fault) the data.json files on a thread while the data.info files are parsed here
        int map_rc = 0;
        std::string map_err;
        std::thread mapper([&] {
            for (int r = 0; r < n_dirs && !map_rc; r++) {
                map_rc = json[(size_t)r].open(std::string(input_dirs[r]) + "/data.json");
                if (map_rc) map_err = g_err;                       // g_err is thread-local
            }
        });
        int rc = 0;
        for (int r = 0; r < n_dirs && !rc; r++) rc = parse_info(input_dirs[r], r, all, n_dirs > 1 ? &index : nullptr, info_text[(size_t)r]);
        trace.mark("data.info parsed");
        mapper.join();
        trace.mark("data.json mapped (wait)");
        if (rc) return rc;
        if (map_rc) { g_err = map_err; return map_rc; }
    }
    std::vector<SiteRef> sites;
    for (auto &s : all) if (s.n_reads >= min_reads) sites.push_back(std::move(s));
    if (sites.empty()) return fail(M6A_IO_EFORMAT, "no site with at least %d reads", min_reads);
    const int64_t S = (int64_t)sites.size();

    std::unordered_map<std::string, int> norm_ix;
    for (int i = 0; i < n_norm; i++) norm_ix.emplace(std::string(norm_kmers + 5 * i, 5), i);

    m6a_sites *res = new (std::nothrow) m6a_sites;
    if (!res) return fail(M6A_IO_ENOMEM, "out of memory");
    res->n_rep = n_dirs;
    res->off.assign((size_t)S + 1, 0);
    for (int64_t s = 0; s < S; s++) res->off[(size_t)s + 1] = res->off[(size_t)s] + sites[(size_t)s].n_reads;
    const int64_t R = res->off[(size_t)S];
    try {
        res->X.resize((size_t)R * 9);
        res->site_kmers.resize((size_t)S * 3);
        res->tx_pos.resize((size_t)S);
        res->read_ids.resize((size_t)R);
        res->read_rep.resize((size_t)R);
        res->tx_ids.resize((size_t)S);
        res->kmer5.resize((size_t)S);
    } catch (const std::bad_alloc &) {
        delete res;
        return fail(M6A_IO_ENOMEM, "out of memory for %lld reads", (long long)R);
    }

    trace.mark("filter, offsets, buffers");
    const int nw = n_workers(n_threads, S);
    std::vector<std::string> errs((size_t)nw);
    std::vector<int> rcs((size_t)nw, 0);
    auto work = [&](int w) {
        std::vector<double> vals;
        // contiguous site ranges balanced by read count
        const int64_t r_lo = R * w / nw, r_hi = R * (w + 1) / nw;
        int64_t s0 = std::lower_bound(res->off.begin(), res->off.end() - 1, r_lo) - res->off.begin();
        int64_t s1 = (w == nw - 1) ? S : std::lower_bound(res->off.begin(), res->off.end() - 1, r_hi) - res->off.begin();
        for (int64_t s = s0; s < s1; s++) {
            const SiteRef &sr = sites[(size_t)s];
            std::string tx(sr.tx);
            std::string kmer;
            int64_t row = res->off[(size_t)s];
            for (size_t ip = 0; ip < sr.n_parts(); ip++) {
                const Part &pt = sr.parts_begin()[ip];
                const Mapped &m = json[(size_t)pt.rep];
                if (pt.start < 0 || pt.end > (int64_t)m.n || pt.start >= pt.end) {
                    rcs[(size_t)w] = fail(M6A_IO_EFORMAT, "site %s:%lld: byte range outside data.json", tx.c_str(), (long long)sr.pos);
                    errs[(size_t)w] = g_err; return;
                }
                vals.clear();
                std::string k;
                int ncol = 0;
                int rc = parse_record(m.p + pt.start, m.p + pt.end, tx, sr.pos, k, vals, ncol);
                if (rc) { rcs[(size_t)w] = rc; errs[(size_t)w] = g_err; return; }
                if (kmer.empty()) kmer = k;
                else if (kmer != k) { rcs[(size_t)w] = fail(M6A_IO_EFORMAT, "replicates disagree on the sequence of %s:%lld", tx.c_str(), (long long)sr.pos); errs[(size_t)w] = g_err; return; }
                if (kmer.size() != 7 || ncol != 10) {
                    // data prepared with n_neighbors != 1 (the reference's own slice for that case,
                    // data_utils.py:276-277, yields a 4-mer and fails too)
                    rcs[(size_t)w] = fail(M6A_IO_EFORMAT, "site %s:%lld: %zu-mer with %d columns; only dataprep n_neighbors=1 (7-mer, 10 columns) is supported",
                                          tx.c_str(), (long long)sr.pos, kmer.size(), ncol);
                    errs[(size_t)w] = g_err; return;
                }
                const int64_t nrow = (int64_t)vals.size() / 10;
                if (row + nrow > res->off[(size_t)s + 1]) { rcs[(size_t)w] = fail(M6A_IO_EFORMAT, "site %s:%lld has more reads than data.info says", tx.c_str(), (long long)sr.pos); errs[(size_t)w] = g_err; return; }
                double mean[9], sd[9];
                for (int c = 0; c < 3; c++) {
                    const std::string k5 = kmer.substr((size_t)c, 5);
                    if (n_norm) {
                        auto it = norm_ix.find(k5);
                        if (it == norm_ix.end()) { rcs[(size_t)w] = fail(M6A_IO_EFORMAT, "no normalisation factors for %s", k5.c_str()); errs[(size_t)w] = g_err; return; }
                        for (int j = 0; j < 3; j++) { mean[3 * c + j] = norm_mean[3 * it->second + j]; sd[3 * c + j] = norm_std[3 * it->second + j]; }
                    }
                }
                for (int64_t i = 0; i < nrow; i++) {
                    const double *v = vals.data() + 10 * i;
                    float *x = res->X.data() + 9 * (row + i);
                    for (int j = 0; j < 9; j++) x[j] = n_norm ? (float)((v[j] - mean[j]) / sd[j]) : (float)v[j];
                    res->read_ids[(size_t)(row + i)] = v[9];
                    res->read_rep[(size_t)(row + i)] = pt.rep;
                }
                row += nrow;
            }
            if (row != res->off[(size_t)s + 1]) { rcs[(size_t)w] = fail(M6A_IO_EFORMAT, "site %s:%lld: data.info says %lld reads, data.json has %lld", tx.c_str(), (long long)sr.pos, (long long)sr.n_reads, (long long)(row - res->off[(size_t)s])); errs[(size_t)w] = g_err; return; }
            for (int c = 0; c < 3; c++) {
                auto it = vocab().find(kmer.substr((size_t)c, 5));
                if (it == vocab().end()) { rcs[(size_t)w] = fail(M6A_IO_EFORMAT, "site %s:%lld: %s is not a DRACH context", tx.c_str(), (long long)sr.pos, kmer.c_str()); errs[(size_t)w] = g_err; return; }
                res->site_kmers[(size_t)(3 * s + c)] = (uint8_t)it->second;
            }
            res->tx_pos[(size_t)s] = sr.pos;
            res->tx_ids[(size_t)s] = std::move(tx);
            res->kmer5[(size_t)s] = kmer.substr(1, 5);
        }
    };
    std::vector<std::thread> th;
    for (int w = 1; w < nw; w++) th.emplace_back(work, w);
    work(0);
    for (auto &t : th) t.join();
    trace.mark("records parsed (workers)");
    for (int w = 0; w < nw; w++)
        if (rcs[(size_t)w]) { g_err = errs[(size_t)w]; delete res; return rcs[(size_t)w]; }
    res->view_owned();
    *out = res;
    return M6A_IO_OK;
}

void m6a_io_free(m6a_sites *s) { delete s; }
int64_t m6a_io_n_sites(const m6a_sites *s) { return s ? s->nS : 0; }
int64_t m6a_io_n_reads(const m6a_sites *s) { return s ? s->nR : 0; }
int m6a_io_n_replicates(const m6a_sites *s) { return s ? s->n_rep : 0; }
const float *m6a_io_X(const m6a_sites *s) { return s->vX; }
const uint8_t *m6a_io_site_kmers(const m6a_sites *s) { return s->vK; }
const int64_t *m6a_io_off(const m6a_sites *s) { return s->vOff; }
const int64_t *m6a_io_tx_pos(const m6a_sites *s) { return s->vPos; }
const double *m6a_io_read_ids(const m6a_sites *s) { return s->vIds; }
const int32_t *m6a_io_read_rep(const m6a_sites *s) { return s->vRep; }
const char *m6a_io_store_tag(const m6a_sites *s) { return s ? s->tag.c_str() : ""; }

// ---- binary site store ------------------------------------------------------------------------------------
// One file, little-endian, every array 64-byte aligned:
//   header (128 B) | off i64[S+1] | tx_pos i64[S] | site_kmers u8[S][3] | kmer5 char[S][5] | tx_off i64[S+1] |
//   tx bytes | read_ids f64[R] | read_rep i32[R] | X f32[R][9]
namespace {
struct StoreHeader {
    char magic[8];               // "M6ASITES"
    uint32_t version, n_rep;
    int64_t S, R, tx_bytes;
    char tag[64];                // what the features were normalised with (the caller's label, e.g. "norm_hct116.npz min_reads=20")
    char pad[24];
};
static_assert(sizeof(StoreHeader) == 128, "store header is 128 bytes");
size_t align64(size_t x) { return (x + 63) & ~(size_t)63; }
struct StoreLayout { size_t off, pos, km, k5, txo, txb, ids, rep, x, end; };
StoreLayout store_layout(int64_t S, int64_t R, int64_t tx_bytes)
{
    StoreLayout l;
    size_t p = sizeof(StoreHeader);
    l.off = p; p = align64(p + (size_t)(S + 1) * 8);
    l.pos = p; p = align64(p + (size_t)S * 8);
    l.km = p; p = align64(p + (size_t)S * 3);
    l.k5 = p; p = align64(p + (size_t)S * 5);
    l.txo = p; p = align64(p + (size_t)(S + 1) * 8);
    l.txb = p; p = align64(p + (size_t)tx_bytes);
    l.ids = p; p = align64(p + (size_t)R * 8);
    l.rep = p; p = align64(p + (size_t)R * 4);
    l.x = p; p = p + (size_t)R * 9 * 4;
    l.end = p;
    return l;
}
}  // namespace

int m6a_io_save_store(const m6a_sites *s, const char *path, const char *tag)
{
    if (!s || !path) return fail(M6A_IO_EINVAL, "null argument");
    const int64_t S = s->nS, R = s->nR;
    std::vector<int64_t> txo((size_t)S + 1, 0);
    for (int64_t i = 0; i < S; i++) txo[(size_t)i + 1] = txo[(size_t)i] + (int64_t)s->tx_ids[(size_t)i].size();
    StoreHeader h;
    std::memset(&h, 0, sizeof h);
    std::memcpy(h.magic, "M6ASITES", 8);
    h.version = 1; h.n_rep = (uint32_t)s->n_rep; h.S = S; h.R = R; h.tx_bytes = txo[(size_t)S];
    if (tag) std::strncpy(h.tag, tag, sizeof h.tag - 1);
    const StoreLayout l = store_layout(S, R, h.tx_bytes);
    const std::string tmp = std::string(path) + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return fail(M6A_IO_EIO, "cannot write %s", tmp.c_str());
    bool ok = true;
    size_t at = 0;
    auto put = [&](size_t where, const void *p, size_t n) {
        static const char zeros[64] = {0};
        while (ok && at < where) { const size_t k = std::min<size_t>(64, where - at); ok = fwrite(zeros, 1, k, f) == k; at += k; }
        if (ok && n) { ok = fwrite(p, 1, n, f) == n; at += n; }
    };
    std::string k5((size_t)S * 5, ' '), txb;
    txb.reserve((size_t)h.tx_bytes);
    for (int64_t i = 0; i < S; i++) {
        std::memcpy(&k5[(size_t)i * 5], s->kmer5[(size_t)i].data(), std::min<size_t>(5, s->kmer5[(size_t)i].size()));
        txb += s->tx_ids[(size_t)i];
    }
    put(0, &h, sizeof h);
    put(l.off, s->vOff, (size_t)(S + 1) * 8);
    put(l.pos, s->vPos, (size_t)S * 8);
    put(l.km, s->vK, (size_t)S * 3);
    put(l.k5, k5.data(), k5.size());
    put(l.txo, txo.data(), txo.size() * 8);
    put(l.txb, txb.data(), txb.size());
    put(l.ids, s->vIds, (size_t)R * 8);
    put(l.rep, s->vRep, (size_t)R * 4);
    put(l.x, s->vX, (size_t)R * 9 * 4);
    if (fclose(f) != 0) ok = false;
    if (!ok || rename(tmp.c_str(), path) != 0) { remove(tmp.c_str()); return fail(M6A_IO_EIO, "cannot write %s", path); }
    return M6A_IO_OK;
}

int m6a_io_open_store(const char *path, m6a_sites **out)
{
    if (!path || !out) return fail(M6A_IO_EINVAL, "null argument");
    *out = nullptr;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(M6A_IO_EIO, "cannot open %s", path);
    struct stat st;
    if (fstat(fd, &st) != 0 || (size_t)st.st_size < sizeof(StoreHeader)) { close(fd); return fail(M6A_IO_EFORMAT, "%s is not a site store", path); }
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return fail(M6A_IO_EIO, "cannot map %s", path);
    const char *b = (const char *)m;
    StoreHeader h;
    std::memcpy(&h, b, sizeof h);
    const bool sane = std::memcmp(h.magic, "M6ASITES", 8) == 0 && h.version == 1 && h.S >= 0 && h.R >= 0 && h.tx_bytes >= 0 &&
                      h.S < ((int64_t)1 << 40) && h.R < ((int64_t)1 << 44) && h.tx_bytes < ((int64_t)1 << 40);
    StoreLayout l{};
    if (sane) l = store_layout(h.S, h.R, h.tx_bytes);
    if (!sane || l.end != (size_t)st.st_size) { munmap(m, (size_t)st.st_size); return fail(M6A_IO_EFORMAT, "%s: bad header or truncated store", path); }
    m6a_sites *s = new (std::nothrow) m6a_sites;
    if (!s) { munmap(m, (size_t)st.st_size); return fail(M6A_IO_ENOMEM, "out of memory"); }
    s->map = m; s->map_len = (size_t)st.st_size;
    s->n_rep = (int)h.n_rep; s->nS = h.S; s->nR = h.R;
    h.tag[sizeof h.tag - 1] = 0;
    s->tag = h.tag;
    s->vOff = (const int64_t *)(b + l.off); s->vPos = (const int64_t *)(b + l.pos); s->vK = (const uint8_t *)(b + l.km);
    s->vIds = (const double *)(b + l.ids); s->vRep = (const int32_t *)(b + l.rep); s->vX = (const float *)(b + l.x);
    const int64_t *txo = (const int64_t *)(b + l.txo);
    bool good = s->vOff[0] == 0 && s->vOff[h.S] == h.R && txo[0] == 0 && txo[h.S] == h.tx_bytes;
    for (int64_t i = 0; good && i < h.S; i++) good = s->vOff[i + 1] >= s->vOff[i] && txo[i + 1] >= txo[i];
    if (!good) { delete s; return fail(M6A_IO_EFORMAT, "%s: inconsistent offsets", path); }
    // k-mer ids index the encoder's 66-row embedding table on the GPU: a corrupt byte must not get that far
    for (int64_t i = 0; good && i < 3 * h.S; i++) good = s->vK[i] < 66;
    if (!good) { delete s; return fail(M6A_IO_EFORMAT, "%s: k-mer id out of range (vocabulary has 66 entries)", path); }
    s->tx_ids.resize((size_t)h.S); s->kmer5.resize((size_t)h.S);
    for (int64_t i = 0; i < h.S; i++) {
        s->tx_ids[(size_t)i].assign(b + l.txb + txo[i], (size_t)(txo[i + 1] - txo[i]));
        s->kmer5[(size_t)i].assign(b + l.k5 + i * 5, 5);
    }
    *out = s;
    return M6A_IO_OK;
}
const char *m6a_io_tx_id(const m6a_sites *s, int64_t i) { return s->tx_ids[(size_t)i].c_str(); }
const char *m6a_io_kmer5(const m6a_sites *s, int64_t i) { return s->kmer5[(size_t)i].c_str(); }

int m6a_io_format_f16(double v, char *buf336) { const int k = format_f16(v, buf336); buf336[k] = 0; return k; }

int m6a_io_write_csv(const m6a_sites *s, const char *out_dir, const float *read_prob, const float *site_prob,
                     const double *mod_ratio, int write_header, int n_threads)
{
    return m6a_io_write_csv_n(s, out_dir, read_prob, site_prob, mod_ratio, write_header, n_threads, -1);
}

}  // extern "C"

namespace {

template <class F>
void on_threads_io(int nw, int n_items, F &&f)
{
    std::atomic<int> next{0};
    auto worker = [&]() { for (int k; (k = next.fetch_add(1)) < n_items;) f(k); };
    std::vector<std::thread> th;
    for (int t = 1; t < std::min(nw, n_items); t++) th.emplace_back(worker);
    worker();
    for (auto &t : th) t.join();
}

// The rows of sites [i0, i1) appended to `a` (data.site_proba.csv) and `b` (data.indiv_proba.csv).  read_prob / site_prob /
// mod_ratio hold the values of sites [site_base, ...) and reads [read_base, ...): the whole job (bases 0) or one rank's shard.
void format_rows(const m6a_sites *s, int64_t i0, int64_t i1, const float *read_prob, const float *site_prob, const double *mod_ratio,
                 int64_t site_base, int64_t read_base, std::string &a, std::string &b)
{
    char buf[512];
    std::string head;
    for (int64_t i = i0; i < i1; i++) {
        const int64_t r0 = s->vOff[i], r1 = s->vOff[i + 1];
        // '%s,%d,%s,%.16f,%s,%.16f'  (inference_utils.py:62)
        a += s->tx_ids[(size_t)i];
        int k = 0;
        buf[k++] = ',';
        k += format_i64((long long)s->vPos[i], buf + k);
        buf[k++] = ',';
        k += format_i64((long long)(r1 - r0), buf + k);
        buf[k++] = ',';
        k += format_f16((double)site_prob[i - site_base], buf + k);
        buf[k++] = ',';
        a.append(buf, (size_t)k);
        a += s->kmer5[(size_t)i];
        k = 0;
        buf[k++] = ',';
        k += format_f16(mod_ratio[i - site_base], buf + k);
        buf[k++] = '\n';
        a.append(buf, (size_t)k);
        // '%s,%d,%s,%.16f'  (inference_utils.py:66); read ids: str(float64), or "<int>_<rep>"
        head.assign(s->tx_ids[(size_t)i]);
        k = 0;
        buf[k++] = ',';
        k += format_i64((long long)s->vPos[i], buf + k);
        buf[k++] = ',';
        head.append(buf, (size_t)k);
        for (int64_t r = r0; r < r1; r++) {
            b += head;
            k = 0;
            const double id = s->vIds[r];
            if (s->n_rep > 1) {
                k += format_i64((long long)id, buf + k);
                buf[k++] = '_';
                k += format_i64((long long)s->vRep[r], buf + k);
            } else if (id == std::floor(id) && std::fabs(id) < 1e15 && !std::signbit(id)) {
                k += format_i64((long long)id, buf + k);                     // str(float64) of an integral value
                buf[k++] = '.';
                buf[k++] = '0';
            } else {
                format_py_float(id, b);
            }
            buf[k++] = ',';
            k += format_f16((double)read_prob[r - read_base], buf + k);
            buf[k++] = '\n';
            b.append(buf, (size_t)k);
        }
    }
}

const char kSiteHeader[] = "transcript_id,transcript_position,n_reads,probability_modified,kmer,mod_ratio\n";
const char kIndivHeader[] = "transcript_id,transcript_position,read_index,probability_modified\n";

// Sites [A, B) formatted on all threads a round of chunks at a time (chunks of at most 2^20 reads bound the text held in
// memory, ~64 MB per worker; at least 2^14 so tiny jobs do not spawn idle threads); `sink(site_text, indiv_text)` gets the
// chunks in site order.
template <class RoundSink>
int format_site_rounds(const m6a_sites *s, int64_t A, int64_t B, const float *read_prob, const float *site_prob, const double *mod_ratio,
                       int64_t site_base, int64_t read_base, int n_threads, RoundSink &&round_sink)
{
    const int nw = n_workers(n_threads, B - A);
    const int64_t R = s->vOff[B] - s->vOff[A];
    const int64_t chunk_reads = std::max<int64_t>(1 << 14, std::min<int64_t>(1 << 20, (R + nw - 1) / nw));
    int64_t s_begin = A;
    while (s_begin < B) {
        std::vector<int64_t> cuts{s_begin};
        for (int w = 0; w < nw && cuts.back() < B; w++) {
            const int64_t target = std::min(s->vOff[B], s->vOff[cuts.back()] + chunk_reads);
            int64_t e = std::upper_bound(s->vOff, s->vOff + s->nS + 1, target) - s->vOff - 1;
            e = std::min<int64_t>(B, std::max<int64_t>(e, cuts.back() + 1));
            cuts.push_back(e);
        }
        const int nc = (int)cuts.size() - 1;
        std::vector<std::string> site_txt((size_t)nc), indiv_txt((size_t)nc);
        auto work = [&](int w) {
            const int64_t i0 = cuts[(size_t)w], i1 = cuts[(size_t)w + 1];
            site_txt[(size_t)w].reserve((size_t)(i1 - i0) * 96);
            indiv_txt[(size_t)w].reserve((size_t)(s->vOff[i1] - s->vOff[i0]) * 64);
            format_rows(s, i0, i1, read_prob, site_prob, mod_ratio, site_base, read_base, site_txt[(size_t)w], indiv_txt[(size_t)w]);
        };
        std::vector<std::thread> th;
        for (int w = 1; w < nc; w++) th.emplace_back(work, w);
        work(0);
        for (auto &t : th) t.join();
        const int rc = round_sink(site_txt, indiv_txt, nw);                        // the chunks of one round, in site order
        if (rc) return rc;
        s_begin = cuts.back();
    }
    return 0;
}

// ... with a sink per chunk, called in site order (it may move the strings out)
template <class Sink>
int format_site_range(const m6a_sites *s, int64_t A, int64_t B, const float *read_prob, const float *site_prob, const double *mod_ratio,
                      int64_t site_base, int64_t read_base, int n_threads, Sink &&sink)
{
    return format_site_rounds(s, A, B, read_prob, site_prob, mod_ratio, site_base, read_base, n_threads,
                              [&](std::vector<std::string> &a, std::vector<std::string> &b, int) {
                                  for (size_t w = 0; w < a.size(); w++) {
                                      const int rc = sink(a[w], b[w]);
                                      if (rc) return rc;
                                  }
                                  return 0;
                              });
}

// every chunk of a round pwrite()n at its offset by its own thread: the page-cache copy of a 436 MB data.indiv_proba.csv
// is worth several threads (one thread's write() was most of the writer's time)
bool pwrite_all(int fd, const char *p, size_t n, int64_t at)
{
    while (n) {
        const ssize_t w = ::pwrite(fd, p, n, (off_t)at);
        if (w < 0) { if (errno == EINTR) continue; return false; }
        p += w; n -= (size_t)w; at += w;
    }
    return true;
}

int pwrite_round(int fd_site, int fd_indiv, const std::vector<std::string> &a, const std::vector<std::string> &b, int64_t &at_site, int64_t &at_indiv, int nw)
{
    const size_t nc = a.size();
    std::vector<int64_t> oa(nc), ob(nc);
    for (size_t w = 0; w < nc; w++) { oa[w] = at_site; at_site += (int64_t)a[w].size(); ob[w] = at_indiv; at_indiv += (int64_t)b[w].size(); }
    std::atomic<bool> ok{true};
    on_threads_io(std::min<int>(nw, (int)nc), (int)nc, [&](int w) {
        if (!pwrite_all(fd_site, a[(size_t)w].data(), a[(size_t)w].size(), oa[(size_t)w]) ||
            !pwrite_all(fd_indiv, b[(size_t)w].data(), b[(size_t)w].size(), ob[(size_t)w])) ok = false;
    });
    return ok ? 0 : M6A_IO_EIO;
}

int check_range(const m6a_sites *s, int64_t a, int64_t b)
{
    if (!s) return fail(M6A_IO_EINVAL, "null argument");
    if (a < 0 || b < a || b > m6a_io_n_sites(s)) return fail(M6A_IO_EINVAL, "site range [%lld, %lld) outside the job's %lld sites", (long long)a, (long long)b, (long long)m6a_io_n_sites(s));
    return 0;
}

}  // namespace

extern "C" {

int m6a_io_write_csv_n(const m6a_sites *s, const char *out_dir, const float *read_prob, const float *site_prob,
                       const double *mod_ratio, int write_header, int n_threads, int64_t n_sites_limit)
{
    if (!s || !out_dir || !read_prob || !site_prob || !mod_ratio) return fail(M6A_IO_EINVAL, "null argument");
    const int64_t S = n_sites_limit >= 0 ? std::min<int64_t>(n_sites_limit, m6a_io_n_sites(s)) : m6a_io_n_sites(s);
    const std::string fs = std::string(out_dir) + "/data.site_proba.csv", fi = std::string(out_dir) + "/data.indiv_proba.csv";
    const int flags = O_WRONLY | O_CREAT | (write_header ? O_TRUNC : 0);
    const int f = ::open(fs.c_str(), flags, 0644);
    if (f < 0) return fail(M6A_IO_EIO, "cannot open %s", fs.c_str());
    const int g = ::open(fi.c_str(), flags, 0644);
    if (g < 0) { ::close(f); return fail(M6A_IO_EIO, "cannot open %s", fi.c_str()); }
    int64_t at_site = 0, at_indiv = 0;
    bool ok = true;
    if (write_header) {
        ok = pwrite_all(f, kSiteHeader, sizeof(kSiteHeader) - 1, 0) && pwrite_all(g, kIndivHeader, sizeof(kIndivHeader) - 1, 0);
        at_site = (int64_t)sizeof(kSiteHeader) - 1; at_indiv = (int64_t)sizeof(kIndivHeader) - 1;
    } else {                                                   // append: behind whatever the files hold
        struct stat st;
        ok = fstat(f, &st) == 0;
        at_site = ok ? (int64_t)st.st_size : 0;
        ok = ok && fstat(g, &st) == 0;
        at_indiv = ok ? (int64_t)st.st_size : 0;
    }
    int rc = ok ? 0 : fail(M6A_IO_EIO, "cannot write into %s", out_dir);
    // rows are formatted on all threads a round of chunks at a time; every chunk is written at its offset by its own thread
    if (!rc)
        rc = format_site_rounds(s, 0, S, read_prob, site_prob, mod_ratio, 0, 0, n_threads,
                                [&](std::vector<std::string> &a, std::vector<std::string> &b, int nw) {
                                    if (pwrite_round(f, g, a, b, at_site, at_indiv, nw)) return fail(M6A_IO_EIO, "short write in %s", out_dir);
                                    return 0;
                                });
    if (::close(f) != 0 && !rc) rc = fail(M6A_IO_EIO, "cannot close %s", fs.c_str());
    if (::close(g) != 0 && !rc) rc = fail(M6A_IO_EIO, "cannot close %s", fi.c_str());
    return rc;
}

// Checksum of the arrays a shard's rows are formatted from: every site value, and the read probabilities sampled at a stride
// that keeps the pass under ~1 M loads (plus both ends) -- a guard against stale text, not a hash of the job.
static uint64_t csv_args_sum(const float *rp, int64_t nr, const float *sp, const double *mr, int64_t ns)
{
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)nr ^ ((uint64_t)ns << 32);
    auto mix = [&](uint64_t v) { h = (h ^ v) * 0x100000001B3ull; h ^= h >> 29; };
    for (int64_t i = 0; i < ns; i++) { uint32_t u; uint64_t w; memcpy(&u, sp + i, 4); memcpy(&w, mr + i, 8); mix(u); mix(w); }
    const int64_t step = std::max<int64_t>(1, nr >> 20);
    for (int64_t i = 0; i < nr; i += step) { uint32_t u; memcpy(&u, rp + i, 4); mix(u); }
    for (int64_t i = std::max<int64_t>(0, nr - 64); i < nr; i++) { uint32_t u; memcpy(&u, rp + i, 4); mix(u); }
    return h;
}

int64_t m6a_io_csv_header_bytes(int which) { return which == 0 ? (int64_t)sizeof(kSiteHeader) - 1 : (int64_t)sizeof(kIndivHeader) - 1; }

int m6a_io_csv_shard_size(const m6a_sites *s, const float *read_prob, const float *site_prob, const double *mod_ratio,
                          int64_t site_begin, int64_t site_end, int n_threads, int64_t *site_bytes, int64_t *indiv_bytes)
{
    int rc = check_range(s, site_begin, site_end);
    if (rc) return rc;
    if (!site_bytes || !indiv_bytes || (site_end > site_begin && (!read_prob || !site_prob || !mod_ratio))) return fail(M6A_IO_EINVAL, "null argument");
    int64_t na = 0, nb = 0;
    m6a_sites::CsvKeep &keep = const_cast<m6a_sites *>(s)->csv_keep;     // a cache, not part of the sites' value
    keep.clear();
    const char *km = getenv("M6A_IO_CSV_KEEP_MB");
    const int64_t budget = (int64_t)(km ? atoll(km) : 1024) << 20;
    bool keeping = budget > 0;
    rc = format_site_range(s, site_begin, site_end, read_prob, site_prob, mod_ratio, site_begin, s->vOff[site_begin], n_threads,
                           [&](std::string &a, std::string &b) {
                               na += (int64_t)a.size(); nb += (int64_t)b.size();
                               if (keeping && na + nb > budget) { keeping = false; keep.clear(); }
                               if (keeping) { keep.site.push_back(std::move(a)); keep.indiv.push_back(std::move(b)); }
                               return 0;
                           });
    if (!rc && keeping) {
        keep.a = site_begin; keep.b = site_end; keep.rp = read_prob; keep.sp = site_prob; keep.mr = mod_ratio;
        keep.sum = csv_args_sum(read_prob, s->vOff[site_end] - s->vOff[site_begin], site_prob, mod_ratio, site_end - site_begin);
    }
    else keep.clear();
    *site_bytes = na; *indiv_bytes = nb;
    return rc;
}

int m6a_io_csv_shard_write(const m6a_sites *s, const char *out_dir, const float *read_prob, const float *site_prob,
                           const double *mod_ratio, int64_t site_begin, int64_t site_end, int n_threads,
                           int64_t site_offset, int64_t indiv_offset, int write_header, int64_t site_total, int64_t indiv_total)
{
    int rc = check_range(s, site_begin, site_end);
    if (rc) return rc;
    if (!out_dir || site_offset < 0 || indiv_offset < 0 || (site_end > site_begin && (!read_prob || !site_prob || !mod_ratio))) return fail(M6A_IO_EINVAL, "bad argument");
    const std::string fs = std::string(out_dir) + "/data.site_proba.csv", fi = std::string(out_dir) + "/data.indiv_proba.csv";
    const int f = ::open(fs.c_str(), O_WRONLY | O_CREAT, 0644);
    if (f < 0) return fail(M6A_IO_EIO, "cannot open %s", fs.c_str());
    const int g = ::open(fi.c_str(), O_WRONLY | O_CREAT, 0644);
    if (g < 0) { ::close(f); return fail(M6A_IO_EIO, "cannot open %s", fi.c_str()); }
    auto put = [&](int fd, const char *p, size_t n, int64_t at) {
        while (n) {
            const ssize_t w = ::pwrite(fd, p, n, (off_t)at);
            if (w < 0) { if (errno == EINTR) continue; return false; }
            p += w; n -= (size_t)w; at += w;
        }
        return true;
    };
    bool ok = true;
    if (write_header) {
        // the rank that writes the headers also gives the files their final size: whatever an earlier run left behind them is cut
        ok = put(f, kSiteHeader, sizeof(kSiteHeader) - 1, 0) && put(g, kIndivHeader, sizeof(kIndivHeader) - 1, 0);
        if (ok && site_total >= 0) ok = ftruncate(f, (off_t)site_total) == 0;
        if (ok && indiv_total >= 0) ok = ftruncate(g, (off_t)indiv_total) == 0;
    }
    int64_t sa = site_offset, sb = indiv_offset;
    m6a_sites::CsvKeep &keep = const_cast<m6a_sites *>(s)->csv_keep;
    if (ok && keep.a == site_begin && keep.b == site_end && keep.rp == read_prob && keep.sp == site_prob && keep.mr == mod_ratio &&
        keep.sum == csv_args_sum(read_prob, s->vOff[site_end] - s->vOff[site_begin], site_prob, mod_ratio, site_end - site_begin)) {
        // the text m6a_io_csv_shard_size formatted a moment ago
        ok = pwrite_round(f, g, keep.site, keep.indiv, sa, sb, n_workers(n_threads, (int64_t)keep.site.size())) == 0;
        keep.clear();
    } else if (ok)
        rc = format_site_rounds(s, site_begin, site_end, read_prob, site_prob, mod_ratio, site_begin, s->vOff[site_begin], n_threads,
                                [&](std::vector<std::string> &a, std::vector<std::string> &b, int nw) {
                                    if (pwrite_round(f, g, a, b, sa, sb, nw)) return fail(M6A_IO_EIO, "short write in %s", out_dir);
                                    return 0;
                                });
    if (::close(f) != 0) ok = false;
    if (::close(g) != 0) ok = false;
    if (!ok && !rc) rc = fail(M6A_IO_EIO, "cannot write into %s", out_dir);
    return rc;
}

}  // extern "C"


// =====================================================================================================
// dataprep: eventalign.txt -> eventalign.index, data.json, data.info, data.log
// (m6anet/scripts/dataprep.py:54-70 -> m6anet/utils/dataprep_utils.py)
// =====================================================================================================
namespace {

struct IdxRow { std::string tx; long long read; int64_t start, end; };

// repr(float): shortest digits that round-trip; fixed notation for 1e-4 <= |x| < 1e16, else exponent
void py_repr(double v, std::string &out)
{
    if (v == 0) { out += std::signbit(v) ? "-0.0" : "0.0"; return; }
    if (!std::isfinite(v)) { out += std::isnan(v) ? "NaN" : (v < 0 ? "-Infinity" : "Infinity"); return; }
    char buf[64];
    const double av = std::fabs(v);
    if (av >= 1e-4 && av < 1e16) {
        // repr's fixed-notation range: the shortest digits that round-trip, written positionally -- exactly what
        // std::to_chars(fixed) without a precision produces; an integral value gets its ".0"
        auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::fixed);
        if (std::memchr(buf, '.', (size_t)(r.ptr - buf)) == nullptr) { *r.ptr++ = '.'; *r.ptr++ = '0'; }
        out.append(buf, (size_t)(r.ptr - buf));
        return;
    }
    auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);   // shortest, d.ddde+XX
    std::string sci(buf, r.ptr);
    size_t epos = sci.find('e');
    std::string mant = sci.substr(0, epos);
    const int exp10 = std::stoi(sci.substr(epos + 1));
    bool neg = false;
    if (mant[0] == '-') { neg = true; mant.erase(0, 1); }
    std::string digits;
    for (char c : mant) if (c != '.') digits += c;
    if (neg) out += '-';
    if (exp10 >= -4 && exp10 < 16) {
        if (exp10 < 0) {
            out += "0.";
            out.append((size_t)(-exp10 - 1), '0');
            out += digits;
        } else if ((int)digits.size() <= exp10 + 1) {
            out += digits;
            out.append((size_t)(exp10 + 1 - (int)digits.size()), '0');
            out += ".0";
        } else {
            out.append(digits, 0, (size_t)exp10 + 1);
            out += '.';
            out.append(digits, (size_t)exp10 + 1, std::string::npos);
        }
    } else {
        out += digits[0];
        if (digits.size() > 1) { out += '.'; out.append(digits, 1, std::string::npos); }
        char eb[16];
        snprintf(eb, sizeof eb, "e%c%02d", exp10 < 0 ? '-' : '+', std::abs(exp10));
        out += eb;
    }
}

// np.round(x, d): x * 10^d, rint (half to even), / 10^d
inline double np_round(double x, double scale) { return std::nearbyint(x * scale) / scale; }

// repr() of a value that np_round produced (k / scale, scale = 10 or 1000): its shortest round-trip digits ARE the decimal
// k / scale with trailing zeros dropped -- no shortest-digits search needed (a third of data.json's numbers are means rounded
// to one decimal; with --compress all of them have three).  Only where that is provably what repr gives: 0 < |v| < 1e9 (a
// decimal step of 1/scale is then far above half an ulp, so no shorter string can round to the same double) and v is the
// double nearest to k / scale; everything else takes py_repr.
inline bool repr_rounded(double v, double scale, int digits, std::string &out)
{
    const double av = std::fabs(v);
    if (!(av > 0 && av < 1e9)) return false;
    const long long k = std::llrint(av * scale);
    if (k == 0 || (double)k / scale != av) return false;
    const long long sc = (long long)scale, ip = k / sc;
    long long fr = k % sc;
    char buf[40];
    int n = 0;
    if (v < 0) buf[n++] = '-';
    n += format_i64(ip, buf + n);
    buf[n++] = '.';
    int nd = digits;
    while (nd > 1 && fr % 10 == 0) { fr /= 10; --nd; }
    for (int i = nd - 1; i >= 0; i--) { buf[n + i] = (char)('0' + fr % 10); fr /= 10; }
    n += nd;
    out.append(buf, (size_t)n);
    return true;
}

// pandas groupby sum (pandas/_libs/groupby.pyx group_sum): Kahan-compensated
struct Kahan {
    double sum = 0, comp = 0;
    void add(double v)
    {
        const double y = v - comp, t = sum + y;
        comp = t - sum - y;
        if (comp != comp) comp = 0;
        sum = t;
    }
};

// k-mers are views into the mapped eventalign.txt (alive for the whole call): no allocation per event
struct Pos { long long position; std::string_view kmer; double dwell, sd, mean; };

// the 18 DRACH 5-mers (m6anet/utils/dataprep_utils.py: the centre must be one of them): [AGT][GA]AC[ACT]
inline bool is_drach(std::string_view k)
{
    return k.size() == 5 && (k[0] == 'A' || k[0] == 'G' || k[0] == 'T') && (k[1] == 'G' || k[1] == 'A') && k[2] == 'A' && k[3] == 'C' &&
           (k[4] == 'A' || k[4] == 'C' || k[4] == 'T');
}

// One line: the offsets of its first 15 tab-separated fields and the line's end, in ONE pass over the bytes (round 6; rounds 3-5:
// memchr for the newline, then a byte loop over the line for the tabs -- the parse was 40 % of the transcript pass).  16 bytes at a
// time with SSE2 (baseline x86-64) while a whole vector fits before `safe` (the end of the mapping: a 16-byte load must not cross
// it), byte by byte for the last few.  fe[i] = end of field i; returns the number of fields seen (<= 16) and sets le to the newline
// (or e).  Fields beyond the 16th are not recorded.
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
inline int scan_line(const char *p, const char *e, const char *safe, const char *(&fe)[16], const char *&le)
{
    int n = 0;
    const char *q = p;
#if defined(__SSE2__)
    const __m128i tab = _mm_set1_epi8('\t'), nl = _mm_set1_epi8('\n');
    while (q + 16 <= safe && q < e) {
        const __m128i v = _mm_loadu_si128((const __m128i *)q);
        unsigned mt = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(v, tab)), mn = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(v, nl));
        if (q + 16 > e) {                                  // the run ends inside this vector: ignore what lies beyond it
            const unsigned keep = (1u << (unsigned)(e - q)) - 1u;
            mt &= keep; mn &= keep;
        }
        if (mn) {
            const unsigned first_nl = (unsigned)__builtin_ctz(mn);
            mt &= (1u << first_nl) - 1u;
            while (mt) { if (n < 16) fe[n++] = q + __builtin_ctz(mt); mt &= mt - 1; }
            le = q + first_nl;
            if (n < 16) fe[n++] = le;
            return n;
        }
        while (mt) { if (n < 16) fe[n++] = q + __builtin_ctz(mt); mt &= mt - 1; }
        q += 16;
    }
#else
    (void)safe;
#endif
    for (; q < e; ++q) {
        if (*q == '\n') break;
        if (*q == '\t' && n < 16) fe[n++] = q;
    }
    le = q;                                                // the newline, or e
    if (n < 16) fe[n++] = q;
    return n;
}

// combine() for the lines of ONE (contig, read) run: per position the length-weighted means
// (dataprep_utils.py:269-325).  Returns false on malformed input.  `safe` = end of the mapping.
bool combine_read(const char *p, const char *e, const char *safe, std::vector<Pos> &out)
{
    struct Ev { long long position; std::string_view kmer; double mean, sd, len_s; long long length; };
    // scratch that keeps its capacity from run to run (one pair per worker thread): a run is tens of events, and
    // allocating / freeing two vectors per run was visible in the profile
    static thread_local std::vector<Ev> evs;
    static thread_local std::vector<uint32_t> order;
    evs.clear();
    // position, start_idx, end_idx are integers in every eventalign.txt (pandas reads them as int64): plain digits
    // take the integer path, anything else (a sign, a dot, an exponent) the general number parser
    auto int_field = [](const char *p, const char *e, long long &out) {
        if (p >= e || e - p > 18) return false;
        long long v = 0;
        for (const char *q = p; q < e; ++q) {
            if (*q < '0' || *q > '9') return false;
            v = v * 10 + (*q - '0');
        }
        out = v;
        return true;
    };
    auto any_field = [&](const char *p, const char *e, long long &out) {
        if (int_field(p, e, out)) return true;
        double d;
        Cursor c{p, e};
        if (!c.num(d)) return false;
        out = (long long)d;
        return true;
    };
    // the three float fields: digits [. digits] with at most 15 digit characters is what nanopolish writes ("95.31", "0.00299") --
    // mantissa / 10^frac, both exact doubles, one correctly rounded division (Clinger's fast path, as in Cursor::num, without its
    // per-character case analysis); anything else (a sign, an exponent, more digits, a name) goes to Cursor::num, same result
    auto float_field = [](const char *p, const char *e, double &out) {
        static const double p10[] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};
        const char *q = p;
        uint64_t mant = 0;
        while (q < e && (unsigned)(*q - '0') < 10u) mant = mant * 10 + (uint64_t)(*q++ - '0');
        int nd = (int)(q - p), frac = 0;
        if (q < e && *q == '.') {
            const char *f = ++q;
            while (q < e && (unsigned)(*q - '0') < 10u) mant = mant * 10 + (uint64_t)(*q++ - '0');
            frac = (int)(q - f);
            nd += frac;
        }
        if (q != e || nd == 0 || nd > 15) {
            Cursor c{p, e};
            return c.num(out);
        }
        out = (double)mant / p10[frac];
        return true;
    };
    bool sorted = true;
    while (p < e) {
        const char *fe[16], *le;
        const int nf = scan_line(p, e, safe, fe, le);
        const char *eol = le;
        if (eol > p && eol[-1] == '\r') --eol;
        if (eol > p) {
            if (nf < 15) return false;
            if (fe[nf - 1] > eol) fe[nf - 1] = eol;        // the last field ends before a '\r'
            // field i = [fe[i-1] + 1, fe[i]); field 0 starts at p
            const char *b2 = fe[1] + 1, *b9 = fe[8] + 1;
            // reference_kmer == model_kmer  (dataprep_utils.py:287)
            if ((fe[2] - b2) == (fe[9] - b9) && memcmp(b2, b9, (size_t)(fe[2] - b2)) == 0) {
                Ev ev;
                long long st, ed;
                if (!any_field(fe[0] + 1, fe[1], ev.position) || !float_field(fe[5] + 1, fe[6], ev.mean) || !float_field(fe[6] + 1, fe[7], ev.sd) ||
                    !float_field(fe[7] + 1, fe[8], ev.len_s) ||
                    !any_field(fe[12] + 1, fe[13], st) || !any_field(fe[13] + 1, fe[14], ed)) return false;
                ev.length = ed - st;
                ev.kmer = std::string_view(b2, (size_t)(fe[2] - b2));
                if (!evs.empty()) {
                    const Ev &pv = evs.back();
                    if (pv.position > ev.position || (pv.position == ev.position && pv.kmer > ev.kmer)) sorted = false;
                }
                evs.push_back(ev);
            }
        }
        p = le + 1;
    }
    // groupby(['read_index','contig','position','reference_kmer']): sorted keys, rows in file order.  A run that is already
    // in key order (the usual case: nanopolish writes a read's events by position) needs no sort -- a stable sort keeps it as is.
    order.resize(evs.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = (uint32_t)i;
    if (!sorted)
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b2) {
            if (evs[a].position != evs[b2].position) return evs[a].position < evs[b2].position;
            return evs[a].kmer < evs[b2].kmer;
        });
    size_t i = 0;
    while (i < order.size()) {
        size_t j = i;
        Kahan sm, ss, sd;
        long long total = 0;
        const Ev &first = evs[order[i]];
        while (j < order.size() && evs[order[j]].position == first.position && evs[order[j]].kmer == first.kmer) {
            const Ev &ev = evs[order[j]];
            const double len = (double)ev.length;
            sm.add(ev.mean * len);
            ss.add(ev.sd * len);
            sd.add(ev.len_s * len);
            total += ev.length;
            ++j;
        }
        Pos ps;
        ps.position = first.position;
        ps.kmer = first.kmer;
        ps.mean = np_round(sm.sum / (double)total, 10.0);      // (sum_norm_mean/total_length).round(1)
        ps.sd = ss.sum / (double)total;
        ps.dwell = sd.sum / (double)total;
        out.push_back(ps);
        i = j;
    }
    return true;
}

struct IdxRun {                                       // tx = id in order of first appearance
    uint32_t tx; long long read; int64_t start, end;
    IdxRun() {}                                       // no zero-fill on vector::resize: the rows are first touched by the threads that fill them
    IdxRun(uint32_t t, long long r, int64_t a, int64_t b) : tx(t), read(r), start(a), end(b) {}
};

struct SiteRow { long long pos; std::string kmer; long long read; size_t f; };   // f = offset of its 3 (2w+1) features

struct TxOut {
    std::string json;                                  // all records of the transcript
    std::vector<std::array<long long, 4>> recs;        // pos, offset in json, length, n_reads
    bool wanted = false, logged = false;
    int rc = 0;
    std::string err;
};

// One transcript: combine -> runs of consecutive positions -> +-w window -> DRACH centre -> group by position
// (parallel_preprocess_tx :328-396 + preprocess_tx :399-488 + filter_events :51-168)
void preprocess_transcript(const char *base, size_t file_size, const std::string &tx, const std::vector<IdxRun> &idx,
                           const std::vector<uint32_t> &rows, int readcount_min, int readcount_max, int min_segment_count,
                           int w, int compress, TxOut &o)
{
    // data_dict: read_index -> combined events, insertion order of first appearance (a repeated
    // read_index overwrites its entry but keeps its place, like a Python dict)
    std::vector<long long> read_order;
    std::unordered_map<long long, std::vector<Pos>> by_read;
    int readcount = 0;
    for (uint32_t ri : rows) {
        const IdxRun &r = idx[ri];
        if (r.start < 0 || r.end > (int64_t)file_size || r.start > r.end) { o.rc = M6A_IO_EFORMAT; o.err = "index row outside eventalign.txt"; return; }
        std::vector<Pos> ps;
        if (!combine_read(base + r.start, base + r.end, base + file_size, ps)) { o.rc = M6A_IO_EFORMAT; o.err = "malformed eventalign line for " + tx; return; }
        if (ps.size() > 1) {                          // `if data.size > 1`
            if (!by_read.count(r.read)) read_order.push_back(r.read);
            by_read[r.read] = std::move(ps);
        }
        if (++readcount > readcount_max) break;      // (sic) up to readcount_max + 1 reads
    }
    if (readcount < readcount_min) return;
    o.wanted = true;
    const size_t W = (size_t)w, NF = 3 * (2 * W + 1);
    std::vector<SiteRow> sites;
    std::vector<double> feat;
    for (long long rd : read_order) {
        const std::vector<Pos> &ps = by_read[rd];     // sorted by position already
        size_t a = 0;
        while (a < ps.size()) {                       // runs of consecutive positions (partition_into_continuous_positions)
            size_t b = a + 1;
            while (b < ps.size() && ps[b].position == ps[b - 1].position + 1) ++b;
            if (b - a >= 2 * W + 1) {
                for (size_t i = a + W; i + W < b; i++) {
                    if (!is_drach(ps[i].kmer)) continue;
                    SiteRow sr;
                    sr.pos = ps[i].position + 2;      // centre of the 5-mer
                    sr.kmer.assign(ps[i - W].kmer);   // combine_sequence: first 5-mer + the last base of every later one
                    for (size_t k = i - W + 1; k <= i + W; k++) sr.kmer += ps[k].kmer.back();
                    sr.f = feat.size();
                    for (size_t k = i - W; k <= i + W; k++) {      // roll(): previous ..., centre, next ...; [dwell, sd, mean] each
                        feat.push_back(ps[k].dwell); feat.push_back(ps[k].sd); feat.push_back(ps[k].mean);
                    }
                    sr.read = rd;
                    sites.push_back(std::move(sr));
                }
            }
            a = b;
        }
    }
    if (sites.empty()) return;                        // preprocess_tx returns before its log line (dataprep_utils.py:415,431)
    o.logged = true;
    // reference: np.argsort(positions) (unstable, machine-dependent order inside a position);
    // here: stable, i.e. reads stay in index order inside a position
    std::stable_sort(sites.begin(), sites.end(), [](const SiteRow &x, const SiteRow &y) { return x.pos < y.pos; });
    size_t i = 0;
    while (i < sites.size()) {
        size_t j = i;
        while (j < sites.size() && sites[j].pos == sites[i].pos) {
            if (sites[j].kmer != sites[i].kmer) { o.rc = M6A_IO_EFORMAT; o.err = "reads disagree on the sequence at " + tx + ":" + std::to_string(sites[i].pos); return; }
            ++j;
        }
        if ((int)(j - i) >= min_segment_count) {
            const size_t start = o.json.size();
            o.json.reserve(start + (j - i) * (NF * 22 + 28) + tx.size() + 64);
            o.json += "{\"" + tx + "\":{\"" + std::to_string(sites[i].pos) + "\":{\"" + sites[i].kmer + "\":[";
            for (size_t k = i; k < j; k++) {
                o.json += k == i ? "[" : ",[";
                for (size_t c = 0; c < NF; c++) {
                    const double v = feat[sites[k].f + c];
                    if (compress) {
                        const double r = np_round(v, 1000.0);
                        if (!repr_rounded(r, 1000.0, 3, o.json)) py_repr(r, o.json);
                    } else if (c % 3 != 2 || !repr_rounded(v, 10.0, 1, o.json)) {    // c % 3 == 2: the mean, already rounded to one decimal
                        py_repr(v, o.json);
                    }
                    o.json += ',';
                }
                if (sites[k].read >= 0 && sites[k].read < (1LL << 53)) {      // repr(float(int)): the digits and ".0"
                    char ib[24];
                    const int n = format_i64(sites[k].read, ib);
                    o.json.append(ib, (size_t)n);
                    o.json += ".0";
                } else py_repr((double)sites[k].read, o.json);
                o.json += ']';
            }
            o.json += "]}}}\n";
            o.recs.push_back({sites[i].pos, (long long)start, (long long)(o.json.size() - start), (long long)(j - i)});
        }
        i = j;
    }
}

// One byte range of eventalign.txt (whole lines): its contiguous (contig, read_index) runs; contig names are interned per
// range first (views into the mapping), merged into the file-wide table afterwards
struct LocalRun { uint32_t tx; long long read; int64_t start, end; };
struct IndexChunk {
    std::vector<LocalRun> runs;
    std::vector<std::string_view> names;            // local id -> name, in order of first appearance in the range
    std::vector<uint32_t> global;                   // local id -> file-wide id (filled by the sequential merge)
    size_t out = 0;                                 // where this range's rows start in the file-wide index
    bool drop_first = false;                        // its first run continues the previous range's last run
    std::string text;                               // its rows of eventalign.index
    int rc = 0;
    int64_t bad_at = -1;
};

void index_range(const char *base, const char *p, const char *end, IndexChunk &out)
{
    std::unordered_map<std::string_view, uint32_t> ids;
    while (p < end) {
        const char *le = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *next = le ? le + 1 : end;
        const char *t1 = (const char *)memchr(p, '\t', (size_t)(next - p));
        if (!t1) { p = next; continue; }
        const char *q = t1;
        for (int k = 0; k < 2 && q; k++) q = (const char *)memchr(q + 1, '\t', (size_t)(next - q - 1));
        if (!q) { out.rc = M6A_IO_EFORMAT; out.bad_at = (int64_t)(p - base); return; }
        const long long read = atoll(q + 1);
        const std::string_view name(p, (size_t)(t1 - p));
        if (out.runs.empty() || out.runs.back().read != read || out.names[out.runs.back().tx] != name) {
            uint32_t id;
            if (!out.runs.empty() && out.names[out.runs.back().tx] == name) id = out.runs.back().tx;
            else {
                auto it = ids.find(name);
                if (it == ids.end()) { id = (uint32_t)out.names.size(); ids.emplace(name, id); out.names.push_back(name); }
                else id = it->second;
            }
            out.runs.push_back(LocalRun{id, read, (int64_t)(p - base), (int64_t)(p - base)});
        }
        out.runs.back().end = (int64_t)(next - base);
        p = next;
    }
}

inline void append_ll(std::string &s, long long v)
{
    char b[24];
    auto r = std::to_chars(b, b + sizeof b, v);
    s.append(b, (size_t)(r.ptr - b));
}

// An exception on a worker thread would be std::terminate, one on the calling thread would unwind through the C ABI: both
// are caught here, the remaining items are dropped, and ONE std::bad_alloc is thrown on the calling thread after every worker
// has been joined -- m6a_io_dataprep turns it into M6A_IO_ENOMEM.
template <class F>
void on_threads(int nw, int n_items, F &&f)
{
    std::atomic<int> next{0};
    std::atomic<bool> threw{false};
    auto worker = [&]() {
        for (int k; (k = next.fetch_add(1)) < n_items;) {
            try { f(k); } catch (...) { threw = true; next = n_items; }
        }
    };
    std::vector<std::thread> th;
    try {
        for (int t = 1; t < std::min(nw, n_items); t++) th.emplace_back(worker);
    } catch (...) { /* fewer threads than asked for: the ones that started and this one do the work */ }
    worker();
    for (auto &t : th) t.join();
    if (threw) throw std::bad_alloc();
}

}  // namespace

extern "C" int m6a_io_py_repr(double v, char *buf40)
{
    std::string s;
    py_repr(v, s);
    if (s.size() >= 40) return -1;
    std::memcpy(buf40, s.c_str(), s.size() + 1);
    return (int)s.size();
}

// the fast path of the data.json writer for numbers np.round produced (repr_rounded above): the text, or -1 where it declines
// and py_repr is used instead.  Exported so the tests can pin it against Python's repr(round(x, digits)).
extern "C" int m6a_io_repr_rounded(double v, int digits, char *buf40)
{
    if (digits != 1 && digits != 3) return -1;
    std::string s;
    if (!repr_rounded(v, digits == 1 ? 10.0 : 1000.0, digits, s) || s.size() >= 40) return -1;
    std::memcpy(buf40, s.c_str(), s.size() + 1);
    return (int)s.size();
}

static int dataprep_impl(const char *eventalign_path, const char *out_dir, int n_threads,
                         int readcount_min, int readcount_max, int min_segment_count, int n_neighbors,
                         int compress, int skip_index);

// Test hook (tests/test_dataprep.py): M6A_IO_TEST_THROW=index|transcript|index_file|bookkeeping makes that phase run out of memory
// once, on whichever thread gets there first -- what the try/catch blocks below are for cannot be provoked reliably otherwise.
// Compiled ONLY into test builds (-DM6A_IO_TEST_HOOKS: tests/test_dataprep.py and tests/sanitize.sh build their own copy); the
// shipped libm6a_io.so reads no such variable.
#ifdef M6A_IO_TEST_HOOKS
static void test_throw(const char *where)
{
    static const char *want = getenv("M6A_IO_TEST_THROW");
    if (want && !strcmp(want, where)) throw std::bad_alloc();
}
#else
static inline void test_throw(const char *) {}
#endif

// What dataprep holds open while code that can throw runs (std::bad_alloc in its bookkeeping, std::system_error from a
// std::thread constructor): closed on every exit path, the stdio streams BEFORE the buffers setvbuf gave them go away (a guard is
// declared after what it must outlive), and outputs a failed run leaves half-written are removed -- a truncated data.json next
// to a data.info that indexes into it is worse than no output.
struct FileCloser {
    FILE *f = nullptr;
    explicit FileCloser(FILE *f_) : f(f_) {}
    FileCloser(const FileCloser &) = delete;
    FileCloser &operator=(const FileCloser &) = delete;
    int close() { FILE *g = f; f = nullptr; return g ? fclose(g) : 0; }
    ~FileCloser() { (void)close(); }
};
struct FdCloser {
    int fd = -1;
    explicit FdCloser(int fd_) : fd(fd_) {}
    FdCloser(const FdCloser &) = delete;
    FdCloser &operator=(const FdCloser &) = delete;
    int release() { const int g = fd; fd = -1; return g; }
    ~FdCloser() { if (fd >= 0) (void)::close(fd); }
};
struct PartialOutputs {
    std::vector<std::string> paths;
    bool keep = false;
    void add(const std::string &p) { paths.push_back(p); }
    ~PartialOutputs() { if (!keep) for (const std::string &p : paths) (void)::unlink(p.c_str()); }
};

// No exception crosses the C ABI: the files this targets run to hundreds of GB, and the index (32 B per read run), the
// per-transcript buffers and the writers' text can all exhaust memory -- that is M6A_IO_ENOMEM, not an aborted interpreter.
extern "C" int m6a_io_dataprep(const char *eventalign_path, const char *out_dir, int n_threads,
                               int readcount_min, int readcount_max, int min_segment_count, int n_neighbors,
                               int compress, int skip_index)
{
    try {
        return dataprep_impl(eventalign_path, out_dir, n_threads, readcount_min, readcount_max, min_segment_count, n_neighbors,
                             compress, skip_index);
    } catch (const std::bad_alloc &) {
        return fail(M6A_IO_ENOMEM, "dataprep: out of memory");
    } catch (const std::exception &e) {
        return fail(M6A_IO_EIO, "dataprep: %s", e.what());
    } catch (...) {
        return fail(M6A_IO_EIO, "dataprep: unexpected exception");
    }
}

static int dataprep_impl(const char *eventalign_path, const char *out_dir, int n_threads,
                         int readcount_min, int readcount_max, int min_segment_count, int n_neighbors,
                         int compress, int skip_index)
{
    if (!eventalign_path || !out_dir) return fail(M6A_IO_EINVAL, "null argument");
    if (n_neighbors < 1 || n_neighbors > 16) return fail(M6A_IO_EINVAL, "n_neighbors must be 1..16");
    PhaseTrace trace;
    Mapped ev;
    const char *pm = getenv("M6A_IO_POPULATE_MAX_MB");
    int rc = ev.open(eventalign_path, (size_t)(pm ? atoll(pm) : 2048) << 20);
    if (rc) return rc;
    const char *base = ev.p, *end = ev.p + ev.n;
    const std::string dir(out_dir);
    trace.mark("dataprep: map");

    // ---- index (parallel_index, dataprep_utils.py:187-266): one row per contiguous (contig, read_index) run.
    // Byte ranges of whole lines on all threads; the runs that meet at a range boundary are stitched afterwards.
    std::vector<IdxRun> idx;
    std::vector<std::string> tx_names;                 // id -> name, ids in order of first appearance
    std::unordered_map<std::string, uint32_t> tx_ids;
    auto intern = [&](const char *p, size_t n) -> uint32_t {
        if (!idx.empty() && tx_names[idx.back().tx].size() == n && memcmp(tx_names[idx.back().tx].data(), p, n) == 0) return idx.back().tx;
        std::string key(p, n);
        auto it = tx_ids.find(key);
        if (it != tx_ids.end()) return it->second;
        const uint32_t id = (uint32_t)tx_names.size();
        tx_ids.emplace(key, id);
        tx_names.push_back(std::move(key));
        return id;
    };
    const std::string idx_path = dir + "/eventalign.index";
    // declared before the index writer and the stream guards: destroyed after them, i.e. after the writer thread has been joined
    // and every stream closed -- then, unless the run succeeded, the files this call created are removed
    PartialOutputs partial;
    // eventalign.index is written BEHIND the transcript pass: nothing downstream reads the file (the pass works from `idx`), and
    // writing 2.6 GB of it was 1.1 s of a 7.2 s run with every worker waiting at two barriers per batch (format, then pwrite).
    // Declared after `idx` / `tx_names`, which it reads: its destructor joins the thread before they go, on every return path.
    struct IndexFileWriter {
        std::thread th;
        std::atomic<bool> ok{true}, oom{false};
        void wait() { if (th.joinable()) th.join(); }
        ~IndexFileWriter() { wait(); }
    } idx_writer;
    if (skip_index) {
        FILE *f = fopen(idx_path.c_str(), "r");
        if (!f) return fail(M6A_IO_EIO, "--skip_index but %s does not exist", idx_path.c_str());
        FileCloser fc(f);                                  // idx.push_back / intern can throw
        char line[4096];
        bool first = true;
        while (fgets(line, sizeof line, f)) {
            if (first) { first = false; continue; }
            char *c = strrchr(line, ',');
            if (!c) continue;
            IdxRun r;
            r.end = atoll(c + 1); *c = 0;
            c = strrchr(line, ','); if (!c) continue; r.start = atoll(c + 1); *c = 0;
            c = strrchr(line, ','); if (!c) continue; r.read = atoll(c + 1); *c = 0;
            r.tx = intern(line, strlen(line));
            idx.push_back(r);
        }
        (void)fc.close();
    } else {
        const char *body = (const char *)memchr(base, '\n', ev.n);
        if (!body) return fail(M6A_IO_EFORMAT, "%s: no header line", eventalign_path);
        ++body;
        const size_t n_body = (size_t)(end - body);
        const char *rk = getenv("M6A_IO_INDEX_RANGE_KB");                                      // tests: small ranges on small files
        const size_t min_range = std::max<size_t>(1, (size_t)(rk ? atoll(rk) : 8192)) << 10;  // a range is at least 8 MB
        const int nw = n_workers(n_threads, (int64_t)std::max<size_t>(1, n_body / min_range));
        const int NC = (int)std::max<size_t>(1, std::min<size_t>((size_t)nw * 4, n_body / min_range + 1));
        std::vector<const char *> cut((size_t)NC + 1, end);
        cut[0] = body;
        for (int k = 1; k < NC; k++) {
            const char *p = body + n_body / (size_t)NC * (size_t)k;
            if (p < cut[(size_t)k - 1]) p = cut[(size_t)k - 1];
            const char *nl = p < end ? (const char *)memchr(p, '\n', (size_t)(end - p)) : nullptr;
            cut[(size_t)k] = nl ? nl + 1 : end;
        }
        std::vector<IndexChunk> chunks((size_t)NC);
        on_threads(nw, NC, [&](int k) { if (k == NC / 2) test_throw("index"); index_range(base, cut[(size_t)k], cut[(size_t)k + 1], chunks[(size_t)k]); });
        trace.mark("dataprep: index ranges");
        for (const auto &c : chunks)
            if (c.rc) return fail(c.rc, "%s: short line at byte %lld", eventalign_path, (long long)c.bad_at);
        // sequential, per RANGE (not per run): contig names into the file-wide table in order of first appearance; a first
        // run that continues the previous range's last one (same contig, same read, adjacent bytes) is folded into it
        size_t total = 0, n_local = 0;
        for (const auto &c : chunks) n_local += c.names.size();
        std::unordered_map<std::string_view, uint32_t> view_ids;      // keys are views into the mapping: nothing is copied to look a name up
        view_ids.reserve(n_local);
        tx_names.reserve(n_local);
        IndexChunk *open_c = nullptr;                                  // the range holding the file's last run so far
        for (auto &c : chunks) {
            c.global.resize(c.names.size());
            for (size_t i = 0; i < c.names.size(); i++) {
                auto it = view_ids.find(c.names[i]);
                if (it == view_ids.end()) {
                    const uint32_t id = (uint32_t)tx_names.size();
                    tx_names.emplace_back(c.names[i]);
                    view_ids.emplace(c.names[i], id);
                    c.global[i] = id;
                } else c.global[i] = it->second;
            }
            if (!c.runs.empty() && open_c) {
                LocalRun &last = open_c->runs.back();
                const LocalRun &first = c.runs.front();
                if (last.end == first.start && last.read == first.read && open_c->global[last.tx] == c.global[first.tx]) {
                    last.end = first.end;
                    c.drop_first = true;
                }
            }
            c.out = total;
            total += c.runs.size() - (c.drop_first ? 1 : 0);
            if (c.runs.size() > (c.drop_first ? 1u : 0u)) open_c = &c;
        }
        if (total > 0xffffffffull) return fail(M6A_IO_EINVAL, "more than 2^32 index rows");
        trace.mark("dataprep: index names merged");
        idx.resize(total);
        // every range fills its rows of the index; their text is formatted and written a batch of ranges at a time, so the
        // index file (a tenth of the eventalign.txt) never sits in memory as a whole
        on_threads(nw, NC, [&](int k) {
            IndexChunk &c = chunks[(size_t)k];
            size_t o = c.out;
            for (size_t i = c.drop_first ? 1 : 0; i < c.runs.size(); i++) {
                const LocalRun &r = c.runs[i];
                idx[o++] = IdxRun{c.global[r.tx], r.read, r.start, r.end};
            }
            std::vector<LocalRun>().swap(c.runs);
        });
        trace.mark("dataprep: index rows filled");
        // the index file: on a background thread (see idx_writer above) -- a batch of row ranges is formatted on a quarter of
        // the workers, then every range pwrite()s its own text at its offset, while the transcript pass runs on all of them
        const int fd = ::open(idx_path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) return fail(M6A_IO_EIO, "cannot write %s", idx_path.c_str());
        FdCloser fd_guard(fd);                             // until the writer thread exists and owns it (std::thread's constructor can throw)
        partial.add(idx_path);
        test_throw("bookkeeping");
        std::vector<size_t> bounds((size_t)NC + 1, total);
        for (int k = 0; k < NC; k++) bounds[(size_t)k] = chunks[(size_t)k].out;
        const int nbg = std::max(1, nw / 4);
        idx_writer.th = std::thread([&idx, &tx_names, &idx_writer, fd, nbg, bounds = std::move(bounds)]() {
          try {
            test_throw("index_file");
            static const char kIdxHeader[] = "transcript_id,read_index,pos_start,pos_end\n";
            int64_t file_off = (int64_t)sizeof(kIdxHeader) - 1;
            bool io_ok = ::pwrite(fd, kIdxHeader, sizeof(kIdxHeader) - 1, 0) == (ssize_t)(sizeof(kIdxHeader) - 1);
            const int n_ranges = (int)bounds.size() - 1, batch = std::max(1, nbg * 2);
            std::vector<std::string> text((size_t)batch);
            std::vector<int64_t> at((size_t)batch);
            for (int k0 = 0; k0 < n_ranges && io_ok; k0 += batch) {
                const int k1 = std::min(n_ranges, k0 + batch);
                on_threads(nbg, k1 - k0, [&](int kk) {
                    std::string &t = text[(size_t)kk];
                    t.clear();
                    const size_t o0 = bounds[(size_t)(k0 + kk)], o1 = bounds[(size_t)(k0 + kk) + 1];
                    t.reserve((o1 - o0) * 48);
                    for (size_t o = o0; o < o1; o++) {
                        const IdxRun &r = idx[o];
                        t += tx_names[r.tx];
                        t += ',';
                        append_ll(t, r.read);
                        t += ',';
                        append_ll(t, (long long)r.start);
                        t += ',';
                        append_ll(t, (long long)r.end);
                        t += '\n';
                    }
                });
                for (int k = k0; k < k1; k++) { at[(size_t)(k - k0)] = file_off; file_off += (int64_t)text[(size_t)(k - k0)].size(); }
                std::atomic<bool> ok{true};
                on_threads(nbg, k1 - k0, [&](int kk) {
                    const char *p = text[(size_t)kk].data();
                    size_t n = text[(size_t)kk].size();
                    int64_t o = at[(size_t)kk];
                    while (n) {
                        const ssize_t w = ::pwrite(fd, p, n, (off_t)o);
                        if (w < 0) { if (errno == EINTR) continue; ok = false; break; }
                        p += w; n -= (size_t)w; o += w;
                    }
                });
                io_ok = ok;
            }
            if (::close(fd) != 0) io_ok = false;
            if (!io_ok) idx_writer.ok = false;
          } catch (...) {                              // out of memory for a batch of text: reported as the file's failure
            (void)::close(fd);
            idx_writer.ok = false;
            idx_writer.oom = true;
          }
        });
        (void)fd_guard.release();                          // the thread closes it
    }
    trace.mark("dataprep: index file handed to its writer");
    if (ev.p && ev.n) (void)madvise((void *)ev.p, ev.n, MADV_NORMAL);       // the transcript pass jumps between a read's runs

    // ---- transcripts in order of first appearance (= id order), with their index rows in file order
    const int64_t NT = (int64_t)tx_names.size();
    std::vector<std::vector<uint32_t>> tx_rows((size_t)NT);
    if (idx.size() > 0xffffffffull) return fail(M6A_IO_EINVAL, "more than 2^32 index rows");
    {
        std::vector<uint32_t> cnt((size_t)NT, 0);
        for (const IdxRun &r : idx) cnt[r.tx]++;
        for (int64_t t = 0; t < NT; t++) tx_rows[(size_t)t].reserve(cnt[(size_t)t]);
    }
    for (size_t i = 0; i < idx.size(); i++) tx_rows[idx[i].tx].push_back((uint32_t)i);

    trace.mark("dataprep: rows per transcript");
    // ---- per transcript on all threads, written in transcript order AS SOON AS every earlier one is written: memory holds a
    // bounded window of finished transcripts, not the whole data.json
    std::vector<char> jbuf(4 << 20), ibuf(1 << 20);     // before the streams that use them: the guards below close first
    FILE *fj = fopen((dir + "/data.json").c_str(), "w"), *fi = fopen((dir + "/data.info").c_str(), "w"), *fl = fopen((dir + "/data.log").c_str(), "w");
    FileCloser cj(fj), ci(fi), cl(fl);
    if (fj) partial.add(dir + "/data.json");
    if (fi) partial.add(dir + "/data.info");
    if (fl) partial.add(dir + "/data.log");
    if (!fj || !fi || !fl) return fail(M6A_IO_EIO, "cannot write into %s", out_dir);
    setvbuf(fj, jbuf.data(), _IOFBF, jbuf.size());
    setvbuf(fi, ibuf.data(), _IOFBF, ibuf.size());
    fputs("transcript_id,transcript_position,start,end,n_reads\n", fi);
    const int nw = n_workers(n_threads, NT);
    // How far the workers may run ahead of the writer is bounded by the BYTES of finished-but-unwritten JSON, not by a
    // transcript count: transcripts differ in size by orders of magnitude, and with a window of 8 per thread everybody
    // stood waiting behind each large one (16 threads were 7 x one thread; profiles/r04_dataprep_sweep.txt)
    const char *pb = getenv("M6A_IO_PENDING_MB");
    const size_t pending_budget = (size_t)(pb && atoll(pb) > 0 ? atoll(pb) : 256) << 20;
    const int64_t window = std::max<int64_t>(4096, (int64_t)nw * 256);      // and a count, so that `outs` bookkeeping stays small
    size_t pending_bytes = 0;                          // JSON of deposited, not yet written transcripts (guarded by mu)
    std::vector<std::unique_ptr<TxOut>> outs((size_t)NT);
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<int64_t> next{0};
    int64_t written = 0;                               // transcripts [0, written) are on disk (guarded by mu)
    bool writing = false, failed = false;
    int fail_rc = 0;
    std::string fail_msg;
    long long off = 0;                                 // bytes of data.json written so far (writer only)
    size_t peak_pending = 0;
    auto write_one = [&](int64_t t, const TxOut &o) {
        if (!o.wanted) return;
        if (!o.json.empty()) fwrite(o.json.data(), 1, o.json.size(), fj);
        for (const auto &r : o.recs)
            fprintf(fi, "%s,%lld,%lld,%lld,%lld\n", tx_names[(size_t)t].c_str(), r[0], off + r[1], off + r[1] + r[2], r[3]);
        off += (long long)o.json.size();
        // logged like the reference: only transcripts that yielded at least one DRACH window (dataprep_utils.py:415,431,472)
        if (o.logged) fprintf(fl, "%s: Data preparation ... Done.\n", tx_names[(size_t)t].c_str());
    };
    auto worker_body = [&]() {
        for (;;) {
            const int64_t t = next.fetch_add(1);
            if (t >= NT) break;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return failed || t == written || (t < written + window && pending_bytes < pending_budget); });
                if (failed) break;
            }
            std::unique_ptr<TxOut> o(new TxOut);
            if (t == NT / 2) test_throw("transcript");
            preprocess_transcript(base, ev.n, tx_names[(size_t)t], idx, tx_rows[(size_t)t], readcount_min, readcount_max,
                                  min_segment_count, n_neighbors, compress, *o);
            std::vector<uint32_t>().swap(tx_rows[(size_t)t]);
            std::unique_lock<std::mutex> lk(mu);
            if (o->rc && !failed) { failed = true; fail_rc = o->rc; fail_msg = o->err; cv.notify_all(); }
            pending_bytes += o->json.size();
            peak_pending = std::max(peak_pending, pending_bytes);
            outs[(size_t)t] = std::move(o);
            if (writing || failed) continue;
            // this thread becomes the writer for as long as the next transcript in line is finished
            writing = true;
            while (!failed && written < NT && outs[(size_t)written]) {
                std::vector<std::unique_ptr<TxOut>> batch;
                const int64_t first = written;
                int64_t k = written;
                while (k < NT && outs[(size_t)k]) batch.push_back(std::move(outs[(size_t)k++]));
                size_t batch_bytes = 0;
                for (const auto &bo : batch) batch_bytes += bo->json.size();
                lk.unlock();
                for (size_t b = 0; b < batch.size(); b++) write_one(first + (int64_t)b, *batch[b]);
                batch.clear();
                lk.lock();
                written = k;
                pending_bytes -= batch_bytes;
                cv.notify_all();
            }
            writing = false;
        }
    };
    // a worker that runs out of memory (the transcript's feature / JSON buffers) fails the job instead of the process: the
    // others see `failed` at their next wait and leave
    auto worker = [&]() {
        try {
            worker_body();
        } catch (...) {
            std::lock_guard<std::mutex> lk(mu);
            if (!failed) { failed = true; fail_rc = M6A_IO_ENOMEM; fail_msg = "dataprep: out of memory in the transcript pass"; }
            writing = false;
            cv.notify_all();
        }
    };
    {
        std::vector<std::thread> th;
        try {
            for (int w = 1; w < nw; w++) th.emplace_back(worker);
        } catch (...) { /* fewer threads than asked for */ }
        worker();
        for (auto &t : th) t.join();
    }
    int bad = 0;
    bad |= cj.close(); bad |= ci.close(); bad |= cl.close();
    if (failed) return fail(fail_rc, "%s", fail_msg.c_str());
    if (written != NT) return fail(M6A_IO_EIO, "internal: %lld of %lld transcripts written", (long long)written, (long long)NT);
    if (bad) return fail(M6A_IO_EIO, "cannot close outputs in %s", out_dir);
    trace.mark("dataprep: transcripts");
    idx_writer.wait();
    if (idx_writer.oom) return fail(M6A_IO_ENOMEM, "out of memory while writing %s", idx_path.c_str());
    if (!idx_writer.ok) return fail(M6A_IO_EIO, "cannot write %s", idx_path.c_str());
    trace.mark("dataprep: index file finished behind them");
    if (trace.on) fprintf(stderr, "m6a_io: dataprep peak of finished-but-unwritten json: %.1f MB (budget %.0f MB, window %lld transcripts)\n", peak_pending / 1e6, pending_budget / 1e6, (long long)window);
    partial.keep = true;
    return M6A_IO_OK;
}
