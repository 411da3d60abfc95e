// stub_rccl.cpp -- TEST INFRASTRUCTURE, never shipped, never loaded by the product on its own.
//
// A stand-in for librccl that lets SEVERAL RANKS SHARE ONE GPU: the twelve nccl* entry points m6a_comm.hip binds at run time
// (M6A_RCCL_LIB names the library to bind), carried over files in /dev/shm and staged through host memory.  A real RCCL refuses
// two ranks on one device, and a test lease is one device, so without this the W > 1 branch of gather_group -- receive offsets
// (cuts[r] - cuts[0]) * esz, ragged / empty shards, the group that must be closed when a Send fails -- would first execute on
// somebody's 8-GPU node.  tests/test_gpu_comm_stub.py builds it with hipcc and runs 2 / 3 / 8 ranks.
//
// Semantics kept from RCCL, as far as m6a_comm.hip relies on them:
//   * ncclCommInitRank is collective: it returns when all `world` ranks of the same unique id have arrived;
//   * ncclSend / ncclRecv inside ncclGroupStart .. ncclGroupEnd only QUEUE; the exchange happens at ncclGroupEnd, ordered after
//     everything queued on the stream before it (the stub synchronises the stream, copies D2H, publishes; receives poll, copy
//     H2D) and complete when ncclGroupEnd returns -- stronger than RCCL's stream-asynchronous completion, never weaker;
//   * messages between a (source, destination) pair match in order; counts are ELEMENTS of the given datatype;
//   * a failing call returns a non-zero ncclResult_t and ncclGetErrorString names it.
// Test hooks: STUB_RCCL_FAIL_SEND=<k> makes the k-th ncclSend of the process fail (k counts from 1); stub_rccl_group_depth()
// reports the calling thread's open-group depth; STUB_RCCL_LOG=<file> appends one line per executed send / receive.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

struct UniqueId { char internal[128]; };

struct Comm {
    std::string token;
    int rank = 0, world = 0, device = 0;
    std::vector<long long> sent, received;      // per peer: messages exchanged so far (pairwise order)
};

struct Op { bool send; const void *src; void *dst; size_t bytes; int peer; Comm *comm; hipStream_t stream; };

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
std::atomic<int> g_sends{0};

const char *kErrors[] = {"no error (stub)", "unhandled cuda error (stub)", "unhandled system error (stub)", "internal error (stub)",
                         "invalid argument (stub)", "invalid usage (stub)", "remote error (stub)", "in progress (stub)"};

size_t dtype_bytes(int t)
{
    switch (t) {                // ncclDataType_t
    case 0: case 1: return 1;   // int8, uint8
    case 2: case 3: return 4;   // int32, uint32
    case 4: case 5: return 8;   // int64, uint64
    case 6: return 2;           // float16
    case 7: return 4;           // float32
    case 8: return 8;           // float64
    case 9: return 2;           // bfloat16
    default: return 0;
    }
}

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

double timeout_s()
{
    const char *e = getenv("STUB_RCCL_TIMEOUT");
    return e && atof(e) > 0 ? atof(e) : 120.0;
}

std::string base(const std::string &token) { return "/dev/shm/m6astub_" + token; }

bool write_file(const std::string &path, const void *data, size_t n)
{
    const std::string tmp = path + ".tmp" + std::to_string((long long)getpid());
    const int fd = ::open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
    if (fd < 0) return false;
    const char *p = (const char *)data;
    size_t left = n;
    while (left) {
        const ssize_t w = ::write(fd, p, left);
        if (w <= 0) { ::close(fd); ::unlink(tmp.c_str()); return false; }
        p += w; left -= (size_t)w;
    }
    ::close(fd);
    return ::rename(tmp.c_str(), path.c_str()) == 0;      // atomic publish: a reader sees the whole message or none
}

bool wait_for(const std::string &path)
{
    const double t0 = now_s();
    struct stat st;
    while (::stat(path.c_str(), &st) != 0) {
        if (now_s() - t0 > timeout_s()) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    return true;
}

void log_line(const char *what, const Comm *c, int peer, size_t bytes, const void *ptr)
{
    const char *f = getenv("STUB_RCCL_LOG");
    if (!f) return;
    char line[256];
    const int n = snprintf(line, sizeof line, "%s rank=%d peer=%d bytes=%zu ptr=%p\n", what, c->rank, peer, bytes, ptr);
    const int fd = ::open(f, O_WRONLY | O_CREAT | O_APPEND, 0600);
    if (fd >= 0) { (void)!::write(fd, line, (size_t)n); ::close(fd); }
}

int run_send(const Op &o)
{
    Comm *c = o.comm;
    std::vector<char> host(o.bytes);
    if (hipStreamSynchronize(o.stream) != hipSuccess) return 1;
    if (o.bytes && hipMemcpy(host.data(), o.src, o.bytes, hipMemcpyDefault) != hipSuccess) return 1;
    const long long seq = c->sent[(size_t)o.peer]++;
    const std::string path = base(c->token) + "_m_" + std::to_string(c->rank) + "_" + std::to_string(o.peer) + "_" + std::to_string(seq);
    if (!write_file(path, host.data(), o.bytes)) return 2;
    log_line("send", c, o.peer, o.bytes, o.src);
    return 0;
}

int run_recv(const Op &o)
{
    Comm *c = o.comm;
    const long long seq = c->received[(size_t)o.peer]++;
    const std::string path = base(c->token) + "_m_" + std::to_string(o.peer) + "_" + std::to_string(c->rank) + "_" + std::to_string(seq);
    if (!wait_for(path)) return 6;
    std::vector<char> host(o.bytes);
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return 2;
    struct stat st;
    if (fstat(fd, &st) != 0 || (size_t)st.st_size != o.bytes) { ::close(fd); return 4; }     // sender and receiver disagree on the size
    size_t got = 0;
    while (got < o.bytes) {
        const ssize_t r = ::read(fd, host.data() + got, o.bytes - got);
        if (r <= 0) { ::close(fd); return 2; }
        got += (size_t)r;
    }
    ::close(fd);
    ::unlink(path.c_str());
    if (hipStreamSynchronize(o.stream) != hipSuccess) return 1;
    if (o.bytes && hipMemcpy(o.dst, host.data(), o.bytes, hipMemcpyDefault) != hipSuccess) return 1;
    log_line("recv", c, o.peer, o.bytes, o.dst);
    return 0;
}

int flush_ops()
{
    std::vector<Op> ops;
    ops.swap(g_ops);
    int first = 0;
    for (const Op &o : ops) if (o.send) { const int e = run_send(o); if (e && !first) first = e; }      // every send first: nobody blocks on us
    for (const Op &o : ops) if (!o.send) { const int e = run_recv(o); if (e && !first) first = e; }
    return first;
}

}  // namespace

extern "C" {

int ncclGetVersion(int *v) { if (!v) return 4; *v = 99999; return 0; }      // no RCCL release: the stub says what it is

const char *ncclGetErrorString(int e) { return e >= 0 && e < 8 ? kErrors[e] : "unknown error (stub)"; }

int ncclGetUniqueId(UniqueId *id)
{
    if (!id) return 4;
    static std::atomic<int> n{0};
    std::memset(id->internal, 0, sizeof id->internal);
    snprintf(id->internal, sizeof id->internal, "%lld_%lld_%d", (long long)getpid(),
             (long long)std::chrono::steady_clock::now().time_since_epoch().count(), n++);
    return 0;
}

int ncclCommInitRank(void **comm, int world, UniqueId id, int rank)
{
    if (!comm || world < 1 || rank < 0 || rank >= world) return 4;
    id.internal[sizeof id.internal - 1] = 0;
    Comm *c = new Comm;
    c->token = id.internal; c->rank = rank; c->world = world;
    c->sent.assign((size_t)world, 0); c->received.assign((size_t)world, 0);
    if (hipGetDevice(&c->device) != hipSuccess) { delete c; return 1; }
    // collective: everybody announces itself, everybody waits for everybody
    if (!write_file(base(c->token) + "_r" + std::to_string(rank), "1", 1)) { delete c; return 2; }
    for (int r = 0; r < world; r++)
        if (!wait_for(base(c->token) + "_r" + std::to_string(r))) { delete c; return 6; }
    *comm = c;
    return 0;
}

int ncclCommDestroy(void *comm)
{
    Comm *c = (Comm *)comm;
    if (!c) return 4;
    // leave a tombstone and remove the announcement when the last rank has gone (best effort: files in /dev/shm of a test run)
    (void)write_file(base(c->token) + "_d" + std::to_string(c->rank), "1", 1);
    bool all = true;
    struct stat st;
    for (int r = 0; r < c->world; r++) all = all && ::stat((base(c->token) + "_d" + std::to_string(r)).c_str(), &st) == 0;
    if (all)
        for (int r = 0; r < c->world; r++) {
            ::unlink((base(c->token) + "_d" + std::to_string(r)).c_str());
            ::unlink((base(c->token) + "_r" + std::to_string(r)).c_str());
        }
    delete c;
    return 0;
}

int ncclCommCount(void *comm, int *n) { if (!comm || !n) return 4; *n = ((Comm *)comm)->world; return 0; }
int ncclCommUserRank(void *comm, int *r) { if (!comm || !r) return 4; *r = ((Comm *)comm)->rank; return 0; }
int ncclCommCuDevice(void *comm, int *d) { if (!comm || !d) return 4; *d = ((Comm *)comm)->device; return 0; }

int ncclGroupStart() { g_depth++; return 0; }

int ncclGroupEnd()
{
    if (g_depth <= 0) return 5;
    if (--g_depth > 0) return 0;
    return flush_ops();
}

int ncclSend(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream)
{
    Comm *c = (Comm *)comm;
    const size_t esz = dtype_bytes(dtype);
    if (!c || !esz || peer < 0 || peer >= c->world || (!buf && count)) return 4;
    const int k = ++g_sends;
    const char *f = getenv("STUB_RCCL_FAIL_SEND");
    if (f && atoi(f) == k) return 1;
    g_ops.push_back(Op{true, buf, nullptr, count * esz, peer, c, stream});
    return g_depth > 0 ? 0 : flush_ops();
}

int ncclRecv(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream)
{
    Comm *c = (Comm *)comm;
    const size_t esz = dtype_bytes(dtype);
    if (!c || !esz || peer < 0 || peer >= c->world || (!buf && count)) return 4;
    g_ops.push_back(Op{false, nullptr, buf, count * esz, peer, c, stream});
    return g_depth > 0 ? 0 : flush_ops();
}

int stub_rccl_group_depth() { return g_depth; }

}  // extern "C"
