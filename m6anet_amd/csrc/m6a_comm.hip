// m6a_comm.hip -- RCCL bound at run time and the job's one exchange (split out of m6a_api.hip; internal declarations: m6a_ctx.h)
#include "m6a_ctx.h"

using namespace m6a_detail;

namespace m6a_detail {

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

void comm_release(m6a_ctx *c)
{
    if (!c->comm) return;
    Rccl *R = rccl();
    if (R->CommDestroy) (void)R->CommDestroy(c->comm);
    c->comm = nullptr;
}

#define RCCLCHK(c, R, expr)                                                                             \
    do {                                                                                                \
        const int e_ = (expr);                                                                          \
        if (e_ != 0) return fail((c), M6A_EHIP, "%s: %s", #expr, (R)->GetErrorString ? (R)->GetErrorString(e_) : "RCCL error"); \
    } while (0)


}  // namespace m6a_detail

extern "C" {

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


}  // extern "C"
