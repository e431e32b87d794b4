// RCCL over xGMI for the time-sliced objective: run-time binding (no link-time dependency), one communicator per handle.
#include "cmax_comm.h"

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enums only: every entry point is resolved with dlsym

#include <cstring>
#include <mutex>

#include "cmax_common.h"

namespace cmax {

namespace {

struct RcclApi {
    void *dso = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    std::string why;
    std::string path;  // file the bound ncclGetVersion lives in (dladdr): torch's copy when the process holds one
};

RcclApi g_api;
std::once_flag g_api_once;

void load_api() {
    // 1. the copy this process already holds (torch links its own librccl.so, SONAME librccl.so.1): two RCCL copies in
    //    one process would each bring their own topology detection and IPC handles;  2. the ROCm installation's.
    const char *names[] = {"librccl.so.1", "librccl.so"};
    for (const char *n : names) {
        g_api.dso = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (g_api.dso) break;
    }
    if (!g_api.dso) {
        const char *paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *p : paths) {
            g_api.dso = dlopen(p, RTLD_NOW | RTLD_GLOBAL);
            if (g_api.dso) break;
        }
    }
    if (!g_api.dso) {
        const char *e = dlerror();
        g_api.why = std::string("librccl could not be loaded: ") + (e ? e : "unknown dlopen error");
        return;
    }
#define CMAX_SYM(field, name)                                                \
    g_api.field = reinterpret_cast<decltype(g_api.field)>(dlsym(g_api.dso, name)); \
    if (!g_api.field) {                                                      \
        g_api.why = std::string("librccl lacks ") + name;                    \
        return;                                                              \
    }
    CMAX_SYM(GetVersion, "ncclGetVersion")
    CMAX_SYM(GetUniqueId, "ncclGetUniqueId")
    CMAX_SYM(CommInitRank, "ncclCommInitRank")
    CMAX_SYM(CommDestroy, "ncclCommDestroy")
    CMAX_SYM(AllReduce, "ncclAllReduce")
    CMAX_SYM(GroupStart, "ncclGroupStart")
    CMAX_SYM(GroupEnd, "ncclGroupEnd")
    CMAX_SYM(GetErrorString, "ncclGetErrorString")
#undef CMAX_SYM
    Dl_info info;
    if (dladdr(reinterpret_cast<void *>(g_api.GetVersion), &info) && info.dli_fname) g_api.path = info.dli_fname;
    g_api.ok = true;
}

int api_ready() {
    std::call_once(g_api_once, load_api);
    if (!g_api.ok) {
        set_error("RCCL unavailable: %s", g_api.why.c_str());
        return CMAX_ENODEV;
    }
    return 0;
}

// RCCL failures are reported as CMAX_ECOMM with the RCCL message in cmax_last_error()
#define CMAX_CHECK_RCCL(expr)                                                                       \
    do {                                                                                            \
        ncclResult_t _r = (expr);                                                                   \
        if (_r != ncclSuccess) {                                                                    \
            set_error("%s:%d %s -> RCCL error %d: %s", __FILE__, __LINE__, #expr, (int)_r, g_api.GetErrorString(_r)); \
            return CMAX_ECOMM;                                                                      \
        }                                                                                           \
    } while (0)

ncclDataType_t nccl_type(CommType t) { return t == kCommF64 ? ncclFloat64 : ncclFloat32; }
ncclRedOp_t nccl_op(CommOp o) { return o == kCommMin ? ncclMin : (o == kCommMax ? ncclMax : ncclSum); }

}  // namespace

struct Comm {
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
    hipStream_t last_stream = nullptr;  // stream of the last collective (synchronised before the communicator is destroyed)
    bool used = false;
};

static_assert(sizeof(ncclUniqueId) == CMAX_COMM_ID_BYTES, "cmax_hip.h's CMAX_COMM_ID_BYTES must be sizeof(ncclUniqueId)");

int comm_available(char *path_out, int cap) {
    int rc = api_ready();
    if (path_out && cap > 0) {
        std::strncpy(path_out, rc ? "" : g_api.path.c_str(), (size_t)cap - 1);
        path_out[cap - 1] = '\0';
    }
    return rc;
}

int comm_version() {
    if (api_ready()) return 0;
    int v = 0;
    return g_api.GetVersion(&v) == ncclSuccess ? v : 0;
}

int comm_unique_id(void *id128_host) {
    int rc = api_ready();
    if (rc) return rc;
    ncclUniqueId id;
    CMAX_CHECK_RCCL(g_api.GetUniqueId(&id));
    std::memcpy(id128_host, &id, sizeof(id));
    return 0;
}

int comm_create(const void *id128_host, int nranks, int rank, Comm **out) {
    int rc = api_ready();
    if (rc) return rc;
    ncclUniqueId id;
    std::memcpy(&id, id128_host, sizeof(id));
    Comm *c = new Comm();
    c->nranks = nranks;
    c->rank = rank;
    ncclResult_t r = g_api.CommInitRank(&c->comm, nranks, id, rank);  // collective over the ranks: blocks until all have called
    if (r != ncclSuccess) {
        set_error("ncclCommInitRank(nranks=%d, rank=%d) -> RCCL error %d: %s", nranks, rank, (int)r, g_api.GetErrorString(r));
        delete c;
        return CMAX_ECOMM;
    }
    *out = c;
    return 0;
}

void comm_destroy(Comm *c) {
    if (!c) return;
    // collectives of prepared calls may still be queued: ncclCommDestroy on a busy communicator is undefined
    if (c->used) (void)hipStreamSynchronize(c->last_stream);
    if (c->comm && g_api.ok) (void)g_api.CommDestroy(c->comm);
    delete c;
}

int comm_nranks(const Comm *c) { return c ? c->nranks : 1; }
int comm_rank(const Comm *c) { return c ? c->rank : 0; }

int comm_allreduce(Comm *c, void *buf, size_t count, CommType type, CommOp op, hipStream_t s) {
    if (!c || count == 0) return 0;  // a 1-rank communicator still goes through RCCL (world-1 tests run the real path)
    c->last_stream = s;
    c->used = true;
    CMAX_CHECK_RCCL(g_api.AllReduce(buf, buf, count, nccl_type(type), nccl_op(op), c->comm, s));
    return 0;
}

int comm_allreduce_group(Comm *c, void *const *bufs, const size_t *counts, const CommType *types, int n, CommOp op, hipStream_t s) {
    if (!c || n == 0) return 0;
    if (n == 1) return comm_allreduce(c, bufs[0], counts[0], types[0], op, s);
    c->last_stream = s;
    c->used = true;
    CMAX_CHECK_RCCL(g_api.GroupStart());
    for (int i = 0; i < n; ++i) {
        if (counts[i] == 0) continue;
        ncclResult_t r = g_api.AllReduce(bufs[i], bufs[i], counts[i], nccl_type(types[i]), nccl_op(op), c->comm, s);
        if (r != ncclSuccess) {
            (void)g_api.GroupEnd();
            set_error("ncclAllReduce inside a group -> RCCL error %d: %s", (int)r, g_api.GetErrorString(r));
            return CMAX_ECOMM;
        }
    }
    CMAX_CHECK_RCCL(g_api.GroupEnd());
    return 0;
}

}  // namespace cmax
