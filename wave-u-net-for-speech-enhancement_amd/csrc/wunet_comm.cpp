// Flat-buffer RCCL all-reduce behind the C ABI (include/wunet_hip.h: wunet_comm_*): the data-parallel exchange step of the path
// (replaces what torch.nn.DataParallel does implicitly, /root/reference/trainer/base_trainer.py:26-27).
// One communicator per process and GPU (ncclCommInitRank on the calling thread's current device); the all-reduce is enqueued on the
// CALLER's stream - so it orders like any kernel of the library and can be captured into the same hipGraph as the step around it.
// RCCL is bound at run time (dlopen): the library still loads, and everything else works, on a box without it; the first
// wunet_comm_* call then fails with a message.  The handful of RCCL declarations needed are restated here (rccl.h, ROCm 7: the
// NCCL 2.x API) so that the CPU test build of this file needs no ROCm headers.
#include <dlfcn.h>

#include "wunet_host.h"

namespace {

struct UniqueId { char internal[WUNET_COMM_ID_BYTES]; };     // ncclUniqueId (NCCL_UNIQUE_ID_BYTES == 128)
typedef void* Comm;                                          // ncclComm_t
enum { kSuccess = 0, kFloat32 = 7, kSum = 0 };               // ncclSuccess, ncclFloat32, ncclSum

struct Api {
    void* lib = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;
};

Api& api()
{
    static Api a;
    static std::once_flag once;
    std::call_once(once, [] {
#ifdef WUNET_EMU
        a.why = "the CPU test build has no RCCL (world size 1 runs without it)";
#else
        // the copy the process already holds (torch's "nccl" backend IS this library on ROCm) before a second one
        const char* env = getenv("WUNET_RCCL_PATH");
        const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (int pass = 0; pass < 2 && !a.lib; ++pass)
            for (const char* n : names) {
                if (!n) continue;
                a.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
                if (a.lib) break;
            }
        if (!a.lib) {
            const char* e = dlerror();             // (a second dlerror() returns null: it clears the message)
            a.why = std::string("librccl.so not found: ") + (e ? e : "no loader message");
            return;
        }
        a.GetUniqueId = (int (*)(UniqueId*))dlsym(a.lib, "ncclGetUniqueId");
        a.CommInitRank = (int (*)(Comm*, int, UniqueId, int))dlsym(a.lib, "ncclCommInitRank");
        a.CommDestroy = (int (*)(Comm))dlsym(a.lib, "ncclCommDestroy");
        a.AllReduce = (int (*)(const void*, void*, size_t, int, int, Comm, hipStream_t))dlsym(a.lib, "ncclAllReduce");
        a.GetErrorString = (const char* (*)(int))dlsym(a.lib, "ncclGetErrorString");
        if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce) { a.why = "librccl.so lacks the NCCL 2.x entry points"; a.lib = nullptr; }
#endif
    });
    return a;
}

int rccl_fail(const char* what, int rc)
{
    Api& a = api();
    return wunet_host::fail(WUNET_E_RUNTIME, "%s failed: %s", what, a.GetErrorString ? a.GetErrorString(rc) : "RCCL error");
}

}  // namespace

struct wunet_comm {
    int world = 1, rank = 0;
    Comm comm = nullptr;        // nullptr at world size 1 without RCCL (the reduction over one rank is the identity)
};

using wunet_host::fail;

extern "C" {

int wunet_comm_unique_id(unsigned char* id)
{
    if (!id) return fail(WUNET_E_ARG, "null id");
    Api& a = api();
    if (!a.lib) { memset(id, 0, WUNET_COMM_ID_BYTES); return fail(WUNET_E_RUNTIME, "RCCL unavailable: %s", a.why.c_str()); }
    UniqueId u;
    const int rc = a.GetUniqueId(&u);
    if (rc != kSuccess) return rccl_fail("ncclGetUniqueId", rc);
    memcpy(id, u.internal, WUNET_COMM_ID_BYTES);
    return WUNET_OK;
}

int wunet_comm_create(const unsigned char* id, int world, int rank, wunet_comm** out)
{
    if (!out || world < 1 || rank < 0 || rank >= world) return fail(WUNET_E_ARG, "bad communicator arguments (world=%d rank=%d)", world, rank);
    Api& a = api();
    wunet_comm* c = new wunet_comm();
    c->world = world; c->rank = rank;
    if (!a.lib) {
        if (world == 1) { *out = c; return WUNET_OK; }
        delete c;
        return fail(WUNET_E_RUNTIME, "RCCL unavailable: %s", a.why.c_str());
    }
    if (!id) { delete c; return fail(WUNET_E_ARG, "null id"); }
    UniqueId u;
    memcpy(u.internal, id, WUNET_COMM_ID_BYTES);
    const int rc = a.CommInitRank(&c->comm, world, u, rank);
    if (rc != kSuccess) { delete c; return rccl_fail("ncclCommInitRank", rc); }
    *out = c;
    return WUNET_OK;
}

int wunet_comm_allreduce_sum(wunet_comm* c, float* buf, size_t count, void* stream)
{
    if (!c || (!buf && count)) return fail(WUNET_E_ARG, "null argument");
    if (!c->comm || count == 0) return WUNET_OK;            // one rank, no RCCL: nothing to add
    const int rc = api().AllReduce(buf, buf, count, kFloat32, kSum, c->comm, (hipStream_t)stream);
    if (rc != kSuccess) return rccl_fail("ncclAllReduce", rc);
    return WUNET_OK;
}

int wunet_comm_world(const wunet_comm* c) { return c ? c->world : 0; }

void wunet_comm_destroy(wunet_comm* c)
{
    if (!c) return;
    if (c->comm) api().CommDestroy(c->comm);
    delete c;
}

}  // extern "C"
