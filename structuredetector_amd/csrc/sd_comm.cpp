// Gradient exchange: RCCL all-reduce behind the C ABI.  RCCL is resolved at run time (dlopen) so that the library loads on
// hosts without it and shares the instance PyTorch-ROCm already mapped (same soname, librccl.so.1).
#include <dlfcn.h>
#include <stdint.h>
#include <string.h>
#include <mutex>
#include "../../include/sdnet_hip.h"

namespace sd {
void set_error(const char* fmt, ...);

// the few RCCL entry points used, declared locally (rccl.h: ncclUniqueId is 128 opaque bytes, ncclFloat32 = 7, ncclSum = 0)
struct UniqueId { char internal[SD_COMM_ID_BYTES]; };
typedef int (*fn_get_id)(UniqueId*);
typedef int (*fn_init_rank)(void**, int, UniqueId, int);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, void*, void*);
typedef int (*fn_destroy)(void*);
typedef const char* (*fn_err)(int);

struct Rccl {
    void* handle = nullptr;
    fn_get_id get_id = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_destroy destroy = nullptr;
    fn_err err = nullptr;
};
static Rccl g_rccl;
static std::once_flag g_once;

static const Rccl* rccl() {
    std::call_once(g_once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void* h = nullptr;
        for (const char* n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h) return;
        Rccl r;
        r.handle = h;
        r.get_id = (fn_get_id)dlsym(h, "ncclGetUniqueId");
        r.init_rank = (fn_init_rank)dlsym(h, "ncclCommInitRank");
        r.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
        r.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
        r.err = (fn_err)dlsym(h, "ncclGetErrorString");
        if (r.get_id && r.init_rank && r.all_reduce && r.destroy && r.err) g_rccl = r;
    });
    if (!g_rccl.handle) {
        set_error("RCCL is not available: dlopen(librccl.so.1) failed or lacks the ncclAllReduce entry points");
        return nullptr;
    }
    return &g_rccl;
}

static int nccl_status(const Rccl* r, int rc, const char* what) {
    if (rc != 0) set_error("%s failed: %s", what, r->err(rc));
    return rc;
}
}  // namespace sd

extern "C" {

int sd_allreduce_unique_id(void* id_out) {
    if (!id_out) { sd::set_error("sd_allreduce_unique_id: null output"); return SD_ERR_INVALID; }
    const sd::Rccl* r = sd::rccl();
    if (!r) return SD_ERR_INVALID;
    sd::UniqueId id;
    int rc = sd::nccl_status(r, r->get_id(&id), "ncclGetUniqueId");
    if (rc == 0) memcpy(id_out, id.internal, SD_COMM_ID_BYTES);
    return rc;
}

int sd_allreduce_init(const void* id, int rank, int world, void** comm_out) {
    if (!id || !comm_out || world < 1 || rank < 0 || rank >= world) {
        sd::set_error("sd_allreduce_init: bad arguments (rank %d of %d)", rank, world);
        return SD_ERR_INVALID;
    }
    const sd::Rccl* r = sd::rccl();
    if (!r) return SD_ERR_INVALID;
    sd::UniqueId uid;
    memcpy(uid.internal, id, SD_COMM_ID_BYTES);
    void* comm = nullptr;
    int rc = sd::nccl_status(r, r->init_rank(&comm, world, uid, rank), "ncclCommInitRank");
    *comm_out = rc == 0 ? comm : nullptr;
    return rc;
}

int sd_allreduce_run(void* comm, float* buf, int64_t count, sd_stream_t stream) {
    if (!comm || (!buf && count > 0) || count < 0) { sd::set_error("sd_allreduce_run: bad arguments"); return SD_ERR_INVALID; }
    if (count == 0) return 0;
    const sd::Rccl* r = sd::rccl();
    if (!r) return SD_ERR_INVALID;
    return sd::nccl_status(r, r->all_reduce(buf, buf, (size_t)count, /*ncclFloat32*/ 7, /*ncclSum*/ 0, comm, stream), "ncclAllReduce");
}

int sd_allreduce_destroy(void* comm) {
    if (!comm) return 0;
    const sd::Rccl* r = sd::rccl();
    if (!r) return SD_ERR_INVALID;
    return sd::nccl_status(r, r->destroy(comm), "ncclCommDestroy");
}

}  // extern "C"
