// The flush of SURVEY.md section 8e behind the C-ABI: ONE ncclAllReduce(ncclDouble, ncclSum) of the accumulated correlation
// sums over the ranks of a node (RCCL over xGMI), for consumers that have no torch.distributed.
//
// librccl is bound at run time (dlopen at the first sc_comm_* call): the engine library has no link-time dependency on it
// and single-GPU users never load it.  Host code only -- the reduction kernels are RCCL's.
#include "sc_common.h"
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>          // types and enumerators only; every function is resolved with dlsym

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    char error[256] = "";
};

RcclApi g_rccl;

template <class F>
bool resolve(F &fn, const char *name) {
    fn = reinterpret_cast<F>(dlsym(g_rccl.handle, name));
    if (!fn) snprintf(g_rccl.error, sizeof(g_rccl.error), "librccl has no symbol %s", name);
    return fn != nullptr;
}

// 0 when the library and every entry point are there (idempotent; the first failure is remembered in g_rccl.error)
int load_rccl() {
    if (g_rccl.AllReduce) return SC_OK;
    if (!g_rccl.handle) {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};       // no environment knobs
        for (const char *n : names) {
            g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (g_rccl.handle) break;
        }
        if (!g_rccl.handle) {
            snprintf(g_rccl.error, sizeof(g_rccl.error), "cannot load librccl (%s)", dlerror());
            return sc_fail(SC_ERR_UNSUPPORTED, "%s", g_rccl.error);
        }
    }
    const bool ok = resolve(g_rccl.GetUniqueId, "ncclGetUniqueId") && resolve(g_rccl.CommInitRank, "ncclCommInitRank")
                    && resolve(g_rccl.CommDestroy, "ncclCommDestroy") && resolve(g_rccl.CommCount, "ncclCommCount")
                    && resolve(g_rccl.CommUserRank, "ncclCommUserRank") && resolve(g_rccl.GetErrorString, "ncclGetErrorString")
                    && resolve(g_rccl.GetVersion, "ncclGetVersion") && resolve(g_rccl.AllReduce, "ncclAllReduce");
    if (!ok) {
        g_rccl.AllReduce = nullptr;
        return sc_fail(SC_ERR_UNSUPPORTED, "%s", g_rccl.error);
    }
    return SC_OK;
}

int rccl_fail(const char *what, ncclResult_t r) {
    return sc_fail(SC_ERR_LAUNCH, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
}

}  // namespace

extern "C" int sc_comm_available(void) {
    if (load_rccl() != SC_OK) return 0;
    int version = 0;
    return g_rccl.GetVersion(&version) == ncclSuccess ? version : 0;
}

extern "C" int sc_comm_unique_id(void *id_out) {
    if (!id_out) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_comm_unique_id: NULL destination");
    if (int rc = load_rccl()) return rc;
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
    static_assert(sizeof(id) == SC_COMM_ID_BYTES, "ncclUniqueId is not 128 bytes");
    memcpy(id_out, &id, sizeof(id));
    return SC_OK;
}

extern "C" int sc_comm_init(const void *id, int32_t nranks, int32_t rank, void **comm_out) {
    if (!id || !comm_out || nranks < 1 || rank < 0 || rank >= nranks)
        return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_comm_init: id/comm_out NULL or rank %d outside [0, %d)", rank, nranks);
    if (int rc = load_rccl()) return rc;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm = nullptr;
    const ncclResult_t r = g_rccl.CommInitRank(&comm, nranks, uid, rank);      // collective: every rank calls it
    if (r != ncclSuccess) return rccl_fail("ncclCommInitRank", r);
    *comm_out = comm;
    return SC_OK;
}

extern "C" int sc_comm_destroy(void *comm) {
    if (!comm) return SC_OK;
    if (int rc = load_rccl()) return rc;
    const ncclResult_t r = g_rccl.CommDestroy(static_cast<ncclComm_t>(comm));
    return r == ncclSuccess ? SC_OK : rccl_fail("ncclCommDestroy", r);
}

extern "C" int sc_comm_rank_count(void *comm, int32_t *rank_out, int32_t *nranks_out) {
    if (!comm) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_comm_rank_count: NULL communicator");
    if (int rc = load_rccl()) return rc;
    int rank = 0, count = 0;
    ncclResult_t r = g_rccl.CommUserRank(static_cast<ncclComm_t>(comm), &rank);
    if (r == ncclSuccess) r = g_rccl.CommCount(static_cast<ncclComm_t>(comm), &count);
    if (r != ncclSuccess) return rccl_fail("ncclCommUserRank/Count", r);
    if (rank_out) *rank_out = rank;
    if (nranks_out) *nranks_out = count;
    return SC_OK;
}

extern "C" int sc_flush_allreduce(double *sums, int64_t count, void *comm, void *stream) {
    if (count < 0 || (count > 0 && !sums)) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_flush_allreduce: bad buffer");
    if (!comm) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_flush_allreduce: NULL communicator (sc_comm_init first)");
    if (count == 0) return SC_OK;
    if (int rc = load_rccl()) return rc;
    const ncclResult_t r = g_rccl.AllReduce(sums, sums, (size_t)count, ncclDouble, ncclSum, static_cast<ncclComm_t>(comm),
                                            static_cast<hipStream_t>(stream));
    return r == ncclSuccess ? SC_OK : rccl_fail("ncclAllReduce", r);
}
