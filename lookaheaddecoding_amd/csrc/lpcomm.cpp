// Lookahead parallelism over RCCL / xGMI behind the C ABI: an opaque communicator handle with explicit
// create / destroy, and the one collective of a step - the all-gather of every rank's fixed int32 record.
//
// Replaces, for a caller that binds the C ABI without torch.distributed: the process-group init of lade/utils.py:28-33
// (dist.init_process_group + torch.cuda.set_device) and the four pickled object collectives of a step
// (lade/decoding.py:1024, :1057, :1090, :1096, :1106), which the record protocol folds into ONE ncclAllGather of
// rec_words int32 per rank (tens of bytes: latency bound, issued on the step's own stream so that it is ordered with
// lade_lp_pack before it and lade_lp_reduce_apply after it, and capturable with them).
//
// RCCL is bound at run time (dlopen by soname): a process that already carries an RCCL instance - torch ships one with the same
// soname - keeps exactly that one, and a build box without RCCL still builds and loads the library (the entry points then fail
// with LADE_E_LIMIT and a message).  The unique id travels between the ranks by whatever channel the host already has
// (the reference: the TCP store behind init_process_group); rank 0 obtains it from lade_lp_unique_id.
#include <dlfcn.h>
#include <string.h>

#include <mutex>
#include <string>

// types, enums and prototypes only: the entry points are resolved at run time (see above), nothing links librccl.  A build box
// without the RCCL headers still builds the library: the handful of declarations used here are then restated (NCCL's stable C API).
#include <hip/hip_runtime.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2 } ncclDataType_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId* uniqueId);
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId commId, int rank);
ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclCommCount(const ncclComm_t comm, int* count);
const char* ncclGetErrorString(ncclResult_t result);
}
#endif

#include "common.hpp"

namespace {

struct Rccl {
    void* h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
    std::string why;          // dlopen / dlsym failure text, captured once (dlerror() clears itself when read)
};

// bound once per process, thread-safe: lookahead-parallel ranks may be threads of one process (tests)
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);       // an instance the process already carries (torch's) wins
            if (r.h) break;
        }
        for (int i = 0; !r.h && i < 3; ++i) {
            r.h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
            if (!r.h) { const char* e = dlerror(); if (e) r.why = e; }
        }
        if (!r.h) { if (r.why.empty()) r.why = "librccl.so.1 not found"; return; }
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
        r.AllGather = (decltype(r.AllGather))dlsym(r.h, "ncclAllGather");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
        r.CommCount = (decltype(r.CommCount))dlsym(r.h, "ncclCommCount");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
        r.ok = r.GetUniqueId && r.CommInitRank && r.AllGather && r.CommDestroy && r.CommCount;
        if (!r.ok) r.why = "missing symbols in librccl";
    });
    return r;
}

struct LpComm {
    ncclComm_t comm;
    int rank, world;
};

const char* err_text(Rccl& r, ncclResult_t e) { return r.GetErrorString ? r.GetErrorString(e) : "rccl error"; }

}  // namespace

#define RCCL_OR_FAIL(what)                                                                                     \
    Rccl& R = rccl();                                                                                          \
    LADE_REQUIRE(R.ok, LADE_E_LIMIT, what ": RCCL (librccl.so.1) is not available in this process: %s", R.why.c_str())

extern "C" int lade_lp_unique_id(void* id128) {
    LADE_REQUIRE(id128, LADE_E_ARG, "lade_lp_unique_id: null buffer");
    RCCL_OR_FAIL("lade_lp_unique_id");
    ncclUniqueId id;
    const ncclResult_t e = R.GetUniqueId(&id);
    LADE_REQUIRE(e == ncclSuccess, LADE_E_LAUNCH, "lade_lp_unique_id: %s", err_text(R, e));
    memcpy(id128, &id, sizeof(id));
    return LADE_OK;
}

extern "C" int lade_lp_comm_create(const void* id128, int32_t rank, int32_t world, void** comm_out) {
    LADE_REQUIRE(id128 && comm_out && world >= 1 && rank >= 0 && rank < world, LADE_E_ARG, "lade_lp_comm_create: rank=%d world=%d", rank, world);
    RCCL_OR_FAIL("lade_lp_comm_create");
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    const ncclResult_t e = R.CommInitRank(&c, world, id, rank);      // binds to the calling thread's current HIP device (lade/utils.py:32)
    LADE_REQUIRE(e == ncclSuccess && c, LADE_E_LAUNCH, "lade_lp_comm_create: ncclCommInitRank: %s", err_text(R, e));
    // the communicator must span exactly the ranks the caller believes in: a mismatch (stale id, wrong world) fails here, loudly
    int n = -1;
    const ncclResult_t e2 = R.CommCount(c, &n);
    if (e2 != ncclSuccess || n != world) {
        (void)R.CommDestroy(c);
        LADE_REQUIRE(false, LADE_E_LAUNCH, "lade_lp_comm_create: communicator spans %d ranks, expected %d (%s)", n, world, err_text(R, e2));
    }
    *comm_out = new LpComm{c, rank, world};
    return LADE_OK;
}

extern "C" int lade_lp_allgather(void* comm, const int32_t* send, int32_t* recv, int32_t words_per_rank, void* stream) {
    LADE_REQUIRE(comm && send && recv && words_per_rank > 0, LADE_E_ARG, "lade_lp_allgather: bad args (words=%d)", words_per_rank);
    RCCL_OR_FAIL("lade_lp_allgather");
    LpComm* lc = (LpComm*)comm;
    const ncclResult_t e = R.AllGather(send, recv, (size_t)words_per_rank, ncclInt32, lc->comm, (hipStream_t)stream);
    LADE_REQUIRE(e == ncclSuccess, LADE_E_LAUNCH, "lade_lp_allgather: %s", err_text(R, e));
    return LADE_OK;
}

extern "C" int lade_lp_comm_count(void* comm, int32_t* ranks_out) {
    LADE_REQUIRE(comm && ranks_out, LADE_E_ARG, "lade_lp_comm_count: null argument");
    RCCL_OR_FAIL("lade_lp_comm_count");
    int n = 0;
    const ncclResult_t e = R.CommCount(((LpComm*)comm)->comm, &n);
    LADE_REQUIRE(e == ncclSuccess, LADE_E_LAUNCH, "lade_lp_comm_count: %s", err_text(R, e));
    *ranks_out = n;
    return LADE_OK;
}

extern "C" int lade_lp_comm_destroy(void* comm) {
    if (!comm) return LADE_OK;
    LpComm* lc = (LpComm*)comm;
    const ncclComm_t c = lc->comm;
    delete lc;                                       // the handle is gone whatever happens below
    RCCL_OR_FAIL("lade_lp_comm_destroy");
    const ncclResult_t e = R.CommDestroy(c);
    LADE_REQUIRE(e == ncclSuccess, LADE_E_LAUNCH, "lade_lp_comm_destroy: %s", err_text(R, e));
    return LADE_OK;
}
