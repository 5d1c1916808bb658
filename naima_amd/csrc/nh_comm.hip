// nh_comm.hip -- the one collective of the path: an all-gather of the walkers a
// rank has just moved, straight on RCCL over xGMI.  librccl is
// dlopen()ed on first use so that single-GPU users never load it.
#include <dlfcn.h>

#include <strings.h>

#include <cstdlib>
#include <string>
#include <rccl/rccl.h>

#include "nh_common.h"

static_assert(NH_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

struct rccl_api {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t,
                            hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
};

static rccl_api g_rccl;

static int rccl_load() {
  if (g_rccl.lib) return NH_OK;
  // NCCL_DEBUG=VERSION makes RCCL write a banner to STDOUT (where a caller may owe someone
  // exactly one line of output); real debug levels are left alone
  if (const char* dbg = getenv("NCCL_DEBUG"))
    if (strcasecmp(dbg, "VERSION") == 0) unsetenv("NCCL_DEBUG");
  // The RCCL that belongs to the HIP runtime THIS library is linked against, by absolute
  // path.  A process that has imported PyTorch (torch.distributed hands the unique id
  // around) already holds PyTorch's own bundled librccl.so.1 -- bound to PyTorch's own
  // bundled HIP runtime, a second one in the process -- and a dlopen by soname would
  // return that copy, whose ncclCommInitRank then fails on our streams ("unhandled cuda
  // error").  RTLD_DEEPBIND keeps the two copies' symbols apart.
  std::string first = "/opt/rocm/lib/librccl.so.1";
  if (const char* rp = getenv("ROCM_PATH")) first = std::string(rp) + "/lib/librccl.so.1";
  const char* names[] = {first.c_str(), "/opt/rocm/lib/librccl.so.1", "librccl.so.1",
                         "librccl.so"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
    if (h) break;
  }
  if (!h) return nh_set_error(NH_ECOMM, "cannot dlopen librccl: %s", dlerror());
#define NH_SYM(field, sym)                                                           \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, sym));            \
  if (!g_rccl.field) return nh_set_error(NH_ECOMM, "librccl lacks symbol %s", sym);
  NH_SYM(GetUniqueId, "ncclGetUniqueId")
  NH_SYM(CommInitRank, "ncclCommInitRank")
  NH_SYM(CommDestroy, "ncclCommDestroy")
  NH_SYM(AllGather, "ncclAllGather")
  NH_SYM(GetErrorString, "ncclGetErrorString")
  NH_SYM(CommCount, "ncclCommCount")
  NH_SYM(CommUserRank, "ncclCommUserRank")
  NH_SYM(CommCuDevice, "ncclCommCuDevice")
#undef NH_SYM
  g_rccl.lib = h;
  return NH_OK;
}

#define NH_CHECK_RCCL(expr)                                                               \
  do {                                                                                    \
    ncclResult_t _r = (expr);                                                             \
    if (_r != ncclSuccess)                                                                \
      return nh_set_error(NH_ECOMM, "%s failed: %s", #expr, g_rccl.GetErrorString(_r));   \
  } while (0)

// librccl loaded and every symbol resolved?  No communicator, no collective: what every rank
// checks (and agrees on over the control plane) BEFORE anybody enters ncclCommInitRank, which
// blocks until all ranks have called it
extern "C" int nh_comm_available(void) { return rccl_load(); }

extern "C" int nh_comm_unique_id(char* id_out) {
  NH_REQUIRE(id_out, "id_out is NULL");
  int rc = rccl_load();
  if (rc) return rc;
  ncclUniqueId id;
  NH_CHECK_RCCL(g_rccl.GetUniqueId(&id));
  memcpy(id_out, id.internal, NH_UNIQUE_ID_BYTES);
  return NH_OK;
}

extern "C" int nh_comm_init(nh_ctx* c, int rank, int nranks, const char* idbytes) {
  NH_REQUIRE(c && idbytes && nranks >= 1 && rank >= 0 && rank < nranks, "bad argument");
  NH_REQUIRE(c->comm == nullptr, "communicator already initialised");
  int rc = rccl_load();
  if (rc) return rc;
  NH_CHECK_HIP(hipSetDevice(c->device));
  ncclUniqueId id;
  memcpy(id.internal, idbytes, NH_UNIQUE_ID_BYTES);
  ncclComm_t comm;
  NH_CHECK_RCCL(g_rccl.CommInitRank(&comm, nranks, id, rank));
  c->comm = comm;
  return NH_OK;
}

// What the LIVE communicator says of itself -- ncclCommCount / ncclCommUserRank / ncclCommCuDevice
// -- not what the caller believes it asked for: bench.py prints these (`rccl_nranks`), so that the
// day RCCL sees fewer ranks than the launcher started the line shows it.
extern "C" int nh_comm_info(nh_ctx* c, int* nranks, int* rank, int* device) {
  NH_REQUIRE(c && nranks && rank && device, "bad argument");
  NH_REQUIRE(c->comm != nullptr, "nh_comm_init has not been called");
  ncclComm_t comm = reinterpret_cast<ncclComm_t>(c->comm);
  NH_CHECK_RCCL(g_rccl.CommCount(comm, nranks));
  NH_CHECK_RCCL(g_rccl.CommUserRank(comm, rank));
  NH_CHECK_RCCL(g_rccl.CommCuDevice(comm, device));
  return NH_OK;
}

extern "C" int nh_comm_destroy(nh_ctx* c) {
  if (c && c->comm) {
    g_rccl.CommDestroy(reinterpret_cast<ncclComm_t>(c->comm));
    c->comm = nullptr;
  }
  return NH_OK;
}

extern "C" int nh_comm_allgather(nh_ctx* c, const double* send, double* recv, long long count) {
  NH_REQUIRE(c && send && recv && count >= 0, "bad argument");
  NH_REQUIRE(c->comm != nullptr, "nh_comm_init has not been called");
  nh_prof_scope ps(c, NH_K_GLUE);
  NH_CHECK_RCCL(g_rccl.AllGather(send, recv, (size_t)count, ncclDouble,
                                 reinterpret_cast<ncclComm_t>(c->comm), c->stream));
  return NH_OK;
}
