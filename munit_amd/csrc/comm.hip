// Data-parallel exchange for hosts that do not bring a communicator of their own (SURVEY.md section 8b lists
// munit_comm_{init,allreduce,destroy} among the boundary's entry points): a thin layer over RCCL, resolved at RUN time with
// dlopen -- the library has no link-time dependency on librccl, and inside a PyTorch process dlopen("librccl.so.1") returns the
// copy torch has already loaded (same soname), so no second RCCL enters the process.  The Python host of this repository does
// NOT use these entry points: its exchange is torch.distributed's process group (the launch contract), see DESIGN.md section 6.
#include "common.h"
#include <dlfcn.h>
#include <atomic>
#include <cstring>
#include <mutex>

namespace {
struct UniqueId { char internal[128]; };                 // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128), passed by value
typedef int (*get_unique_id_t)(UniqueId*);
typedef int (*comm_init_rank_t)(void**, int, UniqueId, int);
typedef int (*all_reduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*comm_destroy_t)(void*);
typedef const char* (*get_error_string_t)(int);
constexpr int NCCL_FLOAT32 = 7, NCCL_SUM = 0;           // ncclDataType_t / ncclRedOp_t values of rccl.h

struct Rccl {
  void* handle = nullptr;
  get_unique_id_t get_unique_id = nullptr;
  comm_init_rank_t comm_init_rank = nullptr;
  all_reduce_t all_reduce = nullptr;
  comm_destroy_t comm_destroy = nullptr;
  get_error_string_t error_string = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mutex;                 // first use from two host threads, and shutdown against a first use
std::atomic<int> g_live_comms{0};        // communicators created through this library and not yet destroyed

// Resolves librccl once; callers hold no lock afterwards (the table is written before `handle` is published and never changes
// while a communicator lives: munit_shutdown refuses to unload it then).
bool load_rccl() {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (g_rccl.handle != nullptr) return true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (h != nullptr) break;
  }
  if (h == nullptr) {
    munit_set_error("munit_comm: librccl not found (%s)", dlerror());
    return false;
  }
  Rccl r;
  r.handle = h;
  r.get_unique_id = (get_unique_id_t)dlsym(h, "ncclGetUniqueId");
  r.comm_init_rank = (comm_init_rank_t)dlsym(h, "ncclCommInitRank");
  r.all_reduce = (all_reduce_t)dlsym(h, "ncclAllReduce");
  r.comm_destroy = (comm_destroy_t)dlsym(h, "ncclCommDestroy");
  r.error_string = (get_error_string_t)dlsym(h, "ncclGetErrorString");
  if (!r.get_unique_id || !r.comm_init_rank || !r.all_reduce || !r.comm_destroy) {
    munit_set_error("munit_comm: librccl lacks an expected symbol");
    dlclose(h);
    return false;
  }
  g_rccl = r;
  return true;
}

int check(int rc, const char* what) {
  if (rc == 0) return MUNIT_OK;
  if (g_rccl.error_string != nullptr) munit_set_error("munit_comm: %s failed: %s", what, g_rccl.error_string(rc));
  else munit_set_error("munit_comm: %s failed: ncclResult_t %d (this librccl exports no ncclGetErrorString)", what, rc);
  return MUNIT_ERR_LAUNCH;
}
}  // namespace

extern "C" int munit_comm_unique_id(void* id_out, size_t bytes) {
  MUNIT_CHECK_ARG(id_out != nullptr && bytes >= sizeof(UniqueId), "comm_unique_id: need a 128-byte buffer");
  if (!load_rccl()) return MUNIT_ERR_LAUNCH;
  UniqueId id;
  const int rc = check(g_rccl.get_unique_id(&id), "ncclGetUniqueId");
  if (rc == MUNIT_OK) memcpy(id_out, &id, sizeof(id));
  return rc;
}

extern "C" int munit_comm_init(munit_comm_t* comm, int rank, int world, const void* unique_id) {
  MUNIT_CHECK_ARG(comm != nullptr && unique_id != nullptr && world >= 1 && rank >= 0 && rank < world, "comm_init: bad arguments");
  if (!load_rccl()) return MUNIT_ERR_LAUNCH;
  UniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  void* c = nullptr;
  const int rc = check(g_rccl.comm_init_rank(&c, world, id, rank), "ncclCommInitRank");
  if (rc == MUNIT_OK) {
    *comm = c;
    g_live_comms.fetch_add(1);
  }
  return rc;
}

extern "C" int munit_comm_allreduce(munit_comm_t comm, float* buf, size_t count, munit_stream_t stream) {
  MUNIT_CHECK_ARG(comm != nullptr && (buf != nullptr || count == 0), "comm_allreduce: null argument");
  if (!load_rccl()) return MUNIT_ERR_LAUNCH;
  if (count == 0) return MUNIT_OK;
  return check(g_rccl.all_reduce(buf, buf, count, NCCL_FLOAT32, NCCL_SUM, comm, (hipStream_t)stream), "ncclAllReduce");
}

extern "C" int munit_comm_destroy(munit_comm_t comm) {
  if (comm == nullptr) return MUNIT_OK;
  if (!load_rccl()) return MUNIT_ERR_LAUNCH;
  const int rc = check(g_rccl.comm_destroy(comm), "ncclCommDestroy");
  if (rc == MUNIT_OK) g_live_comms.fetch_sub(1);
  return rc;
}

// Frees what the library keeps between calls: the RCCL handle.  Refused (MUNIT_ERR_ARG, nothing unloaded) while a
// communicator created by munit_comm_init is still alive: its owner destroys it first.
// HIP modules, the cached stream-wait events and the thread-local error string are process-lifetime state of the runtime.
extern "C" int munit_shutdown(void) {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (g_live_comms.load() > 0) {
    munit_set_error("munit_shutdown: %d communicator(s) still alive; call munit_comm_destroy first", g_live_comms.load());
    return MUNIT_ERR_ARG;
  }
  if (g_rccl.handle != nullptr) {
    dlclose(g_rccl.handle);
    g_rccl = Rccl{};
  }
  return MUNIT_OK;
}
