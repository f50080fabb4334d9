// Native RCCL communicator: direct ncclAllReduce / ncclReduceScatter / ncclAllGather / ncclBroadcast calls on a HIP
// stream WE own, without going through torch.distributed's process group (no work objects, no Python per collective
// beyond one ctypes call, capturable into a hipGraph together with the kernels around it).
//
// RCCL is resolved at run time with dlopen (the process must use ONE RCCL: the copy torch ships is already mapped, so
// the Python side passes that path), hence no link-time dependency and a CPU box can still load this library.
// The reference's only collective is a blocking per-parameter all_reduce through c10d
// (/root/reference/src/pytorch/CNN/main.py:134-140).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

namespace {

// The few RCCL entry points we use; enum values are those of rccl.h (stable NCCL ABI).
typedef struct { char internal[128]; } UniqueId;
typedef void* Comm;
enum { kSum = 0, kAvg = 4 };
enum { kFloat32 = 7, kBfloat16 = 9 };

struct Api {
  void* handle = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*ReduceScatter)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*GetVersion)(int*) = nullptr;
  int (*CommCount)(Comm, int*) = nullptr;
  int (*CommUserRank)(Comm, int*) = nullptr;
  int (*CommAbort)(Comm) = nullptr;
  bool ok() const { return handle != nullptr; }
} api;

template <typename F> bool sym(void* h, const char* name, F& out) {
  out = reinterpret_cast<F>(dlsym(h, name));
  return out != nullptr;
}

int dtype_of(int code) { return code == 1 ? kBfloat16 : kFloat32; }
int op_of(int code) { return code == 1 ? kAvg : kSum; }

}  // namespace

extern "C" {

// Map RCCL.  `path` may be null (then "librccl.so" from the loader path).  0 on success.
int ddl_comm_load(const char* path) {
  if (api.ok()) return 0;
  void* h = dlopen(path != nullptr ? path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (h == nullptr) return -1;
  Api a;
  a.handle = h;
  const bool all = sym(h, "ncclGetUniqueId", a.GetUniqueId) && sym(h, "ncclCommInitRank", a.CommInitRank) &&
                   sym(h, "ncclCommDestroy", a.CommDestroy) && sym(h, "ncclAllReduce", a.AllReduce) &&
                   sym(h, "ncclReduceScatter", a.ReduceScatter) && sym(h, "ncclAllGather", a.AllGather) &&
                   sym(h, "ncclBroadcast", a.Broadcast) && sym(h, "ncclGetVersion", a.GetVersion) &&
                   sym(h, "ncclCommCount", a.CommCount) && sym(h, "ncclCommUserRank", a.CommUserRank) &&
                   sym(h, "ncclCommAbort", a.CommAbort);
  if (!all) return -2;
  api = a;
  return 0;
}

int ddl_comm_version() {
  int v = 0;
  if (!api.ok() || api.GetVersion(&v) != 0) return -1;
  return v;
}

// 128-byte rendezvous token, created on rank 0 and handed to every rank out of band (TCP store).
int ddl_comm_unique_id(void* out128) {
  if (!api.ok() || out128 == nullptr) return -1;
  UniqueId id;
  const int rc = api.GetUniqueId(&id);
  if (rc != 0) return rc;
  memcpy(out128, id.internal, sizeof(id.internal));
  return 0;
}

// Collective over all ranks: the calling thread's current HIP device becomes this rank's device.
int ddl_comm_init(const void* id128, int rank, int world, void** comm_out) {
  if (!api.ok() || id128 == nullptr || comm_out == nullptr || world < 1 || rank < 0 || rank >= world) return -1;
  UniqueId id;
  memcpy(id.internal, id128, sizeof(id.internal));
  Comm c = nullptr;
  const int rc = api.CommInitRank(&c, world, id, rank);
  if (rc != 0) return rc;
  *comm_out = c;
  return 0;
}

int ddl_comm_destroy(void* comm) {
  if (!api.ok() || comm == nullptr) return -1;
  return api.CommDestroy((Comm)comm);
}

// Tear a communicator down without waiting for outstanding collectives (a rank that gives up on a hung job).
int ddl_comm_abort(void* comm) {
  if (!api.ok() || comm == nullptr) return -1;
  return api.CommAbort((Comm)comm);
}

// What the communicator itself reports: number of ranks (ncclCommCount) and this rank's index; negative on error.
int ddl_comm_count(void* comm) {
  int n = 0;
  if (!api.ok() || comm == nullptr || api.CommCount((Comm)comm, &n) != 0) return -1;
  return n;
}

int ddl_comm_user_rank(void* comm) {
  int r = 0;
  if (!api.ok() || comm == nullptr || api.CommUserRank((Comm)comm, &r) != 0) return -1;
  return r;
}

// dtype: 0 = fp32, 1 = bf16.  op: 0 = sum, 1 = average.  In place when send == recv.
int ddl_comm_all_reduce(void* comm, const void* send, void* recv, long long count, int dtype, int op, void* stream) {
  if (!api.ok() || comm == nullptr || send == nullptr || recv == nullptr || count <= 0) return -1;
  return api.AllReduce(send, recv, (size_t)count, dtype_of(dtype), op_of(op), (Comm)comm, (hipStream_t)stream);
}

// recv gets this rank's `recv_count` elements of the reduction of `send` (world * recv_count elements).
int ddl_comm_reduce_scatter(void* comm, const void* send, void* recv, long long recv_count, int dtype, int op,
                            void* stream) {
  if (!api.ok() || comm == nullptr || send == nullptr || recv == nullptr || recv_count <= 0) return -1;
  return api.ReduceScatter(send, recv, (size_t)recv_count, dtype_of(dtype), op_of(op), (Comm)comm, (hipStream_t)stream);
}

// recv (world * send_count elements) gets every rank's `send` in rank order; in place when send == recv + rank * n.
int ddl_comm_all_gather(void* comm, const void* send, void* recv, long long send_count, int dtype, void* stream) {
  if (!api.ok() || comm == nullptr || send == nullptr || recv == nullptr || send_count <= 0) return -1;
  return api.AllGather(send, recv, (size_t)send_count, dtype_of(dtype), (Comm)comm, (hipStream_t)stream);
}

int ddl_comm_broadcast(void* comm, void* buf, long long count, int dtype, int root, void* stream) {
  if (!api.ok() || comm == nullptr || buf == nullptr || count <= 0) return -1;
  return api.Broadcast(buf, buf, (size_t)count, dtype_of(dtype), root, (Comm)comm, (hipStream_t)stream);
}

}  // extern "C"
