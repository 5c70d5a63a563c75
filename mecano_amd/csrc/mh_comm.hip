// mh_comm.hip -- the multi-GPU entry points of include/mecano_hip.h: one process per GPU, the batch sharded by rows, RCCL over xGMI.
//
// Every configuration is independent and the model is read-only, so the data path has no collective at all; what a
// host needs is (1) the slice of the batch a rank owns, (2) one broadcast of the robot description from the rank that has it, and
// (3) optionally the ranks' output rows side by side on every rank.  mecano_amd/distributed.py does this over torch.distributed for the
// Python host; these are the same three operations for a host that has no torch (the Java shim: HipCommunicator.java).
//
// librccl.so.1 is opened on the first call: the library carries no link-time dependency on RCCL and loads on a box without it -- and no
// build-time dependency either: the handful of RCCL types and constants the eleven entry points need are declared below (they are part of
// RCCL's / NCCL's stable C ABI: ncclUniqueId is 128 opaque bytes, ncclSuccess 0, ncclInt8 0, ncclInt32 2, ncclSum 0), so the library builds
// on a box without the RCCL headers.
#include "mecano_hip.h"
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

extern "C" mh_status mh_internal_fail(mh_status code, const char *message); // mh_api.hip: sets mh_last_error() of the calling thread

namespace
{
// ---- the slice of RCCL's C ABI this file calls (rccl.h / nccl.h), declared locally
typedef struct ncclComm *ncclComm_t;
constexpr int NCCL_UNIQUE_ID_BYTES = 128;
typedef struct
{
   char internal[NCCL_UNIQUE_ID_BYTES];
} ncclUniqueId;
typedef int ncclResult_t;   // enum ncclResult_t: ncclSuccess = 0
typedef int ncclDataType_t; // enum ncclDataType_t
typedef int ncclRedOp_t;    // enum ncclRedOp_t
constexpr ncclResult_t ncclSuccess = 0;
constexpr ncclDataType_t ncclChar = 0, ncclInt32 = 2;
constexpr ncclRedOp_t ncclSum = 0;
struct Rccl
{
   void *handle = nullptr;
   ncclResult_t (*get_unique_id)(ncclUniqueId *) = nullptr;
   ncclResult_t (*comm_init_rank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
   ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
   ncclResult_t (*comm_count)(const ncclComm_t, int *) = nullptr;
   ncclResult_t (*comm_user_rank)(const ncclComm_t, int *) = nullptr;
   ncclResult_t (*broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
   ncclResult_t (*all_gather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
   ncclResult_t (*all_reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
   ncclResult_t (*group_start)(void) = nullptr;
   ncclResult_t (*group_end)(void) = nullptr;
   const char *(*error_string)(ncclResult_t) = nullptr;
   bool ok = false;
   char why[256] = "";
};
Rccl g_rccl;
std::once_flag g_rccl_once;

const Rccl &rccl()
{
   std::call_once(g_rccl_once, [] {
      Rccl &r = g_rccl;
      const char *names[] = {getenv("MH_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
      for (const char *name : names)
         if (name && *name && (r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL)))
            break;
      if (!r.handle)
      {
         snprintf(r.why, sizeof r.why, "librccl.so.1 could not be opened (%s)", dlerror());
         return;
      }
#define MH_RCCL_SYM(field, symbol)                                                  \
   r.field = (decltype(r.field))dlsym(r.handle, symbol);                            \
   if (!r.field)                                                                    \
   {                                                                                \
      snprintf(r.why, sizeof r.why, "librccl.so.1 has no symbol %s", symbol);       \
      return;                                                                       \
   }
      MH_RCCL_SYM(get_unique_id, "ncclGetUniqueId")
      MH_RCCL_SYM(comm_init_rank, "ncclCommInitRank")
      MH_RCCL_SYM(comm_destroy, "ncclCommDestroy")
      MH_RCCL_SYM(comm_count, "ncclCommCount")
      MH_RCCL_SYM(comm_user_rank, "ncclCommUserRank")
      MH_RCCL_SYM(broadcast, "ncclBroadcast")
      MH_RCCL_SYM(all_gather, "ncclAllGather")
      MH_RCCL_SYM(all_reduce, "ncclAllReduce")
      MH_RCCL_SYM(group_start, "ncclGroupStart")
      MH_RCCL_SYM(group_end, "ncclGroupEnd")
      MH_RCCL_SYM(error_string, "ncclGetErrorString")
#undef MH_RCCL_SYM
      r.ok = true;
   });
   return g_rccl;
}

mh_status failf(mh_status code, const char *fmt, const char *a, const char *b = "")
{
   char msg[512];
   snprintf(msg, sizeof msg, fmt, a, b);
   return mh_internal_fail(code, msg);
}
#define RCCL_READY()                                                     \
   const Rccl &R = rccl();                                               \
   if (!R.ok)                                                            \
      return failf(MH_ERR_NO_DEVICE, "RCCL is not available: %s", R.why)
#define RCCL_TRY(expr)                                                   \
   do                                                                    \
   {                                                                     \
      const ncclResult_t r_ = (expr);                                    \
      if (r_ != ncclSuccess)                                             \
         return failf(MH_ERR_HIP, "%s: %s", #expr, R.error_string(r_));  \
   } while (0)
#define HIPC_TRY(expr)                                                   \
   do                                                                    \
   {                                                                     \
      const hipError_t e_ = (expr);                                      \
      if (e_ != hipSuccess)                                              \
         return failf(e_ == hipErrorOutOfMemory ? MH_ERR_OUT_OF_MEMORY : MH_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
   } while (0)
} // namespace

struct mh_comm
{
   ncclComm_t comm = nullptr;
   int rank = 0, world = 1;
   void *scratch = nullptr; // device staging of mh_comm_broadcast_host, the word mh_comm_barrier reduces
   size_t scratch_bytes = 0;
};

namespace
{
mh_status ensure_scratch(mh_comm *c, size_t bytes)
{
   if (c->scratch_bytes >= bytes)
      return MH_OK;
   if (c->scratch)
      (void)hipFree(c->scratch);
   c->scratch = nullptr, c->scratch_bytes = 0;
   HIPC_TRY(hipMalloc(&c->scratch, bytes));
   c->scratch_bytes = bytes;
   return MH_OK;
}
} // namespace

extern "C" {

// rows [lo, hi) of a batch of B owned by `rank`: contiguous, sizes differ by at most one (mecano_amd/distributed.py: shard_range)
mh_status mh_shard_range(int64_t B, int32_t rank, int32_t world, int64_t *lo_out, int64_t *hi_out)
{
   if (B < 0 || world < 1 || rank < 0 || rank >= world || !lo_out || !hi_out)
      return mh_internal_fail(MH_ERR_INVALID_ARGUMENT, "mh_shard_range: need B >= 0, 0 <= rank < world and two output pointers");
   const int64_t base = B / world, rem = B % world;
   const int64_t lo = rank * base + (rank < rem ? rank : rem);
   *lo_out = lo, *hi_out = lo + base + (rank < rem ? 1 : 0);
   return MH_OK;
}

mh_status mh_comm_unique_id(void *id_out)
{
   static_assert(MH_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the header's id size is RCCL's");
   if (!id_out)
      return mh_internal_fail(MH_ERR_INVALID_ARGUMENT, "mh_comm_unique_id: NULL output");
   RCCL_READY();
   ncclUniqueId id;
   RCCL_TRY(R.get_unique_id(&id));
   memcpy(id_out, id.internal, MH_COMM_ID_BYTES);
   return MH_OK;
}

mh_status mh_comm_create(const void *id_in, int32_t rank, int32_t world, mh_comm_t *comm_out)
{
   if (!id_in || !comm_out || world < 1 || rank < 0 || rank >= world)
      return mh_internal_fail(MH_ERR_INVALID_ARGUMENT, "mh_comm_create: need an id, 0 <= rank < world and an output pointer");
   *comm_out = nullptr;
   RCCL_READY();
   ncclUniqueId id;
   memcpy(id.internal, id_in, MH_COMM_ID_BYTES);
   mh_comm *c = new mh_comm;
   const ncclResult_t r = R.comm_init_rank(&c->comm, world, id, rank); // on the calling thread's current device
   if (r != ncclSuccess)
   {
      delete c;
      return failf(MH_ERR_HIP, "ncclCommInitRank: %s", R.error_string(r));
   }
   // what the communicator itself counted, not what the caller said
   if (R.comm_count(c->comm, &c->world) != ncclSuccess || R.comm_user_rank(c->comm, &c->rank) != ncclSuccess || c->world != world || c->rank != rank)
   {
      (void)R.comm_destroy(c->comm);
      delete c;
      return mh_internal_fail(MH_ERR_HIP, "mh_comm_create: the communicator reports another rank / size than it was created with");
   }
   *comm_out = c;
   return MH_OK;
}

mh_status mh_comm_destroy(mh_comm_t c)
{
   if (!c)
      return MH_OK;
   const Rccl &R = rccl();
   if (c->scratch)
      (void)hipFree(c->scratch);
   const ncclResult_t r = R.ok && c->comm ? R.comm_destroy(c->comm) : ncclSuccess;
   delete c;
   return r == ncclSuccess ? MH_OK : failf(MH_ERR_HIP, "ncclCommDestroy: %s", R.error_string(r));
}

mh_status mh_comm_size(mh_comm_t c, int32_t *rank_out, int32_t *world_out)
{
   if (!c)
      return mh_internal_fail(MH_ERR_INVALID_ARGUMENT, "mh_comm_size: NULL communicator");
   if (rank_out)
      *rank_out = c->rank;
   if (world_out)
      *world_out = c->world;
   return MH_OK;
}

mh_status mh_comm_broadcast(mh_comm_t c, void *device_buf, size_t bytes, int32_t root, void *stream)
{
   if (!c || (!device_buf && bytes) || root < 0 || root >= c->world)
      return mh_internal_fail(MH_ERR_INVALID_ARGUMENT, "mh_comm_broadcast: NULL communicator / buffer, or root outside the communicator");
   if (bytes == 0)
      return MH_OK;
   RCCL_READY();
   RCCL_TRY(R.broadcast(device_buf, device_buf, bytes, ncclChar, root, c->comm, (hipStream_t)stream));
   return MH_OK;
}

// host buffer in, host buffer out (the robot description: a few KB): staged through device scratch, returns when the bytes are there
mh_status mh_comm_broadcast_host(mh_comm_t c, void *host_buf, size_t bytes, int32_t root)
{
   if (!c || (!host_buf && bytes) || root < 0 || root >= c->world)
      return mh_internal_fail(MH_ERR_INVALID_ARGUMENT, "mh_comm_broadcast_host: NULL communicator / buffer, or root outside the communicator");
   if (bytes == 0)
      return MH_OK;
   RCCL_READY();
   const mh_status st = ensure_scratch(c, bytes);
   if (st != MH_OK)
      return st;
   if (c->rank == root)
      HIPC_TRY(hipMemcpy(c->scratch, host_buf, bytes, hipMemcpyHostToDevice));
   RCCL_TRY(R.broadcast(c->scratch, c->scratch, bytes, ncclChar, root, c->comm, (hipStream_t) nullptr));
   HIPC_TRY(hipStreamSynchronize(nullptr));
   if (c->rank != root)
      HIPC_TRY(hipMemcpy(host_buf, c->scratch, bytes, hipMemcpyDeviceToHost));
   return MH_OK;
}

// The operations mh_comm_all_gather_rows issues on rank `rank` of `world`, without issuing them (host-only; no RCCL, no device): step k is
// a broadcast of steps[k].bytes bytes from rank steps[k].root into [recv_offset, recv_offset + bytes) of every rank's output; on the root
// the bytes come from its local rows (send_local = 1: send buffer != receive buffer), everywhere else the operation is in place.  Equal
// shards without forcing: ONE step with root = -1, the plain all-gather (bytes = one shard).  Every rank computes the same list (same
// roots, sizes and offsets in the same order) -- which is what a grouped collective needs, and what tests/test_distributed_cpu.py checks
// for every world size up to nine without any transport.
mh_status mh_comm_gather_plan(int64_t B_total, size_t row_bytes, int32_t rank, int32_t world, int32_t force_ragged, mh_gather_step *steps,
                              int32_t cap, int32_t *n_steps_out)
{
   if (B_total < 0 || world < 1 || rank < 0 || rank >= world || !n_steps_out || (cap > 0 && !steps))
      return mh_internal_fail(MH_ERR_INVALID_ARGUMENT, "mh_comm_gather_plan: need B_total >= 0, 0 <= rank < world and an output count");
   int n = 0;
   auto put = [&](int root, int send_local, int64_t ofs, int64_t bytes) {
      if (n < cap)
         steps[n] = mh_gather_step{root, send_local, ofs, bytes};
      n++;
   };
   if (B_total > 0 && row_bytes > 0)
   {
      if (B_total % world == 0 && !force_ragged)
         put(-1, 1, (int64_t)rank * (B_total / world) * (int64_t)row_bytes, (B_total / world) * (int64_t)row_bytes);
      else
         for (int r = 0; r < world; r++)
         {
            int64_t rl = 0, rh = 0;
            (void)mh_shard_range(B_total, r, world, &rl, &rh);
            if (rh > rl) // (more ranks than rows: nothing to send, and every rank skips the same r)
               put(r, r == rank ? 1 : 0, rl * (int64_t)row_bytes, (rh - rl) * (int64_t)row_bytes);
         }
   }
   *n_steps_out = n;
   return MH_OK;
}

// every rank passes the rows mh_shard_range gives it ([hi - lo][row_bytes], device) and receives all B_total rows in batch order.
// Equal shards: one all-gather.  Ragged shards (sizes differ by one): one broadcast per rank inside a group -- still one fused operation,
// no padding and no compaction pass.  The list of operations is mh_comm_gather_plan's.
mh_status mh_comm_all_gather_rows(mh_comm_t c, const void *local_rows, int64_t B_total, size_t row_bytes, void *all_rows_out, void *stream)
{
   if (!c || B_total < 0 || (!all_rows_out && B_total && row_bytes))
      return mh_internal_fail(MH_ERR_INVALID_ARGUMENT, "mh_comm_all_gather_rows: NULL communicator / output, or a negative batch size");
   if (B_total == 0 || row_bytes == 0)
      return MH_OK;
   int64_t lo = 0, hi = 0;
   (void)mh_shard_range(B_total, c->rank, c->world, &lo, &hi);
   if (hi > lo && !local_rows)
      return mh_internal_fail(MH_ERR_INVALID_ARGUMENT, "mh_comm_all_gather_rows: NULL local rows");
   RCCL_READY();
   char *const out = (char *)all_rows_out;
   const char *force = getenv("MH_COMM_RAGGED"); // =1: the grouped-broadcast path for equal shards too (tests on one rank)
   std::vector<mh_gather_step> steps((size_t)c->world + 1);
   int32_t n = 0;
   if (const mh_status st = mh_comm_gather_plan(B_total, row_bytes, c->rank, c->world, force && atoi(force) ? 1 : 0, steps.data(), (int32_t)steps.size(), &n); st != MH_OK)
      return st;
   if (n == 1 && steps[0].root < 0)
   {
      RCCL_TRY(R.all_gather(local_rows, out, (size_t)steps[0].bytes, ncclChar, c->comm, (hipStream_t)stream));
      return MH_OK;
   }
   RCCL_TRY(R.group_start());
   for (int k = 0; k < n; k++)
   {
      void *recv = out + steps[k].recv_offset;
      const ncclResult_t e = R.broadcast(steps[k].send_local ? local_rows : recv, recv, (size_t)steps[k].bytes, ncclChar, steps[k].root, c->comm, (hipStream_t)stream);
      if (e != ncclSuccess)
      { // an error at this point is an argument error, and every rank computed the same arguments: they all leave the group here
         (void)R.group_end();
         return failf(MH_ERR_HIP, "ncclBroadcast (ragged all-gather): %s", R.error_string(e));
      }
   }
   RCCL_TRY(R.group_end());
   return MH_OK;
}

// returns when every rank has reached it and the work queued on `stream` before it has finished (bench-style region brackets)
mh_status mh_comm_barrier(mh_comm_t c, void *stream)
{
   if (!c)
      return mh_internal_fail(MH_ERR_INVALID_ARGUMENT, "mh_comm_barrier: NULL communicator");
   RCCL_READY();
   const mh_status st = ensure_scratch(c, sizeof(int));
   if (st != MH_OK)
      return st;
   HIPC_TRY(hipMemsetAsync(c->scratch, 0, sizeof(int), (hipStream_t)stream));
   RCCL_TRY(R.all_reduce(c->scratch, c->scratch, 1, ncclInt32, ncclSum, c->comm, (hipStream_t)stream));
   HIPC_TRY(hipStreamSynchronize((hipStream_t)stream));
   return MH_OK;
}

} // extern "C"
