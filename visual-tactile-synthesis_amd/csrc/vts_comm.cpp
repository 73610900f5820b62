// Optional collective of the data-parallel path: the sum-all-reduce of one flat fp32 gradient bucket as an explicit REDUCE-SCATTER +
// ALL-GATHER on the library's own RCCL communicator and side stream (on 8 fully connected MI355X every rank reduces
// 1/8 of the bucket and the seven xGMI links of a GPU carry one slice each, instead of whatever algorithm ncclAllReduce picks for a
// <= 8.5 MB message).  Replaces what the reference gets from nn.DataParallel's reduce_add_coalesced
// (/root/reference/models/base_model.py:104-108).  Off by default (vts/ddp.py: VTS_DDP_DIRECT=1): no multi-GPU node was available to
// compare it with torch.distributed's all_reduce; the one-rank form runs on the test box.
//
// RCCL is loaded with dlopen (first the copy already in the process -- PyTorch's --, then the system one), so libvts_hip.so itself has
// no link-time dependency on it and single-GPU users never touch it.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include "vts_internal.h"

namespace {

typedef struct { char internal[128]; } ncclUniqueId_;
typedef void* ncclComm_;
enum { ncclFloat_ = 7, ncclSum_ = 0 };

struct Rccl {
  void* h = nullptr;
  int (*GetUniqueId)(ncclUniqueId_*) = nullptr;
  int (*CommInitRank)(ncclComm_*, int, ncclUniqueId_, int) = nullptr;
  int (*CommDestroy)(ncclComm_) = nullptr;
  int (*ReduceScatter)(const void*, void*, size_t, int, int, ncclComm_, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r.h ? &r : nullptr;
  tried = true;
  const char* names[] = {getenv("VTS_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    if (!n) continue;
    r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);      // the copy torch.distributed already loaded, if any
    if (r.h) break;
  }
  for (const char* n : names) {
    if (r.h) break;
    if (n) r.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
  }
  if (!r.h) return nullptr;
#define SYM(field, name) *(void**)(&r.field) = dlsym(r.h, name)
  SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
  SYM(ReduceScatter, "ncclReduceScatter"); SYM(AllGather, "ncclAllGather"); SYM(AllReduce, "ncclAllReduce");
  SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.ReduceScatter || !r.AllGather || !r.AllReduce) {
    r.h = nullptr;
    return nullptr;
  }
  return &r;
}

struct Comm {
  ncclComm_ comm;
  int rank, world;
  hipStream_t side;
  hipEvent_t ready, done;
};

#define RCCL_CHECK(expr, what)                                                                                    \
  do {                                                                                                            \
    const int rc__ = (expr);                                                                                      \
    if (rc__ != 0) {                                                                                              \
      vts_set_error("%s: RCCL error %d (%s)", what, rc__, R->GetErrorString ? R->GetErrorString(rc__) : "?");   \
      return VTS_ERR_LAUNCH;                                                                                      \
    }                                                                                                             \
  } while (0)

}  // namespace

extern "C" int vts_comm_unique_id(void* id128) {
  VTS_CHECK_ARG(id128, "vts_comm_unique_id: null pointer");
  Rccl* R = rccl();
  VTS_CHECK_ARG(R, "vts_comm_unique_id: librccl.so could not be loaded (set VTS_RCCL_LIB)");
  ncclUniqueId_ id;
  RCCL_CHECK(R->GetUniqueId(&id), "ncclGetUniqueId");
  memcpy(id128, &id, sizeof id);
  return VTS_OK;
}

extern "C" int vts_comm_init(const void* id128, int rank, int world, void** comm) {
  VTS_CHECK_ARG(id128 && comm && world >= 1 && rank >= 0 && rank < world, "vts_comm_init: bad arguments (rank %d of %d)", rank, world);
  Rccl* R = rccl();
  VTS_CHECK_ARG(R, "vts_comm_init: librccl.so could not be loaded (set VTS_RCCL_LIB)");
  Comm* c = (Comm*)calloc(1, sizeof(Comm));
  VTS_CHECK_ARG(c, "vts_comm_init: out of host memory");
  ncclUniqueId_ id;
  memcpy(&id, id128, sizeof id);
  if (R->CommInitRank(&c->comm, world, id, rank) != 0) {
    free(c);
    vts_set_error("vts_comm_init: ncclCommInitRank failed (rank %d of %d)", rank, world);
    return VTS_ERR_LAUNCH;
  }
  c->rank = rank;
  c->world = world;
  if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->ready) (void)hipEventDestroy(c->ready);
    if (c->side) (void)hipStreamDestroy(c->side);
    (void)R->CommDestroy(c->comm);
    free(c);
    vts_set_error("vts_comm_init: could not create the side stream / events");
    return VTS_ERR_LAUNCH;
  }
  *comm = c;
  return VTS_OK;
}

// Which elements of an n-element bucket rank `rank` of `world` reduces: its slice [offset, offset + chunk) of the reduce-scatter and the
// [tail_offset, tail_offset + tail) remainder every rank all-reduces.  Host arithmetic only (no GPU, no RCCL): vts_allreduce_flat_async
// takes its offsets from here, and the CPU tests replay the same plan over gloo at world sizes the test box has no GPUs for.
extern "C" int vts_allreduce_slice_plan(int64_t n, int world, int rank, int64_t* offset, int64_t* chunk, int64_t* tail_offset, int64_t* tail) {
  VTS_CHECK_ARG(offset && chunk && tail_offset && tail, "vts_allreduce_slice_plan: null pointer");
  VTS_CHECK_ARG(n >= 1 && world >= 1 && rank >= 0 && rank < world, "vts_allreduce_slice_plan: bad arguments (n %lld, rank %d of %d)", (long long)n, rank, world);
  *chunk = n / world;
  *offset = (int64_t)rank * *chunk;
  *tail_offset = *chunk * world;
  *tail = n - *tail_offset;
  return VTS_OK;
}

// buf[0 .. n) <- sum over ranks, in place, on the communicator's side stream: the side stream first waits for everything enqueued on
// `producer_stream` so far (the backward that filled the bucket).  Rank r reduces slice r of `chunk = n / world` elements
// (ncclReduceScatter in place: recvbuff = sendbuff + r * chunk), all slices are gathered back (ncclAllGather in place); the
// n - chunk * world tail elements, if any, go through a small ncclAllReduce.
extern "C" int vts_allreduce_flat_async(void* comm, float* buf, int64_t n, void* producer_stream) {
  VTS_CHECK_ARG(comm && buf && n >= 1, "vts_allreduce_flat_async: bad arguments");
  Rccl* R = rccl();
  Comm* c = (Comm*)comm;
  if (hipEventRecord(c->ready, (hipStream_t)producer_stream) != hipSuccess || hipStreamWaitEvent(c->side, c->ready, 0) != hipSuccess) {
    vts_set_error("vts_allreduce_flat_async: stream ordering failed");
    return VTS_ERR_LAUNCH;
  }
  int64_t off = 0, chunk = 0, tail_off = 0, tail = 0;
  if (vts_allreduce_slice_plan(n, c->world, c->rank, &off, &chunk, &tail_off, &tail) != VTS_OK) return VTS_ERR_ARG;
  if (chunk > 0) {
    RCCL_CHECK(R->ReduceScatter(buf, buf + off, (size_t)chunk, ncclFloat_, ncclSum_, c->comm, c->side), "ncclReduceScatter");
    RCCL_CHECK(R->AllGather(buf + off, buf, (size_t)chunk, ncclFloat_, c->comm, c->side), "ncclAllGather");
  }
  if (tail > 0) RCCL_CHECK(R->AllReduce(buf + tail_off, buf + tail_off, (size_t)tail, ncclFloat_, ncclSum_, c->comm, c->side), "ncclAllReduce (tail)");
  if (hipEventRecord(c->done, c->side) != hipSuccess) {
    vts_set_error("vts_allreduce_flat_async: event record failed");
    return VTS_ERR_LAUNCH;
  }
  return VTS_OK;
}

// makes `consumer_stream` (the stream the optimiser step is enqueued on) wait for the last vts_allreduce_flat_async of this communicator
extern "C" int vts_allreduce_flat_wait(void* comm, void* consumer_stream) {
  VTS_CHECK_ARG(comm, "vts_allreduce_flat_wait: null communicator");
  Comm* c = (Comm*)comm;
  if (hipStreamWaitEvent((hipStream_t)consumer_stream, c->done, 0) != hipSuccess) {
    vts_set_error("vts_allreduce_flat_wait: stream wait failed");
    return VTS_ERR_LAUNCH;
  }
  return VTS_OK;
}

extern "C" int vts_comm_destroy(void* comm) {
  if (!comm) return VTS_OK;
  Rccl* R = rccl();
  Comm* c = (Comm*)comm;
  (void)hipStreamSynchronize(c->side);
  if (R) (void)R->CommDestroy(c->comm);
  (void)hipEventDestroy(c->ready);
  (void)hipEventDestroy(c->done);
  (void)hipStreamDestroy(c->side);
  free(c);
  return VTS_OK;
}
