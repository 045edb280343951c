// ParameterServer replacement: sum-all-reduce of gradient buffers over RCCL / xGMI.
// The reference selects MatConvNet's ParameterServer with method 'tmove'
// (emoVoxCeleb/run_distillation.m:88,181): per-parameter push / sync / pull between one MATLAB
// worker per GPU.  Here: one process per GPU, one ncclAllReduce(sum, fp32) per flat bucket.
#include <rccl/rccl.h>

#include "xm_common.h"

#include <cstdlib>
#include <vector>

static ncclComm_t g_comm = nullptr;
static int g_world = 1;
static int g_inited = 0;                   // xm_comm_init succeeded (a real single worker has no communicator)
static int g_force_single = 0;             // debugging: a real 1-rank communicator (xm_debug_comm_force_single)
// ParameterServer push/sync: the exchange runs on the communicator's OWN stream so that a bucket's all-reduce
// overlaps whatever the producer stream does next (the rest of the backward pass)
static hipStream_t g_ps_stream = nullptr;
static std::vector<hipEvent_t> g_ps_events;
static size_t g_ps_next = 0;
static int g_ps_pending = 0;

static hipEvent_t ps_event() {
  if (g_ps_next == g_ps_events.size()) {
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    g_ps_events.push_back(e);
  }
  return g_ps_events[g_ps_next++];
}

#define XM_NCCL(expr)                                                                   \
  do {                                                                                  \
    ncclResult_t r__ = (expr);                                                          \
    if (r__ != ncclSuccess)                                                             \
      return xm::fail(XM_EHIP, "%s -> %s", #expr, ncclGetErrorString(r__));             \
  } while (0)

extern "C" {

int xm_comm_unique_id(void *id128) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  if (!id128) return xm::fail(XM_EINVAL, "comm: NULL id buffer");
  ncclUniqueId id;
  XM_NCCL(ncclGetUniqueId(&id));
  memcpy(id128, &id, sizeof id);
  return XM_OK;
}

int xm_comm_init(const void *id128, int rank, int world) {
  if (g_comm) return xm::fail(XM_EINVAL, "comm: already initialised");
  if (world < 1 || rank < 0 || rank >= world) return xm::fail(XM_EINVAL, "comm: bad rank/world");
  if (world == 1 && !g_force_single) {  // single worker: ParameterServer is bypassed (numel(gpus) == 1)
    g_world = 1;
    g_inited = 1;
    return XM_OK;
  }
  if (!id128) return xm::fail(XM_EINVAL, "comm: NULL id");
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  XM_NCCL(ncclCommInitRank(&g_comm, world, id, rank));
  g_world = world;
  g_inited = 1;
  return XM_OK;
}

}  // extern "C"
// test switch behind xm_debug_set("comm_single", v): xm_comm_init with world 1 creates a real one-rank communicator
namespace xm {
int comm_force_single(int on) {
  const int old = g_force_single ? 1 : 0;
  g_force_single = on != 0;
  return old;
}
}  // namespace xm
extern "C" {

int xm_comm_count(int *ranks) {
  if (!ranks) return xm::fail(XM_EINVAL, "comm: NULL output");
  *ranks = 1;
  if (g_comm) XM_NCCL(ncclCommCount(g_comm, ranks));
  return XM_OK;
}

// ParameterServer.push: start the sum of `buf` over all workers as soon as everything already enqueued on
// `producer_stream` has finished; returns at once (the exchange runs on the communicator's stream).
// An exchange without a communicator is a no-op ONLY after a successful single-worker xm_comm_init; a host that
// skipped the init (or whose init failed) must not train on silently un-exchanged derivatives.
static int comm_missing(const char *what) {
  if (g_comm) return 0;
  if (g_inited && g_world == 1) return 1;   // real single worker
  xm::fail(XM_EINVAL, "%s: no communicator (xm_comm_init was not called or failed)", what);
  return -1;
}

int xm_parserv_push(float *buf, size_t n, void *producer_stream) {
  if (int m = comm_missing("parserv push")) return m > 0 ? XM_OK : XM_EINVAL;
  if (n == 0) return XM_OK;
  if (!buf) return xm::fail(XM_EINVAL, "parserv: NULL buffer");
  if (!g_ps_stream) XM_HIP(hipStreamCreateWithFlags(&g_ps_stream, hipStreamNonBlocking));
  hipEvent_t e = ps_event();
  if (!e) return xm::fail(XM_EHIP, "parserv: event creation failed");
  XM_HIP(hipEventRecord(e, (hipStream_t)producer_stream));
  XM_HIP(hipStreamWaitEvent(g_ps_stream, e, 0));
  XM_NCCL(ncclAllReduce(buf, buf, n, ncclFloat, ncclSum, g_comm, g_ps_stream));
  ++g_ps_pending;
  return XM_OK;
}

// ParameterServer.sync + pull: `consumer_stream` waits for every push issued since the last sync; the summed
// values are then in place.  Does not block the host.
int xm_parserv_sync(void *consumer_stream) {
  if (int m = comm_missing("parserv sync")) {
    g_ps_next = 0;
    return m > 0 ? XM_OK : XM_EINVAL;
  }
  if (!g_ps_pending) {
    g_ps_next = 0;
    return XM_OK;
  }
  hipEvent_t e = ps_event();
  if (!e) return xm::fail(XM_EHIP, "parserv: event creation failed");
  XM_HIP(hipEventRecord(e, g_ps_stream));
  XM_HIP(hipStreamWaitEvent((hipStream_t)consumer_stream, e, 0));
  g_ps_pending = 0;
  g_ps_next = 0;   // events are re-recorded next step; stream order keeps earlier waits valid
  return XM_OK;
}

int xm_allreduce_sum_f32(float *buf, size_t n, void *stream) {
  if (int m = comm_missing("allreduce")) return m > 0 ? XM_OK : XM_EINVAL;
  if (n == 0) return XM_OK;
  if (!buf) return xm::fail(XM_EINVAL, "comm: NULL buffer");
  XM_NCCL(ncclAllReduce(buf, buf, n, ncclFloat, ncclSum, g_comm, (hipStream_t)stream));
  return XM_OK;
}

int xm_comm_destroy(void) {
  if (g_ps_stream) {
    (void)hipStreamSynchronize(g_ps_stream);
    (void)hipStreamDestroy(g_ps_stream);
    g_ps_stream = nullptr;
  }
  for (hipEvent_t e : g_ps_events) (void)hipEventDestroy(e);
  g_ps_events.clear();
  g_ps_next = 0;
  g_ps_pending = 0;
  if (g_comm) {
    XM_NCCL(ncclCommDestroy(g_comm));
    g_comm = nullptr;
  }
  g_world = 1;
  g_inited = 0;
  return XM_OK;
}
}
