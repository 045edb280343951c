// ParameterServer replacement: sum-all-reduce of gradient buffers over RCCL / xGMI.
// The reference selects MatConvNet's ParameterServer with method 'tmove'
// (emoVoxCeleb/run_distillation.m:88,181): per-parameter push / sync / pull between one MATLAB
// worker per GPU.  Here: one process per GPU, one ncclAllReduce(sum, fp32) per flat bucket.
#include <rccl/rccl.h>

#include "xm_common.h"

static ncclComm_t g_comm = nullptr;
static int g_world = 1;

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
  g_world = world;
  if (world == 1) return XM_OK;  // single worker: ParameterServer is bypassed (numel(gpus) == 1)
  if (!id128) return xm::fail(XM_EINVAL, "comm: NULL id");
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  XM_NCCL(ncclCommInitRank(&g_comm, world, id, rank));
  return XM_OK;
}

int xm_allreduce_sum_f32(float *buf, size_t n, void *stream) {
  if (g_world == 1 || n == 0) return XM_OK;
  if (!g_comm) return xm::fail(XM_EINVAL, "comm: xm_comm_init has not been called");
  if (!buf) return xm::fail(XM_EINVAL, "comm: NULL buffer");
  XM_NCCL(ncclAllReduce(buf, buf, n, ncclFloat, ncclSum, g_comm, (hipStream_t)stream));
  return XM_OK;
}

int xm_comm_destroy(void) {
  if (g_comm) {
    XM_NCCL(ncclCommDestroy(g_comm));
    g_comm = nullptr;
  }
  g_world = 1;
  return XM_OK;
}
}
