/*
 * xm_mex.h -- helpers shared by the MEX gateways that put libxmodal_hip.so behind MatConvNet's MATLAB operator
 * names (vl_nnconv, vl_nnpool, vl_nnbnorm, vl_nnrelu, vl_nnsigmoid, vl_nnsoftmaxt, vl_nnsoftmaxceloss, vl_nnloss).
 *
 * NOT BUILT BY THIS REPO: the image has no MATLAB.  tests/test_mex_sources.py only syntax-checks these files against
 * a declaration-only stand-in for mex.h (tests/mex_stub/), so they are a documented binding, not a tested one.
 *
 *   mex -I../include vl_nnconv.cpp -L../mcncrossmodalemotions_amd -lxmodal_hip      (no -lmwgpu: see below)
 *
 * MATLAB's gpuArray / mxGPUArray is CUDA-only; there is no device-array type on an MI355X host.  A tensor argument
 * is therefore one of
 *   (a) an `xmArray` object (mex/matlab/xmArray.m): an opaque handle {ptr: uint64 device address, sz: 1x4 double}
 *       whose storage came from xm_device_alloc.  Zero copy: the handle's address goes straight into the C ABI, and
 *       the outputs are handles too -- a dagnn network whose parameters / inputs were moved with xmArray(...) keeps
 *       every intermediate on the device, exactly like the gpuArray path of the reference;
 *   (b) a host `single` array.  Staged: uploaded into a temporary device buffer, outputs downloaded into fresh host
 *       arrays, temporaries freed at gateway exit.  Correct and convenient for tests, PCIe-bound for training.
 * Mixed calls return handles when ANY tensor input is a handle.
 *
 * Conventions mirrored from matlab/src/vl_nn*.cu of MatConvNet: positional tensors first, then 'name', value options
 * (case-insensitive); backward mode when DZDY is present; class must be single.
 */
#pragma once
#include <cctype>
#include <cstdint>
#include <cstring>
#include <vector>

#include "mex.h"
#include "xmodal.h"

inline void xm_check(int rc) {
  if (rc != XM_OK) mexErrMsgIdAndTxt("XM:error", "%s", xm_last_error());
}

/* Once per gateway (the state lives in libxmodal_hip.so, shared by all of them): every gateway issues its operator on
 * the null stream, one call after the other -- MatConvNet's own sequence (cnn_train_dag -> net.eval -> vl_nn*,
 * run_distillation.m:170-182) -- so the host declares XM_EXEC_SINGLE_STREAM (include/xmodal.h: kernels that are faster
 * alone but poor neighbours become candidates; an explicit statement, the library infers nothing from the call history). */
inline void xm_mex_startup() {
  static bool done = false;
  if (done) return;
  if (xm_version() < 105) mexErrMsgIdAndTxt("XM:version", "libxmodal_hip.so is older than ABI revision 105.");
  xm_check(xm_set_exec_hint(XM_EXEC_SINGLE_STREAM));
  done = true;
}

struct XmTensor {
  const float *ptr = nullptr;  // device address
  int d[4] = {1, 1, 1, 1};
  bool empty = true;
  bool handle = false;         // came in as an xmArray (not owned by the gateway)
  size_t numel() const { return (size_t)d[0] * d[1] * d[2] * d[3]; }
};

/* everything a gateway call allocates temporarily; freed by the destructor, also when mexErrMsgIdAndTxt long-jumps
 * out (MATLAB runs C++ destructors of the gateway frame only on normal return, so errors are raised AFTER cleanup:
 * see XmCall::fail) */
struct XmCall {
  std::vector<void *> temps;
  bool any_handle = false;

  XmCall() { xm_mex_startup(); }
  ~XmCall() { release(); }
  void release() {
    for (void *p : temps) xm_device_free(p);
    temps.clear();
  }
  [[noreturn]] void fail(const char *id, const char *msg) {
    release();
    mexErrMsgIdAndTxt(id, "%s", msg);
    throw 0;  // not reached
  }
  void check(int rc) {
    if (rc != XM_OK) fail("XM:error", xm_last_error());
  }

  XmTensor input(const mxArray *a, const char *name) {
    XmTensor t;
    if (a == nullptr || mxIsEmpty(a)) return t;
    if (mxIsClass(a, "xmArray")) {
      mxArray *p = mxGetProperty(a, 0, "ptr"), *s = mxGetProperty(a, 0, "sz");
      if (!p || !s) fail("XM:badHandle", "xmArray without ptr / sz properties.");
      t.ptr = (const float *)(uintptr_t)(*(const uint64_t *)mxGetData(p));
      const double *sz = mxGetPr(s);
      size_t n = mxGetNumberOfElements(s);
      if (n > 4) fail("XM:tooManyDims", "tensor has more than 4 dimensions.");
      for (size_t i = 0; i < n; ++i) t.d[i] = (int)sz[i];
      t.handle = true;
      t.empty = t.numel() == 0;
      any_handle = true;
      return t;
    }
    if (!mxIsSingle(a)) {
      char msg[128];
      snprintf(msg, sizeof msg, "%s must be of class SINGLE (or an xmArray handle).", name);
      fail("XM:needSingle", msg);
    }
    mwSize nd = mxGetNumberOfDimensions(a);
    if (nd > 4) fail("XM:tooManyDims", "tensor has more than 4 dimensions.");
    const mwSize *dims = mxGetDimensions(a);
    for (mwSize i = 0; i < nd; ++i) t.d[i] = (int)dims[i];
    void *dev = nullptr;
    check(xm_device_alloc(&dev, t.numel() * sizeof(float)));
    temps.push_back(dev);
    check(xm_device_upload(dev, mxGetData(a), t.numel() * sizeof(float)));
    t.ptr = (const float *)dev;
    t.empty = false;
    return t;
  }

  /* device buffer for an output of size h x w x c x n */
  struct Out {
    float *ptr = nullptr;
    int d[4] = {0, 0, 0, 0};
  };
  Out output(int h, int w, int c, int n) {
    Out o;
    o.d[0] = h, o.d[1] = w, o.d[2] = c, o.d[3] = n;
    void *dev = nullptr;
    check(xm_device_alloc(&dev, (size_t)h * w * c * n * sizeof(float)));
    temps.push_back(dev);
    o.ptr = (float *)dev;
    return o;
  }

  /* hand an output to MATLAB: a handle (ownership moves to the xmArray object, which frees it in its delete
   * method) when any input was a handle, else a host array filled by a download */
  mxArray *deliver(const Out &o) {
    mwSize dims[4] = {(mwSize)o.d[0], (mwSize)o.d[1], (mwSize)o.d[2], (mwSize)o.d[3]};
    if (any_handle) {
      mxArray *args[2];
      args[0] = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
      *(uint64_t *)mxGetData(args[0]) = (uint64_t)(uintptr_t)o.ptr;
      args[1] = mxCreateDoubleMatrix(1, 4, mxREAL);
      for (int i = 0; i < 4; ++i) mxGetPr(args[1])[i] = (double)o.d[i];
      mxArray *obj = nullptr;
      for (size_t i = 0; i < temps.size(); ++i)
        if (temps[i] == (void *)o.ptr) temps.erase(temps.begin() + i--);   // no longer a temporary
      if (mexCallMATLAB(1, &obj, 2, args, "xmArray") != 0) fail("XM:handle", "xmArray constructor failed.");
      return obj;
    }
    mxArray *host = mxCreateNumericArray(4, dims, mxSINGLE_CLASS, mxREAL);
    check(xm_device_download(mxGetData(host), o.ptr, (size_t)o.d[0] * o.d[1] * o.d[2] * o.d[3] * sizeof(float)));
    return host;
  }
};

/* 'stride' / 'pad' / 'dilate' value -> up to 4 ints, MatConvNet broadcasting rules */
inline void xm_intvec(XmCall &call, const mxArray *v, int *out, int want, const char *name) {
  size_t n = mxGetNumberOfElements(v);
  const double *p = mxGetPr(v);
  if (n == 1) {
    for (int i = 0; i < want; ++i) out[i] = (int)p[0];
  } else if (want == 4 && n == 2) {
    out[0] = out[1] = (int)p[0];
    out[2] = out[3] = (int)p[1];
  } else if ((int)n == want) {
    for (int i = 0; i < want; ++i) out[i] = (int)p[i];
  } else {
    char msg[96];
    snprintf(msg, sizeof msg, "%s has the wrong number of elements.", name);
    call.fail("XM:invalidArgument", msg);
  }
}

inline bool xm_streq(const mxArray *a, const char *s) {
  char buf[64];
  if (!mxIsChar(a) || mxGetString(a, buf, sizeof buf)) return false;
  for (char *c = buf; *c; ++c) *c = (char)tolower(*c);
  return strcmp(buf, s) == 0;
}
/* options MatConvNet accepts and this backend has no use for */
inline bool xm_ignored_option(const mxArray *a) {
  return xm_streq(a, "cudnn") || xm_streq(a, "nocudnn") || xm_streq(a, "verbose");
}
