/*
 * xm_mex.h -- helpers shared by the MEX gateways that put libxmodal_hip.so behind MatConvNet's
 * MATLAB operator names.  NOT COMPILED IN THIS REPO'S CI: the build image has no MATLAB (no mex.h,
 * no mxGPUArray); the sources are the reference-side binding a maintainer adds.
 *
 *   mex -I../include vl_nnconv.cpp -L../mcncrossmodalemotions_amd -lxmodal_hip -lmwgpu
 *
 * Conventions mirrored from matlab/src/vl_nn*.cu of MatConvNet: positional tensors first, then
 * 'name', value options (case-insensitive); backward mode when DZDY is present; gpuArray single
 * inputs only (this library has no CPU path -- a CPU array raises the same error MatConvNet
 * raises for an unsupported class).
 */
#pragma once
#include <cstring>
#include <string>
#include <vector>

#include "gpu/mxGPUArray.h"
#include "mex.h"
#include "xmodal.h"

struct XmTensor {
  mxGPUArray const *gpu = nullptr;
  const float *ptr = nullptr;
  int d[4] = {1, 1, 1, 1};
  bool empty = true;
};

inline XmTensor xm_input(const mxArray *a, const char *name) {
  XmTensor t;
  if (mxIsEmpty(a)) return t;
  if (!mxIsGPUArray(a))
    mexErrMsgIdAndTxt("XM:needGpuArray", "%s must be a gpuArray (this build has no CPU path).", name);
  t.gpu = mxGPUCreateFromMxArray(a);
  if (mxGPUGetClassID(t.gpu) != mxSINGLE_CLASS)
    mexErrMsgIdAndTxt("XM:needSingle", "%s must be of class SINGLE.", name);
  mwSize nd = mxGPUGetNumberOfDimensions(t.gpu);
  if (nd > 4) mexErrMsgIdAndTxt("XM:tooManyDims", "%s has more than 4 dimensions.", name);
  const mwSize *dims = mxGPUGetDimensions(t.gpu);
  for (mwSize i = 0; i < nd; ++i) t.d[i] = (int)dims[i];
  t.ptr = (const float *)mxGPUGetDataReadOnly(t.gpu);
  t.empty = false;
  return t;
}

inline float *xm_output(mxArray **out, mxGPUArray **keep, int h, int w, int c, int n) {
  mwSize dims[4] = {(mwSize)h, (mwSize)w, (mwSize)c, (mwSize)n};
  *keep = mxGPUCreateGPUArray(4, dims, mxSINGLE_CLASS, mxREAL, MX_GPU_DO_NOT_INITIALIZE);
  *out = mxGPUCreateMxArrayOnGPU(*keep);
  return (float *)mxGPUGetData(*keep);
}

inline void xm_check(int rc) {
  if (rc != XM_OK) mexErrMsgIdAndTxt("XM:error", "%s", xm_last_error());
}

/* 'stride' / 'pad' / 'dilate' value -> up to 4 ints, MatConvNet broadcasting rules */
inline void xm_intvec(const mxArray *v, int *out, int want, const char *name) {
  size_t n = mxGetNumberOfElements(v);
  const double *p = mxGetPr(v);
  if (n == 1) for (int i = 0; i < want; ++i) out[i] = (int)p[0];
  else if (want == 4 && n == 2) { out[0] = out[1] = (int)p[0]; out[2] = out[3] = (int)p[1]; }
  else if ((int)n == want) for (int i = 0; i < want; ++i) out[i] = (int)p[i];
  else mexErrMsgIdAndTxt("XM:invalidArgument", "%s has the wrong number of elements.", name);
}

inline bool xm_streq(const mxArray *a, const char *s) {
  char buf[64];
  if (!mxIsChar(a) || mxGetString(a, buf, sizeof buf)) return false;
  for (char *c = buf; *c; ++c) *c = (char)tolower(*c);
  return strcmp(buf, s) == 0;
}
