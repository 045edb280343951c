/* Y = VL_NNBNORM(X, G, B, 'epsilon', E, 'moments', M);  [DX, DG, DB, MOMENTS] = VL_NNBNORM(X, G, B, DZDY, ...)
 * Drop-in for matlab/src/vl_nnbnorm.cu, backed by xm_nnbnorm_forward / xm_nnbnorm_backward. */
#include "xm_mex.h"

void mexFunction(int nout, mxArray *out[], int nin, mxArray const *in[]) {
  if (nin < 3) mexErrMsgIdAndTxt("XM:invalidArgument", "The arguments are less than three.");
  mxInitGPU();
  float eps = 1e-4f;
  XmTensor mom;
  int next = 3;
  bool backward = nin > 3 && !mxIsChar(in[3]);
  if (backward) next = 4;
  for (; next < nin; ++next) {
    if (xm_streq(in[next], "epsilon")) eps = (float)mxGetScalar(in[++next]);
    else if (xm_streq(in[next], "moments")) mom = xm_input(in[++next], "MOMENTS");
    else if (xm_streq(in[next], "cudnn") || xm_streq(in[next], "nocudnn") || xm_streq(in[next], "verbose")) {}
    else mexErrMsgIdAndTxt("XM:invalidArgument", "Unknown option.");
  }
  XmTensor x = xm_input(in[0], "X"), g = xm_input(in[1], "G"), b = xm_input(in[2], "B");
  const int H = x.d[0], W = x.d[1], Cc = x.d[2], N = x.d[3];
  mxGPUArray *k0 = nullptr, *k1 = nullptr, *k2 = nullptr, *k3 = nullptr;
  if (!backward) {
    float *y = xm_output(&out[0], &k0, H, W, Cc, N);
    float *mo = nout > 1 ? xm_output(&out[1], &k1, Cc, 2, 1, 1) : nullptr;
    xm_check(xm_nnbnorm_forward(x.ptr, H, W, Cc, N, g.ptr, b.ptr, eps, mom.empty ? nullptr : mom.ptr, y, mo,
                                nullptr));
  } else {
    XmTensor dz = xm_input(in[3], "DZDY");
    float *dx = xm_output(&out[0], &k0, H, W, Cc, N);
    float *dg = xm_output(&out[1], &k1, Cc, 1, 1, 1);
    float *db = xm_output(&out[2], &k2, Cc, 1, 1, 1);
    float *mo = nout > 3 ? xm_output(&out[3], &k3, Cc, 2, 1, 1) : nullptr;
    xm_check(xm_nnbnorm_backward(x.ptr, H, W, Cc, N, g.ptr, b.ptr, dz.ptr, eps,
                                 mom.empty ? nullptr : mom.ptr, dx, dg, db, mo, nullptr));
    if (dz.gpu) mxGPUDestroyGPUArray(dz.gpu);
  }
  for (mxGPUArray *k : {k0, k1, k2, k3}) if (k) mxGPUDestroyGPUArray(k);
  for (mxGPUArray const *k : {x.gpu, g.gpu, b.gpu, mom.gpu}) if (k) mxGPUDestroyGPUArray(k);
}
