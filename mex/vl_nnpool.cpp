/* Y = VL_NNPOOL(X, POOL, 'stride', S, 'pad', P, 'method', 'max'|'avg');  DX = VL_NNPOOL(X, POOL, DZDY, ...)
 * Drop-in for matlab/src/vl_nnpool.cu, backed by xm_nnpool_forward / xm_nnpool_backward. */
#include "xm_mex.h"

void mexFunction(int nout, mxArray *out[], int nin, mxArray const *in[]) {
  (void)nout;
  if (nin < 2) mexErrMsgIdAndTxt("XM:invalidArgument", "The arguments are less than two.");
  mxInitGPU();
  int pool[2], stride[2] = {1, 1}, pad[4] = {0, 0, 0, 0}, method = XM_POOL_MAX;
  xm_intvec(in[1], pool, 2, "POOL");
  int next = 2;
  bool backward = nin > 2 && !mxIsChar(in[2]);
  if (backward) next = 3;
  for (; next < nin; ++next) {
    if (xm_streq(in[next], "stride")) xm_intvec(in[++next], stride, 2, "STRIDE");
    else if (xm_streq(in[next], "pad")) xm_intvec(in[++next], pad, 4, "PAD");
    else if (xm_streq(in[next], "method")) {
      ++next;
      if (xm_streq(in[next], "max")) method = XM_POOL_MAX;
      else if (xm_streq(in[next], "avg")) method = XM_POOL_AVG;
      else mexErrMsgIdAndTxt("XM:invalidArgument", "METHOD is not a supported method.");
    } else if (xm_streq(in[next], "cudnn") || xm_streq(in[next], "nocudnn") || xm_streq(in[next], "verbose")) {}
    else mexErrMsgIdAndTxt("XM:invalidArgument", "Unknown option.");
  }
  XmTensor x = xm_input(in[0], "X");
  const int H = x.d[0], W = x.d[1], Cc = x.d[2], N = x.d[3];
  mxGPUArray *ko = nullptr;
  if (!backward) {
    int Ho = xm_out_size(H, pad[0], pad[1], pool[0], 1, stride[0]);
    int Wo = xm_out_size(W, pad[2], pad[3], pool[1], 1, stride[1]);
    float *y = xm_output(&out[0], &ko, Ho, Wo, Cc, N);
    xm_check(xm_nnpool_forward(x.ptr, H, W, Cc, N, pool[0], pool[1], stride[0], stride[1], pad[0], pad[1],
                               pad[2], pad[3], method, y, nullptr));
  } else {
    XmTensor dz = xm_input(in[2], "DZDY");
    float *dx = xm_output(&out[0], &ko, H, W, Cc, N);
    xm_check(xm_nnpool_backward(x.ptr, H, W, Cc, N, pool[0], pool[1], stride[0], stride[1], pad[0], pad[1],
                                pad[2], pad[3], method, dz.ptr, dx, nullptr));
    if (dz.gpu) mxGPUDestroyGPUArray(dz.gpu);
  }
  if (ko) mxGPUDestroyGPUArray(ko);
  if (x.gpu) mxGPUDestroyGPUArray(x.gpu);
}
