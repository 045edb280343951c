/* Y = VL_NNPOOL(X, POOL, 'stride', S, 'pad', P, 'method', 'max'|'avg');  DX = VL_NNPOOL(X, POOL, DZDY, ...)
 * Drop-in for matlab/src/vl_nnpool.cu, backed by xm_nnpool_forward / xm_nnpool_backward. */
#include "xm_mex.h"

void mexFunction(int nout, mxArray *out[], int nin, mxArray const *in[]) {
  (void)nout;
  XmCall call;
  if (nin < 2) call.fail("XM:invalidArgument", "The arguments are less than two.");
  int pool[2], stride[2] = {1, 1}, pad[4] = {0, 0, 0, 0}, method = XM_POOL_MAX;
  xm_intvec(call, in[1], pool, 2, "POOL");
  int next = 2;
  const bool backward = nin > 2 && !mxIsChar(in[2]);
  if (backward) next = 3;
  for (; next < nin; ++next) {
    if (xm_streq(in[next], "stride") && next + 1 < nin) xm_intvec(call, in[++next], stride, 2, "STRIDE");
    else if (xm_streq(in[next], "pad") && next + 1 < nin) xm_intvec(call, in[++next], pad, 4, "PAD");
    else if (xm_streq(in[next], "method") && next + 1 < nin) {
      ++next;
      if (xm_streq(in[next], "max")) method = XM_POOL_MAX;
      else if (xm_streq(in[next], "avg")) method = XM_POOL_AVG;
      else call.fail("XM:invalidArgument", "METHOD is not a supported method.");
    } else if (xm_ignored_option(in[next])) {}
    else call.fail("XM:invalidArgument", "Unknown option.");
  }
  XmTensor x = call.input(in[0], "X");
  const int H = x.d[0], W = x.d[1], Cc = x.d[2], N = x.d[3];
  if (!backward) {
    const int Ho = xm_out_size(H, pad[0], pad[1], pool[0], 1, stride[0]);
    const int Wo = xm_out_size(W, pad[2], pad[3], pool[1], 1, stride[1]);
    XmCall::Out y = call.output(Ho > 0 ? Ho : 0, Wo > 0 ? Wo : 0, Cc, N);
    call.check(xm_nnpool_forward(x.ptr, H, W, Cc, N, pool[0], pool[1], stride[0], stride[1], pad[0], pad[1], pad[2],
                                 pad[3], method, y.ptr, nullptr));
    out[0] = call.deliver(y);
  } else {
    XmTensor dz = call.input(in[2], "DZDY");
    XmCall::Out dx = call.output(H, W, Cc, N);
    call.check(xm_nnpool_backward(x.ptr, H, W, Cc, N, pool[0], pool[1], stride[0], stride[1], pad[0], pad[1], pad[2],
                                  pad[3], method, dz.ptr, dx.ptr, nullptr));
    out[0] = call.deliver(dx);
  }
}
