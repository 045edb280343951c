/* [Y, MOMENTS] = VL_NNBNORM(X, G, B, 'epsilon', E, 'moments', M)
 * [DX, DG, DB, MOMENTS] = VL_NNBNORM(X, G, B, DZDY, ...)
 * Drop-in for matlab/src/vl_nnbnorm.cu, backed by xm_nnbnorm_forward / xm_nnbnorm_backward. */
#include "xm_mex.h"

void mexFunction(int nout, mxArray *out[], int nin, mxArray const *in[]) {
  XmCall call;
  if (nin < 3) call.fail("XM:invalidArgument", "The arguments are less than three.");
  float eps = 1e-4f;
  XmTensor mom;
  int next = 3;
  const bool backward = nin > 3 && !mxIsChar(in[3]);
  if (backward) next = 4;
  for (; next < nin; ++next) {
    if (xm_streq(in[next], "epsilon") && next + 1 < nin) eps = (float)mxGetScalar(in[++next]);
    else if (xm_streq(in[next], "moments") && next + 1 < nin) mom = call.input(in[++next], "MOMENTS");
    else if (xm_ignored_option(in[next])) {}
    else call.fail("XM:invalidArgument", "Unknown option.");
  }
  XmTensor x = call.input(in[0], "X"), g = call.input(in[1], "G"), b = call.input(in[2], "B");
  const int H = x.d[0], W = x.d[1], Cc = x.d[2], N = x.d[3];
  if (!backward) {
    XmCall::Out y = call.output(H, W, Cc, N), mo = call.output(Cc, 2, 1, 1);
    call.check(xm_nnbnorm_forward(x.ptr, H, W, Cc, N, g.ptr, b.ptr, eps, mom.empty ? nullptr : mom.ptr, y.ptr, mo.ptr,
                                  nullptr));
    out[0] = call.deliver(y);
    if (nout > 1) out[1] = call.deliver(mo);
  } else {
    XmTensor dz = call.input(in[3], "DZDY");
    XmCall::Out dx = call.output(H, W, Cc, N), dg = call.output(Cc, 1, 1, 1), db = call.output(Cc, 1, 1, 1),
                mo = call.output(Cc, 2, 1, 1);
    call.check(xm_nnbnorm_backward(x.ptr, H, W, Cc, N, g.ptr, b.ptr, dz.ptr, eps, mom.empty ? nullptr : mom.ptr,
                                   dx.ptr, dg.ptr, db.ptr, mo.ptr, nullptr));
    out[0] = call.deliver(dx);
    if (nout > 1) out[1] = call.deliver(dg);
    if (nout > 2) out[2] = call.deliver(db);
    if (nout > 3) out[3] = call.deliver(mo);
  }
}
