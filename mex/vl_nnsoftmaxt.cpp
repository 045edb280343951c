/* Y = VL_NNSOFTMAXT(X, 'temperature', T, 'dim', D);  DZDX = VL_NNSOFTMAXT(X, DZDY, ...)
 * mcnExtraLayers M-file (softmax with temperature along dimension D, default 3).  Reference use:
 * emoVoxCeleb/student_stats.m:95 (vl_nnsoftmaxt(logits, 'dim', 2)).  Gateway over xm_nnsoftmaxt /
 * xm_nnsoftmaxt_backward: the tensor is viewed as (HW = prod(size(1:D-1))) x (C = size(D)) x (N = the rest). */
#include "xm_mex.h"

void mexFunction(int nout, mxArray *out[], int nin, mxArray const *in[]) {
  (void)nout;
  XmCall call;
  if (nin < 1) call.fail("XM:invalidArgument", "Not enough arguments.");
  float T = 1.f;
  int dim = 3, next = 1;
  const bool backward = nin > 1 && !mxIsChar(in[1]) && !mxIsEmpty(in[1]);
  if (nin > 1 && !mxIsChar(in[1])) next = 2;
  for (; next < nin; ++next) {
    if (xm_streq(in[next], "temperature") && next + 1 < nin) T = (float)mxGetScalar(in[++next]);
    else if (xm_streq(in[next], "dim") && next + 1 < nin) dim = (int)mxGetScalar(in[++next]);
    else call.fail("XM:invalidArgument", "Unknown option.");
  }
  if (dim < 1 || dim > 4) call.fail("XM:invalidArgument", "DIM must be between 1 and 4.");
  XmTensor x = call.input(in[0], "X");
  int HW = 1, N = 1;
  for (int i = 0; i < dim - 1; ++i) HW *= x.d[i];
  for (int i = dim; i < 4; ++i) N *= x.d[i];
  const int Cc = x.d[dim - 1];
  XmCall::Out y = call.output(x.d[0], x.d[1], x.d[2], x.d[3]);
  if (!backward) {
    call.check(xm_nnsoftmaxt(x.ptr, HW, Cc, N, T, y.ptr, nullptr));
  } else {
    XmTensor dz = call.input(in[1], "DZDY");
    call.check(xm_nnsoftmaxt_backward(x.ptr, dz.ptr, HW, Cc, N, T, y.ptr, nullptr));
  }
  out[0] = call.deliver(y);
}
