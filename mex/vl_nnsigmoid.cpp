/* Y = VL_NNSIGMOID(X);  DZDX = VL_NNSIGMOID(X, DZDY)
 * M-file upstream (matlab/vl_nnsigmoid.m); gateway over xm_nnsigmoid.  Reference use: the SE gate of
 * senet50-ferplus (teacher/ferPlusZoo.m:93-101 loads the graph). */
#include "xm_mex.h"

void mexFunction(int nout, mxArray *out[], int nin, mxArray const *in[]) {
  (void)nout;
  XmCall call;
  if (nin < 1) call.fail("XM:invalidArgument", "Not enough arguments.");
  XmTensor x = call.input(in[0], "X");
  XmTensor dz;
  const bool backward = nin > 1 && !mxIsEmpty(in[1]);
  if (backward) dz = call.input(in[1], "DZDY");
  XmCall::Out y = call.output(x.d[0], x.d[1], x.d[2], x.d[3]);
  call.check(xm_nnsigmoid(x.ptr, x.numel(), backward ? dz.ptr : nullptr, y.ptr, nullptr));
  out[0] = call.deliver(y);
}
