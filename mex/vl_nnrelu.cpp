/* Y = VL_NNRELU(X, 'leak', L);  DZDX = VL_NNRELU(X, DZDY, 'leak', L)
 * MatConvNet ships this operator as an M-file (matlab/vl_nnrelu.m: max(x, 0) on a gpuArray); on an MI355X host
 * there is no gpuArray arithmetic, so it is a gateway over xm_nnrelu.  Call sites in the reference: every
 * dagnn.ReLU of the student / teacher graphs (emoVoxCeleb/emoVoxZoo.m:44,199 load them). */
#include "xm_mex.h"

void mexFunction(int nout, mxArray *out[], int nin, mxArray const *in[]) {
  (void)nout;
  XmCall call;
  if (nin < 1) call.fail("XM:invalidArgument", "Not enough arguments.");
  float leak = 0.f;
  int next = 1;
  const bool backward = nin > 1 && !mxIsChar(in[1]) && !mxIsEmpty(in[1]);
  if (nin > 1 && !mxIsChar(in[1])) next = 2;
  for (; next < nin; ++next) {
    if (xm_streq(in[next], "leak") && next + 1 < nin) leak = (float)mxGetScalar(in[++next]);
    else call.fail("XM:invalidArgument", "Unknown option.");
  }
  XmTensor x = call.input(in[0], "X");
  XmTensor dz;
  if (backward) dz = call.input(in[1], "DZDY");
  XmCall::Out y = call.output(x.d[0], x.d[1], x.d[2], x.d[3]);
  call.check(xm_nnrelu(x.ptr, x.numel(), leak, backward ? dz.ptr : nullptr, y.ptr, nullptr));
  out[0] = call.deliver(y);
}
