/* Y = VL_NNSOFTMAXCELOSS(X, P, 'temperature', T, 'logitTargets', TF, 'instanceWeights', W)
 * DZDX = VL_NNSOFTMAXCELOSS(X, P, DZDY, ...)
 * mcnExtraLayers M-file behind dagnn.SoftmaxCELoss -- the distillation loss of the reference:
 * dagnn.SoftmaxCELoss('temperature', 2, 'logitTargets', true) at emoVoxCeleb/emoVoxZoo.m:152
 * (also teacher/ferPlusZoo.m:244).  X, P: 1 x 1 x C x N.  Gateway over xm_nnsoftmaxceloss. */
#include "xm_mex.h"

void mexFunction(int nout, mxArray *out[], int nin, mxArray const *in[]) {
  (void)nout;
  XmCall call;
  if (nin < 2) call.fail("XM:invalidArgument", "Not enough arguments.");
  float T = 1.f;
  int logitTargets = 0, next = 2;
  XmTensor w;
  const bool backward = nin > 2 && !mxIsChar(in[2]) && !mxIsEmpty(in[2]);
  if (nin > 2 && !mxIsChar(in[2])) next = 3;
  for (; next < nin; ++next) {
    if (xm_streq(in[next], "temperature") && next + 1 < nin) T = (float)mxGetScalar(in[++next]);
    else if (xm_streq(in[next], "logittargets") && next + 1 < nin) logitTargets = mxGetScalar(in[++next]) != 0;
    else if (xm_streq(in[next], "instanceweights") && next + 1 < nin) w = call.input(in[++next], "INSTANCEWEIGHTS");
    else if (xm_streq(in[next], "tol") && next + 1 < nin) ++next;   /* clamp of the M-file's log(): the kernel uses log-sum-exp */
    else call.fail("XM:invalidArgument", "Unknown option.");
  }
  XmTensor x = call.input(in[0], "X"), p = call.input(in[1], "P");
  if (x.d[0] != 1 || x.d[1] != 1) call.fail("XM:invalidArgument", "X must be 1 x 1 x C x N.");
  const int Cc = x.d[2], N = x.d[3];
  if (!backward) {
    XmCall::Out y = call.output(1, 1, 1, 1);
    call.check(xm_nnsoftmaxceloss(x.ptr, p.ptr, Cc, N, T, logitTargets, w.empty ? nullptr : w.ptr, nullptr, y.ptr,
                                  nullptr));
    out[0] = call.deliver(y);
  } else {
    XmTensor dz = call.input(in[2], "DZDY");   /* scalar: 1 for {'objective', 1} */
    XmCall::Out dx = call.output(1, 1, Cc, N);
    call.check(xm_nnsoftmaxceloss(x.ptr, p.ptr, Cc, N, T, logitTargets, w.empty ? nullptr : w.ptr, dz.ptr, dx.ptr,
                                  nullptr));
    out[0] = call.deliver(dx);
  }
}
