/* Y = VL_NNLOSS(X, C, 'loss', 'softmaxlog'|'classerror');  DZDX = VL_NNLOSS(X, C, DZDY, 'loss', ...)
 * M-file upstream (matlab/vl_nnloss.m).  Reference use: dagnn.Loss('loss','softmaxlog') and the 'classerror'
 * metric layers at emoVoxCeleb/emoVoxZoo.m:149,160 and teacher/ferPlusZoo.m:242,252.  X: 1 x 1 x C x N, C: 1-based
 * labels (single, 1 x 1 x 1 x N).  Gateway over xm_nnloss; other loss types of vl_nnloss are not on this path. */
#include "xm_mex.h"

void mexFunction(int nout, mxArray *out[], int nin, mxArray const *in[]) {
  (void)nout;
  XmCall call;
  if (nin < 2) call.fail("XM:invalidArgument", "Not enough arguments.");
  int loss = XM_LOSS_SOFTMAXLOG, next = 2;
  const bool backward = nin > 2 && !mxIsChar(in[2]) && !mxIsEmpty(in[2]);
  if (nin > 2 && !mxIsChar(in[2])) next = 3;
  for (; next < nin; ++next) {
    if (xm_streq(in[next], "loss") && next + 1 < nin) {
      ++next;
      if (xm_streq(in[next], "softmaxlog")) loss = XM_LOSS_SOFTMAXLOG;
      else if (xm_streq(in[next], "classerror")) loss = XM_LOSS_CLASSERROR;
      else call.fail("XM:notSupported", "only 'softmaxlog' and 'classerror' are on the built path.");
    } else call.fail("XM:invalidArgument", "Unknown option.");
  }
  XmTensor x = call.input(in[0], "X"), c = call.input(in[1], "C");
  if (x.d[0] != 1 || x.d[1] != 1) call.fail("XM:invalidArgument", "X must be 1 x 1 x C x N.");
  const int Cc = x.d[2], N = x.d[3];
  if (!backward) {
    XmCall::Out y = call.output(1, 1, 1, 1);
    call.check(xm_nnloss(x.ptr, c.ptr, Cc, N, loss, nullptr, y.ptr, nullptr));
    out[0] = call.deliver(y);
  } else {
    XmTensor dz = call.input(in[2], "DZDY");
    XmCall::Out dx = call.output(1, 1, Cc, N);
    call.check(xm_nnloss(x.ptr, c.ptr, Cc, N, loss, dz.ptr, dx.ptr, nullptr));
    out[0] = call.deliver(dx);
  }
}
