/* [Y, MASK] = VL_NNDROPOUT(X, 'rate', R);  Y = VL_NNDROPOUT(X, 'mask', M);  DZDX = VL_NNDROPOUT(X, DZDY, 'mask', M)
 * MatConvNet ships this operator as an M-file (matlab/vl_nndropout.m: a gpuArray rand + product); on an MI355X host it
 * is a gateway over xm_nndropout_forward / xm_nndropout_apply.  Call sites in the reference: the dagnn.DropOut layers
 * emoVoxCeleb/emoVoxZoo.m:116-135,272-277 puts behind fc6 / fc7 when opts.dropout > 0.
 * Extension options 'seed', 'offset' (doubles holding integers): the mask comes from the library's stateless Philox
 * stream (include/xmodal.h) -- MATLAB's generator is not available on the device; a dagnn.DropOut block keeps a running
 * offset (+ ceil(numel / 4) per call) so that masks never repeat. */
#include "xm_mex.h"

void mexFunction(int nout, mxArray *out[], int nin, mxArray const *in[]) {
  XmCall call;
  if (nin < 1) call.fail("XM:invalidArgument", "Not enough arguments.");
  float rate = 0.5f;
  unsigned long long seed = 0, offset = 0;
  const mxArray *maskArg = nullptr;
  int next = 1;
  const bool backward = nin > 1 && !mxIsChar(in[1]) && !mxIsEmpty(in[1]);
  if (nin > 1 && !mxIsChar(in[1])) next = 2;
  for (; next < nin; ++next) {
    if (xm_streq(in[next], "rate") && next + 1 < nin) rate = (float)mxGetScalar(in[++next]);
    else if (xm_streq(in[next], "mask") && next + 1 < nin) maskArg = in[++next];
    else if (xm_streq(in[next], "seed") && next + 1 < nin) seed = (unsigned long long)mxGetScalar(in[++next]);
    else if (xm_streq(in[next], "offset") && next + 1 < nin) offset = (unsigned long long)mxGetScalar(in[++next]);
    else call.fail("XM:invalidArgument", "Unknown option.");
  }
  XmTensor x = call.input(in[0], "X");
  XmCall::Out y = call.output(x.d[0], x.d[1], x.d[2], x.d[3]);
  if (backward || maskArg) {
    if (!maskArg) call.fail("XM:invalidArgument", "The backward call needs the 'mask' of the forward call.");
    XmTensor m = call.input(maskArg, "MASK");
    if (m.numel() != x.numel()) call.fail("XM:invalidArgument", "MASK must have the size of X.");
    XmTensor src = backward ? call.input(in[1], "DZDY") : x;
    call.check(xm_nndropout_apply(src.ptr, m.ptr, x.numel(), y.ptr, nullptr));
    out[0] = call.deliver(y);
    return;
  }
  XmCall::Out mask = call.output(x.d[0], x.d[1], x.d[2], x.d[3]);
  call.check(xm_nndropout_forward(x.ptr, x.numel(), rate, seed, offset, y.ptr, mask.ptr, nullptr));
  out[0] = call.deliver(y);
  if (nout > 1) out[1] = call.deliver(mask);
}
