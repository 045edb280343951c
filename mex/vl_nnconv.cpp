/* Y = VL_NNCONV(X, F, B, 'stride', S, 'pad', P, 'dilate', D)
 * [DX, DF, DB] = VL_NNCONV(X, F, B, DZDY, ..., 'NoDerData', 'NoDerFilters', 'NoDerBiases')
 * Drop-in for MatConvNet's matlab/src/vl_nnconv.cu, backed by xm_nnconv_forward / xm_nnconv_backward.
 * Tensor arguments: xmArray handles (zero copy) or host singles (staged) -- see xm_mex.h. */
#include "xm_mex.h"

void mexFunction(int nout, mxArray *out[], int nin, mxArray const *in[]) {
  XmCall call;
  if (nin < 3) call.fail("XM:invalidArgument", "There are less than three arguments.");
  int stride[2] = {1, 1}, pad[4] = {0, 0, 0, 0}, dil[2] = {1, 1};
  bool derData = true, derFilt = true, derBias = true;
  int next = 3;
  const bool backward = nin > 3 && !mxIsChar(in[3]);
  if (backward) next = 4;
  for (; next < nin; ++next) {
    if (xm_streq(in[next], "stride") && next + 1 < nin) xm_intvec(call, in[++next], stride, 2, "STRIDE");
    else if (xm_streq(in[next], "pad") && next + 1 < nin) xm_intvec(call, in[++next], pad, 4, "PAD");
    else if (xm_streq(in[next], "dilate") && next + 1 < nin) xm_intvec(call, in[++next], dil, 2, "DILATE");
    else if (xm_streq(in[next], "noderdata")) derData = false;
    else if (xm_streq(in[next], "noderfilters")) derFilt = false;
    else if (xm_streq(in[next], "noderbiases")) derBias = false;
    else if (xm_ignored_option(in[next])) {}
    else call.fail("XM:invalidArgument", "Unknown option.");
  }
  XmTensor x = call.input(in[0], "X"), f = call.input(in[1], "F"), b = call.input(in[2], "B");
  const int H = x.d[0], W = x.d[1], Cc = x.d[2], N = x.d[3];
  const int FH = f.d[0], FW = f.d[1], FC = f.d[2], K = f.d[3];
  const int Ho = xm_out_size(H, pad[0], pad[1], FH, dil[0], stride[0]);
  const int Wo = xm_out_size(W, pad[2], pad[3], FW, dil[1], stride[1]);
  if (!backward) {
    XmCall::Out y = call.output(Ho > 0 ? Ho : 0, Wo > 0 ? Wo : 0, K, N);
    call.check(xm_nnconv_forward(x.ptr, H, W, Cc, N, f.ptr, FH, FW, FC, K, b.empty ? nullptr : b.ptr, y.ptr,
                                 stride[0], stride[1], pad[0], pad[1], pad[2], pad[3], dil[0], dil[1],
                                 nullptr /* the null stream: MATLAB's MEX calls are serial */));
    out[0] = call.deliver(y);
    return;
  }
  XmTensor dz = call.input(in[3], "DZDY");
  XmCall::Out dx, df, db;
  if (derData) dx = call.output(H, W, Cc, N);
  if (derFilt) df = call.output(FH, FW, FC, K);
  if (derBias && !b.empty) db = call.output(K, 1, 1, 1);
  call.check(xm_nnconv_backward(x.ptr, H, W, Cc, N, f.ptr, FH, FW, FC, K, dz.ptr, dx.ptr, df.ptr, db.ptr, stride[0],
                                stride[1], pad[0], pad[1], pad[2], pad[3], dil[0], dil[1], nullptr));
  out[0] = dx.ptr ? call.deliver(dx) : mxCreateDoubleMatrix(0, 0, mxREAL);
  if (nout > 1) out[1] = df.ptr ? call.deliver(df) : mxCreateDoubleMatrix(0, 0, mxREAL);
  if (nout > 2) out[2] = db.ptr ? call.deliver(db) : mxCreateDoubleMatrix(0, 0, mxREAL);
}
