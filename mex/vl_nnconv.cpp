/* Y = VL_NNCONV(X, F, B, 'stride', S, 'pad', P, 'dilate', D)
 * [DX, DF, DB] = VL_NNCONV(X, F, B, DZDY, ..., 'NoDerData', 'NoDerFilters', 'NoDerBiases')
 * Drop-in for MatConvNet's matlab/src/vl_nnconv.cu, backed by xm_nnconv_forward / _backward. */
#include "xm_mex.h"

void mexFunction(int nout, mxArray *out[], int nin, mxArray const *in[]) {
  if (nin < 3) mexErrMsgIdAndTxt("XM:invalidArgument", "There are less than three arguments.");
  mxInitGPU();
  int stride[2] = {1, 1}, pad[4] = {0, 0, 0, 0}, dil[2] = {1, 1};
  bool derData = true, derFilt = true, derBias = true;
  int next = 3;
  bool backward = nin > 3 && !mxIsChar(in[3]);
  if (backward) next = 4;
  for (; next < nin; ++next) {
    if (xm_streq(in[next], "stride")) xm_intvec(in[++next], stride, 2, "STRIDE");
    else if (xm_streq(in[next], "pad")) xm_intvec(in[++next], pad, 4, "PAD");
    else if (xm_streq(in[next], "dilate")) xm_intvec(in[++next], dil, 2, "DILATE");
    else if (xm_streq(in[next], "noderdata")) derData = false;
    else if (xm_streq(in[next], "noderfilters")) derFilt = false;
    else if (xm_streq(in[next], "noderbiases")) derBias = false;
    else if (xm_streq(in[next], "cudnn") || xm_streq(in[next], "nocudnn") || xm_streq(in[next], "verbose")) {}
    else mexErrMsgIdAndTxt("XM:invalidArgument", "Unknown option.");
  }
  XmTensor x = xm_input(in[0], "X"), f = xm_input(in[1], "F"), b = xm_input(in[2], "B");
  const int H = x.d[0], W = x.d[1], Cc = x.d[2], N = x.d[3];
  const int FH = f.d[0], FW = f.d[1], FC = f.d[2], K = f.d[3];
  const int Ho = xm_out_size(H, pad[0], pad[1], FH, dil[0], stride[0]);
  const int Wo = xm_out_size(W, pad[2], pad[3], FW, dil[1], stride[1]);
  mxGPUArray *ky = nullptr, *kdx = nullptr, *kdf = nullptr, *kdb = nullptr;
  if (!backward) {
    float *y = xm_output(&out[0], &ky, Ho, Wo, K, N);
    xm_check(xm_nnconv_forward(x.ptr, H, W, Cc, N, f.ptr, FH, FW, FC, K, b.empty ? nullptr : b.ptr, y,
                               stride[0], stride[1], pad[0], pad[1], pad[2], pad[3], dil[0], dil[1],
                               nullptr /* MATLAB's default stream */));
  } else {
    XmTensor dz = xm_input(in[3], "DZDY");
    float *dx = derData ? xm_output(&out[0], &kdx, H, W, Cc, N) : nullptr;
    float *df = derFilt ? xm_output(&out[1], &kdf, FH, FW, FC, K) : nullptr;
    float *db = (derBias && !b.empty) ? xm_output(&out[2], &kdb, K, 1, 1, 1) : nullptr;
    if (!derData) out[0] = mxCreateDoubleMatrix(0, 0, mxREAL);
    if (!derFilt && nout > 1) out[1] = mxCreateDoubleMatrix(0, 0, mxREAL);
    if (!db && nout > 2) out[2] = mxCreateDoubleMatrix(0, 0, mxREAL);
    xm_check(xm_nnconv_backward(x.ptr, H, W, Cc, N, f.ptr, FH, FW, FC, K, dz.ptr, dx, df, db, stride[0],
                                stride[1], pad[0], pad[1], pad[2], pad[3], dil[0], dil[1], nullptr));
    if (dz.gpu) mxGPUDestroyGPUArray(dz.gpu);
  }
  for (mxGPUArray *k : {ky, kdx, kdf, kdb}) if (k) mxGPUDestroyGPUArray(k);
  for (mxGPUArray const *k : {x.gpu, f.gpu, b.gpu}) if (k) mxGPUDestroyGPUArray(k);
}
