/* XM_DEVICE(CMD, ...) -- everything around the vl_nn* operators that a dagnn / cnn_train_dag run touches device memory
 * for, on a host without gpuArray (mex/matlab/xmArray.m is the MATLAB face of it):
 *
 *   h   = xm_device('upload', A)                 host single -> new device buffer, returns xmArray
 *   A   = xm_device('download', h)               xmArray -> host single of size h.sz          (gather)
 *         xm_device('free', ptr)                  release a buffer (xmArray.delete)
 *         xm_device('sync')                       wait for the device
 *   y   = xm_device('sum', a, b)                  dagnn.Sum forward: y = a + b                  (xm_sum2)
 *         xm_device('sgd', w, m, der, lr, momentum, weightDecay, batchSize)
 *                                                 accumulateGradients, trainMethod 'gradient', IN PLACE on w, m
 *         xm_device('average', w, der, lr, denom) trainMethod 'average' (bnorm moments), in place
 *         xm_device('scale', x, a)                x <- a x in place (moments * worker batch size before the exchange)
 *   id  = xm_device('comm_id')                    128-byte uint8 unique id (lab 1; labBroadcast it)
 *         xm_device('comm_init', id, labindex-1, numlabs)
 *         xm_device('push', h)                    ParameterServer.push: start the sum over workers (overlapped)
 *         xm_device('sync_params')                ParameterServer.sync / pull
 *   n   = xm_device('comm_count')
 *
 * Reference call sites these replace: gpuArray(...) / gather(...) in getBatchEmoVoxCeleb.m:197-205 and
 * fetch_emovoxceleb_imdb.m:129-131; cnn_train_dag's accumulateGradients + ParameterServer (run_distillation.m:88,
 * 170-182).  Not built here (no MATLAB): see xm_mex.h. */
#include "xm_mex.h"

static XmTensor need_handle(XmCall &call, const mxArray *a, const char *name) {
  if (!mxIsClass(a, "xmArray")) call.fail("XM:needHandle", "in-place commands need xmArray handles.");
  return call.input(a, name);
}

void mexFunction(int nout, mxArray *out[], int nin, mxArray const *in[]) {
  (void)nout;
  XmCall call;
  if (nin < 1 || !mxIsChar(in[0])) call.fail("XM:invalidArgument", "usage: xm_device(cmd, ...)");
  if (xm_streq(in[0], "upload") && nin == 2) {
    call.any_handle = true;   // the result is a handle
    if (!mxIsSingle(in[1])) call.fail("XM:needSingle", "upload takes a SINGLE array.");
    mwSize nd = mxGetNumberOfDimensions(in[1]);
    const mwSize *dims = mxGetDimensions(in[1]);
    if (nd > 4) call.fail("XM:tooManyDims", "tensor has more than 4 dimensions.");
    int d[4] = {1, 1, 1, 1};
    for (mwSize i = 0; i < nd; ++i) d[i] = (int)dims[i];
    XmCall::Out o = call.output(d[0], d[1], d[2], d[3]);
    call.check(xm_device_upload(o.ptr, mxGetData(in[1]), (size_t)d[0] * d[1] * d[2] * d[3] * sizeof(float)));
    out[0] = call.deliver(o);
  } else if (xm_streq(in[0], "download") && nin == 2) {
    XmTensor t = need_handle(call, in[1], "H");
    mwSize dims[4] = {(mwSize)t.d[0], (mwSize)t.d[1], (mwSize)t.d[2], (mwSize)t.d[3]};
    out[0] = mxCreateNumericArray(4, dims, mxSINGLE_CLASS, mxREAL);
    call.check(xm_device_download(mxGetData(out[0]), t.ptr, t.numel() * sizeof(float)));
  } else if (xm_streq(in[0], "free") && nin == 2) {
    call.check(xm_device_free((void *)(uintptr_t)(*(const uint64_t *)mxGetData(in[1]))));
  } else if (xm_streq(in[0], "sync")) {
    call.check(xm_device_synchronize());
  } else if (xm_streq(in[0], "sum") && nin == 3) {
    XmTensor a = need_handle(call, in[1], "A"), b = need_handle(call, in[2], "B");
    if (a.numel() != b.numel()) call.fail("XM:invalidArgument", "sum: sizes differ.");
    XmCall::Out y = call.output(a.d[0], a.d[1], a.d[2], a.d[3]);
    call.check(xm_sum2(a.ptr, b.ptr, a.numel(), 0, y.ptr, nullptr));
    out[0] = call.deliver(y);
  } else if (xm_streq(in[0], "sgd") && nin == 8) {
    XmTensor w = need_handle(call, in[1], "W"), m = need_handle(call, in[2], "M"), d = need_handle(call, in[3], "DER");
    call.check(xm_sgd_update((float *)w.ptr, (float *)m.ptr, d.ptr, w.numel(), (float)mxGetScalar(in[4]),
                             (float)mxGetScalar(in[5]), (float)mxGetScalar(in[6]), (float)mxGetScalar(in[7]), nullptr));
  } else if (xm_streq(in[0], "average") && nin == 5) {
    XmTensor w = need_handle(call, in[1], "W"), d = need_handle(call, in[2], "DER");
    call.check(xm_average_update((float *)w.ptr, d.ptr, w.numel(), (float)mxGetScalar(in[3]),
                                 (float)mxGetScalar(in[4]), nullptr));
  } else if (xm_streq(in[0], "scale") && nin == 3) {
    XmTensor x = need_handle(call, in[1], "X");
    call.check(xm_scale_f32((float *)x.ptr, x.numel(), (float)mxGetScalar(in[2]), nullptr));
  } else if (xm_streq(in[0], "comm_id")) {
    out[0] = mxCreateNumericMatrix(1, 128, mxUINT8_CLASS, mxREAL);
    call.check(xm_comm_unique_id(mxGetData(out[0])));
  } else if (xm_streq(in[0], "comm_init") && nin == 4) {
    if (mxGetNumberOfElements(in[1]) != 128) call.fail("XM:invalidArgument", "the unique id has 128 bytes.");
    call.check(xm_comm_init(mxGetData(in[1]), (int)mxGetScalar(in[2]), (int)mxGetScalar(in[3])));
  } else if (xm_streq(in[0], "push") && nin == 2) {
    XmTensor x = need_handle(call, in[1], "X");
    call.check(xm_parserv_push((float *)x.ptr, x.numel(), nullptr));
  } else if (xm_streq(in[0], "sync_params")) {
    call.check(xm_parserv_sync(nullptr));
  } else if (xm_streq(in[0], "comm_count")) {
    int n = 1;
    call.check(xm_comm_count(&n));
    out[0] = mxCreateDoubleScalar((double)n);
  } else {
    call.fail("XM:invalidArgument", "unknown command or wrong number of arguments.");
  }
}
