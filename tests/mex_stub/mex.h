/* Declaration-only stand-in for MATLAB's mex.h / matrix.h, just wide enough to SYNTAX-CHECK the gateways under mex/
 * (tests/test_mex_sources.py: g++ -fsyntax-only).  Nothing here is linked or run; the signatures follow the
 * documented MATLAB C Matrix / MEX API.  It does not make the gateways "built": there is no MATLAB in the image. */
#pragma once
#include <cstddef>
#include <cstdio>
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef enum { mxUNKNOWN_CLASS = 0, mxDOUBLE_CLASS = 6, mxSINGLE_CLASS = 7, mxUINT8_CLASS = 9, mxUINT64_CLASS = 15 } mxClassID;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
extern "C" {
bool mxIsEmpty(const mxArray *);
bool mxIsSingle(const mxArray *);
bool mxIsChar(const mxArray *);
bool mxIsClass(const mxArray *, const char *);
mwSize mxGetNumberOfDimensions(const mxArray *);
const mwSize *mxGetDimensions(const mxArray *);
size_t mxGetNumberOfElements(const mxArray *);
void *mxGetData(const mxArray *);
double *mxGetPr(const mxArray *);
double mxGetScalar(const mxArray *);
int mxGetString(const mxArray *, char *, mwSize);
mxArray *mxGetProperty(const mxArray *, mwIndex, const char *);
mxArray *mxCreateNumericArray(mwSize, const mwSize *, mxClassID, mxComplexity);
mxArray *mxCreateNumericMatrix(mwSize, mwSize, mxClassID, mxComplexity);
mxArray *mxCreateDoubleMatrix(mwSize, mwSize, mxComplexity);
mxArray *mxCreateDoubleScalar(double);
int mexCallMATLAB(int, mxArray *[], int, mxArray *[], const char *);
[[noreturn]] void mexErrMsgIdAndTxt(const char *, const char *, ...);
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);
}
