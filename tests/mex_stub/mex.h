/* Declarations of the part of MATLAB's MEX C API (R2018a interleaved-complex API) that
 * mex/mmc_hip_mex.cpp uses, for a syntax and type check of the gateway where MATLAB is absent.
 * Nothing here is implemented: the test compiles with -fsyntax-only and never links. */
#ifndef MMC_TEST_MEX_STUB_H
#define MMC_TEST_MEX_STUB_H
#include <stddef.h>
#include <stdbool.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef bool mxLogical;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef enum { mxDOUBLE_CLASS = 6 } mxClassID;
bool mxIsChar(const mxArray *);
bool mxIsDouble(const mxArray *);
bool mxIsComplex(const mxArray *);
bool mxIsSparse(const mxArray *);
bool mxIsLogical(const mxArray *);
bool mxIsStruct(const mxArray *);
bool mxIsEmpty(const mxArray *);
bool mxIsNaN(double);
int mxGetString(const mxArray *, char *, mwSize);
double mxGetScalar(const mxArray *);
double *mxGetDoubles(const mxArray *);
mxLogical *mxGetLogicals(const mxArray *);
size_t mxGetNumberOfElements(const mxArray *);
size_t mxGetM(const mxArray *);
size_t mxGetN(const mxArray *);
mxArray *mxGetField(const mxArray *, mwSize, const char *);
void mxSetField(mxArray *, mwSize, const char *, mxArray *);
mxArray *mxCreateDoubleScalar(double);
mxArray *mxCreateDoubleMatrix(mwSize, mwSize, mxComplexity);
mxArray *mxCreateNumericArray(mwSize, const mwSize *, mxClassID, mxComplexity);
mxArray *mxCreateStructMatrix(mwSize, mwSize, int, const char **);
void mxDestroyArray(mxArray *);
int mexCallMATLAB(int, mxArray **, int, mxArray **, const char *);
int mexPrintf(const char *, ...);
void mexErrMsgIdAndTxt(const char *, const char *, ...);
int mexAtExit(void (*)(void));
void mexLock(void);
void mexUnlock(void);
bool mexIsLocked(void);
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);
#ifdef __cplusplus
}
#endif
#endif
