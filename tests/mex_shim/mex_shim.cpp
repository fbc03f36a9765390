// mex_shim.cpp -- implementation of tests/mex_shim/mex.h (see there).  Test infrastructure.
#include "mex.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

struct mxArray_tag {
    int classid = mxDOUBLE_CLASS;
    std::vector<mwSize> dims;
    std::vector<char> data;
    std::vector<std::string> fnames;
    std::vector<mxArray *> fields;
};
static size_t elem_size(int c) { return c == mxDOUBLE_CLASS ? 8 : c == mxINT32_CLASS ? 4 : c == mxCHAR_CLASS ? 1 : 0; }
static size_t numel(const mxArray *a)
{
    size_t n = 1;
    for (mwSize d : a->dims) n *= d;
    return a->dims.empty() ? 0 : n;
}
extern "C" {
mxArray *mxCreateNumericArray(mwSize ndim, const mwSize *dims, mxClassID classid, mxComplexity)
{
    mxArray *a = new mxArray_tag();
    a->classid = classid;
    a->dims.assign(dims, dims + ndim);
    while (a->dims.size() < 2) a->dims.push_back(1);
    a->data.assign(numel(a) * elem_size(classid), 0);      // MATLAB zero-fills
    return a;
}
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity f)
{
    const mwSize d[2] = {m, n};
    return mxCreateNumericArray(2, d, mxDOUBLE_CLASS, f);
}
mxArray *mxCreateString(const char *str)
{
    const mwSize d[2] = {1, strlen(str)};
    mxArray *a = mxCreateNumericArray(2, d, mxCHAR_CLASS, mxREAL);
    memcpy(a->data.data(), str, strlen(str));
    return a;
}
mxArray *mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char **fieldnames)
{
    mxArray *a = new mxArray_tag();
    a->classid = mxSTRUCT_CLASS;
    a->dims = {m, n};
    for (int i = 0; i < nfields; i++) { a->fnames.push_back(fieldnames[i]); a->fields.push_back(nullptr); }
    return a;
}
void mxSetFieldByNumber(mxArray *pa, mwSize index, int fieldnumber, mxArray *value)
{
    if (index != 0 || fieldnumber < 0 || fieldnumber >= (int)pa->fields.size()) throw MexError{"shim:field", "bad struct field index"};
    pa->fields[(size_t)fieldnumber] = value;
}
int mxShimNumberOfFields(const mxArray *pa) { return (int)pa->fields.size(); }
mxArray *mxShimGetFieldByNumber(const mxArray *pa, int f) { return pa->fields[(size_t)f]; }
const char *mxShimGetFieldName(const mxArray *pa, int f) { return pa->fnames[(size_t)f].c_str(); }
int mxShimClass(const mxArray *pa) { return pa->classid; }
void mxDestroyArray(mxArray *pa)
{
    if (!pa) return;
    for (mxArray *f : pa->fields) mxDestroyArray(f);
    delete pa;
}
double *mxGetPr(const mxArray *pa) { return pa->classid == mxDOUBLE_CLASS && !pa->data.empty() ? (double *)pa->data.data() : nullptr; }
void *mxGetData(const mxArray *pa) { return pa->data.empty() ? nullptr : (void *)pa->data.data(); }
size_t mxGetM(const mxArray *pa) { return pa->dims.empty() ? 0 : pa->dims[0]; }
size_t mxGetN(const mxArray *pa)
{
    if (pa->dims.size() < 2) return 0;
    size_t n = 1;
    for (size_t i = 1; i < pa->dims.size(); i++) n *= pa->dims[i];
    return n;
}
size_t mxGetNumberOfElements(const mxArray *pa) { return numel(pa); }
mwSize mxGetNumberOfDimensions(const mxArray *pa) { return pa->dims.size(); }
const mwSize *mxGetDimensions(const mxArray *pa) { return pa->dims.data(); }
bool mxIsEmpty(const mxArray *pa) { return numel(pa) == 0; }
double mxGetScalar(const mxArray *pa)
{
    if (numel(pa) == 0) throw MexError{"shim:scalar", "mxGetScalar of an empty array"};
    if (pa->classid == mxDOUBLE_CLASS) return *(const double *)pa->data.data();
    if (pa->classid == mxINT32_CLASS) return (double)*(const int32_t *)pa->data.data();
    return (double)(unsigned char)pa->data[0];
}
int mxGetString(const mxArray *pa, char *str, mwSize len)
{
    if (pa->classid != mxCHAR_CLASS || len == 0) return 1;
    const size_t n = numel(pa);
    const size_t k = n < len - 1 ? n : len - 1;
    memcpy(str, pa->data.data(), k);
    str[k] = 0;
    return n > len - 1 ? 1 : 0;
}
static std::vector<void (*)(void)> g_at_exit;
int mexAtExit(void (*exit_fcn)(void))
{
    for (auto f : g_at_exit) if (f == exit_fcn) return 0;       // MATLAB keeps one exit function per MEX file
    g_at_exit.push_back(exit_fcn);
    return 0;
}
void mxShimRunAtExit(void)
{
    for (auto f : g_at_exit) f();
    g_at_exit.clear();
}
void mexErrMsgTxt(const char *msg) { throw MexError{"", msg}; }
void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw MexError{id, buf};
}
}
