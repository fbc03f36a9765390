// driver.cpp -- calls the mexFunction of one of the gateways in matlab/ with arrays read from a file and writes what it
// returned (tests/test_mex_boundary.py).  The gateways are compiled with -DmexFunction=mex_<name>.
//   driver <epiekf|batch|rt|sim|pipeline> <in.bin> <out.bin> <nlhs>
// File format: int32 count, then per array { int32 class (6 double, 12 int32, 4 char), int32 ndim, int64 dims[ndim],
// raw column-major data }.  A struct result is written field by field, in field order.
#include "mex.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
extern "C" {
void mex_epiekf(int, mxArray *[], int, const mxArray *[]);
void mex_batch(int, mxArray *[], int, const mxArray *[]);
void mex_rt(int, mxArray *[], int, const mxArray *[]);
void mex_sim(int, mxArray *[], int, const mxArray *[]);
void mex_pipeline(int, mxArray *[], int, const mxArray *[]);
}
static size_t esz(int c) { return c == mxDOUBLE_CLASS ? 8 : c == mxINT32_CLASS ? 4 : 1; }
static void write_array(FILE *f, const mxArray *a)
{
    const int32_t cls = mxShimClass(a), nd = (int32_t)mxGetNumberOfDimensions(a);
    fwrite(&cls, 4, 1, f); fwrite(&nd, 4, 1, f);
    for (int i = 0; i < nd; i++) { const int64_t v = (int64_t)mxGetDimensions(a)[i]; fwrite(&v, 8, 1, f); }
    const size_t bytes = mxGetNumberOfElements(a) * esz(cls);
    if (bytes) fwrite(mxGetData(a), 1, bytes, f);
}
int main(int argc, char **argv)
{
    if (argc != 5) { fprintf(stderr, "usage: driver <gateway> <in> <out> <nlhs>\n"); return 2; }
    FILE *f = fopen(argv[2], "rb");
    if (!f) { perror("in"); return 2; }
    int32_t count = 0;
    if (fread(&count, 4, 1, f) != 1) return 2;
    std::vector<const mxArray *> prhs;
    for (int k = 0; k < count; k++) {
        int32_t cls, nd;
        if (fread(&cls, 4, 1, f) != 1 || fread(&nd, 4, 1, f) != 1) return 2;
        std::vector<mwSize> dims((size_t)nd);
        for (int i = 0; i < nd; i++) { int64_t v; if (fread(&v, 8, 1, f) != 1) return 2; dims[(size_t)i] = (mwSize)v; }
        mxArray *a = mxCreateNumericArray((mwSize)nd, dims.data(), (mxClassID)cls, mxREAL);
        const size_t bytes = mxGetNumberOfElements(a) * esz(cls);
        if (bytes && fread(mxGetData(a), 1, bytes, f) != bytes) return 2;
        prhs.push_back(a);
    }
    fclose(f);
    const int nlhs = atoi(argv[4]);
    mxArray *plhs[8] = {nullptr};
    try {
        if (!strcmp(argv[1], "epiekf")) mex_epiekf(nlhs, plhs, (int)prhs.size(), prhs.data());
        else if (!strcmp(argv[1], "batch")) mex_batch(nlhs, plhs, (int)prhs.size(), prhs.data());
        else if (!strcmp(argv[1], "rt")) mex_rt(nlhs, plhs, (int)prhs.size(), prhs.data());
        else if (!strcmp(argv[1], "sim")) mex_sim(nlhs, plhs, (int)prhs.size(), prhs.data());
        else if (!strcmp(argv[1], "pipeline")) mex_pipeline(nlhs, plhs, (int)prhs.size(), prhs.data());
        else { fprintf(stderr, "unknown gateway\n"); return 2; }
    } catch (const MexError &e) {
        fprintf(stderr, "MEXERROR[%s]: %s\n", e.id.c_str(), e.msg.c_str());
        mxShimRunAtExit();
        return 3;
    }
    FILE *o = fopen(argv[3], "wb");
    if (!o) { perror("out"); return 2; }
    std::vector<const mxArray *> res;
    if (plhs[0] && mxShimClass(plhs[0]) == mxSTRUCT_CLASS) {
        for (int i = 0; i < mxShimNumberOfFields(plhs[0]); i++) res.push_back(mxShimGetFieldByNumber(plhs[0], i));
    } else {
        for (int i = 0; i < (nlhs > 1 ? nlhs : 1); i++) if (plhs[i]) res.push_back(plhs[i]);
    }
    const int32_t n = (int32_t)res.size();
    fwrite(&n, 4, 1, o);
    for (const mxArray *a : res) write_array(o, a);
    fclose(o);
    mxShimRunAtExit();        // what `clear mex` does: the gateways hand the library's pools and worker threads back
    return 0;
}
