// oracle_shim.cpp -- compress()/decompress() with the reference's C++ linkage
// (compress.h:12-18, decompress.h:11-17), implemented by the CPU ORACLE.
//
// TEST INFRASTRUCTURE ONLY.  It exists so the reference's own tests.cpp can be
// run against the oracle in a container without a GPU (oracle/Makefile target
// ref_tests_oracle).  The product never links this file.
#include <cstdlib>

#include "wah_oracle.h"

unsigned int *compress(unsigned int *data_cpu, unsigned long long int dataSize, unsigned long long int *outputSize,
                       float *tH2D, float *tKernel, float *tD2H) {
    unsigned int *out = (unsigned int *)std::malloc(sizeof(unsigned int) * (wah_oracle_max_words(dataSize) + 1));
    const unsigned long long c = wah_oracle_compress(data_cpu, dataSize, out);
    if (outputSize) *outputSize = c;
    if (tH2D) *tH2D = 0.f;
    if (tKernel) *tKernel = 0.f;
    if (tD2H) *tD2H = 0.f;
    return out;
}

unsigned int *decompress(unsigned int *data, unsigned long long int dataSize, unsigned long long int *outSize,
                         float *tH2D, float *tKernel, float *tD2H) {
    const unsigned long long g = wah_oracle_decoded_groups(data, dataSize);
    // the reference returns a buffer of G words of which ceil(31G/32) are meaningful (decompress.cu:127)
    unsigned int *out = (unsigned int *)std::calloc(g + 1, sizeof(unsigned int));
    const unsigned long long n = wah_oracle_decompress(data, dataSize, out);
    if (outSize) *outSize = n;
    if (tH2D) *tH2D = 0.f;
    if (tKernel) *tKernel = 0.f;
    if (tD2H) *tD2H = 0.f;
    return out;
}
