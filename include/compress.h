// compress.h -- C++-linkage drop-in for the reference's compress entry point.
//
// Same symbol (_Z8compressPjyPyPfS1_S1_), argument meaning and ownership rules
// as /root/reference/compress.h:12-18 + compress.cu:41-209, so source.cpp and
// tests.cpp of the reference build against libwah_hip.so without edits.
// It forwards to wah_compress() (include/wah.h), which documents the contract.
#ifndef WAH_DROPIN_COMPRESS_H_
#define WAH_DROPIN_COMPRESS_H_

unsigned int *compress(unsigned int *data_cpu,               // host bitmap, dataSize words, not modified
                       unsigned long long int dataSize,      // words
                       unsigned long long int *outputSize,   // [out] compressed words; may be NULL
                       float *pTransferToDeviceTime,         // [out] ms, may be NULL
                       float *pCompressionTime,              // [out] ms, may be NULL
                       float *ptranserFromDeviceTime);       // [out] ms, may be NULL

#endif  // WAH_DROPIN_COMPRESS_H_
