// decompress.h -- C++-linkage drop-in for the reference's decompress entry point.
//
// Same symbol (_Z10decompressPjyPyPfS1_S1_), argument meaning and ownership
// rules as /root/reference/decompress.h:11-17 + decompress.cu:18-141.
// It forwards to wah_decompress() (include/wah.h).
#ifndef WAH_DROPIN_DECOMPRESS_H_
#define WAH_DROPIN_DECOMPRESS_H_

unsigned int *decompress(unsigned int *data,                   // host compressed stream, dataSize words
                         unsigned long long int dataSize,      // words
                         unsigned long long int *outSize,      // [out] ceil(31*G/32) decoded words; may be NULL
                         float *pTransferToDeviceTime,         // [out] ms, may be NULL
                         float *pCompressionTime,              // [out] ms, may be NULL
                         float *ptranserFromDeviceTime);       // [out] ms, may be NULL

#endif  // WAH_DROPIN_DECOMPRESS_H_
