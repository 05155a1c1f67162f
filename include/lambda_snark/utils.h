/*
 * lambda_snark/utils.h — discrete Gaussian sampler.
 *
 * Drop-in for cpp-core/include/lambda_snark/utils.h:27 (implementation cpp-core/src/utils.cpp:132-146):
 * CDT table to +-max(8, ceil(12 sigma)) built in long double on the host, first-index-with-cdf>=u
 * selection, sign from a second 64-bit word, two's-complement int64 stored in uint64.
 */
#pragma once

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 0 on success; -1 if output is NULL, len == 0, sigma is not finite or <= 0 (utils.cpp:133), or on a
 * device failure.  Entropy: one fresh 64-bit seed from the OS per call, expanded on the GPU. */
int sample_gaussian(uint64_t* output, size_t len, double sigma)
#ifdef __cplusplus
    noexcept
#endif
    ;

#ifdef __cplusplus
}
#endif
