/*
 * lambda_snark/ntt.h — negacyclic NTT over Z_q[X]/(X^n+1), MI355X backend.
 *
 * Same five symbols, signatures and error behaviour as the reference header
 * cpp-core/include/lambda_snark/ntt.h (implementation: cpp-core/src/ntt.cpp).  Buffers are caller-owned
 * HOST memory; the library moves them through the GPU.  Device-resident / batched entry points are in
 * lambda_snark/batch.h.
 */
#pragma once

#include "lambda_snark/types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* reference ntt.h:25 */
typedef struct NttContext NttContext;

/* reference ntt.h:34, ntt.cpp:30-70.  NULL unless n is a power of two in [2, 131072], 2 <= q < 2^61,
 * q prime and q == 1 (mod 2n).  psi is the numerically smallest primitive 2n-th root of unity (SEAL
 * NTTTables semantics).  Also NULL (with a message on stderr) when no usable GPU is present. */
NttContext* ntt_context_create(uint64_t q, uint32_t n) LSR_NOEXCEPT;

/* reference ntt.h:41, ntt.cpp:72-74.  NULL-safe. */
void ntt_context_free(NttContext* ctx) LSR_NOEXCEPT;

/* reference ntt.h:55-59, ntt.cpp:76-89.  In place, natural order in, bit-reversed evaluation order out:
 * out[i] = sum_j a_j psi^((2*bitrev(i)+1) j) mod q, every out[i] in [0,q).  0 on success; -1 if ctx or
 * coeffs is NULL, n != ctx degree, or a device error occurred. */
int ntt_forward(const NttContext* ctx, uint64_t* coeffs, uint32_t n) LSR_NOEXCEPT;

/* reference ntt.h:69-73, ntt.cpp:91-104.  Exact inverse of ntt_forward (bit-reversed in, natural out,
 * scaled by n^-1, canonical). */
int ntt_inverse(const NttContext* ctx, uint64_t* evals, uint32_t n) LSR_NOEXCEPT;

/* reference ntt.h:86-92, ntt.cpp:106-119.  result[i] = a[i]*b[i] mod q (canonical, any 64-bit inputs).
 * Silent no-op if any pointer is NULL; n is not validated against the context; result may alias a or b. */
void ntt_mul_pointwise(const NttContext* ctx, uint64_t* result, const uint64_t* a, const uint64_t* b,
                       uint32_t n) LSR_NOEXCEPT;

#ifdef __cplusplus
}
#endif
