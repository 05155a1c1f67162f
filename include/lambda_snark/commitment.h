/*
 * lambda_snark/commitment.h — Module-LWE vector commitment, MI355X backend.
 *
 * Same seven symbols and signatures as the reference header cpp-core/include/lambda_snark/commitment.h
 * (implementation: cpp-core/src/commitment.cpp).  The reference implements them with SEAL BFV symmetric
 * encryption and ignores `seed`; this library implements the header's documented contract
 * (commitment.h:43-52: c = A*s + M*message + e, seed => deterministic) as a rank-k Module-LWE
 * encryption over R_q = Z_q[X]/(X^n+1); the scheme is written down in DESIGN.md.
 */
#pragma once

#include "lambda_snark/types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* reference commitment.h:31, commitment.cpp:102-132.  NULL params => NULL.  ring_degree must be a power
 * of two in [2, 131072]; module_rank 0 is read as 1, > 16 is rejected; sigma must be finite, > 0.
 * params->modulus is used when it is a prime == 1 (mod 2n) with 2^40 <= q < 2^61, otherwise an internal
 * 44-bit NTT prime is selected (the reference ignores the field entirely: commitment.cpp:106-111).
 * Fresh key per context (as the reference); set LAMBDA_SNARK_KEY_SEED for a reproducible key. */
LweContext* lwe_context_create(const PublicParams* params) LSR_NOEXCEPT;

/* reference commitment.h:38, commitment.cpp:134-136.  NULL-safe; zeroizes the secret key. */
void lwe_context_free(LweContext* ctx) LSR_NOEXCEPT;

/* reference commitment.h:58-63, commitment.cpp:138-164.  message is truncated / zero-padded to n slots,
 * each slot taken mod the plaintext modulus t.  seed != 0 => deterministic; 0 => fresh entropy.
 * NULL on NULL ctx/message or device failure (message on stderr). */
LweCommitment* lwe_commit(LweContext* ctx, const uint64_t* message, size_t msg_len, uint64_t seed) LSR_NOEXCEPT;

/* reference commitment.h:70, commitment.cpp:166-177.  Zeroizes, NULL-safe. */
void lwe_commitment_free(LweCommitment* comm) LSR_NOEXCEPT;

/* reference commitment.h:78, commitment.cpp:179-198. */
LweCommitment* lwe_commitment_clone(const LweCommitment* comm) LSR_NOEXCEPT;

/* reference commitment.h:94-100, commitment.cpp:200-232.  1 valid / 0 invalid / -1 error.  Trapdoor
 * (secret-key) check like the reference's decrypt; `opening` is ignored (commitment.cpp:205). */
int lwe_verify_opening(const LweContext* ctx, const LweCommitment* commitment, const uint64_t* message,
                       size_t msg_len, const LweOpening* opening) LSR_NOEXCEPT;

/* reference commitment.h:113-118, commitment.cpp:234-276.  sum_i (coeffs[i] mod t) * commitments[i];
 * NULL entries skipped; NULL if nothing to combine. */
LweCommitment* lwe_linear_combine(const LweContext* ctx, const LweCommitment** commitments,
                                  const uint64_t* coeffs, size_t count) LSR_NOEXCEPT;

#ifdef __cplusplus
}
#endif
