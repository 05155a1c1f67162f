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
 * Fresh keys per context from 256 bits of OS entropy each (the reference: commitment.cpp:118-121); the public matrix and
 * the secret come from separate keys.  Reproducible keys only through lsr_lwe_context_create_seeded (batch.h).
 * NOISE BUDGET: decoding tolerates noise below Delta/2 = q/(2t).  A (sigma, n, k) whose fresh commitments could exceed it
 * — 8 * sqrt(2 k n) * sigma^2 >= Delta/2 — is refused here (NULL, message on stderr and in lsr_last_error) rather than
 * failing at verify time; with the 44-bit modulus and n = 4096, k = 2 that limit is sigma of about 90. */
LweContext* lwe_context_create(const PublicParams* params) LSR_NOEXCEPT;

/* reference commitment.h:38, commitment.cpp:134-136.  NULL-safe; zeroizes the secret key. */
void lwe_context_free(LweContext* ctx) LSR_NOEXCEPT;

/* reference commitment.h:58-63, commitment.cpp:138-164.  message is truncated / zero-padded to n slots, each slot taken
 * mod the plaintext modulus t (lsr_lwe_plain_modulus, about 2^20: SEAL's Batching(n, 20) prime).  The reference encodes
 * out-of-range words unchecked too; like there, such a word is bound only through its residue and never opens as given
 * (see lwe_verify_opening) — commit to limbs (lsr_words_to_limbs, batch.h) to bind wide field elements in full.
 * seed == 0: blinding from 256 bits of fresh OS entropy.  seed != 0: deterministic in (seed, message, context) — the stream
 * key is a PRF of all three, so a seed reused for another message or context never repeats the blinding (the reference
 * ignores `seed` and is never deterministic: commitment.cpp:142).  A non-zero seed carries at most 64 bits of entropy: a
 * caller that wants the hiding of security_level draws it fresh per proof or passes 0.
 * NULL on NULL ctx/message or device failure (message on stderr). */
LweCommitment* lwe_commit(LweContext* ctx, const uint64_t* message, size_t msg_len, uint64_t seed) LSR_NOEXCEPT;

/* reference commitment.h:70, commitment.cpp:166-177.  Zeroizes, NULL-safe. */
void lwe_commitment_free(LweCommitment* comm) LSR_NOEXCEPT;

/* reference commitment.h:78, commitment.cpp:179-198. */
LweCommitment* lwe_commitment_clone(const LweCommitment* comm) LSR_NOEXCEPT;

/* reference commitment.h:94-100, commitment.cpp:200-232.  1 valid / 0 invalid / -1 error.  Trapdoor
 * (secret-key) check like the reference's decrypt; `opening` is ignored (commitment.cpp:205).  The decoded slots are
 * compared with the message words AS GIVEN (OR of XOR, commitment.cpp:223-228): a claimed word >= t gives 0 even when it
 * is congruent to the committed one. */
int lwe_verify_opening(const LweContext* ctx, const LweCommitment* commitment, const uint64_t* message,
                       size_t msg_len, const LweOpening* opening) LSR_NOEXCEPT;

/* reference commitment.h:113-118, commitment.cpp:234-276.  sum_i (coeffs[i] mod t) * commitments[i];
 * NULL entries skipped; NULL if nothing to combine.
 * RANGE: the result's noise is sum_i (coeffs[i] mod t) times that of a fresh commitment.  The reference's 72-bit SEAL
 * modulus absorbs any coefficients below t; this library's default 44-bit modulus decodes while
 * sum_i (coeffs[i] mod t) * 8 sqrt(2 k n) sigma^2 < q/(2t) — about 800 at n = 4096, k = 2, sigma = 3.19 — and beyond that
 * the call returns NULL with a message (stderr, lsr_last_error) instead of a commitment that cannot be opened.  For the
 * reference's full range create the context with a 60-bit NTT prime as params->modulus (e.g. 1152921504606584833 for
 * n <= 131072): it is honoured and gives sum of coefficients up to about 2^26.  Inputs that are themselves combinations
 * carry their own accumulated noise, which the library cannot see. */
LweCommitment* lwe_linear_combine(const LweContext* ctx, const LweCommitment** commitments,
                                  const uint64_t* coeffs, size_t count) LSR_NOEXCEPT;

#ifdef __cplusplus
}
#endif
