/*
 * lambda_snark/prover.h — ADDITIVE entry points for the prover-side polynomial path:
 * the cyclic NTT of rust-api/lambda-snark/src/ntt.rs and the NTT-path quotient polynomial of
 * rust-api/lambda-snark/src/r1cs.rs:474-506, on the same MI355X butterfly kernels as ntt.h.
 *
 * The reference computes these in Rust on the host (there is no C symbol to replace); a maintainer binds
 * them from `lagrange_interpolate_ntt` (r1cs.rs:772-793) and `compute_quotient_poly` — INTEGRATION.md §6.
 * Field: F_q with q = NTT_MODULUS = 2^64 - 2^32 + 1 and its 2^32-th root NTT_PRIMITIVE_ROOT
 * (rust-api/lambda-snark-core/src/lib.rs:58,78); other primes q < 2^61 work for the transforms.
 */
#pragma once

#include "lambda_snark/ntt.h"
#include "lambda_snark/r1cs.h"
#include "lambda_snark/types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* NTT_MODULUS and NTT_PRIMITIVE_ROOT (lambda-snark-core/src/lib.rs:58,78) */
uint64_t lsr_prover_modulus(void) LSR_NOEXCEPT;
uint64_t lsr_prover_root_2_32(void) LSR_NOEXCEPT;
/* compute_root_of_unity(n, NTT_MODULUS, NTT_PRIMITIVE_ROOT) (ntt.rs:226-233); 0 unless n = 2^k <= 2^32 */
uint64_t lsr_prover_root_of_unity(uint64_t n) LSR_NOEXCEPT;

/* Context for cyclic transforms of size n = 2^k in [2, 131072] over prime q with omega a primitive n-th root
 * (omega = 0: the reference's root, q must then be NTT_MODULUS).  device -1 = default.  Free with ntt_context_free.
 * The ntt.h / batch.h transforms accept such a context and then run the cyclic butterfly network in its
 * native order: forward = natural in -> bit-reversed out, inverse = bit-reversed in -> natural out. */
NttContext* lsr_cyclic_ntt_context_create(uint64_t q, uint32_t n, uint64_t omega, int device) LSR_NOEXCEPT;
int lsr_ntt_context_is_cyclic(const NttContext* ctx) LSR_NOEXCEPT;

/* ntt_forward / ntt_inverse of ntt.rs:117-201 for `batch` contiguous vectors of ctx->n words, natural order in
 * and out, host buffers, in place.  Inputs must be < q.  0 / -1. */
int lsr_cyclic_ntt_forward_batch(const NttContext* ctx, uint64_t* values, size_t batch) LSR_NOEXCEPT;
int lsr_cyclic_ntt_inverse_batch(const NttContext* ctx, uint64_t* values, size_t batch) LSR_NOEXCEPT;
/* device-resident bit-reversal of `batch` vectors of 2^logn words (d_out != d_in), asynchronous on `stream` */
int lsr_bit_reverse_device(uint64_t* d_out, const uint64_t* d_in, int logn, size_t batch, void* stream) LSR_NOEXCEPT;

/* ---- quotient polynomial Q = (A*B - C) / (X^m - 1) on the NTT path (r1cs.rs:386-389: m = 2^k, q = NTT_MODULUS) ---- */
typedef struct LsrQuotientPlan LsrQuotientPlan;
/* m = number of constraints, a power of two in [1, 131072]; NULL otherwise or without a GPU */
LsrQuotientPlan* lsr_quotient_plan_create(uint32_t m, int device) LSR_NOEXCEPT;
void lsr_quotient_plan_free(LsrQuotientPlan* plan) LSR_NOEXCEPT;
uint32_t lsr_quotient_plan_size(const LsrQuotientPlan* plan) LSR_NOEXCEPT;
/* `batch` independent instances.  a/b/c_evals = [batch][m] constraint evaluations (A z, B z, C z of
 * compute_constraint_evals, r1cs.rs:296-304), values < q.  quotient = [batch][m] receives the coefficients of Q
 * (zero padded); quotient_len[i] = the length compute_quotient_poly would return (trailing zeros trimmed, >= 1), or 0
 * when the division leaves a remainder — the reference's Err "remainder non-zero (witness invalid)" (r1cs.rs:1050-1054);
 * the m quotient words of such an instance are unspecified.  Host buffers.  0 / -1. */
int lsr_quotient_batch(LsrQuotientPlan* plan, const uint64_t* a_evals, const uint64_t* b_evals, const uint64_t* c_evals,
                       size_t batch, uint64_t* quotient, uint32_t* quotient_len) LSR_NOEXCEPT;
/* same on device-resident buffers, asynchronous on `stream`; d_quotient [batch][m], d_quotient_len [batch].  A plan owns one
 * workspace: the library runs the calls on one plan one behind the other whatever streams they are given (a call waits on the
 * host for the previous asynchronous call's last kernel before it enqueues — use one plan per stream for concurrency); the
 * first call of a given batch size allocates, so make it outside a stream capture. */
int lsr_quotient_batch_device(LsrQuotientPlan* plan, const uint64_t* d_a_evals, const uint64_t* d_b_evals,
                              const uint64_t* d_c_evals, size_t batch, uint64_t* d_quotient, uint32_t* d_quotient_len,
                              void* stream) LSR_NOEXCEPT;

/* ---- compute_quotient_poly(witness) in full, for one R1CS and many witnesses (r1cs.rs:474-506) ----
 * The three sparse products of compute_constraint_evals (r1cs.rs:296-304, SparseMatrix::mul_vec sparse_matrix.rs:259-289:
 * values and witness words reduced mod q as unsigned integers) run on the device in front of the pipeline above.
 * A, B, C: m x n_vars in the FFI's coordinate form (r1cs.h; duplicate (row, col) entries add up), m = 2^k in [1, 131072],
 * modulus NTT_MODULUS.  The matrices are copied; NULL on bad shapes / indices or without a GPU. */
typedef struct LsrR1csProver LsrR1csProver;
LsrR1csProver* lsr_r1cs_prover_create(const SparseMatrix* A, const SparseMatrix* B, const SparseMatrix* C, int device) LSR_NOEXCEPT;
void     lsr_r1cs_prover_free(LsrR1csProver* prover) LSR_NOEXCEPT;
uint32_t lsr_r1cs_prover_num_constraints(const LsrR1csProver* prover) LSR_NOEXCEPT;
uint32_t lsr_r1cs_prover_num_variables(const LsrR1csProver* prover) LSR_NOEXCEPT;
/* witnesses = [batch][n_vars] (host).  a/b/c_evals = [batch][m] receive A z, B z, C z (compute_constraint_evals). 0 / -1. */
int lsr_r1cs_constraint_evals_batch(LsrR1csProver* prover, const uint64_t* witnesses, size_t batch, uint64_t* a_evals,
                                    uint64_t* b_evals, uint64_t* c_evals) LSR_NOEXCEPT;
/* quotient / quotient_len as in lsr_quotient_batch; quotient_len[i] = 0 <=> witness i does not satisfy the R1CS
 * (is_satisfied, r1cs.rs:148-172 — the reference's Err "Witness does not satisfy R1CS constraints"). 0 / -1. */
int lsr_r1cs_quotient_batch(LsrR1csProver* prover, const uint64_t* witnesses, size_t batch, uint64_t* quotient,
                            uint32_t* quotient_len) LSR_NOEXCEPT;

#ifdef __cplusplus
}
#endif
