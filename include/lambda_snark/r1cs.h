/*
 * lambda_snark/r1cs.h — SEAL/NTL-free form of the R1CS shim the Rust workspace links against.
 *
 * NOT part of the GPU hot path (SURVEY.md §2 marks it out of scope); it exists so that a workspace which swaps
 * cpp-core for this library still resolves the five `lambda_snark_r1cs_*` symbols that
 * rust-api/lambda-snark-core/src/r1cs.rs:121-141 declares.  Struct layouts are those of
 * cpp-core/include/lambda_snark/r1cs.h:38-79 without the `#include <NTL/ZZ_p.h>` (:26); behaviour follows
 * cpp-core/src/ffi.cpp:27-105 and cpp-core/src/r1cs.cpp:95-180, including the signed reading of 64-bit values
 * (`static_cast<long>`, r1cs.cpp:122-124,165-167: 0xFFFF…FFFF means -1).  Host code, plain 128-bit arithmetic.
 */
#pragma once

#include <stdbool.h>

#include "lambda_snark/types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {   /* reference r1cs.h:38-42 */
    uint32_t row;
    uint32_t col;
    uint64_t value;
} SparseEntry;

typedef struct {   /* reference r1cs.h:49-54 */
    SparseEntry* entries;
    size_t       n_entries;
    uint32_t     n_rows;
    uint32_t     n_cols;
} SparseMatrix;

typedef struct {   /* reference r1cs.h:76-79 */
    uint64_t* values;
    size_t    len;
} R1CSWitness;

/* ffi.cpp:27-50.  Deep-copies the matrices (r1cs.cpp:33-46).  NULL => ERR_NULL_PTR; row/column counts of A, B, C
 * differ => ERR_INVALID_PARAMS. */
LambdaSnarkError lambda_snark_r1cs_create(const SparseMatrix* A, const SparseMatrix* B, const SparseMatrix* C,
                                          uint64_t modulus, void** out_r1cs) LSR_NOEXCEPT;
/* ffi.cpp:59-79, r1cs.cpp:95-128.  (A z) o (B z) == C z over Z_modulus.  Wrong length or z[0] != 1 =>
 * ERR_INVALID_PARAMS; an entry outside the matrix/witness => ERR_CRYPTO_FAILED. */
LambdaSnarkError lambda_snark_r1cs_validate_witness(void* r1cs, const R1CSWitness* witness, bool* out_valid) LSR_NOEXCEPT;
void     lambda_snark_r1cs_free(void* r1cs) LSR_NOEXCEPT;                 /* ffi.cpp:85-89 */
uint32_t lambda_snark_r1cs_num_constraints(void* r1cs) LSR_NOEXCEPT;      /* ffi.cpp:94-97 */
uint32_t lambda_snark_r1cs_num_variables(void* r1cs) LSR_NOEXCEPT;        /* ffi.cpp:102-105 */

#ifdef __cplusplus
}
#endif
