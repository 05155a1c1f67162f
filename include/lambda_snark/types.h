/*
 * lambda_snark/types.h — FFI-visible types of the commitment kernel's C-ABI.
 *
 * Drop-in for the reference header cpp-core/include/lambda_snark/types.h: every struct layout and
 * enum value below is what lambda-snark-sys's bindgen consumes (rust-api/lambda-snark-sys/build.rs:185-207),
 * so they are ABI.  Reference lines are cited per item.
 */
#pragma once

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
#define LSR_NOEXCEPT noexcept
extern "C" {
#else
#define LSR_NOEXCEPT
#endif

/* opaque handle (reference types.h:27; layout private to the library) */
typedef struct LweContext LweContext;

/* reference types.h:36-39.  data[0] = payload byte length, data[1..] = payload zero-padded to 8 bytes
 * (framing of cpp-core/src/commitment.cpp:44-60).  Rust dereferences both fields
 * (rust-api/lambda-snark/src/commitment.rs:88-93). */
typedef struct {
    uint64_t* data;
    size_t    len;
} LweCommitment;

/* reference types.h:44-47 (ignored by lwe_verify_opening, as in commitment.cpp:205) */
typedef struct {
    uint64_t* randomness;
    size_t    rand_len;
} LweOpening;

/* reference types.h:52-55 */
typedef enum {
    PROFILE_SCALAR_A = 0,
    PROFILE_RING_B   = 1,
} ProfileType;

/* reference types.h:60-67 (32 bytes: u32,u32,u64,u32,u32,f64) */
typedef struct {
    ProfileType profile;
    uint32_t    security_level;
    uint64_t    modulus;
    uint32_t    ring_degree;
    uint32_t    module_rank;
    double      sigma;
} PublicParams;

/* reference types.h:72-78 */
typedef enum {
    LAMBDA_SNARK_OK                 = 0,
    LAMBDA_SNARK_ERR_NULL_PTR       = 1,
    LAMBDA_SNARK_ERR_INVALID_PARAMS = 2,
    LAMBDA_SNARK_ERR_ALLOC_FAILED   = 3,
    LAMBDA_SNARK_ERR_CRYPTO_FAILED  = 4,
} LambdaSnarkError;

#ifdef __cplusplus
}
#endif
