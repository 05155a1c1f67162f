/*
 * lsr_oracle.h — CPU oracle for the Lambda-SNARK-R hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (lambda-snark-r_amd/, include/) may include,
 * link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and
 * only as the checker / the CPU baseline.
 *
 * What it restates (file:line into /root/reference):
 *   - cpp-core/src/ntt.cpp:30-119      C-ABI validation + error contract of ntt_*
 *   - Microsoft SEAL >= 4.1.2 (cpp-core/vcpkg.json:8-9, NOT vendored in the reference tree):
 *     util/ntt.cpp (NTTTables), util/dwthandler.h (Harvey lazy CT / GS), util/numth.cpp
 *     (try_minimal_primitive_root), util/uintarithsmallmod.h (multiply_uint_mod[_lazy]) — the
 *     published algorithm, restated from its documented semantics (SURVEY.md §8(a) N2–N5).
 *   - cpp-core/src/utils.cpp:26-146    CDT Gaussian sampler (table + branch-free scan)
 *   - cpp-core/include/lambda_snark/commitment.h:43-52 + cpp-core/src/commitment.cpp:44-60,138-276
 *     commitment contract and wire framing (data[0] = payload byte length).
 *
 * PARITY PINNING:
 *   - pinned by the reference's own tests: ntt round trip (test_ntt.cpp:47-68), pointwise 2*3=6
 *     (test_ntt.cpp:70-81), error codes (test_ntt.cpp:83-90), the ROOTS_OF_UNITY table
 *     (rust-api/lambda-snark/src/r1cs.rs:534-547) for modular exponentiation, sampler moments
 *     (test_utils.cpp:26-70), commitment behaviour (test_commitment.cpp:37-166).
 *   - forward-NTT output VALUES / ordering / choice of psi: *parity unpinned by the reference* (it
 *     holds no forward KAT and cannot be built here: SEAL absent).  They are pinned instead by the
 *     mathematical definition out[i] = a(psi^(2*bitrev(i)+1)) with psi the minimal primitive 2n-th
 *     root (checked by an independent O(n^2) evaluation in tests) and by the survey's check values.
 *   - sampler: besides test_utils.cpp's moment bounds, the reference's own utils.cpp compiled into oracle/_ref
 *     (`make ref`) is compared with this restatement in tests/test_reference_sampler.py (distribution against the
 *     restated table, tail bound, argument contract).
 *   - commitment bytes: unpinnable (reference is non-deterministic, commitment.cpp:142); bit-exactness
 *     is GPU-vs-this-oracle under the same seeds.
 */
#ifndef LSR_ORACLE_H
#define LSR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------- number theory ---------- */
uint64_t oracle_mulmod(uint64_t a, uint64_t b, uint64_t q);
uint64_t oracle_powmod(uint64_t a, uint64_t e, uint64_t q);
int      oracle_is_prime(uint64_t q);
/* minimal primitive 2n-th root of unity mod q (SEAL try_minimal_primitive_root); 0 if none */
uint64_t oracle_minimal_primitive_root(uint64_t two_n, uint64_t q);
/* largest prime p < 2^bits with p == 1 (mod factor) (SEAL get_primes(factor,bits,1)); 0 if none */
uint64_t oracle_largest_prime_1mod(uint64_t factor, int bits);

/* ---------- NTT (SEAL NTTTables + Harvey butterflies) ---------- */
typedef struct oracle_ntt oracle_ntt;
oracle_ntt* oracle_ntt_create(uint64_t q, uint32_t n);          /* NULL per ntt.cpp:30-70 rules */
void        oracle_ntt_free(oracle_ntt* t);
uint64_t    oracle_ntt_root(const oracle_ntt* t);               /* psi */
/* root_powers[i].operand (bit-reversed psi powers) / inv_root_powers (SEAL scrambled layout) */
void        oracle_ntt_tables(const oracle_ntt* t, uint64_t* root_powers, uint64_t* inv_root_powers);
int         oracle_ntt_forward(const oracle_ntt* t, uint64_t* a, uint32_t n);   /* 0 / -1 */
int         oracle_ntt_inverse(const oracle_ntt* t, uint64_t* a, uint32_t n);   /* 0 / -1 */
void        oracle_ntt_mul_pointwise(const oracle_ntt* t, uint64_t* r, const uint64_t* a, const uint64_t* b, uint32_t n);
/* batched helpers for the CPU baseline (contiguous polys) */
int         oracle_ntt_forward_batch(const oracle_ntt* t, uint64_t* a, size_t batch);
int         oracle_ntt_inverse_batch(const oracle_ntt* t, uint64_t* a, size_t batch);
/* definition-level check: out[i] = sum_j a_j psi^((2*bitrev(i)+1) j), O(n^2) */
void        oracle_ntt_forward_naive(const oracle_ntt* t, const uint64_t* a, uint64_t* out);

/* ---------- splitmix64 input generator (SURVEY.md §8(c) check values) ---------- */
void oracle_splitmix_fill(uint64_t seed, uint64_t q, uint64_t* out, size_t len);

/* ---------- ChaCha20 block function (RFC 8439 §2.3) ---------- */
void oracle_chacha20_block(const uint32_t key[8], uint32_t counter, const uint32_t nonce[3], uint32_t out[16]);
/* the library's seeded stream: 64-bit words of object (seed, domain, index); word w lives in
 * block w/8 at position w%8 (little-endian pairs of ChaCha words). */
void oracle_stream_words(uint64_t seed, uint32_t domain, uint64_t index, uint64_t first_word, uint64_t* out, size_t count);

/* ---------- CDT Gaussian sampler (utils.cpp:26-146) ---------- */
/* builds the table; returns number of entries (<= cap) or 0 on invalid sigma */
size_t oracle_gaussian_cdf(double sigma, uint64_t* cdf, size_t cap);
/* reference-equivalent sampler on fresh entropy (std::random_device analogue: /dev/urandom) */
int oracle_sample_gaussian(uint64_t* out, size_t len, double sigma);
/* seeded variant: sample i uses stream word i — low bit = sign, upper 63 bits against the table >> 1 */
int oracle_sample_gaussian_seeded(uint64_t* out, size_t len, double sigma, uint64_t seed, uint32_t domain, uint64_t index);

/* ---------- Module-LWE commitment (definition: DESIGN.md §commitment) ---------- */
typedef struct oracle_lwe oracle_lwe;
/* internal parameter selection */
/* key schedule: context keys from a non-zero key seed; per-commitment key = PRF(seed, id, embedded message) */
void oracle_context_keys(uint64_t key_seed, uint32_t pub[8], uint32_t sec[8], uint32_t id[4]);
void oracle_commit_key(uint64_t seed, const uint32_t id[4], const uint64_t* msg, size_t copy, uint64_t t, uint32_t out[8]);
uint64_t oracle_lwe_select_modulus(uint64_t requested_q, uint32_t n);
oracle_lwe* oracle_lwe_create(uint64_t requested_q, uint32_t n, uint32_t k, double sigma, uint64_t key_seed);
void        oracle_lwe_free(oracle_lwe* c);
uint64_t    oracle_lwe_q(const oracle_lwe* c);
uint64_t    oracle_lwe_t(const oracle_lwe* c);
size_t      oracle_lwe_commit_words(const oracle_lwe* c);       /* total words incl. data[0] */
/* writes commitment words (data[0] = payload byte length) into out; 0 ok / -1 error */
int  oracle_lwe_commit(const oracle_lwe* c, const uint64_t* msg, size_t msg_len, uint64_t seed, uint64_t* out);
int  oracle_lwe_verify(const oracle_lwe* c, const uint64_t* comm, size_t comm_len, const uint64_t* msg, size_t msg_len);
int  oracle_lwe_linear_combine(const oracle_lwe* c, const uint64_t* const* comms, const size_t* comm_lens,
                               const uint64_t* coeffs, size_t count, uint64_t* out);
/* the metric workload (config 3): u = INTT(A^T NTT(r)) + e1 ; r,e1,u are [k][n], A_hat is [k][k][n] */
void oracle_lwe_public_matrix(const oracle_lwe* c, uint64_t* a_hat);
void oracle_mlwe_matvec(const oracle_ntt* t, uint32_t k, const uint64_t* a_hat, const uint64_t* r,
                        const uint64_t* e1, uint64_t* u);

/* ---------- prover-side polynomial path (lsr_prover_oracle.c; rust-api/lambda-snark/src/{ntt,r1cs}.rs) ---------- */
uint64_t oracle_prover_modulus(void);
uint64_t oracle_prover_root_2_32(void);
uint64_t oracle_root_of_unity(uint64_t n, uint64_t q, uint64_t root_2_32);
int      oracle_cyclic_ntt_forward(uint64_t* data, size_t n, uint64_t q, uint64_t omega);
int      oracle_cyclic_ntt_inverse(uint64_t* data, size_t n, uint64_t q, uint64_t omega);
void     oracle_cyclic_ntt_naive(const uint64_t* in, uint64_t* out, size_t n, uint64_t q, uint64_t omega);
uint64_t oracle_eval_poly(const uint64_t* poly, size_t len, uint64_t x, uint64_t q);
void     oracle_sparse_mul_vec(const uint32_t* rows, const uint32_t* cols, const uint64_t* vals, size_t n_entries, const uint64_t* v,
                               uint64_t q, uint64_t* out, size_t n_rows);
size_t   oracle_quotient_ntt_path(const uint64_t* a_evals, const uint64_t* b_evals, const uint64_t* c_evals, size_t m,
                                  uint64_t q, uint64_t root_2_32, uint64_t* quotient);

#ifdef __cplusplus
}
#endif
#endif
