/*
 * lambda_snark/batch.h — ADDITIVE entry points (not in the reference): batched and device-resident
 * forms of the hot path, seeded/deterministic variants, and introspection.
 *
 * Why they exist: the reference C-ABI moves ONE polynomial through host pointers per call
 * (cpp-core/include/lambda_snark/ntt.h:55-92, commitment.h:58-63); a GPU cannot reach the headline
 * metric (batched degree-2^16 NTTs/s, commits/s) through that, so SURVEY.md §8(b) asks for batched
 * twins.  Every symbol here is `lsr_`-prefixed or `*_batch`-suffixed; the reference symbols keep their
 * exact semantics.  All are extern "C", plain pointers and sizes; `stream` is a hipStream_t passed as
 * void* (NULL = the default stream of the context's device).
 */
#pragma once

#include "lambda_snark/commitment.h"
#include "lambda_snark/ntt.h"
#include "lambda_snark/types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- runtime ---------------- */
/* number of visible HIP devices (0 when there is no GPU / no driver) */
int lsr_device_count(void) LSR_NOEXCEPT;
/* last error message of the calling thread ("" if none) */
const char* lsr_last_error(void) LSR_NOEXCEPT;
/* library / kernel-variant description, e.g. "lambda_snark_core hip gfx950 r3 (...)" */
const char* lsr_version(void) LSR_NOEXCEPT;

/* ---------------- NTT: contexts on a chosen device ---------------- */
/* like ntt_context_create, on HIP device `device` (-1 = LAMBDA_SNARK_DEVICE env, else LOCAL_RANK, else 0; an index that is
 * not a visible device is an error — NULL and a message — never wrapped around onto another rank's GPU) */
NttContext* lsr_ntt_context_create_on(uint64_t q, uint32_t n, int device) LSR_NOEXCEPT;
int      lsr_ntt_context_device(const NttContext* ctx) LSR_NOEXCEPT;
uint64_t lsr_ntt_context_root(const NttContext* ctx) LSR_NOEXCEPT;   /* psi */
/* 1 if the context computes with the exact FP64-FMA Barrett kernels (q < 2^45), 0 for u64 Shoup */
int      lsr_ntt_context_uses_f64(const NttContext* ctx) LSR_NOEXCEPT;
/* force the arithmetic flavour of FUTURE contexts: 0 auto, 1 u64 Shoup always (testing) */
void     lsr_set_arith_mode(int mode) LSR_NOEXCEPT;

/* ---------------- NTT: batched, host buffers ([batch][n] contiguous) ---------------- */
int ntt_forward_batch(const NttContext* ctx, uint64_t* polys, size_t batch) LSR_NOEXCEPT;
int ntt_inverse_batch(const NttContext* ctx, uint64_t* polys, size_t batch) LSR_NOEXCEPT;
/* result/a/b are [batch][n]; returns 0 / -1 (unlike the void single-poly form) */
int ntt_mul_pointwise_batch(const NttContext* ctx, uint64_t* result, const uint64_t* a, const uint64_t* b,
                            size_t batch) LSR_NOEXCEPT;

/* ---------------- NTT: batched, device-resident, asynchronous on `stream` ---------------- */
int lsr_ntt_forward_batch_device(const NttContext* ctx, uint64_t* d_polys, size_t batch, void* stream) LSR_NOEXCEPT;
int lsr_ntt_inverse_batch_device(const NttContext* ctx, uint64_t* d_polys, size_t batch, void* stream) LSR_NOEXCEPT;
int lsr_ntt_mul_pointwise_device(const NttContext* ctx, uint64_t* d_result, const uint64_t* d_a,
                                 const uint64_t* d_b, size_t count, void* stream) LSR_NOEXCEPT;

/* ---------------- Gaussian sampler: seeded / device ---------------- */
/* sample i of object (seed, domain, index) uses ChaCha20 stream word i (low bit: sign; upper 63 bits: the uniform
 * value compared with the CDT table at 63-bit precision);
 * output = two's-complement int64 like sample_gaussian. Host buffer. */
int lsr_sample_gaussian_seeded(uint64_t* output, size_t len, double sigma, uint64_t seed, uint32_t domain,
                               uint64_t index) LSR_NOEXCEPT;
/* CDT table exactly as cpp-core/src/utils.cpp:26-75 (host long double); returns entry count or 0 */
size_t lsr_gaussian_cdf(double sigma, uint64_t* cdf, size_t cap) LSR_NOEXCEPT;

/* ---------------- commitment: seeded contexts, batches, the metric workload ---------------- */
/* lwe_context_create with an explicit key seed and device (-1 = default).  key_seed != 0: every key of the context is
 * derived from it — reproducible contexts for tests and for replicating ONE context on several devices; such a context is
 * only as secret as the 64-bit seed.  key_seed == 0: 256-bit OS entropy, exactly lwe_context_create. */
LweContext* lsr_lwe_context_create_seeded(const PublicParams* params, uint64_t key_seed, int device) LSR_NOEXCEPT;
/* The modulus to put into PublicParams.modulus for a context whose lwe_linear_combine has the reference's range
 * (commitment.cpp:88-96,247-266: any coefficient below the plaintext modulus): the largest 60-bit prime = 1 (mod 2 ring_degree).
 * A default context (any modulus no transform can use: 2^44 + 1, 12289 ...) commits under a 44-bit prime on the FP64 kernels and
 * refuses combinations whose centred coefficients sum beyond ~800 (noise budget); under the 60-bit prime the bound is ~2^25 and
 * the u64 Harvey/Shoup kernels run (about 1.5x the time per transform).  0 for an unsupported ring_degree. */
uint64_t lsr_lwe_wide_modulus(uint32_t ring_degree) LSR_NOEXCEPT;
uint64_t lsr_lwe_modulus(const LweContext* ctx) LSR_NOEXCEPT;         /* internal q actually used */
uint64_t lsr_lwe_plain_modulus(const LweContext* ctx) LSR_NOEXCEPT;   /* t */
uint32_t lsr_lwe_ring_degree(const LweContext* ctx) LSR_NOEXCEPT;
uint32_t lsr_lwe_module_rank(const LweContext* ctx) LSR_NOEXCEPT;
size_t   lsr_lwe_commitment_words(const LweContext* ctx) LSR_NOEXCEPT; /* LweCommitment.len */
const NttContext* lsr_lwe_ntt_context(const LweContext* ctx) LSR_NOEXCEPT;
/* copy the public matrix A_hat ([k][k][n], NTT domain) to a host buffer */
int lsr_lwe_public_matrix(const LweContext* ctx, uint64_t* a_hat) LSR_NOEXCEPT;

/* `batch` commitments in one device pass.  messages = [batch][msg_len]; seeds[batch] (0 = fresh; semantics of lwe_commit);
 * out[batch] receives commitments to be freed with lwe_commitment_free.  0 / -1. */
int lwe_commit_batch(LweContext* ctx, const uint64_t* messages, size_t msg_len, size_t batch,
                     const uint64_t* seeds, LweCommitment** out) LSR_NOEXCEPT;

/* The same `batch` commitments written back to back into ONE caller-owned host array
 * out_words[batch][lsr_lwe_commitment_words(ctx)] — row i holds exactly the words lwe_commit_batch would put in
 * out[i]->data (data[0] = payload byte length, commitment.cpp:44-60) — with no per-commitment allocation: the rows are
 * assembled on the device and come back in one copy.  0 / -1. */
int lsr_lwe_commit_batch_flat(LweContext* ctx, const uint64_t* messages, size_t msg_len, size_t batch,
                              const uint64_t* seeds, uint64_t* out_words) LSR_NOEXCEPT;
/* The same with the rows left in DEVICE memory d_out_words[batch][lsr_lwe_commitment_words(ctx)] on the context's device
 * (messages and seeds are still host arrays); returns after the rows are complete.  For chaining with
 * lsr_fs_challenge_batch_device before the rows travel to the host. */
int lsr_lwe_commit_batch_flat_device(LweContext* ctx, const uint64_t* messages, size_t msg_len, size_t batch,
                                     const uint64_t* seeds, uint64_t* d_out_words) LSR_NOEXCEPT;

/* Whole commitments without a byte of host traffic (round 3): the batched, device-resident form of lwe_commit
 * (cpp-core/src/commitment.cpp:138-164, contract cpp-core/include/lambda_snark/commitment.h:43-63) and of lwe_verify_opening
 * (commitment.cpp:200-232, commitment.h:80-99).  lsr_lwe_commit_keys derives, on the host, the per-commitment
 * 256-bit stream keys exactly as lwe_commit does (seed != 0: PRF of seed, context id and embedded message; seed == 0: fresh OS
 * entropy) into out_keys[batch][4]; lsr_lwe_commit_rows_device then turns DEVICE arrays d_keys[batch][4] and
 * d_messages[batch][msg_len] into the wire rows d_rows[batch][lsr_lwe_commitment_words(ctx)] — word for word what
 * lsr_lwe_commit_batch_flat returns for the same keys — asynchronously on `stream`.  Contexts with ring_degree 4096 (FP64
 * flavour, rank <= 4, sigma <= ~6.9) run it as ONE launch with one workgroup per commitment: r, e1, e2 are sampled in the
 * lanes, transformed and multiplied in LDS, and only the finished row is written; ring_degree 2^16 / 2^17 sample inside the
 * strided transform rounds (three launches per chunk).  lsr_lwe_pipeline names the path a context takes: "tile", "fused",
 * "fused-matvec" (only the matrix-vector workload is fused), "general".
 * Calls on one context are ordered one behind the other (each waits for the previous call's last kernel), whatever streams the
 * caller passes; the synchronous entry points of the same context (lwe_commit, lwe_verify_opening, lwe_linear_combine, the batch
 * and sharded calls) and lwe_context_free wait for a pending asynchronous call before they touch the context's workspaces.
 * Results are ready when `stream` has drained.  0 / -1.
 * lsr_lwe_commit_keys_device derives the same keys ON THE DEVICE from device-resident messages (seeds: a HOST array, every seed
 * non-zero — seed 0 means fresh OS entropy, which only the host call can serve: -1), asynchronously on `stream`: the host
 * derivation hashes every embedded message word (about 10 us per full-length message at n = 4096 on one core), which for long
 * messages costs several times what the commitments themselves take on the GPU. */
int lsr_lwe_commit_keys(const LweContext* ctx, const uint64_t* messages, size_t msg_len, size_t batch, const uint64_t* seeds,
                        uint64_t* out_keys) LSR_NOEXCEPT;
int lsr_lwe_commit_keys_device(LweContext* ctx, const uint64_t* d_messages, size_t msg_len, size_t batch, const uint64_t* seeds,
                               uint64_t* d_keys, void* stream) LSR_NOEXCEPT;
int lsr_lwe_commit_rows_device(LweContext* ctx, const uint64_t* d_messages, size_t msg_len, size_t batch,
                               const uint64_t* d_keys, uint64_t* d_rows, void* stream) LSR_NOEXCEPT;
/* `count` openings of device-resident rows against device-resident claimed messages (1 <= msg_len <= ring_degree):
 * d_results[i] = 1 / 0 / -1 with the meaning of lwe_verify_opening.  Asynchronous on `stream`.  0 / -1. */
int lsr_lwe_verify_rows_device(const LweContext* ctx, const uint64_t* d_rows, const uint64_t* d_messages, size_t msg_len,
                               size_t count, int* d_results, void* stream) LSR_NOEXCEPT;
const char* lsr_lwe_pipeline(const LweContext* ctx) LSR_NOEXCEPT;

/* `count` openings in one device pass.  messages = [count][msg_len]; results[i] = 1 / 0 / -1 with the meaning of
 * lwe_verify_opening (cpp-core/src/commitment.cpp:200-232) for (commitments[i], messages[i]); NULL entries => -1.
 * Returns 0, or -1 if the call itself failed. */
int lwe_verify_opening_batch(const LweContext* ctx, const LweCommitment* const* commitments, const uint64_t* messages,
                             size_t msg_len, size_t count, int* results) LSR_NOEXCEPT;
/* The same for `count` commitments stored back to back (rows of lsr_lwe_commit_batch_flat): words[count][lsr_lwe_commitment_words]. */
int lsr_lwe_verify_opening_batch_flat(const LweContext* ctx, const uint64_t* words, const uint64_t* messages,
                                      size_t msg_len, size_t count, int* results) LSR_NOEXCEPT;

/* The Module-LWE matrix–vector workload of BASELINE config 3, device-resident:
 *   u_j = INTT( A_hat^T . NTT(r_j) ) + e1_j   for j < batch;   r, e1, u are [batch][k][n] in [0,q).
 * d_e1 != NULL: the blinding residues are read from it (seeds may be NULL).
 * d_e1 == NULL: e1 is sampled on the device, component i of vector j from the raw-seed stream (seeds[j], domain 5, i)
 *               (key = {seed, "LSR1", "STRM"}: a reproducible workload stream, not a commitment's message-bound key);
 *               `seeds` is then a HOST array of `batch` seeds and the call returns after the work has finished.
 * d_r: contexts that qualify for the fused pipeline (FP64 flavour, n = 2^16 or 2^17, rank <= 4) only read it; otherwise it
 * is overwritten with NTT(r) — treat its contents as unspecified after the call.  0 / -1. */
int lsr_mlwe_matvec_batch_device(const LweContext* ctx, uint64_t* d_r, const uint64_t* d_e1, uint64_t* d_u,
                                 size_t batch, const uint64_t* seeds, void* stream) LSR_NOEXCEPT;

/* Field elements wider than the plaintext modulus t (about 2^20).  lwe_commit embeds every message word mod t — the
 * reference's BatchEncoder takes out-of-range words unchecked and lwe_verify_opening compares the decoded slots with the
 * message words as given (cpp-core/src/commitment.cpp:152,223-226), so there, as here, a word >= t is bound only through
 * its residue and never opens.  A caller that needs a 44- or 64-bit coefficient bound in full commits to its LIMBS:
 * limbs[i * limbs_per_word + l] = (words[i] >> (l * limb_bits)) & (2^limb_bits - 1), e.g. limb_bits = 16, limbs_per_word = 4
 * (64-bit words) or limb_bits = 15, limbs_per_word = 3 (44-bit field); every limb is < t, the message grows by the factor
 * limbs_per_word (<= ring_degree slots in all) and opens word for word.  Returns the number of limbs (count *
 * limbs_per_word; also when `words` or `limbs` is NULL, for sizing), 0 on invalid limb parameters.  Host only. */
size_t lsr_words_to_limbs(const uint64_t* words, size_t count, unsigned limb_bits, unsigned limbs_per_word,
                          uint64_t* limbs) LSR_NOEXCEPT;

/* The blinding residues alone: d_e1[batch][k][n] in [0,q), component i of vector j from the seeded CDT stream
 * (seeds[j], domain 5, i) — exactly what lsr_mlwe_matvec_batch_device samples when d_e1 == NULL.  `seeds` is a HOST
 * array; the call returns after the samples are complete.  0 / -1. */
int lsr_lwe_sample_blinding_device(const LweContext* ctx, uint64_t* d_e1, size_t batch, const uint64_t* seeds,
                                   void* stream) LSR_NOEXCEPT;

/* Synthetic inputs of SURVEY.md section 8(d): d_out[objects][len], element i of object o = the (i+1)-th output of
 * splitmix64 seeded with seed_base + o, reduced mod q (q = 0: the raw 64-bit word).  Asynchronous on `stream`, launched on
 * the calling thread's current HIP device.  0 / -1. */
int lsr_fill_splitmix_device(uint64_t* d_out, size_t objects, size_t len, uint64_t seed_base, uint64_t q,
                             void* stream) LSR_NOEXCEPT;

/* ---------------- several devices of one node inside ONE call (BASELINE config 4) ----------------
 * Independent polynomials / commitments: the batch is cut into `shards` contiguous slices (lsr_shard_bounds: sizes differ
 * by at most one, earlier shards take the extra), shard g is driven by ctxs[g] on ITS device from its own host thread and
 * stream, and every slice of the result is copied device -> host straight into its place in the caller's single array
 * (allocate it with lsr_host_alloc_pinned for full PCIe speed).  No collective, no device-to-device traffic; the shared
 * resource is host PCIe / DRAM bandwidth.  The reference has no counterpart (single-threaded CPU library; its only
 * concurrency statement is `unsafe impl Send`, rust-api/lambda-snark/src/context.rs:76).  The contexts of a call must be
 * distinct objects; several may live on the same device (that is how a one-GPU box exercises this path). */
void  lsr_shard_bounds(size_t batch, int shards, int index, size_t* first, size_t* count) LSR_NOEXCEPT;
void* lsr_host_alloc_pinned(size_t bytes) LSR_NOEXCEPT;     /* NULL on failure */
void  lsr_host_free_pinned(void* p) LSR_NOEXCEPT;
/* the same commitment context (same keys: same A_hat, s, b_hat, context id) on another device; commitments made by either
 * replica verify and combine under the other.  NULL on failure. */
LweContext* lsr_lwe_context_replicate(const LweContext* ctx, int device) LSR_NOEXCEPT;
/* ntt_forward_batch / ntt_inverse_batch over `shards` contexts of the same (q, n): polys is ONE host array [batch][n] */
int lsr_ntt_forward_batch_sharded(const NttContext* const* ctxs, int shards, uint64_t* polys, size_t batch) LSR_NOEXCEPT;
int lsr_ntt_inverse_batch_sharded(const NttContext* const* ctxs, int shards, uint64_t* polys, size_t batch) LSR_NOEXCEPT;
/* lsr_lwe_commit_batch_flat over replicas of one context: host messages [batch][msg_len], seeds [batch] (or NULL), rows
 * written to out_words[batch][lsr_lwe_commitment_words] — bit-identical to the one-device call. */
int lsr_lwe_commit_batch_flat_sharded(LweContext* const* ctxs, int shards, const uint64_t* messages, size_t msg_len,
                                      size_t batch, const uint64_t* seeds, uint64_t* out_words) LSR_NOEXCEPT;
/* The config-4 workload: u_j = INTT(A_hat^T NTT(r_j)) + e1_j with DEVICE-resident inputs per shard and a HOST gather.
 * d_r[g], d_e1[g]: arrays on ctxs[g]'s device holding rows [first_g, first_g + count_g) of the batch ([count_g][k][n]);
 * host_u: one array [batch][k][n].  Each shard works through its slice in pieces: the device -> host copy of one piece runs on
 * a copy stream under the kernels of the next, so a shard takes about max(kernels, gather) plus one piece, not their sum.
 * seconds (optional, 2 doubles): over the shards, [0] the longest kernel time of a slice, [1] the longest wall time until a
 * slice's last byte is in host memory.  _stats: the same two figures for every shard, per_shard[shards][2].  0 / -1. */
int lsr_mlwe_matvec_batch_sharded(LweContext* const* ctxs, int shards, uint64_t* const* d_r, const uint64_t* const* d_e1,
                                  size_t batch, uint64_t* host_u, double* seconds) LSR_NOEXCEPT;
int lsr_mlwe_matvec_batch_sharded_stats(LweContext* const* ctxs, int shards, uint64_t* const* d_r,
                                        const uint64_t* const* d_e1, size_t batch, uint64_t* host_u,
                                        double* per_shard) LSR_NOEXCEPT;

/* ---------------- Fiat–Shamir consumer of the commitment words (host, no GPU needed) ---------------- */
/* The transcript of rust-api/lambda-snark/src/challenge.rs:102-134: SHA3-256 over "LAMBDA-SNARK-R-FS-v1", the
 * public inputs and ALL commitment words (each length-prefixed, little-endian); alpha = LE64(h[0..8]) mod modulus.
 * hash32 may be NULL.  0 / -1. */
int lsr_fs_challenge(const uint64_t* public_inputs, size_t n_inputs, const LweCommitment* commitment, uint64_t modulus,
                     uint64_t* alpha, uint8_t* hash32) LSR_NOEXCEPT;
/* `count` transcripts at once: commitments as rows words[count][words_per_commitment] (lsr_lwe_commit_batch_flat), public
 * inputs [count][n_inputs]; alphas[count], hashes32 (optional) [count][32].  SHA3 is sequential inside a transcript and
 * independent across them: a pool of `threads` host threads (0 = up to 16) shares the rows.  0 / -1. */
int lsr_fs_challenge_batch_flat(const uint64_t* public_inputs, size_t n_inputs, const uint64_t* words,
                                size_t words_per_commitment, size_t count, uint64_t modulus, uint64_t* alphas,
                                uint8_t* hashes32, unsigned threads) LSR_NOEXCEPT;
/* The same on device-resident arrays (8-byte aligned), one lane per transcript, asynchronous on `stream`: d_words
 * [count][words_per_commitment] (e.g. from lsr_lwe_commit_batch_flat_device), d_public_inputs [count][n_inputs] (may be a
 * previous call's d_alphas with n_inputs = 1: the second challenge of prove_r1cs, lib.rs:768), d_alphas [count],
 * d_hashes32 (optional) [count][32].  The kernel is launched on the calling thread's current HIP device (where the arrays
 * must live).  The cost is flat up to 65 536 transcripts (a wavefront per SIMD).  0 / -1. */
int lsr_fs_challenge_batch_device(const uint64_t* d_public_inputs, size_t n_inputs, const uint64_t* d_words,
                                  size_t words_per_commitment, size_t count, uint64_t modulus, uint64_t* d_alphas,
                                  uint8_t* d_hashes32, void* stream) LSR_NOEXCEPT;

/* ---------------- host-only number theory (usable without a GPU) ---------------- */
uint64_t lsr_minimal_primitive_root(uint64_t q, uint32_t n) LSR_NOEXCEPT;   /* 0 if none */
uint64_t lsr_select_commit_modulus(uint64_t requested_q, uint32_t n) LSR_NOEXCEPT;
uint64_t lsr_plain_modulus(uint32_t n) LSR_NOEXCEPT;                        /* SEAL Batching(n,20) */

#ifdef __cplusplus
}
#endif
