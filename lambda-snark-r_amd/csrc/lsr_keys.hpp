// Key schedule of the seeded streams (host side).  All randomness of a context or a commitment comes from ChaCha20 streams
// (lsr_sampler.hpp) keyed with 256-bit keys; this file derives those keys.
//
// Why it exists (round-1 advisor findings, cpp-core/src/commitment.cpp:141-157 for contrast: the reference ignores `seed`
// and draws fresh SEAL randomness for every call):
//   * a commitment's blinding (r, e1, e2) must never repeat across DIFFERENT messages or contexts, even when a caller
//     reuses `seed` (the Rust prover commits Q and Q' with one seed, lib.rs:758/905) — otherwise u coincides and
//     v1 - v2 = Delta (m1 - m2) leaks the message difference.  The per-commitment key is therefore a PRF of
//     (seed, context id, message): PRF_seed-chain( universal hash of the embedded message ), see derive_commit_key;
//   * with seed == 0 (and key_seed == 0) the keys are 256 bits of OS entropy, not a 64-bit value;
//   * the public matrix A comes from a public key, the secret s / e from a separate secret key.
// (seed, message, context) -> commitment stays deterministic, which the bit-exact parity tests and the header contract
// (commitment.h:52) need.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

namespace lsr {

struct StreamKey {
    uint32_t w[8];
};

// four-character labels of the key schedule, little endian (shared with the device derivation, lsr_commit_keys.hpp)
constexpr uint32_t key_tag(char a, char b, char c, char d) {
    return (uint32_t)(unsigned char)a | ((uint32_t)(unsigned char)b << 8) | ((uint32_t)(unsigned char)c << 16) | ((uint32_t)(unsigned char)d << 24);
}
constexpr uint32_t kTagLsr2 = key_tag('L', 'S', 'R', '2'), kTagCommit = key_tag('C', 'M', 'I', 'T'), kTagHashPoints = key_tag('H', 'P', 'N', 'T'),
                   kTagCommitKey1 = key_tag('C', 'K', 'Y', '1'), kTagCommitKey2 = key_tag('C', 'K', 'Y', '2');
constexpr uint64_t kHashPrime61 = (1ull << 61) - 1;

// the key as the four little-endian 64-bit words the device kernels read
inline void key_words(const StreamKey& k, uint64_t out[4]) {
    for (int i = 0; i < 4; ++i) out[i] = (uint64_t)k.w[2 * i] | ((uint64_t)k.w[2 * i + 1] << 32);
}
inline std::vector<uint64_t> key_words(const StreamKey& k) {
    std::vector<uint64_t> v(4);
    key_words(k, v.data());
    return v;
}

// RFC 8439 §2.3 block function
void chacha20_block_host(const uint32_t key[8], uint32_t counter, const uint32_t nonce[3], uint32_t out[16]);
// first 8 words of the block keyed by `key` at counter = label, nonce = {a, b, c}: ChaCha20 as a PRF
StreamKey kdf(const StreamKey& key, uint32_t label, uint32_t a, uint32_t b, uint32_t c);
// the key of a raw 64-bit-seed stream {seed_lo, seed_hi, "LSR1", "STRM", 0, 0, 0, 0}: seeded sampler / workload entry
// points (lsr_sample_gaussian_seeded, lsr_mlwe_matvec_batch_device) — reproducible test streams, not commitment-grade
StreamKey expand_seed64(uint64_t seed);
// 32 bytes from the OS (getrandom(2))
StreamKey fresh_key();
void os_entropy_fill(void* dst, size_t bytes);

struct ContextKeys {
    StreamKey pub;     // A_hat
    StreamKey sec;     // s, e
    uint32_t id[4];    // public identifier bound to both; enters every per-commitment key
};
// key_seed != 0: derived from {key_seed, "LSR2", "MSTR"} (reproducible contexts: tests, replicated multi-GPU contexts);
// key_seed == 0: 256 bits of OS entropy each (the reference draws a fresh key per context, commitment.cpp:118-121)
ContextKeys derive_context_keys(uint64_t key_seed);

// Per-commitment stream key for seed != 0.  m = the `copy` message words that are embedded (each taken mod t), i.e. the
// plaintext polynomial.  h_a = sum_i (m_i mod t) x_a^(i+1) mod (2^61 - 1) for two seed-derived secret points x_1, x_2 is a
// universal hash (collision probability <= (n / 2^61)^2 per pair of messages); the key is PRF(seed, id)(h_1) chained with h_2.
StreamKey derive_commit_key(uint64_t seed, const uint32_t id[4], const uint64_t* message, size_t copy, uint64_t t);

}  // namespace lsr
