// Host-side number theory and table construction for the MI355X commitment kernel.
// Product code (not the oracle).  Semantics follow SEAL 4.1's NTTTables / numth as used by the
// reference (cpp-core/src/ntt.cpp:46-59) — see SURVEY.md §8(a) row N2.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

namespace lsr {

using u128 = unsigned __int128;

inline uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a * b) % q); }
uint64_t powmod(uint64_t a, uint64_t e, uint64_t q);
inline uint64_t invmod_prime(uint64_t a, uint64_t q) { return powmod(a, q - 2, q); }
bool is_prime_u64(uint64_t n);
inline uint32_t bit_reverse(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1u); x >>= 1; }
    return r;
}

// Numerically smallest primitive (2n)-th root of unity mod prime q; 0 if q != 1 (mod 2n).
uint64_t minimal_primitive_root_2n(uint64_t q, uint32_t n);
// Largest prime p < 2^bits with p == 1 (mod factor); 0 if none above 2^(bits-1).
uint64_t largest_prime_congruent_one(uint64_t factor, int bits);

// Validation of (q, n) exactly as ntt_context_create must behave (ntt.cpp:30-70 + SEAL ctor rules).
bool ntt_params_valid(uint64_t q, uint32_t n, int* logn_out);

// Twiddle tables in "stage order": entry [m + i] (m = 2^s, i < m) is the twiddle of butterfly group i
// of the stage with m groups — forward: psi^bitrev(m+i); inverse: its modular inverse.  Entry [0] = 1.
struct TwiddleTables {
    std::vector<uint64_t> fwd;   // [n]
    std::vector<uint64_t> inv;   // [n]
    uint64_t psi = 0;
    uint64_t n_inv = 0;
};
TwiddleTables build_twiddles(uint64_t q, uint32_t n, int logn, uint64_t psi);

// Cyclic transform of the prover's polynomial path (rust-api/lambda-snark/src/ntt.rs:117-201): same butterfly network,
// other twiddles.  Entry [m + i] = omega^((n/2m) * bitrev_{log m}(i)) — group i of the m-group stage reduces modulo
// X^(n/m) - omega^((n/m) * bitrev(i)), so its twiddle is the square root of that constant; inverse table = inverses.
// Output of the forward network is f(omega^bitrev(i)) at slot i.
constexpr uint64_t kProverModulus = 0xFFFFFFFF00000001ull;       // NTT_MODULUS, lambda-snark-core/src/lib.rs:58
constexpr uint64_t kProverRoot2_32 = 1753635133440165772ull;     // NTT_PRIMITIVE_ROOT, lib.rs:78
// Montgomery form w 2^64 mod NTT_MODULUS of a residue: how the Goldilocks kernels hold their multipliers (gold_mul_mont)
inline uint64_t prover_montgomery(uint64_t w) { return mulmod(w % kProverModulus, 0xFFFFFFFFull, kProverModulus); }   // 2^64 = 2^32 - 1
// n = 2^k in [2, 131072], prime q (q = NTT_MODULUS or q < 2^61), omega of exact order n.
bool cyclic_params_valid(uint64_t q, uint32_t n, uint64_t omega, int* logn_out);
// omega_n of compute_root_of_unity (ntt.rs:226-233); 0 unless q = NTT_MODULUS
uint64_t prover_root_of_unity(uint64_t q, uint64_t n);
TwiddleTables build_cyclic_twiddles(uint64_t q, uint32_t n, int logn, uint64_t omega);

// Shoup quotient floor(w * 2^64 / q)
inline uint64_t shoup_quotient(uint64_t w, uint64_t q) { return (uint64_t)(((u128)w << 64) / q); }

// CDT table of the reference sampler (cpp-core/src/utils.cpp:26-75). Empty on invalid sigma.
std::vector<uint64_t> gaussian_cdf(double sigma);
// Entries of `cdf` a scan at 63-bit precision has to visit: everything from the first entry whose upper 63 bits are all ones on
// can never be below a 63-bit uniform value, so `count += (cdf[k] >> 1 < u)` stops changing there (sigma = 3.19: 30 of 40).  The
// scan loops over k + 1 < entries, hence first saturated index + 1.  Same samples, bit for bit.
uint32_t gaussian_scan_entries(const std::vector<uint64_t>& cdf);

// Commitment parameter selection (DESIGN.md "Commitment definition").
uint64_t select_commit_modulus(uint64_t requested, uint32_t n);
uint64_t plain_modulus_for(uint32_t n);

uint64_t os_entropy64();

}  // namespace lsr
