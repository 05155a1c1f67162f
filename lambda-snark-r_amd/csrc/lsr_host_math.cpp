#include "lsr_host_math.hpp"
#include "lsr_keys.hpp"

#include <cmath>
#include <limits>
#include <random>

namespace lsr {

uint64_t powmod(uint64_t a, uint64_t e, uint64_t q) {
    uint64_t result = 1 % q;
    a %= q;
    for (; e; e >>= 1) {
        if (e & 1) result = mulmod(result, a, q);
        a = mulmod(a, a, q);
    }
    return result;
}

// Deterministic Miller–Rabin: the first twelve primes as witnesses are sufficient below 2^64.
bool is_prime_u64(uint64_t n) {
    if (n < 2) return false;
    constexpr uint64_t witnesses[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    for (uint64_t p : witnesses) {
        if (n == p) return true;
        if (n % p == 0) return false;
    }
    uint64_t odd = n - 1;
    int twos = 0;
    while ((odd & 1) == 0) { odd >>= 1; ++twos; }
    for (uint64_t a : witnesses) {
        uint64_t x = powmod(a, odd, n);
        if (x == 1 || x == n - 1) continue;
        bool witness_of_compositeness = true;
        for (int r = 1; r < twos && witness_of_compositeness; ++r) {
            x = mulmod(x, x, n);
            if (x == n - 1) witness_of_compositeness = false;
        }
        if (witness_of_compositeness) return false;
    }
    return true;
}

uint64_t minimal_primitive_root_2n(uint64_t q, uint32_t n) {
    const uint64_t order = 2ull * n;
    if (q < 3 || (q - 1) % order != 0) return 0;
    const uint64_t cofactor = (q - 1) / order;
    // Any element of exact order 2n: g^cofactor whose n-th power is -1.
    uint64_t gen = 0;
    for (uint64_t g = 2; g < q && g < 100000; ++g) {
        const uint64_t cand = powmod(g, cofactor, q);
        if (powmod(cand, n, q) == q - 1) { gen = cand; break; }
    }
    if (!gen) return 0;
    // The primitive 2n-th roots are exactly the odd powers of gen; take the smallest.
    const uint64_t step = mulmod(gen, gen, q);
    uint64_t smallest = gen, walk = gen;
    for (uint32_t i = 1; i < n; ++i) {
        walk = mulmod(walk, step, q);
        if (walk < smallest) smallest = walk;
    }
    return smallest;
}

uint64_t largest_prime_congruent_one(uint64_t factor, int bits) {
    if (factor == 0 || bits < 2 || bits > 61) return 0;
    const uint64_t top = (1ull << bits) - 1;
    const uint64_t floor_bound = 1ull << (bits - 1);
    for (uint64_t cand = (top / factor) * factor + 1; cand > floor_bound; cand -= factor) {
        if (is_prime_u64(cand)) return cand;
        if (cand < factor) break;
    }
    return 0;
}

bool ntt_params_valid(uint64_t q, uint32_t n, int* logn_out) {
    if (n == 0 || (n & (n - 1)) != 0) return false;   // ntt.cpp:31, :41-44
    if (q < 2 || (q >> 61) != 0) return false;        // seal::Modulus range
    int logn = 0;
    while ((1u << logn) < n) ++logn;
    if (logn < 1 || logn > 17) return false;          // SEAL NTTTables degree range
    if ((q - 1) % (2ull * n) != 0) return false;
    if (!is_prime_u64(q)) return false;
    if (logn_out) *logn_out = logn;
    return true;
}

TwiddleTables build_twiddles(uint64_t q, uint32_t n, int logn, uint64_t psi) {
    TwiddleTables t;
    t.psi = psi;
    t.fwd.assign(n, 1);
    t.inv.assign(n, 1);
    // psi^e for e = 0..n-1 lands at slot bitrev(e); slots with index m+i (m = 2^s) then hold
    // psi^bitrev(m+i), which is what butterfly group i of the m-group stage multiplies by.
    uint64_t power = 1;
    for (uint32_t e = 0; e < n; ++e) {
        t.fwd[bit_reverse(e, logn)] = power;
        power = mulmod(power, psi, q);
    }
    const uint64_t psi_inv = invmod_prime(psi, q);
    power = 1;
    for (uint32_t e = 0; e < n; ++e) {
        t.inv[bit_reverse(e, logn)] = power;   // (psi^-1)^e at the same slot = inverse of fwd entry
        power = mulmod(power, psi_inv, q);
    }
    t.n_inv = invmod_prime(n % q, q);
    return t;
}

bool cyclic_params_valid(uint64_t q, uint32_t n, uint64_t omega, int* logn_out) {
    if (n < 2 || n > 131072 || (n & (n - 1))) return false;
    if (q != kProverModulus && (q >> 61)) return false;
    if (q < 3 || (q - 1) % n != 0 || !is_prime_u64(q)) return false;
    if (omega == 0 || omega >= q || powmod(omega, n / 2, q) != q - 1) return false;   // exact order n
    int logn = 0;
    while ((1u << logn) < n) ++logn;
    if (logn_out) *logn_out = logn;
    return true;
}

uint64_t prover_root_of_unity(uint64_t q, uint64_t n) {
    if (q != kProverModulus || n == 0 || (n & (n - 1)) || n > (1ull << 32)) return 0;
    return powmod(kProverRoot2_32, (1ull << 32) / n, q);
}

TwiddleTables build_cyclic_twiddles(uint64_t q, uint32_t n, int logn, uint64_t omega) {
    TwiddleTables t;
    t.psi = omega;
    t.fwd.assign(n, 1);
    t.inv.assign(n, 1);
    std::vector<uint64_t> pw(n / 2 + 1), pw_inv(n / 2 + 1);     // omega^e, omega^-e for e <= n/2
    const uint64_t omega_inv = invmod_prime(omega, q);
    pw[0] = pw_inv[0] = 1;
    for (uint32_t e = 1; e <= n / 2; ++e) {
        pw[e] = mulmod(pw[e - 1], omega, q);
        pw_inv[e] = mulmod(pw_inv[e - 1], omega_inv, q);
    }
    for (int s = 0; s < logn; ++s) {
        const uint32_t m = 1u << s, unit = n / (2 * m);
        for (uint32_t i = 0; i < m; ++i) {
            const uint32_t e = unit * bit_reverse(i, s);         // < n/2
            t.fwd[m + i] = pw[e];
            t.inv[m + i] = pw_inv[e];
        }
    }
    t.n_inv = invmod_prime(n % q, q);
    return t;
}

std::vector<uint64_t> gaussian_cdf(double sigma) {
    std::vector<uint64_t> cdf;
    if (!(sigma > 0.0) || !std::isfinite(sigma)) return cdf;
    const long double s = static_cast<long double>(sigma);
    long double tail = std::ceil(12.0L * s);
    if (tail < 8.0L) tail = 8.0L;
    const size_t last = static_cast<size_t>(tail);
    if (last > (1u << 20)) return cdf;
    std::vector<long double> mass(last + 1);
    long double total = 0.0L;
    for (size_t k = 0; k <= last; ++k) {
        const long double kk = static_cast<long double>(k) * static_cast<long double>(k);
        long double m = std::exp(-kk / (2.0L * s * s));
        if (k) m *= 2.0L;
        mass[k] = m;
        total += m;
    }
    const long double full = static_cast<long double>(std::numeric_limits<uint64_t>::max());
    cdf.assign(last + 1, 0);
    if (total == 0.0L) {
        cdf[last] = std::numeric_limits<uint64_t>::max();
        return cdf;
    }
    const long double scale = full / total;
    long double running = 0.0L;
    for (size_t k = 0; k <= last; ++k) {
        running += mass[k];
        const long double v = running * scale;
        cdf[k] = v >= full ? std::numeric_limits<uint64_t>::max() : (v <= 0.0L ? 0 : static_cast<uint64_t>(v));
    }
    cdf[last] = std::numeric_limits<uint64_t>::max();
    return cdf;
}

uint32_t gaussian_scan_entries(const std::vector<uint64_t>& cdf) {
    size_t first = 0;
    while (first < cdf.size() && (cdf[first] >> 1) != 0x7FFFFFFFFFFFFFFFull) ++first;
    return static_cast<uint32_t>(std::min(cdf.size(), first + 1));
}

uint64_t select_commit_modulus(uint64_t requested, uint32_t n) {
    if (n < 2 || (n & (n - 1)) != 0 || n > 131072) return 0;
    const bool usable = requested >= (1ull << 40) && requested < (1ull << 61) &&
                        (requested - 1) % (2ull * n) == 0 && is_prime_u64(requested);
    if (usable) return requested;
    if (n <= 4096) return 17592169062401ull;   // the reference's 44-bit NTT prime (r1cs.rs:527)
    return largest_prime_congruent_one(2ull * n, 44);
}

uint64_t plain_modulus_for(uint32_t n) {
    if (n < 2 || (n & (n - 1)) != 0 || n > 131072) return 0;
    return largest_prime_congruent_one(2ull * n, 20);   // SEAL PlainModulus::Batching(n, 20)
}

uint64_t os_entropy64() {
    uint64_t v = 0;
    os_entropy_fill(&v, sizeof v);
    return v;
}

}  // namespace lsr
