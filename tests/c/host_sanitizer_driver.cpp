// Native driver for the library's HOST-only code (number theory, table construction, CDT table, SHA3 transcript)
// under ASan + UBSan; compiled with g++ from lsr_host_math.cpp and lsr_transcript.cpp, no HIP involved.
#include <cstdint>
#include <cstdio>
#include <vector>

#include "lambda_snark/types.h"
#include "lsr_host_math.hpp"

extern "C" int lsr_fs_challenge(const uint64_t*, size_t, const LweCommitment*, uint64_t, uint64_t*, uint8_t*) noexcept;

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "FAIL %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main() {
    using namespace lsr;
    CHECK(is_prime_u64(17592169062401ull) && !is_prime_u64(17592186044417ull) && is_prime_u64(2) && !is_prime_u64(1));
    CHECK(minimal_primitive_root_2n(12289, 256) == 3 && minimal_primitive_root_2n(17592169062401ull, 4096) == 1299579534ull);
    CHECK(minimal_primitive_root_2n(17592169062401ull, 65536) == 0);
    int logn = 0;
    CHECK(ntt_params_valid(17592182243329ull, 65536, &logn) && logn == 16 && !ntt_params_valid(12289, 3, &logn) && !ntt_params_valid(12289, 0, &logn));
    const TwiddleTables t = build_twiddles(12289, 256, 8, 3);
    CHECK(t.fwd.size() == 256 && t.fwd[0] == 1 && mulmod(t.fwd[37], t.inv[37], 12289) == 1 && mulmod(t.n_inv, 256, 12289) == 1);
    CHECK(gaussian_cdf(3.19).size() == 40 && gaussian_cdf(0.0).empty() && gaussian_cdf(3.19).back() == ~0ull);
    CHECK(select_commit_modulus(12289, 4096) == 17592169062401ull && select_commit_modulus(5, 65536) == 17592182243329ull && select_commit_modulus(5, 3) == 0);
    CHECK(plain_modulus_for(4096) == 1032193 && plain_modulus_for(65536) == 786433);
    CHECK(largest_prime_congruent_one(0, 20) == 0 && largest_prime_congruent_one(8192, 1) == 0);
    // prover path: cyclic tables over the Goldilocks field and a 44-bit field
    const uint64_t gq = kProverModulus, w8 = prover_root_of_unity(gq, 8);
    CHECK(w8 != 0 && powmod(w8, 4, gq) == gq - 1 && prover_root_of_unity(gq, 12) == 0 && prover_root_of_unity(12289, 8) == 0);
    CHECK(prover_root_of_unity(gq, 1ull << 32) == kProverRoot2_32 && prover_root_of_unity(gq, 1ull << 33) == 0);
    CHECK(cyclic_params_valid(gq, 8, w8, &logn) && logn == 3 && !cyclic_params_valid(gq, 8, 5, &logn) && !cyclic_params_valid(gq, 6, w8, &logn));
    CHECK(!cyclic_params_valid(gq - 2, 8, w8, &logn) && !cyclic_params_valid(gq, 1u << 18, w8, &logn) && !cyclic_params_valid(gq, 8, gq, &logn));
    for (uint32_t n : {2u, 8u, 131072u}) {
        const uint64_t w = prover_root_of_unity(gq, n);
        int ln = 0;
        CHECK(cyclic_params_valid(gq, n, w, &ln));
        const TwiddleTables c = build_cyclic_twiddles(gq, n, ln, w);
        CHECK(c.fwd.size() == n && c.fwd[0] == 1 && c.fwd[1] == 1 && mulmod(c.n_inv, n, gq) == 1);
        for (uint32_t i = 0; i < n; i += (n > 64 ? 4099 : 1)) CHECK(mulmod(c.fwd[i], c.inv[i], gq) == 1);
        if (n >= 8)   // stage-order layout: entry m + i = w^((n/2m) bitrev(i))
            CHECK(c.fwd[3] == powmod(w, n / 4, gq) && c.fwd[5] == powmod(w, n / 4, gq) && c.fwd[6] == powmod(w, n / 8, gq) &&
                  c.fwd[7] == powmod(w, 3 * (n / 8), gq));
    }
    std::vector<uint64_t> words(12293, 0x0123456789abcdefull);
    LweCommitment com{words.data(), words.size()};
    uint64_t inputs[] = {1, 471}, alpha = 0;
    uint8_t digest[32];
    CHECK(lsr_fs_challenge(inputs, 2, &com, 17592186044417ull, &alpha, digest) == 0 && alpha < 17592186044417ull);
    CHECK(lsr_fs_challenge(nullptr, 0, &com, 12289, &alpha, nullptr) == 0);
    CHECK(lsr_fs_challenge(nullptr, 2, &com, 12289, &alpha, nullptr) == -1);
    std::puts("host sanitizer driver ok");
    return 0;
}
