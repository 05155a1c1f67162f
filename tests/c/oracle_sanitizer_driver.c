/* Native driver that runs every oracle entry point under AddressSanitizer + UndefinedBehaviorSanitizer
 * (CPU only; GPU sanitizers are not available on this pool).  Built and run by tests/test_sanitizers.py. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lsr_oracle.h"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAIL %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main(void) {
    const uint64_t q = 17592169062401ULL;
    const uint32_t n = 1024;
    oracle_ntt* t = oracle_ntt_create(q, n);
    CHECK(t != NULL);
    uint64_t* a = malloc(sizeof(uint64_t) * 3 * n);
    uint64_t* b = malloc(sizeof(uint64_t) * 3 * n);
    oracle_splitmix_fill(1, q, a, 3 * n);
    memcpy(b, a, sizeof(uint64_t) * 3 * n);
    CHECK(oracle_ntt_forward_batch(t, b, 3) == 0);
    CHECK(oracle_ntt_inverse_batch(t, b, 3) == 0);
    CHECK(memcmp(a, b, sizeof(uint64_t) * 3 * n) == 0);
    oracle_ntt_mul_pointwise(t, b, a, a, n);
    uint64_t* naive = malloc(sizeof(uint64_t) * n);
    oracle_ntt* small = oracle_ntt_create(12289, 64);
    oracle_splitmix_fill(2, 12289, naive, 64);
    uint64_t out64[64], in64[64];
    memcpy(in64, naive, sizeof in64);
    oracle_ntt_forward_naive(small, in64, out64);
    CHECK(oracle_ntt_forward(small, in64, 64) == 0 && memcmp(in64, out64, sizeof in64) == 0);
    CHECK(oracle_ntt_create(12289, 3) == NULL && oracle_ntt_create(q, 65536) == NULL);
    /* sampler */
    uint64_t cdf[64];
    CHECK(oracle_gaussian_cdf(3.19, cdf, 64) == 40 && oracle_gaussian_cdf(3.19, cdf, 8) == 0);
    uint64_t g[1001];
    CHECK(oracle_sample_gaussian_seeded(g, 1001, 3.19, 7, 5, 2) == 0);
    CHECK(oracle_sample_gaussian(g, 17, 3.2) == 0 && oracle_sample_gaussian(NULL, 1, 3.2) == -1);
    uint64_t words[19];
    oracle_stream_words(9, 1, 2, 5, words, 19);
    /* commitment */
    oracle_lwe* c = oracle_lwe_create(12289, 256, 2, 3.19, 11);
    CHECK(c != NULL);
    const size_t w = oracle_lwe_commit_words(c);
    uint64_t *c1 = malloc(8 * w), *c2 = malloc(8 * w), *c3 = malloc(8 * w);
    uint64_t m1[] = {1, 2, 3, 4}, m2[] = {5, 6, 7, 8}, comb[] = {17, 22, 27, 32}, cf[] = {2, 3};
    CHECK(oracle_lwe_commit(c, m1, 4, 21, c1) == 0 && oracle_lwe_commit(c, m2, 4, 22, c2) == 0);
    CHECK(oracle_lwe_verify(c, c1, w, m1, 4) == 1 && oracle_lwe_verify(c, c1, w, m2, 4) == 0);
    const uint64_t* both[] = {c1, c2};
    const size_t lens[] = {w, w};
    CHECK(oracle_lwe_linear_combine(c, both, lens, cf, 2, c3) == 0);
    CHECK(oracle_lwe_verify(c, c3, w, comb, 4) == 1);
    CHECK(oracle_lwe_verify(c, c3, 3, comb, 4) == -1);
    uint64_t long_msg[300];
    for (int i = 0; i < 300; ++i) long_msg[i] = (uint64_t)i;
    CHECK(oracle_lwe_commit(c, long_msg, 300, 5, c1) == 0 && oracle_lwe_verify(c, c1, w, long_msg, 256) == 1 && oracle_lwe_verify(c, c1, w, long_msg, 300) == 0);
    free(c1); free(c2); free(c3); free(a); free(b); free(naive);
    oracle_lwe_free(c); oracle_ntt_free(t); oracle_ntt_free(small);
    {   /* prover path: ntt.rs known answers, a quotient with and without remainder, the sparse product */
        const uint64_t gq = oracle_prover_modulus(), g = oracle_prover_root_2_32();
        uint64_t v[8] = {1, 2, 3, 4, 5, 6, 7, 8}, out[8];
        const uint64_t w8 = oracle_root_of_unity(8, gq, g);
        oracle_cyclic_ntt_naive(v, out, 8, gq, w8);
        if (oracle_cyclic_ntt_forward(v, 8, gq, w8) != 0 || v[0] != 36 || memcmp(v, out, sizeof v) != 0) return 1;
        if (oracle_cyclic_ntt_inverse(v, 8, gq, w8) != 0 || v[0] != 1 || v[7] != 8) return 1;
        if (oracle_cyclic_ntt_forward(v, 6, gq, w8) != -1 || oracle_root_of_unity(12, gq, g) != 0) return 1;
        uint64_t a[2] = {2, 6}, b[2] = {3, 4}, c[2] = {6, 24}, quot[2];
        if (oracle_quotient_ntt_path(a, b, c, 2, gq, g, quot) != 1 || quot[0] != 1) return 1;
        c[1] = 25;
        if (oracle_quotient_ntt_path(a, b, c, 2, gq, g, quot) != 0) return 1;
        uint64_t one = 7, zero = 0;
        if (oracle_quotient_ntt_path(&one, &one, &one, 1, gq, g, quot) != 0 || oracle_quotient_ntt_path(&zero, &one, &zero, 1, gq, g, quot) != 1) return 1;
        const uint32_t rows[3] = {0, 1, 0}, cols[3] = {1, 0, 1};
        const uint64_t vals[3] = {5, ~0ull, 2}, z[2] = {3, 4};
        oracle_sparse_mul_vec(rows, cols, vals, 3, z, gq, out, 2);
        if (out[0] != 28 || out[1] != (uint64_t)(((unsigned __int128)(~0ull % gq) * 3) % gq)) return 1;
        if (oracle_eval_poly(v, 8, 2, gq) != 1 + 4 + 12 + 32 + 80 + 192 + 448 + 1024) return 1;
    }
    puts("oracle sanitizer driver ok");
    return 0;
}
