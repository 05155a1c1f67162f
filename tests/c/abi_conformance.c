/*
 * Dependency-free conformance runner for the C-ABI: the assertions of the reference's gtest files
 *   cpp-core/tests/test_ntt.cpp, cpp-core/tests/test_commitment.cpp, cpp-core/tests/test_utils.cpp
 * and of rust-api/lambda-snark-sys/src/lib.rs:24-43, restated in plain C11 (gtest is not available and the
 * point is to show that the headers are valid C and that a non-Python caller links and runs).
 * Build: gcc -std=c11 -Iinclude tests/c/abi_conformance.c -Llambda-snark-r_amd/lib -llambda_snark_core -lm
 * Exit status 0 = every check passed.  With `--no-gpu` it checks only what must hold without a device.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lambda_snark/batch.h"
#include "lambda_snark/commitment.h"
#include "lambda_snark/ntt.h"
#include "lambda_snark/prover.h"
#include "lambda_snark/types.h"
#include "lambda_snark/utils.h"

static int failures = 0, checks = 0;
#define CHECK(cond)                                                                      \
    do {                                                                                 \
        ++checks;                                                                        \
        if (!(cond)) { ++failures; fprintf(stderr, "FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); } \
    } while (0)

static void test_without_device(void) {
    uint64_t buf[8] = {0};
    CHECK(ntt_forward(NULL, buf, 8) == -1);
    CHECK(lwe_context_create(NULL) == NULL);                 /* lambda-snark-sys lib.rs:27-32 */
    ntt_context_free(NULL);
    lwe_context_free(NULL);
    lwe_commitment_free(NULL);
    CHECK(lwe_commit(NULL, NULL, 0, 0) == NULL);
    CHECK(sample_gaussian(NULL, 16, 3.2) == -1);             /* test_utils.cpp:29 */
    CHECK(sample_gaussian(buf, 0, 3.2) == -1);
    CHECK(sample_gaussian(buf, 8, 0.0) == -1);
    CHECK(sample_gaussian(buf, 8, INFINITY) == -1);
    CHECK(ntt_context_create(17592169062401ULL, 65536) == NULL);   /* SURVEY.md F5 */
    CHECK(sizeof(PublicParams) == 32);
    CHECK(sizeof(LweCommitment) == 16 && sizeof(LweOpening) == 16);
    CHECK(PROFILE_SCALAR_A == 0 && PROFILE_RING_B == 1 && LAMBDA_SNARK_ERR_CRYPTO_FAILED == 4);
    CHECK(lsr_plain_modulus(4096) == 1032193);
    CHECK(lsr_quotient_plan_create(3, -1) == NULL && lsr_cyclic_ntt_context_create(12289, 8, 5, -1) == NULL);
    CHECK(lsr_quotient_batch(NULL, NULL, NULL, NULL, 1, NULL, NULL) == -1);
    CHECK(lsr_lwe_commit_batch_flat(NULL, buf, 1, 1, NULL, buf) == -1);
}

/* cpp-core/tests/test_ntt.cpp */
static void test_ntt(void) {
    const uint64_t q = 12289;
    const uint32_t n = 256;
    NttContext* ctx = ntt_context_create(q, n);              /* SetUp, :13-17 */
    CHECK(ctx != NULL);
    if (!ctx) return;
    uint64_t ones[256], orig[256] = {1, 2, 3, 4, 5, 6, 7, 8}, x[256], a[256], b[256], r[256];
    for (uint32_t i = 0; i < n; ++i) ones[i] = 1;
    CHECK(ntt_forward(ctx, ones, n) == 0);                   /* ForwardNttBasic :33-38 */
    for (uint32_t i = 0; i < n; ++i) ones[i] = 1;
    CHECK(ntt_inverse(ctx, ones, n) == 0);                   /* InverseNttBasic :40-45 */
    memcpy(x, orig, sizeof x);
    CHECK(ntt_forward(ctx, x, n) == 0);
    CHECK(ntt_inverse(ctx, x, n) == 0);
    CHECK(memcmp(x, orig, sizeof x) == 0);                   /* ForwardInverseIdentity :47-68 */
    for (uint32_t i = 0; i < n; ++i) { a[i] = 2; b[i] = 3; r[i] = 0; }
    ntt_mul_pointwise(ctx, r, a, b, n);
    for (uint32_t i = 0; i < n; ++i) CHECK(r[i] == 6);       /* PointwiseMultiplication :70-81 */
    CHECK(ntt_forward(NULL, x, n) == -1);                    /* NullPointerHandling :83-90 */
    CHECK(ntt_forward(ctx, NULL, n) == -1);
    CHECK(ntt_forward(ctx, x, n / 2) == -1);                 /* ntt.cpp:81 */
    ntt_context_free(ctx);
    ntt_context_free(NULL);
}

/* cpp-core/tests/test_commitment.cpp */
static void test_commitment(void) {
    PublicParams params;
    params.profile = PROFILE_RING_B;
    params.security_level = 128;
    params.modulus = 12289;
    params.ring_degree = 4096;
    params.module_rank = 2;
    params.sigma = 3.19;
    LweContext* ctx = lwe_context_create(&params);           /* SetUp :12-23 */
    CHECK(ctx != NULL);
    if (!ctx) return;
    {
        uint64_t message[] = {1, 2, 3, 4};
        LweCommitment* comm = lwe_commit(ctx, message, 4, 0x1234);   /* CommitBasic :37-47 */
        CHECK(comm != NULL && comm->len > 0 && comm->data != NULL);
        lwe_commitment_free(comm);
    }
    {
        uint64_t m1[] = {1, 2, 3}, m2[] = {4, 5, 6};
        LweCommitment* c1 = lwe_commit(ctx, m1, 3, 0);       /* CommitBinding :49-75 */
        LweCommitment* c2 = lwe_commit(ctx, m2, 3, 0);
        CHECK(c1 && c2 && c1->len > 0 && c2->len > 0);
        CHECK(c1 && c2 && memcmp(c1->data, c2->data, c1->len * 8) != 0);
        lwe_commitment_free(c1); lwe_commitment_free(c2);
        c1 = lwe_commit(ctx, m1, 3, 0x1234);                 /* CommitDifferentMessages :77-100 */
        c2 = lwe_commit(ctx, m2, 3, 0x1234);
        CHECK(c1 && c2 && memcmp(c1->data, c2->data, (c1->len < c2->len ? c1->len : c2->len) * 8) != 0);
        lwe_commitment_free(c1); lwe_commitment_free(c2);
    }
    CHECK(lwe_commit(NULL, NULL, 0, 0) == NULL);             /* NullPointerHandling :102-113 */
    CHECK(lwe_commit(ctx, NULL, 10, 0) == NULL);
    lwe_commitment_free(NULL);
    {
        uint64_t message[] = {7, 11, 13, 17}, wrong[] = {7, 11 ^ 1, 13, 17}, randomness = 0;
        LweCommitment* comm = lwe_commit(ctx, message, 4, 0);
        LweOpening opening = {&randomness, 1};
        CHECK(lwe_verify_opening(ctx, comm, message, 4, &opening) == 1);   /* VerifyOpeningMatchesMessage :115-132 */
        CHECK(lwe_verify_opening(ctx, comm, wrong, 4, &opening) == 0);
        LweCommitment* twin = lwe_commitment_clone(comm);
        CHECK(twin && twin->len == comm->len && memcmp(twin->data, comm->data, comm->len * 8) == 0);
        lwe_commitment_free(twin);
        lwe_commitment_free(comm);
    }
    {
        uint64_t m1[4] = {1, 2, 3, 4}, m2[4] = {5, 6, 7, 8}, coeffs[] = {2, 3}, expected[4], randomness = 0;
        LweCommitment* c1 = lwe_commit(ctx, m1, 4, 0);
        LweCommitment* c2 = lwe_commit(ctx, m2, 4, 0);
        const LweCommitment* inputs[] = {c1, c2};
        LweCommitment* combined = lwe_linear_combine(ctx, inputs, coeffs, 2);   /* LinearCombination… :134-166 */
        CHECK(combined != NULL);
        for (int i = 0; i < 4; ++i) expected[i] = coeffs[0] * m1[i] + coeffs[1] * m2[i];
        LweOpening opening = {&randomness, 1};
        CHECK(lwe_verify_opening(ctx, combined, expected, 4, &opening) == 1);
        expected[0] += 1;
        CHECK(lwe_verify_opening(ctx, combined, expected, 4, &opening) == 0);
        lwe_commitment_free(c1); lwe_commitment_free(c2); lwe_commitment_free(combined);
    }
    {   /* the batched twins a C caller would use (batch.h): the same commitment as a row of one array, opened in one pass; the
           reference's parameters take the one-workgroup-per-commitment path */
        enum { B = 3, ML = 4 };
        uint64_t msgs[B * ML] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12}, seeds[B] = {0x1234, 0x1235, 0x1236}, keys[B * 4], keys2[B * 4];
        const size_t words = lsr_lwe_commitment_words(ctx);
        CHECK(words == 5 + 3 * 4096);
        CHECK(strcmp(lsr_lwe_pipeline(ctx), "tile") == 0);
        uint64_t* rows = (uint64_t*)malloc(B * words * sizeof(uint64_t));
        int verdicts[B] = {9, 9, 9};
        CHECK(rows != NULL && lsr_lwe_commit_batch_flat(ctx, msgs, ML, B, seeds, rows) == 0);
        LweCommitment* single = lwe_commit(ctx, msgs + ML, ML, seeds[1]);
        CHECK(single != NULL && single->len == words && memcmp(single->data, rows + words, words * sizeof(uint64_t)) == 0);
        lwe_commitment_free(single);
        CHECK(rows[0] == 8 * (words - 1));                      /* data[0] = payload byte length, commitment.cpp:44-60 */
        msgs[2 * ML] ^= 1;                                       /* third claim wrong */
        CHECK(lsr_lwe_verify_opening_batch_flat(ctx, rows, msgs, ML, B, verdicts) == 0);
        CHECK(verdicts[0] == 1 && verdicts[1] == 1 && verdicts[2] == 0);
        msgs[2 * ML] ^= 1;
        CHECK(lsr_lwe_commit_keys(ctx, msgs, ML, B, seeds, keys) == 0 && lsr_lwe_commit_keys(ctx, msgs, ML, B, seeds, keys2) == 0);
        CHECK(memcmp(keys, keys2, sizeof keys) == 0);           /* seed != 0: deterministic in (seed, context, message) */
        seeds[0] = 0;
        CHECK(lsr_lwe_commit_keys(ctx, msgs, ML, B, seeds, keys2) == 0 && memcmp(keys, keys2, 32) != 0);   /* seed == 0: fresh entropy */
        CHECK(lsr_lwe_commit_keys_device(ctx, NULL, 0, B, seeds, (uint64_t*)keys2, NULL) == -1);           /* ... which the device variant refuses */
        free(rows);
    }
    lwe_context_free(ctx);
}

/* cpp-core/tests/test_utils.cpp */
static void test_sampler(void) {
    enum { kSamples = 4096 };
    static uint64_t buffer[kSamples];
    CHECK(sample_gaussian(buffer, kSamples, 3.2) == 0);      /* EmpiricalMomentsWithinBounds :35-70 */
    double mean = 0.0, m2 = 0.0;
    long positives = 0, negatives = 0;
    for (int i = 0; i < kSamples; ++i) {
        const int64_t value = (int64_t)buffer[i];
        const double xv = (double)value, delta = xv - mean;
        mean += delta / (double)(i + 1);
        m2 += delta * (xv - mean);
        positives += value > 0;
        negatives += value < 0;
    }
    const double sigma_emp = sqrt(m2 / (double)(kSamples - 1));
    CHECK(fabs(mean) < 0.5);
    CHECK(fabs(sigma_emp - 3.2) < 0.8);
    CHECK(positives > kSamples / 4 && negatives > kSamples / 4);
    CHECK(labs(positives - negatives) < kSamples / 5);
}

/* rust-api/lambda-snark/src/ntt.rs tests + r1cs.rs:1706-1721, from plain C */
static void test_prover_path(void) {
    const uint64_t q = lsr_prover_modulus();
    CHECK(q == 18446744069414584321ULL && lsr_prover_root_of_unity(2) == q - 1);
    NttContext* t2 = lsr_cyclic_ntt_context_create(q, 2, 0, -1);
    NttContext* t8 = lsr_cyclic_ntt_context_create(q, 8, 0, -1);
    CHECK(t2 != NULL && t8 != NULL);
    if (t2 && t8) {
        uint64_t v2[2] = {1, 2}, v8[8] = {1, 2, 3, 4, 5, 6, 7, 8};
        CHECK(lsr_cyclic_ntt_forward_batch(t2, v2, 1) == 0 && v2[0] == 3 && v2[1] == q - 1);     /* ntt.rs:293-296 */
        CHECK(lsr_cyclic_ntt_inverse_batch(t2, v2, 1) == 0 && v2[0] == 1 && v2[1] == 2);
        CHECK(lsr_cyclic_ntt_forward_batch(t8, v8, 1) == 0 && v8[0] == 36);                      /* ntt.rs:327 */
        CHECK(lsr_cyclic_ntt_inverse_batch(t8, v8, 1) == 0 && v8[0] == 1 && v8[7] == 8);
    }
    ntt_context_free(t2);
    ntt_context_free(t8);
    LsrQuotientPlan* plan = lsr_quotient_plan_create(2, -1);
    CHECK(plan != NULL && lsr_quotient_plan_size(plan) == 2);
    if (plan) {
        /* two gates 2*3 = 6, 6*4 = 24 (r1cs.rs:1090-1113), then a wrong product */
        uint64_t a[4] = {2, 6, 2, 6}, b[4] = {3, 4, 3, 4}, c[4] = {6, 24, 6, 25}, quot[4];
        uint32_t len[2] = {99, 99};
        CHECK(lsr_quotient_batch(plan, a, b, c, 2, quot, len) == 0);
        CHECK(len[0] >= 1 && len[0] <= 2 && len[1] == 0);
        /* interpolants on the domain {1, -1}: A = 4 - 2X, B = (7 - X)/2, C = 15 - 9X  =>  A B - C = X^2 - 1  =>  Q = 1 */
        CHECK(len[0] == 1 && quot[0] == 1);
    }
    lsr_quotient_plan_free(plan);
    lsr_quotient_plan_free(NULL);
}

int main(int argc, char** argv) {
    const int no_gpu = argc > 1 && strcmp(argv[1], "--no-gpu") == 0;
    test_without_device();
    if (no_gpu) {
        CHECK(lsr_device_count() > 0 || ntt_context_create(12289, 256) == NULL);   /* fails loudly, no CPU fallback */
    } else {
        test_ntt();
        test_commitment();
        test_sampler();
        test_prover_path();
    }
    printf("%s: %d checks, %d failures (%s)\n", failures ? "FAILED" : "ok", checks, failures, lsr_version());
    return failures ? 1 : 0;
}
