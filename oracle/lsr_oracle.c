/*
 * lsr_oracle.c — CPU oracle (plain C) for the Lambda-SNARK-R hot path.  TEST INFRASTRUCTURE ONLY:
 * see lsr_oracle.h for the rules, the reference lines each function follows, and what is / is not
 * pinned by the reference's own tests.
 */
#include "lsr_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------------------------------ */
/* number theory                                                                                */
/* ------------------------------------------------------------------------------------------ */

/* SEAL multiply_uint_mod: 128-bit product, Barrett-reduced to [0,q).  The value is determined by
 * (a*b) mod q, so a plain 128-bit remainder restates it (reference call site: ntt.cpp:116-118). */
uint64_t oracle_mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a * b) % q); }

uint64_t oracle_powmod(uint64_t a, uint64_t e, uint64_t q) {
    uint64_t r = 1 % q;
    a %= q;
    while (e) {
        if (e & 1) r = oracle_mulmod(r, a, q);
        a = oracle_mulmod(a, a, q);
        e >>= 1;
    }
    return r;
}

/* deterministic Miller–Rabin for 64-bit integers */
int oracle_is_prime(uint64_t n) {
    static const uint64_t small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    if (n < 2) return 0;
    for (size_t i = 0; i < sizeof small / sizeof *small; ++i) {
        if (n == small[i]) return 1;
        if (n % small[i] == 0) return 0;
    }
    uint64_t d = n - 1;
    int s = 0;
    while (!(d & 1)) { d >>= 1; ++s; }
    for (size_t i = 0; i < sizeof small / sizeof *small; ++i) {
        uint64_t x = oracle_powmod(small[i], d, n);
        if (x == 1 || x == n - 1) continue;
        int comp = 1;
        for (int r = 1; r < s; ++r) {
            x = oracle_mulmod(x, x, n);
            if (x == n - 1) { comp = 0; break; }
        }
        if (comp) return 0;
    }
    return 1;
}

/* SEAL util::try_minimal_primitive_root(degree = 2n, modulus): any primitive degree-th root (SEAL
 * draws random candidates g^((q-1)/degree) and keeps one with root^(degree/2) == q-1), then the
 * numerically smallest over all its odd powers.  For prime q the group is cyclic, so the minimum
 * does not depend on which primitive root was found: a deterministic candidate scan is equivalent. */
uint64_t oracle_minimal_primitive_root(uint64_t degree, uint64_t q) {
    if (q < 2 || degree < 2 || (degree & (degree - 1))) return 0;
    uint64_t group = q - 1;
    if (group % degree) return 0;
    uint64_t quot = group / degree;
    uint64_t root = 0;
    for (uint64_t g = 2; g < 2 + 4096 && g < q; ++g) {
        uint64_t c = oracle_powmod(g, quot, q);
        if (c != 0 && oracle_powmod(c, degree >> 1, q) == q - 1) { root = c; break; }
    }
    if (!root) return 0;
    uint64_t gen_sq = oracle_mulmod(root, root, q);
    uint64_t cur = root, best = root;
    for (uint64_t i = 0; i < degree; i += 2) {
        if (cur < best) best = cur;
        cur = oracle_mulmod(cur, gen_sq, q);
    }
    return best;
}

/* SEAL util::get_primes(factor, bit_size, 1): value = ((2^bits - 1)/factor)*factor + 1, step -factor,
 * stop at 2^(bits-1). */
uint64_t oracle_largest_prime_1mod(uint64_t factor, int bits) {
    if (bits < 2 || bits > 61 || factor == 0) return 0;
    uint64_t value = ((((uint64_t)1 << bits) - 1) / factor) * factor + 1;
    uint64_t lower = (uint64_t)1 << (bits - 1);
    while (value > lower) {
        if (oracle_is_prime(value)) return value;
        if (value < factor) break;
        value -= factor;
    }
    return 0;
}

static uint32_t bitrev(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

/* ------------------------------------------------------------------------------------------ */
/* NTT tables + Harvey butterflies                                                              */
/* ------------------------------------------------------------------------------------------ */

typedef struct { uint64_t operand, quotient; } mm_operand;   /* SEAL MultiplyUIntModOperand */

struct oracle_ntt {
    uint64_t q;
    uint32_t n;
    int logn;
    uint64_t root, inv_root;
    mm_operand* root_powers;      /* [bitrev(i)] = psi^i                      */
    mm_operand* inv_root_powers;  /* [bitrev(i-1)+1] = psi^-i, [0] = 1        */
    mm_operand inv_degree;        /* n^-1                                     */
};

static mm_operand mm_set(uint64_t w, uint64_t q) {
    mm_operand o;
    o.operand = w;
    o.quotient = (uint64_t)(((u128)w << 64) / q);
    return o;
}

/* SEAL multiply_uint_mod_lazy: result in [0, 2q) */
static inline uint64_t mul_lazy(uint64_t x, mm_operand w, uint64_t q) {
    uint64_t hi = (uint64_t)(((u128)x * w.quotient) >> 64);
    return x * w.operand - hi * q;
}

/* follows cpp-core/src/ntt.cpp:30-70 (+ SEAL Modulus / NTTTables constructors) */
oracle_ntt* oracle_ntt_create(uint64_t q, uint32_t n) {
    if (n == 0) return NULL;                          /* ntt.cpp:31 */
    if (n & (n - 1)) return NULL;                     /* ntt.cpp:41-44 */
    if (q < 2 || (q >> 61)) return NULL;              /* seal::Modulus: 2 <= q < 2^61 (q==0 fails later) */
    int logn = 0;
    while (((uint32_t)1 << logn) < n) ++logn;
    if (logn < 1 || logn > 17) return NULL;           /* NTTTables: SEAL_POLY_MOD_DEGREE_MIN..MAX */
    /* SEAL does not test primality; for composite q its randomized root search is either always
     * failing or run-to-run random.  The oracle (and the product) reject composites. */
    if (!oracle_is_prime(q)) return NULL;
    uint64_t root = oracle_minimal_primitive_root((uint64_t)2 * n, q);
    if (!root) return NULL;                           /* NTTTables throws "invalid modulus" */
    oracle_ntt* t = (oracle_ntt*)calloc(1, sizeof *t);
    if (!t) return NULL;
    t->q = q; t->n = n; t->logn = logn; t->root = root;
    t->inv_root = oracle_powmod(root, q - 2, q);
    t->root_powers = (mm_operand*)malloc(sizeof(mm_operand) * n);
    t->inv_root_powers = (mm_operand*)malloc(sizeof(mm_operand) * n);
    if (!t->root_powers || !t->inv_root_powers) { oracle_ntt_free(t); return NULL; }
    t->root_powers[0] = mm_set(1, q);
    t->inv_root_powers[0] = mm_set(1, q);
    uint64_t p = root;
    for (uint32_t i = 1; i < n; ++i) {
        t->root_powers[bitrev(i, logn)] = mm_set(p, q);
        p = oracle_mulmod(p, root, q);
    }
    p = t->inv_root;
    for (uint32_t i = 1; i < n; ++i) {
        t->inv_root_powers[bitrev(i - 1, logn) + 1] = mm_set(p, q);
        p = oracle_mulmod(p, t->inv_root, q);
    }
    t->inv_degree = mm_set(oracle_powmod(n, q - 2, q), q);
    return t;
}

void oracle_ntt_free(oracle_ntt* t) {
    if (!t) return;
    free(t->root_powers);
    free(t->inv_root_powers);
    free(t);
}

uint64_t oracle_ntt_root(const oracle_ntt* t) { return t ? t->root : 0; }

void oracle_ntt_tables(const oracle_ntt* t, uint64_t* rp, uint64_t* irp) {
    for (uint32_t i = 0; i < t->n; ++i) {
        if (rp) rp[i] = t->root_powers[i].operand;
        if (irp) irp[i] = t->inv_root_powers[i].operand;
    }
}

/* SEAL DWTHandler::transform_to_rev + final correction of ntt_negacyclic_harvey */
static void fwd_core(const oracle_ntt* t, uint64_t* v) {
    const uint64_t q = t->q, two_q = 2 * q;
    const uint32_t n = t->n;
    const mm_operand* roots = t->root_powers;
    uint32_t gap = n >> 1;
    for (uint32_t m = 1; m < n; m <<= 1, gap >>= 1) {
        uint32_t offset = 0;
        for (uint32_t i = 0; i < m; ++i) {
            const mm_operand r = *++roots;
            uint64_t* x = v + offset;
            uint64_t* y = x + gap;
            for (uint32_t j = 0; j < gap; ++j) {
                uint64_t u = *x - (two_q & (uint64_t)(-(int64_t)(*x >= two_q)));   /* guard */
                uint64_t w = mul_lazy(*y, r, q);
                *x++ = u + w;
                *y++ = u + two_q - w;
            }
            offset += gap << 1;
        }
    }
    for (uint32_t i = 0; i < n; ++i) {
        if (v[i] >= two_q) v[i] -= two_q;
        if (v[i] >= q) v[i] -= q;
    }
}

/* SEAL DWTHandler::transform_from_rev with scalar = n^-1 + final correction */
static void inv_core(const oracle_ntt* t, uint64_t* v) {
    const uint64_t q = t->q, two_q = 2 * q;
    const uint32_t n = t->n;
    const mm_operand* roots = t->inv_root_powers;
    uint32_t gap = 1;
    uint32_t m = n >> 1;
    for (; m > 1; m >>= 1, gap <<= 1) {
        uint32_t offset = 0;
        for (uint32_t i = 0; i < m; ++i) {
            const mm_operand r = *++roots;
            uint64_t* x = v + offset;
            uint64_t* y = x + gap;
            for (uint32_t j = 0; j < gap; ++j) {
                uint64_t u = *x, w = *y;
                uint64_t s = u + w;
                *x++ = s - (two_q & (uint64_t)(-(int64_t)(s >= two_q)));
                *y++ = mul_lazy(u + two_q - w, r, q);
            }
            offset += gap << 1;
        }
    }
    {   /* last stage, n^-1 folded in */
        const mm_operand r = *++roots;
        const mm_operand scaled = mm_set(oracle_mulmod(r.operand, t->inv_degree.operand, q), q);
        uint64_t* x = v;
        uint64_t* y = v + gap;
        for (uint32_t j = 0; j < gap; ++j) {
            uint64_t u = *x - (two_q & (uint64_t)(-(int64_t)(*x >= two_q)));
            uint64_t w = *y;
            uint64_t s = u + w;
            s -= two_q & (uint64_t)(-(int64_t)(s >= two_q));
            *x++ = mul_lazy(s, t->inv_degree, q);
            *y++ = mul_lazy(u + two_q - w, scaled, q);
        }
    }
    for (uint32_t i = 0; i < n; ++i)
        if (v[i] >= q) v[i] -= q;
}

/* ntt.cpp:76-89 */
int oracle_ntt_forward(const oracle_ntt* t, uint64_t* a, uint32_t n) {
    if (!t || !a || n != t->n) return -1;
    fwd_core(t, a);
    return 0;
}
/* ntt.cpp:91-104 */
int oracle_ntt_inverse(const oracle_ntt* t, uint64_t* a, uint32_t n) {
    if (!t || !a || n != t->n) return -1;
    inv_core(t, a);
    return 0;
}
/* ntt.cpp:106-119: silent no-op on NULL, n not validated */
void oracle_ntt_mul_pointwise(const oracle_ntt* t, uint64_t* r, const uint64_t* a, const uint64_t* b, uint32_t n) {
    if (!t || !r || !a || !b) return;
    for (uint32_t i = 0; i < n; ++i) r[i] = oracle_mulmod(a[i], b[i], t->q);
}

int oracle_ntt_forward_batch(const oracle_ntt* t, uint64_t* a, size_t batch) {
    if (!t || !a) return -1;
    for (size_t b = 0; b < batch; ++b) fwd_core(t, a + b * (size_t)t->n);
    return 0;
}
int oracle_ntt_inverse_batch(const oracle_ntt* t, uint64_t* a, size_t batch) {
    if (!t || !a) return -1;
    for (size_t b = 0; b < batch; ++b) inv_core(t, a + b * (size_t)t->n);
    return 0;
}

void oracle_ntt_forward_naive(const oracle_ntt* t, const uint64_t* a, uint64_t* out) {
    const uint64_t q = t->q;
    for (uint32_t i = 0; i < t->n; ++i) {
        uint64_t x = oracle_powmod(t->root, 2 * (uint64_t)bitrev(i, t->logn) + 1, q);
        uint64_t acc = 0;   /* Horner from the top */
        for (uint32_t j = t->n; j-- > 0;) acc = (oracle_mulmod(acc, x, q) + a[j] % q) % q;
        out[i] = acc;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* input generator                                                                              */
/* ------------------------------------------------------------------------------------------ */
void oracle_splitmix_fill(uint64_t seed, uint64_t q, uint64_t* out, size_t len) {
    uint64_t x = seed;
    for (size_t i = 0; i < len; ++i) {
        x += 0x9E3779B97F4A7C15ull;
        uint64_t z = x;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        out[i] = q ? z % q : z;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* ChaCha20 (RFC 8439 §2.3) and the seeded word stream                                          */
/* ------------------------------------------------------------------------------------------ */
#define ROTL32(v, c) (((v) << (c)) | ((v) >> (32 - (c))))
#define QR(a, b, c, d)                                                                             \
    a += b; d ^= a; d = ROTL32(d, 16); c += d; b ^= c; b = ROTL32(b, 12);                          \
    a += b; d ^= a; d = ROTL32(d, 8);  c += d; b ^= c; b = ROTL32(b, 7)

void oracle_chacha20_block(const uint32_t key[8], uint32_t counter, const uint32_t nonce[3], uint32_t out[16]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                      key[4], key[5], key[6], key[7], counter, nonce[0], nonce[1], nonce[2]};
    uint32_t x[16];
    memcpy(x, s, sizeof x);
    for (int i = 0; i < 10; ++i) {
        QR(x[0], x[4], x[8], x[12]); QR(x[1], x[5], x[9], x[13]); QR(x[2], x[6], x[10], x[14]); QR(x[3], x[7], x[11], x[15]);
        QR(x[0], x[5], x[10], x[15]); QR(x[1], x[6], x[11], x[12]); QR(x[2], x[7], x[8], x[13]); QR(x[3], x[4], x[9], x[14]);
    }
    for (int i = 0; i < 16; ++i) out[i] = x[i] + s[i];
}

/* a stream is (256-bit key, domain, index): nonce = {domain, index_lo, index_hi}; counter = block */
static void key_block(const uint32_t key[8], uint32_t domain, uint64_t index, uint32_t block, uint64_t w[8]) {
    const uint32_t nonce[3] = {domain, (uint32_t)index, (uint32_t)(index >> 32)};
    uint32_t o[16];
    oracle_chacha20_block(key, block, nonce, o);
    for (int j = 0; j < 8; ++j) w[j] = (uint64_t)o[2 * j] | ((uint64_t)o[2 * j + 1] << 32);
}
/* raw 64-bit test seed: key = {seed_lo, seed_hi, 'LSR1', 'STRM', 0,0,0,0} */
static void expand_seed(uint64_t seed, uint32_t key[8]) {
    const uint32_t k[8] = {(uint32_t)seed, (uint32_t)(seed >> 32), 0x3152534Cu, 0x4D525453u, 0, 0, 0, 0};
    memcpy(key, k, sizeof k);
}
static void stream_block(uint64_t seed, uint32_t domain, uint64_t index, uint32_t block, uint64_t w[8]) {
    uint32_t key[8];
    expand_seed(seed, key);
    key_block(key, domain, index, block, w);
}

/* ---- key schedule (restates the library's definition, DESIGN.md "Commitment definition"; there is no reference
 * counterpart: cpp-core/src/commitment.cpp:141-157 ignores `seed` and draws fresh SEAL randomness) ---- */
#define TAG(a, b, c, d) ((uint32_t)(a) | ((uint32_t)(b) << 8) | ((uint32_t)(c) << 16) | ((uint32_t)(d) << 24))
static void kdf(const uint32_t key[8], uint32_t label, uint32_t a, uint32_t b, uint32_t c, uint32_t out[8]) {
    const uint32_t nonce[3] = {a, b, c};
    uint32_t o[16];
    oracle_chacha20_block(key, label, nonce, o);
    memcpy(out, o, 8 * sizeof(uint32_t));
}
void oracle_context_keys(uint64_t key_seed, uint32_t pub[8], uint32_t sec[8], uint32_t id[4]) {
    const uint32_t master[8] = {(uint32_t)key_seed, (uint32_t)(key_seed >> 32), TAG('L', 'S', 'R', '2'), TAG('M', 'S', 'T', 'R'), 0, 0, 0, 0};
    uint32_t idk[8];
    kdf(master, TAG('P', 'U', 'B', 'K'), 0, 0, 0, pub);
    kdf(master, TAG('S', 'E', 'C', 'K'), 0, 0, 0, sec);
    kdf(sec, TAG('C', 'T', 'I', 'D'), pub[0], pub[1], pub[2], idk);
    memcpy(id, idk, 4 * sizeof(uint32_t));
}
static const uint64_t P61 = ((uint64_t)1 << 61) - 1;
static uint64_t mul61(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a * b) % P61); }
void oracle_commit_key(uint64_t seed, const uint32_t id[4], const uint64_t* msg, size_t copy, uint64_t t, uint32_t out[8]) {
    const uint32_t base[8] = {(uint32_t)seed, (uint32_t)(seed >> 32), TAG('L', 'S', 'R', '2'), TAG('C', 'M', 'I', 'T'), id[0], id[1], id[2], id[3]};
    uint32_t pts[8], step[8];
    kdf(base, TAG('H', 'P', 'N', 'T'), 0, 0, 0, pts);
    const uint64_t x1 = ((((uint64_t)pts[1] << 32) | pts[0]) & P61) % P61, x2 = ((((uint64_t)pts[3] << 32) | pts[2]) & P61) % P61;
    uint64_t h1 = 0, h2 = 0, p1 = x1, p2 = x2;
    for (size_t i = 0; i < copy; ++i) {
        const uint64_t m = msg[i] % t;
        h1 = (h1 + mul61(m, p1)) % P61;
        h2 = (h2 + mul61(m, p2)) % P61;
        p1 = mul61(p1, x1);
        p2 = mul61(p2, x2);
    }
    kdf(base, TAG('C', 'K', 'Y', '1'), (uint32_t)h1, (uint32_t)(h1 >> 32), 0, step);
    kdf(step, TAG('C', 'K', 'Y', '2'), (uint32_t)h2, (uint32_t)(h2 >> 32), 0, out);
}

void oracle_stream_words(uint64_t seed, uint32_t domain, uint64_t index, uint64_t first, uint64_t* out, size_t count) {
    uint64_t w[8];
    uint64_t have = (uint64_t)-1;
    for (size_t i = 0; i < count; ++i) {
        uint64_t idx = first + i;
        if (idx / 8 != have) { have = idx / 8; stream_block(seed, domain, index, (uint32_t)have, w); }
        out[i] = w[idx % 8];
    }
}

/* ------------------------------------------------------------------------------------------ */
/* CDT Gaussian sampler — utils.cpp:26-75 (table), :95-121 (scan), :132-146 (entry point)       */
/* ------------------------------------------------------------------------------------------ */
size_t oracle_gaussian_cdf(double sigma, uint64_t* cdf, size_t cap) {
    if (!(sigma > 0.0) || !isfinite(sigma)) return 0;
    const long double s = (long double)sigma, s2 = s * s;
    long double bound = ceill(12.0L * s);
    if (bound < 8.0L) bound = 8.0L;
    const size_t max_index = (size_t)bound;
    if (max_index + 1 > cap) return 0;
    long double* w = (long double*)malloc(sizeof(long double) * (max_index + 1));
    if (!w) return 0;
    long double sum = 0.0L;
    for (size_t k = 0; k <= max_index; ++k) {
        long double kk = (long double)k * (long double)k;
        long double wt = expl(-kk / (2.0L * s2));
        if (k > 0) wt *= 2.0L;
        w[k] = wt;
        sum += wt;
    }
    const long double maxu = (long double)UINT64_MAX;
    const long double scale = maxu / sum;
    long double cum = 0.0L;
    for (size_t k = 0; k <= max_index; ++k) {
        cum += w[k];
        long double v = cum * scale;
        if (v >= maxu) cdf[k] = UINT64_MAX;
        else if (v <= 0.0L) cdf[k] = 0;
        else cdf[k] = (uint64_t)v;
    }
    cdf[max_index] = UINT64_MAX;
    free(w);
    return max_index + 1;
}

static int64_t cdt_pick(const uint64_t* cdf, size_t entries, uint64_t u, uint64_t sign_word) {
    uint32_t chosen = (uint32_t)(entries - 1);
    uint64_t found = 0;
    for (size_t k = 0; k < entries; ++k) {   /* branch-free first-k-with-cdf>=u, utils.cpp:101-108 */
        uint64_t ge = (uint64_t)(cdf[k] >= u);
        uint64_t sel = ge & (1ull ^ found);
        uint32_t m32 = (uint32_t)(-(int32_t)sel);
        chosen = (chosen & ~m32) | ((uint32_t)k & m32);
        found |= sel;
    }
    uint64_t sign = (sign_word & 1ull) & (uint64_t)(chosen != 0);
    int64_t mag = (int64_t)chosen;
    return sign ? -mag : mag;
}

/* the library's seeded form: ONE stream word per sample — low bit = sign, upper 63 bits against the table at 63-bit
 * precision (first k with cdf[k] >> 1 >= word >> 1); same scan, same sign rule */
static int64_t cdt_pick_word(const uint64_t* cdf, size_t entries, uint64_t word) {
    uint32_t chosen = (uint32_t)(entries - 1);
    uint64_t found = 0;
    const uint64_t u = word >> 1;
    for (size_t k = 0; k < entries; ++k) {
        uint64_t ge = (uint64_t)((cdf[k] >> 1) >= u);
        uint64_t sel = ge & (1ull ^ found);
        uint32_t m32 = (uint32_t)(-(int32_t)sel);
        chosen = (chosen & ~m32) | ((uint32_t)k & m32);
        found |= sel;
    }
    uint64_t sign = (word & 1ull) & (uint64_t)(chosen != 0);
    int64_t mag = (int64_t)chosen;
    return sign ? -mag : mag;
}

int oracle_sample_gaussian(uint64_t* out, size_t len, double sigma) {
    if (!out || len == 0 || !(sigma > 0.0) || !isfinite(sigma)) return -1;   /* utils.cpp:133 */
    uint64_t cdf[4096];
    size_t entries = oracle_gaussian_cdf(sigma, cdf, 4096);
    if (!entries) return -1;
    FILE* f = fopen("/dev/urandom", "rb");
    if (!f) return -1;
    for (size_t i = 0; i < len; ++i) {
        uint64_t r[2];
        if (fread(r, sizeof r, 1, f) != 1) { fclose(f); return -1; }
        out[i] = (uint64_t)cdt_pick(cdf, entries, r[0], r[1]);
    }
    fclose(f);
    return 0;
}

static int sample_gaussian_keyed(uint64_t* out, size_t len, double sigma, const uint32_t key[8], uint32_t domain, uint64_t index) {
    if (!out || len == 0 || !(sigma > 0.0) || !isfinite(sigma)) return -1;
    uint64_t cdf[4096];
    size_t entries = oracle_gaussian_cdf(sigma, cdf, 4096);
    if (!entries) return -1;
    uint64_t w[8];
    for (size_t i = 0; i < len; ++i) {
        if ((i & 7) == 0) key_block(key, domain, index, (uint32_t)(i >> 3), w);
        out[i] = (uint64_t)cdt_pick_word(cdf, entries, w[i & 7]);
    }
    return 0;
}
int oracle_sample_gaussian_seeded(uint64_t* out, size_t len, double sigma, uint64_t seed, uint32_t domain, uint64_t index) {
    uint32_t key[8];
    expand_seed(seed, key);
    return sample_gaussian_keyed(out, len, sigma, key, domain, index);
}

/* ------------------------------------------------------------------------------------------ */
/* Module-LWE commitment ("Commitment definition")                                   */
/* ------------------------------------------------------------------------------------------ */
enum { DOM_A = 1, DOM_S = 2, DOM_E = 3, DOM_R = 4, DOM_E1 = 5, DOM_E2 = 6 };
/* "LSRC0001" little-endian */
static const uint64_t kMagic = 0x313030304352534CULL;

struct oracle_lwe {
    uint64_t q, t, delta;
    uint32_t n, k;
    double sigma;
    double noise_unit;           /* 8 sqrt(2 k n) sigma^2 */
    uint32_t pub[8], sec[8], id[4];
    oracle_ntt* ntt;
    uint64_t* a_hat;   /* [k][k][n], NTT domain */
    uint64_t* s_hat;   /* [k][n], NTT domain   */
    uint64_t* b_hat;   /* [k][n], NTT domain   */
};

uint64_t oracle_lwe_select_modulus(uint64_t req, uint32_t n) {
    if (n < 2 || (n & (n - 1)) || n > 131072) return 0;
    if (req >= ((uint64_t)1 << 40) && req < ((uint64_t)1 << 61) && (req - 1) % (2ull * n) == 0 && oracle_is_prime(req)) return req;
    if (n <= 4096) return 17592169062401ull;              /* the north-star 44-bit prime (r1cs.rs:527) */
    return oracle_largest_prime_1mod(2ull * n, 44);       /* n=2^16 -> 17592182243329 */
}

static void uniform_poly(const uint32_t key[8], uint32_t domain, uint64_t index, uint64_t q, uint64_t* out, uint32_t n) {
    uint64_t w[8];
    for (uint32_t i = 0; i < n; ++i) {
        if ((i & 7) == 0) key_block(key, domain, index, i >> 3, w);
        out[i] = (uint64_t)(((u128)w[i & 7] * q) >> 64);
    }
}

static void gaussian_poly(const uint32_t key[8], uint32_t domain, uint64_t index, double sigma, uint64_t q, uint64_t* out, uint32_t n) {
    sample_gaussian_keyed(out, n, sigma, key, domain, index);
    for (uint32_t i = 0; i < n; ++i) {
        int64_t v = (int64_t)out[i];
        out[i] = v < 0 ? q - (uint64_t)(-v) : (uint64_t)v;
    }
}

oracle_lwe* oracle_lwe_create(uint64_t req_q, uint32_t n, uint32_t k, double sigma, uint64_t key_seed) {
    if (k == 0) k = 1;
    if (k > 16 || !(sigma > 0.0) || !isfinite(sigma) || sigma > 1024.0) return NULL;
    uint64_t q = oracle_lwe_select_modulus(req_q, n);
    if (!q) return NULL;
    uint64_t t = oracle_largest_prime_1mod(2ull * n, 20);   /* SEAL PlainModulus::Batching(n, 20), commitment.cpp:111 */
    if (!t) return NULL;
    oracle_lwe* c = (oracle_lwe*)calloc(1, sizeof *c);
    c->q = q; c->t = t; c->delta = q / t; c->n = n; c->k = k; c->sigma = sigma;
    c->noise_unit = 8.0 * sqrt(2.0 * k * n) * sigma * sigma;
    if (c->noise_unit >= 0.5 * (double)c->delta) { free(c); return NULL; }   /* fresh commitments could fail to open */
    oracle_context_keys(key_seed, c->pub, c->sec, c->id);
    c->ntt = oracle_ntt_create(q, n);
    if (!c->ntt) { free(c); return NULL; }
    c->a_hat = (uint64_t*)malloc(sizeof(uint64_t) * k * k * n);
    c->s_hat = (uint64_t*)malloc(sizeof(uint64_t) * k * n);
    c->b_hat = (uint64_t*)malloc(sizeof(uint64_t) * k * n);
    uint64_t* tmp = (uint64_t*)malloc(sizeof(uint64_t) * n);
    for (uint32_t i = 0; i < k; ++i)
        for (uint32_t j = 0; j < k; ++j) uniform_poly(c->pub, DOM_A, (uint64_t)i * k + j, q, c->a_hat + ((size_t)i * k + j) * n, n);
    for (uint32_t j = 0; j < k; ++j) {
        gaussian_poly(c->sec, DOM_S, j, sigma, q, c->s_hat + (size_t)j * n, n);
        fwd_core(c->ntt, c->s_hat + (size_t)j * n);
    }
    for (uint32_t i = 0; i < k; ++i) {
        uint64_t* b = c->b_hat + (size_t)i * n;
        gaussian_poly(c->sec, DOM_E, i, sigma, q, b, n);
        fwd_core(c->ntt, b);
        for (uint32_t j = 0; j < k; ++j) {
            const uint64_t* a = c->a_hat + ((size_t)i * k + j) * n;
            const uint64_t* s = c->s_hat + (size_t)j * n;
            for (uint32_t x = 0; x < n; ++x) b[x] = (b[x] + oracle_mulmod(a[x], s[x], q)) % q;
        }
    }
    free(tmp);
    return c;
}

void oracle_lwe_free(oracle_lwe* c) {
    if (!c) return;
    oracle_ntt_free(c->ntt);
    free(c->a_hat); free(c->s_hat); free(c->b_hat);
    free(c);
}
uint64_t oracle_lwe_q(const oracle_lwe* c) { return c->q; }
uint64_t oracle_lwe_t(const oracle_lwe* c) { return c->t; }
size_t oracle_lwe_commit_words(const oracle_lwe* c) { return 1 + 4 + (size_t)(c->k + 1) * c->n; }
void oracle_lwe_public_matrix(const oracle_lwe* c, uint64_t* a_hat) { memcpy(a_hat, c->a_hat, sizeof(uint64_t) * c->k * c->k * c->n); }

/* u[j] = INTT(sum_i A[i][j] o NTT(r_i)) + e1[j] */
void oracle_mlwe_matvec(const oracle_ntt* t, uint32_t k, const uint64_t* a_hat, const uint64_t* r, const uint64_t* e1, uint64_t* u) {
    const uint32_t n = t->n;
    const uint64_t q = t->q;
    uint64_t* rh = (uint64_t*)malloc(sizeof(uint64_t) * k * n);
    memcpy(rh, r, sizeof(uint64_t) * k * n);
    for (uint32_t i = 0; i < k; ++i) fwd_core(t, rh + (size_t)i * n);
    for (uint32_t j = 0; j < k; ++j) {
        uint64_t* o = u + (size_t)j * n;
        for (uint32_t x = 0; x < n; ++x) {
            uint64_t acc = 0;
            for (uint32_t i = 0; i < k; ++i) acc = (acc + oracle_mulmod(a_hat[((size_t)i * k + j) * n + x], rh[(size_t)i * n + x], q)) % q;
            o[x] = acc;
        }
        inv_core(t, o);
        if (e1)
            for (uint32_t x = 0; x < n; ++x) o[x] = (o[x] + e1[(size_t)j * n + x]) % q;
    }
    free(rh);
}

int oracle_lwe_commit(const oracle_lwe* c, const uint64_t* msg, size_t msg_len, uint64_t seed, uint64_t* out) {
    if (!c || !msg || !out) return -1;
    const uint32_t n = c->n, k = c->k;
    const uint64_t q = c->q;
    uint64_t* r = (uint64_t*)malloc(sizeof(uint64_t) * k * n);
    uint64_t* e1 = (uint64_t*)malloc(sizeof(uint64_t) * k * n);
    uint64_t* e2 = (uint64_t*)malloc(sizeof(uint64_t) * n);
    /* per-commitment key = PRF(seed, context id, embedded message); seed == 0 means fresh entropy in the library and has no
       deterministic counterpart — the oracle then uses the all-zero seed's key like any other */
    uint32_t ck[8];
    oracle_commit_key(seed, c->id, msg, msg_len < n ? msg_len : n, c->t, ck);
    for (uint32_t i = 0; i < k; ++i) {
        gaussian_poly(ck, DOM_R, i, c->sigma, q, r + (size_t)i * n, n);
        gaussian_poly(ck, DOM_E1, i, c->sigma, q, e1 + (size_t)i * n, n);
    }
    gaussian_poly(ck, DOM_E2, 0, c->sigma, q, e2, n);
    out[0] = 8ull * (4 + (size_t)(k + 1) * n);
    out[1] = kMagic;
    out[2] = (uint64_t)n | ((uint64_t)k << 32);
    out[3] = q;
    out[4] = c->t;
    uint64_t* u = out + 5;
    uint64_t* v = u + (size_t)k * n;
    oracle_mlwe_matvec(c->ntt, k, c->a_hat, r, e1, u);
    /* v = INTT(<b_hat, r_hat>) + e2 + delta*m */
    for (uint32_t i = 0; i < k; ++i) fwd_core(c->ntt, r + (size_t)i * n);
    for (uint32_t x = 0; x < n; ++x) {
        uint64_t acc = 0;
        for (uint32_t i = 0; i < k; ++i) acc = (acc + oracle_mulmod(c->b_hat[(size_t)i * n + x], r[(size_t)i * n + x], q)) % q;
        v[x] = acc;
    }
    inv_core(c->ntt, v);
    const size_t copy = msg_len < n ? msg_len : n;          /* commitment.cpp:146-149 truncation */
    /* The message rides as round(q m / t) = floor((q m + t/2) / t), not as floor(q/t) m: with q = Delta t + rho a sum of commitments
     * carries Delta M for the integer M = sum c_i m_i, and Delta M = Delta (M mod t) - rho floor(M / t) (mod q) — the second term is noise
     * of size rho sum c_i (rho = 53217 for the default modulus: five times the sampled noise), which made combinations with
     * sum c_i > ~130 of large messages undecodable although they passed the noise budget (round 3).  With the rounded scaling
     * sum c_i round(q m_i / t) = (q/t) M + (rounding <= sum c_i / 2) = (q/t)(M mod t) (mod q): no rho term (what SEAL's BFV
     * encryptor does for the same reason, multiply_add_plain_with_scaling_variant). */
    for (uint32_t x = 0; x < n; ++x) {
        uint64_t m = x < copy ? msg[x] % c->t : 0;
        const uint64_t scaled = (uint64_t)((((u128)q * m) + (c->t >> 1)) / c->t);
        v[x] = (v[x] + e2[x] + scaled) % q;
    }
    free(r); free(e1); free(e2);
    return 0;
}

static int parse(const oracle_lwe* c, const uint64_t* comm, size_t len, const uint64_t** u, const uint64_t** v) {
    if (!c || !comm || len < 1) return 0;
    const uint64_t byte_len = comm[0];
    if (byte_len == 0 || byte_len > (len - 1) * 8) return 0;        /* commitment.cpp:71-75 */
    if (byte_len != 8ull * (4 + (size_t)(c->k + 1) * c->n)) return 0;
    if (comm[1] != kMagic || comm[2] != ((uint64_t)c->n | ((uint64_t)c->k << 32)) || comm[3] != c->q || comm[4] != c->t) return 0;
    *u = comm + 5;
    *v = *u + (size_t)c->k * c->n;
    for (size_t i = 0; i < (size_t)(c->k + 1) * c->n; ++i)
        if ((*u)[i] >= c->q) return 0;
    return 1;
}

int oracle_lwe_verify(const oracle_lwe* c, const uint64_t* comm, size_t comm_len, const uint64_t* msg, size_t msg_len) {
    if (!c || !comm || !msg) return -1;
    const uint64_t *u, *v;
    if (!parse(c, comm, comm_len, &u, &v)) return -1;
    const uint32_t n = c->n, k = c->k;
    const uint64_t q = c->q;
    if (msg_len > n) return 0;                                      /* commitment.cpp:219-221 */
    uint64_t* w = (uint64_t*)malloc(sizeof(uint64_t) * n);
    uint64_t* tmp = (uint64_t*)malloc(sizeof(uint64_t) * n);
    memcpy(w, v, sizeof(uint64_t) * n);
    fwd_core(c->ntt, w);
    for (uint32_t j = 0; j < k; ++j) {
        memcpy(tmp, u + (size_t)j * n, sizeof(uint64_t) * n);
        fwd_core(c->ntt, tmp);
        for (uint32_t x = 0; x < n; ++x) w[x] = (w[x] + q - oracle_mulmod(c->s_hat[(size_t)j * n + x], tmp[x], q)) % q;
    }
    inv_core(c->ntt, w);
    uint64_t diff = 0;
    for (size_t i = 0; i < msg_len; ++i) {
        uint64_t dec = (uint64_t)((((u128)w[i] * c->t) + (q >> 1)) / q) % c->t;
        diff |= dec ^ msg[i];                                       /* raw message word, commitment.cpp:224 */
    }
    free(w); free(tmp);
    return diff == 0 ? 1 : 0;
}

int oracle_lwe_linear_combine(const oracle_lwe* c, const uint64_t* const* comms, const size_t* lens, const uint64_t* coeffs, size_t count, uint64_t* out) {
    if (!c || !comms || !coeffs || count == 0 || !out) return -1;   /* commitment.cpp:240-242 */
    const size_t body = (size_t)(c->k + 1) * c->n;
    int has = 0;
    double weight = 0;
    /* coefficients act through their CENTRED representative mod t: c in (t/2, t) is the small negative number c - t, so a
     * subtraction of commitments (coefficient t - 1) costs one unit of noise, not t - 1 */
    for (size_t i = 0; i < count; ++i)
        if (comms[i]) {
            const uint64_t cf = coeffs[i] % c->t;
            weight += (double)(cf > c->t / 2 ? c->t - cf : cf);
        }
    if (weight * (c->noise_unit + 1.0) >= 0.5 * (double)c->delta) return -1;  /* the result could not be opened (noise budget) */
    memset(out, 0, sizeof(uint64_t) * (5 + body));
    for (size_t i = 0; i < count; ++i) {
        if (!comms[i]) continue;                                    /* commitment.cpp:248-250 */
        const uint64_t *u, *v;
        if (!parse(c, comms[i], lens[i], &u, &v)) return -1;
        uint64_t cf = coeffs[i] % c->t;                             /* commitment.cpp:90 */
        if (cf > c->t / 2) cf = c->q - (c->t - cf);                 /* the centred representative, as a residue mod q */
        for (size_t x = 0; x < body; ++x) out[5 + x] = (out[5 + x] + oracle_mulmod(cf, u[x], c->q)) % c->q;
        has = 1;
    }
    if (!has) return -1;                                            /* commitment.cpp:268-270 */
    out[0] = 8ull * (4 + body);
    out[1] = kMagic;
    out[2] = (uint64_t)c->n | ((uint64_t)c->k << 32);
    out[3] = c->q;
    out[4] = c->t;
    return 0;
}
