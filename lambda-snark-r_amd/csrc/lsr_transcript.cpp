// Fiat–Shamir consumer of the commitment words: the transcript of
// rust-api/lambda-snark/src/challenge.rs:102-134 —
//   h = SHA3-256("LAMBDA-SNARK-R-FS-v1" || LE64(#inputs) || LE64(input)... || LE64(#words) || LE64(word)...)
//   alpha = LE64(h[0..8]) mod q
// Host code (FIPS 202 sponge); the commitment words already live in host memory at this boundary.
#include <cstdint>
#include <cstring>

#include "lambda_snark/batch.h"

namespace {

constexpr uint64_t kRoundConstants[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL, 0x000000000000808BULL, 0x0000000080000001ULL,
    0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
    0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
    0x000000000000800AULL, 0x800000008000000AULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
constexpr int kRotation[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};

inline uint64_t rotl(uint64_t v, int s) { return s ? (v << s) | (v >> (64 - s)) : v; }

void keccak_f1600(uint64_t (&a)[25]) {
    for (int round = 0; round < 24; ++round) {
        uint64_t c[5], d[5], b[25];
        for (int x = 0; x < 5; ++x) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ rotl(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; ++i) a[i] ^= d[i % 5];
        for (int x = 0; x < 5; ++x)
            for (int y = 0; y < 5; ++y) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl(a[x + 5 * y], kRotation[x + 5 * y]);
        for (int y = 0; y < 5; ++y)
            for (int x = 0; x < 5; ++x) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= kRoundConstants[round];
    }
}

class Sha3_256 {
public:
    void update(const void* data, size_t len) {
        const auto* p = static_cast<const uint8_t*>(data);
        while (len) {
            const size_t take = len < kRate - fill_ ? len : kRate - fill_;
            for (size_t i = 0; i < take; ++i) xor_byte(fill_ + i, p[i]);
            fill_ += take; p += take; len -= take;
            if (fill_ == kRate) { keccak_f1600(state_); fill_ = 0; }
        }
    }
    void update_le64(uint64_t v) {
        uint8_t b[8];
        for (int i = 0; i < 8; ++i) b[i] = static_cast<uint8_t>(v >> (8 * i));
        update(b, 8);
    }
    void finish(uint8_t out[32]) {
        xor_byte(fill_, 0x06);
        xor_byte(kRate - 1, 0x80);
        keccak_f1600(state_);
        for (int i = 0; i < 32; ++i) out[i] = static_cast<uint8_t>(state_[i / 8] >> (8 * (i % 8)));
    }

private:
    static constexpr size_t kRate = 136;
    void xor_byte(size_t pos, uint8_t v) { state_[pos / 8] ^= static_cast<uint64_t>(v) << (8 * (pos % 8)); }
    uint64_t state_[25] = {};
    size_t fill_ = 0;
};

}  // namespace

extern "C" int lsr_fs_challenge(const uint64_t* public_inputs, size_t n_inputs, const LweCommitment* commitment, uint64_t modulus,
                                uint64_t* alpha, uint8_t* hash32) noexcept {
    if ((!public_inputs && n_inputs) || !commitment || !commitment->data || modulus == 0 || !alpha) return -1;
    Sha3_256 h;
    h.update("LAMBDA-SNARK-R-FS-v1", 20);                       // challenge.rs:107
    h.update_le64(static_cast<uint64_t>(n_inputs));            // :110
    for (size_t i = 0; i < n_inputs; ++i) h.update_le64(public_inputs[i]);   // :113-115
    h.update_le64(static_cast<uint64_t>(commitment->len));     // :119
    for (size_t i = 0; i < commitment->len; ++i) h.update_le64(commitment->data[i]);   // :120-122
    uint8_t digest[32];
    h.finish(digest);
    uint64_t raw = 0;
    for (int i = 0; i < 8; ++i) raw |= static_cast<uint64_t>(digest[i]) << (8 * i);   // :129-131
    *alpha = raw % modulus;
    if (hash32) std::memcpy(hash32, digest, 32);
    return 0;
}
