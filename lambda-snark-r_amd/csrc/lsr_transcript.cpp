// Fiat–Shamir consumer of the commitment words: the transcript of
// rust-api/lambda-snark/src/challenge.rs:102-134 —
//   h = SHA3-256("LAMBDA-SNARK-R-FS-v1" || LE64(#inputs) || LE64(input)... || LE64(#words) || LE64(word)...)
//   alpha = LE64(h[0..8]) mod q
// Host code (FIPS 202 sponge); the commitment words already live in host memory at this boundary.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "lambda_snark/batch.h"

namespace {

constexpr uint64_t kRoundConstants[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL, 0x000000000000808BULL, 0x0000000080000001ULL,
    0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
    0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
    0x000000000000800AULL, 0x800000008000000AULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

static_assert(__BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__, "the transcript absorbs host words as their little-endian bytes");

template <int S>
inline uint64_t rotl(uint64_t v) {
    if constexpr (S == 0) return v;
    else return (v << S) | (v >> (64 - S));
}

// one round on 25 named lanes: theta, rho + pi, chi, iota (FIPS 202 §3.2); A -> E
#define LSR_KECCAK_ROUND(A, E, RC)                                                                                    \
    do {                                                                                                              \
        const uint64_t c0 = A[0] ^ A[5] ^ A[10] ^ A[15] ^ A[20], c1 = A[1] ^ A[6] ^ A[11] ^ A[16] ^ A[21],               \
                       c2 = A[2] ^ A[7] ^ A[12] ^ A[17] ^ A[22], c3 = A[3] ^ A[8] ^ A[13] ^ A[18] ^ A[23],               \
                       c4 = A[4] ^ A[9] ^ A[14] ^ A[19] ^ A[24];                                                         \
        const uint64_t d0 = c4 ^ rotl<1>(c1), d1 = c0 ^ rotl<1>(c2), d2 = c1 ^ rotl<1>(c3), d3 = c2 ^ rotl<1>(c4),       \
                       d4 = c3 ^ rotl<1>(c0);                                                                            \
        uint64_t b0, b1, b2, b3, b4;                                                                                     \
        b0 = A[0] ^ d0; b1 = rotl<44>(A[6] ^ d1); b2 = rotl<43>(A[12] ^ d2); b3 = rotl<21>(A[18] ^ d3); b4 = rotl<14>(A[24] ^ d4); \
        E[0] = b0 ^ (~b1 & b2) ^ (RC); E[1] = b1 ^ (~b2 & b3); E[2] = b2 ^ (~b3 & b4); E[3] = b3 ^ (~b4 & b0); E[4] = b4 ^ (~b0 & b1); \
        b0 = rotl<28>(A[3] ^ d3); b1 = rotl<20>(A[9] ^ d4); b2 = rotl<3>(A[10] ^ d0); b3 = rotl<45>(A[16] ^ d1); b4 = rotl<61>(A[22] ^ d2); \
        E[5] = b0 ^ (~b1 & b2); E[6] = b1 ^ (~b2 & b3); E[7] = b2 ^ (~b3 & b4); E[8] = b3 ^ (~b4 & b0); E[9] = b4 ^ (~b0 & b1); \
        b0 = rotl<1>(A[1] ^ d1); b1 = rotl<6>(A[7] ^ d2); b2 = rotl<25>(A[13] ^ d3); b3 = rotl<8>(A[19] ^ d4); b4 = rotl<18>(A[20] ^ d0); \
        E[10] = b0 ^ (~b1 & b2); E[11] = b1 ^ (~b2 & b3); E[12] = b2 ^ (~b3 & b4); E[13] = b3 ^ (~b4 & b0); E[14] = b4 ^ (~b0 & b1); \
        b0 = rotl<27>(A[4] ^ d4); b1 = rotl<36>(A[5] ^ d0); b2 = rotl<10>(A[11] ^ d1); b3 = rotl<15>(A[17] ^ d2); b4 = rotl<56>(A[23] ^ d3); \
        E[15] = b0 ^ (~b1 & b2); E[16] = b1 ^ (~b2 & b3); E[17] = b2 ^ (~b3 & b4); E[18] = b3 ^ (~b4 & b0); E[19] = b4 ^ (~b0 & b1); \
        b0 = rotl<62>(A[2] ^ d2); b1 = rotl<55>(A[8] ^ d3); b2 = rotl<39>(A[14] ^ d4); b3 = rotl<41>(A[15] ^ d0); b4 = rotl<2>(A[21] ^ d1); \
        E[20] = b0 ^ (~b1 & b2); E[21] = b1 ^ (~b2 & b3); E[22] = b2 ^ (~b3 & b4); E[23] = b3 ^ (~b4 & b0); E[24] = b4 ^ (~b0 & b1); \
    } while (0)

void keccak_f1600(uint64_t (&a)[25]) {
    uint64_t e[25];
    for (int round = 0; round < 24; round += 2) {
        LSR_KECCAK_ROUND(a, e, kRoundConstants[round]);
        LSR_KECCAK_ROUND(e, a, kRoundConstants[round + 1]);
    }
}
#undef LSR_KECCAK_ROUND

class Sha3_256 {
public:
    // bytes are collected in a rate-sized block and folded into the state 17 lanes at a time
    void update(const void* data, size_t len) {
        const auto* p = static_cast<const uint8_t*>(data);
        while (len) {
            const size_t take = len < kRate - fill_ ? len : kRate - fill_;
            std::memcpy(block_ + fill_, p, take);
            fill_ += take; p += take; len -= take;
            if (fill_ == kRate) absorb_block();
        }
    }
    void update_le64(uint64_t v) { update(&v, 8); }     // little-endian host (asserted above)
    void finish(uint8_t out[32]) {
        std::memset(block_ + fill_, 0, kRate - fill_);
        block_[fill_] ^= 0x06;
        block_[kRate - 1] ^= 0x80;
        fill_ = kRate;
        absorb_block();
        std::memcpy(out, state_, 32);
    }

private:
    static constexpr size_t kRate = 136;
    void absorb_block() {
        for (size_t i = 0; i < kRate / 8; ++i) {
            uint64_t lane;
            std::memcpy(&lane, block_ + 8 * i, 8);
            state_[i] ^= lane;
        }
        keccak_f1600(state_);
        fill_ = 0;
    }
    uint64_t state_[25] = {};
    uint8_t block_[kRate] = {};
    size_t fill_ = 0;
};

// one transcript: challenge.rs:102-134
void derive(const uint64_t* public_inputs, size_t n_inputs, const uint64_t* words, size_t n_words, uint64_t modulus, uint64_t* alpha, uint8_t* hash32) {
    Sha3_256 h;
    h.update("LAMBDA-SNARK-R-FS-v1", 20);                       // :107
    h.update_le64(static_cast<uint64_t>(n_inputs));            // :110
    if (n_inputs) h.update(public_inputs, n_inputs * 8);       // :113-115, each word as its 8 little-endian bytes
    h.update_le64(static_cast<uint64_t>(n_words));             // :119
    h.update(words, n_words * 8);                              // :120-122
    uint8_t digest[32];
    h.finish(digest);
    uint64_t raw;
    std::memcpy(&raw, digest, 8);                              // :129-131 (little-endian host)
    *alpha = raw % modulus;
    if (hash32) std::memcpy(hash32, digest, 32);
}

}  // namespace

// `count` transcripts at once — the commitments as rows of one array (lsr_lwe_commit_batch_flat), the public inputs as
// [count][n_inputs] — hashed by a pool of host threads (SHA3 is sequential within a transcript, independent across them).
extern "C" int lsr_fs_challenge_batch_flat(const uint64_t* public_inputs, size_t n_inputs, const uint64_t* words, size_t words_per_commitment,
                                           size_t count, uint64_t modulus, uint64_t* alphas, uint8_t* hashes32, unsigned threads) noexcept {
    if ((!public_inputs && n_inputs) || !words || words_per_commitment == 0 || modulus == 0 || !alphas) return -1;
    if (count == 0) return 0;
    try {
        unsigned workers = threads ? threads : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
        workers = static_cast<unsigned>(std::min<size_t>(workers, count));
        auto span = [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i)
                derive(n_inputs ? public_inputs + i * n_inputs : nullptr, n_inputs, words + i * words_per_commitment, words_per_commitment, modulus,
                       &alphas[i], hashes32 ? hashes32 + 32 * i : nullptr);
        };
        if (workers <= 1) {
            span(0, count);
            return 0;
        }
        std::vector<std::thread> pool;
        const size_t per = (count + workers - 1) / workers;
        for (unsigned w = 0; w < workers; ++w) {
            const size_t lo = w * per, hi = std::min(count, lo + per);
            if (lo < hi) pool.emplace_back(span, lo, hi);
        }
        for (std::thread& t : pool) t.join();
        return 0;
    } catch (...) {
        return -1;
    }
}

extern "C" int lsr_fs_challenge(const uint64_t* public_inputs, size_t n_inputs, const LweCommitment* commitment, uint64_t modulus,
                                uint64_t* alpha, uint8_t* hash32) noexcept {
    if ((!public_inputs && n_inputs) || !commitment || !commitment->data || modulus == 0 || !alpha) return -1;
    derive(public_inputs, n_inputs, commitment->data, commitment->len, modulus, alpha, hash32);
    return 0;
}
