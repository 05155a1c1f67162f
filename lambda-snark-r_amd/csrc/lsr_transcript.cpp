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
#include "lsr_keccak.hpp"

namespace {

static_assert(__BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__, "the transcript absorbs host words as their little-endian bytes");
using lsr::keccak_f1600;

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
