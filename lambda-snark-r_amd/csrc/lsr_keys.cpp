// Key schedule of the seeded streams — see lsr_keys.hpp.
#include "lsr_keys.hpp"

#include <sys/random.h>

#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace lsr {

namespace {
inline uint32_t rotl(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }
inline void quarter(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d) {
    a += b; d ^= a; d = rotl(d, 16);
    c += d; b ^= c; b = rotl(b, 12);
    a += b; d ^= a; d = rotl(d, 8);
    c += d; b ^= c; b = rotl(b, 7);
}
constexpr uint32_t tag(char a, char b, char c, char d) { return key_tag(a, b, c, d); }
constexpr uint64_t kP61 = kHashPrime61;
inline uint64_t mul61(uint64_t a, uint64_t b) {   // a, b < 2^61 - 1
    const unsigned __int128 w = (unsigned __int128)a * b;
    uint64_t r = (uint64_t)(w & kP61) + (uint64_t)(w >> 61);
    r = (r & kP61) + (r >> 61);
    return r >= kP61 ? r - kP61 : r;
}
inline uint64_t point61(uint32_t lo, uint32_t hi) {
    const uint64_t x = (((uint64_t)hi << 32) | lo) & kP61;
    return x == kP61 ? 0 : x;
}
}  // namespace

void chacha20_block_host(const uint32_t key[8], uint32_t counter, const uint32_t nonce[3], uint32_t out[16]) {
    const uint32_t init[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                               key[4], key[5], key[6], key[7], counter, nonce[0], nonce[1], nonce[2]};
    uint32_t x[16];
    std::memcpy(x, init, sizeof x);
    for (int round = 0; round < 10; ++round) {
        quarter(x[0], x[4], x[8], x[12]); quarter(x[1], x[5], x[9], x[13]); quarter(x[2], x[6], x[10], x[14]); quarter(x[3], x[7], x[11], x[15]);
        quarter(x[0], x[5], x[10], x[15]); quarter(x[1], x[6], x[11], x[12]); quarter(x[2], x[7], x[8], x[13]); quarter(x[3], x[4], x[9], x[14]);
    }
    for (int i = 0; i < 16; ++i) out[i] = x[i] + init[i];
}

StreamKey kdf(const StreamKey& key, uint32_t label, uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t nonce[3] = {a, b, c};
    uint32_t block[16];
    chacha20_block_host(key.w, label, nonce, block);
    StreamKey out;
    std::memcpy(out.w, block, sizeof out.w);
    return out;
}

StreamKey expand_seed64(uint64_t seed) {
    return StreamKey{{(uint32_t)seed, (uint32_t)(seed >> 32), tag('L', 'S', 'R', '1'), tag('S', 'T', 'R', 'M'), 0u, 0u, 0u, 0u}};
}

void os_entropy_fill(void* dst, size_t bytes) {
    unsigned char* p = static_cast<unsigned char*>(dst);
    size_t done = 0;
    while (done < bytes) {
        const ssize_t got = getrandom(p + done, bytes - done, 0);
        if (got <= 0) {   // fall back to the device node
            std::FILE* f = std::fopen("/dev/urandom", "rb");
            if (!f || std::fread(p + done, 1, bytes - done, f) != bytes - done) {
                if (f) std::fclose(f);
                throw std::runtime_error("no entropy source (getrandom and /dev/urandom failed)");
            }
            std::fclose(f);
            return;
        }
        done += static_cast<size_t>(got);
    }
}

StreamKey fresh_key() {
    StreamKey k;
    os_entropy_fill(k.w, sizeof k.w);
    return k;
}

ContextKeys derive_context_keys(uint64_t key_seed) {
    ContextKeys out;
    if (key_seed == 0) {
        out.pub = fresh_key();
        out.sec = fresh_key();
    } else {
        const StreamKey master{{(uint32_t)key_seed, (uint32_t)(key_seed >> 32), tag('L', 'S', 'R', '2'), tag('M', 'S', 'T', 'R'), 0u, 0u, 0u, 0u}};
        out.pub = kdf(master, tag('P', 'U', 'B', 'K'), 0, 0, 0);
        out.sec = kdf(master, tag('S', 'E', 'C', 'K'), 0, 0, 0);
    }
    const StreamKey id = kdf(out.sec, tag('C', 'T', 'I', 'D'), out.pub.w[0], out.pub.w[1], out.pub.w[2]);
    std::memcpy(out.id, id.w, sizeof out.id);
    return out;
}

StreamKey derive_commit_key(uint64_t seed, const uint32_t id[4], const uint64_t* message, size_t copy, uint64_t t) {
    const StreamKey base{{(uint32_t)seed, (uint32_t)(seed >> 32), kTagLsr2, kTagCommit, id[0], id[1], id[2], id[3]}};
    const StreamKey points = kdf(base, kTagHashPoints, 0, 0, 0);
    const uint64_t x1 = point61(points.w[0], points.w[1]), x2 = point61(points.w[2], points.w[3]);
    // h_a = sum_i m_i x_a^(i+1), evaluated as four interleaved Horner chains in y = x^4 per point (one product per word and point, eight
    // independent chains for the multiplier's pipeline; the running-power form of round 2 spent two dependent products per word and
    // point: 10 us per full-length message, now about half): h = sum_r x^(r+1) H_r(y), H_r(y) = sum_j m_(4j+r) y^j
    auto reduce = [t](uint64_t w) { return w < t ? w : w % t; };
    const uint64_t y1 = mul61(mul61(x1, x1), mul61(x1, x1)), y2 = mul61(mul61(x2, x2), mul61(x2, x2));
    uint64_t a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
    size_t i = copy;
    while (i % 4 != 0) {                                    // the ragged top: residues r = i mod 4 that own one word more
        --i;
        a1[i % 4] = reduce(message[i]);
        a2[i % 4] = a1[i % 4];
    }
    while (i >= 4) {
        i -= 4;
        for (int r = 0; r < 4; ++r) {
            const uint64_t m = reduce(message[i + r]);
            uint64_t v1 = mul61(a1[r], y1) + m, v2 = mul61(a2[r], y2) + m;
            a1[r] = v1 >= kP61 ? v1 - kP61 : v1;
            a2[r] = v2 >= kP61 ? v2 - kP61 : v2;
        }
    }
    uint64_t h1 = 0, h2 = 0, p1 = x1, p2 = x2;             // x^(r+1)
    for (int r = 0; r < 4; ++r) {
        h1 += mul61(a1[r], p1); if (h1 >= kP61) h1 -= kP61;
        h2 += mul61(a2[r], p2); if (h2 >= kP61) h2 -= kP61;
        p1 = mul61(p1, x1);
        p2 = mul61(p2, x2);
    }
    const StreamKey step = kdf(base, kTagCommitKey1, (uint32_t)h1, (uint32_t)(h1 >> 32), 0);
    return kdf(step, kTagCommitKey2, (uint32_t)h2, (uint32_t)(h2 >> 32), 0);
}

}  // namespace lsr
