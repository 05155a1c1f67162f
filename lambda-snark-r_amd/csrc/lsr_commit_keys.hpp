// Per-commitment stream keys derived ON THE DEVICE (round 3): derive_commit_key of lsr_keys.cpp for a batch whose messages are
// device-resident.  The host derivation walks the embedded message twice-multiplying mod 2^61 - 1 per slot — about 10 us per
// full-length message (n = 4096) on one core, 20 ms for a batch of 16384 on eight threads, against 2.2 ms for the commitments
// themselves (commit_tile_kernel): with long messages the key schedule, not the GPU, set the rate of lsr_lwe_commit_rows_device.
// One wavefront per commitment:
//     base   = {seed, "LSR2", "CMIT", context id}                               (ChaCha20 key)
//     x1, x2 = first two words of block(base; counter "HPNT") mod 2^61 - 1       (secret evaluation points)
//     h_a    = sum_i (m_i mod t) x_a^(i+1) mod 2^61 - 1                          lane l takes i = l, l + 64, ...: Horner in x^64 from the
//                                                                                top, times x^(l+1); wavefront reduction
//     key    = block( block(base; "CKY1", h1) ; "CKY2", h2 )                     (lane 0)
// Word for word the host's keys (tests/test_commit_rows_gpu.py); seed == 0 (fresh OS entropy, commitment.h:52) stays a host matter.
#pragma once

#include "lsr_commit_tile.hpp"
#include "lsr_keys.hpp"

namespace lsr {

constexpr int kKeyThreads = 64;      // one wavefront per commitment: the three cipher blocks cost a wavefront the same whether 1 or 64 lanes need them

__device__ __forceinline__ uint64_t mul61_dev(uint64_t a, uint64_t b) {     // a, b < 2^61 - 1 (lsr_keys.cpp mul61)
    const uint64_t lo = a * b, hi = __umul64hi(a, b);                       // hi < 2^58
    uint64_t r = (lo & kHashPrime61) + ((lo >> 61) | (hi << 3));
    r = (r & kHashPrime61) + (r >> 61);
    return r >= kHashPrime61 ? r - kHashPrime61 : r;
}
__device__ __forceinline__ uint64_t add61_dev(uint64_t a, uint64_t b) {     // a, b < 2^61 - 1
    const uint64_t r = a + b;
    return r >= kHashPrime61 ? r - kHashPrime61 : r;
}
__device__ __forceinline__ uint64_t point61_dev(uint64_t word) {
    const uint64_t x = word & kHashPrime61;
    return x == kHashPrime61 ? 0 : x;
}
// x^e for 1 <= e <= 64 and, from the same chain of squarings, *step = x^64
__device__ __forceinline__ uint64_t pow61_dev(uint64_t x, uint32_t e, uint64_t* step) {
    uint64_t result = 1, square = x;
#pragma unroll
    for (int bit = 0; bit < 6; ++bit) {
        if ((e >> bit) & 1u) result = mul61_dev(result, square);
        square = mul61_dev(square, square);
    }
    *step = square;                                        // x^(2^6)
    return (e >> 6) ? square : result;                     // e = 64 is the only exponent with bit 6 set
}
static_assert(kKeyThreads == 64, "pow61_dev and the reduction of commit_keys_kernel assume one wavefront per commitment");

struct CommitKeysJob {
    uint64_t* keys;              // [batch][4]
    const uint64_t* msgs;        // [batch][msg_len]; only read when copy > 0
    const uint64_t* seeds;       // [batch], every one non-zero (checked on the host)
    uint64_t msg_len;
    uint32_t copy;               // min(msg_len, n) embedded slots (commitment.cpp:146-149)
    uint32_t id[4];              // context id (lsr_keys.hpp ContextKeys)
    uint64_t t;
};

__global__ void __launch_bounds__(kKeyThreads) commit_keys_kernel(CommitKeysJob job) {
    const uint32_t lane = threadIdx.x;
    const size_t j = blockIdx.x;
    const uint64_t seed = job.seeds[j];
    const uint64_t base[4] = {seed, (uint64_t)kTagLsr2 | ((uint64_t)kTagCommit << 32), (uint64_t)job.id[0] | ((uint64_t)job.id[1] << 32),
                              (uint64_t)job.id[2] | ((uint64_t)job.id[3] << 32)};
    uint64_t w[8];
    stream_block(base, 0u, 0ull, kTagHashPoints, w);                        // kdf(base, "HPNT", 0, 0, 0): the same in every lane
    const uint64_t x1 = point61_dev(w[0]), x2 = point61_dev(w[1]);
    uint64_t h1 = 0, h2 = 0;
    if (lane < job.copy) {
        // this lane's words m_l, m_(l+64), ... by Horner's rule in y = x^64, highest index first (one product per word and point; the
        // straightforward sum of m_i x^(i+1) with a running power costs two), then one multiplication by x^(l+1)
        const PlainScale scale = make_plain_scale(job.t, job.t);            // only t and 1/t are used by mod_plain
        uint64_t s1, s2;
        const uint64_t p1 = pow61_dev(x1, lane + 1u, &s1), p2 = pow61_dev(x2, lane + 1u, &s2);
        const uint64_t* const msg = job.msgs + j * job.msg_len + lane;
        uint64_t a1 = 0, a2 = 0;
        for (uint32_t jj = (job.copy - lane + kKeyThreads - 1u) / kKeyThreads; jj-- > 0;) {
            const uint64_t m = (uint64_t)mod_plain(msg[(size_t)jj * kKeyThreads], scale);   // < t < 2^21
            a1 = add61_dev(mul61_dev(a1, s1), m);
            a2 = add61_dev(mul61_dev(a2, s2), m);
        }
        h1 = mul61_dev(a1, p1);
        h2 = mul61_dev(a2, p2);
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) {
        h1 = add61_dev(h1, (uint64_t)__shfl_xor((unsigned long long)h1, off));
        h2 = add61_dev(h2, (uint64_t)__shfl_xor((unsigned long long)h2, off));
    }
    if (lane == 0) {
        uint64_t step[8], key[8];
        stream_block(base, (uint32_t)h1, h1 >> 32, kTagCommitKey1, step);   // kdf(base, "CKY1", lo(h1), hi(h1), 0)
        stream_block(step, (uint32_t)h2, h2 >> 32, kTagCommitKey2, key);    // kdf(step, "CKY2", lo(h2), hi(h2), 0)
#pragma unroll
        for (int i = 0; i < 4; ++i) job.keys[4 * j + i] = key[i];
    }
}

}  // namespace lsr
