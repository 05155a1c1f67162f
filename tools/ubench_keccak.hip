// Microbenchmark: SHA3-256 of many independent messages on the GPU, ONE lane per message (state in registers, no cross-lane
// traffic).  2048 messages of 12293 u64 words (the reference-size commitment) with the 4-byte phase of the transcript
// (a 20-byte tag in front).  Reports time and checks one digest against a host implementation.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_keccak.hip -o tools/bin/ubench_keccak
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__host__ __device__ constexpr uint64_t rc(int i) {
    constexpr uint64_t k[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL, 0x000000000000808BULL,
                                0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008AULL, 0x0000000000000088ULL,
                                0x0000000080008009ULL, 0x000000008000000AULL, 0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL,
                                0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
                                0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    return k[i];
}
template <int S> __host__ __device__ inline uint64_t rotl(uint64_t v) { if constexpr (S == 0) return v; else return (v << S) | (v >> (64 - S)); }

#define ROUND(A, E, RC)                                                                                                  \
    do {                                                                                                                 \
        const uint64_t c0 = A[0] ^ A[5] ^ A[10] ^ A[15] ^ A[20], c1 = A[1] ^ A[6] ^ A[11] ^ A[16] ^ A[21],                  \
                       c2 = A[2] ^ A[7] ^ A[12] ^ A[17] ^ A[22], c3 = A[3] ^ A[8] ^ A[13] ^ A[18] ^ A[23],                  \
                       c4 = A[4] ^ A[9] ^ A[14] ^ A[19] ^ A[24];                                                            \
        const uint64_t d0 = c4 ^ rotl<1>(c1), d1 = c0 ^ rotl<1>(c2), d2 = c1 ^ rotl<1>(c3), d3 = c2 ^ rotl<1>(c4), d4 = c3 ^ rotl<1>(c0); \
        uint64_t b0, b1, b2, b3, b4;                                                                                        \
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

__host__ __device__ inline void keccak_f(uint64_t (&a)[25]) {
    uint64_t e[25];
#pragma unroll 1
    for (int r = 0; r < 24; r += 2) { ROUND(a, e, rc(r)); ROUND(e, a, rc(r + 1)); }
}

// the byte stream is tag(20) || W[0..M); its 64-bit units are U[0] = tag[0..8), U[1] = tag[8..16),
// U[2] = tag[16..20) | lo32(W[0]) << 32, U[t] = hi32(W[t-3]) | lo32(W[t-2]) << 32 for t >= 3; data ends after 20 + 8M bytes.

__global__ void __launch_bounds__(64) hash_rows(const uint64_t* __restrict__ rows, size_t M, size_t count, const uint64_t* __restrict__ tag_units,
                                                uint64_t* __restrict__ digests) {
    const size_t r = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (r >= count) return;
    const uint64_t* W = rows + r * M;
    uint64_t a[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) a[i] = 0;
    const size_t total_bytes = 20 + 8 * M;
    const size_t full_blocks = total_bytes / 136;
    // unit t (t >= 3) = hi32(W[t-3]) | lo32(W[t-2]) << 32; W[-1], W[-2].. handled through tag_units for t < 3
    uint64_t prev = 0;   // W[t-3] of the next unit
    size_t t = 0;
    for (size_t blk = 0; blk < full_blocks; ++blk) {
#pragma unroll
        for (int i = 0; i < 17; ++i, ++t) {
            uint64_t u;
            if (t < 2) u = tag_units[t];
            else if (t == 2) { prev = W[0]; u = tag_units[2] | (prev << 32); }
            else { const uint64_t next = W[t - 2]; u = (prev >> 32) | (next << 32); prev = next; }
            a[i] ^= u;
        }
        keccak_f(a);
    }
    // last, padded block: remaining data bytes rem = total - 136 full (rem = 4 mod 8)
    const size_t rem = total_bytes - full_blocks * 136;
#pragma unroll
    for (int i = 0; i < 17; ++i, ++t) {
        const size_t off = (size_t)i * 8;
        uint64_t u = 0;
        if (off + 8 <= rem) {          // full data unit
            if (t < 2) u = tag_units[t];
            else if (t == 2) { prev = W[0]; u = tag_units[2] | (prev << 32); }
            else { const uint64_t next = W[t - 2]; u = (prev >> 32) | (next << 32); prev = next; }
        } else if (off < rem) {        // the unit where the data ends (4 data bytes), then the 0x06 domain byte
            u = (t >= 3 ? (prev >> 32) : tag_units[2]) | (0x06ULL << 32);
        } else if (off == rem) {
            u = 0x06ULL;
        }
        if (i == 16) u ^= 0x8000000000000000ULL;
        a[i] ^= u;
    }
    keccak_f(a);
    for (int i = 0; i < 4; ++i) digests[r * 4 + i] = a[i];
}

// host reference (byte-wise sponge)
static void host_sha3(const uint8_t* msg, size_t len, uint64_t out[4]) {
    uint64_t a[25] = {0};
    uint8_t block[136];
    size_t pos = 0;
    while (len - pos >= 136) {
        for (int i = 0; i < 17; ++i) { uint64_t l; memcpy(&l, msg + pos + 8 * i, 8); a[i] ^= l; }
        keccak_f(a); pos += 136;
    }
    memset(block, 0, 136); memcpy(block, msg + pos, len - pos); block[len - pos] ^= 0x06; block[135] ^= 0x80;
    for (int i = 0; i < 17; ++i) { uint64_t l; memcpy(&l, block + 8 * i, 8); a[i] ^= l; }
    keccak_f(a);
    memcpy(out, a, 32);
}

int main() {
    const size_t count = 2048, M = 12293 + 2;   // [n_inputs = 0][n_words][words...]: what the transcript hashes after the tag
    std::vector<uint64_t> rows(count * M);
    for (size_t i = 0; i < rows.size(); ++i) rows[i] = i * 0x9E3779B97F4A7C15ULL + 12345;
    const char* tag = "LAMBDA-SNARK-R-FS-v1";
    uint64_t tag_units[3] = {0, 0, 0};
    memcpy(&tag_units[0], tag, 8); memcpy(&tag_units[1], tag + 8, 8); memcpy(&tag_units[2], tag + 16, 4);
    uint64_t *d_rows, *d_tag, *d_dig;
    CK(hipMalloc(&d_rows, rows.size() * 8)); CK(hipMalloc(&d_tag, 24)); CK(hipMalloc(&d_dig, count * 32));
    CK(hipMemcpy(d_rows, rows.data(), rows.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_tag, tag_units, 24, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(hash_rows, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, 0, d_rows, M, count, d_tag, d_dig);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    std::vector<uint64_t> dig(count * 4);
    CK(hipMemcpy(dig.data(), d_dig, count * 32, hipMemcpyDeviceToHost));
    int bad = 0;
    for (size_t r : {size_t(0), size_t(1), size_t(777), count - 1}) {
        std::vector<uint8_t> msg(20 + 8 * M);
        memcpy(msg.data(), tag, 20); memcpy(msg.data() + 20, rows.data() + r * M, 8 * M);
        uint64_t want[4]; host_sha3(msg.data(), msg.size(), want);
        bad += memcmp(want, &dig[r * 4], 32) != 0;
    }
    printf("one lane per transcript: %zu transcripts of %zu bytes in %.3f ms = %.1f K transcripts/s, %.2f GB/s; digest mismatches: %d\n", count,
           20 + 8 * M, best, count / best, count * (20 + 8 * M) / best / 1e6, bad);
    return bad != 0;
}
