// Keccak-f[1600] (FIPS 202 §3) shared by the host transcript code (lsr_transcript.cpp) and the device kernel that hashes a
// batch of transcripts, one lane per transcript (lsr_transcript_gpu.hip).
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define LSR_HD __host__ __device__
#else
#define LSR_HD
#endif

namespace lsr {

LSR_HD constexpr uint64_t keccak_round_constant(int i) {
    constexpr uint64_t k[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL, 0x000000000000808BULL, 0x0000000080000001ULL,
        0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
        0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
        0x000000000000800AULL, 0x800000008000000AULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    return k[i];
}

template <int S>
LSR_HD inline uint64_t keccak_rotl(uint64_t v) {
    if constexpr (S == 0) return v;
    else return (v << S) | (v >> (64 - S));
}

// one round on 25 named lanes: theta, rho + pi, chi, iota (FIPS 202 §3.2); A -> E
#define LSR_KECCAK_ROUND(A, E, RC)                                                                                    \
    do {                                                                                                              \
        const uint64_t c0 = A[0] ^ A[5] ^ A[10] ^ A[15] ^ A[20], c1 = A[1] ^ A[6] ^ A[11] ^ A[16] ^ A[21],               \
                       c2 = A[2] ^ A[7] ^ A[12] ^ A[17] ^ A[22], c3 = A[3] ^ A[8] ^ A[13] ^ A[18] ^ A[23],               \
                       c4 = A[4] ^ A[9] ^ A[14] ^ A[19] ^ A[24];                                                         \
        const uint64_t d0 = c4 ^ keccak_rotl<1>(c1), d1 = c0 ^ keccak_rotl<1>(c2), d2 = c1 ^ keccak_rotl<1>(c3), d3 = c2 ^ keccak_rotl<1>(c4),       \
                       d4 = c3 ^ keccak_rotl<1>(c0);                                                                            \
        uint64_t b0, b1, b2, b3, b4;                                                                                     \
        b0 = A[0] ^ d0; b1 = keccak_rotl<44>(A[6] ^ d1); b2 = keccak_rotl<43>(A[12] ^ d2); b3 = keccak_rotl<21>(A[18] ^ d3); b4 = keccak_rotl<14>(A[24] ^ d4); \
        E[0] = b0 ^ (~b1 & b2) ^ (RC); E[1] = b1 ^ (~b2 & b3); E[2] = b2 ^ (~b3 & b4); E[3] = b3 ^ (~b4 & b0); E[4] = b4 ^ (~b0 & b1); \
        b0 = keccak_rotl<28>(A[3] ^ d3); b1 = keccak_rotl<20>(A[9] ^ d4); b2 = keccak_rotl<3>(A[10] ^ d0); b3 = keccak_rotl<45>(A[16] ^ d1); b4 = keccak_rotl<61>(A[22] ^ d2); \
        E[5] = b0 ^ (~b1 & b2); E[6] = b1 ^ (~b2 & b3); E[7] = b2 ^ (~b3 & b4); E[8] = b3 ^ (~b4 & b0); E[9] = b4 ^ (~b0 & b1); \
        b0 = keccak_rotl<1>(A[1] ^ d1); b1 = keccak_rotl<6>(A[7] ^ d2); b2 = keccak_rotl<25>(A[13] ^ d3); b3 = keccak_rotl<8>(A[19] ^ d4); b4 = keccak_rotl<18>(A[20] ^ d0); \
        E[10] = b0 ^ (~b1 & b2); E[11] = b1 ^ (~b2 & b3); E[12] = b2 ^ (~b3 & b4); E[13] = b3 ^ (~b4 & b0); E[14] = b4 ^ (~b0 & b1); \
        b0 = keccak_rotl<27>(A[4] ^ d4); b1 = keccak_rotl<36>(A[5] ^ d0); b2 = keccak_rotl<10>(A[11] ^ d1); b3 = keccak_rotl<15>(A[17] ^ d2); b4 = keccak_rotl<56>(A[23] ^ d3); \
        E[15] = b0 ^ (~b1 & b2); E[16] = b1 ^ (~b2 & b3); E[17] = b2 ^ (~b3 & b4); E[18] = b3 ^ (~b4 & b0); E[19] = b4 ^ (~b0 & b1); \
        b0 = keccak_rotl<62>(A[2] ^ d2); b1 = keccak_rotl<55>(A[8] ^ d3); b2 = keccak_rotl<39>(A[14] ^ d4); b3 = keccak_rotl<41>(A[15] ^ d0); b4 = keccak_rotl<2>(A[21] ^ d1); \
        E[20] = b0 ^ (~b1 & b2); E[21] = b1 ^ (~b2 & b3); E[22] = b2 ^ (~b3 & b4); E[23] = b3 ^ (~b4 & b0); E[24] = b4 ^ (~b0 & b1); \
    } while (0)

LSR_HD inline void keccak_f1600(uint64_t (&a)[25]) {
    uint64_t e[25];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int round = 0; round < 24; round += 2) {
        LSR_KECCAK_ROUND(a, e, keccak_round_constant(round));
        LSR_KECCAK_ROUND(e, a, keccak_round_constant(round + 1));
    }
}
#undef LSR_KECCAK_ROUND

constexpr unsigned kSha3Rate = 136;   // SHA3-256: 1088-bit rate

}  // namespace lsr
