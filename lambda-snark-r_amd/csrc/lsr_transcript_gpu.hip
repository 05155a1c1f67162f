// A batch of Fiat–Shamir transcripts on the device (rust-api/lambda-snark/src/challenge.rs:102-134), for commitments that are
// already there as rows of one array (lsr_lwe_commit_batch_flat_device).  SHA3-256 is sequential inside a transcript, so the
// parallelism is across transcripts: ONE LANE per transcript with the 1600-bit state in registers and no cross-lane traffic
// (tools/ubench_keccak: 2048 reference-size transcripts in 9.6 ms, and the same time up to 65 536 of them — a wavefront per
// SIMD — where 16 host threads need 21 ms per 2048).
#include "lambda_snark/batch.h"
#include "lsr_keccak.hpp"
#include "lsr_runtime.hpp"

namespace lsr {

// The transcript is tag(20 bytes) || V[0..M) with the virtual word array
//   V = [n_inputs][inputs...][n_words][words...],  M = n_inputs + n_words + 2,
// i.e. its 64-bit units are U[0] = tag[0..8), U[1] = tag[8..16), U[2] = tag[16..20) | lo32(V[0]) << 32 and
// U[t] = hi32(V[t-3]) | lo32(V[t-2]) << 32 for t >= 3; the data ends 4 bytes into unit M + 2.
struct TranscriptView {
    const uint64_t* inputs;
    const uint64_t* words;
    uint64_t n_inputs, n_words;
    __device__ __forceinline__ uint64_t at(uint64_t i) const {
        if (i == 0) return n_inputs;
        if (i <= n_inputs) return inputs[i - 1];
        if (i == n_inputs + 1) return n_words;
        return words[i - n_inputs - 2];
    }
};

__global__ void __launch_bounds__(64) fs_challenge_rows_kernel(const uint64_t* __restrict__ public_inputs, uint64_t n_inputs,
                                                                const uint64_t* __restrict__ words, uint64_t words_per_commitment, uint64_t count,
                                                                uint64_t modulus, uint64_t* __restrict__ alphas, uint64_t* __restrict__ hashes) {
    const uint64_t r = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (r >= count) return;
    const TranscriptView v{public_inputs + r * n_inputs, words + r * words_per_commitment, n_inputs, words_per_commitment};
    const uint64_t M = n_inputs + words_per_commitment + 2;
    const uint64_t tag0 = 0x532D4144424D414CULL, tag1 = 0x462D522D4B52414EULL, tag2 = 0x31762D53ULL;   // "LAMBDA-S" "NARK-R-F" "S-v1"
    uint64_t a[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) a[i] = 0;
    const uint64_t total_bytes = 20 + 8 * M;
    const uint64_t full_blocks = total_bytes / kSha3Rate;
    const uint64_t bulk_from = n_inputs + 2;          // V[i] = words[i - bulk_from] from here on
    uint64_t prev = 0;                                // V[t - 3] of the unit about to be formed
    uint64_t t = 0;
    for (uint64_t blk = 0; blk < full_blocks; ++blk) {
        if (t >= bulk_from + 3) {                     // every unit of this block comes from the commitment words alone
            const uint64_t* w = v.words + (t - 2 - bulk_from);
#pragma unroll
            for (int i = 0; i < 17; ++i) {
                const uint64_t next = w[i];
                a[i] ^= (prev >> 32) | (next << 32);
                prev = next;
            }
            t += 17;
        } else {
#pragma unroll
            for (int i = 0; i < 17; ++i, ++t) {
                uint64_t u;
                if (t == 0) u = tag0;
                else if (t == 1) u = tag1;
                else {
                    const uint64_t next = v.at(t - 2);
                    u = (t == 2 ? tag2 : (prev >> 32)) | (next << 32);
                    prev = next;
                }
                a[i] ^= u;
            }
        }
        keccak_f1600(a);
    }
    // last, padded block: the data ends 4 bytes into one of its units (total = 4 mod 8)
    const uint64_t rem = total_bytes - full_blocks * kSha3Rate;
#pragma unroll
    for (int i = 0; i < 17; ++i, ++t) {
        const uint64_t off = (uint64_t)i * 8;
        uint64_t u = 0;
        if (off + 8 <= rem) {
            if (t == 0) u = tag0;
            else if (t == 1) u = tag1;
            else {
                const uint64_t next = v.at(t - 2);
                u = (t == 2 ? tag2 : (prev >> 32)) | (next << 32);
                prev = next;
            }
        } else if (off < rem) {
            u = (t == 2 ? tag2 : (prev >> 32)) | (0x06ULL << 32);      // 4 data bytes, then the SHA-3 domain byte
        }
        if (i == 16) u ^= 0x8000000000000000ULL;
        a[i] ^= u;
    }
    keccak_f1600(a);
    alphas[r] = a[0] % modulus;                                          // challenge.rs:129-133
    if (hashes) {
#pragma unroll
        for (int i = 0; i < 4; ++i) hashes[r * 4 + i] = a[i];
    }
}

}  // namespace lsr

extern "C" int lsr_fs_challenge_batch_device(const uint64_t* d_public_inputs, size_t n_inputs, const uint64_t* d_words, size_t words_per_commitment,
                                             size_t count, uint64_t modulus, uint64_t* d_alphas, uint8_t* d_hashes32, void* stream) noexcept {
    if ((!d_public_inputs && n_inputs) || !d_words || words_per_commitment == 0 || modulus == 0 || !d_alphas) return -1;
    if (count == 0) return 0;
    try {
        hipLaunchKernelGGL(lsr::fs_challenge_rows_kernel, dim3(static_cast<unsigned>((count + 63) / 64)), dim3(64), 0, static_cast<hipStream_t>(stream),
                           d_public_inputs, (uint64_t)n_inputs, d_words, (uint64_t)words_per_commitment, (uint64_t)count, modulus, d_alphas,
                           reinterpret_cast<uint64_t*>(d_hashes32));
        LSR_HIP(hipGetLastError());
        return 0;
    } catch (const std::exception& e) {
        lsr::set_last_error(std::string("lsr_fs_challenge_batch_device: ") + e.what());
        return -1;
    } catch (...) {
        return -1;
    }
}
