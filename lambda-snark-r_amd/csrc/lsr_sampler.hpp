// Seeded randomness on the device: ChaCha20 word stream, uniform residues, CDT discrete Gaussian.
//
// The reference sampler (cpp-core/src/utils.cpp:95-146) draws two 64-bit words per sample from
// std::random_device: one compared against the CDT table (first k with cdf[k] >= u), one whose low bit is
// the sign.  Here the words come from a counter-based ChaCha20 stream so that a seed reproduces the
// output on any device: object (key, domain, index) owns a stream; 64-bit word w of it is ChaCha block
// w/8, 32-bit words 2(w%8) (low half) and 2(w%8)+1 (high half).  Sample i uses ONE word, word i: its low
// bit is the sign, its upper 63 bits are the uniform value compared with the table at 63-bit precision
// (first k with cdf[k] >> 1 >= word >> 1) — half the cipher work of spending a second word on one sign
// bit; the table itself is only accurate to about 2^-63 (long double accumulation, utils.cpp:26-75).
//   key   = 256 bits from the key schedule of lsr_keys.hpp (per context / per commitment), or the expansion
//           { seed_lo, seed_hi, "LSR1", "STRM", 0, 0, 0, 0 } of a raw 64-bit test seed
//   nonce = { domain, index_lo, index_hi }          counter = block number
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

namespace lsr {

enum StreamDomain : uint32_t { kDomA = 1, kDomS = 2, kDomE = 3, kDomR = 4, kDomE1 = 5, kDomE2 = 6, kDomUser = 16 };

__device__ __forceinline__ uint32_t rotl32(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }

__device__ __forceinline__ void chacha_quarter(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d) {
    a += b; d ^= a; d = rotl32(d, 16);
    c += d; b ^= c; b = rotl32(b, 12);
    a += b; d ^= a; d = rotl32(d, 8);
    c += d; b ^= c; b = rotl32(b, 7);
}

// RFC 8439 §2.3 block function; returns the block as eight little-endian 64-bit words.
// key4: the 256-bit key as four little-endian 64-bit words
__device__ __forceinline__ void stream_block(const uint64_t* __restrict__ key4, uint32_t domain, uint64_t index, uint32_t block, uint64_t (&w)[8]) {
    const uint64_t k0 = key4[0], k1 = key4[1], k2 = key4[2], k3 = key4[3];
    const uint32_t init[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u,
                               (uint32_t)k0, (uint32_t)(k0 >> 32), (uint32_t)k1, (uint32_t)(k1 >> 32), (uint32_t)k2, (uint32_t)(k2 >> 32),
                               (uint32_t)k3, (uint32_t)(k3 >> 32), block, domain, (uint32_t)index, (uint32_t)(index >> 32)};
    uint32_t x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = init[i];
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        chacha_quarter(x[0], x[4], x[8], x[12]);
        chacha_quarter(x[1], x[5], x[9], x[13]);
        chacha_quarter(x[2], x[6], x[10], x[14]);
        chacha_quarter(x[3], x[7], x[11], x[15]);
        chacha_quarter(x[0], x[5], x[10], x[15]);
        chacha_quarter(x[1], x[6], x[11], x[12]);
        chacha_quarter(x[2], x[7], x[8], x[13]);
        chacha_quarter(x[3], x[4], x[9], x[14]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = (uint64_t)(x[2 * j] + init[2 * j]) | ((uint64_t)(x[2 * j + 1] + init[2 * j + 1]) << 32);
}

// Magnitudes of COUNT samples at once: first k with cdf63[k] >= u[s] (cdf63 = table >> 1, non-decreasing, last entry 2^63 - 1;
// u = stream word >> 1) — the value the reference's scan selects (utils.cpp:101-108) — computed the way the reference computes
// it: a branch-free pass over the WHOLE table, count += (cdf63[k] < u).  Every lane reads the same LDS word per step (a
// broadcast), so neither the instruction stream nor the LDS access pattern depends on the secret uniform words.
template <int COUNT>
__device__ __forceinline__ void cdt_scan(const uint64_t* cdf63, uint32_t entries, const uint64_t (&u)[COUNT], uint32_t (&magnitude)[COUNT]) {
#pragma unroll
    for (int s = 0; s < COUNT; ++s) magnitude[s] = 0;
    for (uint32_t k = 0; k + 1 < entries; ++k) {      // the last entry is 2^63 - 1: never below u
        const uint64_t c = cdf63[k];
#pragma unroll
        for (int s = 0; s < COUNT; ++s) magnitude[s] += (c < u[s]) ? 1u : 0u;
    }
}

// The same count as a branch-free BINARY SEARCH over a table that lives in the lanes of the wavefront (round 3): lane l holds
// entry l (63-bit, padded with 2^63 - 1, which is never below u), and a probe is a ds_bpermute_b32 gather — the LDS crossbar, no LDS
// memory, no banks, so neither the instruction stream nor any memory access pattern depends on the secret words, exactly as in
// the linear pass.  STEPS = 5 serves tables of <= 32 entries (sigma <= ~3.4: every reference parameter set), STEPS = 6 <= 64
// entries; a sample costs about 15 VALU + 10 DS instructions (6-step form: 27 + 22) instead of 2 (entries - 1) VALU ones, and the gathers run on the
// LDS pipe beside the cipher's VALU work: 209 -> 276 G samples/s with the ChaCha20 stream (tools/ubench_sampler.hip,
// profiles/r03_ubench_sampler.txt: the cipher alone 327 G/s).  EVERY lane of the wavefront must be active (an inactive source
// lane reads as zero).  The table is non-decreasing (a cumulative sum), which is all the search needs.
struct LaneTable {
    uint32_t lo, hi;       // entry l (l = lane % 32), 63-bit, padded with 2^63 - 1
    uint32_t lo2, hi2;     // entry 32 + l (tables of 33..64 entries; padding otherwise)
};
__device__ __forceinline__ uint32_t lane_table_steps(uint32_t entries) { return entries <= 32u ? 5u : (entries <= 64u ? 6u : 0u); }   // 0: use cdt_scan
// cdf: the 64-bit table (global or LDS); entries: scanned entries (the last one is all ones).
// Every probe reads from lanes 0..31 only: a ds_bpermute whose source lanes l and l + 32 differ lands on one LDS bank twice and takes
// longer (tools/ubench_sampler.hip: 28.4 against 24.3 cycles for random sources among all 64 lanes, 24.3 for any pattern inside
// 0..31, broadcasts included), which would make the search time depend on the secret words.  A 64-entry table therefore lives in
// TWO register pairs of the same 32 lanes, and the 6-step search fetches both candidates and selects.
__device__ __forceinline__ LaneTable lane_table_load(const uint64_t* cdf, uint32_t entries) {
    const uint32_t l = threadIdx.x & 31u;
    const uint64_t a = l < entries ? cdf[l] >> 1 : (~0ull >> 1);
    const uint64_t b = l + 32u < entries ? cdf[l + 32u] >> 1 : (~0ull >> 1);
    return LaneTable{(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
}
template <int STEPS, int COUNT>
__device__ __forceinline__ void cdt_search(const LaneTable& tab, const uint64_t (&u)[COUNT], uint32_t (&magnitude)[COUNT]) {
    static_assert(STEPS == 5 || STEPS == 6, "tables of up to 32 or up to 64 entries");
#pragma unroll
    for (int s = 0; s < COUNT; ++s) {
        // STEPS == 6: the first probe is entry 31 for every sample (lane 31, a broadcast); it decides which half of the table the
        // remaining five steps search
        bool upper = false;
        if constexpr (STEPS == 6) {
            const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(31 * 4, (int)tab.lo);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(31 * 4, (int)tab.hi);
            upper = (((uint64_t)hi << 32) | lo) < u[s];
        }
        int at = 15 * 4;                                     // byte address of the probed lane: (found + step - 1) * 4
#pragma unroll
        for (int step = 16; step >= 1; step >>= 1) {
            uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)tab.lo);
            uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)tab.hi);
            if constexpr (STEPS == 6) {
                const uint32_t lo2 = (uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)tab.lo2);
                const uint32_t hi2 = (uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)tab.hi2);
                lo = upper ? lo2 : lo;
                hi = upper ? hi2 : hi;
            }
            const bool below = (((uint64_t)hi << 32) | lo) < u[s];
            at += step > 1 ? (below ? 2 * step : -2 * step) : (below ? 4 : 0);
        }
        magnitude[s] = ((uint32_t)at >> 2) + (upper ? 32u : 0u);
    }
}
// magnitudes of the eight samples of one stream block, by whichever form the table's size admits (wavefront-uniform choice)
__device__ __forceinline__ void cdt_magnitudes(const LaneTable& tab, const uint64_t* cdf63_lds, uint32_t entries, const uint64_t (&u)[8],
                                               uint32_t (&magnitude)[8]) {
    const uint32_t steps = lane_table_steps(entries);
    if (steps == 5u) cdt_search<5, 8>(tab, u, magnitude);
    else if (steps == 6u) cdt_search<6, 8>(tab, u, magnitude);
    else cdt_scan<8>(cdf63_lds, entries, u, magnitude);
}

// sign applied to a magnitude without a branch (utils.cpp:114-120): residue in [0,q), or two's-complement int64 when q == 0
__device__ __forceinline__ uint64_t gaussian_value(uint32_t magnitude, uint64_t word, uint64_t q) {
    const uint64_t m = magnitude;
    const uint64_t sign = (word & 1ull) & (uint64_t)(m != 0);
    return q ? (sign ? q - m : m) : (m ^ (0ull - sign)) + sign;
}

constexpr int kSamplesPerBlock = 8;     // one ChaCha20 block = eight 64-bit words = eight samples

struct GaussianJob {
    uint64_t* out;            // [objects][samples]
    const uint64_t* keys;     // device array [groups][4]: the 256-bit stream key of object o is keys[4 (o / components) ..]
    uint64_t index_base;      // stream index of object o = index_base + o % components
    uint32_t components;
    uint32_t domain;
    uint64_t samples;         // per object
    uint64_t objects;
    uint64_t q;               // 0: two's-complement int64 (utils.cpp:142); else residue in [0,q)
};

// the same sampler run INSIDE the pass that consumes the samples (ntt_strided_round_sampled, lsr_ntt_kernels.hpp): object = polynomial
struct BlindSampler {
    const uint64_t* keys = nullptr;     // [objects / components][4]
    const uint64_t* cdf = nullptr;      // CDT table (64-bit thresholds)
    uint32_t entries = 0, components = 1, domain = 0;
    uint64_t* side = nullptr;           // split sampling: int8 samples of the first half of the rows, [polynomial][column][2^R / 16] words
};

void launch_gaussian(const GaussianJob& job, const uint64_t* d_cdf, uint32_t entries, hipStream_t stream);
// three independent jobs in one launch (same table)
void launch_gaussian3(const GaussianJob& a, const GaussianJob& b, const GaussianJob& c, const uint64_t* d_cdf, uint32_t entries, hipStream_t stream);
// out[o][i] = floor(word_i * q / 2^64), object o = (keys[4 (o / components) ..], domain, index_base + o % components)
void launch_uniform(uint64_t* out, const uint64_t* d_keys, uint64_t index_base, uint32_t components, uint32_t domain,
                    uint64_t samples, uint64_t objects, uint64_t q, hipStream_t stream);

}  // namespace lsr
