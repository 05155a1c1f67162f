// Batched negacyclic NTT kernels for gfx950 (replaces SEAL's ntt_negacyclic_harvey /
// inverse_ntt_negacyclic_harvey behind cpp-core/src/ntt.cpp:84,99).
//
// Decomposition.  A length-n = 2^L transform is L butterfly stages; stage s pairs coefficients whose
// index differs in bit (L-1-s) and multiplies by twiddle[2^s + (index >> (L-s))].  Stages are grouped
// into radix-16 ROUNDS of four consecutive index bits that one lane holds in 16 registers:
//   * tile kernel   — one 256-lane workgroup owns 4096 contiguous residues (one n=4096 polynomial, or
//     4096/n smaller ones, or one 4096-block of a larger one) and runs every round on index bits
//     [0,12) with the tile staged in LDS between rounds (padded so that every ds_read/ds_write_b64
//     of a round is bank-conflict free); global traffic is fully coalesced on both ends.
//   * strided round kernel — index bits >= 12 (n > 4096): 16 registers hold residues n/16 apart, all
//     256 lanes of a workgroup walk consecutive addresses, no LDS.
// The intermediate array between two kernels of one transform is private, so it is left in the
// arithmetic's raw element form (f64 bit patterns for ArithF64) — no conversion at pass boundaries.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>

#include "lsr_arith.hpp"

namespace lsr {

constexpr int kTileLog = 12;
constexpr uint32_t kTile = 1u << kTileLog;
constexpr int kThreads = 256;
constexpr int kRegs = 16;
constexpr uint32_t kLdsWords = kTile + (kTile >> 4);   // one pad word per 16

template <class A>
struct RoundConsts {
    typename A::twid n_inv;          // n^-1
    typename A::twid w_last_scaled;  // inverse twiddle of the last GS stage times n^-1
};

__device__ __forceinline__ uint32_t lds_slot(uint32_t idx) { return idx + (idx >> 4); }

template <class A> __device__ __forceinline__ uint64_t elem_bits(typename A::elem v);
template <> __device__ __forceinline__ uint64_t elem_bits<ArithF64>(double v) { return (uint64_t)__double_as_longlong(v); }
template <> __device__ __forceinline__ uint64_t elem_bits<ArithU64>(uint64_t v) { return v; }
template <class A> __device__ __forceinline__ typename A::elem elem_from_bits(uint64_t b);
template <> __device__ __forceinline__ double elem_from_bits<ArithF64>(uint64_t b) { return __longlong_as_double((long long)b); }
template <> __device__ __forceinline__ uint64_t elem_from_bits<ArithU64>(uint64_t b) { return b; }

// Tile index of register k for lane t in a round that keeps bits [LO, LO+R) in registers.  With R < 4
// a lane carries 2^(4-R) independent groups; their selector goes to the top tile bits [8+R, 12).
template <int LO, int R>
__device__ __forceinline__ uint32_t lane_base(uint32_t t) {
    return (t & ((1u << LO) - 1u)) | ((t >> LO) << (LO + R));
}
template <int LO, int R>
__host__ __device__ constexpr uint32_t reg_offset(int k) {
    return (uint32_t)((k & ((1 << R) - 1)) << LO) | (uint32_t)((k >> R) << (8 + R));
}

// ---- butterflies of one round, forward (Cooley–Tukey, high bit first) -----------------------------
template <class A, int LO, int R>
__device__ __forceinline__ void forward_round(typename A::elem (&v)[kRegs], uint32_t base_idx, uint32_t block_pos,
                                              uint32_t nmask, const ModParams& p, const typename A::twid* __restrict__ tw) {
    constexpr int G = 1 << (4 - R);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const uint32_t pos0 = (block_pos + (base_idx | ((uint32_t)g << (8 + R)))) & nmask;
#pragma unroll
        for (int b = LO + R - 1; b >= LO; --b) {
            const int half = 1 << (b - LO);
            const uint32_t tw_base = (1u << (p.logn - 1 - b)) + (pos0 >> (b + 1));
#pragma unroll
            for (int u = 0; u < (1 << (LO + R - 1 - b)); ++u) {
                const typename A::twid w = A::load_tw(tw, tw_base + u);
#pragma unroll
                for (int l = 0; l < half; ++l) {
                    const int kx = (g << R) | (u << (b - LO + 1)) | l;
                    A::ct(v[kx], v[kx + half], w, p);
                }
            }
        }
    }
}

// ---- butterflies of one round, inverse (Gentleman–Sande, low bit first) ---------------------------
template <class A, int LO, int R>
__device__ __forceinline__ void inverse_round(typename A::elem (&v)[kRegs], uint32_t base_idx, uint32_t block_pos,
                                              uint32_t nmask, const ModParams& p, const typename A::twid* __restrict__ tw,
                                              const RoundConsts<A>& cs) {
    constexpr int G = 1 << (4 - R);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const uint32_t pos0 = (block_pos + (base_idx | ((uint32_t)g << (8 + R)))) & nmask;
#pragma unroll
        for (int b = LO; b < LO + R; ++b) {
            const int half = 1 << (b - LO);
            const bool last_stage = (b == p.logn - 1);   // wave-uniform
            const uint32_t tw_base = (1u << (p.logn - 1 - b)) + (pos0 >> (b + 1));
#pragma unroll
            for (int u = 0; u < (1 << (LO + R - 1 - b)); ++u) {
                const typename A::twid w = A::load_tw(tw, tw_base + u);
#pragma unroll
                for (int l = 0; l < half; ++l) {
                    const int kx = (g << R) | (u << (b - LO + 1)) | l;
                    if (last_stage) A::gs_scaled(v[kx], v[kx + half], cs.w_last_scaled, cs.n_inv, p);
                    else A::gs(v[kx], v[kx + half], w, p);
                }
            }
        }
    }
}

// ---- tile kernel -----------------------------------------------------------------------------------
// LT = number of low index bits this kernel transforms (min(L,12)).  RAW_IN / RAW_OUT: the global array
// holds raw element bit patterns (pass boundary of a two-pass transform) instead of canonical uint64.
template <class A, int LT, bool RAW_IN, bool RAW_OUT>
__global__ void __launch_bounds__(kThreads) ntt_tile_forward(uint64_t* __restrict__ data, size_t total, ModParams p,
                                                               const typename A::twid* __restrict__ tw) {
    __shared__ uint64_t lds[kLdsWords];
    using elem = typename A::elem;
    const uint32_t t = threadIdx.x;
    const size_t tile_base = (size_t)blockIdx.x * kTile;
    const uint32_t nmask = (1u << p.logn) - 1u;
    const uint32_t block_pos = (uint32_t)(tile_base & nmask);
    elem v[kRegs];

    auto run_round = [&](auto lo_tag, auto r_tag, bool first) {
        constexpr int LO = decltype(lo_tag)::value;
        constexpr int R = decltype(r_tag)::value;
        const uint32_t base = lane_base<LO, R>(t);
        if (first) {
#pragma unroll
            for (int k = 0; k < kRegs; ++k) {
                const size_t gi = tile_base + (base | reg_offset<LO, R>(k));
                const uint64_t raw = gi < total ? data[gi] : 0;
                v[k] = RAW_IN ? elem_from_bits<A>(raw) : A::load(raw, p);
            }
        } else {
#pragma unroll
            for (int k = 0; k < kRegs; ++k) v[k] = elem_from_bits<A>(lds[lds_slot(base | reg_offset<LO, R>(k))]);
        }
        forward_round<A, LO, R>(v, base, block_pos, nmask, p, tw);
#pragma unroll
        for (int k = 0; k < kRegs; ++k) lds[lds_slot(base | reg_offset<LO, R>(k))] = elem_bits<A>(v[k]);
        __syncthreads();
    };

    constexpr int REM = LT % 4;
    constexpr int FULL = LT / 4;
    if constexpr (FULL >= 1) run_round(std::integral_constant<int, LT - 4>{}, std::integral_constant<int, 4>{}, true);
    if constexpr (FULL >= 2) run_round(std::integral_constant<int, LT - 8>{}, std::integral_constant<int, 4>{}, false);
    if constexpr (FULL >= 3) run_round(std::integral_constant<int, LT - 12>{}, std::integral_constant<int, 4>{}, false);
    if constexpr (REM > 0) run_round(std::integral_constant<int, 0>{}, std::integral_constant<int, REM>{}, FULL == 0);

    // coalesced write-out: lane t stores tile indices t + 256 k
#pragma unroll
    for (int k = 0; k < kRegs; ++k) {
        const uint32_t idx = t + (uint32_t)k * kThreads;
        const size_t gi = tile_base + idx;
        const uint64_t bits = lds[lds_slot(idx)];
        if (gi < total) data[gi] = RAW_OUT ? bits : A::store_canonical(elem_from_bits<A>(bits), p);
    }
}

template <class A, int LT, bool RAW_IN, bool RAW_OUT>
__global__ void __launch_bounds__(kThreads) ntt_tile_inverse(uint64_t* __restrict__ data, size_t total, ModParams p,
                                                               const typename A::twid* __restrict__ tw, RoundConsts<A> cs) {
    __shared__ uint64_t lds[kLdsWords];
    using elem = typename A::elem;
    const uint32_t t = threadIdx.x;
    const size_t tile_base = (size_t)blockIdx.x * kTile;
    const uint32_t nmask = (1u << p.logn) - 1u;
    const uint32_t block_pos = (uint32_t)(tile_base & nmask);
    elem v[kRegs];

    // coalesced read-in to LDS (raw element bits)
#pragma unroll
    for (int k = 0; k < kRegs; ++k) {
        const uint32_t idx = t + (uint32_t)k * kThreads;
        const size_t gi = tile_base + idx;
        const uint64_t raw = gi < total ? data[gi] : 0;
        lds[lds_slot(idx)] = RAW_IN ? raw : elem_bits<A>(A::load(raw, p));
    }
    __syncthreads();

    auto run_round = [&](auto lo_tag, auto r_tag, bool last) {
        constexpr int LO = decltype(lo_tag)::value;
        constexpr int R = decltype(r_tag)::value;
        const uint32_t base = lane_base<LO, R>(t);
#pragma unroll
        for (int k = 0; k < kRegs; ++k) v[k] = elem_from_bits<A>(lds[lds_slot(base | reg_offset<LO, R>(k))]);
        inverse_round<A, LO, R>(v, base, block_pos, nmask, p, tw, cs);
        const bool final_values = last && (LT == p.logn);   // these are outputs of the n^-1-scaled stage
        if (!final_values) {
#pragma unroll
            for (int k = 0; k < kRegs; ++k) A::end_of_inverse_round(v[k], p);
        }
        if (last) {
#pragma unroll
            for (int k = 0; k < kRegs; ++k) {
                const size_t gi = tile_base + (base | reg_offset<LO, R>(k));
                if (gi < total) data[gi] = RAW_OUT ? elem_bits<A>(v[k]) : A::store_reduced(v[k], p);
            }
        } else {
#pragma unroll
            for (int k = 0; k < kRegs; ++k) lds[lds_slot(base | reg_offset<LO, R>(k))] = elem_bits<A>(v[k]);
            __syncthreads();
        }
    };

    constexpr int REM = LT % 4;
    constexpr int FULL = LT / 4;
    if constexpr (REM > 0) run_round(std::integral_constant<int, 0>{}, std::integral_constant<int, REM>{}, FULL == 0);
    if constexpr (FULL >= 1) run_round(std::integral_constant<int, REM>{}, std::integral_constant<int, 4>{}, FULL == 1);
    if constexpr (FULL >= 2) run_round(std::integral_constant<int, REM + 4>{}, std::integral_constant<int, 4>{}, FULL == 2);
    if constexpr (FULL >= 3) run_round(std::integral_constant<int, REM + 8>{}, std::integral_constant<int, 4>{}, FULL == 3);
}

// ---- strided round kernel (index bits [lo, lo+R), lo >= 12) ----------------------------------------
template <class A, int R, bool INVERSE, bool RAW_IN, bool RAW_OUT>
__global__ void __launch_bounds__(kThreads) ntt_strided_round(uint64_t* __restrict__ data, size_t total, int lo, ModParams p,
                                                                const typename A::twid* __restrict__ tw, RoundConsts<A> cs) {
    using elem = typename A::elem;
    constexpr int N = 1 << R;
    const size_t group = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (group >= (total >> R)) return;
    const size_t low = group & (((size_t)1 << lo) - 1);
    const size_t idx0 = ((group >> lo) << (lo + R)) | low;
    const uint32_t nmask = (1u << p.logn) - 1u;
    const uint32_t pos0 = (uint32_t)(idx0 & nmask);
    elem v[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const uint64_t raw = data[idx0 + ((size_t)k << lo)];
        v[k] = RAW_IN ? elem_from_bits<A>(raw) : A::load(raw, p);
    }
    if (!INVERSE) {
#pragma unroll
        for (int j = R - 1; j >= 0; --j) {
            const int b = lo + j;
            const int half = 1 << j;
            const uint32_t tw_base = (1u << (p.logn - 1 - b)) + (pos0 >> (b + 1));
#pragma unroll
            for (int u = 0; u < (1 << (R - 1 - j)); ++u) {
                const typename A::twid w = A::load_tw(tw, tw_base + u);
#pragma unroll
                for (int l = 0; l < half; ++l) {
                    const int kx = (u << (j + 1)) | l;
                    A::ct(v[kx], v[kx + half], w, p);
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int b = lo + j;
            const int half = 1 << j;
            const bool last_stage = (b == p.logn - 1);
            const uint32_t tw_base = (1u << (p.logn - 1 - b)) + (pos0 >> (b + 1));
#pragma unroll
            for (int u = 0; u < (1 << (R - 1 - j)); ++u) {
                const typename A::twid w = A::load_tw(tw, tw_base + u);
#pragma unroll
                for (int l = 0; l < half; ++l) {
                    const int kx = (u << (j + 1)) | l;
                    if (last_stage) A::gs_scaled(v[kx], v[kx + half], cs.w_last_scaled, cs.n_inv, p);
                    else A::gs(v[kx], v[kx + half], w, p);
                }
            }
        }
        if (lo + R != p.logn) {
#pragma unroll
            for (int k = 0; k < N; ++k) A::end_of_inverse_round(v[k], p);
        }
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
        uint64_t out;
        if (RAW_OUT) out = elem_bits<A>(v[k]);
        else out = INVERSE ? A::store_reduced(v[k], p) : A::store_canonical(v[k], p);
        data[idx0 + ((size_t)k << lo)] = out;
    }
}

// ---- pointwise product (ntt.cpp:106-119) ------------------------------------------------------------
static __global__ void __launch_bounds__(kThreads) pointwise_mul_kernel(uint64_t* __restrict__ out, const uint64_t* __restrict__ a,
                                                                   const uint64_t* __restrict__ b, size_t count, ModParams p) {
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < count; i += stride) out[i] = mulmod_barrett128(a[i], b[i], p);
}

}  // namespace lsr
