// Batched negacyclic NTT kernels for gfx950 (replaces SEAL's ntt_negacyclic_harvey /
// inverse_ntt_negacyclic_harvey behind cpp-core/src/ntt.cpp:84,99).
//
// Decomposition.  A length-n = 2^L transform is L butterfly stages; stage s pairs coefficients whose
// index differs in bit (L-1-s) and multiplies by twiddle[2^s + (index >> (L-s))].  Stages are grouped
// into radix-16 ROUNDS of four consecutive index bits that one lane holds in 16 registers:
//   * tile kernel   — one 256-lane workgroup owns 4096 contiguous residues (one n=4096 polynomial, or
//     4096/n smaller ones, or one 4096-block of a larger one) and runs every round on index bits
//     [0,12) with the tile staged in LDS between rounds (one pad word per 16 keeps the b64 accesses of
//     the radix-16 rounds off each other's banks; PMC: 9 % residual conflict cycles); global traffic is
//     fully coalesced on both ends and goes through buffer resources (SGPR base, one lane offset,
//     immediate register offsets, hardware clipping of a partial last tile).  The 15 twiddles a lane
//     needs for round r+1 are requested before the butterflies of round r start, so their L2 latency
//     hides behind 256 FP64 instructions.
//   * strided round kernel — the top 4 (n = 2^17: 5) index bits of n > 4096: 16 registers hold residues
//     n/16 apart, all 256 lanes of a workgroup walk consecutive addresses, no LDS, twiddles are the
//     first 15 table entries (scalar loads).
// The intermediate array between two kernels of one transform is private, so it is left in the
// arithmetic's raw element form (f64 bit patterns for ArithF64) — no conversion at pass boundaries.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>

#include "lsr_arith.hpp"
#include "lsr_sampler.hpp"

namespace lsr {

constexpr int kTileLog = 12;
constexpr uint32_t kTile = 1u << kTileLog;
constexpr int kThreads = 256;
constexpr int kRegs = 16;
constexpr int kRoundTwiddles = 15;
constexpr uint32_t kLdsWords = kTile + (kTile >> 4);   // one pad word per 16

template <class A>
struct RoundConsts {
    typename A::twid n_inv;          // n^-1
    typename A::twid w_last_scaled;  // inverse twiddle of the last GS stage times n^-1
};

// padded LDS slot of tile index idx.  For idx = lane part | constant part with disjoint bits the slot splits into
// lds_slot(lane) + lds_slot(constant), so every access is one lane address plus an immediate offset.
__host__ __device__ constexpr uint32_t lds_slot(uint32_t idx) { return idx + (idx >> 4); }

template <class A> __device__ __forceinline__ uint64_t elem_bits(typename A::elem v);
template <> __device__ __forceinline__ uint64_t elem_bits<ArithF64>(double v) { return (uint64_t)__double_as_longlong(v); }
template <> __device__ __forceinline__ uint64_t elem_bits<ArithU64>(uint64_t v) { return v; }
template <> __device__ __forceinline__ uint64_t elem_bits<ArithGold>(uint64_t v) { return v; }
template <class A> __device__ __forceinline__ typename A::elem elem_from_bits(uint64_t b);
template <> __device__ __forceinline__ double elem_from_bits<ArithF64>(uint64_t b) { return __longlong_as_double((long long)b); }
template <> __device__ __forceinline__ uint64_t elem_from_bits<ArithU64>(uint64_t b) { return b; }
template <> __device__ __forceinline__ uint64_t elem_from_bits<ArithGold>(uint64_t b) { return b; }

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Round schedule of the tile kernel for LT transformed bits: full radix-16 rounds from the top bit down,
// then one remainder round of LT%4 bits at the bottom.  The inverse runs the same rounds in reverse.
template <int LT, int I>
struct TileRound {
    static constexpr int kFull = LT / 4;
    static constexpr int kRem = LT % 4;
    static constexpr int kCount = kFull + (kRem ? 1 : 0);
    static constexpr int R = I < kFull ? 4 : kRem;
    static constexpr int LO = I < kFull ? LT - 4 * (I + 1) : 0;
};

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

// ---- twiddles of one round into registers -------------------------------------------------------------
// Order: group, then stage in execution order (forward: high bit first; inverse: low bit first), then u.
// SKIP_TOP: the stage of bit LO+R-1 takes its multipliers from RoundConsts (last inverse stage).
template <class A, int LO, int R, bool INVERSE, bool SKIP_TOP>
__device__ __forceinline__ void load_round_twiddles(typename A::twid (&w)[kRoundTwiddles], uint32_t base_idx, uint32_t block_pos,
                                                    uint32_t nmask, int logn, rsrc_t tw) {
    constexpr int G = 1 << (4 - R);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const uint32_t pos0 = (block_pos + (base_idx | ((uint32_t)g << (8 + R)))) & nmask;
        static_for<0, R>([&](auto sc) {
            constexpr int step = decltype(sc)::value;
            constexpr int b = INVERSE ? LO + step : LO + R - 1 - step;
            if constexpr (!(SKIP_TOP && b == LO + R - 1)) {
                constexpr int count = 1 << (LO + R - 1 - b);
                // slots before this run: earlier groups, then earlier stages of this group
                constexpr int per_group = (1 << R) - 1 - (SKIP_TOP ? 1 : 0);
                constexpr int before = INVERSE ? ((1 << R) - (1 << (R - step))) : ((1 << step) - 1) - (SKIP_TOP && step > 0 ? 1 : 0);
                const uint32_t tw_base = (1u << (logn - 1 - b)) + (pos0 >> (b + 1));
                A::template load_tw_run<count>(tw, tw_base, &w[g * per_group + before]);
            }
        });
    }
}

// ---- butterflies of one round, forward (Cooley–Tukey, high bit first) -----------------------------
// TOP: the round holds the transform's first stages (group index 0 above it).  Flavours with A::kUnitTopTwiddles (cyclic tables:
// entry m + 0 of every stage is omega^0 = 1, lsr_host_math.cpp build_cyclic_twiddles) then skip the product of the u = 0 butterflies —
// 15 of the 32 butterflies of a radix-16 top round, 15.6 % of a 4096-point transform's products.
template <class A, int LO, int R, bool TOP = false>
__device__ __forceinline__ void forward_round(typename A::elem (&v)[kRegs], const typename A::twid (&w)[kRoundTwiddles], const ModParams& p) {
    constexpr int G = 1 << (4 - R);
    int slot = 0;
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int j = R - 1; j >= 0; --j) {          // register bit j <-> index bit LO + j
            const int half = 1 << j;
#pragma unroll
            for (int u = 0; u < (1 << (R - 1 - j)); ++u) {
                const typename A::twid tw = w[slot++];
#pragma unroll
                for (int l = 0; l < half; ++l) {
                    const int kx = (g << R) | (u << (j + 1)) | l;
                    if constexpr (TOP && A::kUnitTopTwiddles) {
                        if (u == 0) A::ct_unit(v[kx], v[kx + half], p);
                        else A::ct(v[kx], v[kx + half], tw, p);
                    } else {
                        A::ct(v[kx], v[kx + half], tw, p);
                    }
                }
            }
        }
    }
}

// ---- butterflies of one round, inverse (Gentleman–Sande, low bit first) ---------------------------
// FINAL: bit LO+R-1 is the transform's last stage (n^-1 folded in, SEAL transform_from_rev's scalar path)
template <class A, int LO, int R, bool FINAL>
__device__ __forceinline__ void inverse_round(typename A::elem (&v)[kRegs], const typename A::twid (&w)[kRoundTwiddles], const ModParams& p,
                                              const RoundConsts<A>& cs) {
    constexpr int G = 1 << (4 - R);
    int slot = 0;
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int half = 1 << j;
            if (FINAL && j == R - 1) {
#pragma unroll
                for (int l = 0; l < half; ++l) A::gs_scaled(v[(g << R) | l], v[((g << R) | l) + half], cs.w_last_scaled, cs.n_inv, p);
                continue;
            }
#pragma unroll
            for (int u = 0; u < (1 << (R - 1 - j)); ++u) {
                const typename A::twid tw = w[slot++];
#pragma unroll
                for (int l = 0; l < half; ++l) {
                    const int kx = (g << R) | (u << (j + 1)) | l;
                    if constexpr (FINAL && A::kUnitTopTwiddles) {      // the last round = the forward transform's top round: u = 0 is omega^0
                        if (u == 0) A::gs_unit(v[kx], v[kx + half], p);
                        else A::gs(v[kx], v[kx + half], tw, p);
                    } else {
                        A::gs(v[kx], v[kx + half], tw, p);
                    }
                }
            }
        }
    }
}

// ---- tile kernel -----------------------------------------------------------------------------------
// LT = number of low index bits this kernel transforms (min(L,12)).  RAW_IN / RAW_OUT: the global array
// holds raw element bit patterns (pass boundary of a two-pass transform) instead of canonical uint64.
// Global and twiddle accesses go through buffer resources: the tile's base sits in SGPRs, each lane carries
// one byte offset per mapping, register offsets are immediates, and a partial last tile is clipped by the
// resource's size (loads return 0, stores are dropped).
// src (optional): read the operands from there instead of `data` (same layout) — an out-of-place first pass.
// vblock: the tile this workgroup transforms (the hardware block index when the pass is a kernel of its own)
// Elementwise work fused into the read-in of a forward tile pass (the prover's quotient pipeline, lsr_prover.hip; Goldilocks only):
//   MODE 1: every operand is first multiplied by the word of `x1` at the same index (standard form) — the coset product a b;
//   MODE 2: the operands are the c_k of R1CS constraints and x1 = a, x2 = b: a_k b_k == c_k is tested on the way in, a failing
//           constraint marks its instance in `bad` (is_satisfied, r1cs.rs:148-172).  The transform itself is unchanged.
//   MODE 3: MODE 1 on the way in, and on the way OUT the finish of the quotient pipeline (lsr_prover.hip step 6): the transform's
//           output z never reaches memory; Q[j] = (2m)^-1 c_hat[p] - psi^-j (2m)^-1 z[p], p = bitrev(j), is formed where the last round
//           leaves z in registers (c_hat and the untwist table read at the same indices), scattered to its natural-order slot of the
//           LDS tile and streamed out coalesced, with the per-instance highest non-zero index (one atomicMax per wavefront).
struct FuseIn {
    const uint64_t* x1 = nullptr;
    const uint64_t* x2 = nullptr;
    uint32_t* bad = nullptr;
    // MODE 3
    const uint64_t* chat = nullptr;       // [instances][m], same tiling as the data
    const uint64_t* untwist = nullptr;    // [m], Montgomery form
    uint64_t half_m_inv = 0;              // (2m)^-1, Montgomery form
    uint64_t* quotient = nullptr;         // [instances][m], natural order
    uint32_t* top = nullptr;              // [instances]: 1 + highest non-zero index
};
template <class A, int LT, bool RAW_IN, bool RAW_OUT, int MODE = 0>
__device__ __forceinline__ void tile_forward_body(uint64_t* __restrict__ data, size_t total, const ModParams& p,
                                                  const typename A::twid* __restrict__ tw, const uint64_t* __restrict__ src, uint32_t vblock,
                                                  const FuseIn& fuse = FuseIn{}) {
    __shared__ uint64_t lds[kLdsWords];
    using elem = typename A::elem;
    using twid = typename A::twid;
    constexpr int NR = TileRound<LT, 0>::kCount;
    const uint32_t t = threadIdx.x;
    const size_t tile_base = (size_t)vblock * kTile;
    const uint32_t nmask = (1u << p.logn) - 1u;
    const uint32_t block_pos = (uint32_t)(tile_base & nmask);
    const size_t left = total - tile_base;
    const uint32_t tile_bytes = left >= kTile ? kTile * 8u : (uint32_t)left * 8u;
    const rsrc_t tile = make_rsrc(data + tile_base, tile_bytes);
    const rsrc_t from = make_rsrc(const_cast<uint64_t*>(src ? src : data) + tile_base, tile_bytes);
    const rsrc_t table = make_rsrc(tw, (uint32_t)sizeof(twid) << p.logn);
    elem v[kRegs];
    twid w[2][kRoundTwiddles];

    {   // first round: operands straight from global memory in the round's own mapping
        constexpr int LO = TileRound<LT, 0>::LO, R = TileRound<LT, 0>::R;
        const uint32_t base = lane_base<LO, R>(t);
        uint64_t raw[kRegs];
#pragma unroll
        for (int k = 0; k < kRegs; ++k) raw[k] = buf_load64<RAW_OUT ? 0 : kAuxStream>(from, base * 8u, reg_offset<LO, R>(k) * 8u);
        load_round_twiddles<A, LO, R, false, false>(w[0], base, block_pos, nmask, p.logn, table);
        if constexpr (MODE == 0) {
#pragma unroll
            for (int k = 0; k < kRegs; ++k) v[k] = RAW_IN ? elem_from_bits<A>(raw[k]) : A::load(raw[k], p);
        } else {
            static_assert(std::is_same_v<A, ArithGold> && !RAW_IN, "fused elementwise work: first pass of a Goldilocks transform");
            const rsrc_t r1 = make_rsrc(fuse.x1 + tile_base, tile_bytes);
            if constexpr (MODE == 1 || MODE == 3) {
                uint64_t o1[kRegs];
#pragma unroll
                for (int k = 0; k < kRegs; ++k) o1[k] = buf_load64(r1, base * 8u, reg_offset<LO, R>(k) * 8u);
#pragma unroll
                for (int k = 0; k < kRegs; ++k) v[k] = gold_mul(A::load(raw[k], p), A::load(o1[k], p));
            } else {
                // a and b come in two halves: three operand sets live at once cost 138 VGPRs and a wave of occupancy
                const rsrc_t r2 = make_rsrc(fuse.x2 + tile_base, tile_bytes);
#pragma unroll
                for (int k = 0; k < kRegs; ++k) v[k] = A::load(raw[k], p);
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    constexpr int H = kRegs / 2;
                    uint64_t o1[H], o2[H];
#pragma unroll
                    for (int k = 0; k < H; ++k) {
                        o1[k] = buf_load64(r1, base * 8u, reg_offset<LO, R>(half * H + k) * 8u);
                        o2[k] = buf_load64(r2, base * 8u, reg_offset<LO, R>(half * H + k) * 8u);
                    }
#pragma unroll
                    for (int k = 0; k < H; ++k) {
                        const uint32_t at = base + reg_offset<LO, R>(half * H + k);     // index within the tile
                        if (at * 8u < tile_bytes && gold_mul(A::load(o1[k], p), A::load(o2[k], p)) != v[half * H + k])
                            atomicOr(&fuse.bad[(tile_base + at) >> p.logn], 1u);         // only ever taken for an unsatisfied instance
                    }
                }
            }
        }
    }

    static_for<0, NR>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
        constexpr int LO = TileRound<LT, I>::LO, R = TileRound<LT, I>::R;
        const uint32_t base = lane_base<LO, R>(t);
        if constexpr (I + 1 < NR) {   // request the next round's twiddles before this round's arithmetic
            constexpr int LO1 = TileRound<LT, I + 1>::LO, R1 = TileRound<LT, I + 1>::R;
            load_round_twiddles<A, LO1, R1, false, false>(w[(I + 1) & 1], lane_base<LO1, R1>(t), block_pos, nmask, p.logn, table);
        }
        forward_round<A, LO, R, I == 0 && !RAW_IN && (LO + R == LT)>(v, w[I & 1], p);   // !RAW_IN: this pass is the whole transform (LT = log n)
        uint64_t* const row = lds + lds_slot(base);
#pragma unroll
        for (int k = 0; k < kRegs; ++k) row[lds_slot(reg_offset<LO, R>(k))] = elem_bits<A>(v[k]);
        __syncthreads();
        if constexpr (I + 1 < NR) {
            constexpr int LO1 = TileRound<LT, I + 1>::LO, R1 = TileRound<LT, I + 1>::R;
            const uint64_t* const row1 = lds + lds_slot(lane_base<LO1, R1>(t));
#pragma unroll
            for (int k = 0; k < kRegs; ++k) v[k] = elem_from_bits<A>(row1[lds_slot(reg_offset<LO1, R1>(k))]);
        }
    });

    // coalesced write-out: lane t stores tile indices t + 256 k
    const uint64_t* const col = lds + lds_slot(t);
    if constexpr (MODE == 3) {
        // The finish, in the coalesced mapping (the last round's own mapping holds 16 consecutive words per lane: c_hat and the
        // untwist table would be read with a 128-byte lane stride, measured 100 us per 2^24 words).  Lane t takes z at tile indices
        // p = t + 256 k (bit-reversed order), forms Q there, and a second trip through the tile puts Q at its natural index.
        // Slot map of that trip: nat + (nat >> S), S = log m - 6 — a wavefront's 64 natural indices are 2^S apart (bit reversal
        // of consecutive p), the pad spreads them over all banks; the coalesced read-back stays conflict-free.
        constexpr int S = LT >= 10 ? LT - 6 : 4;
        const rsrc_t chat = make_rsrc(fuse.chat + tile_base, tile_bytes);
        const rsrc_t untw = make_rsrc(fuse.untwist, 8u << p.logn);
        const rsrc_t out = make_rsrc(fuse.quotient + tile_base, tile_bytes);
        uint64_t h[kRegs];
        {
            uint64_t c[kRegs], uw[kRegs];
#pragma unroll
            for (int k = 0; k < kRegs; ++k) {
                c[k] = buf_load64<kAuxStream>(chat, t * 8u, (uint32_t)k * kThreads * 8u);
                uw[k] = buf_load64(untw, ((block_pos + t + (uint32_t)k * kThreads) & nmask) * 8u, 0);
            }
#pragma unroll
            for (int k = 0; k < kRegs; ++k)   // z is a lazy 64-bit representative (ArithGold::ct): gold_mul_mont takes any
                h[k] = gold_sub(gold_mul_mont(A::load(c[k], p), fuse.half_m_inv), gold_mul_mont(col[lds_slot((uint32_t)k * kThreads)], uw[k]));
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kRegs; ++k) {
            const uint32_t at = t + (uint32_t)k * kThreads;
            const uint32_t nat = (at & ~nmask) | (__brev(at & nmask) >> (32 - LT));       // LT = log m (single-pass transform)
            lds[nat + (nat >> S)] = h[k];
        }
        __syncthreads();
        constexpr int kGroup = LT >= 8 ? 1 << (LT - 8) : 1;     // registers per instance
#pragma unroll
        for (int g = 0; g < kRegs / kGroup; ++g) {
            uint32_t best = 0;                                   // wave-uniform for log m >= 6
#pragma unroll
            for (int kk = 0; kk < kGroup; ++kk) {
                const int k = g * kGroup + kk;
                const uint32_t at = t + (uint32_t)k * kThreads;
                const bool live = at * 8u < tile_bytes;
                const uint64_t q = lds[at + (at >> S)];
                buf_store64<kAuxStream>(out, t * 8u, (uint32_t)k * kThreads * 8u, q);
                if constexpr (LT >= 6) {   // a wavefront's 64 consecutive words belong to one instance: one atomic per wavefront and instance
                    const uint64_t nz = __ballot(live && q != 0);
                    if (nz) best = ((at & ~63u) & nmask) + 64u - (uint32_t)__clzll(nz);
                } else if (live && q != 0) {
                    atomicMax(&fuse.top[(tile_base + at) >> p.logn], (at & nmask) + 1u);
                }
            }
            if constexpr (LT >= 6) {
                const uint32_t first = t + (uint32_t)(g * kGroup) * kThreads;
                if (best && (t & 63u) == 0) atomicMax(&fuse.top[(tile_base + first) >> p.logn], best);
            }
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < kRegs; ++k) {
        const uint64_t bits = col[lds_slot((uint32_t)k * kThreads)];
        buf_store64<RAW_OUT ? 0 : kAuxStream>(tile, t * 8u, (uint32_t)k * kThreads * 8u, RAW_OUT ? bits : A::store_canonical(elem_from_bits<A>(bits), p));
    }
}
template <class A, int LT, bool RAW_IN, bool RAW_OUT>
__global__ void __launch_bounds__(kThreads) ntt_tile_forward(uint64_t* __restrict__ data, size_t total, ModParams p,
                                                               const typename A::twid* __restrict__ tw, const uint64_t* __restrict__ src = nullptr) {
    tile_forward_body<A, LT, RAW_IN, RAW_OUT>(data, total, p, tw, src, blockIdx.x);
}
template <class A, int LT, int MODE>
__global__ void __launch_bounds__(kThreads) ntt_tile_forward_fused(uint64_t* __restrict__ data, size_t total, ModParams p,
                                                                     const typename A::twid* __restrict__ tw, const uint64_t* __restrict__ src, FuseIn fuse) {
    tile_forward_body<A, LT, false, false, MODE>(data, total, p, tw, src, blockIdx.x, fuse);
}

// `add` (optional, only when !RAW_OUT): canonical residues added to the outputs on the final store — the fused
// discrete-Gaussian blinding add of the commitment (u = INTT(...) + e1).
// PRE: every input word is first multiplied by pre[its index within the polynomial] (canonical residues) — a diagonal
// operator fused into the read-in (the coset twist of the prover's quotient pipeline, lsr_prover.hip).
template <class A, int LT, bool RAW_IN, bool RAW_OUT, bool PRE = false>
__device__ __forceinline__ void tile_inverse_body(uint64_t* __restrict__ data, size_t total, const ModParams& p,
                                                  const typename A::twid* __restrict__ tw, const RoundConsts<A>& cs,
                                                  const uint64_t* __restrict__ add, const uint64_t* __restrict__ pre, uint32_t vblock) {
    __shared__ uint64_t lds[kLdsWords];
    using elem = typename A::elem;
    using twid = typename A::twid;
    constexpr int NR = TileRound<LT, 0>::kCount;
    const uint32_t t = threadIdx.x;
    const size_t tile_base = (size_t)vblock * kTile;
    const uint32_t nmask = (1u << p.logn) - 1u;
    const uint32_t block_pos = (uint32_t)(tile_base & nmask);
    const size_t left = total - tile_base;
    const uint32_t tile_bytes = left >= kTile ? kTile * 8u : (uint32_t)left * 8u;
    const rsrc_t tile = make_rsrc(data + tile_base, tile_bytes);
    const rsrc_t table = make_rsrc(tw, (uint32_t)sizeof(twid) << p.logn);
    elem v[kRegs];
    twid w[2][kRoundTwiddles];

    {   // coalesced read-in to LDS (raw element bits); the first round's twiddles ride along
        uint64_t raw[kRegs];
        // single-pass inverse transforms (n <= 4096): a fresh workgroup puts its loads in flight ahead of its neighbours' arithmetic
        // (-4 % at n = 4096; the same in the forward kernel or in the passes of a two-pass transform loses 2-3 %: profiles/r02b_tile_prio.txt)
        constexpr bool kLoadPrio = !RAW_IN && !RAW_OUT;
        if constexpr (kLoadPrio) __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int k = 0; k < kRegs; ++k) raw[k] = buf_load64<RAW_OUT ? 0 : kAuxStream>(tile, t * 8u, (uint32_t)k * kThreads * 8u);
        if constexpr (kLoadPrio) __builtin_amdgcn_s_setprio(0);
        constexpr int J = NR - 1;
        constexpr int LO = TileRound<LT, J>::LO, R = TileRound<LT, J>::R;
        load_round_twiddles<A, LO, R, true, (NR == 1) && !RAW_OUT>(w[0], lane_base<LO, R>(t), block_pos, nmask, p.logn, table);
        uint64_t* const col = lds + lds_slot(t);
        if constexpr (PRE) {
            static_assert(!RAW_IN, "the fused diagonal multiply belongs to the first pass");
            const rsrc_t diag = make_rsrc(pre, 8u << p.logn);
            uint64_t d[kRegs];
#pragma unroll
            for (int k = 0; k < kRegs; ++k) d[k] = buf_load64(diag, ((block_pos + t + (uint32_t)k * kThreads) & nmask) * 8u, 0);
#pragma unroll
            for (int k = 0; k < kRegs; ++k) col[lds_slot((uint32_t)k * kThreads)] = elem_bits<A>(A::pre_mul(A::load(raw[k], p), d[k], p));
        } else {
#pragma unroll
            for (int k = 0; k < kRegs; ++k) col[lds_slot((uint32_t)k * kThreads)] = RAW_IN ? raw[k] : elem_bits<A>(A::load(raw[k], p));
        }
    }
    __syncthreads();

    static_for<0, NR>([&](auto ic) {
        constexpr int I = decltype(ic)::value;          // I-th inverse round = forward round NR-1-I
        constexpr int J = NR - 1 - I;
        constexpr int LO = TileRound<LT, J>::LO, R = TileRound<LT, J>::R;
        constexpr bool kLast = (I == NR - 1);
        constexpr bool kFinal = kLast && !RAW_OUT;      // outputs of the n^-1-scaled stage (LT == log n)
        const uint32_t base = lane_base<LO, R>(t);
        uint64_t* const row = lds + lds_slot(base);
#pragma unroll
        for (int k = 0; k < kRegs; ++k) v[k] = elem_from_bits<A>(row[lds_slot(reg_offset<LO, R>(k))]);
        if constexpr (!kLast) {
            constexpr int J1 = J - 1;
            constexpr int LO1 = TileRound<LT, J1>::LO, R1 = TileRound<LT, J1>::R;
            constexpr bool kNextFinal = (I + 1 == NR - 1) && !RAW_OUT;
            load_round_twiddles<A, LO1, R1, true, kNextFinal>(w[(I + 1) & 1], lane_base<LO1, R1>(t), block_pos, nmask, p.logn, table);
        }
        uint64_t blind[(kLast && !RAW_OUT) ? kRegs : 1];
        if constexpr (kLast && !RAW_OUT) {   // request the blinding residues before the last round's arithmetic
            if (add != nullptr) {
                const rsrc_t extra = make_rsrc(add + tile_base, tile_bytes);
#pragma unroll
                for (int k = 0; k < kRegs; ++k) blind[k] = buf_load64(extra, base * 8u, reg_offset<LO, R>(k) * 8u);
            }
        }
        inverse_round<A, LO, R, kFinal>(v, w[I & 1], p, cs);
        if constexpr (!kFinal) {
            // inner rounds re-centre only the outputs that can exceed the next round's 2 q input bound (lsr_arith.hpp needs_recentre);
            // the last round of a first pass (RAW_OUT) re-centres everything: the strided round that follows runs up to five stages
            constexpr bool kAll = kLast || !A::kPartialRecentre;
#pragma unroll
            for (int k = 0; k < kRegs; ++k)
                if (kAll || A::template needs_recentre<R>(k & ((1 << R) - 1))) A::end_of_inverse_round(v[k], p);
        }
        if constexpr (kLast) {
            if (!RAW_OUT && add != nullptr) {
#pragma unroll
                for (int k = 0; k < kRegs; ++k)
                    buf_store64<kAuxStream>(tile, base * 8u, reg_offset<LO, R>(k) * 8u, A::store_reduced_plus(v[k], blind[k], p));
            } else {
#pragma unroll
                for (int k = 0; k < kRegs; ++k)
                    buf_store64<RAW_OUT ? 0 : kAuxStream>(tile, base * 8u, reg_offset<LO, R>(k) * 8u, RAW_OUT ? elem_bits<A>(v[k]) : A::store_reduced(v[k], p));
            }
        } else {
#pragma unroll
            for (int k = 0; k < kRegs; ++k) row[lds_slot(reg_offset<LO, R>(k))] = elem_bits<A>(v[k]);
            __syncthreads();
        }
    });
}
template <class A, int LT, bool RAW_IN, bool RAW_OUT, bool PRE = false>
__global__ void __launch_bounds__(kThreads) ntt_tile_inverse(uint64_t* __restrict__ data, size_t total, ModParams p,
                                                               const typename A::twid* __restrict__ tw, RoundConsts<A> cs,
                                                               const uint64_t* __restrict__ add, const uint64_t* __restrict__ pre = nullptr) {
    tile_inverse_body<A, LT, RAW_IN, RAW_OUT, PRE>(data, total, p, tw, cs, add, pre, blockIdx.x);
}

// ---- strided round kernel (the TOP R index bits: lo + R == log n) ------------------------------------
// Its butterfly groups are indexed by the polynomial only, so stage j (register bit j) uses the table entries
// 2^(R-1-j) + u for every lane of every polynomial: compile-time indices, scalar loads, no VGPRs for twiddles.
// ADD: canonical residues `add` are added to the outputs in the final store (inverse, last pass only).
// Forward rounds (ADD is then false): a non-null `add` is the array to READ the operands from (out-of-place first pass).
// SM (sampling mode; 1 and 2 with ADD): the blinding residues are not read from `add` but SAMPLED in the passes around the middle
// stage — the CDT Gaussian of lsr_sampler.hpp, object = polynomial (key bs.keys[4 (poly / components)], stream index
// poly % components).  A workgroup owns 256 columns x 2^R rows of one polynomial = 2^R * 32 ChaCha blocks of 8 consecutive
// coefficients; the samples change hands through LDS.  The operand loads are issued first and are in flight under the cipher
// work, so the integer-bound sampler and the memory-bound round share a pass, and e1 never exists in memory as residues.
//   SM = 1: the last inverse round samples all 2^R rows itself (2^R / 8 blocks per lane);
//   SM = 3 + SM = 2: the FORWARD round of the same chunk (its VALU is as idle as the inverse round's) samples rows [0, 2^R / 2) and
//            leaves them in bs.side as int8, one byte per row packed per column ([polynomial][column][2^R / 16] words: 1/16 of a
//            polynomial pass); the inverse round reads that and samples rows [2^R / 2, 2^R) — the cipher work split over two passes.
//            Needs table entries <= 127 (magnitudes fit a byte).
template <class A, int R, bool INVERSE, bool RAW_IN, bool RAW_OUT, bool ADD, int SM>
__device__ __forceinline__ void strided_round_body(uint64_t* __restrict__ data, size_t total, int lo, ModParams p,
                                                   const typename A::twid* __restrict__ tw, RoundConsts<A> cs,
                                                   const uint64_t* __restrict__ add, const BlindSampler& bs, uint32_t vblock = blockIdx.x,
                                                   uint32_t vthread = threadIdx.x) {
    // vblock / vthread: the 256-lane group this call plays (a kernel that hosts the round as ONE of its roles, lsr_commit_fused.hpp
    // mlwe_mixed, passes its own numbering; sampling modes use the real indices and LDS and are not available there)
    static_assert(!ADD || (INVERSE && !RAW_OUT), "the fused add belongs to the last inverse pass");
    static_assert(SM == 0 || R >= 4, "2^R / 16 blocks per lane and half");
    static_assert((SM != 1 && SM != 2) || ADD, "sampling replaces the read of the blinding residues");
    static_assert(SM != 3 || (!INVERSE && !ADD), "the forward round only prepares samples");
    using elem = typename A::elem;
    constexpr int N = 1 << R;
    const size_t group = (size_t)vblock * kThreads + vthread;
    if (group >= (total >> R)) return;          // whole workgroups: total >> R is a multiple of kThreads (2^lo >= 512)
    const size_t low = group & (((size_t)1 << lo) - 1);
    const size_t idx0 = ((group >> lo) << (lo + R)) | low;
    elem v[N];
    uint64_t extra[ADD ? N : 1];
    const uint64_t* const from = (!INVERSE && add != nullptr) ? add : data;
    // streaming policy: the last pass of an inverse transform, and operands read out of place (the caller's array is read once)
    const bool stream_in = (LSR_NT_LAST_PASS && INVERSE && !RAW_OUT) || (LSR_NT_COMMIT_INPUTS && !INVERSE && add != nullptr);
    if (stream_in) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const uint64_t raw = __builtin_nontemporal_load(from + idx0 + ((size_t)k << lo));
            v[k] = RAW_IN ? elem_from_bits<A>(raw) : A::load(raw, p);
        }
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const uint64_t raw = from[idx0 + ((size_t)k << lo)];
            v[k] = RAW_IN ? elem_from_bits<A>(raw) : A::load(raw, p);
        }
    }
    if constexpr (SM != 0) {
        extern __shared__ uint64_t lds_words[];
        uint64_t* const cdf63 = lds_words;
        uint64_t* const tile_words = lds_words + ((bs.entries + 1u) & ~1u);
        const bool in_lanes = lane_table_steps(bs.entries) != 0u;            // the table in the wavefront's lanes (cdt_search), else in LDS
        if (!in_lanes) {
            for (uint32_t i = threadIdx.x; i < bs.entries; i += kThreads) cdf63[i] = bs.cdf[i] >> 1;
            __syncthreads();
        }
        const LaneTable tab = in_lanes ? lane_table_load(bs.cdf, bs.entries) : LaneTable{0u, 0u};
        const size_t g0 = (size_t)blockIdx.x * kThreads;               // workgroup-uniform: polynomial and first column
        const uint32_t poly = (uint32_t)(g0 >> lo), low0 = (uint32_t)(g0 & (((size_t)1 << lo) - 1));
        const uint64_t* const key = bs.keys + 4 * (size_t)(poly / bs.components);
        constexpr int kRows = SM == 1 ? N : N / 2;                      // rows sampled by this pass
        constexpr int kRow0 = SM == 2 ? N / 2 : 0;                      // first of them
        constexpr int kSideWords = N / 16;                              // packed words per column in bs.side
#pragma unroll 1
        for (int h = 0; h < kRows / 8; ++h) {
            const uint32_t b = (uint32_t)h * kThreads + threadIdx.x, row = b >> 5, cb = b & 31u;
            uint64_t w[8], u[8];
            stream_block(key, bs.domain, poly % bs.components, ((((kRow0 + row) << lo) + low0) >> 3) + cb, w);
#pragma unroll
            for (int i = 0; i < 8; ++i) u[i] = w[i] >> 1;
            uint32_t magnitude[8];
            cdt_magnitudes(tab, cdf63, bs.entries, u, magnitude);
            if constexpr (SM == 3) {                                    // int8 tile [row][256 columns]: eight samples = one LDS word
                uint64_t packed = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint32_t m = magnitude[i], sign = (uint32_t)(w[i] & 1ull);
                    packed |= (uint64_t)((sign ? 0u - m : m) & 0xFFu) << (8 * i);
                }
                tile_words[row * (kThreads / 8) + cb] = packed;
            } else {                                                    // int32 tile [row][256 columns]
                int32_t* const tile = reinterpret_cast<int32_t*>(tile_words);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int32_t m = (int32_t)magnitude[i], sign = (int32_t)(w[i] & 1ull);
                    tile[row * kThreads + cb * 8 + i] = sign ? -m : m;
                }
            }
        }
        __syncthreads();
        if constexpr (SM == 3) {                                        // column j: one byte per row, packed row-major into words
            const uint8_t* const bytes = reinterpret_cast<const uint8_t*>(tile_words);
            uint64_t* const dst = bs.side + ((size_t)poly << lo) * kSideWords + (size_t)(low0 + threadIdx.x) * kSideWords;
#pragma unroll
            for (int wd = 0; wd < kSideWords; ++wd) {
                uint64_t packed = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) packed |= (uint64_t)bytes[(wd * 8 + i) * kThreads + threadIdx.x] << (8 * i);
                dst[wd] = packed;
            }
        } else {
            const int32_t* const tile = reinterpret_cast<const int32_t*>(tile_words);
            if constexpr (SM == 2) {
                const uint64_t* const src = bs.side + ((size_t)poly << lo) * kSideWords + (size_t)(low0 + threadIdx.x) * kSideWords;
#pragma unroll
                for (int wd = 0; wd < kSideWords; ++wd) {
                    const uint64_t packed = __builtin_nontemporal_load(src + wd);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int32_t e = (int32_t)(int8_t)(packed >> (8 * i));
                        extra[wd * 8 + i] = e < 0 ? p.q - (uint64_t)(-e) : (uint64_t)e;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < kRows; ++k) {
                const int32_t e = tile[k * kThreads + threadIdx.x];
                extra[kRow0 + k] = e < 0 ? p.q - (uint64_t)(-e) : (uint64_t)e;
            }
        }
    } else if constexpr (ADD) {   // the blinding residues travel with the operands, not behind the arithmetic
#pragma unroll
        for (int k = 0; k < N; ++k) extra[k] = LSR_NT_COMMIT_INPUTS ? __builtin_nontemporal_load(add + idx0 + ((size_t)k << lo)) : add[idx0 + ((size_t)k << lo)];
    }
    if (!INVERSE) {
#pragma unroll
        for (int j = R - 1; j >= 0; --j) {
            const int half = 1 << j;
#pragma unroll
            for (int u = 0; u < (1 << (R - 1 - j)); ++u) {
                const typename A::twid w = tw[(1 << (R - 1 - j)) + u];
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
            const int half = 1 << j;
            if (!RAW_OUT && j == R - 1) {   // the transform's last stage (bit log n - 1)
#pragma unroll
                for (int l = 0; l < half; ++l) A::gs_scaled(v[l], v[l + half], cs.w_last_scaled, cs.n_inv, p);
                continue;
            }
#pragma unroll
            for (int u = 0; u < (1 << (R - 1 - j)); ++u) {
                const typename A::twid w = tw[(1 << (R - 1 - j)) + u];
#pragma unroll
                for (int l = 0; l < half; ++l) {
                    const int kx = (u << (j + 1)) | l;
                    A::gs(v[kx], v[kx + half], w, p);
                }
            }
        }
        if (RAW_OUT) {
#pragma unroll
            for (int k = 0; k < N; ++k) A::end_of_inverse_round(v[k], p);
        }
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const size_t gi = idx0 + ((size_t)k << lo);
        uint64_t out;
        if (RAW_OUT) out = elem_bits<A>(v[k]);
        else if constexpr (ADD) out = A::store_reduced_plus(v[k], extra[k], p);
        else out = INVERSE ? A::store_reduced(v[k], p) : A::store_canonical(v[k], p);
        if constexpr (LSR_NT_LAST_PASS && INVERSE && !RAW_OUT) __builtin_nontemporal_store(out, data + gi);
        else data[gi] = out;
    }
}

template <class A, int R, bool INVERSE, bool RAW_IN, bool RAW_OUT, bool ADD>
__global__ void __launch_bounds__(kThreads) ntt_strided_round(uint64_t* __restrict__ data, size_t total, int lo, ModParams p,
                                                                const typename A::twid* __restrict__ tw, RoundConsts<A> cs,
                                                                const uint64_t* __restrict__ add) {
    strided_round_body<A, R, INVERSE, RAW_IN, RAW_OUT, ADD, 0>(data, total, lo, p, tw, cs, add, BlindSampler{});
}

// last pass of an inverse transform with the blinding residues sampled in place (dynamic LDS: table + sample tile); HALF: the first
// half of the rows comes from bs.side, written by ntt_strided_round_sampling of the same chunk
template <class A, int R, bool RAW_IN, bool HALF>
__global__ void __launch_bounds__(kThreads) ntt_strided_round_sampled(uint64_t* __restrict__ data, size_t total, int lo, ModParams p,
                                                                        const typename A::twid* __restrict__ tw, RoundConsts<A> cs, BlindSampler bs) {
    strided_round_body<A, R, true, RAW_IN, false, true, HALF ? 2 : 1>(data, total, lo, p, tw, cs, nullptr, bs);
}
// first (out-of-place, raw-out) round of a forward transform that also samples the first half of the rows of the blinding
// polynomial of the same index into bs.side
template <class A, int R>
__global__ void __launch_bounds__(kThreads) ntt_strided_round_sampling(uint64_t* __restrict__ data, size_t total, int lo, ModParams p,
                                                                         const typename A::twid* __restrict__ tw, RoundConsts<A> cs,
                                                                         const uint64_t* __restrict__ src, BlindSampler bs) {
    strided_round_body<A, R, false, false, true, false, 3>(data, total, lo, p, tw, cs, src, bs);
}

// ---- pointwise product (ntt.cpp:106-119) ------------------------------------------------------------
static __global__ void __launch_bounds__(kThreads) pointwise_mul_kernel(uint64_t* __restrict__ out, const uint64_t* __restrict__ a,
                                                                          const uint64_t* __restrict__ b, size_t count, ModParams p) {
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < count; i += stride) out[i] = mulmod_barrett128(a[i], b[i], p);
}

static __global__ void __launch_bounds__(kThreads) pointwise_mul_gold_kernel(uint64_t* __restrict__ out, const uint64_t* __restrict__ a,
                                                                               const uint64_t* __restrict__ b, size_t count) {
    const size_t stride = (size_t)gridDim.x * kThreads;
    const ModParams unused{};
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < count; i += stride)
        out[i] = gold_mul(ArithGold::load(a[i], unused), ArithGold::load(b[i], unused));
}

// out[b][i] = in[b][bitrev(i)] — boundary permutation between the natural order of rust-api/lambda-snark/src/ntt.rs and the
// bit-reversed order the butterfly network produces/consumes
static __global__ void __launch_bounds__(kThreads) bit_reverse_kernel(uint64_t* __restrict__ out, const uint64_t* __restrict__ in, int logn,
                                                                        size_t total) {
    const size_t stride = (size_t)gridDim.x * kThreads;
    const uint32_t mask = (1u << logn) - 1u;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += stride) {
        const uint32_t lane = (uint32_t)i & mask;
        out[i] = in[(i - lane) + (__brev(lane) >> (32 - logn))];
    }
}

}  // namespace lsr
