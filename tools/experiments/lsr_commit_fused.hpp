// Fused middle of the Module-LWE matrix–vector product u = INTT(A_hat^T NTT(r)) for n >= 4096 (K8 of SURVEY.md §2a):
// one workgroup owns the same 4096-residue block T of the K polynomials of one witness vector and runs
//     the forward tile rounds of r_0..r_{K-1} (in lockstep: one twiddle fetch serves all K),
//     u_c[T] = sum_i A_hat[i][c][T] * r_hat_i[T]  from registers, c = 0..K-1,
//     the inverse tile rounds of each u_c,
// so NTT(r) and A_hat^T NTT(r) never travel to HBM: per witness vector the unfused sequence moves (forward tile pass,
// matvec, inverse tile pass) 6 K polynomial reads/writes, this kernel 2 K.  n = 4096: the whole product in one launch (the
// blinding add rides in the last store); n > 4096: between the strided top-bits round of the forward transform (raw
// elements in) and that of the inverse (raw elements out), exactly where ntt_tile_forward / ntt_tile_inverse sit.
// FP64 flavour only (q < 2^45); other moduli and ranks use the unfused kernels.
#pragma once

#include "lsr_ntt_kernels.hpp"

namespace lsr {

template <int K, int LT, bool TWO_PASS>
__global__ void __launch_bounds__(kThreads, 2) mlwe_tile_fused(const uint64_t* __restrict__ r, uint64_t* __restrict__ u, const double* __restrict__ mat,
                                                              size_t vectors, ModParams p, const double* __restrict__ fwd_tw,
                                                              const double* __restrict__ inv_tw, RoundConsts<ArithF64> cs,
                                                              const uint64_t* __restrict__ add) {
    using A = ArithF64;
    __shared__ uint64_t lds[kLdsWords];
    constexpr int NR = TileRound<LT, 0>::kCount;
    const uint32_t t = threadIdx.x;
    // (witness vector j, block T).  Workgroups b and b + 8 share an XCD (observed round-robin placement; speed only): with at
    // least 8 blocks per polynomial an XCD then sees 1/8 of the block positions, so its slice of A_hat stays in its L2.
    const int tp_log = p.logn - kTileLog;
    const uint32_t b = blockIdx.x;
    uint32_t T;
    size_t j;
    if (tp_log >= 3) {
        const uint32_t rest = b >> 3;
        T = (b & 7u) | ((rest & ((1u << (tp_log - 3)) - 1u)) << 3);
        j = rest >> (tp_log - 3);
    } else {
        T = b & ((1u << tp_log) - 1u);
        j = b >> tp_log;
    }
    if (j >= vectors) return;
    const uint32_t nmask = (1u << p.logn) - 1u;
    const uint32_t block_pos = T << kTileLog;
    const rsrc_t ftable = make_rsrc(fwd_tw, 8u << p.logn);
    const rsrc_t itable = make_rsrc(inv_tw, 8u << p.logn);
    double R[K][kRegs];
    double w[2][kRoundTwiddles];

    // ---- forward tile rounds, K polynomials in lockstep -------------------------------------------------------------
    {
        constexpr int LO = TileRound<LT, 0>::LO, RR = TileRound<LT, 0>::R;
        const uint32_t base = lane_base<LO, RR>(t);
        uint64_t raw[K][kRegs];
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const rsrc_t tile = make_rsrc(r + (((j * K + i) << p.logn) + block_pos), kTile * 8u);
#pragma unroll
            for (int k = 0; k < kRegs; ++k) raw[i][k] = buf_load64(tile, base * 8u, reg_offset<LO, RR>(k) * 8u);
        }
        load_round_twiddles<A, LO, RR, false, false>(w[0], base, block_pos, nmask, p.logn, ftable);
#pragma unroll
        for (int i = 0; i < K; ++i)
#pragma unroll
            for (int k = 0; k < kRegs; ++k) R[i][k] = TWO_PASS ? elem_from_bits<A>(raw[i][k]) : A::load(raw[i][k], p);
    }
    static_for<0, NR>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
        constexpr int LO = TileRound<LT, I>::LO, RR = TileRound<LT, I>::R;
        if constexpr (I + 1 < NR) {
            constexpr int LO1 = TileRound<LT, I + 1>::LO, R1 = TileRound<LT, I + 1>::R;
            load_round_twiddles<A, LO1, R1, false, false>(w[(I + 1) & 1], lane_base<LO1, R1>(t), block_pos, nmask, p.logn, ftable);
        }
#pragma unroll
        for (int i = 0; i < K; ++i) forward_round<A, LO, RR>(R[i], w[I & 1], p);
        if constexpr (I + 1 < NR) {   // regroup each polynomial's block through the LDS tile
            constexpr int LO1 = TileRound<LT, I + 1>::LO, R1 = TileRound<LT, I + 1>::R;
            uint64_t* const row = lds + lds_slot(lane_base<LO, RR>(t));
            const uint64_t* const row1 = lds + lds_slot(lane_base<LO1, R1>(t));
#pragma unroll
            for (int i = 0; i < K; ++i) {
#pragma unroll
                for (int k = 0; k < kRegs; ++k) row[lds_slot(reg_offset<LO, RR>(k))] = elem_bits<A>(R[i][k]);
                __syncthreads();
#pragma unroll
                for (int k = 0; k < kRegs; ++k) R[i][k] = elem_from_bits<A>(row1[lds_slot(reg_offset<LO1, R1>(k))]);
                __syncthreads();
            }
        }
    });

    // ---- per output component: product with the matrix column from registers, then the inverse tile rounds -----------
    constexpr int LOl = TileRound<LT, NR - 1>::LO, Rl = TileRound<LT, NR - 1>::R;
    const uint32_t basel = lane_base<LOl, Rl>(t);
    static_for<0, K>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        double v[kRegs];
        load_round_twiddles<A, LOl, Rl, true, (NR == 1) && !TWO_PASS>(w[0], basel, block_pos, nmask, p.logn, itable);
#pragma unroll
        for (int k = 0; k < kRegs; ++k) v[k] = 0.0;
#pragma unroll
        for (int i = 0; i < K; ++i) {   // u_c = sum_i A_hat[i][c] r_hat_i
            const rsrc_t col = make_rsrc(mat + (((size_t)(i * K + c) << p.logn) + block_pos), kTile * 8u);
            uint64_t m[kRegs];
#pragma unroll
            for (int k = 0; k < kRegs; k += 2) buf_load128(col, basel * 8u, reg_offset<LOl, Rl>(k) * 8u, m[k], m[k + 1]);
#pragma unroll
            for (int k = 0; k < kRegs; ++k) v[k] += mulmod_f64(R[i][k], __longlong_as_double((long long)m[k]), p.qd, p.inv_qd);
        }
#pragma unroll
        for (int k = 0; k < kRegs; ++k) v[k] = recentre_f64(v[k], p.qd, p.inv_qd);
        const rsrc_t out = make_rsrc(u + (((j * K + c) << p.logn) + block_pos), kTile * 8u);

        static_for<0, NR>([&](auto ic) {
            constexpr int I = decltype(ic)::value;          // I-th inverse round = forward round NR-1-I
            constexpr int J = NR - 1 - I;
            constexpr int LO = TileRound<LT, J>::LO, RR = TileRound<LT, J>::R;
            constexpr bool kLast = (I == NR - 1);
            constexpr bool kFinal = kLast && !TWO_PASS;
            const uint32_t base = lane_base<LO, RR>(t);
            uint64_t* const row = lds + lds_slot(base);
            if constexpr (I > 0) {
#pragma unroll
                for (int k = 0; k < kRegs; ++k) v[k] = elem_from_bits<A>(row[lds_slot(reg_offset<LO, RR>(k))]);
            }
            if constexpr (!kLast) {
                constexpr int LO1 = TileRound<LT, J - 1>::LO, R1 = TileRound<LT, J - 1>::R;
                constexpr bool kNextFinal = (I + 1 == NR - 1) && !TWO_PASS;
                load_round_twiddles<A, LO1, R1, true, kNextFinal>(w[(I + 1) & 1], lane_base<LO1, R1>(t), block_pos, nmask, p.logn, itable);
            }
            uint64_t blind[kFinal ? kRegs : 1];
            if constexpr (kFinal) {
                if (add != nullptr) {
                    const rsrc_t extra = make_rsrc(add + (((j * K + c) << p.logn) + block_pos), kTile * 8u);
#pragma unroll
                    for (int k = 0; k < kRegs; ++k) blind[k] = buf_load64(extra, base * 8u, reg_offset<LO, RR>(k) * 8u);
                }
            }
            inverse_round<A, LO, RR, kFinal>(v, w[I & 1], p, cs);
            if constexpr (!kFinal) {
#pragma unroll
                for (int k = 0; k < kRegs; ++k) A::end_of_inverse_round(v[k], p);
            }
            if constexpr (kLast) {
                if (kFinal && add != nullptr) {
#pragma unroll
                    for (int k = 0; k < kRegs; ++k) buf_store64(out, base * 8u, reg_offset<LO, RR>(k) * 8u, A::store_reduced_plus(v[k], blind[k], p));
                } else {
#pragma unroll
                    for (int k = 0; k < kRegs; ++k)
                        buf_store64(out, base * 8u, reg_offset<LO, RR>(k) * 8u, TWO_PASS ? elem_bits<A>(v[k]) : A::store_reduced(v[k], p));
                }
                if constexpr (NR > 1) __syncthreads();   // the next component reuses the LDS tile
            } else {
#pragma unroll
                for (int k = 0; k < kRegs; ++k) row[lds_slot(reg_offset<LO, RR>(k))] = elem_bits<A>(v[k]);
                __syncthreads();
            }
        });
    });
}

}  // namespace lsr
