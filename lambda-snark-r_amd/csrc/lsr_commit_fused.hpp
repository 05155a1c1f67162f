// Fused middle stage of the Module-LWE matrix–vector product u = INTT(A_hat^T NTT(r)) + e1 for n = 2^16 / 2^17
// ("fused NTT -> MAC -> INTT pipeline"; contract cpp-core/include/lambda_snark/commitment.h:43-52).
//
// Pipeline per chunk of witness vectors (lsr_commit.hip::mlwe_matvec_fused):
//     ntt_strided_round (top index bits, r -> workspace, raw f64)                      4 + 4 polynomial passes per vector
//     mlwe_mid_fused8   (this file: 12 forward stages x K, A_hat^T product, 12 inverse stages x K)      4 + 4
//     ntt_strided_round (top bits of the inverse, n^-1, + e1, canonical store)         4 + 4 + 4
// i.e. 28 polynomial passes per rank-4 commitment instead of 44: NTT(r) and A_hat^T NTT(r) never reach HBM.
//
// One 512-lane workgroup owns the same 4096-residue tile of the K polynomials of one witness vector.  A lane holds 8
// residues (radix-8 rounds: index bits [9,12), [6,9), [3,6), [0,3)), so the K accumulators of the product cost 8 K doubles
// per lane and the kernel stays at 128 VGPRs = 4 waves per SIMD = two workgroups per CU (the round-1 experiment with 16
// residues per lane sat at 256 VGPRs + spills, 2 waves per SIMD: profiles/r01_fused_commit_experiment.txt).  The K
// polynomials go through the tile one after the other; only the accumulators persist.
//
// LDS (69 120 B per workgroup):
//   * the tile, 4608 doubles, at slot(idx) = sum_j W_j idx_j with W = {1,2,4,8,16,32,72,144,289,578,1156,2304}: a weighted
//     sum of index bits is additive over disjoint bit fields (lane part + immediate register offset), and the weights are
//     chosen so that, with the lane maps below, the 32 lanes of a ds_read_b64 group hit 32 different 8-byte banks and the 16
//     lanes of a ds_write_b64 group 16 different ones in every round (tools/experiments/sim_fused8.py checks both and the
//     whole butterfly network against the plain stage loop on the CPU);
//   * twiddles: rounds [9,12) and [6,9) need workgroup- resp. wavefront-uniform multipliers (scalar loads); round [3,6) reads
//     a 448-entry image of the three sub-tables in natural order; round [0,3) multipliers are private to a lane and parked
//     thread-major (7 x 512 doubles) instead of occupying 14 VGPRs.  The image is rebuilt for the inverse direction.
// The matrix is read in a lane-major permuted copy (16 B per lane, coalesced): a_perm[tile][i][c][kp][lane][2].
// FP64 flavour only (q < 2^45); other moduli, ranks > 4 and other degrees use the unfused kernels.
#pragma once

#include "lsr_ntt_kernels.hpp"

#ifndef LSR_F8_TOUCH
#define LSR_F8_TOUCH 1
#endif
// Exchange between the rounds of index bits [3,6) and [0,3): 2 (default) = inside the wavefront with explicit DPP moves
// (quad_perm for lane^1, lane^2; row_shl/row_shr:4 under bank masks for lane^4), 0 = through the LDS tile (+ barrier).  The
// register field swaps places with lane bits 0..2 — an 8 x 8 transpose among 8 neighbouring lanes — so no other lane of the
// workgroup is involved and the exchange needs neither LDS storage nor a barrier (north_star: "ds_swizzle/ds_permute for the
// intra-wavefront transpose stages").  Measured (profiles/r02_fused_wave_exchange.txt, r02_fused_mid_ablation.txt): DPP -21.5 %
// LDS instructions, +14.7 % VALU, pipeline -1.5 %; the __shfl_xor form (-> ds_bpermute) and a barrier-free round trip through
// a wave-private, XOR-swizzled LDS block were both slower and are not kept.
#ifndef LSR_F8_WAVE_XCHG
#define LSR_F8_WAVE_XCHG 2
#endif
// cache policy of the streamed tile operands and results (0 = default, 2 = nt): they pass through the XCD's L2 once, the matrix
// slice is re-read by every workgroup
#ifndef LSR_F8_LOAD_AUX
#define LSR_F8_LOAD_AUX 2
#endif
#ifndef LSR_F8_STORE_AUX
#define LSR_F8_STORE_AUX 2
#endif
// matrix loads one component ahead of their products (double-buffered in the registers the round's multipliers vacate)
#ifndef LSR_F8_MAT_AHEAD
#define LSR_F8_MAT_AHEAD 1
#endif

namespace lsr {

constexpr int kF8Threads = 512;
constexpr int kF8Regs = 8;
constexpr uint32_t kF8TileWords = 4608;
constexpr uint32_t kF8TwShared = 448;              // sub-tables of index bits 5, 4, 3: 64 + 128 + 256 entries
constexpr uint32_t kF8TwPrivate = 7 * kF8Threads;  // index bits 2, 1, 0: 7 multipliers per lane

__host__ __device__ constexpr uint32_t f8_slot(uint32_t idx) {
    return (idx & 63u) + 72u * ((idx >> 6) & 1u) + 144u * ((idx >> 7) & 1u) + 289u * ((idx >> 8) & 1u) + 578u * ((idx >> 9) & 1u) +
           1156u * ((idx >> 10) & 1u) + 2304u * ((idx >> 11) & 1u);
}
// low index bit of the register field of round R (R = 0 is the first forward round)
__host__ __device__ constexpr int f8_lo(int round) { return 9 - 3 * round; }
// tile index of register 0 of lane t in round R
template <int R>
__host__ __device__ constexpr uint32_t f8_base(uint32_t t) {
    if (R == 0) return t;
    if (R == 1) return (t & 63u) | ((t >> 6) << 9);
    if (R == 2) return (t & 7u) | ((t >> 3) << 6);
#if LSR_F8_WAVE_XCHG
    return ((t & 7u) << 3) | ((t >> 3) << 6);        // lane bits 0..2 carry index bits 3..5 after the in-wavefront transpose
#else
    return (((t >> 3) & 31u) << 3) | ((t & 7u) << 8) | ((t >> 8) << 11);
#endif
}

// value of lane (self ^ M), M in {1, 2, 4}
template <int M>
__device__ __forceinline__ double f8_xor_lane(double x) {
    const long long bits = __double_as_longlong(x);
    int lo = (int)bits, hi = (int)(bits >> 32);
    if constexpr (M == 1) {
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false);   // quad_perm:[1,0,3,2]
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false);
    } else if constexpr (M == 2) {
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xF, 0xF, false);   // quad_perm:[2,3,0,1]
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xF, 0xF, false);
    } else {
        const int l0 = lo, h0 = hi;
        lo = __builtin_amdgcn_update_dpp(l0, l0, 0x104, 0xF, 0x5, false);   // row_shl:4 into banks 0, 2 (lanes with bit 2 clear)
        lo = __builtin_amdgcn_update_dpp(lo, l0, 0x114, 0xF, 0xA, false);   // row_shr:4 into banks 1, 3
        hi = __builtin_amdgcn_update_dpp(h0, h0, 0x104, 0xF, 0x5, false);
        hi = __builtin_amdgcn_update_dpp(hi, h0, 0x114, 0xF, 0xA, false);
    }
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// registers <-> lane bits 0..2: element (lane bit s = a, register bit s = b) moves to (lane bit s = b, register bit s = a)
__device__ __forceinline__ void f8_transpose_regs_lanes(double (&v)[kF8Regs], uint32_t t) {
    static_for<0, 3>([&](auto sc) {
        constexpr int S = decltype(sc)::value;
        constexpr int M = 1 << S;
        const bool up = (t >> S) & 1u;
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) {
            if (k & M) continue;
            const double send = up ? v[k] : v[k | M];
            const double recv = f8_xor_lane<M>(send);
            v[k] = up ? recv : v[k];
            v[k | M] = up ? v[k | M] : recv;
        }
    });
}

// multipliers of one round in the order [j = 2][j = 1: u = 0, 1][j = 0: u = 0..3] (j = register bit); `tw(s)` yields the
// s-th of them at its point of use, so that multipliers parked in LDS are not all live at once
template <class TW>
__device__ __forceinline__ void f8_forward_round(double (&v)[kF8Regs], TW&& tw, const ModParams& p) {
    {
        const double w = tw(0);
#pragma unroll
        for (int l = 0; l < 4; ++l) ArithF64::ct(v[l], v[l + 4], w, p);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const double w = tw(1 + u);
#pragma unroll
        for (int l = 0; l < 2; ++l) ArithF64::ct(v[4 * u + l], v[4 * u + l + 2], w, p);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) ArithF64::ct(v[2 * u], v[2 * u + 1], tw(3 + u), p);
}
// RECENTRE: bring the outputs back to |v| <= q/2.  A round multiplies the bound of the sum outputs by 8, and every
// product input must stay below 2^50 > 32 q: from q/2 two rounds may pass (4 q, 32 q) before a re-centring.
template <bool RECENTRE, class TW>
__device__ __forceinline__ void f8_inverse_round(double (&v)[kF8Regs], TW&& tw, const ModParams& p) {
#pragma unroll
    for (int u = 0; u < 4; ++u) ArithF64::gs(v[2 * u], v[2 * u + 1], tw(3 + u), p);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const double w = tw(1 + u);
#pragma unroll
        for (int l = 0; l < 2; ++l) ArithF64::gs(v[4 * u + l], v[4 * u + l + 2], w, p);
    }
    {
        const double w = tw(0);
#pragma unroll
        for (int l = 0; l < 4; ++l) ArithF64::gs(v[l], v[l + 4], w, p);
    }
    if constexpr (RECENTRE) {
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) v[k] = recentre_f64(v[k], p.qd, p.inv_qd);
    }
}

// table indices of the multipliers of a round whose register-0 position inside the polynomial is pos0
template <int R>
__device__ __forceinline__ void f8_uniform_twiddles(double (&w)[7], const double* __restrict__ table, uint32_t pos0, int logn) {
    constexpr int LO = f8_lo(R);
#pragma unroll
    for (int j = 2; j >= 0; --j) {
        const int b = LO + j;
        const uint32_t first = (1u << (logn - 1 - b)) + (pos0 >> (b + 1));
#pragma unroll
        for (int u = 0; u < (1 << (2 - j)); ++u) w[((1 << (2 - j)) - 1) + u] = table[first + u];
    }
}

// (re)build the LDS twiddle image for one direction; the caller separates it from its readers with barriers
__device__ __forceinline__ void f8_fill_twiddles(double* __restrict__ tw_lds, const double* __restrict__ table, uint32_t tile_pos, int logn, uint32_t t) {
    if (t < kF8TwShared) {
        const int b = t < 64 ? 5 : (t < 192 ? 4 : 3);
        const uint32_t e = t < 64 ? t : (t < 192 ? t - 64 : t - 192);
        tw_lds[t] = table[(1u << (logn - 1 - b)) + (tile_pos >> (b + 1)) + e];
    }
    const uint32_t pos0 = tile_pos + f8_base<3>(t);
    double mine[7];
    f8_uniform_twiddles<3>(mine, table, pos0, logn);
#pragma unroll
    for (int s = 0; s < 7; ++s) tw_lds[kF8TwShared + s * kF8Threads + t] = mine[s];
}

// a_perm[tile][i][c][kp][lane][e] = (double) a_hat[i][c][tile * 4096 + f8_base<3>(lane) + 2 kp + e]
static __global__ void __launch_bounds__(256) f8_permute_matrix_kernel(double* __restrict__ a_perm, const uint64_t* __restrict__ a_hat, uint32_t k,
                                                                         int logn) {
    const uint64_t total = (uint64_t)k * k << logn;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += stride) {
        const uint32_t e = g & 1u, lane = (g >> 1) & 511u, kp = (g >> 10) & 3u;
        const uint64_t rest = g >> 12;                       // (tile * k + i) * k + c
        const uint32_t c = rest % k, i = (rest / k) % k;
        const uint64_t tile = rest / ((uint64_t)k * k);
        const uint64_t x = (tile << 12) + f8_base<3>(lane) + 2 * kp + e;
        a_perm[g] = f64_from_u52(a_hat[(((uint64_t)i * k + c) << logn) + x]);
    }
}

// blk: the workgroup's position in the (witness vector, tile) grid — the hardware block index when the stage is a kernel of its
// own, a number handed out by mlwe_mixed (below) when it is one role of a mixed launch
template <int K>
__device__ __forceinline__ void mlwe_mid_fused8_body(uint32_t blk, const uint64_t* __restrict__ rws, uint64_t* __restrict__ u,
                                                     const double* __restrict__ a_perm, uint32_t vectors, const ModParams& p,
                                                     const double* __restrict__ fwd_tw, const double* __restrict__ inv_tw) {
    __shared__ double tile_lds[kF8TileWords];
    __shared__ double tw_lds[kF8TwShared + kF8TwPrivate];
    const uint32_t t = threadIdx.x;
    // (witness vector j, tile).  Workgroups b and b + 8 share an XCD under the observed round-robin placement (speed only):
    // an XCD then sees 2 of the >= 16 tile positions of A_hat, which stay in its L2.
    const int tp_log = p.logn - 12;
    const uint32_t rest = blk >> 3;
    const uint32_t tile = (blk & 7u) | ((rest & ((1u << (tp_log - 3)) - 1u)) << 3);
    const uint32_t j = rest >> (tp_log - 3);
    if (j >= vectors) return;
    const uint32_t tile_pos = tile << 12;
    const uint32_t wave0 = __builtin_amdgcn_readfirstlane(t & ~63u);
    // per-round LDS addresses of register 0 (doubles)
    double* const row0 = tile_lds + f8_slot(f8_base<0>(t));
    double* const row1 = tile_lds + f8_slot(f8_base<1>(t));
    double* const row2 = tile_lds + f8_slot(f8_base<2>(t));
    [[maybe_unused]] double* const row3 = tile_lds + f8_slot(f8_base<3>(t));
    const double* const tw2 = tw_lds;                                   // natural-order sub-tables of bits 5, 4, 3
    const double* const tw3 = tw_lds + kF8TwShared + t;                 // this lane's 7 multipliers, stride 512
    const uint32_t e5 = f8_base<2>(t) >> 6, e4 = f8_base<2>(t) >> 5, e3 = f8_base<2>(t) >> 4;

    double acc[K][kF8Regs];
#pragma unroll
    for (int c = 0; c < K; ++c)
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) acc[c][k] = 0.0;

    f8_fill_twiddles(tw_lds, fwd_tw, tile_pos, p.logn, t);
    double w0[7], w1[7];
    f8_uniform_twiddles<0>(w0, fwd_tw, tile_pos, p.logn);
    f8_uniform_twiddles<1>(w1, fwd_tw, tile_pos + f8_base<1>(wave0), p.logn);
    __syncthreads();

    // ---- forward: the K polynomials of the vector, one after the other ---------------------------------------------------
    const auto tw_r0 = [&](int s_) { return w0[s_]; };
    const auto tw_r1 = [&](int s_) { return w1[s_]; };
    const auto tw_r2 = [&](int s_) { return s_ == 0 ? tw2[e5] : (s_ < 3 ? tw2[64 + e4 + (s_ - 1)] : tw2[192 + e3 + (s_ - 3)]); };
    const auto tw_r3 = [&](int s_) { return tw3[s_ * kF8Threads]; };
#pragma unroll 1
    for (int i = 0; i < K; ++i) {
        double v[kF8Regs];
        {
            const rsrc_t src = make_rsrc(rws + ((((size_t)j * K + i) << p.logn) + tile_pos), 4096u * 8u);
#pragma unroll
            for (int k = 0; k < kF8Regs; ++k) v[k] = __longlong_as_double((long long)buf_load64<LSR_F8_LOAD_AUX>(src, t * 8u, (uint32_t)k * 4096u));
        }
        f8_forward_round(v, tw_r0, p);
        if (i > 0) __syncthreads();                      // the previous polynomial's last LDS reads are done
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) row0[f8_slot((uint32_t)k << 9)] = v[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) v[k] = row1[f8_slot((uint32_t)k << 6)];
#if LSR_F8_TOUCH
        // Pull the next polynomial's tile towards this XCD's L2 three rounds ahead of its use: one dword per 128-byte line
        // (a real prefetch into registers would need 16 more VGPRs than the 128 that two workgroups per CU allow).
        uint32_t touch = 0;
        if (i + 1 < K && t < 256) {
            const rsrc_t nxt = make_rsrc(rws + ((((size_t)j * K + i + 1) << p.logn) + tile_pos), 4096u * 8u);
            touch = __builtin_amdgcn_raw_buffer_load_b32(nxt, (int)(t * 128u), 0, 0);
        }
#endif
        f8_forward_round(v, tw_r1, p);
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) row1[f8_slot((uint32_t)k << 6)] = v[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) v[k] = row2[f8_slot((uint32_t)k << 3)];
        f8_forward_round(v, tw_r2, p);
#if LSR_F8_MAT_AHEAD
        double2 a[4];
        const rsrc_t slab = make_rsrc(a_perm + ((((size_t)tile * K + i) * K) << 12), (uint32_t)K * 4u * 512u * 16u);
        const auto fetch = [&](int c, int kp) {
            uint64_t lo, hi;
            buf_load128(slab, t * 16u, (uint32_t)(c * 4 + kp) * 8192u, lo, hi);
            return make_double2(__longlong_as_double((long long)lo), __longlong_as_double((long long)hi));
        };
#endif
#if LSR_F8_WAVE_XCHG
        f8_transpose_regs_lanes(v, t);
#else
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) row2[f8_slot((uint32_t)k << 3)] = v[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) v[k] = row3[k];
#endif
        f8_forward_round(v, tw_r3, p);
        // acc[c] += A_hat[i][c] o r_hat_i on this lane's 8 positions
        [[maybe_unused]] const double* const mat = a_perm + ((((size_t)tile * K + i) * K) << 12) + (size_t)t * 2;
#if LSR_F8_MAT_AHEAD
        {   // rolling window of four 16-byte matrix loads: each slot is refilled for the next component as soon as its two
            // products are issued, so a wave waits for L2 once per polynomial instead of once per load
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) a[kp] = fetch(0, kp);
            static_for<0, K>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
#pragma unroll
                for (int kp = 0; kp < 4; ++kp) {
                    const double2 cur = a[kp];
                    if constexpr (c + 1 < K) a[kp] = fetch(c + 1, kp);
                    acc[c][2 * kp] += mulmod_f64(v[2 * kp], cur.x, p.qd, p.inv_qd);
                    acc[c][2 * kp + 1] += mulmod_f64(v[2 * kp + 1], cur.y, p.qd, p.inv_qd);
                }
            });
        }
#else
#pragma unroll
        for (int c = 0; c < K; ++c) {
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) {
                const double2 a = *reinterpret_cast<const double2*>(mat + (((size_t)c * 4 + kp) << 10));
                acc[c][2 * kp] += mulmod_f64(v[2 * kp], a.x, p.qd, p.inv_qd);
                acc[c][2 * kp + 1] += mulmod_f64(v[2 * kp + 1], a.y, p.qd, p.inv_qd);
            }
        }
#endif
#if LSR_F8_TOUCH
        asm volatile("" ::"v"(touch));
#endif
    }

    // ---- inverse: each output component through the tile ------------------------------------------------------------------
    __syncthreads();                                     // every lane is done with the forward twiddle image
    f8_fill_twiddles(tw_lds, inv_tw, tile_pos, p.logn, t);
    f8_uniform_twiddles<0>(w0, inv_tw, tile_pos, p.logn);
    f8_uniform_twiddles<1>(w1, inv_tw, tile_pos + f8_base<1>(wave0), p.logn);
    __syncthreads();
    static_for<0, K>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        double x[kF8Regs];
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) x[k] = recentre_f64(acc[c][k], p.qd, p.inv_qd);
        f8_inverse_round<false>(x, tw_r3, p);             // |x| <= 4 q
#if LSR_F8_WAVE_XCHG
        f8_transpose_regs_lanes(x, t);
        f8_inverse_round<true>(x, tw_r2, p);              // 32 q -> q/2
        if constexpr (c > 0) __syncthreads();            // the previous component's last LDS reads are done
#else
        if constexpr (c > 0) __syncthreads();            // the previous component's last LDS reads are done
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) row3[k] = x[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) x[k] = row2[f8_slot((uint32_t)k << 3)];
        f8_inverse_round<true>(x, tw_r2, p);              // 32 q -> q/2
#endif
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) row2[f8_slot((uint32_t)k << 3)] = x[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) x[k] = row1[f8_slot((uint32_t)k << 6)];
        f8_inverse_round<false>(x, tw_r1, p);             // 4 q
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) row1[f8_slot((uint32_t)k << 6)] = x[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) x[k] = row0[f8_slot((uint32_t)k << 9)];
        f8_inverse_round<true>(x, tw_r0, p);              // 32 q -> q/2: what the strided round expects
        const rsrc_t dst = make_rsrc(u + ((((size_t)j * K + c) << p.logn) + tile_pos), 4096u * 8u);
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) buf_store64<LSR_F8_STORE_AUX>(dst, t * 8u, (uint32_t)k * 4096u, (uint64_t)__double_as_longlong(x[k]));
    });
}

template <int K>
__global__ void __launch_bounds__(kF8Threads, 4) mlwe_mid_fused8(const uint64_t* __restrict__ rws, uint64_t* __restrict__ u,
                                                                  const double* __restrict__ a_perm, uint32_t vectors, ModParams p,
                                                                  const double* __restrict__ fwd_tw, const double* __restrict__ inv_tw) {
    mlwe_mid_fused8_body<K>(blockIdx.x, rws, u, a_perm, vectors, p, fwd_tw, inv_tw);
}

// =================================================================================================================================
// MIXED launch (late round 2, n = 2^16, 4 + 12 split).  The three kernels of a chunk are each at their own bound — the strided rounds
// at the memory system's, the middle stage at FP64 issue — and running them on two streams only makes every kernel slower: the
// dispatcher hands a CU to whichever kernel has workgroups waiting, the middle stage (two workgroups = the whole register file of a
// CU) keeps the strided rounds out, and those are latency-bound at the few slots they get (profiles/r02_commit_split_88.txt).
// Here ONE launch carries the workgroups of three independent pieces of work, in an interleaved block order:
//     middle stage of chunk c  |  forward strided round of chunk c + 1 (r -> workspace)  |  inverse strided round (+ e1) of chunk c - 1
// Launches follow each other on one stream, so the dependencies F(c) -> M(c) -> I(c) are launch boundaries and nothing is
// synchronised inside the kernel.  Every workgroup has the middle stage's shape (512 lanes, 128 VGPRs, 69 KB of LDS: two per CU); a
// strided-round workgroup plays two of the round's 256-lane groups and ignores the LDS.  The block order (units of 8 workgroups, so
// that the middle stage keeps its block-index-mod-8 = XCD affinity) decides the mix on a CU: `s_per_m` strided units follow each
// middle unit, and whenever a slot frees up the next block in line takes it — so a CU mostly holds one FP64-bound and one
// memory-bound workgroup, and the two pipes of the CU are busy at the same time.
// =================================================================================================================================
// forward strided round (top four index bits, out of place, raw out) on G independent 256-lane groups per call: G x 16 loads in
// flight per lane.  A strided-round workgroup of the mixed launch holds a slot with the middle stage's 128 VGPRs; with one group
// per lane half of them idle and the round is latency-bound at the few slots it gets.
template <int G>
__device__ __forceinline__ void mixed_forward_round(uint64_t* __restrict__ dst, const uint64_t* __restrict__ src, size_t total, int lo, const ModParams& p,
                                                    const double* __restrict__ tw, uint32_t vblock0, uint32_t vthread) {
    using A = ArithF64;
    double v[G][16];
    size_t idx0[G];
    bool live[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const size_t group = (size_t)(vblock0 + g) * kThreads + vthread;
        live[g] = group < (total >> 4);
        const size_t low = group & (((size_t)1 << lo) - 1);
        idx0[g] = ((group >> lo) << (lo + 4)) | low;
        if (live[g]) {
#pragma unroll
            for (int k = 0; k < 16; ++k) v[g][k] = A::load(__builtin_nontemporal_load(src + idx0[g] + ((size_t)k << lo)), p);
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (!live[g]) continue;
#pragma unroll
        for (int j = 3; j >= 0; --j) {
            const int half = 1 << j;
#pragma unroll
            for (int u = 0; u < (1 << (3 - j)); ++u) {
                const double w = tw[(1 << (3 - j)) + u];
#pragma unroll
                for (int l = 0; l < half; ++l) A::ct(v[g][(u << (j + 1)) | l], v[g][((u << (j + 1)) | l) + half], w, p);
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) dst[idx0[g] + ((size_t)k << lo)] = (uint64_t)__double_as_longlong(v[g][k]);
    }
}

struct MixedJob {
    const uint64_t* m_ws; uint64_t* m_out; uint32_t m_vectors;        // middle stage: workspace -> u (raw), m_vectors witness vectors
    uint64_t* f_dst; const uint64_t* f_src; uint32_t f_polys;         // forward round: caller's r -> workspace (raw), f_polys polynomials
    uint64_t* i_data; const uint64_t* i_add; uint32_t i_polys;        // inverse round in place on u + i_add (canonical, never NULL), i_polys polynomials
    uint32_t units_m, units_f, units_i;                               // units of 8 workgroups per role
    uint32_t s_per_m, periods;                                        // interleaved part: periods x (1 middle unit + s_per_m strided units)
    uint32_t f_groups;                                                // 256-lane groups per half workgroup in the forward role: 1 or 2
};

template <int K>
__global__ void __launch_bounds__(kF8Threads, 4) mlwe_mixed(MixedJob job, const double* __restrict__ a_perm, ModParams p,
                                                             const double* __restrict__ fwd_tw, const double* __restrict__ inv_tw,
                                                             RoundConsts<ArithF64> cs) {
    const uint32_t x = blockIdx.x >> 3, sub = blockIdx.x & 7u;
    const uint32_t per = job.s_per_m + 1u, inter = job.periods * per;
    bool middle;
    uint32_t unit;                                                    // index within its role class (middle | strided)
    if (x < inter) {
        const uint32_t q = x / per, rem = x - q * per;
        middle = rem == 0;
        unit = middle ? q : q * job.s_per_m + rem - 1u;
    } else {
        const uint32_t y = x - inter, left_m = job.units_m - job.periods;
        middle = y < left_m;
        unit = middle ? job.periods + y : job.periods * job.s_per_m + (y - left_m);
    }
    if (middle) {
        mlwe_mid_fused8_body<K>(unit * 8u + sub, job.m_ws, job.m_out, a_perm, job.m_vectors, p, fwd_tw, inv_tw);
        return;
    }
    // the strided roles issue ahead of the middle stage's waves on their SIMD: their few instructions put loads and stores in flight,
    // and the sooner a strided workgroup is through, the sooner its slot is free (2.92-2.96 vs 2.98-2.99 ms per 1024 vectors; the
    // opposite priority loses: profiles/r02_mixed_launch.txt)
    __builtin_amdgcn_s_setprio(3);
    // strided units: forward and inverse alternate while both last
    const uint32_t alt = job.units_f < job.units_i ? job.units_f : job.units_i;
    bool forward;
    uint32_t idx;
    if (unit < 2u * alt) { forward = !(unit & 1u); idx = unit >> 1; }
    else { forward = job.units_f > alt; idx = alt + (unit - 2u * alt); }
    const uint32_t vblock = (idx * 8u + sub) * 2u + (threadIdx.x >> 8), vthread = threadIdx.x & 255u;
    const int lo = p.logn - 4;
    if (forward) {
        if (job.f_groups == 2) mixed_forward_round<2>(job.f_dst, job.f_src, (size_t)job.f_polys << p.logn, lo, p, fwd_tw, vblock * 2u, vthread);
        else mixed_forward_round<1>(job.f_dst, job.f_src, (size_t)job.f_polys << p.logn, lo, p, fwd_tw, vblock, vthread);
    } else {
        strided_round_body<ArithF64, 4, true, true, false, true, 0>(job.i_data, (size_t)job.i_polys << p.logn, lo, p, inv_tw, cs, job.i_add, BlindSampler{},
                                                                    vblock, vthread);
    }
}




// =================================================================================================================================
// 8 + 8 split for n = 2^16 (round 2, second design).  The 4 + 12 pipeline above leaves the fused stage FP64-bound while the two
// strided rounds idle the ALUs at the memory floor; moving four stages of each direction into the outer passes costs those
// passes nothing (they stay memory-bound) and takes a third of the butterflies out of the middle:
//     cols8_forward   bits 15..8  r (canonical) -> workspace (raw)       16 columns x 256 rows per workgroup, one LDS exchange
//     mlwe_mid8<K>    bits 7..0, A_hat^T product, bits 0..7              no workgroup exchange at all
//     cols8_inverse   bits 8..15, n^-1, + e1, canonical store
// With only index bits 0..7 inside, every butterfly of mlwe_mid8 pairs residues of one 256-block; a wavefront owns whole
// blocks, so the regroupings between its three register rounds (bits [5,8), [2,5), [0,2)) are transposes between register
// bits and LANE bits of the same wavefront: v_permlane32_swap / v_permlane16_swap for lane bits 5 and 4, DPP row shifts under
// bank masks for lane bits 3 and 2, quad_perm for lane bit 1 (north_star: "ds_swizzle/ds_permute for the intra-wavefront
// transpose stages").  No LDS tile, no barrier between rounds: the eight wavefronts of a workgroup only share the twiddle image.
// Index algebra checked on the CPU by tools/experiments/sim_mid8.py.
// =================================================================================================================================

// COLS adjacent columns per workgroup (COLS * 16 lanes, 8 * COLS-byte row segments).  16: 256 lanes, 34 KiB of LDS; 32 (default
// since the late round-2 measurement, profiles/r02_cols_width.txt): 512 lanes, 64 KiB — data movement alone 0.809 ms per 4096
// polynomials against 0.867 ms with 128-byte segments and 0.779 ms for the strided round (tools/ubench_move2.hip).
template <int COLS> constexpr int c8_threads() { return COLS * 16; }
// tile position (row, col): COLS = 16 shifts every block of 16 rows by half a bank row so that the 32 lanes of a ds b64 group
// (16 columns x 2 row groups) stay on distinct banks in both exchange patterns; with 32 columns a group is one contiguous row segment
template <int COLS> __host__ __device__ constexpr uint32_t c8_slot(uint32_t row, uint32_t col) {
    return row * COLS + (COLS == 16 ? 16u * (row >> 4) : 0u) + col;
}
template <int COLS> constexpr uint32_t c8_lds_words() { return 256u * COLS + (COLS == 16 ? 256u : 0u); }

// forward: stages of polynomial index bits 15..8 on a tile of COLS adjacent columns x 256 rows (bits 8..15)
// STREAM: every global access with the nt policy, so that the pass leaves the XCD's L2 to a co-resident middle stage (two-lane schedule)
template <bool STREAM, int COLS>
__device__ __forceinline__ void cols8_forward_body(uint32_t blk, uint64_t* __restrict__ dst, const uint64_t* __restrict__ src, uint32_t polys,
                                                   const ModParams& p, const double* __restrict__ tw, double* __restrict__ lds) {
    using A = ArithF64;
    constexpr uint32_t CB = 256 / COLS;
    const uint32_t t = threadIdx.x;
    const uint32_t poly = blk / CB, cb = blk % CB;
    if (poly >= polys) return;
    const uint32_t col = t % COLS, rr = t / COLS;
    const size_t base = ((size_t)poly << 16) + cb * COLS + col;
    double v[16];
    // round 1: registers = bits 12..15, this lane's row bits 8..11 = rr
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint64_t* const at = src + base + ((uint32_t)k << 12) + (rr << 8);
        v[k] = A::load(STREAM ? __builtin_nontemporal_load(at) : *at, p);
    }
#pragma unroll
    for (int j = 3; j >= 0; --j) {
        const int half = 1 << j;
#pragma unroll
        for (int u = 0; u < (1 << (3 - j)); ++u) {
            const double w = tw[(1 << (3 - j)) + u];
#pragma unroll
            for (int l = 0; l < half; ++l) A::ct(v[(u << (j + 1)) | l], v[((u << (j + 1)) | l) + half], w, p);
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[c8_slot<COLS>(((uint32_t)k << 4) + rr, col)] = v[k];
    __syncthreads();
    // round 2: registers = bits 8..11, this lane's row bits 12..15 = rr
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = lds[c8_slot<COLS>((rr << 4) + (uint32_t)k, col)];
#pragma unroll
    for (int j = 3; j >= 0; --j) {                    // polynomial bit b = 8 + j: table[2^(15-b) + (row >> (j + 1))]
        const int half = 1 << j;
#pragma unroll
        for (int u = 0; u < (1 << (3 - j)); ++u) {
            const double w = tw[(1u << (7 - j)) + (rr << (3 - j)) + u];
#pragma unroll
            for (int l = 0; l < half; ++l) A::ct(v[(u << (j + 1)) | l], v[((u << (j + 1)) | l) + half], w, p);
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        uint64_t* const at = dst + base + (rr << 12) + ((uint32_t)k << 8);
        if (STREAM) __builtin_nontemporal_store((uint64_t)__double_as_longlong(v[k]), at);
        else *at = (uint64_t)__double_as_longlong(v[k]);
    }
}
template <bool STREAM, int COLS>
__global__ void __launch_bounds__(COLS * 16) cols8_forward(uint64_t* __restrict__ dst, const uint64_t* __restrict__ src, uint32_t polys, ModParams p,
                                                           const double* __restrict__ tw) {
    __shared__ double lds[c8_lds_words<COLS>()];
    cols8_forward_body<STREAM, COLS>(blockIdx.x, dst, src, polys, p, tw, lds);
}

// inverse: stages of bits 8..15 (Gentleman–Sande), n^-1 folded into the last one, + add (canonical residues, optional), canonical out
template <bool STREAM, int COLS>
__device__ __forceinline__ void cols8_inverse_body(uint32_t blk, uint64_t* __restrict__ data, uint32_t polys, const ModParams& p,
                                                   const double* __restrict__ tw, const RoundConsts<ArithF64>& cs, const uint64_t* __restrict__ add,
                                                   double* __restrict__ lds) {
    using A = ArithF64;
    constexpr uint32_t CB = 256 / COLS;
    const uint32_t t = threadIdx.x;
    const uint32_t poly = blk / CB, cb = blk % CB;
    if (poly >= polys) return;
    const uint32_t col = t % COLS, rr = t / COLS;
    const size_t base = ((size_t)poly << 16) + cb * COLS + col;
    double v[16];
    // round 1: registers = bits 8..11, lane row bits 12..15 = rr
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = __longlong_as_double((long long)__builtin_nontemporal_load(data + base + (rr << 12) + ((uint32_t)k << 8)));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int half = 1 << j;
#pragma unroll
        for (int u = 0; u < (1 << (3 - j)); ++u) {
            const double w = tw[(1u << (7 - j)) + (rr << (3 - j)) + u];
#pragma unroll
            for (int l = 0; l < half; ++l) A::gs(v[(u << (j + 1)) | l], v[((u << (j + 1)) | l) + half], w, p);
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[c8_slot<COLS>((rr << 4) + (uint32_t)k, col)] = recentre_f64(v[k], p.qd, p.inv_qd);
    __syncthreads();
    // round 2: registers = bits 12..15, lane row bits 8..11 = rr
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = lds[c8_slot<COLS>(((uint32_t)k << 4) + rr, col)];
    uint64_t blind[16];
    if (add != nullptr) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint64_t* const at = add + base + ((uint32_t)k << 12) + (rr << 8);
            blind[k] = STREAM ? __builtin_nontemporal_load(at) : *at;
        }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int half = 1 << j;
#pragma unroll
        for (int u = 0; u < (1 << (3 - j)); ++u) {
            const double w = tw[(1 << (3 - j)) + u];
#pragma unroll
            for (int l = 0; l < half; ++l) A::gs(v[(u << (j + 1)) | l], v[((u << (j + 1)) | l) + half], w, p);
        }
    }
#pragma unroll
    for (int l = 0; l < 8; ++l) A::gs_scaled(v[l], v[l + 8], cs.w_last_scaled, cs.n_inv, p);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint64_t out = add != nullptr ? A::store_reduced_plus(v[k], blind[k], p) : A::store_reduced(v[k], p);
        __builtin_nontemporal_store(out, data + base + ((uint32_t)k << 12) + (rr << 8));
    }
}
template <bool STREAM, int COLS>
__global__ void __launch_bounds__(COLS * 16) cols8_inverse(uint64_t* __restrict__ data, uint32_t polys, ModParams p, const double* __restrict__ tw,
                                                           RoundConsts<ArithF64> cs, const uint64_t* __restrict__ add) {
    __shared__ double lds[c8_lds_words<COLS>()];
    cols8_inverse_body<STREAM, COLS>(blockIdx.x, data, polys, p, tw, cs, add, lds);
}

// ---- the middle stage ----
// WAVES = 8: one 512-lane workgroup per 4096-residue tile.  WAVES = 4: one 256-lane workgroup per HALF tile (index bit 11 from
// the block index) with its LDS padded to 40.5 KiB, so that a CU takes exactly three of them (three waves per SIMD, at most 136
// VGPRs each) and keeps one wave slot per SIMD, 104 VGPRs and 34 KiB of LDS free: room for one workgroup of cols8_forward /
// cols8_inverse of a NEIGHBOURING chunk.  The FP64-bound middle stage and the memory-bound outer passes then run side by side
// on every CU (tools/ubench_concurrency.hip: two such kernels on two streams overlap completely when the first one's
// residency leaves room; the memory passes lose nothing at one workgroup per CU, profiles/r02_ubench_move2.txt).
template <int WAVES> constexpr uint32_t m8_entries(int b) { return (1u << (11 - b)) / (8 / WAVES); }      // image entries of stage b
template <int WAVES> constexpr uint32_t m8_image_offset(int b) {
    uint32_t off = 0;
    for (int s = 7; s > b; --s) off += m8_entries<WAVES>(s);
    return off;
}
template <int WAVES> constexpr uint32_t m8_image_words() { return m8_image_offset<WAVES>(0) + m8_entries<WAVES>(0); }
template <int WAVES> constexpr int m8_wave_bits() { return WAVES == 8 ? 3 : 2; }
// tile index of register k of (virtual) lane T in the first (registers = bits 5..7) and last (registers = bits 0, 1, 4) layout;
// T = lane | wave << 6 over the whole 4096-residue tile (with 4 waves per workgroup, T bit 8 = the workgroup's half)
__host__ __device__ constexpr uint32_t m8_idx_a(uint32_t T, uint32_t k) { return ((T >> 1) & 31u) | (k << 5) | ((T & 1u) << 8) | ((T >> 6) << 9); }
__host__ __device__ constexpr uint32_t m8_idx_c(uint32_t T, uint32_t k) {
    return (k & 3u) | (((T >> 1) & 3u) << 2) | ((k >> 2) << 4) | (((T >> 3) & 7u) << 5) | ((T & 1u) << 8) | ((T >> 6) << 9);
}
// compacted lane bits that select a twiddle of round a / round b (the other lane bits do not reach above the stage's bit)
__host__ __device__ constexpr uint32_t m8_sel_a(uint32_t T) { return (T & 1u) | ((T >> 6) << 1); }
__host__ __device__ constexpr uint32_t m8_sel_b(uint32_t T) { return (T & 1u) | ((T >> 3) << 1); }
// index bits 5..11 of this lane's residues once bits 2..4 are in registers (rounds b and c)
__host__ __device__ constexpr uint32_t m8_high(uint32_t T) { return ((T >> 3) & 7u) | ((T & 1u) << 3) | ((T >> 6) << 4); }

// one transpose step: register pair (lo = register bit clear, hi = set) against lane bit LB
template <int LB>
__device__ __forceinline__ void m8_swap_step(double& lo, double& hi, uint32_t t) {
    const long long lb = __double_as_longlong(lo), hb = __double_as_longlong(hi);
    unsigned a0 = (unsigned)lb, a1 = (unsigned)(lb >> 32), b0 = (unsigned)hb, b1 = (unsigned)(hb >> 32);
    if constexpr (LB == 5) {          // v_permlane32_swap: lanes 32..63 of the first operand <-> lanes 0..31 of the second
        const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        a0 = r0[0]; b0 = r0[1]; a1 = r1[0]; b1 = r1[1];
    } else if constexpr (LB == 4) {   // v_permlane16_swap: odd rows of the first operand <-> even rows of the second
        const auto r0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
        const auto r1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
        a0 = r0[0]; b0 = r0[1]; a1 = r1[0]; b1 = r1[1];
    } else if constexpr (LB == 3 || LB == 2) {
        // lanes with the bit set take the partner's `hi` into their `lo` (row_shr), lanes with it clear the partner's `lo` into
        // their `hi` (row_shl); bank masks select the lanes: bit 3 -> banks {2,3} / {0,1}, bit 2 -> banks {1,3} / {0,2}
        constexpr int N = 1 << LB;
        constexpr int up = LB == 3 ? 0xC : 0xA, down = LB == 3 ? 0x3 : 0x5;
        const int na0 = __builtin_amdgcn_update_dpp((int)a0, (int)b0, 0x110 + N, 0xF, up, false);
        const int na1 = __builtin_amdgcn_update_dpp((int)a1, (int)b1, 0x110 + N, 0xF, up, false);
        const int nb0 = __builtin_amdgcn_update_dpp((int)b0, (int)a0, 0x100 + N, 0xF, down, false);
        const int nb1 = __builtin_amdgcn_update_dpp((int)b1, (int)a1, 0x100 + N, 0xF, down, false);
        a0 = (unsigned)na0; a1 = (unsigned)na1; b0 = (unsigned)nb0; b1 = (unsigned)nb1;
    } else {                          // lane bit 1: quad_perm [2,3,0,1] and a select
        const bool upl = (t >> 1) & 1u;
        const unsigned s0 = upl ? a0 : b0, s1 = upl ? a1 : b1;
        const unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp((int)s0, (int)s0, 0x4E, 0xF, 0xF, false);
        const unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp((int)s1, (int)s1, 0x4E, 0xF, 0xF, false);
        a0 = upl ? r0 : a0; a1 = upl ? r1 : a1;
        b0 = upl ? b0 : r0; b1 = upl ? b1 : r1;
    }
    lo = __longlong_as_double(((long long)a1 << 32) | a0);
    hi = __longlong_as_double(((long long)b1 << 32) | b0);
}
// registers (bit 2, 1, 0) <-> lane bits (5, 4, 3): between the rounds of bits [5,8) and [2,5)
__device__ __forceinline__ void m8_transpose_1(double (&v)[8], uint32_t t) {
#pragma unroll
    for (int k = 0; k < 4; ++k) m8_swap_step<5>(v[k], v[k + 4], t);
#pragma unroll
    for (int k = 0; k < 8; ++k) if (!(k & 2)) m8_swap_step<4>(v[k], v[k + 2], t);
#pragma unroll
    for (int k = 0; k < 8; k += 2) m8_swap_step<3>(v[k], v[k + 1], t);
}
// registers (bit 1, 0) <-> lane bits (2, 1): between the rounds of bits [2,5) and [0,2)
__device__ __forceinline__ void m8_transpose_2(double (&v)[8], uint32_t t) {
#pragma unroll
    for (int k = 0; k < 8; ++k) if (!(k & 2)) m8_swap_step<2>(v[k], v[k + 2], t);
#pragma unroll
    for (int k = 0; k < 8; k += 2) m8_swap_step<1>(v[k], v[k + 1], t);
}

// Twiddle image, both directions at once (image[0 .. W) forward, image[W .. 2W) inverse, W = m8_image_words): every lane
// stores the 20 multipliers it will read (lanes that share one write the same value), every index is computed once and
// nothing of the fill stays live across the kernel.  t = lane within the workgroup (addresses), T = lane within the tile (indices).
template <int WAVES>
__device__ __forceinline__ void m8_fill_twiddles(double* __restrict__ image, const double* forward, const double* inverse, uint32_t tile_pos, uint32_t t,
                                                 uint32_t T) {
    constexpr int WB = m8_wave_bits<WAVES>();
    const rsrc_t tf = make_rsrc(forward, 8u << 16), ti = make_rsrc(inverse, 8u << 16);   // buffer loads: not hoisted, not spilled
    const auto put = [&](uint32_t slot, uint32_t index) {
        image[slot] = __longlong_as_double((long long)buf_load64(tf, index * 8u, 0));
        image[m8_image_words<WAVES>() + slot] = __longlong_as_double((long long)buf_load64(ti, index * 8u, 0));
    };
    const uint32_t sa = m8_sel_a(T), h = m8_high(T), low = (T >> 1) & 3u;
    const uint32_t la = m8_sel_a(t), lb = m8_sel_b(t);
#pragma unroll
    for (int b = 7; b >= 5; --b)
#pragma unroll
        for (uint32_t u = 0; u < (1u << (7 - b)); ++u)
            put(m8_image_offset<WAVES>(b) + (la | (u << (1 + WB))), (1u << (15 - b)) + (tile_pos >> (b + 1)) + (sa << (7 - b)) + u);
#pragma unroll
    for (int b = 4; b >= 2; --b)
#pragma unroll
        for (uint32_t u = 0; u < (1u << (4 - b)); ++u)
            put(m8_image_offset<WAVES>(b) + (lb | (u << (4 + WB))), (1u << (15 - b)) + (tile_pos >> (b + 1)) + (h << (4 - b)) + u);
#pragma unroll
    for (uint32_t g = 0; g < 2; ++g)      // bit 1: the registers above it hold bit 4 (g); bits 2, 3 come from lanes 1, 2
        put(m8_image_offset<WAVES>(1) + (t | (g << (6 + WB))), (1u << 14) + (tile_pos >> 2) + (low | (g << 2) | (h << 3)));
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u)      // bit 0: u = bit 1 | bit 4 << 1
        put(m8_image_offset<WAVES>(0) + (t | (u << (6 + WB))), (1u << 15) + (tile_pos >> 1) + ((u & 1u) | (low << 1) | ((u >> 1) << 3) | (h << 4)));
}

// a_perm8[tile][i][c][kp][lane][e] = (double) a_hat[i][c][tile * 4096 + m8_idx_c(lane, 2 kp + e)]
static __global__ void __launch_bounds__(256) m8_permute_matrix_kernel(double* __restrict__ a_perm, const uint64_t* __restrict__ a_hat, uint32_t k, int logn) {
    const uint64_t total = (uint64_t)k * k << logn;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += stride) {
        const uint32_t e = g & 1u, lane = (g >> 1) & 511u, kp = (g >> 10) & 3u;
        const uint64_t rest = g >> 12;
        const uint32_t c = rest % k, i = (rest / k) % k;
        const uint64_t tile = rest / ((uint64_t)k * k);
        const uint64_t x = (tile << 12) + m8_idx_c(lane, 2 * kp + e);
        a_perm[g] = f64_from_u52(a_hat[(((uint64_t)i * k + c) << logn) + x]);
    }
}

// LDS words of the kernel: the two images, padded for WAVES = 4 so that exactly three workgroups fit a CU (3 x 41472 B + the
// 34816 B of a cols8 workgroup <= 163840 B < 4 x 41472 B)
template <int WAVES> constexpr uint32_t m8_lds_words() { return WAVES == 8 ? 2 * m8_image_words<8>() : 41472u / 8u; }

template <int K, int WAVES>
__device__ __forceinline__ void mlwe_mid8_body(uint32_t blk, const uint64_t* __restrict__ rws, uint64_t* __restrict__ u, const double* __restrict__ a_perm,
                                               uint32_t vectors, const ModParams& p, const double* __restrict__ fwd_tw,
                                               const double* __restrict__ inv_tw, double* __restrict__ image_lds) {
    static_assert(2 * m8_image_words<WAVES>() <= m8_lds_words<WAVES>(), "twiddle images must fit");
    constexpr int WB = m8_wave_bits<WAVES>();
    const double* image = image_lds;                              // forward half first, inverse half for the second phase
    const uint32_t t = threadIdx.x;
    // (witness vector j, tile, half).  Workgroups b and b + 8 share an XCD under the observed round-robin placement (speed only):
    // an XCD sees tiles {x, x + 8}, 1 MiB of a_perm8, which stays in its L2.
    uint32_t rest = blk >> 3;
    const uint32_t tile = (blk & 7u) | ((rest & 1u) << 3);        // n = 2^16: 16 tiles per polynomial
    rest >>= 1;
    uint32_t half = 0;
    if constexpr (WAVES == 4) { half = rest & 1u; rest >>= 1; }
    const uint32_t j = rest;
    if (j >= vectors) return;
    const uint32_t T = t | (half << 8);                           // lane within the 4096-residue tile
    const uint32_t tile_pos = tile << 12;
    const uint32_t base_a = m8_idx_a(T, 0);
    const uint32_t la = m8_sel_a(t), lb = m8_sel_b(t);
    const auto tw_a = [&](int s) {
        return s == 0 ? image[m8_image_offset<WAVES>(7) + la] : (s < 3 ? image[m8_image_offset<WAVES>(6) + (la | ((uint32_t)(s - 1) << (1 + WB)))]
                                                                        : image[m8_image_offset<WAVES>(5) + (la | ((uint32_t)(s - 3) << (1 + WB)))]);
    };
    const auto tw_b = [&](int s) {
        return s == 0 ? image[m8_image_offset<WAVES>(4) + lb] : (s < 3 ? image[m8_image_offset<WAVES>(3) + (lb | ((uint32_t)(s - 1) << (4 + WB)))]
                                                                        : image[m8_image_offset<WAVES>(2) + (lb | ((uint32_t)(s - 3) << (4 + WB)))]);
    };
    double acc[K][8];
#pragma unroll
    for (int c = 0; c < K; ++c)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[c][k] = 0.0;

    m8_fill_twiddles<WAVES>(image_lds, fwd_tw, inv_tw, tile_pos, t, T);
    __syncthreads();
#pragma unroll 1
    for (int i = 0; i < K; ++i) {
        // the twiddle image does not change inside the loop: without this the compiler hoists its twenty reads out of it (40 VGPRs)
        // and spills accumulators instead
        asm volatile("" ::: "memory");
        double v[8];
        {
            const rsrc_t src = make_rsrc(rws + ((((size_t)j * K + i) << p.logn) + tile_pos), 4096u * 8u);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = __longlong_as_double((long long)buf_load64<LSR_F8_LOAD_AUX>(src, base_a * 8u, (uint32_t)k * 256u));
        }
        f8_forward_round(v, tw_a, p);                     // bits 7, 6, 5
        m8_transpose_1(v, t);
        f8_forward_round(v, tw_b, p);                     // bits 4, 3, 2
        m8_transpose_2(v, t);
        {                                                 // bits 1, 0: registers (bit 4 | bit 1 | bit 0)
            const double* const t1 = image + m8_image_offset<WAVES>(1) + t;
            const double* const t0 = image + m8_image_offset<WAVES>(0) + t;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const double w = t1[g << (6 + WB)];
                ArithF64::ct(v[4 * g], v[4 * g + 2], w, p);
                ArithF64::ct(v[4 * g + 1], v[4 * g + 3], w, p);
            }
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) ArithF64::ct(v[2 * q4], v[2 * q4 + 1], t0[q4 << (6 + WB)], p);
        }
        {   // acc[c] += A_hat[i][c] o r_hat_i: rolling window of four 16-byte matrix loads (as in mlwe_mid_fused8)
            double2 a[4];
            const rsrc_t slab = make_rsrc(a_perm + ((((size_t)tile * K + i) * K) << 12), (uint32_t)K * 4u * 512u * 16u);
            const auto fetch = [&](int c, int kp) {
                uint64_t lo, hi;
                buf_load128(slab, T * 16u, (uint32_t)(c * 4 + kp) * 8192u, lo, hi);
                return make_double2(__longlong_as_double((long long)lo), __longlong_as_double((long long)hi));
            };
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) a[kp] = fetch(0, kp);
            static_for<0, K>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
#pragma unroll
                for (int kp = 0; kp < 4; ++kp) {
                    const double2 cur = a[kp];
                    if constexpr (c + 1 < K) a[kp] = fetch(c + 1, kp);
                    acc[c][2 * kp] += mulmod_f64(v[2 * kp], cur.x, p.qd, p.inv_qd);
                    acc[c][2 * kp + 1] += mulmod_f64(v[2 * kp + 1], cur.y, p.qd, p.inv_qd);
                }
            });
        }
    }
    image = image_lds + m8_image_words<WAVES>();          // the same lambdas now read the inverse multipliers
    static_for<0, K>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        double x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = recentre_f64(acc[c][k], p.qd, p.inv_qd);
        {                                                 // bits 0, 1 (|x| <= 2 q afterwards)
            const double* const t1 = image + m8_image_offset<WAVES>(1) + t;
            const double* const t0 = image + m8_image_offset<WAVES>(0) + t;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) ArithF64::gs(x[2 * q4], x[2 * q4 + 1], t0[q4 << (6 + WB)], p);
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const double w = t1[g << (6 + WB)];
                ArithF64::gs(x[4 * g], x[4 * g + 2], w, p);
                ArithF64::gs(x[4 * g + 1], x[4 * g + 3], w, p);
            }
        }
        m8_transpose_2(x, t);
        f8_inverse_round<true>(x, tw_b, p);               // bits 2, 3, 4: 16 q -> q/2
        m8_transpose_1(x, t);
        f8_inverse_round<true>(x, tw_a, p);               // bits 5, 6, 7: 4 q -> q/2, what cols8_inverse expects
        const rsrc_t dst = make_rsrc(u + ((((size_t)j * K + c) << p.logn) + tile_pos), 4096u * 8u);
#pragma unroll
        for (int k = 0; k < 8; ++k) buf_store64<LSR_F8_STORE_AUX>(dst, base_a * 8u, (uint32_t)k * 256u, (uint64_t)__double_as_longlong(x[k]));
    });
}

template <int K>
__global__ void __launch_bounds__(512, 4) mlwe_mid8_w8(const uint64_t* __restrict__ rws, uint64_t* __restrict__ u, const double* __restrict__ a_perm,
                                                       uint32_t vectors, ModParams p, const double* __restrict__ fwd_tw, const double* __restrict__ inv_tw) {
    __shared__ double image_lds[m8_lds_words<8>()];
    mlwe_mid8_body<K, 8>(blockIdx.x, rws, u, a_perm, vectors, p, fwd_tw, inv_tw, image_lds);
}
// 136 VGPRs: three of these waves per SIMD leave 512 - 3 x 136 = 104 registers, what one wave of cols8_inverse needs
template <int K>
__global__ void __launch_bounds__(256, 3) __attribute__((amdgpu_num_vgpr(136)))
mlwe_mid8_w4(const uint64_t* __restrict__ rws, uint64_t* __restrict__ u, const double* __restrict__ a_perm, uint32_t vectors, ModParams p,
             const double* __restrict__ fwd_tw, const double* __restrict__ inv_tw) {
    __shared__ double image_lds[m8_lds_words<4>()];
    mlwe_mid8_body<K, 4>(blockIdx.x, rws, u, a_perm, vectors, p, fwd_tw, inv_tw, image_lds);
}


// Mixed launch of the 8 + 8 split (same scheme and block order as mlwe_mixed): roles = barrier-free middle stage mlwe_mid8 (bits 7..0),
// cols8_forward of the next chunk, cols8_inverse (+ e1) of the previous one — all 512-lane workgroups, 64-65 KB of LDS.
// a_perm8 layout.  `job.f_groups` is ignored (one 32-column tile per workgroup).
template <int K>
__global__ void __launch_bounds__(512, 4) mlwe_mixed88(MixedJob job, const double* __restrict__ a_perm8, ModParams p, const double* __restrict__ fwd_tw,
                                                       const double* __restrict__ inv_tw, RoundConsts<ArithF64> cs) {
    constexpr uint32_t kPool = m8_lds_words<8>() > c8_lds_words<32>() ? m8_lds_words<8>() : c8_lds_words<32>();
    __shared__ double pool[kPool];                                    // one LDS block, whichever role the workgroup plays
    const uint32_t x = blockIdx.x >> 3, sub = blockIdx.x & 7u;
    const uint32_t per = job.s_per_m + 1u, inter = job.periods * per;
    bool middle;
    uint32_t unit;
    if (x < inter) {
        const uint32_t q = x / per, rem = x - q * per;
        middle = rem == 0;
        unit = middle ? q : q * job.s_per_m + rem - 1u;
    } else {
        const uint32_t y = x - inter, left_m = job.units_m - job.periods;
        middle = y < left_m;
        unit = middle ? job.periods + y : job.periods * job.s_per_m + (y - left_m);
    }
    if (middle) {
        mlwe_mid8_body<K, 8>(unit * 8u + sub, job.m_ws, job.m_out, a_perm8, job.m_vectors, p, fwd_tw, inv_tw, pool);
        return;
    }
    const uint32_t alt = job.units_f < job.units_i ? job.units_f : job.units_i;
    bool forward;
    uint32_t idx;
    if (unit < 2u * alt) { forward = !(unit & 1u); idx = unit >> 1; }
    else { forward = job.units_f > alt; idx = alt + (unit - 2u * alt); }
    __builtin_amdgcn_s_setprio(3);
    if (forward) cols8_forward_body<false, 32>(idx * 8u + sub, job.f_dst, job.f_src, job.f_polys, p, fwd_tw, pool);
    else cols8_inverse_body<false, 32>(idx * 8u + sub, job.i_data, job.i_polys, p, inv_tw, cs, job.i_add, pool);
}

}  // namespace lsr
