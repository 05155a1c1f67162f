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
// inverse rounds of the tile pipeline: 1 = re-centre register 0 after every inner round (round 3), 0 = round 2's scheme (nothing
// after rounds 3 and 1, all eight registers after round 2)
#ifndef LSR_F8_RECENTRE_INNER
#define LSR_F8_RECENTRE_INNER 1
#endif
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
// RECENTRE = 2: bring every output back to |v| <= q/2 (the last round of a first pass: the strided round that follows runs up to
// five stages on them).  RECENTRE = 1 (inner rounds, round 3): re-centre only register 0.  Contract between rounds: every input is
// at most 3.5 q.  Then the input of the product of stage j is a difference of two sums of 2^j inputs, at most 2 * 4 * 3.5 q = 28 q
// < 2^50 at the top stage, and the all-sum output (register 0) is at most 8 * 3.5 q = 28 q; a product output is at most 0.875 q
// (=: P) and doubles with every later sum stage: registers 4..7 P, 2..3 2 P, register 1 4 P = 3.5 q — only register 0 can exceed
// the bound.  (Round 2 re-centred all eight registers every second round: 24 instead of 9 instructions per output and lane.)
template <int RECENTRE, class TW>
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
    if constexpr (RECENTRE == 2) {
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) v[k] = recentre_f64(v[k], p.qd, p.inv_qd);
    } else if constexpr (RECENTRE == 1) {
        v[0] = recentre_f64(v[0], p.qd, p.inv_qd);
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

// (re)build the LDS twiddle image for one direction; the caller separates it from its readers with barriers.
// IN_REGS: the lane's seven multipliers of the last round stay in `mine` (14 VGPRs) instead of 28 KB of LDS — for pipelines with
// registers to spare and an occupancy that the LDS footprint limits (openings at n = 4096: three workgroups per CU instead of two)
template <bool IN_REGS = false>
__device__ __forceinline__ void f8_fill_twiddles(double* __restrict__ tw_lds, const double* __restrict__ table, uint32_t tile_pos, int logn, uint32_t t,
                                                 double (&mine)[7]) {
    if (t < kF8TwShared) {
        const int b = t < 64 ? 5 : (t < 192 ? 4 : 3);
        const uint32_t e = t < 64 ? t : (t < 192 ? t - 64 : t - 192);
        tw_lds[t] = table[(1u << (logn - 1 - b)) + (tile_pos >> (b + 1)) + e];
    }
    const uint32_t pos0 = tile_pos + f8_base<3>(t);
    f8_uniform_twiddles<3>(mine, table, pos0, logn);
    if constexpr (!IN_REGS) {
#pragma unroll
        for (int s = 0; s < 7; ++s) tw_lds[kF8TwShared + s * kF8Threads + t] = mine[s];
    }
}

// last inverse round of a transform whose top index bit lies in this tile (n = 4096: the tile IS the polynomial): the stage of
// bit 11 folds n^-1 into both outputs (SEAL's scalar path; RoundConsts), so the outputs are fresh products, |x| < q
template <class TW>
__device__ __forceinline__ void f8_inverse_round_final(double (&v)[kF8Regs], TW&& tw, const RoundConsts<ArithF64>& cs, const ModParams& p) {
#pragma unroll
    for (int u = 0; u < 4; ++u) ArithF64::gs(v[2 * u], v[2 * u + 1], tw(3 + u), p);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const double w = tw(1 + u);
#pragma unroll
        for (int l = 0; l < 2; ++l) ArithF64::gs(v[4 * u + l], v[4 * u + l + 2], w, p);
    }
#pragma unroll
    for (int l = 0; l < 4; ++l) ArithF64::gs_scaled(v[l], v[l + 4], cs.w_last_scaled, cs.n_inv, p);
}

// Lane-major copy of the matrix a tile pipeline multiplies by (16 B per lane, coalesced):
//   out[tile][i][c][kp][lane][e] = (double) column(c0 + c)[i][tile * 4096 + f8_base<3>(lane) + 2 kp + e],   c < nc
// where column x < k is A_hat[i][x] (the commitment's A^T product) and column k is `extra`[i] (b_hat for the scalar component of
// a full commitment, s_hat for an opening).
static __global__ void __launch_bounds__(256) f8_permute_matrix_kernel(double* __restrict__ out, const uint64_t* __restrict__ a_hat,
                                                                         const uint64_t* __restrict__ extra, uint32_t k, uint32_t c0, uint32_t nc, int logn) {
    const uint64_t total = (uint64_t)k * nc << logn;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += stride) {
        const uint32_t e = g & 1u, lane = (g >> 1) & 511u, kp = (g >> 10) & 3u;
        const uint64_t rest = g >> 12;                       // (tile * k + i) * nc + c
        const uint32_t c = rest % nc, i = (rest / nc) % k;
        const uint64_t tile = rest / ((uint64_t)k * nc);
        const uint64_t x = (tile << 12) + f8_base<3>(lane) + 2 * kp + e;
        const uint32_t col = c0 + c;
        const uint64_t word = col < k ? a_hat[(((uint64_t)i * k + col) << logn) + x] : extra[((uint64_t)i << logn) + x];
        out[g] = f64_from_u52(word);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The tile pipeline: K polynomials in, NC linear combinations out, nothing but the ends touches memory.
//   forward: for i < K: src.load(i) -> the 12 low forward stages -> acc[c] += M[i][c] o x_hat_i
//   inverse: for c < NC: the 12 low inverse stages of acc[c] -> sink.store(c)
// `mat` is this tile's slice of a lane-major matrix copy (f8_permute_matrix_kernel): [i][c][kp][lane][2].
// FULL: the tile is the whole polynomial (n = 4096), so the last inverse round carries the n^-1-scaled final stage.
// Src:  load(i, v)  — polynomial i in the layout of round 0 (register k of lane t = tile index t + 512 k), as doubles;
//       ahead(i)    — optional hint issued three rounds before load(i + 1).
// Sink: store(c, x) — output c in the same layout (raw elements, |x| <= q/2 + 1; FULL: |x| < q, fresh products).
// Both may use workgroup barriers (every lane calls them, in the same order).
// ---------------------------------------------------------------------------------------------------------------------------------
// TW3_REGS: see f8_fill_twiddles (tw_lds then has kF8TwShared entries only).
template <int K, int NC, bool FULL, class Src, class Sink, bool TW3_REGS = false>
__device__ __forceinline__ void f8_tile_pipeline(uint32_t tile_pos, Src& src, Sink& sink, const double* __restrict__ mat, const ModParams& p,
                                                 const double* __restrict__ fwd_tw, const double* __restrict__ inv_tw, const RoundConsts<ArithF64>& cs,
                                                 double* __restrict__ tile_lds, double* __restrict__ tw_lds) {
    const uint32_t t = threadIdx.x;
    const uint32_t wave0 = __builtin_amdgcn_readfirstlane(t & ~63u);
    // per-round LDS addresses of register 0 (doubles)
    double* const row0 = tile_lds + f8_slot(f8_base<0>(t));
    double* const row1 = tile_lds + f8_slot(f8_base<1>(t));
    double* const row2 = tile_lds + f8_slot(f8_base<2>(t));
    [[maybe_unused]] double* const row3 = tile_lds + f8_slot(f8_base<3>(t));
    const double* const tw2 = tw_lds;                                   // natural-order sub-tables of bits 5, 4, 3
    const double* const tw3 = tw_lds + kF8TwShared + t;                 // this lane's 7 multipliers, stride 512
    const uint32_t e5 = f8_base<2>(t) >> 6, e4 = f8_base<2>(t) >> 5, e3 = f8_base<2>(t) >> 4;

    double acc[NC][kF8Regs];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) acc[c][k] = 0.0;

    double mine3[7];
    f8_fill_twiddles<TW3_REGS>(tw_lds, fwd_tw, tile_pos, p.logn, t, mine3);
    double w0[7], w1[7];
    f8_uniform_twiddles<0>(w0, fwd_tw, tile_pos, p.logn);
    f8_uniform_twiddles<1>(w1, fwd_tw, tile_pos + f8_base<1>(wave0), p.logn);
    __syncthreads();

    // ---- forward: the K polynomials of the vector, one after the other ---------------------------------------------------
    const auto tw_r0 = [&](int s_) { return w0[s_]; };
    const auto tw_r1 = [&](int s_) { return w1[s_]; };
    const auto tw_r2 = [&](int s_) { return s_ == 0 ? tw2[e5] : (s_ < 3 ? tw2[64 + e4 + (s_ - 1)] : tw2[192 + e3 + (s_ - 3)]); };
    const auto tw_r3 = [&](int s_) {
        if constexpr (TW3_REGS) return mine3[s_];
        else return tw3[s_ * kF8Threads];
    };
#pragma unroll 1
    for (int i = 0; i < K; ++i) {
        double v[kF8Regs];
        src.load(i, v);
        f8_forward_round(v, tw_r0, p);
        if (i > 0) __syncthreads();                      // the previous polynomial's last LDS reads are done
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) row0[f8_slot((uint32_t)k << 9)] = v[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) v[k] = row1[f8_slot((uint32_t)k << 6)];
        const uint32_t hint = src.ahead(i);
        f8_forward_round(v, tw_r1, p);
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) row1[f8_slot((uint32_t)k << 6)] = v[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) v[k] = row2[f8_slot((uint32_t)k << 3)];
        f8_forward_round(v, tw_r2, p);
#if LSR_F8_MAT_AHEAD
        double2 a[4];
        const rsrc_t slab = make_rsrc(mat + (((size_t)i * NC) << 12), (uint32_t)NC * 4u * 512u * 16u);
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
        // acc[c] += M[i][c] o x_hat_i on this lane's 8 positions
        [[maybe_unused]] const double* const mat_i = mat + (((size_t)i * NC) << 12) + (size_t)t * 2;
#if LSR_F8_MAT_AHEAD
        {   // rolling window of four 16-byte matrix loads: each slot is refilled for the next component as soon as its two
            // products are issued, so a wave waits for L2 once per polynomial instead of once per load
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) a[kp] = fetch(0, kp);
            static_for<0, NC>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
#pragma unroll
                for (int kp = 0; kp < 4; ++kp) {
                    const double2 cur = a[kp];
                    if constexpr (c + 1 < NC) a[kp] = fetch(c + 1, kp);
                    acc[c][2 * kp] += mulmod_f64(v[2 * kp], cur.x, p.qd, p.inv_qd);
                    acc[c][2 * kp + 1] += mulmod_f64(v[2 * kp + 1], cur.y, p.qd, p.inv_qd);
                }
            });
        }
#else
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) {
                const double2 a = *reinterpret_cast<const double2*>(mat_i + (((size_t)c * 4 + kp) << 10));
                acc[c][2 * kp] += mulmod_f64(v[2 * kp], a.x, p.qd, p.inv_qd);
                acc[c][2 * kp + 1] += mulmod_f64(v[2 * kp + 1], a.y, p.qd, p.inv_qd);
            }
        }
#endif
        asm volatile("" ::"v"(hint));
    }

    // ---- inverse: each output component through the tile ------------------------------------------------------------------
    __syncthreads();                                     // every lane is done with the forward twiddle image
    f8_fill_twiddles<TW3_REGS>(tw_lds, inv_tw, tile_pos, p.logn, t, mine3);
    f8_uniform_twiddles<0>(w0, inv_tw, tile_pos, p.logn);
    f8_uniform_twiddles<1>(w1, inv_tw, tile_pos + f8_base<1>(wave0), p.logn);
    __syncthreads();
    static_for<0, NC>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        double x[kF8Regs];
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) x[k] = recentre_f64(acc[c][k], p.qd, p.inv_qd);
        f8_inverse_round<LSR_F8_RECENTRE_INNER>(x, tw_r3, p);             // inputs <= q/2; outputs <= 3.5 q
#if LSR_F8_WAVE_XCHG
        f8_transpose_regs_lanes(x, t);
        f8_inverse_round<LSR_F8_RECENTRE_INNER == 1 ? 1 : 2>(x, tw_r2, p);
        if constexpr (c > 0) __syncthreads();            // the previous component's last LDS reads are done
#else
        if constexpr (c > 0) __syncthreads();            // the previous component's last LDS reads are done
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) row3[k] = x[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) x[k] = row2[f8_slot((uint32_t)k << 3)];
        f8_inverse_round<LSR_F8_RECENTRE_INNER == 1 ? 1 : 2>(x, tw_r2, p);
#endif
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) row2[f8_slot((uint32_t)k << 3)] = x[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) x[k] = row1[f8_slot((uint32_t)k << 6)];
        f8_inverse_round<LSR_F8_RECENTRE_INNER>(x, tw_r1, p);
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) row1[f8_slot((uint32_t)k << 6)] = x[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) x[k] = row0[f8_slot((uint32_t)k << 9)];
        if constexpr (FULL) f8_inverse_round_final(x, tw_r0, cs, p);   // sums up to 28 q enter the n^-1 products: |x| < q
        else f8_inverse_round<2>(x, tw_r0, p);            // -> q/2: what the strided round expects
        sink.store(c, x);
    });
}

// (witness vector j, tile) of workgroup `blk`.  Workgroups b and b + 8 share an XCD under the observed round-robin placement (speed
// only): an XCD then sees 2 of the >= 16 tile positions of the matrix, which stay in its L2.  n = 4096: one tile, j = blk.
__device__ __forceinline__ void f8_block_position(uint32_t blk, int logn, uint32_t* j, uint32_t* tile) {
    const int tp_log = logn - 12;
    if (tp_log < 3) { *tile = blk & ((1u << tp_log) - 1u); *j = blk >> tp_log; return; }
    const uint32_t rest = blk >> 3;
    *tile = (blk & 7u) | ((rest & ((1u << (tp_log - 3)) - 1u)) << 3);
    *j = rest >> (tp_log - 3);
}

// polynomials of witness vector j as raw f64 elements (the strided round's output): polynomial i at base + (i << logn)
struct F8RawSource {
    const uint64_t* base;
    int logn;
    uint32_t tile_pos, polys;
    __device__ __forceinline__ void load(int i, double (&v)[kF8Regs]) const {
        const rsrc_t src = make_rsrc(base + (((size_t)i << logn) + tile_pos), 4096u * 8u);
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) v[k] = __longlong_as_double((long long)buf_load64<LSR_F8_LOAD_AUX>(src, threadIdx.x * 8u, (uint32_t)k * 4096u));
    }
    // Pull the next polynomial's tile towards this XCD's L2 three rounds ahead of its use: one dword per 128-byte line
    // (a real prefetch into registers would need 16 more VGPRs than the 128 that two workgroups per CU allow).
    __device__ __forceinline__ uint32_t ahead(int i) const {
#if LSR_F8_TOUCH
        if (i + 1 < (int)polys && threadIdx.x < 256) {
            const rsrc_t nxt = make_rsrc(base + (((size_t)(i + 1) << logn) + tile_pos), 4096u * 8u);
            return __builtin_amdgcn_raw_buffer_load_b32(nxt, (int)(threadIdx.x * 128u), 0, 0);
        }
#endif
        return 0;
    }
};
// component c as raw f64 elements at base + (c << logn): the inverse strided round's input
struct F8RawSink {
    uint64_t* base;
    int logn;
    uint32_t tile_pos;
    __device__ __forceinline__ void store(int c, const double (&x)[kF8Regs]) const {
        const rsrc_t dst = make_rsrc(base + (((size_t)c << logn) + tile_pos), 4096u * 8u);
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) buf_store64<LSR_F8_STORE_AUX>(dst, threadIdx.x * 8u, (uint32_t)k * 4096u, (uint64_t)__double_as_longlong(x[k]));
    }
};

// Middle stage of the n = 2^16 / 2^17 pipelines: witness vector j's K polynomials at rws + j in_pitch (raw, after the forward strided
// round) -> NC components at out + j out_pitch (raw, for the inverse strided round); mat = [tile][K][NC] lane-major slices.
// blk: the workgroup's position in the (witness vector, tile) grid — the hardware block index when the stage is a kernel of its
// own, a number handed out by mlwe_mixed (below) when it is one role of a mixed launch
template <int K, int NC>
__device__ __forceinline__ void mlwe_mid_body(uint32_t blk, const uint64_t* __restrict__ rws, size_t in_pitch, uint64_t* __restrict__ out, size_t out_pitch,
                                              const double* __restrict__ mat, uint32_t vectors, const ModParams& p, const double* __restrict__ fwd_tw,
                                              const double* __restrict__ inv_tw) {
    __shared__ double tile_lds[kF8TileWords];
    __shared__ double tw_lds[kF8TwShared + kF8TwPrivate];
    uint32_t j, tile;
    f8_block_position(blk, p.logn, &j, &tile);
    if (j >= vectors) return;
    const uint32_t tile_pos = tile << 12;
    F8RawSource src{rws + (size_t)j * in_pitch, p.logn, tile_pos, (uint32_t)K};
    F8RawSink sink{out + (size_t)j * out_pitch, p.logn, tile_pos};
    f8_tile_pipeline<K, NC, false>(tile_pos, src, sink, mat + (((size_t)tile * K * NC) << 12), p, fwd_tw, inv_tw, RoundConsts<ArithF64>{}, tile_lds, tw_lds);
}
template <int K>
__device__ __forceinline__ void mlwe_mid_fused8_body(uint32_t blk, const uint64_t* __restrict__ rws, uint64_t* __restrict__ u,
                                                     const double* __restrict__ a_perm, uint32_t vectors, const ModParams& p,
                                                     const double* __restrict__ fwd_tw, const double* __restrict__ inv_tw) {
    mlwe_mid_body<K, K>(blk, rws, (size_t)K << p.logn, u, (size_t)K << p.logn, a_perm, vectors, p, fwd_tw, inv_tw);
}

template <int K>
__global__ void __launch_bounds__(kF8Threads, 4) mlwe_mid_fused8(const uint64_t* __restrict__ rws, uint64_t* __restrict__ u,
                                                                  const double* __restrict__ a_perm, uint32_t vectors, ModParams p,
                                                                  const double* __restrict__ fwd_tw, const double* __restrict__ inv_tw) {
    mlwe_mid_fused8_body<K>(blockIdx.x, rws, u, a_perm, vectors, p, fwd_tw, inv_tw);
}
// the general form: NC components out of K polynomials, explicit pitches (full commitments write into wire rows; openings read one)
template <int K, int NC>
__global__ void __launch_bounds__(kF8Threads, 4) mlwe_mid_general(const uint64_t* __restrict__ rws, size_t in_pitch, uint64_t* __restrict__ out,
                                                                   size_t out_pitch, const double* __restrict__ mat, uint32_t vectors, ModParams p,
                                                                   const double* __restrict__ fwd_tw, const double* __restrict__ inv_tw) {
    mlwe_mid_body<K, NC>(blockIdx.x, rws, in_pitch, out, out_pitch, mat, vectors, p, fwd_tw, inv_tw);
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

}  // namespace lsr
