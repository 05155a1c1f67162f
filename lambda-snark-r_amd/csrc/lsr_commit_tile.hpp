// Whole commitments and whole openings at the reference's ring degree, ONE workgroup each (round 3).
//
// Every reference caller commits with n = 4096, k = 2; a 4096-residue polynomial is exactly one tile of the fused
// pipeline (lsr_commit_fused.hpp), so lwe_commit (cpp-core/src/commitment.cpp:138-164, contract commitment.h:43-52) is one launch:
//     for i < k : r_i <- chi (ChaCha20 stream + CDT search, in the lanes) -> 12 forward stages in LDS
//                 acc[c] += A_hat[i][c] o r_hat_i (c < k),  acc[k] += b_hat[i] o r_hat_i
//     for c <= k: 12 inverse stages (n^-1 folded into the last) -> + e1_c | + e2 + round(q (m mod t) / t) -> the wire row, canonical
// instead of the eight launches of the unfused path (three samplers writing r, e1, e2 to memory, forward transforms, two matrix
// products, inverse transforms, finish, pack): nothing but the finished row is written, nothing but the message is read.
// lwe_verify_opening (commitment.cpp:200-232) is the same pipeline with the row's u as the source, s_hat as the one-column
// matrix and a decode-and-compare sink: one launch reads the row once.
//
// FP64 flavour (q < 2^45), k <= 4, CDT tables of <= 64 entries (sigma <= ~6.9); other contexts use the general kernels.
#pragma once

#include "lsr_commit_fused.hpp"

namespace lsr {

constexpr uint32_t kRowHeaderWords = 5;                           // data[0] + 4 header words (lsr_commit.hip: the wire format)
constexpr uint64_t kRowMagic = 0x313030304352534CULL;             // "LSRC0001"

// floor((hi:lo) / q) for a dividend below 2^61 * 2^21 (quotient fits 64 bits), by Barrett + fix-up
__device__ __forceinline__ uint64_t div128_by_q(uint64_t hi, uint64_t lo, const ModParams& p) {
    const uint64_t c1 = __umul64hi(lo, p.barrett_lo);
    const uint64_t m1_lo = lo * p.barrett_hi, m1_hi = __umul64hi(lo, p.barrett_hi);
    const uint64_t m2_lo = hi * p.barrett_lo, m2_hi = __umul64hi(hi, p.barrett_lo);
    uint64_t s = c1 + m1_lo;
    uint64_t carry = s < c1;
    const uint64_t s2 = s + m2_lo;
    carry += s2 < s;
    uint64_t quot = hi * p.barrett_hi + m1_hi + m2_hi + carry;
    uint64_t rem = lo - quot * p.q;
    // the estimate drops at most three partial carries: three branch-free corrections (no data-dependent trip count: the dividend
    // is t (v - <s, u>), and the opening check is meant to be constant-time, commitment.h:92)
#pragma unroll
    for (int fix = 0; fix < 3; ++fix) {
        const bool over = rem >= p.q;
        rem -= over ? p.q : 0;
        quot += over ? 1 : 0;
    }
    return quot;
}
// round(t w / q) mod t for a canonical residue w: the plaintext slot an opening decodes (commitment.cpp:215-218 via SEAL decrypt).
// w <= q - 1 gives a quotient <= t, so the reduction mod t is one select.
__device__ __forceinline__ uint64_t decode_slot(uint64_t w, uint64_t t, const ModParams& p) {
    const uint64_t lo0 = w * t, hi0 = __umul64hi(w, t);
    const uint64_t lo = lo0 + (p.q >> 1);
    const uint64_t hi = hi0 + (lo < lo0);
    const uint64_t d = div128_by_q(hi, lo, p);
    return d == t ? 0 : d;
}

// The message term of the scalar component: round(q (m mod t) / t) = floor((q m' + t/2) / t) for any 64-bit word m and t < 2^21,
// in FP64 (a 64-bit integer division costs hundreds of instructions; the embed sits in the same lanes as the transforms).
// Why not floor(q/t) m': with q = Delta t + rho a sum of commitments carries Delta M for the INTEGER M = sum c_i m_i, and
// Delta M = Delta (M mod t) - rho floor(M/t) (mod q): the second term is noise of size rho sum c_i, five times the sampled noise at the
// default modulus, which made combinations of large messages undecodable inside the noise budget.  With the rounded scaling the
// sum is (q/t)(M mod t) + (at most sum c_i / 2) (mod q) (SEAL's BFV encryptor scales the same way for the same reason).
// Every step reduces a value below 2^53 with one rounded quotient, an exact FMA remainder and a fix-up of at most one t either way.
struct PlainScale {
    double t, inv_t, delta, rho, half;          // t, 1/t, floor(q/t), q mod t, floor(t/2)
};
__host__ __device__ inline PlainScale make_plain_scale(uint64_t q, uint64_t t) {
    return PlainScale{(double)t, 1.0 / (double)t, (double)(q / t), (double)(q % t), (double)(t >> 1)};
}
// x = k t + r with 0 <= r < t for an integer 0 <= x < 2^53: returns r, *quot = k
__device__ __forceinline__ double divmod_below_2p53(double x, const PlainScale& m, double* quot) {
    double k = __builtin_floor(x * m.inv_t);
    double r = __builtin_fma(-k, m.t, x);                 // |x - k t| < 2 t: exact
    if (r < 0.0) { r += m.t; k -= 1.0; }
    if (r >= m.t) { r -= m.t; k += 1.0; }
    *quot = k;
    return r;
}
__device__ __forceinline__ double mod_plain(uint64_t word, const PlainScale& m) {
    double k;
    const double hi = divmod_below_2p53((double)(uint32_t)(word >> 32), m, &k);
    return divmod_below_2p53(hi * 4294967296.0 + (double)(uint32_t)word, m, &k);    // < 2^21 2^32 + 2^32 < 2^53
}
// round(q (word mod t) / t) = Delta m' + floor((rho m' + t/2) / t): both terms and their sum (< q < 2^45) are exact in FP64
__device__ __forceinline__ double embed_plain(uint64_t word, const PlainScale& m) {
    const double mm = mod_plain(word, m);
    double k;
    (void)divmod_below_2p53(m.rho * mm + m.half, m, &k);                             // rho m' + t/2 < 2^42
    return m.delta * mm + k;
}

// One stream block (eight consecutive coefficients, block number = lane) of the Gaussian object (key, domain, index), left in `stage`
// as signed 16-bit values in coefficient order.  Every lane of the workgroup takes part (lane table).
__device__ __forceinline__ void f8_sample_to_stage(const uint64_t* __restrict__ key, uint32_t domain, uint32_t index, const LaneTable& tab, uint32_t entries,
                                                   int16_t* __restrict__ stage) {
    uint64_t w[8], u[8];
    stream_block(key, domain, index, threadIdx.x, w);
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i] = w[i] >> 1;
    uint32_t magnitude[8];
    cdt_magnitudes(tab, nullptr, entries, u, magnitude);
    uint32_t packed[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t a = magnitude[2 * i], b = magnitude[2 * i + 1];
        const uint32_t sa = (w[2 * i] & 1ull) ? 0u - a : a, sb = (w[2 * i + 1] & 1ull) ? 0u - b : b;
        packed[i] = (sa & 0xFFFFu) | (sb << 16);
    }
    *reinterpret_cast<uint4*>(stage + 8 * threadIdx.x) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
}

struct CommitTileJob {
    uint64_t* rows;              // [batch][5 + (k + 1) n] wire rows, device
    const uint64_t* keys;        // [batch][4] per-commitment stream keys
    const uint64_t* msgs;        // [batch][msg_len]; only read when copy > 0
    uint64_t msg_len, copy;      // copy = min(msg_len, n) slots are embedded (commitment.cpp:146-149)
    const uint64_t* cdf;         // CDT table (64-bit thresholds)
    uint32_t entries;            // scanned entries (<= 64)
    uint32_t batch;
    uint64_t q, t;
};

// r_i sampled where the forward transform wants it
struct CommitTileSource {
    const uint64_t* key;
    LaneTable tab;
    uint32_t entries;
    int16_t* stage;
    __device__ __forceinline__ void load(int i, double (&v)[kF8Regs]) const {
        f8_sample_to_stage(key, kDomR, (uint32_t)i, tab, entries, stage);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) v[k] = (double)stage[threadIdx.x + 512u * (uint32_t)k];
        // the stage is rewritten by the next load / store only after the barriers of the rounds in between
    }
    __device__ __forceinline__ uint32_t ahead(int) const { return 0; }
};
// u_c = . + e1_c,  v = . + e2 + round(q (m mod t) / t): canonical words into the wire row
template <int K>
struct CommitTileSink {
    const CommitTileJob& job;
    const uint64_t* key;
    LaneTable tab;
    int16_t* stage;
    uint64_t* row;               // this commitment's row
    const uint64_t* msg;         // this commitment's message
    const ModParams& p;
    __device__ __forceinline__ void store(int c, const double (&x)[kF8Regs]) const {
        f8_sample_to_stage(key, c < K ? kDomE1 : kDomE2, c < K ? (uint32_t)c : 0u, tab, job.entries, stage);
        __syncthreads();
        uint64_t* const dst = row + kRowHeaderWords + ((size_t)c << 12);
        const PlainScale ps = make_plain_scale(job.q, job.t);
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) {
            const uint32_t idx = threadIdx.x + 512u * (uint32_t)k;
            double v = x[k] + (double)stage[idx];
            if (c == K && idx < job.copy) v += embed_plain(msg[idx], ps);           // < q: exact
            dst[idx] = u52_from_f64(canonical_f64(v, p.qd, p.inv_qd));              // |v| < 2 q + 2^15
        }
    }
};

template <int K>
__global__ void __launch_bounds__(kF8Threads, 4) commit_tile_kernel(CommitTileJob job, const double* __restrict__ ab_perm, ModParams p,
                                                                     const double* __restrict__ fwd_tw, const double* __restrict__ inv_tw,
                                                                     RoundConsts<ArithF64> cs) {
    __shared__ double tile_lds[kF8TileWords];
    __shared__ double tw_lds[kF8TwShared + kF8TwPrivate];
    __shared__ __attribute__((aligned(16))) int16_t stage[4096];
    const uint32_t j = blockIdx.x;
    if (j >= job.batch) return;
    const uint64_t* const key = job.keys + 4 * (size_t)j;
    const LaneTable tab = lane_table_load(job.cdf, job.entries);
    const size_t row_words = kRowHeaderWords + ((size_t)(K + 1) << 12);
    uint64_t* const row = job.rows + (size_t)j * row_words;
    if (threadIdx.x < kRowHeaderWords) {
        const uint32_t w = threadIdx.x;
        row[w] = w == 0 ? 8ull * (row_words - 1) : (w == 1 ? kRowMagic : (w == 2 ? (4096ull | ((uint64_t)K << 32)) : (w == 3 ? job.q : job.t)));
    }
    CommitTileSource src{key, tab, job.entries, stage};
    CommitTileSink<K> sink{job, key, tab, stage, row, job.msgs + (size_t)j * job.msg_len, p};
    f8_tile_pipeline<K, K + 1, true>(0u, src, sink, ab_perm, p, fwd_tw, inv_tw, cs, tile_lds, tw_lds);
}

// ---- openings -------------------------------------------------------------------------------------------------------------------
struct VerifyTileJob {
    const uint64_t* rows;        // [count][5 + (k + 1) n] wire rows, device
    const uint64_t* msgs;        // [count][msg_len] claimed messages (raw words, commitment.cpp:223-226)
    uint64_t msg_len;            // 1 .. n
    unsigned long long* flags;   // [count]: OR over the slots of decoded ^ claimed
    uint32_t* bad;               // [count]: != 0 when the row is not a canonical commitment of this context
    uint32_t count;
    uint64_t q, t;
};
// u_i of the row: canonical words -> elements; a word >= q or a wrong header marks the row
template <int K>
struct VerifyTileSource {
    const uint64_t* row;
    uint32_t* bad;
    uint64_t q;
    __device__ __forceinline__ void load(int i, double (&v)[kF8Regs]) const {
        const uint64_t* const src = row + kRowHeaderWords + ((size_t)i << 12);
        bool ok = true;
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) {
            const uint64_t raw = src[threadIdx.x + 512u * (uint32_t)k];
            ok = ok && raw < q;
            v[k] = f64_from_u52(raw);
        }
        if (!ok) atomicOr(bad, 1u);
    }
    // pull the next polynomial of the row (after the last u_i: v, which the sink reads) towards this XCD's L2 while the current one
    // is being transformed: one dword per 128-byte line, as the middle stage of the large degrees does
    __device__ __forceinline__ uint32_t ahead(int i) const {
        if (threadIdx.x < 256) {
            const rsrc_t nxt = make_rsrc(row + kRowHeaderWords + ((size_t)(i + 1) << 12), 4096u * 8u);
            return __builtin_amdgcn_raw_buffer_load_b32(nxt, (int)(threadIdx.x * 128u), 0, 0);
        }
        return 0;
    }
};
// w = v - INTT(<s_hat, u_hat>), decoded slot by slot and compared with the claimed words as given
template <int K>
struct VerifyTileSink {
    const VerifyTileJob& job;
    const uint64_t* row;
    const uint64_t* msg;
    unsigned long long* flag;
    uint32_t* bad;
    const ModParams& p;
    __device__ __forceinline__ void store(int, const double (&x)[kF8Regs]) const {
        const uint64_t* const vsrc = row + kRowHeaderWords + ((size_t)K << 12);
        uint64_t diff = 0;
        bool ok = true;
#pragma unroll
        for (int k = 0; k < kF8Regs; ++k) {
            const uint32_t idx = threadIdx.x + 512u * (uint32_t)k;
            const uint64_t raw = vsrc[idx];
            ok = ok && raw < job.q;
            if (idx < job.msg_len) {
                const uint64_t w = u52_from_f64(canonical_f64(f64_from_u52(raw) - x[k], p.qd, p.inv_qd));
                diff |= decode_slot(w, job.t, p) ^ msg[idx];
            }
        }
        if (!ok) atomicOr(bad, 1u);
        if (diff) atomicOr(flag, (unsigned long long)diff);
    }
};

// (tools/verify_tile_sweep.sh, 16384 openings: multipliers in LDS, 4 waves/SIMD 0.594 / 0.935 / 0.392 ms at k = 2 / 4 / 1; in registers, 4 waves
// 0.556 / 0.929 / 0.366 ms; in registers, 6 waves 0.560 / 0.897 / 0.378 ms — the gain is the LDS traffic saved, not the third workgroup)
#ifndef LSR_VERIFY_TILE_WAVES
#define LSR_VERIFY_TILE_WAVES 4       // wavefronts per SIMD the kernel is compiled for (6 = three workgroups per CU: 80 VGPRs with two of them
                                      // spilled, 15 % more traffic per opening for the same speed at k = 2 — PMC, profiles/README.md)
#endif
#ifndef LSR_VERIFY_TILE_TW_REGS
#define LSR_VERIFY_TILE_TW_REGS 1     // the last round's multipliers in registers: 40 KB of LDS per workgroup instead of 69 KB
#endif
template <int K>
__global__ void __launch_bounds__(kF8Threads, LSR_VERIFY_TILE_WAVES) verify_tile_kernel(VerifyTileJob job, const double* __restrict__ s_perm, ModParams p,
                                                                                         const double* __restrict__ fwd_tw, const double* __restrict__ inv_tw,
                                                                                         RoundConsts<ArithF64> cs) {
    __shared__ double tile_lds[kF8TileWords];
    __shared__ double tw_lds[kF8TwShared + (LSR_VERIFY_TILE_TW_REGS ? 0 : kF8TwPrivate)];
    const uint32_t j = blockIdx.x;
    if (j >= job.count) return;
    const size_t row_words = kRowHeaderWords + ((size_t)(K + 1) << 12);
    const uint64_t* const row = job.rows + (size_t)j * row_words;
    if (threadIdx.x < kRowHeaderWords) {
        const uint32_t w = threadIdx.x;
        const uint64_t want = w == 0 ? 8ull * (row_words - 1) : (w == 1 ? kRowMagic : (w == 2 ? (4096ull | ((uint64_t)K << 32)) : (w == 3 ? job.q : job.t)));
        if (row[w] != want) atomicOr(&job.bad[j], 1u);
    }
    VerifyTileSource<K> src{row, &job.bad[j], job.q};
    VerifyTileSink<K> sink{job, row, job.msgs + (size_t)j * job.msg_len, &job.flags[j], &job.bad[j], p};
    f8_tile_pipeline<K, 1, true, VerifyTileSource<K>, VerifyTileSink<K>, LSR_VERIFY_TILE_TW_REGS != 0>(0u, src, sink, s_perm, p, fwd_tw, inv_tw, cs, tile_lds, tw_lds);
}

// =================================================================================================================================
// The same two operations at n = 2^16 / 2^17 (a polynomial is 16 / 32 tiles): three launches per chunk of commitments,
//     commit_top_forward  — r_i SAMPLED where the top forward round wants its operands (no array of r exists), raw elements out
//     mlwe_mid_general    — tile pipeline: 12 forward stages, [A^T | b_hat] product, 12 inverse stages, into the wire rows (raw)
//     commit_top_inverse  — top inverse round in place on the rows, + e1_c / + e2 + round(q (m mod t) / t), the blinding sampled in the pass, canonical
// (rank 4: the scalar component is a second, one-column pass of the middle stage — five accumulators do not fit its 128 VGPRs).
// A workgroup of the outer rounds owns 256 columns x 2^R rows of one polynomial = 2^R * 32 stream blocks of 8 consecutive
// coefficients, 2^R / 8 per lane; the samples change hands through an int16 tile in LDS.
// =================================================================================================================================
template <int R>
__device__ __forceinline__ void top_round_sample(int16_t* __restrict__ tile, const uint64_t* __restrict__ key, uint32_t domain, uint32_t index, int lo,
                                                 uint32_t low0, const LaneTable& tab, uint32_t entries) {
    constexpr int N = 1 << R;
#pragma unroll 1
    for (int h = 0; h < N / 8; ++h) {
        const uint32_t b = (uint32_t)h * 256u + threadIdx.x, row = b >> 5, cb = b & 31u;
        uint64_t w[8], u[8];
        stream_block(key, domain, index, (((row << lo) + low0) >> 3) + cb, w);
#pragma unroll
        for (int i = 0; i < 8; ++i) u[i] = w[i] >> 1;
        uint32_t magnitude[8];
        cdt_magnitudes(tab, nullptr, entries, u, magnitude);
        uint32_t packed[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t a = magnitude[2 * i], bb = magnitude[2 * i + 1];
            const uint32_t sa = (w[2 * i] & 1ull) ? 0u - a : a, sb = (w[2 * i + 1] & 1ull) ? 0u - bb : bb;
            packed[i] = (sa & 0xFFFFu) | (sb << 16);
        }
        *reinterpret_cast<uint4*>(tile + row * 256u + cb * 8u) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
    }
}
// the top R forward stages on 2^R registers (table entries 1 .. 2^R - 1: the same for every lane of every polynomial)
template <int R>
__device__ __forceinline__ void top_round_forward(double (&v)[1 << R], const double* __restrict__ tw, const ModParams& p) {
#pragma unroll
    for (int j = R - 1; j >= 0; --j) {
        const int half = 1 << j;
#pragma unroll
        for (int u = 0; u < (1 << (R - 1 - j)); ++u) {
            const double w = tw[(1 << (R - 1 - j)) + u];
#pragma unroll
            for (int l = 0; l < half; ++l) ArithF64::ct(v[(u << (j + 1)) | l], v[((u << (j + 1)) | l) + half], w, p);
        }
    }
}
// the top R inverse stages, n^-1 folded into the last: outputs are fresh products, |v| < q
template <int R>
__device__ __forceinline__ void top_round_inverse(double (&v)[1 << R], const double* __restrict__ tw, const RoundConsts<ArithF64>& cs, const ModParams& p) {
#pragma unroll
    for (int j = 0; j < R - 1; ++j) {
        const int half = 1 << j;
#pragma unroll
        for (int u = 0; u < (1 << (R - 1 - j)); ++u) {
            const double w = tw[(1 << (R - 1 - j)) + u];
#pragma unroll
            for (int l = 0; l < half; ++l) ArithF64::gs(v[(u << (j + 1)) | l], v[((u << (j + 1)) | l) + half], w, p);
        }
    }
#pragma unroll
    for (int l = 0; l < (1 << (R - 1)); ++l) ArithF64::gs_scaled(v[l], v[l + (1 << (R - 1))], cs.w_last_scaled, cs.n_inv, p);
}

struct CommitTopJob {
    uint64_t* rows;              // [vectors][row_words]
    uint64_t* ws;                // [vectors][k][n] raw elements between the forward round and the middle stage
    const uint64_t* keys;        // [vectors][4]
    const uint64_t* msgs;        // [vectors][msg_len]
    uint64_t msg_len, copy;
    const uint64_t* cdf;
    uint32_t entries, vectors, k;
    uint64_t row_words;
    uint64_t q, t;
};

template <int R>
__global__ void __launch_bounds__(256) commit_top_forward_kernel(CommitTopJob job, int lo, ModParams p, const double* __restrict__ tw) {
    constexpr int N = 1 << R;
    __shared__ __attribute__((aligned(16))) int16_t tile[N * 256];
    const size_t g0 = (size_t)blockIdx.x * 256u;
    const uint32_t poly = (uint32_t)(g0 >> lo), low0 = (uint32_t)(g0 & (((size_t)1 << lo) - 1));
    if (poly >= job.vectors * job.k) return;
    const uint32_t j = poly / job.k, i = poly - j * job.k;
    const LaneTable tab = lane_table_load(job.cdf, job.entries);
    top_round_sample<R>(tile, job.keys + 4 * (size_t)j, kDomR, i, lo, low0, tab, job.entries);
    __syncthreads();
    double v[N];
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = (double)tile[k * 256 + threadIdx.x];
    top_round_forward<R>(v, tw, p);
    uint64_t* const dst = job.ws + ((size_t)poly << (lo + R)) + low0 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < N; ++k) dst[(size_t)k << lo] = (uint64_t)__double_as_longlong(v[k]);
}

template <int R>
__global__ void __launch_bounds__(256) commit_top_inverse_kernel(CommitTopJob job, int lo, ModParams p, const double* __restrict__ tw, RoundConsts<ArithF64> cs) {
    constexpr int N = 1 << R;
    __shared__ __attribute__((aligned(16))) int16_t tile[N * 256];
    const size_t g0 = (size_t)blockIdx.x * 256u;
    const uint32_t poly = (uint32_t)(g0 >> lo), low0 = (uint32_t)(g0 & (((size_t)1 << lo) - 1));
    const uint32_t comps = job.k + 1;
    if (poly >= job.vectors * comps) return;
    const uint32_t j = poly / comps, c = poly - j * comps;
    uint64_t* const row = job.rows + (size_t)j * job.row_words;
    uint64_t* const data = row + kRowHeaderWords + ((size_t)c << (lo + R)) + low0 + threadIdx.x;
    double v[N];
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = __longlong_as_double((long long)__builtin_nontemporal_load(data + ((size_t)k << lo)));   // in flight under the cipher
    const LaneTable tab = lane_table_load(job.cdf, job.entries);
    const bool scalar = c == job.k;                                    // v = . + e2 + round(q (m mod t) / t); the others u_c = . + e1_c
    top_round_sample<R>(tile, job.keys + 4 * (size_t)j, scalar ? kDomE2 : kDomE1, scalar ? 0u : c, lo, low0, tab, job.entries);
    __syncthreads();
    top_round_inverse<R>(v, tw, cs, p);
    const PlainScale ps = make_plain_scale(job.q, job.t);
    const uint64_t* const msg = job.msgs + (size_t)j * job.msg_len;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const uint32_t x = ((uint32_t)k << lo) + low0 + threadIdx.x;
        double out = v[k] + (double)tile[k * 256 + threadIdx.x];
        if (scalar && x < job.copy) out += embed_plain(msg[x], ps);
        __builtin_nontemporal_store(u52_from_f64(canonical_f64(out, p.qd, p.inv_qd)), data + ((size_t)k << lo));
    }
    if (c == 0 && low0 == 0 && threadIdx.x < kRowHeaderWords) {
        const uint32_t w = threadIdx.x;
        const uint64_t n = 1ull << (lo + R);
        row[w] = w == 0 ? 8ull * (job.row_words - 1) : (w == 1 ? kRowMagic : (w == 2 ? (n | ((uint64_t)job.k << 32)) : (w == 3 ? job.q : job.t)));
    }
}

// ---- openings at n = 2^16 / 2^17: top forward round straight from the rows (with the header and canonicity screening), the one-column
// middle stage with s_hat, top inverse round into the decode-and-compare ----
struct VerifyTopJob {
    const uint64_t* rows;        // [count][row_words]
    uint64_t* ws;                // [count][k][n] raw: forward round -> middle stage
    uint64_t* ws_out;            // [count][n] raw: middle stage -> inverse round
    const uint64_t* msgs;
    uint64_t msg_len, row_words;
    unsigned long long* flags;
    uint32_t* bad;
    uint32_t count, k;
    uint64_t q, t;
};
template <int R>
__global__ void __launch_bounds__(256) verify_top_forward_kernel(VerifyTopJob job, int lo, ModParams p, const double* __restrict__ tw) {
    constexpr int N = 1 << R;
    const size_t g0 = (size_t)blockIdx.x * 256u;
    const uint32_t poly = (uint32_t)(g0 >> lo), low0 = (uint32_t)(g0 & (((size_t)1 << lo) - 1));
    if (poly >= job.count * job.k) return;
    const uint32_t j = poly / job.k, i = poly - j * job.k;
    const uint64_t* const row = job.rows + (size_t)j * job.row_words;
    const uint64_t* const src = row + kRowHeaderWords + ((size_t)i << (lo + R)) + low0 + threadIdx.x;
    double v[N];
    bool ok = true;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const uint64_t raw = __builtin_nontemporal_load(src + ((size_t)k << lo));
        ok = ok && raw < job.q;
        v[k] = f64_from_u52(raw);
    }
    if (i == 0 && low0 == 0 && threadIdx.x < kRowHeaderWords) {
        const uint32_t w = threadIdx.x;
        const uint64_t n = 1ull << (lo + R);
        const uint64_t want = w == 0 ? 8ull * (job.row_words - 1) : (w == 1 ? kRowMagic : (w == 2 ? (n | ((uint64_t)job.k << 32)) : (w == 3 ? job.q : job.t)));
        ok = ok && row[w] == want;
    }
    if (!ok) atomicOr(&job.bad[j], 1u);
    top_round_forward<R>(v, tw, p);
    uint64_t* const dst = job.ws + ((size_t)poly << (lo + R)) + low0 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < N; ++k) dst[(size_t)k << lo] = (uint64_t)__double_as_longlong(v[k]);
}
template <int R>
__global__ void __launch_bounds__(256) verify_top_inverse_kernel(VerifyTopJob job, int lo, ModParams p, const double* __restrict__ tw, RoundConsts<ArithF64> cs) {
    constexpr int N = 1 << R;
    const size_t g0 = (size_t)blockIdx.x * 256u;
    const uint32_t j = (uint32_t)(g0 >> lo), low0 = (uint32_t)(g0 & (((size_t)1 << lo) - 1));
    if (j >= job.count) return;
    const uint64_t* const data = job.ws_out + ((size_t)j << (lo + R)) + low0 + threadIdx.x;
    const uint64_t* const vsrc = job.rows + (size_t)j * job.row_words + kRowHeaderWords + ((size_t)job.k << (lo + R)) + low0 + threadIdx.x;
    const uint64_t* const msg = job.msgs + (size_t)j * job.msg_len;
    double v[N];
    uint64_t vraw[N];
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = __longlong_as_double((long long)__builtin_nontemporal_load(data + ((size_t)k << lo)));
#pragma unroll
    for (int k = 0; k < N; ++k) vraw[k] = __builtin_nontemporal_load(vsrc + ((size_t)k << lo));
    top_round_inverse<R>(v, tw, cs, p);
    uint64_t diff = 0;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const uint32_t x = ((uint32_t)k << lo) + low0 + threadIdx.x;
        ok = ok && vraw[k] < job.q;
        if (x < job.msg_len) {
            const uint64_t w = u52_from_f64(canonical_f64(f64_from_u52(vraw[k]) - v[k], p.qd, p.inv_qd));
            diff |= decode_slot(w, job.t, p) ^ msg[x];
        }
    }
    if (!ok) atomicOr(&job.bad[j], 1u);
    if (diff) atomicOr(&job.flags[j], (unsigned long long)diff);
}

}  // namespace lsr
