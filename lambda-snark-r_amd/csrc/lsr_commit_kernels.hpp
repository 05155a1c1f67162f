// Elementwise kernels of the GENERAL commitment path (any degree, rank and arithmetic flavour) and of the wire format: matrix–vector
// products in the NTT domain, blinding adds, the scalar component's epilogue, linear combination, decode-and-compare, row packing.
// The fused pipelines (lsr_commit_fused.hpp, lsr_commit_tile.hpp) replace most of them for the shapes they serve; these remain the
// reference the fused paths are tested against word for word (reference: cpp-core/src/commitment.cpp:138-276).
#pragma once

#include <algorithm>
#include <type_traits>

#include "lsr_arith.hpp"
#include "lsr_commit_tile.hpp"

namespace lsr {

constexpr uint64_t kWireMagic = kRowMagic;                // "LSRC0001" (lsr_commit_tile.hpp)
constexpr size_t kHeaderWords = kRowHeaderWords;          // data[0] + 4 header words

// out[j][row][x] = sum_col M[row*row_stride + col*col_stride][x] * vec[j][col][x]  (+ add[row][x])   mod q
// One lane per (j, row, x).  F64 selects the exact FP64-FMA product (q < 2^45).
template <bool F64>
__global__ void __launch_bounds__(256) matvec_kernel(uint64_t* __restrict__ out, const uint64_t* __restrict__ mat, const uint64_t* __restrict__ vec,
                                                       const uint64_t* __restrict__ add, uint32_t rows, uint32_t cols, uint32_t row_stride,
                                                       uint32_t col_stride, uint32_t logn, uint64_t batch, ModParams p) {
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t n = 1ull << logn;
    if (gid >= batch * rows * n) return;
    const uint64_t x = gid & (n - 1);
    const uint64_t jr = gid >> logn;
    const uint64_t row = jr % rows;
    const uint64_t j = jr / rows;
    const uint64_t* v = vec + (j * cols) * n + x;
    const uint64_t* m = mat + (row * row_stride) * n + x;
    uint64_t result;
    if (F64) {
        double acc = add ? f64_from_u52(add[row * n + x]) : 0.0;
        for (uint32_t c = 0; c < cols; ++c)
            acc += mulmod_f64(f64_from_u52(m[(uint64_t)c * col_stride * n]), f64_from_u52(v[(uint64_t)c * n]), p.qd, p.inv_qd);
        result = u52_from_f64(canonical_f64(acc, p.qd, p.inv_qd));
    } else {
        uint64_t acc = add ? add[row * n + x] : 0;
        for (uint32_t c = 0; c < cols; ++c) {
            acc += mulmod_barrett128(m[(uint64_t)c * col_stride * n], v[(uint64_t)c * n], p);
            if (acc >= p.q) acc -= p.q;
        }
        result = acc;
    }
    out[gid] = result;
}

// Rank-specialised form for the commitment's square product.  One lane owns residue x and walks JB consecutive
// witness vectors with the K*K matrix residues of column x held in registers (the matrix is shared by the whole
// batch, so its L2 traffic drops by JB):  out[j][c] = sum_i M[i*K + c] * vec[j][i]  (TRANSPOSED = A^T r, the commit)
// or out[j][c] = sum_i M[c*K + i] * vec[j][i] (+ add[c]) (A s + e, key generation).
template <int K, bool F64, bool TRANSPOSED, int JB>
__global__ void __launch_bounds__(256) matvec_square_kernel(uint64_t* __restrict__ out, const uint64_t* __restrict__ mat, const uint64_t* __restrict__ vec,
                                                              const uint64_t* __restrict__ add, uint32_t logn, uint64_t batch, ModParams p) {
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t n = 1ull << logn;
    const uint64_t x = gid & (n - 1);
    const uint64_t j0 = (gid >> logn) * JB;
    if (j0 >= batch) return;
    using T = typename std::conditional<F64, double, uint64_t>::type;
    T m[K][K], a[K];
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
        for (int c = 0; c < K; ++c) {
            const uint64_t raw = mat[(uint64_t)(TRANSPOSED ? i * K + c : c * K + i) * n + x];
            if constexpr (F64) m[i][c] = f64_from_u52(raw);
            else m[i][c] = raw;
        }
#pragma unroll
    for (int c = 0; c < K; ++c) {
        const uint64_t raw = add ? add[(uint64_t)c * n + x] : 0;
        if constexpr (F64) a[c] = f64_from_u52(raw);
        else a[c] = raw;
    }
#pragma unroll
    for (int jj = 0; jj < JB; ++jj) {
        const uint64_t j = j0 + jj;
        if (j >= batch) break;
        const uint64_t* v = vec + (j * K) * n + x;
        uint64_t* o = out + (j * K) * n + x;
        T r[K], acc[K];
#pragma unroll
        for (int i = 0; i < K; ++i) {
            if constexpr (F64) r[i] = f64_from_u52(v[(uint64_t)i * n]);
            else r[i] = v[(uint64_t)i * n];
        }
#pragma unroll
        for (int c = 0; c < K; ++c) acc[c] = a[c];
#pragma unroll
        for (int i = 0; i < K; ++i)
#pragma unroll
            for (int c = 0; c < K; ++c) {
                if constexpr (F64) {
                    acc[c] += mulmod_f64(m[i][c], r[i], p.qd, p.inv_qd);
                } else {
                    acc[c] += mulmod_barrett128(m[i][c], r[i], p);
                    if (acc[c] >= p.q) acc[c] -= p.q;
                }
            }
#pragma unroll
        for (int c = 0; c < K; ++c) {
            if constexpr (F64) o[(uint64_t)c * n] = u52_from_f64(canonical_f64(acc[c], p.qd, p.inv_qd));
            else o[(uint64_t)c * n] = acc[c];
        }
    }
}

// dst = (dst + a (+ b)) mod q, all canonical
__global__ void __launch_bounds__(256) add_mod_kernel(uint64_t* __restrict__ dst, const uint64_t* __restrict__ a, const uint64_t* __restrict__ b,
                                                        uint64_t count, uint64_t q) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
        uint64_t s = dst[i] + a[i];
        if (s >= q) s -= q;
        if (b) {
            s += b[i];
            if (s >= q) s -= q;
        }
        dst[i] = s;
    }
}

// v[j][x] = (v[j][x] + e2[j][x] + round(q (msg[j][x] mod t) / t)) mod q for x < copy, the message term absent beyond:
// the scalar component's epilogue (e2 blinding + message embedding, commitment.cpp:146-152 truncation/padding)
__global__ void __launch_bounds__(256) finish_v_kernel(uint64_t* __restrict__ v, const uint64_t* __restrict__ e2, const uint64_t* __restrict__ msgs,
                                                         uint64_t msg_len, uint64_t copy, uint32_t logn, uint64_t count, uint64_t delta, uint64_t t,
                                                         uint64_t q) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    const uint64_t nmask = (1ull << logn) - 1;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
        const uint64_t x = i & nmask, j = i >> logn;
        uint64_t s = v[i] + e2[i];
        if (s >= q) s -= q;
        if (x < copy) {
            // round(q m' / t) = delta m' + floor((rho m' + t/2) / t), m' = m mod t (lsr_commit_tile.hpp embed_plain; any q < 2^61 here)
            const uint64_t mm = msgs[j * msg_len + x] % t;
            s += delta * mm + ((q - delta * t) * mm + (t >> 1)) / t;   // < q
            if (s >= q) s -= q;
        }
        v[i] = s;
    }
}

// acc[x] = (acc[x] + sum_i coeffs[i] * terms[i][x]) mod q — the linear combination of commitments (K6, commitment.cpp:247-266):
// every commitment body is read once, the accumulator is read and written once per pass
__global__ void __launch_bounds__(256) combine_kernel(uint64_t* __restrict__ acc, const uint64_t* __restrict__ terms, const uint64_t* __restrict__ coeffs,
                                                       uint32_t count, uint64_t words, ModParams p) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t x = (uint64_t)blockIdx.x * 256 + threadIdx.x; x < words; x += stride) {
        uint64_t s = acc[x];
        for (uint32_t i = 0; i < count; ++i) {
            s += mulmod_barrett128(coeffs[i], terms[(uint64_t)i * words + x], p);
            if (s >= p.q) s -= p.q;
        }
        acc[x] = s;
    }
}

// w = v_hat - <s_hat, u_hat> is computed with matvec (rows = 1) and a subtraction; this kernel does
// dst = (a - dst) mod q
__global__ void __launch_bounds__(256) rsub_mod_kernel(uint64_t* __restrict__ dst, const uint64_t* __restrict__ a, uint64_t count, uint64_t q) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
        const uint64_t d = dst[i], s = a[i];
        dst[i] = s >= d ? s - d : s + q - d;
    }
}

// flag |= OR_i ( round(t * w_i / q) mod t ) xor msg_i   — OR-of-XOR compare of commitment.cpp:223-228
// minuend (optional, canonical residues): decode minuend_i - w_i instead of w_i
__global__ void __launch_bounds__(256) decode_compare_kernel(const uint64_t* __restrict__ w, const uint64_t* __restrict__ msg, uint64_t msg_len,
                                                               uint64_t t, ModParams p, unsigned long long* __restrict__ flag,
                                                               const uint64_t* __restrict__ minuend = nullptr) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint64_t diff = 0;
    if (i < msg_len) {
        const uint64_t wi = minuend ? (minuend[i] >= w[i] ? minuend[i] - w[i] : minuend[i] + p.q - w[i]) : w[i];
        diff = decode_slot(wi, t, p) ^ msg[i];         // raw message word (commitment.cpp:224)
    }
    if (diff) atomicOr(flag, (unsigned long long)diff);
}

// batched form: flags[j] |= OR_i decode(w[j][i]) xor msg[j][i], one lane per (j, i)
__global__ void __launch_bounds__(256) decode_compare_batch_kernel(const uint64_t* __restrict__ w, const uint64_t* __restrict__ msgs, uint64_t msg_len,
                                                                     uint32_t logn, uint64_t count, uint64_t t, ModParams p,
                                                                     unsigned long long* __restrict__ flags) {
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= count * msg_len) return;
    const uint64_t j = gid / msg_len, i = gid - j * msg_len;
    const uint64_t wi = w[(j << logn) + i];
    const uint64_t diff = decode_slot(wi, t, p) ^ msgs[gid];
    if (diff) atomicOr(&flags[j], (unsigned long long)diff);
}

static unsigned grid_for(uint64_t work, unsigned cap = 256 * 16) {
    const uint64_t blocks = (work + 255) / 256;
    return static_cast<unsigned>(std::min<uint64_t>(blocks, cap));
}


// wire format rows [batch][5 + kn + n] assembled on the device (one contiguous copy back instead of a host-side scatter)
__global__ void __launch_bounds__(256) pack_commitments_kernel(uint64_t* __restrict__ out, const uint64_t* __restrict__ u, const uint64_t* __restrict__ v,
                                                                uint64_t kn, uint64_t n, uint64_t batch, uint64_t q, uint64_t t, uint64_t shape) {
    const uint64_t words = kHeaderWords + kn + n;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < batch * words; i += stride) {
        const uint64_t j = i / words, w = i - j * words;
        uint64_t x;
        if (w >= kHeaderWords + kn) x = v[j * n + (w - kHeaderWords - kn)];
        else if (w >= kHeaderWords) x = u[j * kn + (w - kHeaderWords)];
        else x = w == 0 ? 8ull * (words - 1) : (w == 1 ? kWireMagic : (w == 2 ? shape : (w == 3 ? q : t)));
        out[i] = x;
    }
}

// the reverse of pack_commitments_kernel with the checks of parse_commitment and the canonicity screening: rows -> u, v;
// bad[j] != 0 when row j is not a commitment of this context (wrong header) or holds a residue >= q
__global__ void __launch_bounds__(256) unpack_commitments_kernel(const uint64_t* __restrict__ rows, uint64_t* __restrict__ u, uint64_t* __restrict__ v,
                                                                  uint32_t* __restrict__ bad, uint64_t kn, uint64_t n, uint64_t batch, uint64_t q, uint64_t t,
                                                                  uint64_t shape) {
    const uint64_t words = kHeaderWords + kn + n;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < batch * words; i += stride) {
        const uint64_t j = i / words, w = i - j * words;
        const uint64_t x = rows[i];
        bool ok;
        if (w >= kHeaderWords + kn) { v[j * n + (w - kHeaderWords - kn)] = x; ok = x < q; }
        else if (w >= kHeaderWords) { u[j * kn + (w - kHeaderWords)] = x; ok = x < q; }
        else ok = x == (w == 0 ? 8ull * (words - 1) : (w == 1 ? kWireMagic : (w == 2 ? shape : (w == 3 ? q : t))));
        if (!ok) atomicOr(&bad[j], 1u);
    }
}


// flags[j] / bad[j] -> results[j] = -1 (not a canonical commitment of this context) / 1 (opens) / 0
__global__ void __launch_bounds__(256) opening_verdict_kernel(const unsigned long long* __restrict__ flags, const uint32_t* __restrict__ bad,
                                                               int* __restrict__ results, uint64_t count) {
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < count) results[j] = bad[j] ? -1 : (flags[j] == 0 ? 1 : 0);
}

}  // namespace lsr
