// Prover-side polynomial path on the NTT kernels: natural-order cyclic transforms (rust-api/lambda-snark/src/ntt.rs)
// and the NTT-path quotient polynomial (rust-api/lambda-snark/src/r1cs.rs:474-506).  C-ABI in lambda_snark/prover.h.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "lambda_snark/prover.h"
#include "lsr_runtime.hpp"

namespace lsr {

constexpr int kBlock = 256;

static unsigned blocks_for(size_t work, unsigned cap = 256 * 32) {
    return static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>((work + kBlock - 1) / kBlock, cap)));
}

__device__ __forceinline__ uint64_t gold_canon(uint64_t x) { return x >= kGoldilocks ? x - kGoldilocks : x; }

// ---- quotient pipeline -------------------------------------------------------------------------------------------
// r1cs.rs:489-503 computes N = A B - C from the interpolated polynomials and divides by Z_H = X^m - 1, failing when a
// remainder is left.  Equivalent, with transforms of size m only:
//   * the remainder vanishes iff N vanishes on H = {omega^k}, i.e. iff a_k b_k = c_k for every constraint k;
//   * then Q = N / Z_H has degree <= m - 2, so it is fixed by its values on the coset psi H (psi = omega_2m, psi^2 = omega),
//     where Z_H(psi omega^k) = psi^m - 1 = -2:   Q(psi omega^k) = (A B - C)(psi omega^k) / (-2).
// Radix-2 in-place networks permute by bit reversal (P).  With F_w the DFT matrix of root w and a context built on the
// CONJUGATE root omega^-1, the two launches are
//      forward = P F_{1/omega}           : natural-order values -> m x (inverse DFT), bit-reversed
//      inverse = (P F_{1/omega})^-1      = m^-1 F_omega P : bit-reversed coefficients -> natural-order evaluations
// so  e --forward--> m P coeffs --[x psi^bitrev(p) fused into the read-in]--inverse--> evaluations on psi H (natural order)
// needs no permutation and no scaling, and the way back is forward again.
// C enters linearly, so it never goes to the coset: modulo X^m + 1 (whose roots are psi H) Z_H = -2 and C is its own
// remainder, hence Q = (C - (A B mod X^m + 1)) / 2 coefficient by coefficient.  Six transforms in all: A, B, C interpolated
// (c^ = m P c stays where it is), A and B evaluated on the coset and multiplied, one transform back (z = m P (psi^j g_j)_j
// with g = A B mod X^m + 1), and the finish Q_j = (2m)^-1 (c^[p] - psi^-j z[p]), p = bitrev(j) — the only place where m
// words are put in natural order (through LDS).

// test a_k b_k = c_k (is_satisfied, r1cs.rs:148-172) — the first transform reads the evaluations where they lie
__global__ void __launch_bounds__(kBlock) check_kernel(const uint64_t* __restrict__ a, const uint64_t* __restrict__ b, const uint64_t* __restrict__ c,
                                                       uint32_t* __restrict__ bad, int logm, size_t per_vector) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t base = (size_t)blockIdx.x * kBlock; base < per_vector; base += stride) {   // wave-uniform trip count
        const size_t i = base + threadIdx.x;
        const bool live = i < per_vector;
        bool wrong = false;
        if (live) {
            wrong = gold_mul(gold_canon(a[i]), gold_canon(b[i])) != gold_canon(c[i]);
        }
        if (logm >= 6) {   // a wavefront's 64 consecutive constraints belong to one instance
            if (__ballot(wrong) && (threadIdx.x & 63) == 0) atomicOr(&bad[i >> logm], 1u);
        } else if (wrong) {
            atomicOr(&bad[i >> logm], 1u);
        }
    }
}

// evaluations of A B on the coset, written over A's
__global__ void __launch_bounds__(kBlock) product_kernel(uint64_t* __restrict__ a, const uint64_t* __restrict__ b, size_t count) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += stride) a[i] = gold_mul(a[i], b[i]);
}

// compute_constraint_evals (r1cs.rs:296-304): out[mat][inst][row] = sum_e val[e] * z[inst][col[e]] over the row's CSR run.
// One lane per (instance, row); blockIdx.y selects the matrix.  Witness words may be any 64-bit value (v[col] % modulus).
struct CsrView {
    const uint32_t* row_ptr;   // [m + 1]
    const uint32_t* col;
    const uint64_t* val;       // value mod q, in Montgomery form
};
__global__ void __launch_bounds__(kBlock) constraint_evals_kernel(uint64_t* __restrict__ out, CsrView a, CsrView b, CsrView c,
                                                                  const uint64_t* __restrict__ z, uint32_t n_vars, int logm, size_t per_vector) {
    const CsrView mat = blockIdx.y == 0 ? a : (blockIdx.y == 1 ? b : c);
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < per_vector; i += stride) {
        const uint32_t row = (uint32_t)i & ((1u << logm) - 1u);
        const uint64_t* zi = z + (i >> logm) * n_vars;
        uint64_t acc = 0;
        for (uint32_t e = mat.row_ptr[row]; e < mat.row_ptr[row + 1]; ++e) acc = gold_add(acc, gold_mul_mont(zi[mat.col[e]], mat.val[e]));
        out[blockIdx.y * per_vector + i] = acc;
    }
}

constexpr int kSplitTile = 4096;                 // quotient words per workgroup
constexpr int kSplitPerThread = kSplitTile / kBlock;
__device__ __forceinline__ int split_slot(int i) { return i + (i >> 6); }   // one pad word per 64

// z, chat = [instances][m] in bit-reversed order -> quotient[inst][j] = half_m_inv * chat[inst][p] - untwist[p] * z[inst][p],
// p = bitrev(j), untwist[p] = (2m)^-1 psi^-bitrev(p) (both multipliers in Montgomery form); and per instance `top` =
// 1 + highest non-zero index.
// LOGM_HIGH: m >= 4096, one workgroup moves the 4096 words whose index has a fixed middle field (bits 6..logm-7):
// 64-word runs on both the read and the write side.
template <bool LOGM_HIGH>
__global__ void __launch_bounds__(kBlock) finish_quotient_kernel(const uint64_t* __restrict__ z, const uint64_t* __restrict__ chat,
                                                                 const uint64_t* __restrict__ untwist, uint64_t half_m_inv,
                                                                 uint64_t* __restrict__ quotient, uint32_t* __restrict__ top, int logm, size_t total) {
    __shared__ uint64_t tile[kSplitTile + kSplitTile / 64];
    const int t = threadIdx.x;
    const uint32_t mmask = (1u << logm) - 1u;
    if constexpr (LOGM_HIGH) {
        // bit-reversed-order index (A:6 | B | C:6)  ->  natural index (rev C | rev B | rev A)
        const int mid_bits = logm - 12;
        const size_t inst = blockIdx.x >> mid_bits;
        const uint32_t B = blockIdx.x & ((1u << mid_bits) - 1u);
        const uint32_t Brev = mid_bits ? (__brev(B) >> (32 - mid_bits)) : 0u;
        const int lane = t & 63, wave = t >> 6;
#pragma unroll
        for (int r = 0; r < kSplitPerThread; ++r) {
            const int A = wave * kSplitPerThread + r;
            const uint32_t pidx = ((uint32_t)A << (logm - 6)) | (B << 6) | (uint32_t)lane;
            const size_t g = (inst << logm) + pidx;
            tile[A * 65 + lane] = gold_sub(gold_mul_mont(chat[g], half_m_inv), gold_mul_mont(z[g], untwist[pidx]));
        }
        __syncthreads();
        uint32_t best = 0;
#pragma unroll
        for (int r = 0; r < kSplitPerThread; ++r) {
            const int Cout = wave * kSplitPerThread + r;              // top field of the natural index
            const uint32_t nat = ((uint32_t)Cout << (logm - 6)) | (Brev << 6) | (uint32_t)lane;
            const uint64_t h = tile[(__brev((uint32_t)lane) >> 26) * 65 + (__brev((uint32_t)Cout) >> 26)];
            quotient[(inst << logm) + nat] = h;
            if (h != 0) best = nat + 1;                                // nat grows with r within a thread
        }
        for (int off = 32; off; off >>= 1) best = max(best, (uint32_t)__shfl_xor((int)best, off));
        if (best && lane == 0) atomicMax(&top[inst], best);
    } else {
        // m < 4096: the tile holds 2^(12-logm) whole instances; scatter into LDS, stream out in natural order
        const size_t tile_base = (size_t)blockIdx.x * kSplitTile;
#pragma unroll
        for (int r = 0; r < kSplitPerThread; ++r) {
            const int p = r * kBlock + t;
            const size_t g = tile_base + p;
            if (g < total) {
                const uint32_t j = (uint32_t)p & mmask;
                const uint32_t nat = logm ? (__brev(j) >> (32 - logm)) : 0u;
                tile[split_slot((p & ~(int)mmask) | (int)nat)] = gold_sub(gold_mul_mont(chat[g], half_m_inv), gold_mul_mont(z[g], untwist[j]));
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kSplitPerThread; ++r) {
            const int p = r * kBlock + t;
            const size_t g = tile_base + p;
            const bool live = g < total;
            const uint32_t nat = (uint32_t)p & mmask;
            uint64_t h = 0;
            if (live) {
                h = tile[split_slot(p)];
                quotient[g] = h;
            }
            if (logm >= 6) {   // a wavefront's 64 consecutive words belong to one instance: one atomic per wave
                const uint64_t nz = __ballot(live && h != 0);
                if (nz && (t & 63) == 63 - __clzll(nz)) atomicMax(&top[g >> logm], nat + 1);
            } else if (live && h != 0) {
                atomicMax(&top[g >> logm], nat + 1);
            }
        }
    }
}

__global__ void __launch_bounds__(kBlock) quotient_len_kernel(uint32_t* __restrict__ len, const uint32_t* __restrict__ top,
                                                              const uint32_t* __restrict__ bad, size_t batch) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < batch) len[i] = bad[i] ? 0u : (top[i] ? top[i] : 1u);
}

}  // namespace lsr

struct LsrQuotientPlan {
    uint32_t m = 0;
    int logm = 0;
    int device = 0;
    NttContext* ntt = nullptr;                // size m, on the conjugate root omega_m^-1 (absent for m = 1)
    lsr::DeviceBuffer<uint64_t> twist;        // psi^bitrev(p), p < m                  } all three in Montgomery form
    lsr::DeviceBuffer<uint64_t> untwist;      // (2m)^-1 psi^-bitrev(p), p < m         } (gold_mul_mont)
    uint64_t half_m_inv = 0;                  // (2m)^-1                               }
    std::mutex mutex;                         // guards the workspace and `stream`
    lsr::DeviceBuffer<uint64_t> work;         // [3][chunk][m]
    lsr::DeviceBuffer<uint32_t> flags;        // top[chunk], bad[chunk]
    lsr::DeviceBuffer<uint64_t> io;           // host-API staging of the quotient: [chunk][m]
    lsr::DeviceBuffer<uint32_t> io_len;
    size_t chunk = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_last = nullptr;             // end of the last asynchronous call: the next call on this plan (any stream) starts behind it
    // read from the environment ONCE, when the plan is created
    int chunk_log2 = 26;                      // LAMBDA_SNARK_QUOTIENT_CHUNK_LOG2: evaluations per pass and plane
    bool fuse = true;                         // LAMBDA_SNARK_QUOTIENT_FUSE=0: the a b = c test and the coset product as kernels of their own
};

namespace lsr {

// instances per pass: bound the workspace (3 m words per instance; 2^26 evaluations per vector ~ 1.5 GiB; p.chunk_log2: tests force
// several passes with LAMBDA_SNARK_QUOTIENT_CHUNK_LOG2 at plan creation)
static size_t quotient_chunk(const LsrQuotientPlan& p, size_t batch) {
    const size_t cap = std::max<size_t>(1, (size_t(1) << p.chunk_log2) >> p.logm);
    return std::min(batch, cap);
}

static void ensure_workspace(LsrQuotientPlan& p, size_t chunk, bool host_io) {
    if (chunk > p.chunk) {
        p.work.allocate(3 * chunk * p.m);
        p.flags.allocate(2 * chunk);
        p.io.release();
        p.io_len.release();
        p.chunk = chunk;
    }
    if (host_io && p.io.count < p.chunk * p.m) {
        p.io.allocate(p.chunk * p.m);
        p.io_len.allocate(p.chunk);
    }
}

// one pass over `count` <= plan.chunk instances, everything on `s`
static void quotient_pass(LsrQuotientPlan& p, const uint64_t* d_a, const uint64_t* d_b, const uint64_t* d_c, size_t count, uint64_t* d_q,
                          uint32_t* d_len, hipStream_t s) {
    const size_t per_vector = count << p.logm;
    uint64_t* work = p.work.ptr;
    uint32_t* top = p.flags.ptr;
    uint32_t* bad = p.flags.ptr + count;
    zero_words_async(reinterpret_cast<uint64_t*>(p.flags.ptr), count, s);                 // top[count], bad[count]: 2 count u32 = count words
    // m <= 4096 (one tile launch per transform): the two elementwise kernels ride in the read-in of a transform — the a b = c test in
    // C's interpolation, the coset product in the last transform (the transforms are integer-VALU-bound, the extra loads cost nothing
    // and two passes over the planes disappear: profiles/r02b_quotient_fusion.txt).  A plan created under LAMBDA_SNARK_QUOTIENT_FUSE=0 keeps them apart.
    const bool fuse = p.fuse && p.ntt && ntt_forward_can_fuse(*p.ntt);
    const bool fuse_check = fuse && d_a != work;
    if (!fuse_check) hipLaunchKernelGGL(check_kernel, dim3(blocks_for(per_vector)), dim3(kBlock), 0, s, d_a, d_b, d_c, bad, p.logm, per_vector);
    if (p.ntt) {
        if (d_a == work) {                                                               // interpolation: r1cs.rs:489-491
            launch_ntt(*p.ntt, work, 3 * count, false, s);
        } else {   // out of place: the caller's arrays are read, the workspace planes written
            launch_ntt(*p.ntt, work, count, false, s, nullptr, nullptr, d_a);
            launch_ntt(*p.ntt, work + per_vector, count, false, s, nullptr, nullptr, d_b);
            if (fuse_check) launch_ntt_forward_fused(*p.ntt, work + 2 * per_vector, count, s, d_c, 2, d_a, d_b, bad);
            else launch_ntt(*p.ntt, work + 2 * per_vector, count, false, s, nullptr, nullptr, d_c);
        }
        launch_ntt(*p.ntt, work, 2 * count, true, s, nullptr, p.twist.ptr);              // A, B on the coset psi H
        if (fuse) {
            // (a b) -> coefficients (r1cs.rs:495) and the finish below in one launch: the product rides in the transform's read-in, the
            // subtraction of c, the untwist, the bit reversal and the degree bound in its write-out (lsr_ntt_kernels.hpp, MODE 3)
            launch_ntt_forward_finish(*p.ntt, work, count, s, work + per_vector, work + 2 * per_vector, p.untwist.ptr, p.half_m_inv, d_q, top);
        } else {
            hipLaunchKernelGGL(product_kernel, dim3(blocks_for(per_vector)), dim3(kBlock), 0, s, work, work + per_vector, per_vector);   // r1cs.rs:495
            launch_ntt(*p.ntt, work, count, false, s);                                   // back to (twisted, bit-reversed) coefficients
        }
        if (fuse) {
        } else if (p.logm >= 12) {
            hipLaunchKernelGGL(finish_quotient_kernel<true>, dim3(static_cast<unsigned>(per_vector / kSplitTile)), dim3(kBlock), 0, s, work,
                               work + 2 * per_vector, p.untwist.ptr, p.half_m_inv, d_q, top, p.logm, per_vector);
        } else {
            hipLaunchKernelGGL(finish_quotient_kernel<false>, dim3(static_cast<unsigned>((per_vector + kSplitTile - 1) / kSplitTile)), dim3(kBlock), 0, s,
                               work, work + 2 * per_vector, p.untwist.ptr, p.half_m_inv, d_q, top, p.logm, per_vector);
        }
    } else {
        zero_words_async(d_q, per_vector, s);                                            // m = 1: constants, Q = 0 when a b = c
    }
    hipLaunchKernelGGL(quotient_len_kernel, dim3(blocks_for(count, ~0u)), dim3(kBlock), 0, s, d_len, top, bad, count);
    LSR_HIP(hipGetLastError());
}

static void quotient_device(LsrQuotientPlan& p, const uint64_t* d_a, const uint64_t* d_b, const uint64_t* d_c, size_t batch, uint64_t* d_q,
                            uint32_t* d_len, hipStream_t s) {
    DeviceGuard guard(p.device);
    std::lock_guard<std::mutex> lock(p.mutex);
    // the workspace planes are the plan's: calls on one plan run one behind the other whatever streams the caller passes (and a
    // workspace about to grow is not freed under a call that still uses it)
    // (not while `s` records into a HIP graph: a captured sequence is ordered by the capture, lsr_runtime.hpp stream_is_capturing)
    const bool capturing = stream_is_capturing(s);
    if (p.ev_last && !capturing) LSR_HIP(hipEventSynchronize(p.ev_last));
    const size_t chunk = quotient_chunk(p, batch);
    ensure_workspace(p, chunk, false);
    for (size_t done = 0; done < batch; done += chunk) {
        const size_t now = std::min(chunk, batch - done);
        const size_t off = done << p.logm;
        quotient_pass(p, d_a + off, d_b + off, d_c + off, now, d_q + off, d_len + done, s);
    }
    if (capturing) return;
    if (!p.ev_last) LSR_HIP(hipEventCreateWithFlags(&p.ev_last, hipEventDisableTiming));
    LSR_HIP(hipEventRecord(p.ev_last, s));
}

static void quotient_host(LsrQuotientPlan& p, const uint64_t* a, const uint64_t* b, const uint64_t* c, size_t batch, uint64_t* q, uint32_t* len) {
    DeviceGuard guard(p.device);
    std::lock_guard<std::mutex> lock(p.mutex);
    if (p.ev_last) LSR_HIP(hipEventSynchronize(p.ev_last));      // an asynchronous call still using the planes
    const size_t chunk = quotient_chunk(p, batch);
    ensure_workspace(p, chunk, true);
    for (size_t done = 0; done < batch; done += chunk) {
        const size_t now = std::min(chunk, batch - done);
        const size_t off = done << p.logm, words = now << p.logm, bytes = words * 8;
        // straight into the workspace planes: the check then runs in place, without the device-side copy
        LSR_HIP(hipMemcpyAsync(p.work.ptr, a + off, bytes, hipMemcpyHostToDevice, p.stream));
        LSR_HIP(hipMemcpyAsync(p.work.ptr + words, b + off, bytes, hipMemcpyHostToDevice, p.stream));
        LSR_HIP(hipMemcpyAsync(p.work.ptr + 2 * words, c + off, bytes, hipMemcpyHostToDevice, p.stream));
        quotient_pass(p, p.work.ptr, p.work.ptr + words, p.work.ptr + 2 * words, now, p.io.ptr, p.io_len.ptr, p.stream);
        LSR_HIP(hipMemcpyAsync(q + off, p.io.ptr, bytes, hipMemcpyDeviceToHost, p.stream));
        LSR_HIP(hipMemcpyAsync(len + done, p.io_len.ptr, now * sizeof(uint32_t), hipMemcpyDeviceToHost, p.stream));
        LSR_HIP(hipStreamSynchronize(p.stream));
    }
}

static void destroy_plan(LsrQuotientPlan* p) {
    if (!p) return;
    try {
        DeviceGuard guard(p->device);
        if (p->ev_last) {
            (void)hipEventSynchronize(p->ev_last);
            (void)hipEventDestroy(p->ev_last);
        }
        if (p->stream) (void)hipStreamDestroy(p->stream);
        p->work.release();
        p->flags.release();
        p->io.release();
        p->io_len.release();
        p->twist.release();
        p->untwist.release();
    } catch (...) {
    }
    destroy_ntt_context(p->ntt);
    delete p;
}

static LsrQuotientPlan* create_plan(uint32_t m, int device) {
    if (m == 0 || m > 131072 || (m & (m - 1))) {
        set_last_error("lsr_quotient_plan_create: m must be a power of two in [1, 131072] (r1cs.rs:386-389)");
        return nullptr;
    }
    const int devices = visible_device_count();
    if (devices <= 0) {
        set_last_error("lsr_quotient_plan_create: no HIP device visible — this library has no CPU fallback");
        std::fprintf(stderr, "lambda_snark_core: no HIP device visible; the MI355X backend has no CPU fallback\n");
        return nullptr;
    }
    if (device < 0) device = default_device();
    if (device < 0) return nullptr;                          // LOCAL_RANK / LAMBDA_SNARK_DEVICE names no visible device: message already set
    if (device >= devices) {
        set_last_error("lsr_quotient_plan_create: device index out of range");
        return nullptr;
    }
    auto* p = new LsrQuotientPlan;
    p->m = m;
    p->device = device;
    if (const char* e = std::getenv("LAMBDA_SNARK_QUOTIENT_CHUNK_LOG2")) {
        const int v = std::atoi(e);
        if (v >= 1 && v <= 30) p->chunk_log2 = v;
    }
    if (const char* e = std::getenv("LAMBDA_SNARK_QUOTIENT_FUSE")) p->fuse = !(e[0] == '0');
    while ((1u << p->logm) < m) ++p->logm;
    const uint64_t q = kProverModulus;
    if (m >= 2) {
        p->ntt = create_cyclic_ntt_context(q, m, invmod_prime(prover_root_of_unity(q, m), q), device);
        if (!p->ntt) {
            delete p;
            return nullptr;
        }
    }
    try {
        DeviceGuard guard(device);
        if (m >= 2) {
            const uint64_t psi = prover_root_of_unity(q, 2ull * m), psi_inv = invmod_prime(psi, q);
            std::vector<uint64_t> twist(m), untwist(m);
            const uint64_t half_m_inv = invmod_prime((2ull * m) % q, q);
            p->half_m_inv = prover_montgomery(half_m_inv);
            // The plan's own context transforms WITHOUT the m^-1 of its inverse direction (both scaling constants 1: the last stage then
            // costs no product, ArithGold::gs_scaled).  A and B reach the coset m times too large, their product and z carry m^2, and the
            // untwist table takes it back: untwist[p] = (2m)^-1 m^-2 psi^-bitrev(p).  (The context is private to the plan; the cyclic
            // transforms of lsr_cyclic_ntt_* use contexts of their own.)
            p->ntt->n_inv_gold = prover_montgomery(1);
            p->ntt->w_last_scaled_gold = prover_montgomery(1);       // the last stage's twiddle is omega^0
            const uint64_t m_inv = invmod_prime(m % q, q);
            uint64_t up = 1, down = mulmod(half_m_inv, mulmod(m_inv, m_inv, q), q);
            for (uint32_t j = 0; j < m; ++j) {
                twist[bit_reverse(j, p->logm)] = prover_montgomery(up);
                untwist[bit_reverse(j, p->logm)] = prover_montgomery(down);
                up = mulmod(up, psi, q);
                down = mulmod(down, psi_inv, q);
            }
            p->twist.upload(twist);
            p->untwist.upload(untwist);
        }
        LSR_HIP(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    } catch (const std::exception& e) {
        set_last_error(std::string("lsr_quotient_plan_create: ") + e.what());
        destroy_plan(p);
        return nullptr;
    }
    return p;
}

}  // namespace lsr

struct LsrR1csProver {
    uint32_t m = 0, n_vars = 0;
    LsrQuotientPlan* plan = nullptr;
    lsr::DeviceBuffer<uint32_t> row_ptr[3], col[3];
    lsr::DeviceBuffer<uint64_t> val[3];
    lsr::DeviceBuffer<uint64_t> witness;    // [chunk][n_vars]
    size_t witness_chunk = 0;
};

namespace lsr {

static void destroy_prover(LsrR1csProver* r) {
    if (!r) return;
    if (r->plan) {
        try {
            DeviceGuard guard(r->plan->device);
            for (int k = 0; k < 3; ++k) { r->row_ptr[k].release(); r->col[k].release(); r->val[k].release(); }
            r->witness.release();
        } catch (...) {
        }
    }
    destroy_plan(r->plan);
    delete r;
}

static LsrR1csProver* create_prover(const SparseMatrix* const mats[3], int device) {
    const uint32_t m = mats[0]->n_rows, n_vars = mats[0]->n_cols;
    for (int k = 0; k < 3; ++k) {
        if (mats[k]->n_rows != m || mats[k]->n_cols != n_vars || (mats[k]->n_entries && !mats[k]->entries) || mats[k]->n_entries > 0xFFFFFFF0ull) {
            set_last_error("lsr_r1cs_prover_create: A, B, C must share one shape");
            return nullptr;
        }
        for (size_t e = 0; e < mats[k]->n_entries; ++e)
            if (mats[k]->entries[e].row >= m || mats[k]->entries[e].col >= n_vars) {
                set_last_error("lsr_r1cs_prover_create: entry outside the matrix");
                return nullptr;
            }
    }
    if (n_vars == 0) {
        set_last_error("lsr_r1cs_prover_create: no variables");
        return nullptr;
    }
    auto* r = new LsrR1csProver;
    r->m = m;
    r->n_vars = n_vars;
    r->plan = create_plan(m, device);
    if (!r->plan) {
        delete r;
        return nullptr;
    }
    try {
        DeviceGuard guard(r->plan->device);
        for (int k = 0; k < 3; ++k) {   // coordinate form -> CSR (stable counting sort by row)
            const SparseMatrix& M = *mats[k];
            std::vector<uint32_t> ptr(m + 1, 0), cols(M.n_entries);
            std::vector<uint64_t> vals(M.n_entries);
            for (size_t e = 0; e < M.n_entries; ++e) ++ptr[M.entries[e].row + 1];
            for (uint32_t i = 0; i < m; ++i) ptr[i + 1] += ptr[i];
            std::vector<uint32_t> cursor(ptr.begin(), ptr.end() - 1);
            for (size_t e = 0; e < M.n_entries; ++e) {
                const uint32_t at = cursor[M.entries[e].row]++;
                cols[at] = M.entries[e].col;
                vals[at] = prover_montgomery(M.entries[e].value);     // mul_vec: val % modulus (held in Montgomery form)
            }
            r->row_ptr[k].upload(ptr);
            if (M.n_entries == 0) { cols.push_back(0); vals.push_back(0); }   // keep the pointers non-null
            r->col[k].upload(cols);
            r->val[k].upload(vals);
        }
    } catch (const std::exception& e) {
        set_last_error(std::string("lsr_r1cs_prover_create: ") + e.what());
        destroy_prover(r);
        return nullptr;
    }
    return r;
}

// witnesses (host) -> constraint evaluations in the plan's workspace planes, chunk by chunk; then either copy them out
// (evals != nullptr) or run the quotient pipeline on them in place
static void prover_run(LsrR1csProver& r, const uint64_t* witnesses, size_t batch, uint64_t* const evals[3], uint64_t* q, uint32_t* len) {
    LsrQuotientPlan& p = *r.plan;
    DeviceGuard guard(p.device);
    std::lock_guard<std::mutex> lock(p.mutex);
    if (p.ev_last) LSR_HIP(hipEventSynchronize(p.ev_last));      // an asynchronous lsr_quotient_batch_device call still using the planes
    const size_t chunk = quotient_chunk(p, batch);
    ensure_workspace(p, chunk, true);
    if (r.witness_chunk < p.chunk) {
        r.witness.allocate(p.chunk * r.n_vars);
        r.witness_chunk = p.chunk;
    }
    const CsrView a{r.row_ptr[0].ptr, r.col[0].ptr, r.val[0].ptr}, b{r.row_ptr[1].ptr, r.col[1].ptr, r.val[1].ptr},
        c{r.row_ptr[2].ptr, r.col[2].ptr, r.val[2].ptr};
    for (size_t done = 0; done < batch; done += chunk) {
        const size_t now = std::min(chunk, batch - done);
        const size_t per_vector = now << p.logm, off = done << p.logm;
        LSR_HIP(hipMemcpyAsync(r.witness.ptr, witnesses + done * r.n_vars, now * r.n_vars * 8, hipMemcpyHostToDevice, p.stream));
        hipLaunchKernelGGL(constraint_evals_kernel, dim3(blocks_for(per_vector), 3), dim3(kBlock), 0, p.stream, p.work.ptr, a, b, c, r.witness.ptr,
                           r.n_vars, p.logm, per_vector);
        LSR_HIP(hipGetLastError());
        if (evals) {
            for (int k = 0; k < 3; ++k)
                LSR_HIP(hipMemcpyAsync(evals[k] + off, p.work.ptr + k * per_vector, per_vector * 8, hipMemcpyDeviceToHost, p.stream));
        } else {
            quotient_pass(p, p.work.ptr, p.work.ptr + per_vector, p.work.ptr + 2 * per_vector, now, p.io.ptr, p.io_len.ptr, p.stream);
            LSR_HIP(hipMemcpyAsync(q + off, p.io.ptr, per_vector * 8, hipMemcpyDeviceToHost, p.stream));
            LSR_HIP(hipMemcpyAsync(len + done, p.io_len.ptr, now * sizeof(uint32_t), hipMemcpyDeviceToHost, p.stream));
        }
        LSR_HIP(hipStreamSynchronize(p.stream));
    }
}

// natural-order transforms for host callers: ntt.rs:117-201
static void cyclic_host(const NttContext& c, uint64_t* values, size_t batch, bool inverse) {
    DeviceGuard guard(c.device);
    const size_t n = c.degree;
    const size_t chunk = std::max<size_t>(1, std::min<size_t>(batch, (256ull << 20) / (n * 8)));
    DeviceBuffer<uint64_t> x(chunk * n), y(chunk * n);
    std::lock_guard<std::mutex> lock(c.staging_mutex);   // serialises use of work_stream(c)
    for (size_t done = 0; done < batch; done += chunk) {
        const size_t now = std::min(chunk, batch - done);
        LSR_HIP(hipMemcpyAsync(x.ptr, values + done * n, now * n * 8, hipMemcpyHostToDevice, work_stream(c)));
        if (inverse) {
            launch_bit_reverse(y.ptr, x.ptr, c.logn, now, work_stream(c));
            launch_ntt(c, y.ptr, now, true, work_stream(c));
        } else {
            launch_ntt(c, x.ptr, now, false, work_stream(c));
            launch_bit_reverse(y.ptr, x.ptr, c.logn, now, work_stream(c));
        }
        LSR_HIP(hipMemcpyAsync(values + done * n, y.ptr, now * n * 8, hipMemcpyDeviceToHost, work_stream(c)));
        LSR_HIP(hipStreamSynchronize(work_stream(c)));
    }
}

}  // namespace lsr

using lsr::set_last_error;

template <class F>
static int guarded(const char* where, F&& body) noexcept {
    try {
        body();
        return 0;
    } catch (const std::exception& e) {
        set_last_error(std::string(where) + ": " + e.what());
        std::fprintf(stderr, "lambda_snark_core: %s failed: %s\n", where, e.what());
        return -1;
    } catch (...) {
        set_last_error(std::string(where) + ": unknown exception");
        return -1;
    }
}

extern "C" {

uint64_t lsr_prover_modulus(void) noexcept { return lsr::kProverModulus; }
uint64_t lsr_prover_root_2_32(void) noexcept { return lsr::kProverRoot2_32; }
uint64_t lsr_prover_root_of_unity(uint64_t n) noexcept { return lsr::prover_root_of_unity(lsr::kProverModulus, n); }

NttContext* lsr_cyclic_ntt_context_create(uint64_t q, uint32_t n, uint64_t omega, int device) noexcept {
    try {
        return lsr::create_cyclic_ntt_context(q, n, omega, device);
    } catch (...) {
        return nullptr;
    }
}
int lsr_ntt_context_is_cyclic(const NttContext* ctx) noexcept { return ctx && ctx->cyclic ? 1 : 0; }

int lsr_cyclic_ntt_forward_batch(const NttContext* ctx, uint64_t* values, size_t batch) noexcept {
    if (!ctx || !values || !ctx->cyclic) return -1;
    if (batch == 0) return 0;
    return guarded("lsr_cyclic_ntt_forward_batch", [&] { lsr::cyclic_host(*ctx, values, batch, false); });
}
int lsr_cyclic_ntt_inverse_batch(const NttContext* ctx, uint64_t* values, size_t batch) noexcept {
    if (!ctx || !values || !ctx->cyclic) return -1;
    if (batch == 0) return 0;
    return guarded("lsr_cyclic_ntt_inverse_batch", [&] { lsr::cyclic_host(*ctx, values, batch, true); });
}
int lsr_bit_reverse_device(uint64_t* d_out, const uint64_t* d_in, int logn, size_t batch, void* stream) noexcept {
    if (!d_out || !d_in || d_out == d_in || logn < 1 || logn > 31) return -1;
    return guarded("lsr_bit_reverse_device", [&] { lsr::launch_bit_reverse(d_out, d_in, logn, batch, static_cast<hipStream_t>(stream)); });
}

LsrQuotientPlan* lsr_quotient_plan_create(uint32_t m, int device) noexcept {
    try {
        return lsr::create_plan(m, device);
    } catch (...) {
        return nullptr;
    }
}
void lsr_quotient_plan_free(LsrQuotientPlan* plan) noexcept { lsr::destroy_plan(plan); }
uint32_t lsr_quotient_plan_size(const LsrQuotientPlan* plan) noexcept { return plan ? plan->m : 0; }

int lsr_quotient_batch(LsrQuotientPlan* plan, const uint64_t* a_evals, const uint64_t* b_evals, const uint64_t* c_evals, size_t batch,
                       uint64_t* quotient, uint32_t* quotient_len) noexcept {
    if (!plan || !a_evals || !b_evals || !c_evals || !quotient || !quotient_len) return -1;
    if (batch == 0) return 0;
    return guarded("lsr_quotient_batch", [&] { lsr::quotient_host(*plan, a_evals, b_evals, c_evals, batch, quotient, quotient_len); });
}
int lsr_quotient_batch_device(LsrQuotientPlan* plan, const uint64_t* d_a, const uint64_t* d_b, const uint64_t* d_c, size_t batch, uint64_t* d_quotient,
                              uint32_t* d_quotient_len, void* stream) noexcept {
    if (!plan || !d_a || !d_b || !d_c || !d_quotient || !d_quotient_len) return -1;
    if (batch == 0) return 0;
    return guarded("lsr_quotient_batch_device",
                   [&] { lsr::quotient_device(*plan, d_a, d_b, d_c, batch, d_quotient, d_quotient_len, static_cast<hipStream_t>(stream)); });
}

LsrR1csProver* lsr_r1cs_prover_create(const SparseMatrix* A, const SparseMatrix* B, const SparseMatrix* C, int device) noexcept {
    if (!A || !B || !C) return nullptr;
    try {
        const SparseMatrix* const mats[3] = {A, B, C};
        return lsr::create_prover(mats, device);
    } catch (...) {
        return nullptr;
    }
}
void lsr_r1cs_prover_free(LsrR1csProver* prover) noexcept { lsr::destroy_prover(prover); }
uint32_t lsr_r1cs_prover_num_constraints(const LsrR1csProver* prover) noexcept { return prover ? prover->m : 0; }
uint32_t lsr_r1cs_prover_num_variables(const LsrR1csProver* prover) noexcept { return prover ? prover->n_vars : 0; }

int lsr_r1cs_constraint_evals_batch(LsrR1csProver* prover, const uint64_t* witnesses, size_t batch, uint64_t* a_evals, uint64_t* b_evals,
                                    uint64_t* c_evals) noexcept {
    if (!prover || !witnesses || !a_evals || !b_evals || !c_evals) return -1;
    if (batch == 0) return 0;
    return guarded("lsr_r1cs_constraint_evals_batch", [&] {
        uint64_t* const evals[3] = {a_evals, b_evals, c_evals};
        lsr::prover_run(*prover, witnesses, batch, evals, nullptr, nullptr);
    });
}
int lsr_r1cs_quotient_batch(LsrR1csProver* prover, const uint64_t* witnesses, size_t batch, uint64_t* quotient, uint32_t* quotient_len) noexcept {
    if (!prover || !witnesses || !quotient || !quotient_len) return -1;
    if (batch == 0) return 0;
    return guarded("lsr_r1cs_quotient_batch", [&] { lsr::prover_run(*prover, witnesses, batch, nullptr, quotient, quotient_len); });
}

}  // extern "C"
