// Device arithmetic for residues mod q — two flavours behind one butterfly interface.
//
//  * ArithF64 (q < 2^45): residues are kept as IEEE doubles holding exact integers.  A modular product
//    is an exact FP64-FMA Barrett reduction (6 full-rate v_*_f64 instructions, no integer multiplies):
//        h = x*w            (rounded high part)          l = fma(x, w, -h)   (exact low part)
//        k = rint(h / q)    (via h * (1/q))              r = fma(-k, q, h) + l
//    r == x*w (mod q) EXACTLY, with |r| <= 0.875 q, provided |x| < 2^50 and q < 2^45 (proof: DESIGN.md §4).
//    Forward CT butterflies need no per-stage correction (|value| grows by < q per stage, 19 q < 2^50);
//    inverse GS butterflies re-centre every value once per radix-16 round.  The final store maps to the canonical representative in [0,q), so the result
//    is bit-identical to the reference's Harvey/Shoup arithmetic (ntt.cpp:84,99 via SEAL).
//    Measured on MI355X (tools/ubench_arith.hip): 42 cycles per wave-butterfly vs 99 for u64 Shoup.
//  * ArithU64 (any q < 2^61): SEAL's lazy Harvey butterflies with Shoup multiplication, values in
//    [0,4q) forward / [0,2q) inverse — the restated reference algorithm itself.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

namespace lsr {

struct ModParams {
    uint64_t q;        // modulus
    uint64_t two_q;    // 2q
    double qd;         // (double) q
    double inv_qd;     // 1.0 / q rounded to nearest
    uint64_t barrett_hi, barrett_lo;   // floor(2^128 / q) for the pointwise kernel
    int logn;
};

// ---------------------------------------------------------------------------------------------
// exact conversions u64 <-> f64 for integers below 2^52 (one integer op + one FP add each)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double f64_from_u52(uint64_t x) {
    const uint64_t bits = (x & 0x000FFFFFFFFFFFFFull) | 0x4330000000000000ull;   // 2^52 + x
    return __longlong_as_double((long long)bits) - 4503599627370496.0;
}
__device__ __forceinline__ uint64_t u52_from_f64(double v) {   // v an integer in [0, 2^52)
    return (uint64_t)__double_as_longlong(v + 4503599627370496.0) & 0x000FFFFFFFFFFFFFull;
}

__device__ __forceinline__ double mulmod_f64(double x, double w, double q, double inv_q) {
    const double h = x * w;
    const double l = __builtin_fma(x, w, -h);
    const double k = __builtin_rint(h * inv_q);
    const double d = __builtin_fma(-k, q, h);
    return d + l;
}
// bring |v| < 2^50 back to |r| <= q/2 (+1)
__device__ __forceinline__ double recentre_f64(double v, double q, double inv_q) {
    const double k = __builtin_rint(v * inv_q);
    return __builtin_fma(-k, q, v);
}
// Canonical representative in [0,q) of an integer |v| <= 32 q, q < 2^45, in three instructions and no select:
// k = floor(v/q + 2^-46), r = v - k q.  Write v = K q + rho (0 <= rho < q).  The computed v*(1/q) + 2^-46 differs from
// K + rho/q + 2^-46 by less than 64 * 2^-53 = 2^-47 (two roundings of relative size 2^-53 on a value below 32), so it
// lies above K (2^-46 > 2^-47) and below K + 1 (1/q > 2^-45 > 2^-46 + 2^-47): the floor is exactly K and the FMA
// returns rho exactly.
__device__ __forceinline__ double canonical_f64(double v, double q, double inv_q) {
    const double k = __builtin_floor(__builtin_fma(v, inv_q, 0x1p-46));
    return __builtin_fma(-k, q, v);
}

// ---------------------------------------------------------------------------------------------
// buffer-resource addressing: base in SGPRs, one 32-bit lane offset, compile-time offsets in the scalar
// operand — no per-access VALU address arithmetic, and out-of-range lanes are dropped by the hardware.
// ---------------------------------------------------------------------------------------------
using rsrc_t = __amdgpu_buffer_rsrc_t;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// AUX: cache policy immediate of the buffer instruction; kAuxStream = "nt" (streaming: the line is not kept for re-use).
// Measured on the two-pass pattern (tools/ubench_move2.hip, profiles/r02_ubench_move2.txt): nt loads + nt stores on the LAST
// pass of a transform move the data 3 % (n = 2^16) to 10 % (single pass) faster; on a first pass they are slower.
#ifndef LSR_NT_LAST_PASS
#define LSR_NT_LAST_PASS 1
#endif
constexpr int kAuxStream = LSR_NT_LAST_PASS ? 2 : 0;
#ifndef LSR_NT_COMMIT_INPUTS            // streaming loads of arrays read exactly once: out-of-place operands, blinding residues
#define LSR_NT_COMMIT_INPUTS 1
#endif
template <int AUX = 0>
__device__ __forceinline__ uint64_t buf_load64(rsrc_t r, uint32_t lane_bytes, uint32_t const_bytes) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)lane_bytes, (int)const_bytes, AUX);
    return ((uint64_t)v.y << 32) | v.x;
}
__device__ __forceinline__ void buf_load128(rsrc_t r, uint32_t lane_bytes, uint32_t const_bytes, uint64_t& a, uint64_t& b) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_bytes, (int)const_bytes, 0);
    a = ((uint64_t)v.y << 32) | v.x;
    b = ((uint64_t)v.w << 32) | v.z;
}
template <int AUX = 0>
__device__ __forceinline__ void buf_store64(rsrc_t r, uint32_t lane_bytes, uint32_t const_bytes, uint64_t x) {
    u32x2 v;
    v.x = (uint32_t)x;
    v.y = (uint32_t)(x >> 32);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)lane_bytes, (int)const_bytes, AUX);
}

struct ArithF64 {
    static constexpr bool kUnitTopTwiddles = false;   // (cyclic F64 / u64 contexts exist, but only the Goldilocks kernels are product-bound)
    using elem = double;     // residue (exact integer in a double)
    using twid = double;     // twiddle (canonical, as double)

    static __device__ __forceinline__ elem load(uint64_t x, const ModParams&) { return f64_from_u52(x); }
    static __device__ __forceinline__ uint64_t store_canonical(elem v, const ModParams& p) {
        return u52_from_f64(canonical_f64(v, p.qd, p.inv_qd));
    }
    // value already known to satisfy |v| < q (a fresh mulmod result)
    static __device__ __forceinline__ uint64_t store_reduced(elem v, const ModParams& p) {
        return u52_from_f64(v < 0.0 ? v + p.qd : v);
    }
    // canonical (v + e) mod q for |v| < q and a canonical residue e (the fused blinding add)
    // v + e is an exact integer in (-q, 2q): the select-free canonical step (3 instructions) replaces two compare/select pairs —
    // 8 VALU instructions per residue instead of 14 in the inverse round that carries the add
    static __device__ __forceinline__ uint64_t store_reduced_plus(elem v, uint64_t e, const ModParams& p) {
        return u52_from_f64(canonical_f64(v + f64_from_u52(e), p.qd, p.inv_qd));
    }
    static __device__ __forceinline__ twid load_tw(const double* table, uint32_t idx) { return table[idx]; }
    // COUNT consecutive table entries starting at idx (COUNT a power of two, idx a multiple of COUNT)
    template <int COUNT>
    static __device__ __forceinline__ void load_tw_run(rsrc_t table, uint32_t idx, twid* out) {
        if constexpr (COUNT == 1) {
            out[0] = __longlong_as_double((long long)buf_load64(table, idx * 8u, 0));
        } else {
#pragma unroll
            for (int u = 0; u < COUNT; u += 2) {
                uint64_t a, b;
                buf_load128(table, idx * 8u, (uint32_t)u * 8u, a, b);
                out[u] = __longlong_as_double((long long)a);
                out[u + 1] = __longlong_as_double((long long)b);
            }
        }
    }

    // Cooley–Tukey: (x, y) <- (x + w y, x - w y)
    static __device__ __forceinline__ void ct(elem& x, elem& y, twid w, const ModParams& p) {
        const double t = mulmod_f64(y, w, p.qd, p.inv_qd);
        const double a = x;
        x = a + t;
        y = a - t;
    }
    // Gentleman–Sande: (x, y) <- (x + y, (x - y) w)
    static __device__ __forceinline__ void gs(elem& x, elem& y, twid w, const ModParams& p) {
        const double a = x, b = y;
        x = a + b;
        y = mulmod_f64(a - b, w, p.qd, p.inv_qd);
    }
    // last inverse stage with n^-1 folded in: (x, y) <- ((x + y) ninv, (x - y) (w ninv))
    static __device__ __forceinline__ void gs_scaled(elem& x, elem& y, twid w_scaled, twid n_inv, const ModParams& p) {
        const double a = x, b = y;
        x = mulmod_f64(a + b, n_inv, p.qd, p.inv_qd);
        y = mulmod_f64(a - b, w_scaled, p.qd, p.inv_qd);
    }
    static __device__ __forceinline__ void end_of_inverse_round(elem& v, const ModParams& p) { v = recentre_f64(v, p.qd, p.inv_qd); }
    // Which outputs of an R-stage Gentleman–Sande round (register bit j = 1: the product output of stage j) must be re-centred
    // before the NEXT round may take them as inputs.  Contract between rounds: every input satisfies |x| <= 2 q.  Then the input of
    // the product of stage j is a difference of two sums of 2^j inputs, at most 2 * 2^j * 2 q <= 32 q < 2^50 (R <= 4, q < 2^45): exact
    // (DESIGN.md §4).  A product output is at most 0.875 q (=: P) and doubles with every later sum stage, so after the round
    //   bit R-1 set: P        bit R-1 clear, R-2 set: 2 P = 1.75 q        both clear: 4 P, 8 P or the all-sum 2^R * 2 q = 32 q.
    // Only the last class exceeds 2 q: one register in four (round 3: all were re-centred, 15 % of the inverse tile pass).
    static constexpr bool kPartialRecentre = true;
    template <int R>
    static constexpr bool needs_recentre(int stage_bits) {
        return R == 1 ? (stage_bits & 1) == 0 : (stage_bits & (3 << (R - 2))) == 0;
    }
};

// ---------------------------------------------------------------------------------------------
// u64 Harvey / Shoup (SEAL dwthandler.h Arithmetic<uint64_t, MultiplyUIntModOperand, ...>)
// ---------------------------------------------------------------------------------------------
struct ShoupOperand {
    uint64_t w, wq;
};

__device__ __forceinline__ uint64_t mul_shoup_lazy(uint64_t x, ShoupOperand o, uint64_t q) {
    return x * o.w - __umul64hi(x, o.wq) * q;   // in [0, 2q)
}

struct ArithU64 {
    static constexpr bool kUnitTopTwiddles = false;   // (cyclic F64 / u64 contexts exist, but only the Goldilocks kernels are product-bound)
    using elem = uint64_t;
    using twid = ShoupOperand;

    static __device__ __forceinline__ elem load(uint64_t x, const ModParams&) { return x; }
    static __device__ __forceinline__ uint64_t store_canonical(elem v, const ModParams& p) {
        if (v >= p.two_q) v -= p.two_q;
        if (v >= p.q) v -= p.q;
        return v;
    }
    static __device__ __forceinline__ uint64_t store_reduced(elem v, const ModParams& p) { return v >= p.q ? v - p.q : v; }
    static __device__ __forceinline__ uint64_t store_reduced_plus(elem v, uint64_t e, const ModParams& p) {
        uint64_t s = (v >= p.q ? v - p.q : v) + e;
        return s >= p.q ? s - p.q : s;
    }
    static __device__ __forceinline__ twid load_tw(const ShoupOperand* table, uint32_t idx) { return table[idx]; }
    template <int COUNT>
    static __device__ __forceinline__ void load_tw_run(rsrc_t table, uint32_t idx, twid* out) {
#pragma unroll
        for (int u = 0; u < COUNT; ++u) buf_load128(table, idx * 16u, (uint32_t)u * 16u, out[u].w, out[u].wq);
    }

    static __device__ __forceinline__ void ct(elem& x, elem& y, twid w, const ModParams& p) {
        const uint64_t u = x >= p.two_q ? x - p.two_q : x;
        const uint64_t v = mul_shoup_lazy(y, w, p.q);
        x = u + v;
        y = u + p.two_q - v;
    }
    static __device__ __forceinline__ void gs(elem& x, elem& y, twid w, const ModParams& p) {
        const uint64_t u = x, v = y;
        const uint64_t s = u + v;
        x = s >= p.two_q ? s - p.two_q : s;
        y = mul_shoup_lazy(u + p.two_q - v, w, p.q);
    }
    static __device__ __forceinline__ void gs_scaled(elem& x, elem& y, twid w_scaled, twid n_inv, const ModParams& p) {
        const uint64_t u = x >= p.two_q ? x - p.two_q : x, v = y;
        uint64_t s = u + v;
        s = s >= p.two_q ? s - p.two_q : s;
        x = mul_shoup_lazy(s, n_inv, p.q);
        y = mul_shoup_lazy(u + p.two_q - v, w_scaled, p.q);
    }
    static __device__ __forceinline__ void end_of_inverse_round(elem&, const ModParams&) {}
    static constexpr bool kPartialRecentre = false;
    template <int R> static constexpr bool needs_recentre(int) { return true; }
};

// ---------------------------------------------------------------------------------------------
// Goldilocks p = 2^64 - 2^32 + 1 — the prover's NTT field (rust-api/lambda-snark-core/src/lib.rs:58).  Above the 2^61
// limit of the Shoup flavour; uses 2^64 = 2^32 - 1 and 2^96 = -1 (mod p).  Everything stays canonical.
// ---------------------------------------------------------------------------------------------
constexpr uint64_t kGoldilocks = 0xFFFFFFFF00000001ull;
constexpr uint64_t kGoldEpsilon = 0xFFFFFFFFull;   // 2^32 - 1 = 2^64 mod p

__device__ __forceinline__ uint64_t gold_add(uint64_t a, uint64_t b) {   // a, b < p
    unsigned long long s;
    const bool carry = __builtin_uaddll_overflow(a, b, &s);
    return (carry || s >= kGoldilocks) ? s - kGoldilocks : s;              // on carry, s - p wraps to the right residue
}
__device__ __forceinline__ uint64_t gold_sub(uint64_t a, uint64_t b) {
    unsigned long long d;
    const bool borrow = __builtin_usubll_overflow(a, b, &d);
    return borrow ? d + kGoldilocks : d;
}
__device__ __forceinline__ uint64_t gold_mul(uint64_t a, uint64_t b) {
    const unsigned __int128 wide = (unsigned __int128)a * b;
    const uint64_t lo = (uint64_t)wide, hi = (uint64_t)(wide >> 64);
    unsigned long long t0;
    const bool borrow = __builtin_usubll_overflow(lo, hi >> 32, &t0);      // hi_hi 2^96 = -hi_hi
    t0 -= borrow ? kGoldEpsilon : 0ull;                                    // borrow: -2^64 = -(2^32 - 1)
    // hi_lo * 2^64 = hi_lo * (2^32 - 1) = (hi_lo << 32) - hi_lo, spelled in 32-bit halves so that it stays off the multiplier
    const uint32_t h32 = (uint32_t)hi;
    const uint64_t t1 = ((uint64_t)(h32 - (h32 != 0u)) << 32) | (uint64_t)(0u - h32);
    unsigned long long r;
    const bool carry = __builtin_uaddll_overflow(t0, t1, &r);
    r += carry ? kGoldEpsilon : 0ull;                                      // carry: +2^64 = +(2^32 - 1)
    return r >= kGoldilocks ? r - kGoldilocks : r;
}

// x * w mod p for a multiplier held in MONTGOMERY form (w_mont = w 2^64 mod p; every table and constant of the Goldilocks
// kernels is stored that way): with p^-1 = 1 + 2^32 (mod 2^64) the reduction needs no multiplication — m = lo p^-1,
// (x w_mont - m p) / 2^64 = hi - (m p >> 64) in (-p, p).  Any 64-bit x; canonical result.  13 % fewer cycles per butterfly
// than gold_mul (tools/ubench_arith: 124 vs 142).
__device__ __forceinline__ uint64_t gold_mul_mont(uint64_t x, uint64_t w_mont) {
    const unsigned __int128 wide = (unsigned __int128)x * w_mont;
    const uint64_t lo = (uint64_t)wide, hi = (uint64_t)(wide >> 64);
    const uint64_t m = lo + (lo << 32);                                    // lo * (1 + 2^32) mod 2^64
    const uint64_t u = m - (m >> 32) - (uint64_t)(m < (m << 32));          // high word of m * (2^64 - 2^32 + 1)
    unsigned long long t;
    const bool borrow = __builtin_usubll_overflow(hi, u, &t);
    return borrow ? t + kGoldilocks : t;
}

struct ArithGold {
    using elem = uint64_t;
    using twid = uint64_t;   // Montgomery form

    static __device__ __forceinline__ elem load(uint64_t x, const ModParams&) { return x >= kGoldilocks ? x - kGoldilocks : x; }
    // forward (Cooley–Tukey) values are LAZY: any 64-bit representative (round 3, see ct below); one conditional subtraction
    // canonicalises, since 2^64 < 2 p
    static __device__ __forceinline__ uint64_t store_canonical(elem v, const ModParams&) { return v >= kGoldilocks ? v - kGoldilocks : v; }
    static __device__ __forceinline__ uint64_t store_reduced(elem v, const ModParams&) { return v; }
    static __device__ __forceinline__ uint64_t store_reduced_plus(elem v, uint64_t e, const ModParams&) {
        return gold_add(v, e >= kGoldilocks ? e - kGoldilocks : e);
    }
    static __device__ __forceinline__ elem pre_mul(elem v, uint64_t d_mont, const ModParams&) { return gold_mul_mont(v, d_mont); }
    static __device__ __forceinline__ twid load_tw(const uint64_t* table, uint32_t idx) { return table[idx]; }
    template <int COUNT>
    static __device__ __forceinline__ void load_tw_run(rsrc_t table, uint32_t idx, twid* out) {
        if constexpr (COUNT == 1) {
            out[0] = buf_load64(table, idx * 8u, 0);
        } else {
#pragma unroll
            for (int u = 0; u < COUNT; u += 2) buf_load128(table, idx * 8u, (uint32_t)u * 8u, out[u], out[u + 1]);
        }
    }
    // Cooley–Tukey with LAZY sums: x may be any 64-bit representative, the product t is canonical (gold_mul_mont takes any
    // 64-bit multiplicand).  a + t wraps at most once (t < p), and the wrapped sum a + t - 2^64 < p - 1 leaves room for the
    // correction + (2^32 - 1) = + 2^64 mod p; a - t borrows at most once, and the wrapped difference a - t + 2^64 > 2^64 - p =
    // 2^32 - 2 leaves room for - (2^32 - 1).  No comparison with p, no second correction: 110 instead of 123 cycles per
    // wave-butterfly in tools/ubench_arith ("goldilocks mont lazy").  The inverse (Gentleman–Sande) butterflies add two lazy values
    // and stay canonical.
    static __device__ __forceinline__ void ct(elem& x, elem& y, twid w, const ModParams&) {
        const uint64_t t = gold_mul_mont(y, w), a = x;
        unsigned long long s, d;
        const bool carry = __builtin_uaddll_overflow(a, t, &s);
        const bool borrow = __builtin_usubll_overflow(a, t, &d);
        x = s + (carry ? kGoldEpsilon : 0ull);
        y = d - (borrow ? kGoldEpsilon : 0ull);
    }
    static __device__ __forceinline__ void gs(elem& x, elem& y, twid w, const ModParams&) {
        const uint64_t a = x, b = y;
        x = gold_add(a, b);
        y = gold_mul_mont(gold_sub(a, b), w);
    }
    // butterflies whose twiddle is omega^0 = 1 (the u = 0 groups of a cyclic transform's top round, lsr_ntt_kernels.hpp): no product
    static constexpr bool kUnitTopTwiddles = true;          // Goldilocks contexts are cyclic (lsr_ntt.hip build_context)
    static __device__ __forceinline__ void ct_unit(elem& x, elem& y, const ModParams&) {
        const uint64_t a = x, t = y >= kGoldilocks ? y - kGoldilocks : y;      // ct's sums need a canonical second operand
        unsigned long long s, d;
        const bool carry = __builtin_uaddll_overflow(a, t, &s);
        const bool borrow = __builtin_usubll_overflow(a, t, &d);
        x = s + (carry ? kGoldEpsilon : 0ull);
        y = d - (borrow ? kGoldEpsilon : 0ull);
    }
    static __device__ __forceinline__ void gs_unit(elem& x, elem& y, const ModParams&) {
        const uint64_t a = x, b = y;
        x = gold_add(a, b);
        y = gold_sub(a, b);
    }
    // (both constants equal to 1 — Montgomery form 2^32 - 1 — is how a context says "leave the n^-1 out": the quotient plan folds that
    // factor into a table of its own, lsr_prover.hip; a wave-uniform branch, the last stage then costs no product)
    static __device__ __forceinline__ void gs_scaled(elem& x, elem& y, twid w_scaled, twid n_inv, const ModParams&) {
        const uint64_t a = x, b = y;
        if (n_inv == kGoldEpsilon && w_scaled == kGoldEpsilon) {
            x = gold_add(a, b);
            y = gold_sub(a, b);
        } else {
            x = gold_mul_mont(gold_add(a, b), n_inv);
            y = gold_mul_mont(gold_sub(a, b), w_scaled);
        }
    }
    static __device__ __forceinline__ void end_of_inverse_round(elem&, const ModParams&) {}
    static constexpr bool kPartialRecentre = false;
    template <int R> static constexpr bool needs_recentre(int) { return true; }
};

// canonical (a*b) mod q for ANY 64-bit a, b (q < 2^61): 128-bit product + Barrett with floor(2^128/q)
// (same contract as SEAL multiply_uint_mod used at ntt.cpp:117).
__device__ __forceinline__ uint64_t mulmod_barrett128(uint64_t a, uint64_t b, const ModParams& p) {
    const uint64_t lo = a * b, hi = __umul64hi(a, b);
    // q_hat = floor( (hi:lo) * (bhi:blo) / 2^128 ), using the three significant partial products
    const uint64_t c1 = __umul64hi(lo, p.barrett_lo);
    const uint64_t m1_lo = lo * p.barrett_hi, m1_hi = __umul64hi(lo, p.barrett_hi);
    const uint64_t m2_lo = hi * p.barrett_lo, m2_hi = __umul64hi(hi, p.barrett_lo);
    uint64_t s = c1 + m1_lo;
    uint64_t carry = s < c1;
    const uint64_t s2 = s + m2_lo;
    carry += s2 < s;
    const uint64_t q_hat = hi * p.barrett_hi + m1_hi + m2_hi + carry;
    uint64_t r = lo - q_hat * p.q;            // exact remainder is r, r-q or r-2q
    if (r >= p.q) r -= p.q;
    if (r >= p.q) r -= p.q;
    return r;
}

}  // namespace lsr
