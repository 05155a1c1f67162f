// Microbenchmark: which modular-multiply formulation is fastest on gfx950?
// (SURVEY.md §7 H1).  Standalone: hipcc --offload-arch=gfx950 -O3 tools/ubench_arith.hip -o ubench
// Every variant runs ILP independent chains per lane, ITERS times; we report
// wave-instructions-equivalent "ops"/s over the whole chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#include "lsr_arith.hpp"   // the library's Goldilocks arithmetic (tools/Makefile adds the include path)

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int ILP = 8;
constexpr int ITERS = 2048;

static __device__ __forceinline__ uint64_t mulhi64(uint64_t a, uint64_t b) { return __umul64hi(a, b); }

// ---- raw instruction throughput ------------------------------------------------------------
template <int OP>
__global__ void __launch_bounds__(256) k_raw(uint64_t* out, uint64_t seed) {
    uint64_t x[ILP];
    double d[ILP];
    uint32_t u[ILP];
    const uint64_t t = seed + blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < ILP; ++i) { x[i] = t * 0x9E3779B97F4A7C15ull + i; d[i] = (double)(x[i] >> 20); u[i] = (uint32_t)x[i]; }
    const uint32_t cu = (uint32_t)seed | 1u;
    const double cd = 1.0000001 + (double)(seed & 3);
    const double ce = 0.25;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            if (OP == 0) { // v_mad_u64_u32
                x[i] = (uint64_t)(uint32_t)x[i] * cu + x[i];
            } else if (OP == 1) { // v_mul_lo_u32
                u[i] = u[i] * cu;
            } else if (OP == 2) { // v_mul_hi_u32
                u[i] = __umulhi(u[i], cu) + 1u;   // +1 folded? keep: extra add
            } else if (OP == 3) { // v_mul_u32_u24
                u[i] = __umul24(u[i], cu);
            } else if (OP == 4) { // v_fma_f64
                d[i] = __builtin_fma(d[i], cd, ce);
            } else if (OP == 5) { // v_mul_f64
                d[i] = d[i] * cd;
            } else if (OP == 6) { // v_add_f64
                d[i] = d[i] + cd;
            } else if (OP == 7) { // v_rndne_f64
                d[i] = __builtin_rint(d[i]) + 0.0;
                d[i] = d[i] * cd;
            } else if (OP == 8) { // 64-bit integer add (v_add_co + v_addc)
                x[i] = x[i] + (x[i] >> 1) ;
            } else if (OP == 9) { // v_fma_f32 reference
                float f = __builtin_bit_cast(float, u[i]);
                f = __builtin_fmaf(f, 1.0001f, 0.5f);
                u[i] = __builtin_bit_cast(uint32_t, f);
            } else if (OP == 10) { // 64x64 -> hi64
                x[i] = mulhi64(x[i], seed | 0x8000000000000001ull) + 1;
            } else if (OP == 11) { // 64x64 -> lo64
                x[i] = x[i] * (seed | 0x8000000000000001ull) + 1;
            } else if (OP == 12) { // v_floor_f64
                d[i] = __builtin_floor(d[i]);
                d[i] = d[i] * cd;
            } else if (OP == 13) { // mul_hi_u32_u24
                u[i] = __umul24(u[i], cu) ^ __umul24(u[i] >> 3, cu);
            }
        }
    }
    uint64_t acc = 0;
    for (int i = 0; i < ILP; ++i) acc += x[i] + (uint64_t)d[i] + u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// ---- complete CT butterflies ---------------------------------------------------------------
// (A) SEAL-style Harvey lazy butterfly, u64 Shoup multiply.
__global__ void __launch_bounds__(256) k_bfly_shoup(uint64_t* out, uint64_t q, uint64_t w, uint64_t wq) {
    uint64_t X[ILP], Y[ILP];
    const uint64_t t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < ILP; ++i) { X[i] = (t * 0x9E3779B97F4A7C15ull + i) % q; Y[i] = (t * 0xBF58476D1CE4E5B9ull + 7 * i) % q; }
    const uint64_t two_q = 2 * q;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            uint64_t u = X[i] - (X[i] >= two_q ? two_q : 0);
            uint64_t h = mulhi64(Y[i], wq);
            uint64_t v = Y[i] * w - h * q;
            X[i] = u + v;
            Y[i] = u + two_q - v;
        }
    }
    uint64_t acc = 0;
    for (int i = 0; i < ILP; ++i) acc += X[i] ^ Y[i];
    out[t] = acc;
}

// (B) exact FP64-FMA Barrett butterfly, values kept as doubles, no per-stage correction.
__global__ void __launch_bounds__(256) k_bfly_f64(uint64_t* out, double q, double w, double invq) {
    double X[ILP], Y[ILP];
    const uint64_t t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < ILP; ++i) { X[i] = (double)((t * 0x9E3779B97F4A7C15ull + i) >> 21); Y[i] = (double)((t * 0xBF58476D1CE4E5B9ull + 7 * i) >> 21); }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            double h = Y[i] * w;
            double l = __builtin_fma(Y[i], w, -h);
            double k = __builtin_rint(h * invq);
            double d = __builtin_fma(-k, q, h);
            double r = d + l;
            double x = X[i];
            // keep magnitudes bounded like a real NTT (every 16 stages a canonicalisation): emulate by
            // damping X each iteration with a cheap exact op that the real kernel does not have (conservative).
            X[i] = (x + r) * 0.5;
            Y[i] = (x - r);
        }
    }
    double acc = 0;
    for (int i = 0; i < ILP; ++i) acc += X[i] + Y[i];
    out[t] = (uint64_t)acc;
}

// (G) Goldilocks (2^64 - 2^32 + 1) butterfly of the prover path: the library's gold_mul / gold_add / gold_sub.
__global__ void __launch_bounds__(256) k_bfly_gold(uint64_t* out, uint64_t w) {
    uint64_t X[ILP], Y[ILP];
    const uint64_t t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < ILP; ++i) { X[i] = (t * 0x9E3779B97F4A7C15ull + i) >> 1; Y[i] = (t * 0xBF58476D1CE4E5B9ull + 7 * i) >> 1; }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            const uint64_t r = lsr::gold_mul(Y[i], w), x = X[i];
            X[i] = lsr::gold_add(x, r);
            Y[i] = lsr::gold_sub(x, r);
        }
    }
    uint64_t acc = 0;
    for (int i = 0; i < ILP; ++i) acc += X[i] ^ Y[i];
    out[t] = acc;
}

// (H) Goldilocks butterfly with a Montgomery-form twiddle: x * (w 2^64 mod p) reduced with p^-1 = 1 + 2^32 (mod 2^64),
// no multiplication in the reduction, canonical result.
__global__ void __launch_bounds__(256) k_bfly_gold_lazy(uint64_t* out, uint64_t w) {   // x kept in [0, 2^64), t canonical
    uint64_t X[ILP], Y[ILP];
    const uint64_t t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < ILP; ++i) { X[i] = (t * 0x9E3779B97F4A7C15ull + i) >> 1; Y[i] = (t * 0xBF58476D1CE4E5B9ull + 7 * i) >> 1; }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            const uint64_t r = lsr::gold_mul_mont(Y[i], w), x = X[i];
            unsigned long long s, d;
            const bool c = __builtin_uaddll_overflow(x, r, &s);
            const bool b = __builtin_usubll_overflow(x, r, &d);
            X[i] = s + (c ? lsr::kGoldEpsilon : 0ull);
            Y[i] = d - (b ? lsr::kGoldEpsilon : 0ull);
        }
    }
    uint64_t acc = 0;
    for (int i = 0; i < ILP; ++i) acc += X[i] ^ Y[i];
    out[t] = acc;
}
__global__ void __launch_bounds__(256) k_bfly_gold_mont(uint64_t* out, uint64_t w) {
    uint64_t X[ILP], Y[ILP];
    const uint64_t t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < ILP; ++i) { X[i] = (t * 0x9E3779B97F4A7C15ull + i) >> 1; Y[i] = (t * 0xBF58476D1CE4E5B9ull + 7 * i) >> 1; }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            const uint64_t r = lsr::gold_mul_mont(Y[i], w), x = X[i];
            X[i] = lsr::gold_add(x, r);
            Y[i] = lsr::gold_sub(x, r);
        }
    }
    uint64_t acc = 0;
    for (int i = 0; i < ILP; ++i) acc += X[i] ^ Y[i];
    out[t] = acc;
}

// (C) FP64 butterfly using magic-constant rounding instead of v_rndne.
__global__ void __launch_bounds__(256) k_bfly_f64_magic(uint64_t* out, double q, double w, double invq) {
    double X[ILP], Y[ILP];
    const uint64_t t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < ILP; ++i) { X[i] = (double)((t * 0x9E3779B97F4A7C15ull + i) >> 21); Y[i] = (double)((t * 0xBF58476D1CE4E5B9ull + 7 * i) >> 21); }
    const double MAGIC = 6755399441055744.0; // 1.5 * 2^52
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            double h = Y[i] * w;
            double l = __builtin_fma(Y[i], w, -h);
            double k = __builtin_fma(h, invq, MAGIC) - MAGIC;
            double d = __builtin_fma(-k, q, h);
            double r = d + l;
            double x = X[i];
            X[i] = (x + r) * 0.5;
            Y[i] = (x - r);
        }
    }
    double acc = 0;
    for (int i = 0; i < ILP; ++i) acc += X[i] + Y[i];
    out[t] = (uint64_t)acc;
}

// (D) u64 butterfly with a truncated Shoup quotient (3 32-bit multiplies) and a 48-bit remainder.
__global__ void __launch_bounds__(256) k_bfly_shoup_trunc(uint64_t* out, uint64_t q, uint64_t w, uint64_t wq) {
    uint64_t X[ILP], Y[ILP];
    const uint64_t t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < ILP; ++i) { X[i] = (t * 0x9E3779B97F4A7C15ull + i) % q; Y[i] = (t * 0xBF58476D1CE4E5B9ull + 7 * i) % q; }
    const uint64_t eight_q = 8 * q;
    const uint32_t wq1 = (uint32_t)(wq >> 32), wq0 = (uint32_t)wq;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            uint64_t u = X[i] - (X[i] >= eight_q ? eight_q : 0);
            uint64_t y = Y[i] - (Y[i] >= eight_q ? eight_q : 0);
            uint32_t y1 = (uint32_t)(y >> 32), y0 = (uint32_t)y;
            uint64_t h = (uint64_t)y1 * wq1 + (uint64_t)__umulhi(y0, wq1) + (uint64_t)__umulhi(y1, wq0);
            uint64_t v = y * w - h * q;   // in [0, 5q)
            X[i] = u + v;
            Y[i] = u + eight_q - v;
        }
    }
    uint64_t acc = 0;
    for (int i = 0; i < ILP; ++i) acc += X[i] ^ Y[i];
    out[t] = acc;
}

template <typename F>
static int timeit(const char* name, double ops_per_thread, F launch) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int blocks = 256 * 8, threads = 256;
    launch(blocks, threads);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(a));
        launch(blocks, threads);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    double total = ops_per_thread * blocks * threads;
    printf("%-28s %8.3f ms  %9.2f Gops/s (lane-ops)  -> %6.2f cyc/wave-op/SIMD @2.4GHz\n", name, best, total / best * 1e-6,
           (256.0 * 4 * 2.4e9) / (total / 64.0 / (best * 1e-3)));
    return 0;
}

int main() {
    uint64_t* d_out;
    CK(hipMalloc(&d_out, sizeof(uint64_t) * 256 * 8 * 256));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s, CUs %d, clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    const double per = (double)ILP * ITERS;
    const char* names[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32(+add)", "v_mul_u32_u24", "v_fma_f64", "v_mul_f64", "v_add_f64",
                           "v_rndne_f64(+mul)", "add_u64(+shift)", "v_fma_f32", "mulhi64(+add)", "mullo64(+add)", "v_floor_f64(+mul)", "mul_u24 x2 + xor"};
#define RAW(OP) timeit(names[OP], per, [&](int g, int b) { hipLaunchKernelGGL(k_raw<OP>, dim3(g), dim3(b), 0, 0, d_out, 12345ull); });
    RAW(0) RAW(1) RAW(2) RAW(3) RAW(4) RAW(5) RAW(6) RAW(7) RAW(8) RAW(9) RAW(10) RAW(11) RAW(12) RAW(13)
    const uint64_t q = 17592182243329ull, w = 161788235ull;
    const uint64_t wq = (uint64_t)(((unsigned __int128)w << 64) / q);
    printf("--- butterflies (ops = butterflies) ---\n");
    timeit("bfly shoup u64 (Harvey)", per, [&](int g, int b) { hipLaunchKernelGGL(k_bfly_shoup, dim3(g), dim3(b), 0, 0, d_out, q, w, wq); });
    timeit("bfly shoup trunc u64", per, [&](int g, int b) { hipLaunchKernelGGL(k_bfly_shoup_trunc, dim3(g), dim3(b), 0, 0, d_out, q, w, wq); });
    timeit("bfly f64 rndne", per, [&](int g, int b) { hipLaunchKernelGGL(k_bfly_f64, dim3(g), dim3(b), 0, 0, d_out, (double)q, (double)w, 1.0 / (double)q); });
    timeit("bfly goldilocks u64", per, [&](int g, int b) { hipLaunchKernelGGL(k_bfly_gold, dim3(g), dim3(b), 0, 0, d_out, 0x0123456789ABCDEFull); });
    timeit("bfly goldilocks mont lazy", per, [&](int g, int b) { hipLaunchKernelGGL(k_bfly_gold_lazy, dim3(g), dim3(b), 0, 0, d_out, 0x0123456789ABCDEFull); });
    timeit("bfly goldilocks montgomery", per, [&](int g, int b) { hipLaunchKernelGGL(k_bfly_gold_mont, dim3(g), dim3(b), 0, 0, d_out, 0x0123456789ABCDEFull); });
    timeit("bfly f64 magic", per, [&](int g, int b) { hipLaunchKernelGGL(k_bfly_f64_magic, dim3(g), dim3(b), 0, 0, d_out, (double)q, (double)w, 1.0 / (double)q); });
    printf("one n=2^16 NTT = 524288 butterflies: NTT/s = bfly_Gops * 1e9 / 524288\n");
    CK(hipFree(d_out));
    return 0;
}
