// Microbenchmark: where do the cycles of the seeded CDT Gaussian sampler go on gfx950, and which table-scan formulation is
// fastest?  (round 3: the sampler is 45 % of a full lwe_commit at n = 4096 and the bound of the e1-on-device pipeline.)
//   make -C tools bin/ubench_sampler && tools/bin/ubench_sampler
// Every kernel runs `iters` ChaCha blocks (8 samples each) per lane; rates are samples/s over the whole chip, reported for
// 1, 2, 4 and 8 waves per SIMD (blocks of 256 lanes per CU).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <cmath>

#include "lsr_sampler.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

using namespace lsr;

// the library's table for sigma = 3.19 is built on the host with long double (lsr_host_math.cpp); for the microbenchmark any
// non-decreasing table of the same length does (the scans are data-independent by construction)
static std::vector<uint64_t> make_table(double sigma, uint32_t* scan_entries) {
    const int bound = std::max(8, (int)std::ceil(12.0 * sigma));
    std::vector<long double> w(bound + 1);
    long double total = 0;
    for (int k = 0; k <= bound; ++k) { w[k] = (k == 0 ? 1.0L : 2.0L) * expl(-(long double)k * k / (2.0L * sigma * sigma)); total += w[k]; }
    std::vector<uint64_t> cdf(bound + 1);
    long double acc = 0;
    for (int k = 0; k <= bound; ++k) { acc += w[k]; long double v = acc / total * 18446744073709551615.0L; cdf[k] = v >= 18446744073709551615.0L ? ~0ull : (uint64_t)v; }
    cdf[bound] = ~0ull;
    uint32_t e = 0;
    while (e < cdf.size() && (cdf[e] >> 1) != (~0ull >> 1)) ++e;
    *scan_entries = e + 1;
    return cdf;
}

// ---- scan variants: magnitude[s] = #{k : cdf63[k] < u[s]} ---------------------------------------------------------------
// V1: 32-bit borrow chain (v_sub_co, v_subb_co, v_addc)
template <int COUNT>
__device__ __forceinline__ void scan_borrow(const uint64_t* cdf63, uint32_t entries, const uint64_t (&u)[COUNT], uint32_t (&mag)[COUNT]) {
#pragma unroll
    for (int s = 0; s < COUNT; ++s) mag[s] = 0;
    for (uint32_t k = 0; k + 1 < entries; ++k) {
        const uint64_t c = cdf63[k];
#pragma unroll
        for (int s = 0; s < COUNT; ++s) mag[s] += (uint32_t)((c - u[s]) >> 63);     // both below 2^63: the sign of the difference
    }
}
// V2: branch-free binary search over a 32-entry table held in the lanes of one VGPR pair (ds_bpermute gathers: the LDS crossbar,
// no LDS memory, no banks) — 5 steps of {2 bpermute, 1 compare, 2 integer ops}
template <int COUNT>
__device__ __forceinline__ void scan_bperm(uint32_t tab_lo, uint32_t tab_hi, const uint64_t (&u)[COUNT], uint32_t (&mag)[COUNT]) {
#pragma unroll
    for (int s = 0; s < COUNT; ++s) {
        int q = 15 * 4;                                  // byte address of the probe lane: (pos + step - 1) * 4
#pragma unroll
        for (int step = 16; step >= 1; step >>= 1) {
            const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(q, (int)tab_lo);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(q, (int)tab_hi);
            const uint64_t c = ((uint64_t)hi << 32) | lo;
            const bool below = c < u[s];
            if (step > 1) q += below ? 2 * step : -2 * step;
            else q += below ? 4 : 0;
        }
        mag[s] = (uint32_t)q >> 2;
    }
}
// V3: rate probe only (NOT exact): 32-bit compares of the high words
template <int COUNT>
__device__ __forceinline__ void scan_hi32(const uint64_t* cdf63, uint32_t entries, const uint64_t (&u)[COUNT], uint32_t (&mag)[COUNT]) {
#pragma unroll
    for (int s = 0; s < COUNT; ++s) mag[s] = 0;
    for (uint32_t k = 0; k + 1 < entries; ++k) {
        const uint32_t c = (uint32_t)(cdf63[k] >> 32);
#pragma unroll
        for (int s = 0; s < COUNT; ++s) mag[s] += (c < (uint32_t)(u[s] >> 32)) ? 1u : 0u;
    }
}

// MODE 0: cipher only; 1: cipher + library scan; 2: library scan only (u from a cheap recurrence); 3: borrow-chain scan only;
// 4: bpermute scan only; 5: hi32 scan only; 6: cipher + borrow scan; 7: cipher + bpermute scan
template <int MODE>
__global__ void __launch_bounds__(256) k_sampler(uint64_t* out, const uint64_t* key, const uint64_t* cdf_global, uint32_t entries, int iters) {
    __shared__ uint64_t cdf[64];
    for (uint32_t i = threadIdx.x; i < 64; i += 256) cdf[i] = i < entries ? cdf_global[i] >> 1 : (~0ull >> 1);
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t mine = cdf[lane & 31u];
    const uint32_t tab_lo = (uint32_t)mine, tab_hi = (uint32_t)(mine >> 32);
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint64_t acc = 0, x = gid * 0x9E3779B97F4A7C15ull + 1;
    for (int it = 0; it < iters; ++it) {
        uint64_t w[8], u[8];
        if (MODE == 0 || MODE == 1 || MODE >= 6) {
            stream_block(key, 5, gid, (uint32_t)it, w);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) { x = x * 6364136223846793005ull + 1442695040888963407ull; w[i] = x; }
        }
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc ^= w[i];
            continue;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) u[i] = w[i] >> 1;
        uint32_t mag[8];
        if (MODE == 1 || MODE == 2) cdt_scan<8>(cdf, entries, u, mag);
        else if (MODE == 3 || MODE == 6) scan_borrow<8>(cdf, entries, u, mag);
        else if (MODE == 4 || MODE == 7) scan_bperm<8>(tab_lo, tab_hi, u, mag);
        else scan_hi32<8>(cdf, entries, u, mag);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += gaussian_value(mag[i], w[i], 17592169062401ull) << i;
    }
    out[gid] = acc;
}

// raw issue rates with the occupancy the sampler sees
template <int OP>
__global__ void __launch_bounds__(256) k_rate(uint64_t* out, uint64_t seed, int iters) {
    constexpr int ILP = 8;
    uint32_t a[ILP], b[ILP];
    uint64_t x[ILP];
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    for (int i = 0; i < ILP; ++i) { x[i] = (gid + i) * 0x9E3779B97F4A7C15ull + seed; a[i] = (uint32_t)x[i]; b[i] = (uint32_t)(x[i] >> 32); }
    const uint64_t c = seed * 0xBF58476D1CE4E5B9ull | 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            if (OP == 0) a[i] += b[i];                                          // v_add_u32
            else if (OP == 1) a[i] = (a[i] << 7) | (a[i] >> 25);                // v_alignbit_b32
            else if (OP == 2) a[i] ^= b[i];                                     // v_xor_b32
            else if (OP == 3) a[i] += (x[i] < c) ? 1u : 0u;                     // v_cmp_lt_u64 + v_addc
            else if (OP == 4) a[i] += (b[i] < (uint32_t)c) ? 1u : 0u;           // v_cmp_lt_u32 + v_addc
            else if (OP == 5) a[i] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(b[i] & 0xFCu), (int)a[i]) + b[i];   // ds_bpermute_b32, data-dependent lanes
            else if (OP == 7) a[i] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(b[i] & 0x04u), (int)a[i]) + b[i];   // every lane reads lane 0 or 1
            else if (OP == 8) a[i] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((threadIdx.x * 4u) ^ (b[i] & 0x7Cu)) & 0xFCu), (int)a[i]) + b[i];   // a permutation
            else if (OP == 10) a[i] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(b[i] & 0x7Cu), (int)a[i]) + b[i];   // random lanes among 0..31 only: distinct banks
            else if (OP == 9) a[i] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((b[i] & 0x0Cu) | ((threadIdx.x & 1u) << 6)), (int)a[i]) + b[i];   // 64 lanes -> 8 sources
            else if (OP == 6) { a[i] += b[i]; b[i] ^= a[i]; b[i] = (b[i] << 16) | (b[i] >> 16); }   // a ChaCha third: add, xor, rotate
        }
        if (OP == 3 || OP == 4) { for (int i = 0; i < ILP; ++i) { x[i] += a[i]; b[i] += a[i]; } }   // keep the compares loop-variant
    }
    uint64_t acc = 0;
    for (int i = 0; i < ILP; ++i) acc += a[i] + b[i] + x[i];
    out[gid] = acc;
}

template <class F>
static int timeit(const char* name, double units_per_lane, int blocks_per_cu, F&& launch) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int grid = 256 * blocks_per_cu;
    launch(grid);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        launch(grid);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        best = std::min(best, ms);
    }
    const double units = units_per_lane * grid * 256.0;
    // cycles per wave-unit per SIMD at 2.4 GHz: a SIMD runs blocks_per_cu waves; time * clock / (units per wave * waves per SIMD)
    printf("%-34s %d waves/SIMD  %8.3f ms  %9.1f G units/s  %7.2f cyc per wave-unit per SIMD @2.4GHz\n", name, blocks_per_cu, best, units / best * 1e-6,
           best * 1e-3 * 2.4e9 / (units_per_lane * blocks_per_cu));
    return 0;
}

int main() {
    uint32_t entries = 0;
    const std::vector<uint64_t> table = make_table(3.19, &entries);
    printf("table entries %zu, scanned %u\n", table.size(), entries);
    uint64_t *d_out, *d_key, *d_cdf;
    CK(hipMalloc(&d_out, 256ull * 8 * 256 * 8));
    const uint64_t key[4] = {1, 2, 3, 4};
    CK(hipMalloc(&d_key, 32)); CK(hipMemcpy(d_key, key, 32, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_cdf, table.size() * 8)); CK(hipMemcpy(d_cdf, table.data(), table.size() * 8, hipMemcpyHostToDevice));
    const int iters = 256;
    static const char* rate_names[] = {"v_add_u32", "v_alignbit_b32", "v_xor_b32", "v_cmp_lt_u64 + v_addc", "v_cmp_lt_u32 + v_addc", "ds_bpermute_b32 (random lanes)",
                                       "add+xor+rot (3 ops)", "ds_bpermute_b32 (2 source lanes)", "ds_bpermute_b32 (a permutation)", "ds_bpermute_b32 (8 source lanes)",
                                       "ds_bpermute_b32 (random lanes 0..31)"};
    static const char* mode_names[] = {"cipher only (8 samples/unit)", "cipher + library scan", "library scan only", "borrow-chain scan only", "bpermute scan only",
                                       "hi32 scan only (rate probe)", "cipher + borrow scan", "cipher + bpermute scan"};
    for (int bpc : {1, 2, 4, 8}) {
#define RATE(OP) if (timeit(rate_names[OP], 8.0 * 8192, bpc, [&](int g) { hipLaunchKernelGGL(k_rate<OP>, dim3(g), dim3(256), 0, 0, d_out, 12345ull, 8192); })) return 1;
        RATE(0) RATE(1) RATE(2) RATE(3) RATE(4) RATE(5) RATE(6) RATE(7) RATE(8) RATE(9) RATE(10)
#define MODE(M) if (timeit(mode_names[M], 8.0 * iters, bpc, [&](int g) { hipLaunchKernelGGL(k_sampler<M>, dim3(g), dim3(256), 0, 0, d_out, d_key, d_cdf, entries, iters); })) return 1;
        MODE(0) MODE(1) MODE(2) MODE(3) MODE(4) MODE(5) MODE(6) MODE(7)
        printf("\n");
    }
    // the alternative scans must agree with the library scan on the same stream words
    const size_t lanes = 256ull * 2 * 256;
    std::vector<uint64_t> ref(lanes), alt(lanes);
    hipLaunchKernelGGL(k_sampler<1>, dim3(512), dim3(256), 0, 0, d_out, d_key, d_cdf, entries, 64);
    CK(hipMemcpy(ref.data(), d_out, lanes * 8, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(k_sampler<6>, dim3(512), dim3(256), 0, 0, d_out, d_key, d_cdf, entries, 64);
    CK(hipMemcpy(alt.data(), d_out, lanes * 8, hipMemcpyDeviceToHost));
    printf("borrow-chain scan == library scan: %s\n", ref == alt ? "yes" : "NO");
    hipLaunchKernelGGL(k_sampler<7>, dim3(512), dim3(256), 0, 0, d_out, d_key, d_cdf, entries, 64);
    CK(hipMemcpy(alt.data(), d_out, lanes * 8, hipMemcpyDeviceToHost));
    printf("bpermute scan == library scan: %s\n", ref == alt ? "yes" : "NO");
    return 0;
}
