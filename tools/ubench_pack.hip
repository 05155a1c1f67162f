// Microbenchmark: is a two-pass walk faster when the intermediate is stored as 4 B + 2 B planes (6 B/elem) instead of 8 B?
// Pass 1 reads u64 (8 B/lane, rows 2 KiB apart like the strided NTT round) and writes the intermediate; pass 2 reads it
// contiguously and writes u64.  hipcc --offload-arch=gfx950 -O3 tools/ubench_pack.hip -o tools/bin/ubench_pack
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int R = 16;
// pass 1: thread g handles elements idx0 + k*4096 (k<16) of a 65536-element polynomial
template <bool PACKED>
__global__ void __launch_bounds__(256) pass1(const uint64_t* __restrict__ in, uint64_t* __restrict__ mid64, uint32_t* __restrict__ lo, uint16_t* __restrict__ hi, size_t total) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= total / R) return;
    const size_t idx0 = ((g >> 12) << 16) | (g & 4095);
    uint64_t v[R];
#pragma unroll
    for (int k = 0; k < R; ++k) v[k] = in[idx0 + ((size_t)k << 12)];
#pragma unroll
    for (int k = 0; k < R; ++k) v[k] = (v[k] * 3 + k) & 0xFFFFFFFFFFFFull;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const size_t i = idx0 + ((size_t)k << 12);
        if (PACKED) { lo[i] = (uint32_t)v[k]; hi[i] = (uint16_t)(v[k] >> 32); }
        else mid64[i] = v[k];
    }
}
// pass 2: contiguous 4096-element tiles, lane t handles t + 256 k
template <bool PACKED>
__global__ void __launch_bounds__(256) pass2(uint64_t* __restrict__ out, const uint64_t* __restrict__ mid64, const uint32_t* __restrict__ lo, const uint16_t* __restrict__ hi, size_t total) {
    const size_t base = (size_t)blockIdx.x * 4096 + threadIdx.x;
    uint64_t v[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const size_t i = base + (size_t)k * 256;
        if (PACKED) v[k] = (uint64_t)lo[i] | ((uint64_t)hi[i] << 32);
        else v[k] = mid64[i];
    }
#pragma unroll
    for (int k = 0; k < R; ++k) out[base + (size_t)k * 256] = v[k] * 5 + 1;
}

int main() {
    const size_t polys = 4096, n = 65536, total = polys * n, chunk_polys = 512, chunk = chunk_polys * n;
    uint64_t *data, *mid64; uint32_t* lo; uint16_t* hi;
    CK(hipMalloc(&data, total * 8)); CK(hipMalloc(&mid64, chunk * 8)); CK(hipMalloc(&lo, chunk * 4)); CK(hipMalloc(&hi, chunk * 2));
    CK(hipMemset(data, 1, total * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int variant = 0; variant < 3; ++variant) {
        float best = 1e9;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(a));
            for (size_t c = 0; c < polys; c += chunk_polys) {
                uint64_t* d = data + c * n;
                const unsigned g1 = (unsigned)(chunk / R / 256), g2 = (unsigned)(chunk / 4096);
                if (variant == 0) {        // in place, 8 B intermediate (what the library does today)
                    hipLaunchKernelGGL(pass1<false>, dim3(g1), dim3(256), 0, 0, d, d, lo, hi, chunk);
                    hipLaunchKernelGGL(pass2<false>, dim3(g2), dim3(256), 0, 0, d, d, lo, hi, chunk);
                } else if (variant == 1) { // out of place, 8 B intermediate in a scratch buffer
                    hipLaunchKernelGGL(pass1<false>, dim3(g1), dim3(256), 0, 0, d, mid64, lo, hi, chunk);
                    hipLaunchKernelGGL(pass2<false>, dim3(g2), dim3(256), 0, 0, d, mid64, lo, hi, chunk);
                } else {                   // 4 B + 2 B planes in scratch buffers
                    hipLaunchKernelGGL(pass1<true>, dim3(g1), dim3(256), 0, 0, d, mid64, lo, hi, chunk);
                    hipLaunchKernelGGL(pass2<true>, dim3(g2), dim3(256), 0, 0, d, mid64, lo, hi, chunk);
                }
            }
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        const char* names[] = {"in-place 8B intermediate", "scratch 8B intermediate", "scratch 4B+2B planes"};
        printf("%-28s %.3f ms per 4096 polys  (%.2f M transforms/s equivalent)\n", names[variant], best, polys / best / 1e3);
    }
    return 0;
}
