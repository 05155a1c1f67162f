// Microbenchmark (round 2): do an FP64-bound kernel of limited residency and a memory-bound kernel, launched on two streams,
// really run side by side on the CUs?  (The co-residency question of the commitment pipeline: DESIGN.md §5.)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_concurrency.hip -o tools/bin/ubench_concurrency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(512) alu_kernel(double* out, int iters) {
    extern __shared__ double pad[];
    double a = threadIdx.x * 1e-3, b = 1.000001, c = 0.5, d = 0.25;
    for (int i = 0; i < iters; ++i) {
        a = __builtin_fma(a, b, c); d = __builtin_fma(d, b, a); c = __builtin_fma(c, b, d); a = __builtin_fma(a, d, c);
        d = __builtin_fma(d, b, a); c = __builtin_fma(c, b, d); a = __builtin_fma(a, b, c); d = __builtin_fma(d, a, c);
    }
    if (a + c + d == 12345.0) out[blockIdx.x] = a + pad[0];
}
__global__ void __launch_bounds__(256) mem_kernel(uint64_t* __restrict__ d, size_t total) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= total / 16) return;
    const size_t idx0 = ((g >> 12) << 16) | (g & 4095);
    uint64_t v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = d[idx0 + ((size_t)k << 12)];
#pragma unroll
    for (int k = 0; k < 16; ++k) d[idx0 + ((size_t)k << 12)] = v[k] * 3 + v[(k + 1) & 15];
}

int main() {
    const size_t total = (size_t)1 << 28;      // 2 GiB
    uint64_t* data; double* out;
    CK(hipMalloc(&data, total * 8)); CK(hipMalloc(&out, 1 << 20)); CK(hipMemset(data, 1, total * 8));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    CK(hipFuncSetAttribute((const void*)alu_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 120000));
    auto wall = [&](auto&& f) { hipDeviceSynchronize(); auto t0 = std::chrono::steady_clock::now(); f(); hipDeviceSynchronize();
                                return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    for (unsigned lds : {0u, 70000u, 100000u}) {
        for (int blocks : {256, 1024}) {
            const int iters = blocks == 256 ? 40000 : 10000;
            auto A = [&](hipStream_t s) { hipLaunchKernelGGL(alu_kernel, dim3(blocks), dim3(512), lds, s, out, iters); };
            auto M = [&](hipStream_t s) { for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(mem_kernel, dim3((unsigned)(total / 16 / 256)), dim3(256), 0, s, data, total); };
            A(s1); M(s2); hipDeviceSynchronize();
            const double ta = wall([&] { A(s1); }), tm = wall([&] { M(s2); });
            const double both = wall([&] { A(s1); M(s2); }), both_rev = wall([&] { M(s2); A(s1); });
            printf("alu: %4d blocks x 512 lanes, %6u B LDS (%s)  alone %.3f ms | mem (4 x 2 GiB r+w) alone %.3f ms | together %.3f ms (mem first: %.3f) | sum %.3f max %.3f\n",
                   blocks, lds, lds >= 82000 ? "1 per CU" : (lds ? "2 per CU" : "4 per CU"), ta, tm, both, both_rev, ta + tm, ta > tm ? ta : tm);
        }
    }
    return 0;
}
