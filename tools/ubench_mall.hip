// Microbenchmark: does the 256 MiB Infinity Cache absorb a write-then-read hand-off between two kernels?
// hipcc --offload-arch=gfx950 -O3 tools/ubench_mall.hip -o tools/bin/ubench_mall
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_read(const ulonglong2* __restrict__ in, uint64_t* sink, size_t n16) {
    uint64_t acc = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) { ulonglong2 v = in[i]; acc += v.x ^ v.y; }
    if (acc == 0x1234567) sink[0] = acc;
}
__global__ void __launch_bounds__(256) k_write(ulonglong2* __restrict__ out, size_t n16, uint64_t seed) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) out[i] = make_ulonglong2(i + seed, i ^ seed);
}
__global__ void __launch_bounds__(256) k_rmw(ulonglong2* __restrict__ buf, size_t n16, uint64_t seed) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) { ulonglong2 v = buf[i]; v.x += seed; v.y ^= seed; buf[i] = v; }
}

int main() {
    const size_t max_bytes = 4ull << 30;
    ulonglong2* buf; uint64_t* sink;
    CK(hipMalloc(&buf, max_bytes)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(buf, 1, max_bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int grid = 256 * 8;
    const size_t sizes_mib[] = {32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048};
    printf("%8s %14s %14s %14s %16s\n", "MiB", "read GB/s", "write GB/s", "rmw GB/s(r+w)", "w-then-r GB/s");
    for (size_t mib : sizes_mib) {
        const size_t bytes = mib << 20, n16 = bytes / 16;
        // number of disjoint regions to cycle through so that total traffic is comparable; all regions inside 4 GiB.
        const int reps = 20;
        float ms;
        // (1) repeated read of the same region
        hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, buf, sink, n16);
        CK(hipEventRecord(a));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, buf, sink, n16);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        const double rd = (double)bytes * reps / ms * 1e-6;
        // (2) repeated write
        CK(hipEventRecord(a));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, buf, n16, (uint64_t)r);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        const double wr = (double)bytes * reps / ms * 1e-6;
        // (3) in-place read-modify-write of the same region
        CK(hipEventRecord(a));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_rmw, dim3(grid), dim3(256), 0, 0, buf, n16, (uint64_t)r);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        const double rmw = 2.0 * bytes * reps / ms * 1e-6;
        // (4) walk through 4 GiB in regions of this size: rmw pass 1 then rmw pass 2 on the SAME region before moving on
        const size_t regions = max_bytes / bytes;
        CK(hipEventRecord(a));
        for (size_t g = 0; g < regions; ++g) {
            ulonglong2* p = buf + g * n16;
            hipLaunchKernelGGL(k_rmw, dim3(grid), dim3(256), 0, 0, p, n16, 1ull);
            hipLaunchKernelGGL(k_rmw, dim3(grid), dim3(256), 0, 0, p, n16, 2ull);
        }
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        const double two_pass = 4.0 * max_bytes / ms * 1e-6;   // bytes moved by both passes (r+w each)
        printf("%8zu %14.0f %14.0f %14.0f %16.0f   (two-pass walk over 4 GiB: %.3f ms)\n", mib, rd, wr, rmw, two_pass, ms);
    }
    return 0;
}
