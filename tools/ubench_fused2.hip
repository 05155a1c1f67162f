// Microbenchmark v2 (round 2): ubench_fused with streaming (nt) policy on the accesses that cross the fabric (first-pass loads,
// last-pass stores) so that only the hand-over lines stay in the XCD's L2, and a finer poll.  Original header:
// Microbenchmark: the two-pass walk of an n = 2^16 transform fused into ONE persistent kernel — teams of T workgroups on
// one XCD own a polynomial, run the strided pass, meet at a team barrier, then run the contiguous pass from the XCD's L2.
// Data movement + token arithmetic only.  hipcc --offload-arch=gfx950 -O3 tools/ubench_fused.hip -o tools/bin/ubench_fused
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int R = 16;
constexpr unsigned kSpinLimit = 1u << 20;

__device__ __forceinline__ uint64_t load_sc1(const uint64_t* p) {
    uint64_t v;
    asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int T, int SLEEP, bool SC1, bool NT, bool L1INV = false>
__global__ void __launch_bounds__(256) fused(uint64_t* __restrict__ data, size_t polys, unsigned* __restrict__ counters, unsigned* __restrict__ masks,
                                             unsigned* __restrict__ err) {
    const int t = threadIdx.x;
    const int b = blockIdx.x;
    const int idx = b >> 3;                              // idx-th workgroup of its XCD under round-robin placement
    const int team = (b & 7) + 8 * (idx / T), member = idx % T;
    const int nteams = gridDim.x / T;
    if (t == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;   // HW_REG_XCC_ID
        atomicOr(&masks[team], 1u << xcc);
    }
    unsigned epoch = 0;
    for (size_t poly = team; poly < polys; poly += nteams) {
        uint64_t* p = data + poly * 65536;
        uint64_t v[R];
        const size_t e = (size_t)member * 256 + t;
#pragma unroll
        for (int k = 0; k < R; ++k) v[k] = NT ? __builtin_nontemporal_load(p + e + ((size_t)k << 12)) : p[e + ((size_t)k << 12)];
#pragma unroll
        for (int k = 0; k < R; ++k) v[k] = (v[k] * 3 + k) & 0xFFFFFFFFFFFFull;
#pragma unroll
        for (int k = 0; k < R; ++k) p[e + ((size_t)k << 12)] = v[k];
        // team barrier: stores drained, one arrival per workgroup, bounded poll, acquire (L1 invalidate) on the reading CU
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        epoch += T;
        if (t == 0) {
            __hip_atomic_fetch_add(&counters[team * 64], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (__hip_atomic_load(&counters[team * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
                __builtin_amdgcn_s_sleep(SLEEP);
                if (++spins > kSpinLimit) { atomicExch(err, 1u); break; }
            }
            if (!SC1 && !L1INV) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __syncthreads();
        if (L1INV) asm volatile("buffer_inv sc0\n\ts_waitcnt vmcnt(0)" ::: "memory");    // this CU's L1 only (workgroup-scope invalidate): the team shares one L2
        const size_t base = (size_t)member * 4096 + t;
        if (SC1) {
#pragma unroll
            for (int k = 0; k < R; ++k) v[k] = load_sc1(p + base + (size_t)k * 256);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
#pragma unroll
            for (int k = 0; k < R; ++k) v[k] = p[base + (size_t)k * 256];
        }
#pragma unroll
        for (int k = 0; k < R; ++k) { if (NT) __builtin_nontemporal_store(v[k] * 5 + 1, p + base + (size_t)k * 256); else p[base + (size_t)k * 256] = v[k] * 5 + 1; }
    }
}

// the two-launch form for comparison (ubench_pack's variant 0)
__global__ void __launch_bounds__(256) pass1(uint64_t* __restrict__ d, size_t total) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= total / R) return;
    const size_t idx0 = ((g >> 12) << 16) | (g & 4095);
    uint64_t v[R];
#pragma unroll
    for (int k = 0; k < R; ++k) v[k] = d[idx0 + ((size_t)k << 12)];
#pragma unroll
    for (int k = 0; k < R; ++k) d[idx0 + ((size_t)k << 12)] = (v[k] * 3 + k) & 0xFFFFFFFFFFFFull;
}
__global__ void __launch_bounds__(256) pass2(uint64_t* __restrict__ d) {
    const size_t base = (size_t)blockIdx.x * 4096 + threadIdx.x;
    uint64_t v[R];
#pragma unroll
    for (int k = 0; k < R; ++k) v[k] = d[base + (size_t)k * 256];
#pragma unroll
    for (int k = 0; k < R; ++k) d[base + (size_t)k * 256] = v[k] * 5 + 1;
}

int main(int argc, char** argv) {
    const size_t polys = 4096, n = 65536, total = polys * n;
    uint64_t* data; unsigned *counters, *masks, *err;
    CK(hipMalloc(&data, total * 8)); CK(hipMalloc(&counters, 4096 * 64 * 4)); CK(hipMalloc(&masks, 4096 * 4)); CK(hipMalloc(&err, 4));
    CK(hipMemset(data, 1, total * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    // reference result of one two-launch walk, for a correctness check of the fused walk
    uint64_t* ref = (uint64_t*)malloc(1 << 20); uint64_t* got = (uint64_t*)malloc(1 << 20);
    {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemset(data, 1, total * 8));
            CK(hipEventRecord(a));
            for (size_t c = 0; c < polys; c += 512) {
                uint64_t* d = data + c * n;
                hipLaunchKernelGGL(pass1, dim3((unsigned)(512 * n / R / 256)), dim3(256), 0, 0, d, 512 * n);
                hipLaunchKernelGGL(pass2, dim3((unsigned)(512 * n / 4096)), dim3(256), 0, 0, d);
            }
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        CK(hipMemcpy(ref, data + 1234 * n, 1 << 20 >> 1, hipMemcpyDeviceToHost));
        printf("two launches per 512-poly chunk          %.3f ms per 4096 polys\n", best);
    }
    const int grids[] = {256, 512, 768, 1024, 1536, 2048};
    for (int variant = 1; variant < 5; ++variant)
    for (int gi = 0; gi < 6; ++gi) {
        const int grid = grids[gi];
        float best = 1e9; unsigned herr = 0; unsigned hmask[128];
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemset(data, 1, total * 8));
            CK(hipMemset(counters, 0, 4096 * 64 * 4)); CK(hipMemset(masks, 0, 4096 * 4)); CK(hipMemset(err, 0, 4));
            CK(hipEventRecord(a));
            if (variant == 0) hipLaunchKernelGGL((fused<16, 32, false, false>), dim3(grid), dim3(256), 0, 0, data, polys, counters, masks, err);
            else if (variant == 1) hipLaunchKernelGGL((fused<16, 32, false, true>), dim3(grid), dim3(256), 0, 0, data, polys, counters, masks, err);
            else if (variant == 2) hipLaunchKernelGGL((fused<16, 1, false, true>), dim3(grid), dim3(256), 0, 0, data, polys, counters, masks, err);
            else if (variant == 3) hipLaunchKernelGGL((fused<16, 32, false, true, true>), dim3(grid), dim3(256), 0, 0, data, polys, counters, masks, err);
            else hipLaunchKernelGGL((fused<16, 8, false, true, true>), dim3(grid), dim3(256), 0, 0, data, polys, counters, masks, err);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
            CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            if (herr) break;
        }
        CK(hipMemcpy(got, data + 1234 * n, 1 << 20 >> 1, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hmask, masks, sizeof hmask, hipMemcpyDeviceToHost));
        int split = 0;
        for (int i = 0; i < grid / 16; ++i) split += __builtin_popcount(hmask[i]) != 1;
        int bad = 0;
        for (size_t i = 0; i < n; ++i) bad += got[i] != ref[i];
        const char* label[] = {"plain sleep32", "nt sleep32", "nt sleep1", "nt L1inv s32", "nt L1inv s8"};
        printf("fused persistent %-14s %4d workgroups (T=16, %2d teams per XCD)  %.3f ms per 4096 polys   err=%u split_teams=%d mismatches=%d\n", label[variant], grid,
               grid / 16 / 8, best, herr, split, bad);
        if (herr) return 2;
    }
    return 0;
}
