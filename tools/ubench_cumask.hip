// Microbenchmark (round 2): partition the chip between two streams with CU masks (hipExtStreamCreateWithCUMask) so that a
// memory-bound kernel (the strided rounds of the commitment) and an FP64-bound kernel (the fused middle stage) run side by side
// on DISJOINT compute units.  Questions: (1) which CUs does mask bit i select (XCC / CU id histogram); (2) how many CUs per XCD
// does a streaming kernel need to keep the fabric busy; (3) do the two kernels then run at their stand-alone speed together?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_cumask.hip -o tools/bin/ubench_cumask
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <map>
#include <set>
#include <vector>
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
// the access pattern of ntt_strided_round<4> at n = 2^16: 16 residues per lane, 32 KiB apart, read-modify-write
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

// ---- per-CU throughput variants (section 4): what bounds a streaming kernel on a FEW compute units? ----
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
template <int MODE> __global__ void __launch_bounds__(256) var_kernel(uint64_t* __restrict__ d, size_t total) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (MODE == 3 || MODE == 6) {                         // 16 B per lane: two adjacent residues, 16 rows -> half the lanes
        if (g >= total / 32) return;
        const size_t idx0 = ((g >> 11) << 16) | ((g & 2047) << 1);
        u64x2 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = MODE == 6 ? __builtin_nontemporal_load((const u64x2*)(d + idx0 + ((size_t)k << 12))) : *(const u64x2*)(d + idx0 + ((size_t)k << 12));
#pragma unroll
        for (int k = 0; k < 16; ++k) { u64x2 o; o.x = v[k].x * 3 + v[(k + 1) & 15].y; o.y = v[k].y * 3 + v[(k + 1) & 15].x;
            if (MODE == 6) __builtin_nontemporal_store(o, (u64x2*)(d + idx0 + ((size_t)k << 12))); else *(u64x2*)(d + idx0 + ((size_t)k << 12)) = o; }
        return;
    }
    if (g >= total / 16) return;
    const size_t idx0 = MODE == 4 ? (g >> 8 << 12) | (g & 255) : ((g >> 12) << 16) | (g & 4095);
    const size_t step = MODE == 4 ? 256 : 4096;
    uint64_t v[16];
    if (MODE != 2) {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = MODE == 5 ? __builtin_nontemporal_load(d + idx0 + k * step) : d[idx0 + k * step];
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = g + k;
    }
    if (MODE == 1) {
        uint64_t acc = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += v[k];
        if (acc == 0x1234567) d[idx0] = acc;
        return;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) { const uint64_t o = v[k] * 3 + v[(k + 1) & 15]; if (MODE == 5) __builtin_nontemporal_store(o, d + idx0 + k * step); else d[idx0 + k * step] = o; }
}

// ---- section 5: what bounds the READ throughput of one CU?  loads per lane, bytes per load, waves per CU (LDS padding) ----
template <int LOADS, int WIDE> __global__ void __launch_bounds__(256) read_kernel(const uint64_t* __restrict__ d, uint64_t* __restrict__ sink, size_t total) {
    extern __shared__ double pad[];
    const size_t lanes_needed = total / (LOADS * (WIDE ? 2 : 1));
    uint64_t acc = 0;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < lanes_needed; g += (size_t)gridDim.x * 256) {
        if (WIDE) {
            const size_t idx0 = (g / 2048) * (2048 * 2 * LOADS) + (g % 2048) * 2;
            u64x2 v[LOADS];
#pragma unroll
            for (int k = 0; k < LOADS; ++k) v[k] = __builtin_nontemporal_load((const u64x2*)(d + idx0 + (size_t)k * 4096));
#pragma unroll
            for (int k = 0; k < LOADS; ++k) acc += v[k].x ^ v[k].y;
        } else {
            const size_t idx0 = (g / 4096) * (4096 * LOADS) + (g % 4096);
            uint64_t v[LOADS];
#pragma unroll
            for (int k = 0; k < LOADS; ++k) v[k] = __builtin_nontemporal_load(d + idx0 + (size_t)k * 4096);
#pragma unroll
            for (int k = 0; k < LOADS; ++k) acc += v[k];
        }
    }
    if (acc == 0x1234567) sink[0] = acc + (uint64_t)pad[0];
}
__global__ void where_kernel(unsigned* out) {
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw;
    }
    // stay a little so that the blocks spread over every allowed CU
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 20000) {}
}

static std::vector<uint32_t> mask_interleaved(int cus_per_xcd, int first, int count) {        // bit = cu * 8 + xcd
    std::vector<uint32_t> m(8, 0);
    for (int cu = first; cu < first + count && cu < cus_per_xcd; ++cu)
        for (int x = 0; x < 8; ++x) { const int bit = cu * 8 + x; m[bit / 32] |= 1u << (bit % 32); }
    return m;
}
static std::vector<uint32_t> mask_linear(int first, int count) {                               // bit = plain CU number
    std::vector<uint32_t> m(8, 0);
    for (int bit = first; bit < first + count && bit < 256; ++bit) m[bit / 32] |= 1u << (bit % 32);
    return m;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("# %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    const size_t total = (size_t)1 << 28;      // 2 GiB of u64
    uint64_t* data; double* out; unsigned* where;
    CK(hipMalloc(&data, total * 8)); CK(hipMalloc(&out, 1 << 20)); CK(hipMemset(data, 1, total * 8));
    CK(hipMalloc(&where, 8192 * 8));
    CK(hipFuncSetAttribute((const void*)alu_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 120000));
    auto wall = [&](auto&& f) { hipDeviceSynchronize(); auto t0 = std::chrono::steady_clock::now(); f(); hipDeviceSynchronize();
                                return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    auto histogram = [&](hipStream_t s, const char* label) {
        hipMemsetAsync(where, 0xff, 8192 * 8, s);
        hipLaunchKernelGGL(where_kernel, dim3(4096), dim3(64), 0, s, where);
        hipStreamSynchronize(s);
        std::vector<unsigned> h(8192); hipMemcpy(h.data(), where, 8192 * 4, hipMemcpyDeviceToHost);
        std::map<unsigned, std::set<unsigned>> per_xcc;
        for (int b = 0; b < 4096; ++b) per_xcc[h[2 * b] & 0xf].insert((h[2 * b + 1] >> 8) & 0xff);   // cu_id[11:8], sh_id[12], se_id[15:13]
        printf("%s:", label);
        size_t cus = 0;
        for (auto& kv : per_xcc) { printf(" xcc%u=%zu", kv.first, kv.second.size()); cus += kv.second.size(); }
        printf("  (distinct CUs %zu)\n", cus);
    };

    // (1) which CUs does a mask select?
    {
        hipStream_t s; auto m = mask_linear(0, 32); CK(hipExtStreamCreateWithCUMask(&s, 8, m.data())); histogram(s, "mask bits 0..31            "); hipStreamDestroy(s);
        m = mask_interleaved(32, 0, 4); CK(hipExtStreamCreateWithCUMask(&s, 8, m.data())); histogram(s, "mask bits {cu*8+x, cu<4}    "); hipStreamDestroy(s);
        m = mask_linear(0, 8); CK(hipExtStreamCreateWithCUMask(&s, 8, m.data())); histogram(s, "mask bits 0..7             "); hipStreamDestroy(s);
        m = mask_linear(0, 256); CK(hipExtStreamCreateWithCUMask(&s, 8, m.data())); histogram(s, "mask bits 0..255 (all)     "); hipStreamDestroy(s);
    }
    // (2) streaming kernel on c CUs per XCD; FP64 kernel on the rest; (3) both together
    for (int c : {32, 16, 12, 8, 6, 4}) {
        hipStream_t sm, sa;
        auto mm = mask_interleaved(32, 0, c);
        auto ma = mask_interleaved(32, c == 32 ? 0 : c, c == 32 ? 32 : 32 - c);
        CK(hipExtStreamCreateWithCUMask(&sm, 8, mm.data())); CK(hipExtStreamCreateWithCUMask(&sa, 8, ma.data()));
        auto M = [&](hipStream_t s) { for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(mem_kernel, dim3((unsigned)(total / 16 / 256)), dim3(256), 0, s, data, total); };
        auto A = [&](hipStream_t s) { hipLaunchKernelGGL(alu_kernel, dim3(2048), dim3(512), 70000, s, out, 5000); };
        M(sm); A(sa); hipDeviceSynchronize();
        const double tm = wall([&] { M(sm); }), ta = wall([&] { A(sa); });
        const double both = wall([&] { A(sa); M(sm); });
        printf("mem on %2d CUs/XCD: 4 x 2 GiB r+w %.3f ms (%.0f GB/s) | fp64 on %2d CUs/XCD %.3f ms | together %.3f ms | max %.3f sum %.3f\n",
               c, tm, 4.0 * 2 * total * 8 / tm / 1e6, c == 32 ? 32 : 32 - c, ta, both, ta > tm ? ta : tm, ta + tm);
        hipStreamDestroy(sm); hipStreamDestroy(sa);
    }
    // (4) per-CU throughput of streaming variants on 8 and 16 CUs per XCD
    for (int c : {8, 16, 32}) {
        hipStream_t sm; auto mm = mask_interleaved(32, 0, c); CK(hipExtStreamCreateWithCUMask(&sm, 8, mm.data()));
        auto run = [&](auto kern, unsigned blocks, const char* label, double bytes) {
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, sm, data, total); hipDeviceSynchronize();
            const double t = wall([&] { for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, sm, data, total); });
            printf("  %2d CUs/XCD  %-34s %.3f ms  %.0f GB/s  (%.1f GB/s per CU)\n", c, label, t, 4 * bytes / t / 1e6, 4 * bytes / t / 1e6 / (c * 8));
        };
        const unsigned nb = (unsigned)(total / 16 / 256);
        run(var_kernel<0>, nb, "strided 8 B r+w", 2.0 * total * 8);
        run(var_kernel<1>, nb, "strided 8 B read only", 1.0 * total * 8);
        run(var_kernel<2>, nb, "strided 8 B write only", 1.0 * total * 8);
        run(var_kernel<3>, nb / 2, "strided 16 B r+w", 2.0 * total * 8);
        run(var_kernel<4>, nb, "contiguous rows 8 B r+w", 2.0 * total * 8);
        run(var_kernel<5>, nb, "strided 8 B r+w nontemporal", 2.0 * total * 8);
        run(var_kernel<6>, nb / 2, "strided 16 B r+w nontemporal", 2.0 * total * 8);
        hipStreamDestroy(sm);
    }
    // (5) read throughput of 8 CUs per XCD: loads in flight per lane, load width, residency
    {
        const int c = 8;
        hipStream_t sm; auto mm = mask_interleaved(32, 0, c); CK(hipExtStreamCreateWithCUMask(&sm, 8, mm.data()));
        auto run = [&](auto kern, unsigned lds, unsigned blocks, const char* label) {
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, sm, data, (uint64_t*)out, total); hipDeviceSynchronize();
            const double t = wall([&] { for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, sm, data, (uint64_t*)out, total); });
            printf("  read only, %2d CUs/XCD, %-46s %.3f ms  %.0f GB/s  (%.1f GB/s per CU)\n", c, label, t, 4.0 * total * 8 / t / 1e6, 4.0 * total * 8 / t / 1e6 / (c * 8));
        };
        const unsigned many = 1u << 16;
        run(read_kernel<4, 0>, 0, many, "4 x 8 B per lane, full residency");
        run(read_kernel<8, 0>, 0, many, "8 x 8 B per lane, full residency");
        run(read_kernel<16, 0>, 0, many, "16 x 8 B per lane, full residency");
        run(read_kernel<32, 0>, 0, many, "32 x 8 B per lane, full residency");
        run(read_kernel<8, 1>, 0, many, "8 x 16 B per lane, full residency");
        run(read_kernel<16, 1>, 0, many, "16 x 16 B per lane, full residency");
        run(read_kernel<16, 0>, 40000, many, "16 x 8 B per lane, 4 workgroups per CU (LDS pad)");
        run(read_kernel<16, 0>, 80000, many, "16 x 8 B per lane, 2 workgroups per CU (LDS pad)");
        run(read_kernel<16, 0>, 0, 64 * 8, "16 x 8 B per lane, persistent grid 8 WG per CU");
        run(read_kernel<16, 1>, 0, 64 * 8, "16 x 16 B per lane, persistent grid 8 WG per CU");
        hipStreamDestroy(sm);
    }
    return 0;
}
