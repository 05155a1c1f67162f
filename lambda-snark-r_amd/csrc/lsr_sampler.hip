// Discrete-Gaussian / uniform sampling kernels and the sample_gaussian C-ABI
// (reference: cpp-core/src/utils.cpp:132-146, header cpp-core/include/lambda_snark/utils.h:27).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "lambda_snark/batch.h"
#include "lambda_snark/utils.h"
#include "lsr_keys.hpp"
#include "lsr_runtime.hpp"
#include "lsr_sampler.hpp"

namespace lsr {

constexpr int kSamplerThreads = 256;

// One lane = one ChaCha block = eight samples.  The CDT table sits in the lanes of the wavefront (lsr_sampler.hpp, cdt_search) or,
// when it has more than 64 entries, in LDS at 63-bit precision.  Lanes past the end of the job run the cipher and the search like
// everyone else (the lane table needs every lane active) and skip the store.
__device__ __forceinline__ void gaussian_body(const GaussianJob& job, const uint64_t* __restrict__ cdf_global, uint32_t entries) {
    extern __shared__ uint64_t cdf[];
    const bool in_lanes = lane_table_steps(entries) != 0u;
    if (!in_lanes) {
        for (uint32_t i = threadIdx.x; i < entries; i += kSamplerThreads) cdf[i] = cdf_global[i] >> 1;
        __syncthreads();
    }
    const LaneTable tab = in_lanes ? lane_table_load(cdf_global, entries) : LaneTable{0u, 0u};
    const uint64_t blocks_per_object = (job.samples + kSamplesPerBlock - 1) / kSamplesPerBlock;
    const uint64_t lanes = blocks_per_object * job.objects;
    if (lanes == 0) return;                                               // an empty job of a three-job launch: uniform exit
    const uint64_t gid_raw = (uint64_t)blockIdx.x * kSamplerThreads + threadIdx.x;
    const bool live = gid_raw < lanes;
    const uint64_t gid = live ? gid_raw : lanes - 1;
    // lane -> (object, block): ring degrees are powers of two, so the usual case is a shift; a software 64-bit division per
    // lane (three of them) was a tenth of this kernel
    uint64_t object, block;
    if (job.objects == 1) { object = 0; block = gid; }
    else if ((blocks_per_object & (blocks_per_object - 1)) == 0) { object = gid >> (63 - __clzll((long long)blocks_per_object)); block = gid & (blocks_per_object - 1); }
    else { object = gid / blocks_per_object; block = gid - object * blocks_per_object; }
    const uint32_t group = (uint32_t)object / job.components;            // objects < 2^32 (launch_gaussian)
    const uint64_t index = job.index_base + ((uint32_t)object - group * job.components);
    uint64_t w[8];
    stream_block(job.keys + 4 * (size_t)group, job.domain, index, (uint32_t)block, w);
    uint64_t* dst = job.out + object * job.samples + block * kSamplesPerBlock;
    const uint64_t left = job.samples - block * kSamplesPerBlock;
    uint64_t u[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) u[s] = w[s] >> 1;
    uint32_t magnitude[8];
    cdt_magnitudes(tab, cdf, entries, u, magnitude);
    if (!live) return;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if ((uint64_t)s >= left) break;
        dst[s] = gaussian_value(magnitude[s], w[s], job.q);
    }
}
__global__ void __launch_bounds__(kSamplerThreads) gaussian_kernel(GaussianJob job, const uint64_t* __restrict__ cdf_global, uint32_t entries) {
    gaussian_body(job, cdf_global, entries);
}
// three jobs in one launch (blockIdx.y picks the job; the grid is as wide as the widest): r, e1 and e2 of a commitment batch —
// a single legacy lwe_commit call is launch-bound, and these are three of its nine kernels
__global__ void __launch_bounds__(kSamplerThreads) gaussian3_kernel(GaussianJob a, GaussianJob b, GaussianJob c, const uint64_t* __restrict__ cdf_global,
                                                                     uint32_t entries) {
    gaussian_body(blockIdx.y == 0 ? a : (blockIdx.y == 1 ? b : c), cdf_global, entries);
}

// One lane = one ChaCha block = eight uniform residues.
__global__ void __launch_bounds__(kSamplerThreads) uniform_kernel(uint64_t* __restrict__ out, const uint64_t* __restrict__ keys, uint64_t index_base,
                                                                    uint32_t components, uint32_t domain, uint64_t samples, uint64_t objects, uint64_t q) {
    const uint64_t blocks_per_object = (samples + 7) >> 3;
    const uint64_t gid = (uint64_t)blockIdx.x * kSamplerThreads + threadIdx.x;
    if (gid >= blocks_per_object * objects) return;
    const uint64_t object = gid / blocks_per_object;
    const uint64_t block = gid - object * blocks_per_object;
    uint64_t w[8];
    stream_block(keys + 4 * (object / components), domain, index_base + object % components, (uint32_t)block, w);
    uint64_t* dst = out + object * samples + block * 8;
    const uint64_t left = samples - block * 8;
#pragma unroll
    for (int s = 0; s < 8; ++s)
        if ((uint64_t)s < left) dst[s] = __umul64hi(w[s], q);
}

// out[o][i] = splitmix64_{i}(seed_base + o) mod q (q = 0: the raw word) — the synthetic inputs SURVEY.md §8(d) prescribes for
// configs 2 and 3 (one draw per coefficient; state after i+1 steps = seed + (i+1) * golden gamma, so every lane is independent)
__global__ void __launch_bounds__(kSamplerThreads) splitmix_kernel(uint64_t* __restrict__ out, uint64_t objects, uint64_t len, uint64_t seed_base,
                                                                     uint64_t q) {
    const uint64_t stride = (uint64_t)gridDim.x * kSamplerThreads;
    for (uint64_t g = (uint64_t)blockIdx.x * kSamplerThreads + threadIdx.x; g < objects * len; g += stride) {
        const uint64_t o = g / len, i = g - o * len;
        uint64_t z = seed_base + o + (i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        out[g] = q ? z % q : z;
    }
}

void launch_gaussian(const GaussianJob& job, const uint64_t* d_cdf, uint32_t entries, hipStream_t stream) {
    const uint64_t lanes = ((job.samples + kSamplesPerBlock - 1) / kSamplesPerBlock) * job.objects;
    if (!lanes) return;
    if (job.objects > 0xFFFFFFFFull || job.components == 0) throw std::runtime_error("gaussian job: objects must fit 32 bits, components >= 1");
    const unsigned grid = static_cast<unsigned>((lanes + kSamplerThreads - 1) / kSamplerThreads);
    hipLaunchKernelGGL(gaussian_kernel, dim3(grid), dim3(kSamplerThreads), entries * sizeof(uint64_t), stream, job, d_cdf, entries);
    LSR_HIP(hipGetLastError());
}

void launch_gaussian3(const GaussianJob& a, const GaussianJob& b, const GaussianJob& c, const uint64_t* d_cdf, uint32_t entries, hipStream_t stream) {
    uint64_t widest = 0;
    for (const GaussianJob* j : {&a, &b, &c}) {
        if (j->objects > 0xFFFFFFFFull || j->components == 0) throw std::runtime_error("gaussian job: objects must fit 32 bits, components >= 1");
        widest = std::max<uint64_t>(widest, ((j->samples + kSamplesPerBlock - 1) / kSamplesPerBlock) * j->objects);
    }
    if (!widest) return;
    const unsigned grid = static_cast<unsigned>((widest + kSamplerThreads - 1) / kSamplerThreads);
    hipLaunchKernelGGL(gaussian3_kernel, dim3(grid, 3), dim3(kSamplerThreads), entries * sizeof(uint64_t), stream, a, b, c, d_cdf, entries);
    LSR_HIP(hipGetLastError());
}

void launch_uniform(uint64_t* out, const uint64_t* d_keys, uint64_t index_base, uint32_t components, uint32_t domain, uint64_t samples,
                    uint64_t objects, uint64_t q, hipStream_t stream) {
    const uint64_t lanes = ((samples + 7) >> 3) * objects;
    if (!lanes) return;
    const unsigned grid = static_cast<unsigned>((lanes + kSamplerThreads - 1) / kSamplerThreads);
    hipLaunchKernelGGL(uniform_kernel, dim3(grid), dim3(kSamplerThreads), 0, stream, out, d_keys, index_base, components, domain, samples, objects, q);
    LSR_HIP(hipGetLastError());
}

// host-buffer sampler used by both C-ABI entry points
static int sample_to_host(uint64_t* output, size_t len, double sigma, uint64_t seed, uint32_t domain, uint64_t index) {
    const std::vector<uint64_t> table = gaussian_cdf(sigma);
    if (table.empty() || table.size() > 8000) throw std::runtime_error("sigma out of the supported range (table must fit LDS)");
    if ((len + 7) / 8 > 0xFFFFFFFFull) throw std::runtime_error("len exceeds one stream (2^35 samples)");
    if (visible_device_count() <= 0) throw std::runtime_error("no HIP device visible — no CPU fallback");
    const int device = default_device();
    if (device < 0) throw std::runtime_error(last_error_cstr());
    DeviceGuard guard(device);
    DeviceBuffer<uint64_t> d_cdf, d_key, d_out(len);
    d_cdf.upload(table);
    d_key.upload(key_words(expand_seed64(seed)));
    GaussianJob job{d_out.ptr, d_key.ptr, index, 1, domain, len, 1, 0};
    launch_gaussian(job, d_cdf.ptr, gaussian_scan_entries(table), nullptr);
    LSR_HIP(hipMemcpy(output, d_out.ptr, len * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 0;
}

}  // namespace lsr

extern "C" {

int sample_gaussian(uint64_t* output, size_t len, double sigma) noexcept {
    if (!output || len == 0 || !(sigma > 0.0) || !std::isfinite(sigma)) return -1;   // utils.cpp:133
    try {
        // fresh entropy per call, as the reference's std::random_device (utils.cpp:138)
        // (the raw-seed stream carries 64 bits of entropy per call plus a random stream index; the reference's sampler is a
        // test utility too — commitment.cpp never calls it)
        return lsr::sample_to_host(output, len, sigma, lsr::os_entropy64(), lsr::kDomUser, lsr::os_entropy64());
    } catch (const std::exception& e) {
        lsr::set_last_error(std::string("sample_gaussian: ") + e.what());
        std::fprintf(stderr, "lambda_snark_core: sample_gaussian failed: %s\n", e.what());
        return -1;
    } catch (...) {
        return -1;
    }
}

int lsr_fill_splitmix_device(uint64_t* d_out, size_t objects, size_t len, uint64_t seed_base, uint64_t q, void* stream) noexcept {
    if (!d_out) return -1;
    if (objects == 0 || len == 0) return 0;
    try {
        const uint64_t blocks = ((uint64_t)objects * len + lsr::kSamplerThreads - 1) / lsr::kSamplerThreads;
        const unsigned grid = static_cast<unsigned>(std::min<uint64_t>(blocks, 256ull * 64));
        hipLaunchKernelGGL(lsr::splitmix_kernel, dim3(grid), dim3(lsr::kSamplerThreads), 0, static_cast<hipStream_t>(stream), d_out, (uint64_t)objects,
                           (uint64_t)len, seed_base, q);
        LSR_HIP(hipGetLastError());
        return 0;
    } catch (const std::exception& e) {
        lsr::set_last_error(std::string("lsr_fill_splitmix_device: ") + e.what());
        return -1;
    }
}

int lsr_sample_gaussian_seeded(uint64_t* output, size_t len, double sigma, uint64_t seed, uint32_t domain, uint64_t index) noexcept {
    if (!output || len == 0 || !(sigma > 0.0) || !std::isfinite(sigma)) return -1;
    try {
        return lsr::sample_to_host(output, len, sigma, seed, domain, index);
    } catch (const std::exception& e) {
        lsr::set_last_error(std::string("lsr_sample_gaussian_seeded: ") + e.what());
        std::fprintf(stderr, "lambda_snark_core: lsr_sample_gaussian_seeded failed: %s\n", e.what());
        return -1;
    } catch (...) {
        return -1;
    }
}

}  // extern "C"
