// Host runtime shared by the C-ABI translation units: error reporting, device guards, contexts.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "lsr_arith.hpp"
#include "lsr_host_math.hpp"

namespace lsr {

void set_last_error(const std::string& msg);
const char* last_error_cstr();

struct HipFailure : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define LSR_HIP(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t lsr_e_ = (expr);                                                                           \
        if (lsr_e_ != hipSuccess)                                                                             \
            throw ::lsr::HipFailure(std::string(#expr) + ": " + hipGetErrorString(lsr_e_));                  \
    } while (0)

// Device selection that does not leak into the caller's thread state (Rust wrappers are Send, not
// Sync: the calling thread may change between calls — SURVEY.md §8(b) "Threading").
class DeviceGuard {
public:
    explicit DeviceGuard(int device);
    ~DeviceGuard();
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;

private:
    int previous_ = -1;
    bool switched_ = false;
};

int default_device();          // LAMBDA_SNARK_DEVICE, else LOCAL_RANK, else 0; -1 (+ message) when that index is not a visible device
int visible_device_count();    // 0 if the runtime cannot see a GPU

template <class T>
struct DeviceBuffer {
    T* ptr = nullptr;
    size_t count = 0;
    DeviceBuffer() = default;
    explicit DeviceBuffer(size_t n) { allocate(n); }
    ~DeviceBuffer() { release(); }
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    void allocate(size_t n) {
        release();
        if (n) LSR_HIP(hipMalloc(reinterpret_cast<void**>(&ptr), n * sizeof(T)));
        count = n;
    }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        count = 0;
    }
    void upload(const std::vector<T>& host) {
        allocate(host.size());
        if (!host.empty()) LSR_HIP(hipMemcpy(ptr, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    }
};

}  // namespace lsr

// The opaque C-ABI handle (reference: struct NttContext, cpp-core/src/ntt.cpp:21-26).
struct NttContext {
    uint64_t modulus = 0;
    uint32_t degree = 0;
    int logn = 0;
    int device = 0;
    bool use_f64 = false;
    bool gold = false;       // modulus = NTT_MODULUS (2^64 - 2^32 + 1): ArithGold kernels
    bool cyclic = false;     // cyclic twiddle tables (prover path) instead of negacyclic; psi then holds omega
    uint64_t psi = 0;
    lsr::ModParams mod{};
    // stage-order twiddles on the device; only the flavour in use is populated
    lsr::DeviceBuffer<double> fwd_f64, inv_f64;
    lsr::DeviceBuffer<lsr::ShoupOperand> fwd_u64, inv_u64;
    lsr::DeviceBuffer<uint64_t> fwd_gold, inv_gold;
    uint64_t n_inv_gold = 0, w_last_scaled_gold = 0;
    double n_inv_f64 = 0, w_last_scaled_f64 = 0;
    lsr::ShoupOperand n_inv_u64{}, w_last_scaled_u64{};
    // staging for the single-polynomial host-pointer entry points
    mutable std::mutex staging_mutex;
    mutable lsr::DeviceBuffer<uint64_t> staging;   // 3 n words
    // the context's own stream for the host-pointer entry points — created on first use (lsr::work_stream): device-API callers
    // bring their stream, and every stream a process opens competes for the runtime's few hardware queues
    mutable std::mutex stream_mutex;
    mutable hipStream_t stream = nullptr;
};

namespace lsr {

NttContext* create_ntt_context(uint64_t q, uint32_t n, int device);
// cyclic transform over F_q with the given primitive n-th root (0 = the reference's root for NTT_MODULUS)
NttContext* create_cyclic_ntt_context(uint64_t q, uint32_t n, uint64_t omega, int device);
void destroy_ntt_context(NttContext* ctx);
hipStream_t work_stream(const NttContext& ctx);
// asynchronous launches on `stream`, data resident on ctx->device
// add_on_inverse (optional): canonical residues [batch][n] added to the outputs of an inverse transform in its final
// store (the commitment's fused blinding add)
// pre_mul_on_inverse (optional, NTT_MODULUS contexts): residues [n] in Montgomery form (prover_montgomery); input word i of
// every polynomial of an inverse transform is multiplied by entry i as it is read
// forward_source (optional): a forward transform reads its operands from there ([batch][n], canonical) and writes d_data
void launch_ntt(const NttContext& ctx, uint64_t* d_data, size_t batch, bool inverse, hipStream_t stream,
                const uint64_t* add_on_inverse = nullptr, const uint64_t* pre_mul_on_inverse = nullptr,
                const uint64_t* forward_source = nullptr);
// only the strided top-bits round of an n > 4096 FP64-flavour transform: forward reads canonical `src`, writes raw elements
// to `dst` (may alias); inverse turns raw elements into canonical residues (+ optional canonical `add`), in place
// forward transform (Goldilocks, n <= 4096) with elementwise work fused into the read-in: mode 1 = multiply by x1 first (src may be
// NULL = in place), mode 2 = test x1 * x2 == operand and mark failing instances in bad[index >> log n]
bool ntt_forward_can_fuse(const NttContext& ctx);
void launch_ntt_forward_fused(const NttContext& ctx, uint64_t* d_data, size_t batch, hipStream_t stream, const uint64_t* src, int mode, const uint64_t* x1,
                              const uint64_t* x2, uint32_t* bad);
// true while `stream` records into a HIP graph: events recorded there belong to the capture (they cannot be waited for on the
// host), so the per-object ordering brackets stand aside — a captured call sequence is ordered by the capture itself, and the
// caller orders graph launches against other work on the same object
bool stream_is_capturing(hipStream_t stream);
// `words` 64-bit words at `dst` set to zero by a KERNEL on `stream` (device-API paths that may be captured into a HIP graph: a
// captured hipMemsetAsync cleared the verdict state on the first replay only — tools/graph_probe.py, profiles/README.md)
void zero_words_async(uint64_t* dst, size_t words, hipStream_t stream);
void launch_ntt_forward_finish(const NttContext& ctx, const uint64_t* d_data, size_t batch, hipStream_t stream, const uint64_t* x1,
                               const uint64_t* chat, const uint64_t* untwist, uint64_t half_m_inv, uint64_t* quotient, uint32_t* top);
void launch_top_round_forward(const NttContext& ctx, uint64_t* d_dst, const uint64_t* d_src, size_t polys, hipStream_t stream);
void launch_top_round_inverse(const NttContext& ctx, uint64_t* d_data, size_t polys, hipStream_t stream, const uint64_t* add);
// the same round with the added residues sampled in the pass (CDT Gaussian per polynomial, lsr_sampler.hpp) instead of read
struct BlindSampler;
void launch_top_round_inverse_sampled(const NttContext& ctx, uint64_t* d_data, size_t polys, hipStream_t stream, const BlindSampler& sampler);
// split sampling (sampler.side != NULL in both calls): the forward round of a chunk also samples the first half of the rows of the
// residues its inverse round will add, into sampler.side ([polys][n / 2^r][2^r / 16] words, int8 per sample)
void launch_top_round_forward_sampling(const NttContext& ctx, uint64_t* d_dst, const uint64_t* d_src, size_t polys, hipStream_t stream,
                                       const BlindSampler& sampler);
void launch_pointwise(const NttContext& ctx, uint64_t* d_out, const uint64_t* d_a, const uint64_t* d_b, size_t count,
                      hipStream_t stream);
// out[b][i] = in[b][bitrev_logn(i)] (out != in)
void launch_bit_reverse(uint64_t* d_out, const uint64_t* d_in, int logn, size_t batch, hipStream_t stream);
int arith_mode();

}  // namespace lsr
