// Module-LWE commitment on the GPU and the lwe_* C-ABI (reference: cpp-core/src/commitment.cpp).
//
// Scheme (DESIGN.md "Commitment definition"; satisfies the documented contract of
// cpp-core/include/lambda_snark/commitment.h:43-52, which the SEAL-backed reference does not implement):
//   context : A_hat in R_q^{k x k} uniform (NTT domain), s, e <- chi^k, b_hat = A_hat s_hat + e_hat,
//             t = SEAL Batching(n,20) prime, Delta = floor(q/t)
//   commit  : r, e1 <- chi^k, e2 <- chi from the ChaCha20 streams of a per-commitment 256-bit key:
//             PRF(seed, context id, embedded message) for seed != 0, OS entropy for seed == 0 (lsr_keys.hpp)
//             u = INTT(A_hat^T r_hat) + e1                 (the matrix–vector + blinding-add workload)
//             v = INTT(<b_hat, r_hat>) + e2 + round(q (m mod t) / t)      (why not floor(q/t) m)
//   verify  : round(t/q (v - <s,u>)) mod t == m, word for word (commitment.cpp:223-226: decoded ^ message, no reduction of
//             the claimed message: a word >= t never opens)
//   combine : sum_i (c_i mod t) (u_i, v_i)
// Wire format: data[0] = payload bytes; payload = {"LSRC0001", n | k<<32, q, t, u[k][n], v[n]}.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <exception>
#include <memory>
#include <new>
#include <thread>
#include <type_traits>

#include "lambda_snark/batch.h"
#include "lambda_snark/commitment.h"
#include "lsr_arith.hpp"
#include "lsr_commit_fused.hpp"
#include "lsr_commit_kernels.hpp"
#include "lsr_commit_tile.hpp"
#include "lsr_commit_keys.hpp"
#include "lsr_keys.hpp"
#include "lsr_runtime.hpp"
#include "lsr_sampler.hpp"

// full commitments at n >= 2^16, rank 4: 1 = one pass of the middle stage with five accumulators (a few spilled registers),
// 0 = the A^T product and the b_hat product as two passes over the workspace (build-time A/B switch, csrc/Makefile EXTRA=)
#ifndef LSR_K4_SINGLE_PASS
#define LSR_K4_SINGLE_PASS 1
#endif

// ------------------------------------------------------------------------------------------------
// the opaque context (reference: struct LweContext, commitment.cpp:31-40)
// ------------------------------------------------------------------------------------------------
struct LweContext {
    PublicParams params{};
    uint64_t q = 0, t = 0, delta = 0;
    uint32_t n = 0, k = 0;
    int logn = 0;
    double sigma = 0;
    int device = 0;
    NttContext* ntt = nullptr;
    lsr::DeviceBuffer<uint64_t> a_hat, s_hat, b_hat, cdf;
    uint32_t cdf_entries = 0;
    lsr::ContextKeys keys{};          // key schedule of this context (lsr_keys.hpp); keys.sec is secret
    uint64_t key_seed = 0;            // what the context was created from (0 = OS entropy): lets a sharded run replicate it
    double noise_unit = 0;            // 8 sqrt(2 k n) sigma^2: tail bound on the decoding noise of one fresh commitment
    // scratch for commit / verify / combine, grown on demand; guarded by `mutex`
    mutable std::mutex mutex;
    mutable lsr::DeviceBuffer<uint64_t> ws_r, ws_e1, ws_e2, ws_u, ws_v, ws_dm, ws_keys;   // ws_keys: [batch][4]
    mutable lsr::DeviceBuffer<unsigned long long> ws_flag;
    mutable size_t ws_batch = 0, ws_in_batch = 0;
    mutable std::vector<uint64_t> ws_key_host;    // source of an asynchronous upload: must outlive the call that fills it
    // pinned host staging for the gather of a batch (two bulk D2H copies instead of two per commitment)
    mutable uint64_t* host_stage = nullptr;
    mutable size_t host_stage_words = 0;
    // fused matrix–vector pipeline (lsr_commit_fused.hpp): lane-major copy of A_hat, per-lane chunk workspaces, side streams
    static constexpr int kMaxSide = 3;
    lsr::DeviceBuffer<double> a_perm;       // [tile][k][k]: the A^T product (n = 2^16 / 2^17)
    // full commitments and openings on the tile pipeline (lsr_commit_tile.hpp; n = 4096, 2^16, 2^17):
    lsr::DeviceBuffer<double> ab_perm;      // [tile][k][k + 1]: A^T and b_hat in one pass (n = 4096: k <= 4; larger n: k <= 3)
    lsr::DeviceBuffer<double> b_perm;       // [tile][k][1]: b_hat alone (larger n, k = 4: the scalar component is a second pass)
    lsr::DeviceBuffer<double> s_perm;       // [tile][k][1]: s_hat, the one-column matrix of an opening
    mutable lsr::DeviceBuffer<uint64_t> ws_rows;                    // wire rows of a chunk on their way to host allocations
    mutable lsr::DeviceBuffer<unsigned long long> ws_vflags;        // openings: per-row OR of decoded ^ claimed
    mutable uint32_t* ws_vbad = nullptr;                            // openings: per-row "not a canonical commitment of this context" (inside ws_vflags)
    mutable lsr::DeviceBuffer<uint64_t> ws_mid, ws_e1_slots;
    mutable hipStream_t side[kMaxSide] = {nullptr, nullptr, nullptr};
    mutable hipEvent_t ev_fork = nullptr, ev_join[kMaxSide] = {nullptr, nullptr, nullptr};
    mutable int n_side = 0;
    // asynchronous entry points share the context's workspaces and side streams: each call's stream first waits for the previous
    // call's last kernel (recorded here), so calls on one context are ordered whatever streams the caller brings
    mutable hipEvent_t ev_last = nullptr;
    // device -> host gathers of the host-array entry points run on their own stream, so that the copy of one chunk overlaps the
    // kernels of the next (config 4: the gather, not the compute, is the long pole)
    mutable hipStream_t copy_stream = nullptr;
    mutable hipEvent_t ev_chunk[2] = {nullptr, nullptr}, ev_copied[2] = {nullptr, nullptr};
    // pipeline selection, read from the environment ONCE when the context is created (include/lambda_snark/batch.h lists the
    // variables): a context never changes the kernels that sign its commitments under the caller's feet
    struct Tuning {
        bool fused = true;     // LAMBDA_SNARK_COMMIT_FUSED=0: the unfused round-1 kernels (A/B runs and tests)
        bool mixed = true;     // LAMBDA_SNARK_COMMIT_MIXED=0: three launches per chunk instead of one mixed launch (n = 2^16, e1 given)
    } tuning;
    // small commitment batches (a single legacy lwe_commit above all): stream keys and messages go up in ONE copy from page-locked memory
    static constexpr size_t kSmallInWords = 8192 + 256;
    mutable uint64_t* host_in = nullptr;              // page-locked, kSmallInWords
    mutable lsr::DeviceBuffer<uint64_t> ws_in;        // its device twin: [keys 4 b | messages b x msg_len]
    mutable lsr::DeviceBuffer<uint64_t> ws_body;      // lwe_verify_opening: the body u || v of one commitment
    // lsr_lwe_commit_keys_device: the seeds of a batch go up through a page-locked block of their own (the call is asynchronous:
    // ev_seeds = that copy has been read, the block may be rewritten)
    mutable uint64_t* host_seeds = nullptr;
    mutable size_t host_seeds_words = 0;
    mutable lsr::DeviceBuffer<uint64_t> ws_seeds;
    mutable hipEvent_t ev_seeds = nullptr;
};

namespace lsr {

static void matvec(const LweContext& c, uint64_t* out, const uint64_t* mat, const uint64_t* vec, const uint64_t* add, uint32_t rows, uint32_t cols,
                   uint32_t row_stride, uint32_t col_stride, uint64_t batch, hipStream_t s) {
    const uint64_t work = batch * rows * (uint64_t)c.n;
    if (!work) return;
    const unsigned grid = static_cast<unsigned>((work + 255) / 256);
    if (c.ntt->use_f64)
        hipLaunchKernelGGL(matvec_kernel<true>, dim3(grid), dim3(256), 0, s, out, mat, vec, add, rows, cols, row_stride, col_stride, (uint32_t)c.logn, batch, c.ntt->mod);
    else
        hipLaunchKernelGGL(matvec_kernel<false>, dim3(grid), dim3(256), 0, s, out, mat, vec, add, rows, cols, row_stride, col_stride, (uint32_t)c.logn, batch, c.ntt->mod);
    LSR_HIP(hipGetLastError());
}

// out[j] = M^(T) vec[j] (+ add) for the k x k matrix; rank-specialised kernels for k <= 4, generic otherwise
template <int K>
static void launch_square(const LweContext& c, uint64_t* out, const uint64_t* mat, const uint64_t* vec, const uint64_t* add, bool transposed,
                          uint64_t batch, hipStream_t s) {
    constexpr int JB = 4;   // witness vectors per lane
    const uint64_t lanes = ((batch + JB - 1) / JB) * c.n;
    const unsigned grid = static_cast<unsigned>((lanes + 255) / 256);
    const uint32_t logn = static_cast<uint32_t>(c.logn);
    if (c.ntt->use_f64) {
        if (transposed) hipLaunchKernelGGL((matvec_square_kernel<K, true, true, JB>), dim3(grid), dim3(256), 0, s, out, mat, vec, add, logn, batch, c.ntt->mod);
        else hipLaunchKernelGGL((matvec_square_kernel<K, true, false, JB>), dim3(grid), dim3(256), 0, s, out, mat, vec, add, logn, batch, c.ntt->mod);
    } else {
        if (transposed) hipLaunchKernelGGL((matvec_square_kernel<K, false, true, JB>), dim3(grid), dim3(256), 0, s, out, mat, vec, add, logn, batch, c.ntt->mod);
        else hipLaunchKernelGGL((matvec_square_kernel<K, false, false, JB>), dim3(grid), dim3(256), 0, s, out, mat, vec, add, logn, batch, c.ntt->mod);
    }
    LSR_HIP(hipGetLastError());
}

static void matvec_square(const LweContext& c, uint64_t* out, const uint64_t* mat, const uint64_t* vec, const uint64_t* add, bool transposed,
                          uint64_t batch, hipStream_t s) {
    if (!batch) return;
    switch (c.k) {
        case 1: launch_square<1>(c, out, mat, vec, add, transposed, batch, s); break;
        case 2: launch_square<2>(c, out, mat, vec, add, transposed, batch, s); break;
        case 3: launch_square<3>(c, out, mat, vec, add, transposed, batch, s); break;
        case 4: launch_square<4>(c, out, mat, vec, add, transposed, batch, s); break;
        default:
            if (transposed) matvec(c, out, mat, vec, add, c.k, c.k, 1, c.k, batch, s);
            else matvec(c, out, mat, vec, add, c.k, c.k, c.k, 1, batch, s);
    }
}

// staging of a batch's stream keys and messages (every commit path), grown on demand
static void ensure_input_space(const LweContext& c, size_t batch) {
    if (batch <= c.ws_in_batch) return;
    c.ws_dm.allocate(batch * c.n);   // message slots of a batch (at most n per commitment); verify's message buffer
    c.ws_keys.allocate(batch * 4);
    LSR_HIP(hipMemset(c.ws_keys.ptr, 0xA5, batch * 32));   // never a valid stale key: a use before the upload shows up in the parity tests
    if (!c.ws_flag.ptr) c.ws_flag.allocate(1);
    c.ws_in_batch = batch;
}
// arrays of the general (unfused) kernels
static void ensure_workspace(const LweContext& c, size_t batch) {
    ensure_input_space(c, batch);
    if (batch <= c.ws_batch) return;
    const size_t kn = (size_t)c.k * c.n;
    c.ws_r.allocate(batch * kn);
    c.ws_e1.allocate(batch * kn);
    c.ws_u.allocate(batch * kn);
    c.ws_e2.allocate(batch * c.n);
    c.ws_v.allocate(batch * c.n);
    c.ws_batch = batch;
}

// the fused pipeline exists for the FP64 flavour, two-pass degrees whose low pass is a full 4096-residue tile, ranks <= 4
static bool fused_eligible(const LweContext& c) {
    return c.tuning.fused && c.ntt->use_f64 && (c.logn == 16 || c.logn == 17) && c.k >= 1 && c.k <= 4;
}

static bool env_flag(const char* name, bool fallback) {
    const char* e = std::getenv(name);
    return (e && (e[0] == '0' || e[0] == '1') && e[1] == 0) ? e[0] == '1' : fallback;
}

// whole commitments / openings in one workgroup (lsr_commit_tile.hpp): the reference's ring degree, FP64 flavour, rank <= 4, a CDT
// table that fits the lanes of a wavefront
static bool tile_eligible(const LweContext& c) {
    return c.tuning.fused && c.ntt->use_f64 && c.logn == 12 && c.k >= 1 && c.k <= 4 && c.cdf_entries <= 64;
}

static LweContext* create_lwe_context(const PublicParams* params, uint64_t key_seed, int device, const ContextKeys* replicate = nullptr) {
    if (!params) return nullptr;                                   // commitment.cpp:103
    uint32_t k = params->module_rank ? params->module_rank : 1;
    const uint32_t n = params->ring_degree;
    if (k > 16 || !(params->sigma > 0.0) || !std::isfinite(params->sigma) || params->sigma > 256.0) {
        set_last_error("lwe_context_create: module_rank must be <= 16 and sigma finite in (0, 256]");
        std::fprintf(stderr, "lwe_context_create error: unsupported module_rank / sigma\n");
        return nullptr;
    }
    const uint64_t q = select_commit_modulus(params->modulus, n);
    const uint64_t t = q ? plain_modulus_for(n) : 0;
    if (!q || !t) {
        set_last_error("lwe_context_create: ring_degree must be a power of two in [2, 131072]");
        std::fprintf(stderr, "lwe_context_create error: unsupported ring_degree %u\n", n);
        return nullptr;
    }
    std::unique_ptr<LweContext> c(new LweContext);
    c->tuning.fused = env_flag("LAMBDA_SNARK_COMMIT_FUSED", true);
    c->tuning.mixed = env_flag("LAMBDA_SNARK_COMMIT_MIXED", true);
    c->params = *params;
    c->q = q; c->t = t; c->delta = q / t; c->n = n; c->k = k; c->sigma = params->sigma;
    // Noise budget: opening decodes round(t/q (round(q m / t) + <e,r> - <s,e1> + e2)); the noise term is a sum of 2 k n products of
    // two sigma-Gaussians, standard deviation sigma^2 sqrt(2 k n); with an 8-sigma tail it must stay below Delta / 2.
    // A context whose FRESH commitments could fail to verify is refused here instead of failing silently later.
    c->noise_unit = 8.0 * std::sqrt(2.0 * k * n) * params->sigma * params->sigma;
    if (c->noise_unit >= 0.5 * static_cast<double>(c->delta)) {
        set_last_error("lwe_context_create: sigma too large for the commitment modulus (noise budget Delta/2 exceeded); pass a wider NTT prime as modulus");
        std::fprintf(stderr, "lwe_context_create error: sigma %.3f exceeds the noise budget of the %d-bit modulus at n=%u, k=%u\n", params->sigma,
                     64 - __builtin_clzll(q), n, k);
        return nullptr;
    }
    c->ntt = create_ntt_context(q, n, device);
    if (!c->ntt) {
        std::fprintf(stderr, "lwe_context_create error: %s\n", last_error_cstr());
        return nullptr;
    }
    c->device = c->ntt->device;
    c->logn = c->ntt->logn;
    try {
        DeviceGuard guard(c->device);
        const std::vector<uint64_t> table = gaussian_cdf(c->sigma);
        c->cdf.upload(table);
        c->cdf_entries = gaussian_scan_entries(table);   // the saturated tail of the table never changes a 63-bit scan
        c->key_seed = key_seed;
        c->keys = replicate ? *replicate : derive_context_keys(key_seed);
        const size_t kn = (size_t)k * n;
        c->a_hat.allocate((size_t)k * kn);
        c->s_hat.allocate(kn);
        c->b_hat.allocate(kn);
        DeviceBuffer<uint64_t> pub_key, sec_key, e_hat(kn);
        pub_key.upload(key_words(c->keys.pub));
        sec_key.upload(key_words(c->keys.sec));
        hipStream_t s = work_stream(*c->ntt);
        launch_uniform(c->a_hat.ptr, pub_key.ptr, 0, k * k, kDomA, n, (uint64_t)k * k, q, s);
        launch_gaussian(GaussianJob{c->s_hat.ptr, sec_key.ptr, 0, k, kDomS, n, k, q}, c->cdf.ptr, c->cdf_entries, s);
        launch_gaussian(GaussianJob{e_hat.ptr, sec_key.ptr, 0, k, kDomE, n, k, q}, c->cdf.ptr, c->cdf_entries, s);
        launch_ntt(*c->ntt, c->s_hat.ptr, k, false, s);
        launch_ntt(*c->ntt, e_hat.ptr, k, false, s);
        // b_hat[i] = sum_j A_hat[i][j] s_hat[j] + e_hat[i]
        matvec_square(*c, c->b_hat.ptr, c->a_hat.ptr, c->s_hat.ptr, e_hat.ptr, false, 1, s);
        auto permute = [&](DeviceBuffer<double>& out, const uint64_t* extra, uint32_t c0, uint32_t nc) {
            out.allocate((size_t)nc * kn);
            hipLaunchKernelGGL(f8_permute_matrix_kernel, dim3(grid_for((uint64_t)nc * kn)), dim3(256), 0, s, out.ptr, c->a_hat.ptr, extra, k, c0, nc, c->logn);
            LSR_HIP(hipGetLastError());
        };
        if (fused_eligible(*c)) permute(c->a_perm, nullptr, 0u, k);
        if (tile_eligible(*c) || (fused_eligible(*c) && c->cdf_entries <= 64)) {
            if (c->logn == 12 || k <= 3 || LSR_K4_SINGLE_PASS) permute(c->ab_perm, c->b_hat.ptr, 0u, k + 1);
            else permute(c->b_perm, c->b_hat.ptr, k, 1u);
            permute(c->s_perm, c->s_hat.ptr, k, 1u);
        }
        LSR_HIP(hipStreamSynchronize(s));
        LSR_HIP(hipMemset(sec_key.ptr, 0, 32));
    } catch (const std::exception& e) {
        set_last_error(std::string("lwe_context_create: ") + e.what());
        std::fprintf(stderr, "lwe_context_create error: %s\n", e.what());
        destroy_ntt_context(c->ntt);
        return nullptr;
    }
    return c.release();
}

static void destroy_lwe_context(LweContext* c) {
    if (!c) return;
    try {
        DeviceGuard guard(c->device);
        if (c->ev_last) (void)hipEventSynchronize(c->ev_last);      // an asynchronous call still running on a caller's stream: let it finish first
        // zeroize the secret key and the scratch that held commitment randomness (commitment.h:34)
        volatile uint32_t* secret = c->keys.sec.w;
        for (int i = 0; i < 8; ++i) secret[i] = 0;
        volatile uint64_t* hk = c->ws_key_host.data();
        for (size_t i = 0; i < c->ws_key_host.size(); ++i) hk[i] = 0;
        // (ws_mid: the fused pipelines' chunk workspace holds transformed commitment randomness between their launches)
        for (lsr::DeviceBuffer<uint64_t>* b : {&c->s_hat, &c->ws_r, &c->ws_e1, &c->ws_e2, &c->ws_keys, &c->ws_mid})
            if (b->ptr) (void)hipMemset(b->ptr, 0, b->count * 8);
        (void)hipDeviceSynchronize();
        c->a_hat.release(); c->s_hat.release(); c->b_hat.release(); c->cdf.release();
        c->ws_r.release(); c->ws_e1.release(); c->ws_e2.release(); c->ws_u.release(); c->ws_v.release();
        c->ws_dm.release(); c->ws_keys.release(); c->ws_flag.release();
        c->a_perm.release(); c->ab_perm.release(); c->b_perm.release(); c->ws_mid.release(); c->ws_rows.release();
        c->ws_vflags.release();
        if (c->s_perm.ptr) (void)hipMemset(c->s_perm.ptr, 0, c->s_perm.count * 8);
        c->s_perm.release();
        if (c->ws_e1_slots.ptr) (void)hipMemset(c->ws_e1_slots.ptr, 0, c->ws_e1_slots.count * 8);
        c->ws_e1_slots.release();
        for (int i = 0; i < c->n_side; ++i) {
            if (c->side[i]) (void)hipStreamDestroy(c->side[i]);
            if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]);
        }
        if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
        if (c->ev_last) (void)hipEventDestroy(c->ev_last);
        if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
        for (int i = 0; i < 2; ++i) {
            if (c->ev_chunk[i]) (void)hipEventDestroy(c->ev_chunk[i]);
            if (c->ev_copied[i]) (void)hipEventDestroy(c->ev_copied[i]);
        }
        if (c->host_stage) (void)hipHostFree(c->host_stage);
        if (c->host_in) {
            volatile uint64_t* hi = c->host_in;
            for (size_t i = 0; i < LweContext::kSmallInWords; ++i) hi[i] = 0;
            (void)hipHostFree(c->host_in);
        }
        if (c->ws_in.ptr) (void)hipMemset(c->ws_in.ptr, 0, c->ws_in.count * 8);
        c->ws_in.release();
        if (c->host_seeds) {
            volatile uint64_t* hs = c->host_seeds;
            for (size_t i = 0; i < c->host_seeds_words; ++i) hs[i] = 0;
            (void)hipHostFree(c->host_seeds);
        }
        if (c->ws_seeds.ptr) (void)hipMemset(c->ws_seeds.ptr, 0, c->ws_seeds.count * 8);
        c->ws_seeds.release();
        if (c->ev_seeds) (void)hipEventDestroy(c->ev_seeds);
        c->ws_body.release();
    } catch (...) {
    }
    destroy_ntt_context(c->ntt);
    delete c;
}

// Build-time constants of the chunk schedules (round 2 measured the alternatives: profiles/r02_mixed_launch.txt,
// r02_fused_commit_stream_sweep.txt; round 3 removed the environment knobs that selected them):
constexpr size_t kMixedChunkBytes = size_t(64) << 20;     // mixed launches: 2 lanes x 2 slots x 64 MiB stay inside the Infinity Cache
constexpr int kMixedLanes = 2;                            // independent pipelines of mixed launches (lane 0 = the caller's stream)
constexpr uint32_t kMixedForwardGroups = 2;               // 256-lane groups per half workgroup in the forward role
constexpr size_t kFusedChunkBytes = size_t(128) << 20;    // three-launch schedule, blinding residues given
#ifndef LSR_SAMPLED_CHUNK_MIB
#define LSR_SAMPLED_CHUNK_MIB 64
#endif
#ifndef LSR_FUSED_STREAMS
#define LSR_FUSED_STREAMS 2
#endif
constexpr size_t kSampledChunkBytes = size_t(LSR_SAMPLED_CHUNK_MIB) << 20;   // three-launch schedule, blinding residues sampled in the strided rounds
#ifndef LSR_FULL_CHUNK_MIB
#define LSR_FULL_CHUNK_MIB 64
#endif
constexpr size_t kFullCommitChunkBytes = size_t(LSR_FULL_CHUNK_MIB) << 20;   // full commitments at n >= 2^16: witness workspace per chunk
constexpr int kFusedStreams = LSR_FUSED_STREAMS;                          // chunk lanes of the three-launch schedule (lane 0 = the caller's stream)
#ifndef LSR_VERIFY_CHUNK_MIB
#define LSR_VERIFY_CHUNK_MIB 128
#endif
#ifndef LSR_VERIFY_STREAMS
#define LSR_VERIFY_STREAMS 2
#endif
// (tools/verify_sweep.sh, 1024 openings at n = 2^16, k = 4: one lane x 128 MiB 1.77 ms, two lanes x 64 / 128 / 192 / 256 MiB 1.70 / 1.62 / 1.70 / 1.73 ms,
// three lanes x 64 MiB 1.73 ms)
constexpr size_t kVerifyChunkBytes = size_t(LSR_VERIFY_CHUNK_MIB) << 20;  // openings at n >= 2^16: workspace of the u vectors per chunk
constexpr int kVerifyStreams = LSR_VERIFY_STREAMS;                        // chunk lanes of the openings (lane 0 = the caller's stream)

template <int K>
static void launch_mid(const LweContext& c, const uint64_t* ws, uint64_t* d_u, size_t vectors, hipStream_t s) {
    const unsigned grid = static_cast<unsigned>(vectors << (c.logn - 12));
    hipLaunchKernelGGL((mlwe_mid_fused8<K>), dim3(grid), dim3(kF8Threads), 0, s, ws, d_u, c.a_perm.ptr, (uint32_t)vectors, c.ntt->mod, c.ntt->fwd_f64.ptr,
                       c.ntt->inv_f64.ptr);
    LSR_HIP(hipGetLastError());
}

// lanes 1.. of a chunk schedule (lane 0 is the caller's stream: every stream a process opens competes for the runtime's few
// hardware queues, and two lanes that land on one queue overlap nothing — profiles/r02_commit_hw_queues.txt)
static void ensure_side_streams(const LweContext& c, int lanes) {
    while (c.n_side < lanes - 1) {
        LSR_HIP(hipStreamCreateWithFlags(&c.side[c.n_side], hipStreamNonBlocking));
        LSR_HIP(hipEventCreateWithFlags(&c.ev_join[c.n_side], hipEventDisableTiming));
        ++c.n_side;
    }
    if (!c.ev_fork) LSR_HIP(hipEventCreateWithFlags(&c.ev_fork, hipEventDisableTiming));
}
static void fork_lanes(const LweContext& c, hipStream_t s, int lanes) {
    if (lanes <= 1) return;
    LSR_HIP(hipEventRecord(c.ev_fork, s));
    for (int i = 1; i < lanes; ++i) LSR_HIP(hipStreamWaitEvent(c.side[i - 1], c.ev_fork, 0));
}
static void join_lanes(const LweContext& c, hipStream_t s, int lanes) {
    for (int i = 1; i < lanes; ++i) {
        LSR_HIP(hipEventRecord(c.ev_join[i - 1], c.side[i - 1]));
        LSR_HIP(hipStreamWaitEvent(s, c.ev_join[i - 1], 0));
    }
}

// bracket of an asynchronous entry point (caller holds c.mutex): order this call behind the previous one on the same context
// (a stream that records into a HIP graph: no bracket — lsr_runtime.hpp, stream_is_capturing)
static void begin_async(const LweContext& c, hipStream_t s) {
    if (c.ev_last && !stream_is_capturing(s)) LSR_HIP(hipStreamWaitEvent(s, c.ev_last, 0));
}
static void end_async(const LweContext& c, hipStream_t s) {
    if (stream_is_capturing(s)) return;
    if (!c.ev_last) LSR_HIP(hipEventCreateWithFlags(&c.ev_last, hipEventDisableTiming));
    LSR_HIP(hipEventRecord(c.ev_last, s));
}
// a SYNCHRONOUS entry point (its work runs on the context's own stream and is complete when it returns; caller holds c.mutex):
// an asynchronous call enqueued earlier on a caller's stream may still be using the workspaces — wait for it first
static void wait_for_async(const LweContext& c) {
    if (c.ev_last) LSR_HIP(hipEventSynchronize(c.ev_last));
}

// MIXED schedule of the 4 + 12 pipeline (n = 2^16, blinding residues given; caller holds c.mutex): launch t carries the
// middle stage of chunk t, the forward strided round of chunk t + 1 and the inverse strided round of chunk t - 1 as roles of one
// kernel (lsr_commit_fused.hpp, mlwe_mixed), t = -1 .. chunks; two workspace slots alternate.  No events inside a lane: the launch
// boundaries are the dependencies.
template <int K>
static void launch_mixed(const LweContext& c, const MixedJob& job, hipStream_t s) {
    const unsigned grid = (job.units_m + job.units_f + job.units_i) * 8u;
    if (!grid) return;
    hipLaunchKernelGGL((mlwe_mixed<K>), dim3(grid), dim3(kF8Threads), 0, s, job, c.a_perm.ptr, c.ntt->mod, c.ntt->fwd_f64.ptr, c.ntt->inv_f64.ptr,
                       RoundConsts<ArithF64>{c.ntt->n_inv_f64, c.ntt->w_last_scaled_f64});
    LSR_HIP(hipGetLastError());
}

static void mlwe_matvec_mixed(const LweContext& c, const uint64_t* d_r, const uint64_t* d_e1, uint64_t* d_u, size_t batch, hipStream_t s) {
    const uint32_t k = c.k;
    const size_t vec_words = (size_t)k << c.logn;
    const size_t chunk = std::max<size_t>(1, kMixedChunkBytes / (vec_words * 8));
    const long chunks = static_cast<long>((batch + chunk - 1) / chunk);
    // the chunks alternate between independent pipelines of mixed launches on their own streams, so that the drain of one lane's
    // launch is filled by the other's workgroups — all of one shape, so the dispatcher has no reason to starve either
    // (3.05 -> 2.95 ms per 1024 vectors, profiles/r02_mixed_launch.txt)
    const int lanes = static_cast<int>(std::min<long>(kMixedLanes, chunks));
    const size_t slot_words = std::min(chunk, batch) * vec_words;
    if (c.ws_mid.count < slot_words * 2 * lanes) c.ws_mid.allocate(slot_words * 2 * lanes);
    ensure_side_streams(c, lanes);
    fork_lanes(c, s, lanes);
    for (long t = -1;; ++t) {
        bool any = false;
        for (int lane = 0; lane < lanes; ++lane) {
            // lane's sequence: chunks lane, lane + lanes, ...; position t of it is chunk lane + t * lanes
            auto chunk_at = [&](long pos) { return pos < 0 ? -1L : (long)lane + pos * lanes; };
            auto valid = [&](long pos) { const long ci = chunk_at(pos); return ci >= 0 && ci < chunks; };
            auto first_of = [&](long pos) { return (size_t)chunk_at(pos) * chunk; };
            auto count_of = [&](long pos) { return std::min(chunk, batch - first_of(pos)); };
            uint64_t* const ws = c.ws_mid.ptr + (size_t)lane * 2 * slot_words;
            MixedJob job{};
            if (valid(t)) {
                job.m_ws = ws + (size_t)(t & 1) * slot_words;
                job.m_out = d_u + first_of(t) * vec_words;
                job.m_vectors = (uint32_t)count_of(t);
                job.units_m = job.m_vectors << (c.logn - 12 - 3);
            }
            if (valid(t + 1)) {
                job.f_dst = ws + (size_t)((t + 1) & 1) * slot_words;
                job.f_src = d_r + first_of(t + 1) * vec_words;
                job.f_polys = (uint32_t)(count_of(t + 1) * k);
                // a unit = 8 workgroups of 512 lanes x 16 residues = one polynomial; two polynomials with two groups per lane
                job.f_groups = kMixedForwardGroups;
                job.units_f = (job.f_polys + job.f_groups - 1) / job.f_groups;
            }
            if (t >= 1 && valid(t - 1)) {
                job.i_data = d_u + first_of(t - 1) * vec_words;
                job.i_add = d_e1 + first_of(t - 1) * vec_words;
                job.i_polys = (uint32_t)(count_of(t - 1) * k);
                job.units_i = job.i_polys;
            }
            const uint32_t units_s = job.units_f + job.units_i;
            if (!job.units_m && !units_s) continue;
            any = true;
            if (job.units_m && units_s) {
                // strided units that follow one middle unit in the block order, in proportion to the launch's work.  The period
                // (ratio + 1) must stay odd: blocks are dealt to the XCDs and their CUs round-robin, and with an even period the
                // middle-stage workgroups pile up on a fraction of the CUs (profiles/r02_mixed_launch.txt: ratio 3 -> 5.5 ms,
                // ratio 2 -> 3.07 ms per 1024 vectors)
                uint32_t ratio = std::max<uint32_t>(1u, (units_s + job.units_m / 2) / job.units_m);
                ratio += ratio & 1u;
                job.s_per_m = ratio;
                job.periods = std::min(job.units_m, units_s / job.s_per_m);
            }
            hipStream_t st = lane == 0 ? s : c.side[lane - 1];
            switch (k) {
                case 1: launch_mixed<1>(c, job, st); break;
                case 2: launch_mixed<2>(c, job, st); break;
                case 3: launch_mixed<3>(c, job, st); break;
                default: launch_mixed<4>(c, job, st); break;
            }
        }
        if (!any && t >= 0) break;
    }
    join_lanes(c, s, lanes);
}

// Fused pipeline (caller holds c.mutex): per chunk of witness vectors
//   top forward round r -> workspace | 12 forward stages x k, A_hat^T product, 12 inverse stages x k -> u | top inverse round (+ e1)
// with the chunks dealt round-robin to two lanes.  d_r is only read.
// d_e1 == NULL && d_keys != NULL: the blinding residues are SAMPLED inside the two strided rounds of a chunk (domain 5, per-vector
// keys d_keys[batch][4]; the forward round draws the first half of the rows into an int8 side slot, the inverse round the other
// half) — no [batch][k][n] array of e1 ever exists.  Tables with more than 127 entries (sigma > ~13.7) sample everything in the
// inverse round.
static void mlwe_matvec_fused(const LweContext& c, const uint64_t* d_r, const uint64_t* d_e1, uint64_t* d_u, size_t batch, hipStream_t s,
                              const uint64_t* d_keys = nullptr) {
    // default for n = 2^16 with the blinding residues given: mixed launches (one kernel, three roles; see mlwe_matvec_mixed)
    if (c.logn == 16 && d_e1 && !d_keys && c.tuning.mixed) {
        mlwe_matvec_mixed(c, d_r, d_e1, d_u, batch, s);
        return;
    }
    const uint32_t k = c.k;
    const size_t vec_words = (size_t)k << c.logn;
    const bool sample = !d_e1 && d_keys;
    const size_t chunk = std::max<size_t>(1, (sample ? kSampledChunkBytes : kFusedChunkBytes) / (vec_words * 8));
    const int streams = static_cast<int>(std::min<size_t>((size_t)kFusedStreams, (batch + chunk - 1) / chunk));
    ensure_side_streams(c, streams);
    auto lane = [&](size_t i) { return i == 0 ? s : c.side[i - 1]; };
    const size_t slot_words = std::min(chunk, batch) * vec_words;
    if (c.ws_mid.count < slot_words * streams) c.ws_mid.allocate(slot_words * streams);
    const bool split_sampling = sample && c.cdf_entries <= 127;
    const size_t side_words = slot_words / 16;                    // int8 samples of half the rows: 1/16 of the chunk's words
    if (split_sampling && c.ws_e1_slots.count < side_words * streams) c.ws_e1_slots.allocate(side_words * streams);
    fork_lanes(c, s, streams);
    size_t index = 0;
    for (size_t first = 0; first < batch; first += chunk, ++index) {
        const size_t now = std::min(chunk, batch - first);
        hipStream_t st = lane(index % streams);
        uint64_t* const ws = c.ws_mid.ptr + (index % streams) * slot_words;
        uint64_t* const out = d_u + first * vec_words;
        BlindSampler bs{sample ? d_keys + 4 * first : nullptr, c.cdf.ptr, c.cdf_entries, k, kDomE1, nullptr};
        if (split_sampling) bs.side = c.ws_e1_slots.ptr + (index % streams) * side_words;
        if (bs.side) launch_top_round_forward_sampling(*c.ntt, ws, d_r + first * vec_words, now * k, st, bs);
        else launch_top_round_forward(*c.ntt, ws, d_r + first * vec_words, now * k, st);
        switch (k) {
            case 1: launch_mid<1>(c, ws, out, now, st); break;
            case 2: launch_mid<2>(c, ws, out, now, st); break;
            case 3: launch_mid<3>(c, ws, out, now, st); break;
            default: launch_mid<4>(c, ws, out, now, st); break;
        }
        if (sample) launch_top_round_inverse_sampled(*c.ntt, out, now * k, st, bs);
        else launch_top_round_inverse(*c.ntt, out, now * k, st, d_e1 + first * vec_words);
    }
    join_lanes(c, s, streams);
}

// u = INTT(A_hat^T NTT(r)) + e1 on device-resident [batch][k][n] arrays.  Unfused form: r is overwritten by NTT(r), which
// commit_compute then reuses for the scalar component; `allow_fused` callers do not need NTT(r) and leave r untouched when
// the context qualifies for the fused pipeline (they hold c.mutex).
static void mlwe_matvec_device(const LweContext& c, uint64_t* d_r, const uint64_t* d_e1, uint64_t* d_u, size_t batch, hipStream_t s,
                               bool allow_fused = false) {
    const uint32_t k = c.k;
    if (allow_fused && c.a_perm.ptr) {
        mlwe_matvec_fused(c, d_r, d_e1, d_u, batch, s);
        return;
    }
    launch_ntt(*c.ntt, d_r, batch * k, false, s);
    // u[j][col] = sum_i A_hat[i][col] r_hat[j][i]
    matvec_square(c, d_u, c.a_hat.ptr, d_r, nullptr, true, batch, s);
    // inverse transform with the blinding add fused into its final store
    launch_ntt(*c.ntt, d_u, batch * k, true, s, d_e1);
}

// pinned host staging (guarded by c.mutex), grown on demand
static void ensure_host_stage(const LweContext& c, size_t words) {
    if (words <= c.host_stage_words) return;
    if (c.host_stage) (void)hipHostFree(c.host_stage);
    c.host_stage = nullptr;
    c.host_stage_words = 0;
    LSR_HIP(hipHostMalloc(reinterpret_cast<void**>(&c.host_stage), words * 8, hipHostMallocPortable));   // any device of the node may DMA into it
    c.host_stage_words = words;
}

static LweCommitment* new_commitment(size_t words) {
    std::unique_ptr<LweCommitment> out(new LweCommitment);
    out->len = words;
    out->data = new uint64_t[words];
    return out.release();
}

// The per-commitment stream keys and the messages of a batch, staged on the device (enqueued on `s`).
// Host prep (lsr_keys.hpp): seed == 0 => 256 bits of fresh entropy (commitment.h:52), else PRF(seed, context id, embedded message)
// — a reused seed never repeats the blinding across messages or contexts.  Only the first `copy` slots of each message matter
// (commitment.cpp:146-149); rows keep their msg_len pitch.
// keys[j][4] of `batch` commitments (lsr_keys.hpp).  The PRF runs over the embedded message (two multiplications mod 2^61 - 1 per
// slot), so a batch of full-length messages costs the host about 10 us per commitment on one core — more than the GPU spends on it;
// batches above ~0.5 M message words are split over up to eight threads (the derivation is a pure function of its inputs)
static void derive_commit_keys(const LweContext& c, const uint64_t* messages, size_t msg_len, size_t batch, const uint64_t* seeds, uint64_t* keys) {
    const size_t copy = std::min<size_t>(msg_len, c.n);
    auto slice = [&](size_t lo, size_t hi) {
        for (size_t j = lo; j < hi; ++j) {
            const StreamKey key = seeds && seeds[j] ? derive_commit_key(seeds[j], c.keys.id, messages + j * msg_len, copy, c.t) : fresh_key();
            key_words(key, keys + 4 * j);
        }
    };
    const size_t work = batch * (copy + 64);
    const size_t workers = std::min<size_t>(8, std::max<size_t>(1, work >> 19));
    if (workers <= 1 || batch < 2 * workers) { slice(0, batch); return; }
    std::vector<std::thread> pool;
    std::vector<std::exception_ptr> errors(workers);
    const size_t per = (batch + workers - 1) / workers;
    for (size_t w = 0; w < workers; ++w) {
        const size_t lo = w * per, hi = std::min(batch, lo + per);
        if (lo < hi)
            pool.emplace_back([&, w, lo, hi] {
                try { slice(lo, hi); } catch (...) { errors[w] = std::current_exception(); }     // e.g. the OS entropy source failing
            });
    }
    for (std::thread& th : pool) th.join();
    for (const std::exception_ptr& e : errors)
        if (e) std::rethrow_exception(e);
}

struct StagedInputs {
    const uint64_t* d_keys = nullptr;
    const uint64_t* d_msgs = nullptr;
    DeviceBuffer<uint64_t> big_msgs;       // a message batch larger than the context's scratch: dies with this object (after a sync)
};
// the device form of derive_commit_keys (lsr_commit_keys.hpp): d_seeds[batch] all non-zero, d_msgs[batch][msg_len] -> d_keys[batch][4]
static void launch_commit_keys(const LweContext& c, const uint64_t* d_msgs, size_t msg_len, size_t batch, const uint64_t* d_seeds, uint64_t* d_keys,
                               hipStream_t s) {
    CommitKeysJob job{};
    job.keys = d_keys;
    job.msgs = d_msgs;
    job.seeds = d_seeds;
    job.msg_len = msg_len;
    job.copy = static_cast<uint32_t>(std::min<size_t>(msg_len, c.n));
    for (int i = 0; i < 4; ++i) job.id[i] = c.keys.id[i];
    job.t = c.t;
    hipLaunchKernelGGL(commit_keys_kernel, dim3(static_cast<unsigned>(batch)), dim3(kKeyThreads), 0, s, job);
    LSR_HIP(hipGetLastError());
}

static void stage_commit_inputs(const LweContext& c, const uint64_t* messages, size_t msg_len, size_t batch, const uint64_t* seeds, hipStream_t s,
                                StagedInputs* in) {
    ensure_input_space(c, batch);
    std::vector<uint64_t>& key_host = c.ws_key_host;
    key_host.resize(batch * 4);
    const size_t copy = std::min<size_t>(msg_len, c.n);
    // long messages: the keys are derived on the device from the uploaded messages (the host derivation hashes every embedded word —
    // about 10 us per full-length message and core, more than the commitment costs the GPU); seed 0 = fresh OS entropy stays here
    const bool device_keys = seeds && batch * copy >= (size_t(1) << 16) && std::all_of(seeds, seeds + batch, [](uint64_t v) { return v != 0; });
    if (!device_keys) derive_commit_keys(c, messages, msg_len, batch, seeds, key_host.data());
    const size_t in_words = batch * 4 + (copy ? batch * msg_len : 0);
    if (in_words <= LweContext::kSmallInWords) {
        // one upload from page-locked memory instead of two staged ones (every caller synchronises the stream before it returns, so
        // the staging area is free again by the next call)
        if (!c.host_in) {
            LSR_HIP(hipHostMalloc(reinterpret_cast<void**>(&c.host_in), LweContext::kSmallInWords * 8, hipHostMallocPortable));
            c.ws_in.allocate(LweContext::kSmallInWords);
        }
        std::memcpy(c.host_in, key_host.data(), batch * 32);
        if (copy) std::memcpy(c.host_in + batch * 4, messages, batch * msg_len * 8);
        LSR_HIP(hipMemcpyAsync(c.ws_in.ptr, c.host_in, in_words * 8, hipMemcpyHostToDevice, s));
        in->d_keys = c.ws_in.ptr;
        in->d_msgs = c.ws_in.ptr + batch * 4;
        return;
    }
    if (device_keys) {   // (every caller synchronises `s` before it returns: the caller's seed array is read by then)
        if (c.ws_seeds.count < batch) c.ws_seeds.allocate(std::max<size_t>(batch, 4096));
        LSR_HIP(hipMemcpyAsync(c.ws_seeds.ptr, seeds, batch * 8, hipMemcpyHostToDevice, s));
    } else {
        LSR_HIP(hipMemcpyAsync(c.ws_keys.ptr, key_host.data(), batch * 32, hipMemcpyHostToDevice, s));
    }
    uint64_t* d_msgs = c.ws_dm.ptr;
    if (batch * msg_len > c.ws_dm.count) {
        in->big_msgs.allocate(batch * msg_len);
        d_msgs = in->big_msgs.ptr;
    }
    if (copy) LSR_HIP(hipMemcpyAsync(d_msgs, messages, batch * msg_len * 8, hipMemcpyHostToDevice, s));
    if (device_keys) launch_commit_keys(c, d_msgs, msg_len, batch, c.ws_seeds.ptr, c.ws_keys.ptr, s);
    in->d_keys = c.ws_keys.ptr;
    in->d_msgs = d_msgs;
}

// General form (any degree, rank and arithmetic flavour the context supports): r, e1, e2 sampled into arrays, transforms and
// products as separate kernels, the rows assembled at the end.  d_keys [batch][4], d_msgs [batch][msg_len] on the device.
static void commit_rows_general(const LweContext& c, const uint64_t* d_msgs, size_t msg_len, size_t batch, const uint64_t* d_keys, uint64_t* d_rows,
                                hipStream_t s) {
    const uint32_t n = c.n, k = c.k;
    const size_t copy = std::min<size_t>(msg_len, n);
    ensure_workspace(c, batch);
    launch_gaussian3(GaussianJob{c.ws_r.ptr, d_keys, 0, k, kDomR, n, batch * k, c.q}, GaussianJob{c.ws_e1.ptr, d_keys, 0, k, kDomE1, n, batch * k, c.q},
                     GaussianJob{c.ws_e2.ptr, d_keys, 0, 1, kDomE2, n, batch, c.q}, c.cdf.ptr, c.cdf_entries, s);
    mlwe_matvec_device(c, c.ws_r.ptr, c.ws_e1.ptr, c.ws_u.ptr, batch, s);    // leaves r_hat in ws_r
    // v = INTT(<b_hat, r_hat>) + e2 + round(q m / t)
    matvec(c, c.ws_v.ptr, c.b_hat.ptr, c.ws_r.ptr, nullptr, 1, k, 0, 1, batch, s);
    launch_ntt(*c.ntt, c.ws_v.ptr, batch, true, s);
    const uint64_t vcount = (uint64_t)batch * n;
    hipLaunchKernelGGL(finish_v_kernel, dim3(grid_for(vcount)), dim3(256), 0, s, c.ws_v.ptr, c.ws_e2.ptr, d_msgs, (uint64_t)msg_len, (uint64_t)copy,
                       (uint32_t)c.logn, vcount, c.delta, c.t, c.q);
    const uint64_t kn = (uint64_t)k * n, words = kHeaderWords + kn + n;
    hipLaunchKernelGGL(pack_commitments_kernel, dim3(grid_for(batch * words)), dim3(256), 0, s, d_rows, c.ws_u.ptr, c.ws_v.ptr, kn, (uint64_t)n, (uint64_t)batch,
                       c.q, c.t, n | ((uint64_t)k << 32));
    LSR_HIP(hipGetLastError());
}

// n = 4096: one launch, one workgroup per commitment (lsr_commit_tile.hpp)
template <int K>
static void launch_commit_tile(const LweContext& c, const CommitTileJob& job, hipStream_t s) {
    hipLaunchKernelGGL((commit_tile_kernel<K>), dim3(job.batch), dim3(kF8Threads), 0, s, job, c.ab_perm.ptr, c.ntt->mod, c.ntt->fwd_f64.ptr, c.ntt->inv_f64.ptr,
                       RoundConsts<ArithF64>{c.ntt->n_inv_f64, c.ntt->w_last_scaled_f64});
    LSR_HIP(hipGetLastError());
}
static void commit_rows_tile(const LweContext& c, const uint64_t* d_msgs, size_t msg_len, size_t batch, const uint64_t* d_keys, uint64_t* d_rows, hipStream_t s) {
    const size_t row_words = kHeaderWords + ((size_t)c.k + 1) * c.n;
    for (size_t first = 0; first < batch; first += 0x40000000u) {            // grid limit: 2^30 workgroups per launch
        const size_t now = std::min<size_t>(batch - first, 0x40000000u);
        const CommitTileJob job{d_rows + first * row_words, d_keys + 4 * first, d_msgs + first * msg_len, (uint64_t)msg_len, (uint64_t)std::min<size_t>(msg_len, c.n),
                                c.cdf.ptr, c.cdf_entries, (uint32_t)now, c.q, c.t};
        switch (c.k) {
            case 1: launch_commit_tile<1>(c, job, s); break;
            case 2: launch_commit_tile<2>(c, job, s); break;
            case 3: launch_commit_tile<3>(c, job, s); break;
            default: launch_commit_tile<4>(c, job, s); break;
        }
    }
}

// n = 2^16 / 2^17: per chunk of commitments, the top forward round with r sampled in the pass, the tile pipeline with the
// [A^T | b_hat] product writing raw elements into the rows, the top inverse round in place with e1 / e2 / the message in the pass
// (lsr_commit_tile.hpp); the chunks alternate between two lanes like the matrix-vector workload's three-launch schedule
template <int K, int NC>
static void launch_mid_general(const LweContext& c, const uint64_t* ws, size_t in_pitch, uint64_t* out, size_t out_pitch, const double* mat, size_t vectors,
                               hipStream_t s) {
    hipLaunchKernelGGL((mlwe_mid_general<K, NC>), dim3(static_cast<unsigned>(vectors << (c.logn - 12))), dim3(kF8Threads), 0, s, ws, in_pitch, out, out_pitch, mat,
                       (uint32_t)vectors, c.ntt->mod, c.ntt->fwd_f64.ptr, c.ntt->inv_f64.ptr);
    LSR_HIP(hipGetLastError());
}
static void commit_rows_fused(const LweContext& c, const uint64_t* d_msgs, size_t msg_len, size_t batch, const uint64_t* d_keys, uint64_t* d_rows,
                              hipStream_t s) {
    const uint32_t k = c.k, n = c.n;
    const size_t vec_words = (size_t)k << c.logn, row_words = kHeaderWords + ((size_t)k + 1) * n;
    const size_t chunk = std::max<size_t>(1, kFullCommitChunkBytes / (vec_words * 8));
    const int streams = static_cast<int>(std::min<size_t>((size_t)kFusedStreams, (batch + chunk - 1) / chunk));
    ensure_side_streams(c, streams);
    const size_t slot_words = std::min(chunk, batch) * vec_words;
    if (c.ws_mid.count < slot_words * streams) c.ws_mid.allocate(slot_words * streams);
    const RoundConsts<ArithF64> cs{c.ntt->n_inv_f64, c.ntt->w_last_scaled_f64};
    const int r = c.logn - 12, lo = 12;                                  // top R = 4 (n = 2^16) or 5 (2^17) index bits in the outer rounds
    fork_lanes(c, s, streams);
    size_t index = 0;
    for (size_t first = 0; first < batch; first += chunk, ++index) {
        const size_t now = std::min(chunk, batch - first);
        hipStream_t st = index % streams == 0 ? s : c.side[index % streams - 1];
        uint64_t* const ws = c.ws_mid.ptr + (index % streams) * slot_words;
        uint64_t* const rows = d_rows + first * row_words;
        const CommitTopJob job{rows, ws, d_keys + 4 * first, d_msgs + first * msg_len, (uint64_t)msg_len, (uint64_t)std::min<size_t>(msg_len, n), c.cdf.ptr,
                               c.cdf_entries, (uint32_t)now, k, (uint64_t)row_words, c.q, c.t};
        const unsigned grid_f = static_cast<unsigned>((now * k << c.logn) >> (r + 8)), grid_i = static_cast<unsigned>((now * (k + 1) << c.logn) >> (r + 8));
        if (r == 4) hipLaunchKernelGGL((commit_top_forward_kernel<4>), dim3(grid_f), dim3(256), 0, st, job, lo, c.ntt->mod, c.ntt->fwd_f64.ptr);
        else hipLaunchKernelGGL((commit_top_forward_kernel<5>), dim3(grid_f), dim3(256), 0, st, job, lo, c.ntt->mod, c.ntt->fwd_f64.ptr);
        uint64_t* const body = rows + kHeaderWords;
        switch (k) {
            case 1: launch_mid_general<1, 2>(c, ws, vec_words, body, row_words, c.ab_perm.ptr, now, st); break;
            case 2: launch_mid_general<2, 3>(c, ws, vec_words, body, row_words, c.ab_perm.ptr, now, st); break;
            case 3: launch_mid_general<3, 4>(c, ws, vec_words, body, row_words, c.ab_perm.ptr, now, st); break;
            default:
#if LSR_K4_SINGLE_PASS     // five accumulators: 128 VGPRs with a few spilled registers, against a second one-column pass over the workspace
                launch_mid_general<4, 5>(c, ws, vec_words, body, row_words, c.ab_perm.ptr, now, st);
#else
                launch_mid_general<4, 4>(c, ws, vec_words, body, row_words, c.a_perm.ptr, now, st);
                launch_mid_general<4, 1>(c, ws, vec_words, body + vec_words, row_words, c.b_perm.ptr, now, st);
#endif
        }
        if (r == 4) hipLaunchKernelGGL((commit_top_inverse_kernel<4>), dim3(grid_i), dim3(256), 0, st, job, lo, c.ntt->mod, c.ntt->inv_f64.ptr, cs);
        else hipLaunchKernelGGL((commit_top_inverse_kernel<5>), dim3(grid_i), dim3(256), 0, st, job, lo, c.ntt->mod, c.ntt->inv_f64.ptr, cs);
        LSR_HIP(hipGetLastError());
    }
    join_lanes(c, s, streams);
}

// wire rows d_rows[batch][5 + (k + 1) n] of `batch` commitments from device-resident keys and messages, enqueued on `s` (caller
// holds c.mutex): the reference's lwe_commit (commitment.cpp:138-164) for a whole batch without a byte of host traffic
static void commit_rows_device(const LweContext& c, const uint64_t* d_msgs, size_t msg_len, size_t batch, const uint64_t* d_keys, uint64_t* d_rows,
                               hipStream_t s) {
    if (!batch) return;
    if (c.ab_perm.ptr && c.logn == 12) commit_rows_tile(c, d_msgs, msg_len, batch, d_keys, d_rows, s);
    else if (c.a_perm.ptr && (c.ab_perm.ptr || c.b_perm.ptr)) commit_rows_fused(c, d_msgs, msg_len, batch, d_keys, d_rows, s);
    else commit_rows_general(c, d_msgs, msg_len, batch, d_keys, d_rows, s);
}

static void ensure_copy_stream(const LweContext& c) {
    if (c.copy_stream) return;
    // A stream of ANOTHER priority than the compute streams: the runtime multiplexes a process's streams onto a few hardware queues
    // per priority level, and a copy stream that lands on the compute stream's queue serialises with it — the copy of piece i then
    // sits in front of piece i + 1's kernels and nothing overlaps (seen in bench.py, whose process has opened a dozen streams by then:
    // 5.7 ms per 2048 rows against 3.9 ms in a fresh process; profiles/r02_commit_hw_queues.txt is the same effect between lanes).
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || least == greatest ||
        hipStreamCreateWithPriority(&c.copy_stream, hipStreamNonBlocking, greatest) != hipSuccess) {
        (void)hipGetLastError();
        LSR_HIP(hipStreamCreateWithFlags(&c.copy_stream, hipStreamNonBlocking));
    }
    for (int i = 0; i < 2; ++i) {
        LSR_HIP(hipEventCreateWithFlags(&c.ev_chunk[i], hipEventDisableTiming));
        LSR_HIP(hipEventCreateWithFlags(&c.ev_copied[i], hipEventDisableTiming));
    }
}

// `batch` commitments to a HOST array, `chunk` at a time through two row buffers: while chunk i's rows travel to the host on the copy
// stream, chunk i + 1 is staged and computed (caller holds c.mutex)
static void commit_batch_flat_host(const LweContext& c, const uint64_t* messages, size_t msg_len, size_t batch, const uint64_t* seeds, uint64_t* out_words,
                                   size_t chunk) {
    const size_t words = kHeaderWords + ((size_t)c.k + 1) * c.n;
    hipStream_t s = work_stream(*c.ntt);
    ensure_copy_stream(c);
    const size_t slot = std::min(chunk, batch) * words;
    const bool two = batch > chunk;
    if (c.ws_rows.count < slot * (two ? 2 : 1)) c.ws_rows.allocate(slot * (two ? 2 : 1));
    size_t index = 0;
    try {
        for (size_t done = 0; done < batch; done += chunk, ++index) {
            const size_t now = std::min(chunk, batch - done);
            const int b = static_cast<int>(index & 1);
            uint64_t* const rows = c.ws_rows.ptr + (two ? b * slot : 0);
            if (index >= 2) LSR_HIP(hipEventSynchronize(c.ev_copied[b]));            // this buffer's previous rows have left
            StagedInputs in;
            stage_commit_inputs(c, messages + done * msg_len, msg_len, now, seeds ? seeds + done : nullptr, s, &in);
            commit_rows_device(c, in.d_msgs, msg_len, now, in.d_keys, rows, s);
            LSR_HIP(hipEventRecord(c.ev_chunk[b], s));
            LSR_HIP(hipStreamWaitEvent(c.copy_stream, c.ev_chunk[b], 0));
            LSR_HIP(hipMemcpyAsync(out_words + done * words, rows, now * words * 8, hipMemcpyDeviceToHost, c.copy_stream));
            LSR_HIP(hipEventRecord(c.ev_copied[b], c.copy_stream));
            LSR_HIP(hipStreamSynchronize(s));      // the staging areas of the inputs (pinned host block, key vector) are reused by the next chunk
        }
    } catch (...) {
        // the caller gets -1 and may free `out_words` at once: no copy of an earlier piece may still be writing into it
        (void)hipStreamSynchronize(s);
        (void)hipStreamSynchronize(c.copy_stream);
        throw;
    }
    LSR_HIP(hipStreamSynchronize(c.copy_stream));
}

// out_words: host array (the rows come back in one copy) or, with `to_device`, device memory the rows are assembled in
static void commit_chunk_flat(const LweContext& c, const uint64_t* messages, size_t msg_len, size_t batch, const uint64_t* seeds, uint64_t* out_words,
                              bool to_device) {
    const size_t words = kHeaderWords + ((size_t)c.k + 1) * c.n;
    hipStream_t s = work_stream(*c.ntt);
    StagedInputs in;
    stage_commit_inputs(c, messages, msg_len, batch, seeds, s, &in);
    uint64_t* rows = out_words;
    if (!to_device) {
        if (c.ws_rows.count < batch * words) c.ws_rows.allocate(batch * words);
        rows = c.ws_rows.ptr;
    }
    commit_rows_device(c, in.d_msgs, msg_len, batch, in.d_keys, rows, s);
    if (!to_device) LSR_HIP(hipMemcpyAsync(out_words, rows, batch * words * 8, hipMemcpyDeviceToHost, s));
    LSR_HIP(hipStreamSynchronize(s));
}

static void commit_chunk(const LweContext& c, const uint64_t* messages, size_t msg_len, size_t batch, const uint64_t* seeds, LweCommitment** out) {
    const size_t words = kHeaderWords + ((size_t)c.k + 1) * c.n;
    hipStream_t s = work_stream(*c.ntt);
    StagedInputs in;
    stage_commit_inputs(c, messages, msg_len, batch, seeds, s, &in);
    if (c.ws_rows.count < batch * words) c.ws_rows.allocate(batch * words);
    commit_rows_device(c, in.d_msgs, msg_len, batch, in.d_keys, c.ws_rows.ptr, s);
    // gather: the rows of the whole chunk come back in one bulk copy into pinned memory; the per-commitment arrays (which the ABI
    // wants as separate new[] allocations, commitment.cpp:50-57) are filled from there by a few threads
    ensure_host_stage(c, batch * words);
    LSR_HIP(hipMemcpyAsync(c.host_stage, c.ws_rows.ptr, batch * words * 8, hipMemcpyDeviceToHost, s));
    std::vector<LweCommitment*> made(batch, nullptr);
    const size_t workers = std::min<size_t>(8, std::max<size_t>(1, (batch * words * 8) >> 22));   // one thread per ~4 MiB, at most 8
    const size_t per = (batch + workers - 1) / workers;
    auto in_parallel = [&](auto&& body) {                   // body(lo, hi) over disjoint slices of the batch
        if (workers <= 1) { body(size_t(0), batch); return; }
        std::vector<std::thread> pool;
        for (size_t w = 0; w < workers; ++w) {
            const size_t lo = w * per, hi = std::min(batch, lo + per);
            if (lo < hi) pool.emplace_back(body, lo, hi);
        }
        for (std::thread& th : pool) th.join();
    };
    // the allocations (98 KB each at the reference's parameters, first touch included) overlap the copy and each other
    std::atomic<bool> failed{false};
    in_parallel([&](size_t lo, size_t hi) {
        try {
            for (size_t j = lo; j < hi; ++j) made[j] = new_commitment(words);
        } catch (...) {
            failed = true;
        }
    });
    const hipError_t synced = hipStreamSynchronize(s);
    if (failed || synced != hipSuccess) {
        for (LweCommitment* m : made) {
            if (m) { delete[] m->data; delete m; }
        }
        if (failed) throw std::bad_alloc();
        throw HipFailure(std::string("hipStreamSynchronize: ") + hipGetErrorString(synced));
    }
    in_parallel([&](size_t lo, size_t hi) {
        for (size_t j = lo; j < hi; ++j) std::memcpy(made[j]->data, c.host_stage + j * words, words * 8);
    });
    for (size_t j = 0; j < batch; ++j) out[j] = made[j];
}

// parsed view of a commitment that belongs to this context, or false
static bool parse_commitment(const LweContext& c, const LweCommitment* cm, const uint64_t** body) {
    if (!cm || !cm->data || cm->len < 1) return false;
    const uint64_t byte_len = cm->data[0];
    if (byte_len == 0 || byte_len > (cm->len - 1) * 8) return false;       // commitment.cpp:71-75
    const size_t words = kHeaderWords + (size_t)(c.k + 1) * c.n;
    if (byte_len != 8ull * (words - 1)) return false;
    const uint64_t* d = cm->data;
    if (d[1] != kWireMagic || d[2] != ((uint64_t)c.n | ((uint64_t)c.k << 32)) || d[3] != c.q || d[4] != c.t) return false;
    *body = d + kHeaderWords;
    return true;
}

template <int K>
static void launch_verify_tile(const LweContext& c, const VerifyTileJob& job, hipStream_t s) {
    hipLaunchKernelGGL((verify_tile_kernel<K>), dim3(job.count), dim3(kF8Threads), 0, s, job, c.s_perm.ptr, c.ntt->mod, c.ntt->fwd_f64.ptr, c.ntt->inv_f64.ptr,
                       RoundConsts<ArithF64>{c.ntt->n_inv_f64, c.ntt->w_last_scaled_f64});
    LSR_HIP(hipGetLastError());
}

// `count` openings from device-resident wire rows and claimed messages (1 <= msg_len <= n), enqueued on `s`; caller holds c.mutex.
// Leaves the per-row OR of (decoded ^ claimed) in c.ws_vflags and the per-row "not a canonical commitment of this context" mark in
// c.ws_vbad — lwe_verify_opening (commitment.cpp:200-232) for a whole batch:
//   v - INTT(<s_hat, NTT(u)>) decoded slot-wise and compared with the message words as given.
static void verify_rows_device(const LweContext& c, const uint64_t* d_rows, const uint64_t* d_msgs, size_t msg_len, size_t count, hipStream_t s) {
    const uint32_t n = c.n, k = c.k;
    const size_t kn = (size_t)k * n, row = kHeaderWords + kn + n;
    // one allocation, one clear, one copy back: [flags: count x u64 | bad: count x u32]
    const size_t state_words = count + (count + 1) / 2;
    if (c.ws_vflags.count < state_words) c.ws_vflags.allocate(state_words);
    c.ws_vbad = reinterpret_cast<uint32_t*>(c.ws_vflags.ptr + count);
    zero_words_async(reinterpret_cast<uint64_t*>(c.ws_vflags.ptr), state_words, s);
    if (c.s_perm.ptr && c.logn == 12) {     // one launch, one workgroup per opening: the row is read once (lsr_commit_tile.hpp)
        const VerifyTileJob job{d_rows, d_msgs, (uint64_t)msg_len, c.ws_vflags.ptr, c.ws_vbad, (uint32_t)count, c.q, c.t};
        switch (k) {
            case 1: launch_verify_tile<1>(c, job, s); break;
            case 2: launch_verify_tile<2>(c, job, s); break;
            case 3: launch_verify_tile<3>(c, job, s); break;
            default: launch_verify_tile<4>(c, job, s); break;
        }
        return;
    }
    if (c.s_perm.ptr && c.a_perm.ptr) {     // n = 2^16 / 2^17: three launches per chunk, u read once, v read once (lsr_commit_tile.hpp)
        const size_t vec_words = (size_t)k << c.logn;
        const size_t chunk = std::max<size_t>(1, kVerifyChunkBytes / (vec_words * 8));
        const size_t slot = std::min(chunk, count);
        // chunk lanes like the commitments' (commit_rows_fused): a chunk's three launches are a dependent chain whose fills and drains
        // the other lane's kernels cover (one lane: 1.84 ms per 1024 openings at rank 4, neither memory nor FP64 half busy)
        const int streams = static_cast<int>(std::min<size_t>((size_t)kVerifyStreams, (count + chunk - 1) / chunk));
        ensure_side_streams(c, streams);
        const size_t slot_words = slot * (vec_words + n);
        if (c.ws_mid.count < slot_words * streams) c.ws_mid.allocate(slot_words * streams);
        const RoundConsts<ArithF64> cs{c.ntt->n_inv_f64, c.ntt->w_last_scaled_f64};
        const int r = c.logn - 12, lo = 12;
        fork_lanes(c, s, streams);                    // behind the clear of the verdict state above
        size_t index = 0;
        for (size_t first = 0; first < count; first += chunk, ++index) {
            const size_t now = std::min(chunk, count - first);
            hipStream_t st = index % streams == 0 ? s : c.side[index % streams - 1];
            uint64_t* const ws = c.ws_mid.ptr + (index % streams) * slot_words;
            uint64_t* const ws_out = ws + slot * vec_words;
            const VerifyTopJob job{d_rows + first * row, ws, ws_out, d_msgs + first * msg_len, (uint64_t)msg_len, (uint64_t)row, c.ws_vflags.ptr + first,
                                   c.ws_vbad + first, (uint32_t)now, k, c.q, c.t};
            const unsigned grid_f = static_cast<unsigned>((now * k << c.logn) >> (r + 8)), grid_i = static_cast<unsigned>((now << c.logn) >> (r + 8));
            if (r == 4) hipLaunchKernelGGL((verify_top_forward_kernel<4>), dim3(grid_f), dim3(256), 0, st, job, lo, c.ntt->mod, c.ntt->fwd_f64.ptr);
            else hipLaunchKernelGGL((verify_top_forward_kernel<5>), dim3(grid_f), dim3(256), 0, st, job, lo, c.ntt->mod, c.ntt->fwd_f64.ptr);
            switch (k) {
                case 1: launch_mid_general<1, 1>(c, ws, vec_words, ws_out, n, c.s_perm.ptr, now, st); break;
                case 2: launch_mid_general<2, 1>(c, ws, vec_words, ws_out, n, c.s_perm.ptr, now, st); break;
                case 3: launch_mid_general<3, 1>(c, ws, vec_words, ws_out, n, c.s_perm.ptr, now, st); break;
                default: launch_mid_general<4, 1>(c, ws, vec_words, ws_out, n, c.s_perm.ptr, now, st); break;
            }
            if (r == 4) hipLaunchKernelGGL((verify_top_inverse_kernel<4>), dim3(grid_i), dim3(256), 0, st, job, lo, c.ntt->mod, c.ntt->inv_f64.ptr, cs);
            else hipLaunchKernelGGL((verify_top_inverse_kernel<5>), dim3(grid_i), dim3(256), 0, st, job, lo, c.ntt->mod, c.ntt->inv_f64.ptr, cs);
            LSR_HIP(hipGetLastError());
        }
        join_lanes(c, s, streams);
        return;
    }
    // general form: split the rows (header and canonicity checks on the way), transform, product, subtract, inverse, decode
    ensure_workspace(c, count);
    hipLaunchKernelGGL(unpack_commitments_kernel, dim3(grid_for(count * row)), dim3(256), 0, s, d_rows, c.ws_u.ptr, c.ws_v.ptr, c.ws_vbad, (uint64_t)kn,
                       (uint64_t)n, (uint64_t)count, c.q, c.t, (uint64_t)n | ((uint64_t)k << 32));
    launch_ntt(*c.ntt, c.ws_u.ptr, count * k, false, s);
    launch_ntt(*c.ntt, c.ws_v.ptr, count, false, s);
    matvec(c, c.ws_e2.ptr, c.s_hat.ptr, c.ws_u.ptr, nullptr, 1, k, 0, 1, count, s);            // <s_hat, u_hat>
    hipLaunchKernelGGL(rsub_mod_kernel, dim3(grid_for(count * n)), dim3(256), 0, s, c.ws_e2.ptr, c.ws_v.ptr, (uint64_t)count * n, c.q);
    launch_ntt(*c.ntt, c.ws_e2.ptr, count, true, s);
    const uint64_t lanes = (uint64_t)count * msg_len;
    hipLaunchKernelGGL(decode_compare_batch_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s, c.ws_e2.ptr, d_msgs, (uint64_t)msg_len,
                       (uint32_t)c.logn, (uint64_t)count, c.t, c.ntt->mod, c.ws_vflags.ptr);
    LSR_HIP(hipGetLastError());
}

// openings per device pass of the host-pointer entry points: about 1 GiB of rows and scratch
static size_t verify_chunk(const LweContext& c, size_t count) {
    const size_t per_opening = (4 * (size_t)c.k + 5) * c.n * 8;
    return std::max<size_t>(1, std::min<size_t>(count, (1ull << 30) / per_opening));
}

// rows (host, back to back) and messages -> results, `chunk` openings per pass; rows may live in pageable or pinned memory
static void verify_host_rows(const LweContext& c, const uint64_t* rows, const uint64_t* messages, size_t msg_len, size_t count, int* results, hipStream_t s) {
    const size_t row = kHeaderWords + ((size_t)c.k + 1) * c.n;
    const size_t chunk = verify_chunk(c, count);
    if (c.ws_rows.count < chunk * row) c.ws_rows.allocate(chunk * row);
    ensure_input_space(c, chunk);                            // ws_dm: chunk x n message slots (msg_len <= n here), no allocation per call
    uint64_t* const d_msgs = c.ws_dm.ptr;
    std::vector<unsigned long long> state(chunk + (chunk + 1) / 2);
    for (size_t first = 0; first < count; first += chunk) {
        const size_t now = std::min(chunk, count - first);
        LSR_HIP(hipMemcpyAsync(c.ws_rows.ptr, rows + first * row, now * row * 8, hipMemcpyHostToDevice, s));
        LSR_HIP(hipMemcpyAsync(d_msgs, messages + first * msg_len, now * msg_len * 8, hipMemcpyHostToDevice, s));
        verify_rows_device(c, c.ws_rows.ptr, d_msgs, msg_len, now, s);
        LSR_HIP(hipMemcpyAsync(state.data(), c.ws_vflags.ptr, (now + (now + 1) / 2) * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        LSR_HIP(hipStreamSynchronize(s));
        const uint32_t* const host_bad = reinterpret_cast<const uint32_t*>(state.data() + now);
        for (size_t j = 0; j < now; ++j) results[first + j] = host_bad[j] ? -1 : (state[j] == 0 ? 1 : 0);
    }
}

static int verify_opening(const LweContext& c, const LweCommitment* cm, const uint64_t* message, size_t msg_len) {
    const uint64_t* body = nullptr;
    if (!parse_commitment(c, cm, &body)) return -1;
    const size_t body_words = ((size_t)c.k + 1) * c.n;
    if (msg_len == 0 || msg_len > c.n) {                                   // decided without the device
        for (size_t i = 0; i < body_words; ++i)
            if (body[i] >= c.q) return -1;                                  // not a canonical payload
        return msg_len == 0 ? 1 : 0;                                        // commitment.cpp:219-221
    }
    DeviceGuard guard(c.device);
    std::lock_guard<std::mutex> lock(c.mutex);
    wait_for_async(c);
    // a single call is launch-bound: the row goes up in one copy as it is (cm->data IS the wire row) and the canonicity screening
    // happens on the device with everything else
    int result = -1;
    verify_host_rows(c, cm->data, message, msg_len, 1, &result, work_stream(*c.ntt));
    return result;
}

// screening shared by the batched forms when the message length alone decides (0 or > n): -1 / 0 / 1 on the host
static int screen_opening(const LweContext& c, const uint64_t* body, size_t msg_len) {
    const size_t body_words = ((size_t)c.k + 1) * c.n;
    for (size_t x = 0; x < body_words; ++x)
        if (body[x] >= c.q) return -1;
    return msg_len == 0 ? 1 : 0;
}

// Many openings in one device pass.  results[i]: 1 / 0 / -1 exactly as the single call.  The
// commitments' words (cm->data is the wire row) are gathered in pinned memory a chunk at a time and go up in one copy.
static void verify_opening_batch(const LweContext& c, const LweCommitment* const* cms, const uint64_t* messages, size_t msg_len, size_t count,
                                 int* results) {
    const size_t row = kHeaderWords + ((size_t)c.k + 1) * c.n;
    std::vector<size_t> live;           // indices that reach the device
    for (size_t i = 0; i < count; ++i) {
        const uint64_t* body = nullptr;
        if (!cms[i] || !parse_commitment(c, cms[i], &body)) { results[i] = -1; continue; }
        if (msg_len == 0 || msg_len > c.n) { results[i] = screen_opening(c, body, msg_len); continue; }
        live.push_back(i);
    }
    if (live.empty()) return;
    DeviceGuard guard(c.device);
    std::lock_guard<std::mutex> lock(c.mutex);
    wait_for_async(c);
    hipStream_t s = work_stream(*c.ntt);
    const size_t chunk = verify_chunk(c, live.size());
    ensure_host_stage(c, chunk * (row + msg_len));
    uint64_t* const h_rows = c.host_stage;
    uint64_t* const h_msgs = c.host_stage + chunk * row;
    std::vector<int> part(chunk);
    for (size_t first = 0; first < live.size(); first += chunk) {
        const size_t now = std::min(chunk, live.size() - first);
        // the gather is a plain host copy of `now` rows (98 KB each at the reference's parameters): a few threads, one per ~4 MiB
        auto gather = [&](size_t lo, size_t hi) {
            for (size_t j = lo; j < hi; ++j) {
                std::memcpy(h_rows + j * row, cms[live[first + j]]->data, row * 8);
                std::memcpy(h_msgs + j * msg_len, messages + live[first + j] * msg_len, msg_len * 8);
            }
        };
        const size_t workers = std::min<size_t>(8, std::max<size_t>(1, (now * row * 8) >> 22));
        if (workers <= 1) {
            gather(0, now);
        } else {
            std::vector<std::thread> pool;
            const size_t per = (now + workers - 1) / workers;
            for (size_t w = 0; w < workers; ++w) {
                const size_t lo = w * per, hi = std::min(now, lo + per);
                if (lo < hi) pool.emplace_back(gather, lo, hi);
            }
            for (std::thread& th : pool) th.join();
        }
        verify_host_rows(c, h_rows, h_msgs, msg_len, now, part.data(), s);
        for (size_t j = 0; j < now; ++j) results[live[first + j]] = part[j];
    }
}

// the same for commitments stored back to back (rows of lsr_lwe_commit_batch_flat): the rows go up as they are, the header
// and canonicity checks run on the device
static void verify_opening_batch_flat(const LweContext& c, const uint64_t* words, const uint64_t* messages, size_t msg_len, size_t count, int* results) {
    const size_t row = kHeaderWords + ((size_t)c.k + 1) * c.n;
    if (msg_len == 0 || msg_len > c.n) {   // decided by the screening alone: host path
        for (size_t i = 0; i < count; ++i) {
            const LweCommitment view{const_cast<uint64_t*>(words + i * row), row};
            const uint64_t* body = nullptr;
            results[i] = parse_commitment(c, &view, &body) ? screen_opening(c, body, msg_len) : -1;
        }
        return;
    }
    DeviceGuard guard(c.device);
    std::lock_guard<std::mutex> lock(c.mutex);
    wait_for_async(c);
    verify_host_rows(c, words, messages, msg_len, count, results, work_stream(*c.ntt));
}

static LweCommitment* linear_combine(const LweContext& c, const LweCommitment** cms, const uint64_t* coeffs, size_t count) {
    const size_t body_words = (size_t)(c.k + 1) * c.n;
    DeviceGuard guard(c.device);
    std::lock_guard<std::mutex> lock(c.mutex);
    wait_for_async(c);
    hipStream_t s = work_stream(*c.ntt);
    // bodies are gathered `group` at a time (<= 64 MiB) in pinned memory, uploaded in one copy and folded in by one kernel
    const size_t group = std::max<size_t>(1, std::min<size_t>(count, (size_t(64) << 20) / (body_words * 8)));
    DeviceBuffer<uint64_t> acc(body_words), terms(group * body_words), d_coeffs(group);
    ensure_host_stage(c, group * (body_words + 1));
    uint64_t* const h_terms = c.host_stage;
    uint64_t* const h_coeffs = c.host_stage + group * body_words;
    LSR_HIP(hipMemsetAsync(acc.ptr, 0, body_words * 8, s));
    // noise budget of the result: sum_i |c_i| (centred mod t) fresh-commitment noises must still decode (8-sigma tail below Delta / 2).
    // The reference's 72-bit SEAL modulus absorbs any c_i < t (commitment.cpp:88-96,247-266); a 44-bit modulus does not, and a
    // commitment that cannot open is refused here rather than returned.  A 60-bit NTT prime as params->modulus gives the
    // reference's range.
    // Coefficients act through their CENTRED representative mod t (round-2 advisor): c in (t/2, t) is the small negative number
    // c - t, multiplied in as the residue q - (t - c), so subtracting a commitment (coefficient t - 1) costs one unit of noise.
    auto centred = [&](uint64_t coeff, uint64_t* residue) {
        const uint64_t cf = coeff % c.t;
        const bool negative = cf > c.t / 2;
        *residue = negative ? c.q - (c.t - cf) : cf;
        return static_cast<double>(negative ? c.t - cf : cf);
    };
    double weight = 0;
    for (size_t i = 0; i < count; ++i)
        if (cms[i]) { uint64_t unused; weight += centred(coeffs[i], &unused); }
    // + 1: the rounding of the scaled messages (at most 1/2 per commitment, times its coefficient)
    if (weight * (c.noise_unit + 1.0) >= 0.5 * static_cast<double>(c.delta)) {
        set_last_error("lwe_linear_combine: coefficients exceed the noise budget of this context's modulus (sum of |c_i|, c_i centred mod t, too large); "
                       "create the context with a wider NTT prime as modulus");
        std::fprintf(stderr, "lwe_linear_combine error: sum of coefficients %.0f exceeds the noise budget %.0f of the %d-bit modulus\n", weight,
                     0.5 * static_cast<double>(c.delta) / (c.noise_unit + 1.0), 64 - __builtin_clzll(c.q));
        return nullptr;
    }
    bool any = false;
    size_t staged = 0;
    auto flush = [&] {
        if (!staged) return;
        LSR_HIP(hipMemcpyAsync(terms.ptr, h_terms, staged * body_words * 8, hipMemcpyHostToDevice, s));
        LSR_HIP(hipMemcpyAsync(d_coeffs.ptr, h_coeffs, staged * 8, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(combine_kernel, dim3(grid_for(body_words)), dim3(256), 0, s, acc.ptr, terms.ptr, d_coeffs.ptr, (uint32_t)staged,
                           (uint64_t)body_words, c.ntt->mod);
        LSR_HIP(hipGetLastError());
        LSR_HIP(hipStreamSynchronize(s));   // the staging is reused
        staged = 0;
    };
    for (size_t i = 0; i < count; ++i) {
        if (!cms[i]) continue;                                             // commitment.cpp:248-250
        const uint64_t* body = nullptr;
        if (!parse_commitment(c, cms[i], &body)) {
            (void)hipStreamSynchronize(s);
            return nullptr;                                                // commitment.cpp:253-255
        }
        std::memcpy(h_terms + staged * body_words, body, body_words * 8);
        (void)centred(coeffs[i], &h_coeffs[staged]);
        any = true;
        if (++staged == group) flush();
    }
    flush();
    if (!any) return nullptr;                                              // commitment.cpp:268-270
    const size_t words = kHeaderWords + body_words;
    LweCommitment* out = new_commitment(words);
    out->data[0] = 8ull * (words - 1);
    out->data[1] = kWireMagic;
    out->data[2] = (uint64_t)c.n | ((uint64_t)c.k << 32);
    out->data[3] = c.q;
    out->data[4] = c.t;
    try {
        LSR_HIP(hipMemcpyAsync(out->data + kHeaderWords, acc.ptr, body_words * 8, hipMemcpyDeviceToHost, s));
        LSR_HIP(hipStreamSynchronize(s));
    } catch (...) {
        delete[] out->data;
        delete out;
        throw;
    }
    return out;
}

}  // namespace lsr

// one host thread per shard; body(g, first, count) runs with the shard's device current
template <class F>
static int run_shards(const char* where, int shards, size_t batch, LweContext* const* ctxs, F&& body) noexcept {
    std::vector<int> rc(shards, 0);
    std::vector<std::string> errors(shards);
    std::vector<std::thread> pool;
    for (int g = 0; g < shards; ++g)
        pool.emplace_back([&, g] {
            size_t first = 0, count = 0;
            lsr_shard_bounds(batch, shards, g, &first, &count);
            if (count == 0) return;
            try {
                lsr::DeviceGuard guard(ctxs[g]->device);
                body(g, first, count);
            } catch (const std::exception& e) {
                rc[g] = -1;
                errors[g] = e.what();
            } catch (...) {
                rc[g] = -1;
            }
        });
    for (std::thread& th : pool) th.join();
    for (int g = 0; g < shards; ++g)
        if (rc[g] != 0) {
            lsr::set_last_error(std::string(where) + ": shard " + std::to_string(g) + ": " + errors[g]);
            std::fprintf(stderr, "lambda_snark_core: %s failed on shard %d: %s\n", where, g, errors[g].c_str());
            return -1;
        }
    return 0;
}

static bool shards_compatible(LweContext* const* ctxs, int shards) {
    if (!ctxs || shards <= 0) return false;
    for (int g = 0; g < shards; ++g) {
        if (!ctxs[g]) return false;
        if (ctxs[g]->q != ctxs[0]->q || ctxs[g]->n != ctxs[0]->n || ctxs[g]->k != ctxs[0]->k || ctxs[g]->t != ctxs[0]->t) return false;
        if (std::memcmp(ctxs[g]->keys.id, ctxs[0]->keys.id, sizeof ctxs[0]->keys.id) != 0) return false;   // replicas of ONE context
        for (int h = 0; h < g; ++h)
            if (ctxs[h] == ctxs[g]) return false;    // a context serialises its callers: one per shard
    }
    return true;
}

// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

LweContext* lwe_context_create(const PublicParams* params) noexcept {
    try {
        // fresh 256-bit keys per context, like the reference (commitment.cpp:118-121); reproducible keys only through the
        // explicit entry point lsr_lwe_context_create_seeded
        return lsr::create_lwe_context(params, 0, -1);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "lwe_context_create error: %s\n", e.what());
        return nullptr;
    } catch (...) {
        return nullptr;
    }
}

LweContext* lsr_lwe_context_create_seeded(const PublicParams* params, uint64_t key_seed, int device) noexcept {
    try {
        return lsr::create_lwe_context(params, key_seed, device);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "lwe_context_create error: %s\n", e.what());
        return nullptr;
    } catch (...) {
        return nullptr;
    }
}

void lwe_context_free(LweContext* ctx) noexcept { lsr::destroy_lwe_context(ctx); }

uint64_t lsr_lwe_wide_modulus(uint32_t ring_degree) noexcept {
    if (ring_degree < 2 || (ring_degree & (ring_degree - 1)) != 0 || ring_degree > 131072) return 0;
    return lsr::largest_prime_congruent_one(2ull * ring_degree, 60);
}
uint64_t lsr_lwe_modulus(const LweContext* ctx) noexcept { return ctx ? ctx->q : 0; }
uint64_t lsr_lwe_plain_modulus(const LweContext* ctx) noexcept { return ctx ? ctx->t : 0; }
uint32_t lsr_lwe_ring_degree(const LweContext* ctx) noexcept { return ctx ? ctx->n : 0; }
uint32_t lsr_lwe_module_rank(const LweContext* ctx) noexcept { return ctx ? ctx->k : 0; }
size_t lsr_lwe_commitment_words(const LweContext* ctx) noexcept { return ctx ? lsr::kHeaderWords + (size_t)(ctx->k + 1) * ctx->n : 0; }
const NttContext* lsr_lwe_ntt_context(const LweContext* ctx) noexcept { return ctx ? ctx->ntt : nullptr; }

int lsr_lwe_public_matrix(const LweContext* ctx, uint64_t* a_hat) noexcept {
    if (!ctx || !a_hat) return -1;
    try {
        lsr::DeviceGuard guard(ctx->device);
        LSR_HIP(hipMemcpy(a_hat, ctx->a_hat.ptr, ctx->a_hat.count * 8, hipMemcpyDeviceToHost));
        return 0;
    } catch (const std::exception& e) {
        lsr::set_last_error(e.what());
        return -1;
    }
}

int lwe_commit_batch(LweContext* ctx, const uint64_t* messages, size_t msg_len, size_t batch, const uint64_t* seeds, LweCommitment** out) noexcept {
    if (!ctx || !messages || !out) return -1;
    if (batch == 0) return 0;
    try {
        lsr::DeviceGuard guard(ctx->device);
        std::lock_guard<std::mutex> lock(ctx->mutex);
        lsr::wait_for_async(*ctx);
        const size_t per_commit = (3 * (size_t)ctx->k + 3) * ctx->n * 8;
        const size_t chunk = std::max<size_t>(1, std::min<size_t>(batch, (1ull << 30) / per_commit));
        size_t done = 0;
        try {
            for (; done < batch; done += std::min(chunk, batch - done))
                lsr::commit_chunk(*ctx, messages + done * msg_len, msg_len, std::min(chunk, batch - done), seeds ? seeds + done : nullptr, out + done);
        } catch (...) {
            for (size_t j = 0; j < done; ++j) { lwe_commitment_free(out[j]); out[j] = nullptr; }
            throw;
        }
        return 0;
    } catch (const std::exception& e) {
        lsr::set_last_error(std::string("lwe_commit_batch: ") + e.what());
        std::fprintf(stderr, "lwe_commit error: %s\n", e.what());                 // commitment.cpp:158-160
        return -1;
    } catch (...) {
        std::fprintf(stderr, "lwe_commit error: unknown exception\n");           // commitment.cpp:161-163
        return -1;
    }
}

static int commit_batch_flat(const char* where, LweContext* ctx, const uint64_t* messages, size_t msg_len, size_t batch, const uint64_t* seeds,
                             uint64_t* out_words, bool to_device) noexcept {
    if (!ctx || !messages || !out_words) return -1;
    if (batch == 0) return 0;
    try {
        lsr::DeviceGuard guard(ctx->device);
        std::lock_guard<std::mutex> lock(ctx->mutex);
        lsr::wait_for_async(*ctx);
        const size_t words = lsr::kHeaderWords + ((size_t)ctx->k + 1) * ctx->n;
        const size_t per_commit = (4 * (size_t)ctx->k + 5) * ctx->n * 8;
        const size_t chunk = std::max<size_t>(1, std::min<size_t>(batch, (1ull << 30) / per_commit));
        if (!to_device) {
            // host array: pieces of about 32 MiB of rows, so that the copy of one piece (57 GB/s over PCIe: the long pole) runs under
            // the staging and the kernels of the next
            // (page-locked destinations only: a copy into pageable memory is staged synchronously by the runtime and overlaps nothing,
            // so there the rows travel in one large copy as before)
            hipPointerAttribute_t attr{};
            const bool pinned = hipPointerGetAttributes(&attr, out_words) == hipSuccess && attr.type == hipMemoryTypeHost;
            (void)hipGetLastError();
            const size_t piece = pinned ? std::max<size_t>(1, std::min<size_t>(chunk, (size_t(32) << 20) / (words * 8))) : chunk;
            lsr::commit_batch_flat_host(*ctx, messages, msg_len, batch, seeds, out_words, piece);
            return 0;
        }
        for (size_t done = 0; done < batch; done += chunk) {
            const size_t now = std::min(chunk, batch - done);
            lsr::commit_chunk_flat(*ctx, messages + done * msg_len, msg_len, now, seeds ? seeds + done : nullptr, out_words + done * words, true);
        }
        return 0;
    } catch (const std::exception& e) {
        lsr::set_last_error(std::string(where) + ": " + e.what());
        std::fprintf(stderr, "lwe_commit error: %s\n", e.what());
        return -1;
    } catch (...) {
        return -1;
    }
}

int lsr_lwe_commit_batch_flat(LweContext* ctx, const uint64_t* messages, size_t msg_len, size_t batch, const uint64_t* seeds,
                              uint64_t* out_words) noexcept {
    return commit_batch_flat("lsr_lwe_commit_batch_flat", ctx, messages, msg_len, batch, seeds, out_words, false);
}
int lsr_lwe_commit_batch_flat_device(LweContext* ctx, const uint64_t* messages, size_t msg_len, size_t batch, const uint64_t* seeds,
                                     uint64_t* d_out_words) noexcept {
    return commit_batch_flat("lsr_lwe_commit_batch_flat_device", ctx, messages, msg_len, batch, seeds, d_out_words, true);
}

int lsr_lwe_commit_keys(const LweContext* ctx, const uint64_t* messages, size_t msg_len, size_t batch, const uint64_t* seeds, uint64_t* out_keys) noexcept {
    if (!ctx || !out_keys || (!messages && msg_len)) return -1;
    try {
        lsr::derive_commit_keys(*ctx, messages, msg_len, batch, seeds, out_keys);
        return 0;
    } catch (const std::exception& e) {
        lsr::set_last_error(std::string("lsr_lwe_commit_keys: ") + e.what());
        return -1;
    } catch (...) {
        return -1;
    }
}

// the same keys for device-resident messages, derived on the device (lsr_commit_keys.hpp); asynchronous on `stream`
int lsr_lwe_commit_keys_device(LweContext* ctx, const uint64_t* d_messages, size_t msg_len, size_t batch, const uint64_t* seeds, uint64_t* d_keys,
                               void* stream) noexcept {
    if (!ctx || !seeds || !d_keys || (!d_messages && msg_len)) return -1;
    if (batch == 0) return 0;
    for (size_t j = 0; j < batch; ++j)
        if (seeds[j] == 0) {
            lsr::set_last_error("lsr_lwe_commit_keys_device: seed 0 asks for fresh OS entropy (commitment.h:52), which lives on the host — use lsr_lwe_commit_keys");
            return -1;
        }
    if (batch > 0x7fffffffull) {
        lsr::set_last_error("lsr_lwe_commit_keys_device: batch exceeds one launch (2^31 - 1 commitments)");
        return -1;
    }
    try {
        lsr::DeviceGuard guard(ctx->device);
        std::lock_guard<std::mutex> lock(ctx->mutex);
        hipStream_t s = static_cast<hipStream_t>(stream);
        if (lsr::stream_is_capturing(s)) {     // the seeds travel through a host block that a replay would read again, with other seeds in it
            lsr::set_last_error("lsr_lwe_commit_keys_device: not capturable into a HIP graph (host seeds)");
            return -1;
        }
        lsr::begin_async(*ctx, s);
        if (ctx->ev_seeds) LSR_HIP(hipEventSynchronize(ctx->ev_seeds));       // the previous call's upload has left the page-locked block
        if (ctx->host_seeds_words < batch) {
            if (ctx->host_seeds) {
                std::memset(ctx->host_seeds, 0, ctx->host_seeds_words * 8);
                LSR_HIP(hipHostFree(ctx->host_seeds));
                ctx->host_seeds = nullptr;
                ctx->host_seeds_words = 0;
            }
            const size_t words = std::max<size_t>(batch, 4096);
            LSR_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->host_seeds), words * 8, hipHostMallocPortable));
            ctx->host_seeds_words = words;
        }
        if (ctx->ws_seeds.count < batch) ctx->ws_seeds.allocate(std::max<size_t>(batch, 4096));
        std::memcpy(ctx->host_seeds, seeds, batch * 8);
        LSR_HIP(hipMemcpyAsync(ctx->ws_seeds.ptr, ctx->host_seeds, batch * 8, hipMemcpyHostToDevice, s));
        if (!ctx->ev_seeds) LSR_HIP(hipEventCreateWithFlags(&ctx->ev_seeds, hipEventDisableTiming));
        LSR_HIP(hipEventRecord(ctx->ev_seeds, s));
        lsr::launch_commit_keys(*ctx, d_messages, msg_len, batch, ctx->ws_seeds.ptr, d_keys, s);
        lsr::end_async(*ctx, s);
        return 0;
    } catch (const std::exception& e) {
        lsr::set_last_error(std::string("lsr_lwe_commit_keys_device: ") + e.what());
        return -1;
    } catch (...) {
        return -1;
    }
}

int lsr_lwe_commit_rows_device(LweContext* ctx, const uint64_t* d_messages, size_t msg_len, size_t batch, const uint64_t* d_keys, uint64_t* d_rows,
                               void* stream) noexcept {
    if (!ctx || !d_keys || !d_rows || (!d_messages && msg_len)) return -1;
    if (batch == 0) return 0;
    try {
        lsr::DeviceGuard guard(ctx->device);
        std::lock_guard<std::mutex> lock(ctx->mutex);
        hipStream_t s = static_cast<hipStream_t>(stream);
        lsr::begin_async(*ctx, s);
        lsr::commit_rows_device(*ctx, d_messages ? d_messages : d_keys, msg_len, batch, d_keys, d_rows, s);
        lsr::end_async(*ctx, s);
        return 0;
    } catch (const std::exception& e) {
        lsr::set_last_error(std::string("lsr_lwe_commit_rows_device: ") + e.what());
        std::fprintf(stderr, "lwe_commit error: %s\n", e.what());
        return -1;
    } catch (...) {
        return -1;
    }
}

int lsr_lwe_verify_rows_device(const LweContext* ctx, const uint64_t* d_rows, const uint64_t* d_messages, size_t msg_len, size_t count, int* d_results,
                               void* stream) noexcept {
    if (!ctx || !d_rows || !d_messages || !d_results || msg_len == 0 || msg_len > ctx->n) return -1;
    if (count == 0) return 0;
    try {
        lsr::DeviceGuard guard(ctx->device);
        std::lock_guard<std::mutex> lock(ctx->mutex);
        hipStream_t s = static_cast<hipStream_t>(stream);
        lsr::begin_async(*ctx, s);
        lsr::verify_rows_device(*ctx, d_rows, d_messages, msg_len, count, s);
        hipLaunchKernelGGL(lsr::opening_verdict_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, ctx->ws_vflags.ptr, ctx->ws_vbad, d_results,
                           (uint64_t)count);
        LSR_HIP(hipGetLastError());
        lsr::end_async(*ctx, s);
        return 0;
    } catch (const std::exception& e) {
        lsr::set_last_error(std::string("lsr_lwe_verify_rows_device: ") + e.what());
        std::fprintf(stderr, "lwe_verify_opening error: %s\n", e.what());
        return -1;
    } catch (...) {
        return -1;
    }
}

const char* lsr_lwe_pipeline(const LweContext* ctx) noexcept {
    if (!ctx) return "";
    if (ctx->ab_perm.ptr && ctx->logn == 12) return "tile";
    if (ctx->a_perm.ptr) return (ctx->ab_perm.ptr || ctx->b_perm.ptr) ? "fused" : "fused-matvec";
    return "general";
}

LweCommitment* lwe_commit(LweContext* ctx, const uint64_t* message, size_t msg_len, uint64_t seed) noexcept {
    if (!ctx || !message) return nullptr;                                          // commitment.cpp:144
    LweCommitment* out = nullptr;
    if (lwe_commit_batch(ctx, message, msg_len, 1, &seed, &out) != 0) return nullptr;
    return out;
}

void lwe_commitment_free(LweCommitment* comm) noexcept {
    if (!comm) return;
    if (comm->data) {
        volatile uint64_t* p = comm->data;                                         // zeroize (commitment.cpp:169-173)
        for (size_t i = 0; i < comm->len; ++i) p[i] = 0;
        delete[] comm->data;
    }
    delete comm;
}

LweCommitment* lwe_commitment_clone(const LweCommitment* comm) noexcept {
    if (!comm || comm->len == 0 || !comm->data) return nullptr;                    // commitment.cpp:180-182
    auto* clone = new (std::nothrow) LweCommitment;
    if (!clone) return nullptr;
    clone->len = comm->len;
    clone->data = new (std::nothrow) uint64_t[clone->len];
    if (!clone->data) {
        delete clone;
        return nullptr;
    }
    std::memcpy(clone->data, comm->data, clone->len * sizeof(uint64_t));
    return clone;
}

int lwe_verify_opening(const LweContext* ctx, const LweCommitment* commitment, const uint64_t* message, size_t msg_len,
                       const LweOpening* /*opening: ignored, commitment.cpp:205*/) noexcept {
    if (!ctx || !commitment || !message) return -1;                                // commitment.cpp:207
    try {
        return lsr::verify_opening(*ctx, commitment, message, msg_len);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "lwe_verify_opening error: %s\n", e.what());         // commitment.cpp:229-231
        return -1;
    } catch (...) {
        return -1;
    }
}

int lwe_verify_opening_batch(const LweContext* ctx, const LweCommitment* const* commitments, const uint64_t* messages, size_t msg_len, size_t count,
                             int* results) noexcept {
    if (!ctx || !commitments || !messages || !results) return -1;
    if (count == 0) return 0;
    try {
        lsr::verify_opening_batch(*ctx, commitments, messages, msg_len, count, results);
        return 0;
    } catch (const std::exception& e) {
        lsr::set_last_error(std::string("lwe_verify_opening_batch: ") + e.what());
        std::fprintf(stderr, "lwe_verify_opening error: %s\n", e.what());
        return -1;
    } catch (...) {
        return -1;
    }
}

int lsr_lwe_verify_opening_batch_flat(const LweContext* ctx, const uint64_t* words, const uint64_t* messages, size_t msg_len, size_t count,
                                      int* results) noexcept {
    if (!ctx || !words || !messages || !results) return -1;
    if (count == 0) return 0;
    try {
        lsr::verify_opening_batch_flat(*ctx, words, messages, msg_len, count, results);
        return 0;
    } catch (const std::exception& e) {
        lsr::set_last_error(std::string("lsr_lwe_verify_opening_batch_flat: ") + e.what());
        std::fprintf(stderr, "lwe_verify_opening error: %s\n", e.what());
        return -1;
    } catch (...) {
        return -1;
    }
}

LweCommitment* lwe_linear_combine(const LweContext* ctx, const LweCommitment** commitments, const uint64_t* coeffs, size_t count) noexcept {
    if (!ctx || !commitments || !coeffs || count == 0) return nullptr;             // commitment.cpp:240-242
    try {
        return lsr::linear_combine(*ctx, commitments, coeffs, count);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "lwe_linear_combine error: %s\n", e.what());         // commitment.cpp:273-275
        return nullptr;
    } catch (...) {
        return nullptr;
    }
}

size_t lsr_words_to_limbs(const uint64_t* words, size_t count, unsigned limb_bits, unsigned limbs_per_word, uint64_t* limbs) noexcept {
    if (limb_bits == 0 || limb_bits > 32 || limbs_per_word == 0 || limbs_per_word > 64) return 0;
    if (!words || !limbs) return count * limbs_per_word;
    const uint64_t mask = (1ull << limb_bits) - 1;
    for (size_t i = 0; i < count; ++i)
        for (unsigned l = 0; l < limbs_per_word; ++l) {
            const unsigned shift = l * limb_bits;
            limbs[i * limbs_per_word + l] = shift < 64 ? (words[i] >> shift) & mask : 0;
        }
    return count * limbs_per_word;
}

LweContext* lsr_lwe_context_replicate(const LweContext* ctx, int device) noexcept {
    if (!ctx) return nullptr;
    try {
        return lsr::create_lwe_context(&ctx->params, ctx->key_seed, device, &ctx->keys);   // same keys => same A_hat, s, b_hat
    } catch (const std::exception& e) {
        std::fprintf(stderr, "lwe_context_create error: %s\n", e.what());
        return nullptr;
    } catch (...) {
        return nullptr;
    }
}

int lsr_lwe_commit_batch_flat_sharded(LweContext* const* ctxs, int shards, const uint64_t* messages, size_t msg_len, size_t batch, const uint64_t* seeds,
                                      uint64_t* out_words) noexcept {
    if (!shards_compatible(ctxs, shards) || !messages || !out_words) return -1;
    if (batch == 0) return 0;
    const size_t words = lsr::kHeaderWords + ((size_t)ctxs[0]->k + 1) * ctxs[0]->n;
    return run_shards("lsr_lwe_commit_batch_flat_sharded", shards, batch, ctxs, [&](int g, size_t first, size_t count) {
        if (lsr_lwe_commit_batch_flat(ctxs[g], messages + first * msg_len, msg_len, count, seeds ? seeds + first : nullptr, out_words + first * words) != 0)
            throw std::runtime_error(lsr::last_error_cstr());
    });
}

static int mlwe_matvec_batch_sharded(LweContext* const* ctxs, int shards, uint64_t* const* d_r, const uint64_t* const* d_e1, size_t batch, uint64_t* host_u,
                                     double* seconds, double* per_shard) noexcept {
    if (!shards_compatible(ctxs, shards) || !d_r || !d_e1 || !host_u) return -1;
    if (batch == 0) return 0;
    const size_t vec_words = (size_t)ctxs[0]->k * ctxs[0]->n;
    std::vector<double> compute(shards, 0.0), gather(shards, 0.0);
    const int rc = run_shards("lsr_mlwe_matvec_batch_sharded", shards, batch, ctxs, [&](int g, size_t first, size_t count) {
        if (!d_r[g] || !d_e1[g]) throw std::runtime_error("NULL device array for a non-empty shard");
        const LweContext& c = *ctxs[g];
        std::lock_guard<std::mutex> lock(c.mutex);
        lsr::wait_for_async(c);
        lsr::DeviceBuffer<uint64_t> d_u(count * vec_words);
        hipStream_t s = lsr::work_stream(*c.ntt);
        lsr::ensure_copy_stream(c);
        // pieces of the slice: the device -> host copy of piece i (copy stream) runs under the kernels of piece i + 1, so a shard
        // takes about max(compute, gather) + one piece instead of their sum.  Eight pieces, at least 32 vectors each (one chunk of the
        // mixed-launch schedule at rank 4)
        const size_t piece = std::max<size_t>(std::min<size_t>(count, 32), (count + 7) / 8);
        std::vector<hipEvent_t> done_ev;
        hipEvent_t first_copy = nullptr, last_kernel = nullptr;
        LSR_HIP(hipEventCreate(&first_copy));
        LSR_HIP(hipEventCreate(&last_kernel));
        const auto t0 = std::chrono::steady_clock::now();
        try {
            for (size_t lo = 0; lo < count; lo += piece) {
                const size_t now = std::min(piece, count - lo);
                lsr::mlwe_matvec_device(c, d_r[g] + lo * vec_words, d_e1[g] + lo * vec_words, d_u.ptr + lo * vec_words, now, s, true);
                hipEvent_t ev = nullptr;
                LSR_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                done_ev.push_back(ev);
                LSR_HIP(hipEventRecord(ev, s));
                LSR_HIP(hipStreamWaitEvent(c.copy_stream, ev, 0));
                if (lo == 0) LSR_HIP(hipEventRecord(first_copy, c.copy_stream));
                // this shard's slice goes straight into its place in the caller's single (ideally pinned) array
                LSR_HIP(hipMemcpyAsync(host_u + (first + lo) * vec_words, d_u.ptr + lo * vec_words, now * vec_words * 8, hipMemcpyDeviceToHost, c.copy_stream));
            }
            LSR_HIP(hipEventRecord(last_kernel, s));
            LSR_HIP(hipStreamSynchronize(s));
            const auto t1 = std::chrono::steady_clock::now();
            LSR_HIP(hipStreamSynchronize(c.copy_stream));
            const auto t2 = std::chrono::steady_clock::now();
            compute[g] = std::chrono::duration<double>(t1 - t0).count();          // kernels of the whole slice (the copies run beside them)
            gather[g] = std::chrono::duration<double>(t2 - t0).count();           // until the last byte is in host memory: the shard's wall time
        } catch (...) {
            (void)hipStreamSynchronize(s);
            (void)hipStreamSynchronize(c.copy_stream);
            for (hipEvent_t ev : done_ev) (void)hipEventDestroy(ev);
            (void)hipEventDestroy(first_copy); (void)hipEventDestroy(last_kernel);
            throw;
        }
        for (hipEvent_t ev : done_ev) (void)hipEventDestroy(ev);
        (void)hipEventDestroy(first_copy); (void)hipEventDestroy(last_kernel);
    });
    if (seconds) {
        seconds[0] = *std::max_element(compute.begin(), compute.end());
        seconds[1] = *std::max_element(gather.begin(), gather.end());
    }
    if (per_shard)
        for (int g = 0; g < shards; ++g) { per_shard[2 * g] = compute[g]; per_shard[2 * g + 1] = gather[g]; }
    return rc;
}
int lsr_mlwe_matvec_batch_sharded(LweContext* const* ctxs, int shards, uint64_t* const* d_r, const uint64_t* const* d_e1, size_t batch, uint64_t* host_u,
                                  double* seconds) noexcept {
    return mlwe_matvec_batch_sharded(ctxs, shards, d_r, d_e1, batch, host_u, seconds, nullptr);
}
int lsr_mlwe_matvec_batch_sharded_stats(LweContext* const* ctxs, int shards, uint64_t* const* d_r, const uint64_t* const* d_e1, size_t batch, uint64_t* host_u,
                                        double* per_shard) noexcept {
    return mlwe_matvec_batch_sharded(ctxs, shards, d_r, d_e1, batch, host_u, nullptr, per_shard);
}

int lsr_lwe_sample_blinding_device(const LweContext* ctx, uint64_t* d_e1, size_t batch, const uint64_t* seeds, void* stream) noexcept {
    if (!ctx || !d_e1 || !seeds) return -1;
    if (batch == 0) return 0;
    try {
        lsr::DeviceGuard guard(ctx->device);
        hipStream_t s = static_cast<hipStream_t>(stream);
        std::vector<uint64_t> keys(batch * 4);
        for (size_t j = 0; j < batch; ++j) lsr::key_words(lsr::expand_seed64(seeds[j]), keys.data() + 4 * j);
        lsr::DeviceBuffer<uint64_t> d_keys(batch * 4);
        LSR_HIP(hipMemcpyAsync(d_keys.ptr, keys.data(), batch * 32, hipMemcpyHostToDevice, s));
        lsr::launch_gaussian(lsr::GaussianJob{d_e1, d_keys.ptr, 0, ctx->k, lsr::kDomE1, ctx->n, batch * ctx->k, ctx->q}, ctx->cdf.ptr, ctx->cdf_entries, s);
        LSR_HIP(hipStreamSynchronize(s));   // the key arrays die with this scope
        return 0;
    } catch (const std::exception& e) {
        lsr::set_last_error(std::string("lsr_lwe_sample_blinding_device: ") + e.what());
        std::fprintf(stderr, "lambda_snark_core: lsr_lwe_sample_blinding_device failed: %s\n", e.what());
        return -1;
    } catch (...) {
        return -1;
    }
}

int lsr_mlwe_matvec_batch_device(const LweContext* ctx, uint64_t* d_r, const uint64_t* d_e1, uint64_t* d_u, size_t batch, const uint64_t* seeds,
                                 void* stream) noexcept {
    if (!ctx || !d_r || !d_u) return -1;
    if (!d_e1 && !seeds) return -1;
    if (batch == 0) return 0;
    try {
        lsr::DeviceGuard guard(ctx->device);
        hipStream_t s = static_cast<hipStream_t>(stream);
        std::lock_guard<std::mutex> lock(ctx->mutex);
        lsr::begin_async(*ctx, s);
        if (d_e1) {
            lsr::mlwe_matvec_device(*ctx, d_r, d_e1, d_u, batch, s, true);
            lsr::end_async(*ctx, s);
            return 0;
        }
        // e1 sampled on the device from the per-vector raw-seed streams (domain 5), then added
        ctx->ws_key_host.resize(batch * 4);
        for (size_t j = 0; j < batch; ++j) lsr::key_words(lsr::expand_seed64(seeds[j]), ctx->ws_key_host.data() + 4 * j);
        // the workspace is sized BEFORE the keys are staged: ensure_workspace re-allocates ws_keys when the batch grows (round-2
        // advisor: the upload used to precede it, so the sampler of an unfused context read a freed-and-reallocated buffer)
        const bool fused = ctx->a_perm.ptr != nullptr;
        if (!fused) lsr::ensure_workspace(*ctx, batch);
        else lsr::ensure_input_space(*ctx, batch);
        LSR_HIP(hipMemcpyAsync(ctx->ws_keys.ptr, ctx->ws_key_host.data(), batch * 32, hipMemcpyHostToDevice, s));
        if (fused) {
            lsr::mlwe_matvec_fused(*ctx, d_r, nullptr, d_u, batch, s, ctx->ws_keys.ptr);
        } else {
            lsr::launch_gaussian(lsr::GaussianJob{ctx->ws_e1.ptr, ctx->ws_keys.ptr, 0, ctx->k, lsr::kDomE1, ctx->n, batch * ctx->k, ctx->q}, ctx->cdf.ptr,
                                 ctx->cdf_entries, s);
            lsr::mlwe_matvec_device(*ctx, d_r, ctx->ws_e1.ptr, d_u, batch, s, false);
        }
        lsr::end_async(*ctx, s);
        LSR_HIP(hipStreamSynchronize(s));   // seeds is a host array the caller may reuse
        return 0;
    } catch (const std::exception& e) {
        lsr::set_last_error(std::string("lsr_mlwe_matvec_batch_device: ") + e.what());
        std::fprintf(stderr, "lambda_snark_core: lsr_mlwe_matvec_batch_device failed: %s\n", e.what());
        return -1;
    } catch (...) {
        return -1;
    }
}

}  // extern "C"
