// Does a hipMemsetAsync captured into a HIP graph run (and run in order) on every replay?  ROCm 7.2, gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void bump(uint64_t* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1 + i; }
__global__ void copy_out(const uint64_t* p, uint64_t* out, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = p[i]; }
int main() {
    for (int words : {18, 1024, 1 << 20}) {
        uint64_t *buf, *out;
        CK(hipMalloc(&buf, words * 8)); CK(hipMalloc(&out, words * 8));
        CK(hipMemset(buf, 0xFF, words * 8));
        hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        CK(hipMemsetAsync(buf, 0, words * 8, s));
        hipLaunchKernelGGL(bump, dim3((words + 255) / 256), dim3(256), 0, s, buf, words);
        hipLaunchKernelGGL(copy_out, dim3((words + 255) / 256), dim3(256), 0, s, buf, out, words);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
            uint64_t h[3];
            CK(hipMemcpy(h, out, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(h + 1, out + words / 2, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(h + 2, out + words - 1, 8, hipMemcpyDeviceToHost));
            std::printf("words %d replay %d: out[0] = %llu (want 1), out[mid] = %llu (want %d), out[last] = %llu (want %d)\n", words, rep,
                        (unsigned long long)h[0], (unsigned long long)h[1], 1 + words / 2, (unsigned long long)h[2], words);
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(s)); CK(hipFree(buf)); CK(hipFree(out));
    }
    return 0;
}
