// Microbenchmark: what does a dependent phase cost (a) as a kernel of its own inside a HIP graph, (b) as a phase of one
// persistent kernel behind a grid barrier?  Every phase is the chain the pooled-level kernels have: index load ->
// gather -> store, on data the previous phase wrote.
// build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/grid_phase tools/microbench/grid_phase.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kThreads = 1024;

__device__ __forceinline__ void phase_body(const int *__restrict__ idx, const float *__restrict__ in, float *__restrict__ out,
                                           int n, int vb, int chain) {
    int i = vb * kThreads + threadIdx.x;
    if (i >= n) return;
    int j = idx[i];
    for (int c = 1; c < chain; c++) j = idx[j];        // dependent trips
    out[i] = in[j] + 1.0f;
}

__global__ __launch_bounds__(kThreads) void k_phase(const int *idx, const float *in, float *out, int n, int chain) {
    phase_body(idx, in, out, n, blockIdx.x, chain);
}

__device__ __forceinline__ bool grid_barrier(unsigned *counter, unsigned target, int *status) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 20000000) { *status = 1; ok = false; break; }   // bail out instead of hanging the box
        }
        __threadfence();
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(kThreads) void k_persistent(const int *idx, float *a, float *b, int n, int chain, int phases,
                                                         int vblocks, unsigned *counter, int *status) {
    float *in = a, *out = b;
    for (int p = 0; p < phases; p++) {
        for (int vb = blockIdx.x; vb < vblocks; vb += gridDim.x) phase_body(idx, in, out, n, vb, chain);
        if (!grid_barrier(counter, (unsigned)(p + 1) * gridDim.x, status)) return;
        float *t = in; in = out; out = t;
    }
}

int main() {
    const int phases = 32;
    for (int wgs : {4, 16, 64, 256}) {
        for (int chain : {1, 3}) {
            const int n = wgs * kThreads;
            std::vector<int> h(n);
            for (int i = 0; i < n; i++) h[i] = (int)(((long long)i * 7919 + 13) % n);
            int *idx; float *a, *b; unsigned *counter; int *status;
            CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
            CK(hipMalloc(&counter, 4)); CK(hipMalloc(&status, 4));
            CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
            CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4)); CK(hipMemset(status, 0, 4));
            hipStream_t s; CK(hipStreamCreate(&s));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            // (a) graph of `phases` dependent kernels
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            for (int p = 0; p < phases; p++) {
                k_phase<<<wgs, kThreads, 0, s>>>(idx, (p & 1) ? b : a, (p & 1) ? a : b, n, chain);
            }
            CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            float best_g = 1e9f, best_p = 1e9f, best_e = 1e9f;
            for (int r = 0; r < 20; r++) {
                CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best_g) best_g = ms;
            }
            // (a') the same launches issued eagerly
            for (int r = 0; r < 20; r++) {
                CK(hipEventRecord(e0, s));
                for (int p = 0; p < phases; p++) k_phase<<<wgs, kThreads, 0, s>>>(idx, (p & 1) ? b : a, (p & 1) ? a : b, n, chain);
                CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best_e) best_e = ms;
            }
            // (b) one persistent kernel, grid barrier between phases
            for (int r = 0; r < 20; r++) {
                CK(hipMemsetAsync(counter, 0, 4, s));
                CK(hipEventRecord(e0, s));
                k_persistent<<<wgs, kThreads, 0, s>>>(idx, a, b, n, chain, phases, wgs, counter, status);
                CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best_p) best_p = ms;
            }
            // (b') persistent kernel on fewer workgroups than virtual blocks
            float best_h = 1e9f;
            const int half = wgs > 4 ? wgs / 4 : wgs;
            for (int r = 0; r < 20; r++) {
                CK(hipMemsetAsync(counter, 0, 4, s));
                CK(hipEventRecord(e0, s));
                k_persistent<<<half, kThreads, 0, s>>>(idx, a, b, n, chain, phases, wgs, counter, status);
                CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best_h) best_h = ms;
            }
            int st; CK(hipMemcpy(&st, status, 4, hipMemcpyDeviceToHost));
            std::vector<float> ha(n); CK(hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost));
            printf("wgs %3d chain %d: graph %.2f us/phase, eager %.2f, persistent %.2f, persistent on %d wgs %.2f  (status %d, a[0]=%g)\n",
                   wgs, chain, best_g * 1000 / phases, best_e * 1000 / phases, best_p * 1000 / phases, half,
                   best_h * 1000 / phases, st, ha[0]);
            hipFree(idx); hipFree(a); hipFree(b); hipFree(counter); hipFree(status);
        }
    }
    return 0;
}
