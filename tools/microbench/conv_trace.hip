// Where do the ~11 us of a small pooled-level k_conv_fused launch go?  Runs the library's own kernel (gemm.hip is
// included, with DAGR_TRACE recording the 100-MHz clock at its stage boundaries for workgroup 0 when built with -DTRACE)
// on synthetic levels of the sizes a B = 1 window has, inside a HIP graph as the engine's tail does.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Idagr_amd/csrc [-DTRACE] \
//            -o tools/microbench/conv_trace[_t] tools/microbench/conv_trace.hip
#include <hip/hip_runtime.h>
#ifdef TRACE
__device__ long long g_trace[16][16];     // [wave][stage] of workgroup 0
#define DAGR_TRACE(i)                                                                                  \
    do {                                                                                               \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                    \
        if (blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0) g_trace[threadIdx.x >> 6][i] = wall_clock64(); \
    } while (0)
#endif
#include "errors.hip"
#ifndef GEMM_SRC
#define GEMM_SRC "gemm.hip"
#endif
#include GEMM_SRC
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_stamp(long long *p) { *p = wall_clock64(); }

int main(int argc, char **argv) {
    const int reps = 20, nw = 20;
    hipStream_t s; CK(hipStreamCreate(&s));
    // (live nodes, node capacity, cin, cskip, in-degree): the pooled levels of a B = 1 x 25 k and a B = 8 x 100 k window
    const int cfg[][5] = {{36, 70, 66, 0, 5}, {36, 70, 64, 66, 5}, {141, 280, 66, 0, 5}, {561, 1120, 66, 0, 5},
                          {2241, 4480, 18, 0, 5}, {2241, 4480, 64, 18, 5}, {282, 315, 66, 0, 6}, {1122, 1260, 66, 0, 7},
                          {1122, 1260, 64, 66, 7}, {4482, 5040, 66, 0, 7}, {4482, 5040, 64, 66, 7}, {17922, 20160, 18, 0, 8},
                          {17922, 20160, 64, 18, 8}};
    for (const auto &c : cfg) {
        const int n = c[0], n_cap = c[1], cin = c[2], cskip = c[3], deg = c[4], N = 64, rx = 2, ry = 2;
        const int K = 26 * cin + cskip, G = (K + 15) / 16;
        const int E = n * deg;
        std::vector<int> rowptr(n_cap + 1), col(E), code(E), cnt{n, E};
        for (int i = 0; i <= n_cap; i++) rowptr[i] = (i < n ? i : n) * deg;
        for (int e = 0; e < E; e++) { col[e] = (int)(((long long)e * 7919 + 3) % n); code[e] = (e % 5) | ((e / 5 % 5) << 16); }
        std::vector<float> x((size_t)n * cin), w((size_t)(N / 16) * G * 64 * 4);
        for (auto &v : x) v = (float)rand() / RAND_MAX;
        for (auto &v : w) v = (float)rand() / RAND_MAX - 0.5f;
        int *d_rowptr, *d_col, *d_code, *d_cnt; float *d_x, *d_skip, *d_out, *d_w[nw], *d_bias; long long *d_st;
        CK(hipMalloc(&d_rowptr, (n_cap + 1) * 4)); CK(hipMalloc(&d_col, E * 4)); CK(hipMalloc(&d_code, E * 4));
        CK(hipMalloc(&d_cnt, 8)); CK(hipMalloc(&d_x, x.size() * 4 + 4096)); CK(hipMalloc(&d_skip, x.size() * 4 + 4096));
        CK(hipMalloc(&d_out, (size_t)n * N * 4)); CK(hipMalloc(&d_bias, N * 4)); CK(hipMalloc(&d_st, 64 * 8));
        for (int i = 0; i < nw; i++) { CK(hipMalloc(&d_w[i], w.size() * 4)); CK(hipMemcpy(d_w[i], w.data(), w.size() * 4, hipMemcpyHostToDevice)); }
        CK(hipMemcpy(d_rowptr, rowptr.data(), (n_cap + 1) * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_col, col.data(), E * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_code, code.data(), E * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_cnt, cnt.data(), 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_skip, x.data(), x.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(d_bias, 0, N * 4));
        auto conv = [&](int wi) {
            return dagr_spline_conv_fused(d_cnt, n_cap, d_rowptr, d_col, d_code, d_x, cin, cin, cskip ? d_skip : nullptr, cin, cskip,
                                          rx, ry, 4.0f, 4.0f, d_w[wi], d_bias, d_out, N, N, 1, s);
        };
        if (conv(0)) { printf("conv: %s\n", dagr_last_error()); return 1; }
        CK(hipStreamSynchronize(s));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < nw; i++) conv(i);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e9f;
        for (int r = 0; r < reps; r++) {
            CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("n %5d / %5d cin %2d cskip %2d K %4d: %.2f us per launch (graph of %d)", n, n_cap, cin, cskip, K, best * 1000 / nw, nw);
#ifdef TRACE
        // stamp, conv, stamp inside one graph: boundaries and stages of workgroup 0
        hipGraph_t g2; hipGraphExec_t ge2;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        conv(1); k_stamp<<<1, 64, 0, s>>>(d_st); conv(2); k_stamp<<<1, 64, 0, s>>>(d_st + 1);
        CK(hipStreamEndCapture(s, &g2)); CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
        // stage order along the kernel; per stage: wave 0, and the latest wave (the barrier waits for it)
        const int order[] = {0, 1, 7, 8, 2, 9, 3, 4, 10, 5, 11, 6};
        const char *names[] = {"entry", "rowptr", "zero", "root/skip", "col/code", "gathers", "phaseA end", "barrier", "mfma main",
                               "mfma all", "red+barrier", "store+barrier"};
        double w0[12] = {0}, wl[12] = {0}, first[12] = {0};
        double exit_gap = 0;
        const int R = 10;
        for (int r = 0; r < R; r++) {
            CK(hipGraphLaunch(ge2, s)); CK(hipStreamSynchronize(s));
            long long st[2], tr[16][16];
            CK(hipMemcpy(st, d_st, 16, hipMemcpyDeviceToHost));
            CK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_trace), sizeof(tr)));
            for (int k = 0; k < 12; k++) {
                long long mx = tr[0][order[k]], mn = tr[0][order[k]];
                for (int w = 1; w < 16; w++) { mx = tr[w][order[k]] > mx ? tr[w][order[k]] : mx; mn = tr[w][order[k]] < mn ? tr[w][order[k]] : mn; }
                w0[k] += (tr[0][order[k]] - st[0]) * 0.01; wl[k] += (mx - st[0]) * 0.01; first[k] += (mn - st[0]) * 0.01;
            }
            long long mx6 = tr[0][6];
            for (int w = 1; w < 16; w++) mx6 = tr[w][6] > mx6 ? tr[w][6] : mx6;
            exit_gap += (st[1] - mx6) * 0.01;
        }
        printf("\n    us since the previous kernel's stamp (earliest wave / wave 0 / latest wave):");
        for (int k = 0; k < 12; k++) printf("\n      %-14s %6.2f %6.2f %6.2f", names[k], first[k] / R, w0[k] / R, wl[k] / R);
        printf("\n      last wave done -> next kernel's stamp %.2f", exit_gap / R);
#endif
        printf("\n");
    }
    return 0;
}
