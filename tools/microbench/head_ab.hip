// Fused (multi-pass, LDS tile) vs unfused (tap_aggregate -> HBM -> GEMM) pooled-level conv through the C ABI of the
// built library, for the wide rows of the head convs.  us per conv inside a HIP graph of 10.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -o tools/microbench/head_ab tools/microbench/head_ab.hip \
//        -Ldagr_amd/lib -ldagr_hip -Wl,-rpath,'$ORIGIN/../../dagr_amd/lib'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "dagr_hip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    const int nw = 10, reps = 20;
    hipStream_t s; CK(hipStreamCreate(&s));
    // live nodes, capacity, cin, cskip, N, degree
    const int cfg[][6] = {{36, 70, 128, 0, 256, 5},  {36, 70, 256, 0, 7, 5},   {36, 70, 128, 0, 5, 5},
                          {141, 280, 128, 0, 256, 5}, {141, 280, 256, 0, 7, 5}, {141, 280, 128, 0, 5, 5},
                          {282, 315, 128, 0, 256, 6}, {282, 315, 256, 0, 7, 6}, {282, 315, 128, 0, 5, 6},
                          {1122, 1260, 128, 0, 256, 7}, {1122, 1260, 256, 0, 7, 7}, {1122, 1260, 128, 0, 5, 7},
                          {1122, 1260, 130, 130, 128, 7}, {4482, 5040, 130, 130, 128, 7}, {4482, 5040, 98, 98, 96, 7},
                          {17922, 20160, 98, 26, 96, 8}};
    for (const auto &c : cfg) {
        const int n = c[0], n_cap = c[1], cin = c[2], cskip = c[3], N = c[4], deg = c[5], rx = 2, ry = 2;
        const int K = 26 * cin + cskip, G = (K + 15) / 16, E = n * deg;
        std::vector<int> rowptr(n_cap + 1), col(E), code(E), cnt{n, E};
        for (int i = 0; i <= n_cap; i++) rowptr[i] = (i < n ? i : n) * deg;
        for (int e = 0; e < E; e++) { col[e] = (int)(((long long)e * 7919 + 3) % n); code[e] = (e % 5) | ((e / 5 % 5) << 16); }
        const int lds = cskip ? cskip : 1, ldw = (N + 7) / 8 * 8, lda = (K + 3) / 4 * 4;
        std::vector<float> x((size_t)n_cap * cin), xs((size_t)n_cap * lds);
        for (auto &v : x) v = (float)rand() / RAND_MAX;
        for (auto &v : xs) v = (float)rand() / RAND_MAX;
        const size_t wq_elems = (size_t)((N + 15) / 16) * G * 64 * 4, wm_elems = (size_t)K * ldw;
        std::vector<float> w(wq_elems > wm_elems ? wq_elems : wm_elems);
        for (auto &v : w) v = (float)rand() / RAND_MAX - 0.5f;
        int *d_rowptr, *d_col, *d_code, *d_cnt; float *d_x, *d_xs, *d_out, *d_w[nw], *d_bias, *d_A;
        CK(hipMalloc(&d_rowptr, (n_cap + 1) * 4)); CK(hipMalloc(&d_col, E * 4)); CK(hipMalloc(&d_code, E * 4));
        CK(hipMalloc(&d_cnt, 8)); CK(hipMalloc(&d_x, x.size() * 4)); CK(hipMalloc(&d_xs, xs.size() * 4));
        CK(hipMalloc(&d_out, (size_t)n_cap * N * 4)); CK(hipMalloc(&d_bias, N * 4)); CK(hipMalloc(&d_A, (size_t)n_cap * lda * 4));
        for (int i = 0; i < nw; i++) { CK(hipMalloc(&d_w[i], w.size() * 4)); CK(hipMemcpy(d_w[i], w.data(), w.size() * 4, hipMemcpyHostToDevice)); }
        CK(hipMemcpy(d_rowptr, rowptr.data(), (n_cap + 1) * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_col, col.data(), E * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_code, code.data(), E * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_cnt, cnt.data(), 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_xs, xs.data(), xs.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(d_bias, 0, N * 4));
        auto fused = [&](int wi) {
            return dagr_spline_conv_fused(d_cnt, n_cap, d_rowptr, d_col, d_code, d_x, cin, cin, cskip ? d_xs : nullptr, lds, cskip,
                                          rx, ry, 4.0f, 4.0f, d_w[wi], d_bias, d_out, N, N, 1, s);
        };
        auto unfused = [&](int wi) {
            int rc = dagr_spline_tap_aggregate(d_cnt, n_cap, d_rowptr, d_col, d_code, d_x, cin, cin, cskip ? d_xs : nullptr, lds,
                                               cskip, rx, ry, 4.0f, 4.0f, d_A, lda, s);
            if (rc) return rc;
            return dagr_gemm_bias_act(d_cnt, n_cap, d_A, lda, d_w[wi], ldw, d_bias, d_out, N, K, N, 1, s);
        };
        float t[2] = {-1.f, -1.f};
        for (int mode = 0; mode < 2; mode++) {
            if (mode == 0 && dagr_spline_conv_fused_lds_bytes(cin, cskip) > 160 * 1024) continue;
            if ((mode ? unfused(0) : fused(0)) != 0) { printf("call failed: %s\n", dagr_last_error()); return 1; }
            CK(hipStreamSynchronize(s));
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            for (int i = 0; i < nw; i++) { if (mode) unfused(i); else fused(i); }
            CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            float best = 1e9f;
            for (int r = 0; r < reps; r++) {
                CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            t[mode] = best * 1000 / nw;
        }
        printf("n %5d / %5d cin %3d cskip %3d N %3d K %4d: fused %.2f us, unfused (2 launches) %.2f us\n", n, n_cap, cin, cskip, N, K, t[0], t[1]);
        hipFree(d_rowptr); hipFree(d_col); hipFree(d_code); hipFree(d_cnt); hipFree(d_x); hipFree(d_xs); hipFree(d_out); hipFree(d_bias); hipFree(d_A);
        for (int i = 0; i < nw; i++) hipFree(d_w[i]);
    }
    return 0;
}
