// asy_tools.hip -- 1:1 replacements of the reference's native module `asy_tools`
// (src/dagr/asynchronous/asy_tools/main.cu:239-244): the masked row operators its asynchronous (per-event) network
// update is built from.  `indices` (int64[K]) selects the rows of [num_nodes, C] feature matrices that changed; only
// those rows are touched.  Same arguments in the same order as the reference's host functions, shapes riding along as
// integers; same arithmetic order (a k-ordered fma chain per output, bias last; BN as ((x - mean) / sqrt(var + eps)) *
// w + b), so results are bit-identical to the reference's kernels built by hipcc (tests/test_asy_tools_gpu.py).
//
// The reference maps one thread to one (row, output channel) and streams the weight matrix with a stride of Cin per
// lane.  Here a 64-lane wave owns a row: the row of x_in is read once, coalesced, and kept in LDS; lane = output
// channel walks its weight row (K rows share the same few-KiB matrix through L1/L2).
#include "common.hpp"

namespace dagr {
namespace {

constexpr int kRowsPerBlock = kBlock / 64;

template <bool kBias>
__global__ __launch_bounds__(kBlock) void k_masked_lin(const int64_t *__restrict__ indices,
                                                      const float *__restrict__ x_in, float *__restrict__ x_out,
                                                      const float *__restrict__ weight, const float *__restrict__ bias,
                                                      int K, int Cin, int Cout, int add) {
    extern __shared__ float xs[];                       // [kRowsPerBlock][Cin]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * kRowsPerBlock + wv;
    if (i >= K) return;
    const int64_t row = indices[i];
    float *xr = xs + wv * Cin;
    for (int c = lane; c < Cin; c += 64) xr[c] = x_in[row * Cin + c];
    __builtin_amdgcn_wave_barrier();
    for (int co = lane; co < Cout; co += 64) {
        float acc = add ? x_out[row * Cout + co] : 0.0f;
        const float *w = weight + (size_t)co * Cin;
        for (int c = 0; c < Cin; c++) acc = fmaf(xr[c], w[c], acc);      // main.cu:172-174 / :204-206, in order
        if (kBias) acc += bias[co];                                      // main.cu:175
        x_out[row * Cout + co] = acc;
    }
}

__global__ __launch_bounds__(kBlock) void k_masked_isdiff(int64_t *__restrict__ indices, const float *__restrict__ x_old,
                                                         const float *__restrict__ x_new, int K, int C, float atol,
                                                         float rtol) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    if (i >= K) return;
    const int64_t row = indices[i];
    bool diff = false;
    for (int c = lane; c < C; c += 64) {
        const float a = x_old[row * C + c], b = x_new[row * C + c];
        diff |= fabsf(a - b) > fmaf(rtol, b, atol);                      // main.cu:36: atol + rtol * other (sign of other included)
    }
    const bool any = __ballot(diff) != 0;                                // wave = row: every lane sees its own row's verdict
    if (lane == 0 && !any) indices[i] = -1;                              // kept rows keep their index (main.cu:31-39)
}

__global__ __launch_bounds__(kBlock) void k_masked_bn(const int64_t *__restrict__ indices, const float *__restrict__ x,
                                                     float *__restrict__ x_out, const float *__restrict__ mean,
                                                     const float *__restrict__ var, const float *__restrict__ w,
                                                     const float *__restrict__ b, int K, int C, float eps) {
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= (int64_t)K * C) return;
    const int i = (int)(gid / C), c = (int)(gid % C);
    const int64_t at = indices[i] * C + c;
    const float t = (x[at] - mean[c]) / sqrtf(var[c] + eps);             // main.cu:66
    x_out[at] = fmaf(t, w[c], b[c]);
}

}  // namespace
}  // namespace dagr

using namespace dagr;

extern "C" {

int dagr_masked_lin(const int64_t *indices, const float *x_in, float *x_out, const float *weight, const float *bias,
                    int32_t add, int64_t K, int32_t Cin, int32_t Cout, void *stream) {
    DAGR_CHECK_ARG(K >= 0 && Cin >= 1 && Cout >= 1, "bad sizes");
    if (K == 0) return DAGR_OK;
    DAGR_CHECK_ARG(indices && x_in && x_out && weight, "NULL pointer");
    const size_t lds = (size_t)kRowsPerBlock * Cin * 4;
    DAGR_CHECK_ARG(lds <= 64 * 1024, "Cin too large");
    const unsigned grid = (unsigned)ceil_div(K, kRowsPerBlock);
    if (bias)
        k_masked_lin<true><<<grid, kBlock, lds, (hipStream_t)stream>>>(indices, x_in, x_out, weight, bias, (int)K, Cin,
                                                                       Cout, add);
    else
        k_masked_lin<false><<<grid, kBlock, lds, (hipStream_t)stream>>>(indices, x_in, x_out, weight, nullptr, (int)K,
                                                                        Cin, Cout, add);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

int dagr_masked_lin_no_bias(const int64_t *indices, const float *x_in, float *x_out, const float *weight, int32_t add,
                            int64_t K, int32_t Cin, int32_t Cout, void *stream) {
    return dagr_masked_lin(indices, x_in, x_out, weight, nullptr, add, K, Cin, Cout, stream);
}

int dagr_masked_isdiff(int64_t *indices, const float *x_old, const float *x_new, float atol, float rtol, int64_t K,
                       int32_t C, void *stream) {
    DAGR_CHECK_ARG(K >= 0 && C >= 1, "bad sizes");
    if (K == 0) return DAGR_OK;
    DAGR_CHECK_ARG(indices && x_old && x_new, "NULL pointer");
    k_masked_isdiff<<<(unsigned)ceil_div(K, kRowsPerBlock), kBlock, 0, (hipStream_t)stream>>>(indices, x_old, x_new,
                                                                                            (int)K, C, atol, rtol);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

int dagr_masked_inplace_BN(const int64_t *indices, const float *x, float *x_out, const float *running_mean,
                           const float *running_var, const float *weight, const float *bias, float eps, int64_t K,
                           int32_t C, void *stream) {
    DAGR_CHECK_ARG(K >= 0 && C >= 1, "bad sizes");
    if (K == 0) return DAGR_OK;
    DAGR_CHECK_ARG(indices && x && x_out && running_mean && running_var && weight && bias, "NULL pointer");
    k_masked_bn<<<(unsigned)ceil_div(K * C, kBlock), kBlock, 0, (hipStream_t)stream>>>(
        indices, x, x_out, running_mean, running_var, weight, bias, (int)K, C, eps);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

}  // extern "C"
