// train_ops.hip -- backward halves of the training path (BASELINE config 5).
// The reference trains through torch_scatter's autograd (Pooling: scatter_max / scatter_mean, pooling.py:74-77) and
// torch's index_put backward (to_dense, spline_conv.py:80-107).  Both are pure gathers once the forward's choice
// (the arg-max member per (cluster, channel)) is known, so neither needs a float atomic:
//   * k_pool_argmax : arg[c, ch] = LOWEST node index among the members of cluster c whose x equals the pooled maximum
//                     (torch_scatter's CPU reducer keeps the first maximum; its CUDA kernel's choice among ties is
//                     unspecified) -- the only atomic, an integer min
//   * k_pool_grad   : gx[n, ch] = g[cluster[n], ch] if arg == n else 0   (max)   |   g[cluster[n], ch] / count (mean)
//   * k_dense_grad  : gx[n, ch] = gdense[b, ch, cy, cx] for EVERY node inside the map -- also the ones a later node of the
//                     same cell overwrote: torch's index_put backward is grad[indices] over all written rows, and that is
//                     what the reference back-propagates (held to its code by tests/golden/ref_py_model.npz, train_*)
// All HBM streaming: 4 C n bytes read + 4 C n written per pass, coalesced along the channel axis.
#include "common.hpp"

namespace dagr {
namespace {

__global__ __launch_bounds__(kBlock) void k_pool_argmax(const int32_t *__restrict__ cluster, int64_t total, int C,
                                                       const float *__restrict__ x, int ldx,
                                                       const float *__restrict__ xp, int ldp,
                                                       int32_t *__restrict__ arg) {
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= total) return;
    const int n = (int)(gid / C), ch = (int)(gid % C);
    const int c = cluster[n];
    if (c < 0) return;
    if (x[(size_t)n * ldx + ch] == xp[(size_t)c * ldp + ch]) atomicMin(&arg[(size_t)c * C + ch], n);
}

__global__ __launch_bounds__(kBlock) void k_pool_grad(const int32_t *__restrict__ cluster, int64_t total, int C,
                                                     int aggr, const int32_t *__restrict__ arg,
                                                     const int32_t *__restrict__ count, const float *__restrict__ g,
                                                     int ldg, float *__restrict__ gx, int ldgx) {
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= total) return;
    const int n = (int)(gid / C), ch = (int)(gid % C);
    const int c = cluster[n];
    float v = 0.0f;
    if (c >= 0) {
        const float gc = g[(size_t)c * ldg + ch];
        if (aggr == 0)
            v = arg[(size_t)c * C + ch] == n ? gc : 0.0f;
        else
            v = gc / (float)max(count[c], 1);
    }
    gx[(size_t)n * ldgx + ch] = v;
}

__global__ __launch_bounds__(kBlock) void k_dense_grad(int64_t total, int C, const float *__restrict__ pos,
                                                      const int32_t *__restrict__ batch, float vx, float vy, int B,
                                                      int Hc, int Wc, const float *__restrict__ gdense,
                                                      float *__restrict__ gx,
                                                      int ldgx) {
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= total) return;
    const int n = (int)(gid / C), ch = (int)(gid % C);
    const int cx = (int)(pos[3 * n] / vx), cy = (int)(pos[3 * n + 1] / vy), b = batch[n];
    float v = 0.0f;
    if (cx >= 0 && cx < Wc && cy >= 0 && cy < Hc && b >= 0 && b < B)
        v = gdense[(((size_t)b * C + ch) * Hc + cy) * Wc + cx];
    gx[(size_t)n * ldgx + ch] = v;
}

}  // namespace
}  // namespace dagr

using namespace dagr;

extern "C" {

int dagr_pool_argmax(const int32_t *cluster, int32_t n, const float *x, int32_t ldx, int32_t channels,
                     const float *x_pooled, int32_t ldp, int32_t n_clusters, int32_t *arg, void *stream_) {
    DAGR_CHECK_ARG(n >= 0 && n_clusters >= 0 && channels > 0 && ldx >= channels && ldp >= channels, "bad sizes");
    if (n == 0 || n_clusters == 0) return DAGR_OK;
    DAGR_CHECK_ARG(cluster && x && x_pooled && arg, "NULL pointer");
    hipStream_t stream = (hipStream_t)stream_;
    // 0x7f7f7f7f > any node index: clusters without a member keep it (they do not exist in the forward's output)
    DAGR_CHECK_HIP(hipMemsetAsync(arg, 0x7f, (size_t)n_clusters * channels * 4, stream));
    const int64_t total = (int64_t)n * channels;
    k_pool_argmax<<<(unsigned)ceil_div(total, kBlock), kBlock, 0, stream>>>(cluster, total, channels, x, ldx, x_pooled,
                                                                           ldp, arg);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

int dagr_pool_grad(const int32_t *cluster, int32_t n, int32_t channels, int32_t aggr, const int32_t *arg,
                   const int32_t *count, const float *g, int32_t ldg, float *gx, int32_t ldgx, void *stream_) {
    DAGR_CHECK_ARG(n >= 0 && channels > 0 && ldg >= channels && ldgx >= channels, "bad sizes");
    DAGR_CHECK_ARG(aggr == 0 || aggr == 1, "aggr must be 0 (max) or 1 (mean)");
    if (n == 0) return DAGR_OK;
    DAGR_CHECK_ARG(cluster && g && gx, "NULL pointer");
    DAGR_CHECK_ARG(aggr == 0 ? arg != nullptr : count != nullptr, "max needs arg, mean needs count");
    const int64_t total = (int64_t)n * channels;
    k_pool_grad<<<(unsigned)ceil_div(total, kBlock), kBlock, 0, (hipStream_t)stream_>>>(cluster, total, channels, aggr,
                                                                                       arg, count, g, ldg, gx, ldgx);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

int dagr_to_dense_grad(int32_t n, int32_t channels, const float *pos, const int32_t *batch, float vx, float vy,
                       int32_t batch_size, int32_t Hc, int32_t Wc, const float *gdense, float *gx, int32_t ldgx,
                       void *stream_) {
    DAGR_CHECK_ARG(n >= 0 && channels > 0 && batch_size > 0 && Hc > 0 && Wc > 0 && ldgx >= channels, "bad sizes");
    if (n == 0) return DAGR_OK;
    DAGR_CHECK_ARG(pos && batch && gdense && gx, "NULL pointer");
    const int64_t total = (int64_t)n * channels;
    k_dense_grad<<<(unsigned)ceil_div(total, kBlock), kBlock, 0, (hipStream_t)stream_>>>(
        total, channels, pos, batch, vx, vy, batch_size, Hc, Wc, gdense, gx, ldgx);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

}  // extern "C"
