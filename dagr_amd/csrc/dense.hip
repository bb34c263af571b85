// dense.hip -- sparse node features -> dense detection-head maps.
// Reference: to_dense (src/dagr/model/layers/spline_conv.py:80-107): zeroed [B,C,H,W];
// cell = trunc(pos_xy / pooling_xy) (fp32 division, QUIRK-2); dense[batch, :, cy, cx] = x.
// Duplicate targets (only possible through the t == 1.0 leak, QUIRK-1) are last-writer-wins in
// the reference (non-deterministic on GPU); here the highest node index wins, which is what the
// sequential CPU index_put of the oracle does.
#include "common.hpp"

namespace dagr {
namespace {

__global__ __launch_bounds__(kBlock) void k_dense_winner(const int32_t *__restrict__ n_ptr, int n_max,
                                                        const float *__restrict__ pos,
                                                        const int32_t *__restrict__ batch, float vx, float vy, int B,
                                                        int Hc, int Wc, int32_t *__restrict__ winner,
                                                        int32_t *__restrict__ status) {
    const int n_nodes = n_ptr ? min(*n_ptr, n_max) : n_max;
    const int n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= n_nodes) return;
    const int cx = (int)(pos[3 * n] / vx), cy = (int)(pos[3 * n + 1] / vy), b = batch[n];
    if (cx < 0 || cx >= Wc || cy < 0 || cy >= Hc || b < 0 || b >= B) { atomicOr(status, 1); return; }
    atomicMax(&winner[(b * Hc + cy) * Wc + cx], n);
}

__global__ __launch_bounds__(kBlock) void k_dense_write(int B, int C, int Hc, int Wc,
                                                       const int32_t *__restrict__ winner,
                                                       const float *__restrict__ x, int ldx,
                                                       float *__restrict__ dense) {
    const int gid = blockIdx.x * kBlock + threadIdx.x;
    if (gid >= B * C * Hc * Wc) return;
    const int cx = gid % Wc, cy = (gid / Wc) % Hc, ch = (gid / (Wc * Hc)) % C, b = gid / (Wc * Hc * C);
    const int n = winner[(b * Hc + cy) * Wc + cx];
    dense[gid] = n >= 0 ? x[(size_t)n * ldx + ch] : 0.0f;
}

}  // namespace
}  // namespace dagr

using namespace dagr;

extern "C" int dagr_to_dense(const int32_t *n_ptr, int32_t n_max, const float *x, int32_t ldx, int32_t channels,
                             const float *pos, const int32_t *batch, float vx, float vy, int32_t batch_size,
                             int32_t Hc, int32_t Wc, int32_t *winner_scratch, float *dense, int32_t *status,
                             void *stream_) {
    DAGR_CHECK_ARG(batch_size > 0 && Hc > 0 && Wc > 0 && channels > 0, "bad sizes");
    DAGR_CHECK_ARG(winner_scratch && dense && status, "NULL pointer");
    hipStream_t stream = (hipStream_t)stream_;
    const int cells = batch_size * Hc * Wc;
    DAGR_CHECK_HIP(hipMemsetAsync(winner_scratch, 0xff, (size_t)cells * 4, stream));
    if (n_max > 0) {
        DAGR_CHECK_ARG(x && pos && batch, "NULL input");
        k_dense_winner<<<(unsigned)ceil_div(n_max, kBlock), kBlock, 0, stream>>>(n_ptr, n_max, pos, batch, vx, vy,
                                                                               batch_size, Hc, Wc, winner_scratch,
                                                                               status);
        DAGR_CHECK_LAUNCH();
    }
    k_dense_write<<<(unsigned)ceil_div((int64_t)cells * channels, kBlock), kBlock, 0, stream>>>(
        batch_size, channels, Hc, Wc, winner_scratch, x, ldx, dense);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}
