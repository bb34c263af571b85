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

// ---------------------------------------------------------------------------------------------
// y = relu(y + z), in place: the residual join of the image branch's ResNet blocks (net_img.py) as one pass
// over the two activation maps instead of torch's add (new tensor) + in-place clamp.
namespace dagr {
namespace {
__global__ __launch_bounds__(kBlock) void k_add_relu(float *__restrict__ y, const float *__restrict__ z, int64_t n4,
                                                    int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
        float4 a = reinterpret_cast<float4 *>(y)[i];
        const float4 b = reinterpret_cast<const float4 *>(z)[i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        a.x = a.x < 0.f ? 0.f : a.x;   // NaN stays NaN, as torch.relu
        a.y = a.y < 0.f ? 0.f : a.y;
        a.z = a.z < 0.f ? 0.f : a.z;
        a.w = a.w < 0.f ? 0.f : a.w;
        reinterpret_cast<float4 *>(y)[i] = a;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {   // tail
        const int64_t i = (n4 << 2) + threadIdx.x;
        const float v = y[i] + z[i];
        y[i] = v < 0.f ? 0.f : v;
    }
}
}  // namespace
}  // namespace dagr

extern "C" int dagr_add_relu(float *y, const float *z, int64_t n, void *stream) {
    using namespace dagr;
    DAGR_CHECK_ARG(n >= 0, "bad size");
    if (n == 0) return DAGR_OK;
    DAGR_CHECK_ARG(y && z, "NULL pointer");
    DAGR_CHECK_ARG(((uintptr_t)y % 16) == 0 && ((uintptr_t)z % 16) == 0, "buffers must be 16-byte aligned");
    const int64_t n4 = n >> 2;
    const int64_t blocks = ceil_div(n4 > 0 ? n4 : 1, (int64_t)kBlock * 4);
    k_add_relu<<<(unsigned)std::min<int64_t>(blocks, 256 * 16), kBlock, 0, (hipStream_t)stream>>>(y, z, n4, n);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

// ---------------------------------------------------------------------------------------------
// ResNet stem tail in one pass: y = maxpool3x3/s2/p1( relu( x * scale[c] + shift[c] ) ) over a channels-last map
// (torchvision ResNet.forward: bn1 -> relu -> maxpool after the raw conv1 output the reference taps,
// net_img.py:80-84).  One thread per (output pixel, 4 channels); eval-mode BatchNorm as the affine pair
// scale = weight / sqrt(var + eps), shift = bias - mean * scale.
namespace dagr {
namespace {
__global__ __launch_bounds__(kBlock) void k_bn_relu_maxpool(const float *__restrict__ x, int B, int H, int W, int C,
                                                           const float *__restrict__ scale,
                                                           const float *__restrict__ shift, float *__restrict__ y,
                                                           int OH, int OW) {
    const int C4 = C >> 2;
    const int64_t total = (int64_t)B * OH * OW * C4;
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= total) return;
    const int c4 = (int)(gid % C4);
    int64_t r = gid / C4;
    const int ow = (int)(r % OW); r /= OW;
    const int oh = (int)(r % OH);
    const int b = (int)(r / OH);
    const float4 sc = reinterpret_cast<const float4 *>(scale)[c4], sh = reinterpret_cast<const float4 *>(shift)[c4];
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);   // relu output >= 0 and every window holds a valid pixel
#pragma unroll
    for (int dy = 0; dy < 3; dy++) {
        const int ih = 2 * oh - 1 + dy;
        if (ih < 0 || ih >= H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; dx++) {
            const int iw = 2 * ow - 1 + dx;
            if (iw < 0 || iw >= W) continue;
            const float4 v = reinterpret_cast<const float4 *>(x + (((size_t)b * H + ih) * W + iw) * C)[c4];
            m.x = fmaxf(m.x, v.x * sc.x + sh.x);
            m.y = fmaxf(m.y, v.y * sc.y + sh.y);
            m.z = fmaxf(m.z, v.z * sc.z + sh.z);
            m.w = fmaxf(m.w, v.w * sc.w + sh.w);
        }
    }
    reinterpret_cast<float4 *>(y + (((size_t)b * OH + oh) * OW + ow) * C)[c4] = m;
}
}  // namespace
}  // namespace dagr

extern "C" int dagr_bn_relu_maxpool(const float *x_nhwc, int32_t B, int32_t H, int32_t W, int32_t C, const float *scale,
                                    const float *shift, float *y_nhwc, void *stream) {
    using namespace dagr;
    DAGR_CHECK_ARG(B >= 1 && H >= 1 && W >= 1 && C >= 4 && C % 4 == 0, "bad sizes (C must be a multiple of 4)");
    DAGR_CHECK_ARG(x_nhwc && scale && shift && y_nhwc, "NULL pointer");
    DAGR_CHECK_ARG(((uintptr_t)x_nhwc % 16) == 0 && ((uintptr_t)y_nhwc % 16) == 0 && ((uintptr_t)scale % 16) == 0 &&
                       ((uintptr_t)shift % 16) == 0, "buffers must be 16-byte aligned");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const int64_t total = (int64_t)B * OH * OW * (C / 4);
    k_bn_relu_maxpool<<<(unsigned)ceil_div(total, (int64_t)kBlock), kBlock, 0, (hipStream_t)stream>>>(
        x_nhwc, B, H, W, C, scale, shift, y_nhwc, OH, OW);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

// ---------------------------------------------------------------------------------------------
// y = relu(y + bias[c]) in place over a channels-last map: MIOpen's fp32 NHWC convolutions leave the (BN-folded)
// bias and the ReLU to two separate elementwise passes; this is one.
namespace dagr {
namespace {
__global__ __launch_bounds__(kBlock) void k_bias_relu(float *__restrict__ y, const float *__restrict__ bias, int64_t n4,
                                                     int C4) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
        float4 a = reinterpret_cast<float4 *>(y)[i];
        const float4 b = reinterpret_cast<const float4 *>(bias)[i % C4];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        a.x = a.x < 0.f ? 0.f : a.x;
        a.y = a.y < 0.f ? 0.f : a.y;
        a.z = a.z < 0.f ? 0.f : a.z;
        a.w = a.w < 0.f ? 0.f : a.w;
        reinterpret_cast<float4 *>(y)[i] = a;
    }
}
}  // namespace
}  // namespace dagr

namespace dagr {
namespace {
// y = silu(y + bias[c]) in place (yolox BaseConv: conv -> folded BN -> SiLU), v / (1 + exp(-v)) as ATen's silu kernel
__global__ __launch_bounds__(kBlock) void k_bias_silu(float *__restrict__ y, const float *__restrict__ bias, int64_t n4,
                                                     int C4) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
        float4 a = reinterpret_cast<float4 *>(y)[i];
        const float4 b = reinterpret_cast<const float4 *>(bias)[i % C4];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        a.x = a.x / (1.0f + expf(-a.x));
        a.y = a.y / (1.0f + expf(-a.y));
        a.z = a.z / (1.0f + expf(-a.z));
        a.w = a.w / (1.0f + expf(-a.w));
        reinterpret_cast<float4 *>(y)[i] = a;
    }
}
}  // namespace
}  // namespace dagr

extern "C" int dagr_bias_silu(float *y_nhwc, const float *bias, int64_t n, int32_t C, void *stream) {
    using namespace dagr;
    DAGR_CHECK_ARG(n >= 0 && C >= 4 && C % 4 == 0 && n % C == 0, "bad sizes (C must be a multiple of 4 and divide n)");
    if (n == 0) return DAGR_OK;
    DAGR_CHECK_ARG(y_nhwc && bias, "NULL pointer");
    DAGR_CHECK_ARG(((uintptr_t)y_nhwc % 16) == 0 && ((uintptr_t)bias % 16) == 0, "buffers must be 16-byte aligned");
    const int64_t n4 = n >> 2;
    const int64_t blocks = ceil_div(n4, (int64_t)kBlock * 4);
    k_bias_silu<<<(unsigned)std::min<int64_t>(blocks, 256 * 16), kBlock, 0, (hipStream_t)stream>>>(y_nhwc, bias, n4, C / 4);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

extern "C" int dagr_bias_relu(float *y_nhwc, const float *bias, int64_t n, int32_t C, void *stream) {
    using namespace dagr;
    DAGR_CHECK_ARG(n >= 0 && C >= 4 && C % 4 == 0 && n % C == 0, "bad sizes (C must be a multiple of 4 and divide n)");
    if (n == 0) return DAGR_OK;
    DAGR_CHECK_ARG(y_nhwc && bias, "NULL pointer");
    DAGR_CHECK_ARG(((uintptr_t)y_nhwc % 16) == 0 && ((uintptr_t)bias % 16) == 0, "buffers must be 16-byte aligned");
    const int64_t n4 = n >> 2;
    const int64_t blocks = ceil_div(n4, (int64_t)kBlock * 4);
    k_bias_relu<<<(unsigned)std::min<int64_t>(blocks, 256 * 16), kBlock, 0, (hipStream_t)stream>>>(y_nhwc, bias, n4, C / 4);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}
