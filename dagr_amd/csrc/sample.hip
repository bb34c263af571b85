// sample.hip -- image-feature sampling at graph-node positions (only with --use_image).
// Reference: sample_features / _sample_features (src/dagr/model/networks/net.py:193-221):
//   x = 2*(pos_x*W)/(W-1) - 1, y likewise, b = 2*batch/(B'-1) - 1 with B' = max(B, 2);
//   3-D grid_sample(mode='bilinear', align_corners=True, padding zeros) over the volume
//   [C, D=B, h, w]  ->  [N, C], concatenated to x (sampling_skip, net.py:15-17).
// Here: one thread per (node, channel); the feature map is read channels-last ([B,h,w,C], the image
// branch runs in torch.channels_last) so the C channels of a tap are one contiguous row; the result is
// written straight into its column block of the consumer's feature matrix (no torch.cat).  The
// arithmetic follows grid_sample's (unnormalise ((g+1)/2)*(size-1), corner weights, accumulation order
// tnw,tne,tsw,tse,bnw,bne,bsw,bse), including the depth axis.
#include "common.hpp"

namespace dagr {
namespace {

template <typename BatchT>
__global__ __launch_bounds__(kBlock) void k_sample_features(const int32_t *__restrict__ n_ptr, int n_max,
                                                           const float *__restrict__ pos,
                                                           const BatchT *__restrict__ batch,
                                                           const float *__restrict__ feat, int B, int h, int w,
                                                           int C, float fW, float fH, float fWm1, float fHm1,
                                                           float fBm1, float *__restrict__ out, int ldo, int coff) {
    const int n_nodes = n_ptr ? min(*n_ptr, n_max) : n_max;
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int n = (int)(gid / C), c = (int)(gid % C);
    if (n >= n_nodes) return;
    // net.py:196-209
    float gx = pos[3 * (size_t)n] * fW;
    float gy = pos[3 * (size_t)n + 1] * fH;
    float gb = (float)batch[n];
    gx = (2.0f * gx) / fWm1 - 1.0f;
    gy = (2.0f * gy) / fHm1 - 1.0f;
    gb = (2.0f * gb) / fBm1 - 1.0f;
    // grid_sampler_unnormalize, align_corners=True
    const float ix = ((gx + 1.0f) / 2.0f) * (float)(w - 1);
    const float iy = ((gy + 1.0f) / 2.0f) * (float)(h - 1);
    const float iz = ((gb + 1.0f) / 2.0f) * (float)(B - 1);
    const float x0f = floorf(ix), y0f = floorf(iy), z0f = floorf(iz);
    const int x0 = (int)x0f, y0 = (int)y0f, z0 = (int)z0f;
    const int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
    const float wx0 = (x0f + 1.0f) - ix, wx1 = ix - x0f;
    const float wy0 = (y0f + 1.0f) - iy, wy1 = iy - y0f;
    const float wz0 = (z0f + 1.0f) - iz, wz1 = iz - z0f;
    auto tap = [&](int z, int y, int x) -> float {
        if (z < 0 || z >= B || y < 0 || y >= h || x < 0 || x >= w) return 0.0f;
        return feat[(((size_t)z * h + y) * w + x) * C + c];
    };
    float acc = 0.0f;
    acc += tap(z0, y0, x0) * ((wx0 * wy0) * wz0);  // tnw
    acc += tap(z0, y0, x1) * ((wx1 * wy0) * wz0);  // tne
    acc += tap(z0, y1, x0) * ((wx0 * wy1) * wz0);  // tsw
    acc += tap(z0, y1, x1) * ((wx1 * wy1) * wz0);  // tse
    acc += tap(z1, y0, x0) * ((wx0 * wy0) * wz1);  // bnw
    acc += tap(z1, y0, x1) * ((wx1 * wy0) * wz1);  // bne
    acc += tap(z1, y1, x0) * ((wx0 * wy1) * wz1);  // bsw
    acc += tap(z1, y1, x1) * ((wx1 * wy1) * wz1);  // bse
    out[(size_t)n * ldo + coff + c] = acc;
}

}  // namespace
}  // namespace dagr

using namespace dagr;

extern "C" int dagr_sample_features(const int32_t *n_ptr, int32_t n_max, const float *pos, const void *batch,
                                    int32_t batch_is_int64, const float *feat_nhwc, int32_t B, int32_t h, int32_t w,
                                    int32_t C, int32_t width, int32_t height, float *out, int32_t ldo, int32_t coff,
                                    void *stream_) {
    DAGR_CHECK_ARG(n_max >= 0 && B >= 1 && h >= 1 && w >= 1 && C >= 1 && width > 1 && height > 1, "bad sizes");
    if (n_max == 0) return DAGR_OK;
    DAGR_CHECK_ARG(pos && batch && feat_nhwc && out, "NULL pointer");
    hipStream_t stream = (hipStream_t)stream_;
    const float fBm1 = (float)((B > 1 ? B : 2) - 1);  // net.py:208
    const unsigned grid = (unsigned)ceil_div((int64_t)n_max * C, kBlock);
    if (batch_is_int64)
        k_sample_features<int64_t><<<grid, kBlock, 0, stream>>>(n_ptr, n_max, pos, (const int64_t *)batch, feat_nhwc, B,
                                                                h, w, C, (float)width, (float)height,
                                                                (float)(width - 1), (float)(height - 1), fBm1, out,
                                                                ldo, coff);
    else
        k_sample_features<int32_t><<<grid, kBlock, 0, stream>>>(n_ptr, n_max, pos, (const int32_t *)batch, feat_nhwc, B,
                                                                h, w, C, (float)width, (float)height,
                                                                (float)(width - 1), (float)(height - 1), fBm1, out,
                                                                ldo, coff);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}
