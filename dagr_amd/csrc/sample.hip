// sample.hip -- image-feature sampling at graph-node positions (only with --use_image).
// Reference: sample_features / _sample_features (src/dagr/model/networks/net.py:193-221):
//   x = 2*(pos_x*W)/(W-1) - 1, y likewise, b = 2*batch/(B'-1) - 1 with B' = max(B, 2);
//   3-D grid_sample(mode='bilinear', align_corners=True, padding zeros) over the volume
//   [C, D=B, h, w]  ->  [N, C], concatenated to x (sampling_skip, net.py:15-17).
// Here: one thread per (node, 4 channels); the feature map is read channels-last ([B,h,w,C], the image
// branch runs in torch.channels_last) so the C channels of a tap are one contiguous row; the result is
// written straight into its column block of the consumer's feature matrix (no torch.cat).  The
// arithmetic follows grid_sample's (unnormalise ((g+1)/2)*(size-1), corner weights, accumulation order
// tnw,tne,tsw,tse,bnw,bne,bsw,bse), including the depth axis.
#include "common.hpp"

namespace dagr {
namespace {

// V = channels per thread (4 when the channel count, the output row stride and column offset allow 16-byte
// accesses): the coordinate arithmetic -- six IEEE divisions per node -- is shared by V channels and the
// taps become float4 loads.  Taps whose weight is exactly zero are not fetched (on the depth axis the sample
// index is an integer, so the whole z1 plane usually drops out; x*0 contributes +0 for finite features).
template <typename BatchT, int V, bool ALIGNED>
__global__ __launch_bounds__(kBlock) void k_sample_features(const int32_t *__restrict__ n_ptr, int n_max,
                                                           const float *__restrict__ pos,
                                                           const BatchT *__restrict__ batch,
                                                           const float *__restrict__ feat, int B, int h, int w,
                                                           int C, float fW, float fH, float fWm1, float fHm1,
                                                           float fBm1, float *__restrict__ out, int ldo, int coff) {
    const int n_nodes = n_ptr ? min(*n_ptr, n_max) : n_max;
    const int CV = C / V;
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int n = (int)(gid / CV), c = (int)(gid % CV) * V;
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
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; v++) acc[v] = 0.0f;
    auto tap = [&](int z, int y, int x, float wgt) {
        if (wgt == 0.0f || z < 0 || z >= B || y < 0 || y >= h || x < 0 || x >= w) return;
        const float *src = feat + (((size_t)z * h + y) * w + x) * C + c;
        if (V == 4 && ALIGNED) {
            const float4 f = *reinterpret_cast<const float4 *>(src);
            acc[0] += f.x * wgt; acc[1 % V] += f.y * wgt; acc[2 % V] += f.z * wgt; acc[3 % V] += f.w * wgt;
        } else {
#pragma unroll
            for (int v = 0; v < V; v++) acc[v] += src[v] * wgt;
        }
    };
    tap(z0, y0, x0, (wx0 * wy0) * wz0);  // tnw
    tap(z0, y0, x1, (wx1 * wy0) * wz0);  // tne
    tap(z0, y1, x0, (wx0 * wy1) * wz0);  // tsw
    tap(z0, y1, x1, (wx1 * wy1) * wz0);  // tse
    tap(z1, y0, x0, (wx0 * wy0) * wz1);  // bnw
    tap(z1, y0, x1, (wx1 * wy0) * wz1);  // bne
    tap(z1, y1, x0, (wx0 * wy1) * wz1);  // bsw
    tap(z1, y1, x1, (wx1 * wy1) * wz1);  // bse
    float *dst = out + (size_t)n * ldo + coff + c;
    if (V == 4 && ALIGNED) {
        *reinterpret_cast<float4 *>(dst) = make_float4(acc[0], acc[1 % V], acc[2 % V], acc[3 % V]);
    } else {
#pragma unroll
        for (int v = 0; v < V; v++) dst[v] = acc[v];
    }
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
    // 4 channels per thread whenever C allows; float4 accesses only if the feature rows (always, then) and the
    // destination columns are 16-byte aligned (the level-0 input matrix has 19 columns: scalar stores)
    const bool four = (C % 4 == 0) && ((uintptr_t)feat_nhwc % 16 == 0);
    const bool aligned = four && (ldo % 4 == 0) && (coff % 4 == 0) && ((uintptr_t)out % 16 == 0);
    const unsigned grid = (unsigned)ceil_div((int64_t)n_max * (four ? C / 4 : C), kBlock);
#define DAGR_SAMPLE(BT, V, A)                                                                                           \
    k_sample_features<BT, V, A><<<grid, kBlock, 0, stream>>>(n_ptr, n_max, pos, (const BT *)batch, feat_nhwc, B, h, w, C, \
                                                             (float)width, (float)height, (float)(width - 1),          \
                                                             (float)(height - 1), fBm1, out, ldo, coff)
#define DAGR_SAMPLE_BT(BT)                                 \
    do {                                                   \
        if (aligned) DAGR_SAMPLE(BT, 4, true);             \
        else if (four) DAGR_SAMPLE(BT, 4, false);          \
        else DAGR_SAMPLE(BT, 1, false);                    \
    } while (0)
    if (batch_is_int64) DAGR_SAMPLE_BT(int64_t);
    else DAGR_SAMPLE_BT(int32_t);
#undef DAGR_SAMPLE_BT
#undef DAGR_SAMPLE
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}
