// nms.hip -- batched greedy NMS for the detection post-processing (one workgroup per image).
// Reference: postprocess_network_output + batched_nms_coordinate_trick (src/dagr/model/utils.py:25-33,61-110):
// per image a Python loop, class-offset boxes, torchvision.ops.nms (sort by score, suppress IoU > thr).
// Here all B images run in one launch: bitonic sort of (score, index) in LDS, then the sequential greedy
// pass with the candidates' boxes resident in LDS (<= 1024 anchors; DAGR has 175).  Ties in score are
// broken by ascending anchor index (torchvision leaves them unspecified).
#include "common.hpp"

namespace dagr {
namespace {
constexpr int kMaxAnchors = 1024;

__global__ __launch_bounds__(kBlock) void k_nms(const float *__restrict__ boxes, const float *__restrict__ scores,
                                               const int32_t *__restrict__ cls, const uint8_t *__restrict__ valid,
                                               int A, int Apad, float thr, float class_offset,
                                               int32_t *__restrict__ order_out, int32_t *__restrict__ keep_out,
                                               int32_t *__restrict__ n_keep) {
    __shared__ float s_key[kMaxAnchors];
    __shared__ int s_idx[kMaxAnchors];
    __shared__ float4 s_box[kMaxAnchors];
    __shared__ int s_keep[kMaxAnchors];
    __shared__ int s_count;
    const int b = blockIdx.x;
    const float *bx = boxes + (size_t)b * A * 4;
    for (int i = threadIdx.x; i < Apad; i += kBlock) {
        const bool ok = i < A && valid[(size_t)b * A + i];
        s_key[i] = ok ? scores[(size_t)b * A + i] : -INFINITY;
        s_idx[i] = i < A ? i : 0x7fffffff;
    }
    __syncthreads();
    // bitonic sort, descending by score, ascending by index on ties
    for (int k = 2; k <= Apad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < Apad; i += kBlock) {
                const int p = i ^ j;
                if (p > i) {
                    const float ka = s_key[i], kb = s_key[p];
                    const int ia = s_idx[i], ib = s_idx[p];
                    const bool a_first = (ka > kb) || (ka == kb && ia < ib);   // a should precede b
                    const bool up = (i & k) == 0;
                    if (up ? !a_first : a_first) {
                        s_key[i] = kb; s_key[p] = ka; s_idx[i] = ib; s_idx[p] = ia;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < A; i += kBlock) {
        const int a = s_idx[i];
        const bool ok = s_key[i] > -INFINITY;
        float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
            const float off = (float)cls[(size_t)b * A + a] * class_offset;   // idxs * float(max_dim + 1)
            bb = make_float4(bx[4 * a] + off, bx[4 * a + 1] + off, bx[4 * a + 2] + off, bx[4 * a + 3] + off);
        }
        s_box[i] = bb;
        s_keep[i] = ok ? 1 : 0;
    }
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
    for (int i = 0; i < A; i++) {
        if (s_keep[i]) {   // uniform: read after the barrier below
            const float4 bi = s_box[i];
            const float area_i = (bi.z - bi.x) * (bi.w - bi.y);
            for (int j = i + 1 + threadIdx.x; j < A; j += kBlock) {
                if (!s_keep[j]) continue;
                const float4 bj = s_box[j];
                const float w = fmaxf(fminf(bi.z, bj.z) - fmaxf(bi.x, bj.x), 0.f);
                const float h = fmaxf(fminf(bi.w, bj.w) - fmaxf(bi.y, bj.y), 0.f);
                const float inter = w * h;
                const float area_j = (bj.z - bj.x) * (bj.w - bj.y);
                if (inter / (area_i + area_j - inter) > thr) s_keep[j] = 0;
            }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < A; i += kBlock) {
        order_out[(size_t)b * A + i] = s_idx[i];
        keep_out[(size_t)b * A + i] = s_keep[i];
        if (s_keep[i]) atomicAdd(&s_count, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) n_keep[b] = s_count;
}
}  // namespace
}  // namespace dagr

using namespace dagr;

extern "C" int dagr_nms_batched(const float *boxes, const float *scores, const int32_t *cls, const uint8_t *valid,
                                int32_t B, int32_t A, float iou_threshold, float class_offset, int32_t *order_out,
                                int32_t *keep_out, int32_t *n_keep, void *stream) {
    DAGR_CHECK_ARG(B >= 0 && A >= 0 && A <= kMaxAnchors, "A must be <= 1024");
    if (B == 0 || A == 0) return DAGR_OK;
    DAGR_CHECK_ARG(boxes && scores && cls && valid && order_out && keep_out && n_keep, "NULL pointer");
    int Apad = 1;
    while (Apad < A) Apad <<= 1;
    k_nms<<<B, kBlock, 0, (hipStream_t)stream>>>(boxes, scores, cls, valid, A, Apad, iou_threshold, class_offset,
                                                 order_out, keep_out, n_keep);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}
