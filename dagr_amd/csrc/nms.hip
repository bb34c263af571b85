// nms.hip -- batched greedy NMS for the detection post-processing (one workgroup per image).
// Reference: postprocess_network_output + batched_nms_coordinate_trick (src/dagr/model/utils.py:25-33,61-110):
// per image a Python loop, class-offset boxes, torchvision.ops.nms (sort by score, suppress IoU > thr).
// Here all B images run in one launch: bitonic sort of (score, index) in LDS, then the sequential greedy
// pass with the candidates' boxes resident in LDS (<= 1024 anchors; DAGR has 175).  Ties in score are
// broken by ascending anchor index (torchvision leaves them unspecified).
#include "common.hpp"

#include <algorithm>

namespace dagr {
namespace {
constexpr int kMaxAnchors = 1024;
constexpr int kMaskAnchors = 256, kMaskWords = kMaskAnchors / 64;
static_assert(kBlock >= kMaskAnchors, "the rank sort gives one thread to every key");

// Greedy suppression over the score-sorted candidates (s_keep = candidate flags in, survivor flags out): box i, if still
// alive when its turn comes, removes every later box j with IoU(i, j) > thr.  The chain over i is inherently sequential,
// but the pairwise tests are not: for up to 256 candidates (DAGR has 175 anchors) all "i would suppress j" bits are
// computed first -- one wave per row i, one ballot per 64 columns -- and the chain then only ORs 64-bit rows: one wave
// walks it with the removed-set in four registers and the next row prefetched from LDS (~20 cycles per candidate
// instead of a block-wide barrier per candidate: 66 us -> a few us per image at 175 anchors).  Same IoU expression,
// same order, same result as the reference's torchvision.ops.nms on the offset boxes.
__device__ __forceinline__ bool iou_above(const float4 bi, float area_i, const float4 bj, float thr) {
    const float w = fmaxf(fminf(bi.z, bj.z) - fmaxf(bi.x, bj.x), 0.f);
    const float h = fmaxf(fminf(bi.w, bj.w) - fmaxf(bi.y, bj.y), 0.f);
    const float inter = w * h;
    const float area_j = (bj.z - bj.x) * (bj.w - bj.y);
    return inter / (area_i + area_j - inter) > thr;
}

// Descending sort of (key, index) pairs with ties broken by ascending index, in LDS.  Up to 256 candidates: rank sort --
// every element counts the elements that precede it (175 broadcast reads per thread) and drops itself at that position:
// two barriers instead of the 36 of a 256-wide bitonic network, which were most of k_postprocess (54 us per launch).
// Larger inputs keep the bitonic network.  s_idx must hold i at position i on entry (pads: 0x7fffffff, key -inf).
__device__ void sort_desc(int A, int Apad, float *s_key, int *s_idx, float *s_key2) {
    if (A <= kMaskAnchors) {
        float ki = 0.f;
        int rank = 0;
        const int i = threadIdx.x;
        if (i < A) {
            ki = s_key[i];
            for (int j = 0; j < A; j++) {
                const float kj = s_key[j];
                rank += ((kj > ki) || (kj == ki && j < i)) ? 1 : 0;
            }
        }
        __syncthreads();
        if (i < A) { s_key2[rank] = ki; s_idx[rank] = i; }
        __syncthreads();
        if (i < A) s_key[i] = s_key2[i];
        __syncthreads();
        return;
    }
    for (int k = 2; k <= Apad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < Apad; i += (int)blockDim.x) {
                const int q = i ^ j;
                if (q > i) {
                    const float ka = s_key[i], kb = s_key[q];
                    const int ia = s_idx[i], ib = s_idx[q];
                    const bool a_first = (ka > kb) || (ka == kb && ia < ib);   // a should precede b
                    const bool up = (i & k) == 0;
                    if (up ? !a_first : a_first) {
                        s_key[i] = kb; s_key[q] = ka; s_idx[i] = ib; s_idx[q] = ia;
                    }
                }
            }
            __syncthreads();
        }
    }
}

__device__ void greedy_suppress(int A, const float4 *s_box, int *s_keep, float thr,
                                unsigned long long (*s_mask)[kMaskWords]) {
    if (A > kMaskAnchors) {      // generic path: one barrier per candidate
        for (int i = 0; i < A; i++) {
            if (s_keep[i]) {   // uniform: read after the barrier below
                const float4 bi = s_box[i];
                const float area_i = (bi.z - bi.x) * (bi.w - bi.y);
                for (int j = i + 1 + threadIdx.x; j < A; j += (int)blockDim.x)
                    if (s_keep[j] && iou_above(bi, area_i, s_box[j], thr)) s_keep[j] = 0;
            }
            __syncthreads();
        }
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    for (int i = wave; i < A; i += n_waves) {
        const float4 bi = s_box[i];
        const float area_i = (bi.z - bi.x) * (bi.w - bi.y);
        const bool vi = s_keep[i] != 0;
#pragma unroll
        for (int w = 0; w < kMaskWords; w++) {
            const int j = w * 64 + lane;
            const bool hit = vi && j > i && j < A && s_keep[j] != 0 && iou_above(bi, area_i, s_box[j], thr);
            const unsigned long long m = __ballot(hit);
            if (lane == 0) s_mask[i][w] = m;
        }
    }
    __syncthreads();
    if (wave == 0) {
        // The chain.  Everything here is wave-uniform: the removed-set lives in scalar registers, a block of 64 rows is
        // pulled from LDS once (lane l holds row 64 blk + l) and row `bit` is then broadcast with readlane -- no LDS
        // round trip inside the dependent loop.
        unsigned long long removed[kMaskWords] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
        for (int blk = 0; blk < kMaskWords; blk++) {
            if (blk * 64 < A) {
                const int row = blk * 64 + lane;
                unsigned long long mine[kMaskWords];
#pragma unroll
                for (int w = 0; w < kMaskWords; w++) mine[w] = row < A ? s_mask[row][w] : 0ull;
                // only candidates whose row has a bit set can remove anything: walk those (few when boxes barely overlap,
                // all of them around a crowded object)
                bool any = false;
#pragma unroll
                for (int w = 0; w < kMaskWords; w++) any = any || mine[w] != 0ull;
                unsigned long long todo = __ballot(row < A && s_keep[row] != 0 && any);
                while (todo) {
                    const int bit = __ffsll((long long)todo) - 1;
                    todo &= todo - 1ull;
                    const bool alive = ((removed[blk] >> bit) & 1ull) == 0ull;
                    if (alive) {
#pragma unroll
                        for (int w = 0; w < kMaskWords; w++) {
                            if (w >= blk) {      // a row only carries bits of later candidates: words below its own are 0
                                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(mine[w] & 0xffffffffull), bit);
                                const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(mine[w] >> 32), bit);
                                removed[w] |= ((unsigned long long)hi << 32) | lo;
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int w = 0; w < kMaskWords; w++) {
            const int i = w * 64 + lane;
            if (i < A && ((removed[w] >> lane) & 1ull)) s_keep[i] = 0;
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(kBlock) void k_nms(const float *__restrict__ boxes, const float *__restrict__ scores,
                                               const int32_t *__restrict__ cls, const uint8_t *__restrict__ valid,
                                               int A, int Apad, float thr, float class_offset,
                                               int32_t *__restrict__ order_out, int32_t *__restrict__ keep_out,
                                               int32_t *__restrict__ n_keep) {
    __shared__ float s_key[kMaxAnchors];
    __shared__ int s_idx[kMaxAnchors];
    __shared__ float4 s_box[kMaxAnchors];
    __shared__ int s_keep[kMaxAnchors];
    __shared__ unsigned long long s_mask[kMaskAnchors][kMaskWords];
    __shared__ float s_key2[kMaskAnchors];
    __shared__ int s_count;
    const int b = blockIdx.x;
    const float *bx = boxes + (size_t)b * A * 4;
    for (int i = threadIdx.x; i < Apad; i += kBlock) {
        const bool ok = i < A && valid[(size_t)b * A + i];
        const float sc = ok ? scores[(size_t)b * A + i] : -INFINITY;
        s_key[i] = sc == sc ? sc : -INFINITY;       // a NaN key has no rank (the sort needs a total order): never kept
        s_idx[i] = i < A ? i : 0x7fffffff;
    }
    __syncthreads();
    sort_desc(A, Apad, s_key, s_idx, s_key2);      // descending by score, ascending by index on ties
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
    greedy_suppress(A, s_box, s_keep, thr, s_mask);
    for (int i = threadIdx.x; i < A; i += kBlock) {
        order_out[(size_t)b * A + i] = s_idx[i];
        keep_out[(size_t)b * A + i] = s_keep[i];
        if (s_keep[i]) atomicAdd(&s_count, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) n_keep[b] = s_count;
}

// Whole post-processing of one window batch in one launch (model/utils.py:61-110, filtering=True): decoded head
// rows [cx, cy, w, h, obj, cls...] -> xyxy (x1 = cx - w/2, x2 = w + x1: the reference's in-place op order), class
// max / argmax, score = obj * class_conf, confidence mask (score * class_conf >= thr), class-offset greedy NMS, and
// the survivors written front-compacted in descending-score order as rows (x1, y1, x2, y2, score, label).
constexpr int kPostThreads = 1024;   // 16 waves: four per SIMD, so the pairwise-IoU phase hides its VALU latencies
// one image: p = its decoded rows [A][5 + ncls], d = its det rows [A][6], n_keep_b = its survivor count.  All kPostThreads
// threads of the workgroup call it (k_postprocess: one workgroup per image; k_heads_finish: right after the decode).
__device__ __forceinline__ void postprocess_image(const float *__restrict__ p, int A, int Apad, int ncls, float conf_thr,
                                                  float nms_thr, float class_offset, float *__restrict__ d,
                                                  int32_t *__restrict__ n_keep_b) {
    __shared__ float s_key[kMaxAnchors];
    __shared__ int s_idx[kMaxAnchors];
    __shared__ float4 s_raw[kMaxAnchors];     // un-offset boxes, anchor order
    __shared__ float s_score[kMaxAnchors];
    __shared__ int s_cls[kMaxAnchors];
    __shared__ float4 s_box[kMaxAnchors];     // class-offset boxes, score order
    __shared__ int s_keep[kMaxAnchors];
    __shared__ unsigned long long s_mask[kMaskAnchors][kMaskWords];
    __shared__ float s_key2[kMaskAnchors];
    __shared__ int s_scan[kPostThreads / 64];
    const int ld = 5 + ncls;
    for (int i = threadIdx.x; i < Apad; i += kPostThreads) {
        bool ok = false;
        float sc = -INFINITY;
        if (i < A) {
            const float *r = p + (size_t)i * ld;
            const float x1 = r[0] - r[2] / 2.0f, y1 = r[1] - r[3] / 2.0f;
            s_raw[i] = make_float4(x1, y1, r[2] + x1, r[3] + y1);
            float cc = r[5];
            int cp = 0;
            for (int c = 1; c < ncls; c++)
                if (r[5 + c] > cc) { cc = r[5 + c]; cp = c; }
            sc = r[4] * cc;
            s_score[i] = sc;
            s_cls[i] = cp;
            ok = sc * cc >= conf_thr;
        }
        s_key[i] = ok ? sc : -INFINITY;
        s_idx[i] = i < A ? i : 0x7fffffff;
    }
    __syncthreads();
    sort_desc(A, Apad, s_key, s_idx, s_key2);
    for (int i = threadIdx.x; i < A; i += kPostThreads) {
        const bool ok = s_key[i] > -INFINITY;
        float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
            const int a = s_idx[i];
            const float off = (float)s_cls[a] * class_offset;
            const float4 r = s_raw[a];
            bb = make_float4(r.x + off, r.y + off, r.z + off, r.w + off);
        }
        s_box[i] = bb;
        s_keep[i] = ok ? 1 : 0;
    }
    __syncthreads();
    greedy_suppress(A, s_box, s_keep, nms_thr, s_mask);
    // front-compaction: thread t owns the 4 consecutive sorted positions 4t .. 4t+3
    int mine[4], cnt = 0;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int i = 4 * threadIdx.x + u;
        mine[u] = (i < A) ? s_keep[i] : 0;
        cnt += mine[u];
    }
    int total;
    int base = block_exclusive_scan(cnt, s_scan, total);
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int i = 4 * threadIdx.x + u;
        if (i < A && mine[u]) {
            const int a = s_idx[i];
            const float4 r = s_raw[a];
            float *row = d + (size_t)base * 6;
            row[0] = r.x; row[1] = r.y; row[2] = r.z; row[3] = r.w; row[4] = s_score[a]; row[5] = (float)s_cls[a];
            base++;
        }
    }
    if (threadIdx.x == 0) *n_keep_b = total;
}

__global__ __launch_bounds__(kPostThreads) void k_postprocess(const float *__restrict__ pred, int A, int Apad, int ncls,
                                                       float conf_thr, float nms_thr, float class_offset,
                                                       float *__restrict__ det, int32_t *__restrict__ n_keep) {
    const int b = blockIdx.x;
    postprocess_image(pred + (size_t)b * A * (5 + ncls), A, Apad, ncls, conf_thr, nms_thr, class_offset,
                      det + (size_t)b * A * 6, n_keep + b);
}
// collect_outputs + decode_outputs of the eval head (model/networks/dagr.py:283-312) for up to two scales in one launch:
// dense maps [B, 5+C, Hs, Ws] (reg 4 | obj 1 | cls C, raw logits) -> out[B, A, 5+C] with A = sum Hs*Ws, anchors of a scale
// in row-major (y, x) order; xy = (logit + grid) * stride, wh = exp(logit) * stride, obj / cls = sigmoid(logit).
__global__ __launch_bounds__(kBlock) void k_decode_heads(const float *__restrict__ d0, int H0, int W0, float s0,
                                                        const float *__restrict__ d1, int H1, int W1, float s1, int B,
                                                        int CH, float *__restrict__ out) {
    const int A0 = H0 * W0, A1 = d1 ? H1 * W1 : 0, A = A0 + A1;
    const int64_t total = (int64_t)B * A * CH;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int ch = (int)(i % CH);
        const int a = (int)((i / CH) % A);
        const int b = (int)(i / ((int64_t)CH * A));
        const bool first = a < A0;
        const int cell = first ? a : a - A0;
        const int Ws = first ? W0 : W1, HW = first ? A0 : A1;
        const float stride = first ? s0 : s1;
        const float v = (first ? d0 : d1)[((size_t)b * CH + ch) * HW + cell];
        float r;
        if (ch < 2) r = (v + (float)(ch == 0 ? cell % Ws : cell / Ws)) * stride;
        else if (ch < 4) r = expf(v) * stride;
        else r = 1.0f / (1.0f + expf(-v));
        out[i] = r;
    }
}

// to_dense of both head scales + the CNN head's logits + collect_outputs / decode_outputs in ONE launch (spline_conv.py:
// 80-107, dagr.py:219-222,230-234,283-312).  The head maps are tiny (B x 175 anchors): one workgroup keeps both winner
// tables (highest node index per cell, csrc/dense.hip) in LDS, then every thread decodes output elements straight from the
// winning node's predictor row (+ the image branch's logit) -- the dense maps exist only as an optional by-product for
// traces.  Replaces k_dense_winner + k_dense_cells per scale, the concatenation / addition of the CNN maps and
// k_decode_heads: 5 (events-only) to 9 launches at the very end of every window's dependency chain.
struct HeadScale {
    const int32_t *n_ptr; int n_max;
    const float *pred; int ld;              // [n, ld]: reg 4 | obj 1 | cls C
    const float *pos; const int32_t *batch;
    float vx, vy, stride; int Hc, Wc;
    const float *cnn[3]; int cb[3], cc[3], cy[3], cx[3];   // image-branch logits reg / obj / cls with their strides (or NULL)
    float *dense;                           // optional [B, 5 + C, Hc, Wc]: the (fused) logit maps
};

// One workgroup PER IMAGE (round 6; until then one workgroup walked all B images: 21.8 us at B = 8 against 6 at B = 1): the
// image's winner table in LDS (every workgroup scans the -- few hundred -- head nodes and keeps its own image's), its A x CH
// outputs decoded, and -- `post.det` given -- the image's post-processing (model/utils.py:61-110: confidence mask +
// class-offset NMS, postprocess_image above) right behind, on the rows the workgroup has just written: the window's last
// two launches are one, and the B images finish side by side.
struct PostArgs {
    float *det;            // [B, A, 6] or NULL (decode only)
    int32_t *n_keep;       // [B]
    int ncls, Apad;
    float conf_thr, nms_thr, class_offset;
};
__global__ __launch_bounds__(kPostThreads) void k_heads_finish(HeadScale h0, HeadScale h1, int n_scales, int B, int CH,
                                                              float *__restrict__ out, int32_t *__restrict__ status,
                                                              PostArgs post) {
    extern __shared__ int winner[];
    const int b = blockIdx.x;
    const int A0 = h0.Hc * h0.Wc, A1 = n_scales > 1 ? h1.Hc * h1.Wc : 0, A = A0 + A1;
    for (int i = threadIdx.x; i < A; i += blockDim.x) winner[i] = -1;
    __syncthreads();
    for (int s = 0; s < n_scales; s++) {
        const HeadScale &h = s == 0 ? h0 : h1;
        int32_t *w = winner + (s == 0 ? 0 : A0);
        const int n_nodes = h.n_ptr ? min(*h.n_ptr, h.n_max) : h.n_max;
        for (int n = threadIdx.x; n < n_nodes; n += blockDim.x) {
            const int cx = (int)(h.pos[3 * n] / h.vx), cy = (int)(h.pos[3 * n + 1] / h.vy), nb = h.batch[n];
            if (cx < 0 || cx >= h.Wc || cy < 0 || cy >= h.Hc || nb < 0 || nb >= B) {
                if (b == 0) atomicOr(status, 1);
                continue;
            }
            if (nb == b) atomicMax(&w[cy * h.Wc + cx], n);
        }
    }
    __syncthreads();
    float *out_b = out + (size_t)b * A * CH;
    for (int i = threadIdx.x; i < A * CH; i += blockDim.x) {
        const int ch = i % CH, a = i / CH;
        const bool first = a < A0;
        const HeadScale &h = first ? h0 : h1;
        const int cell = first ? a : a - A0;
        const int HW = first ? A0 : A1;
        const int n = winner[a];
        float v = n >= 0 ? h.pred[(size_t)n * h.ld + ch] : 0.0f;
        const int k = ch < 4 ? 0 : (ch == 4 ? 1 : 2);
        if (h.cnn[k]) {
            const int cc_ = ch < 4 ? ch : (ch == 4 ? 0 : ch - 5);
            const int y = cell / h.Wc, x = cell - y * h.Wc;
            v = v + h.cnn[k][(size_t)b * h.cb[k] + (size_t)cc_ * h.cc[k] + (size_t)y * h.cy[k] + (size_t)x * h.cx[k]];
        }
        if (h.dense) h.dense[((size_t)b * CH + ch) * HW + cell] = v;
        float r;
        if (ch < 2) r = (v + (float)(ch == 0 ? cell % h.Wc : cell / h.Wc)) * h.stride;
        else if (ch < 4) r = expf(v) * h.stride;
        else r = 1.0f / (1.0f + expf(-v));
        out_b[i] = r;
    }
    if (post.det) {
        __syncthreads();         // this workgroup wrote the rows it now reads
        postprocess_image(out_b, A, post.Apad, post.ncls, post.conf_thr, post.nms_thr, post.class_offset,
                          post.det + (size_t)b * A * 6, post.n_keep + b);
    }
}
}  // namespace
}  // namespace dagr

using namespace dagr;

static int launch_heads_finish(const dagr_head_scale *scale0, const dagr_head_scale *scale1, int32_t batch_size,
                               int32_t channels, float *out, int32_t *status, const PostArgs &post_in, void *stream) {
    DAGR_CHECK_ARG(scale0 && out && status && batch_size > 0 && channels >= 5, "bad arguments");
    HeadScale h[2] = {};
    const dagr_head_scale *in[2] = {scale0, scale1};
    int cells = 0;
    for (int s = 0; s < (scale1 ? 2 : 1); s++) {
        const dagr_head_scale &d = *in[s];
        DAGR_CHECK_ARG(d.Hc > 0 && d.Wc > 0 && d.n_max >= 0 && d.ld >= channels && d.vx > 0 && d.vy > 0, "bad head scale");
        DAGR_CHECK_ARG(d.n_max == 0 || (d.pred && d.pos && d.batch), "NULL head input");
        h[s].n_ptr = d.n_ptr; h[s].n_max = d.n_max; h[s].pred = d.pred; h[s].ld = d.ld; h[s].pos = d.pos; h[s].batch = d.batch;
        h[s].vx = d.vx; h[s].vy = d.vy; h[s].stride = d.stride; h[s].Hc = d.Hc; h[s].Wc = d.Wc; h[s].dense = d.dense;
        for (int k = 0; k < 3; k++) {
            h[s].cnn[k] = d.cnn[k];
            h[s].cb[k] = d.cnn_stride[k][0]; h[s].cc[k] = d.cnn_stride[k][1];
            h[s].cy[k] = d.cnn_stride[k][2]; h[s].cx[k] = d.cnn_stride[k][3];
        }
        cells += d.Hc * d.Wc;
    }
    const size_t lds = (size_t)cells * 4;
    DAGR_CHECK_ARG(lds <= 16 * 1024, "head maps too large for the per-image finish kernel");
    PostArgs post = post_in;
    if (post.det) {
        DAGR_CHECK_ARG(cells <= kMaxAnchors && post.n_keep && post.ncls == channels - 5, "bad post-processing arguments");
        post.Apad = 1;
        while (post.Apad < cells) post.Apad <<= 1;
    }
    k_heads_finish<<<(unsigned)batch_size, kPostThreads, lds, (hipStream_t)stream>>>(h[0], h[1], scale1 ? 2 : 1, batch_size,
                                                                                    channels, out, status, post);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

extern "C" int dagr_heads_finish(const dagr_head_scale *scale0, const dagr_head_scale *scale1, int32_t batch_size,
                                 int32_t channels, float *out, int32_t *status, void *stream) {
    return launch_heads_finish(scale0, scale1, batch_size, channels, out, status, PostArgs{}, stream);
}

extern "C" int dagr_heads_finish_detect(const dagr_head_scale *scale0, const dagr_head_scale *scale1, int32_t batch_size,
                                        int32_t channels, float *out, int32_t *status, float conf_threshold,
                                        float iou_threshold, float class_offset, float *det, int32_t *n_keep, void *stream) {
    DAGR_CHECK_ARG(det && n_keep, "NULL detection buffers");
    PostArgs post{};
    post.det = det; post.n_keep = n_keep; post.ncls = channels - 5;
    post.conf_thr = conf_threshold; post.nms_thr = iou_threshold; post.class_offset = class_offset;
    return launch_heads_finish(scale0, scale1, batch_size, channels, out, status, post, stream);
}

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

extern "C" int dagr_postprocess(const float *pred, int32_t B, int32_t A, int32_t num_classes, float conf_threshold,
                                float iou_threshold, float class_offset, float *det, int32_t *n_keep, void *stream) {
    DAGR_CHECK_ARG(B >= 0 && A >= 0 && A <= kMaxAnchors && num_classes >= 1, "A must be <= 1024, num_classes >= 1");
    if (B == 0) return DAGR_OK;
    DAGR_CHECK_ARG(pred && det && n_keep, "NULL pointer");
    if (A == 0) {
        DAGR_CHECK_HIP(hipMemsetAsync(n_keep, 0, (size_t)B * 4, (hipStream_t)stream));
        return DAGR_OK;
    }
    int Apad = 1;
    while (Apad < A) Apad <<= 1;
    k_postprocess<<<B, kPostThreads, 0, (hipStream_t)stream>>>(pred, A, Apad, num_classes, conf_threshold, iou_threshold,
                                                         class_offset, det, n_keep);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

extern "C" int dagr_decode_heads(const float *dense0, int32_t H0, int32_t W0, float stride0, const float *dense1,
                                 int32_t H1, int32_t W1, float stride1, int32_t B, int32_t channels, float *out,
                                 void *stream) {
    DAGR_CHECK_ARG(B >= 0 && channels >= 5 && H0 > 0 && W0 > 0 && (!dense1 || (H1 > 0 && W1 > 0)), "bad sizes");
    if (B == 0) return DAGR_OK;
    DAGR_CHECK_ARG(dense0 && out, "NULL pointer");
    const int64_t total = (int64_t)B * (H0 * W0 + (dense1 ? H1 * W1 : 0)) * channels;
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(total, kBlock), 1024);
    k_decode_heads<<<grid, kBlock, 0, (hipStream_t)stream>>>(dense0, H0, W0, stride0, dense1, H1, W1, stride1, B, channels,
                                                            out);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}
