// async_update.hip -- incremental (reset=False) insertion of a micro-batch of events into a resident window.
//
// Reference: EV_TGN.forward with reset=False (src/dagr/model/layers/ev_tgn.py:45-56) -> AsyncGraph.forward
// (graph/ev_graph.py:63-103): the new events are pushed into the persistent B x Q x H x W FIFO volume
// (insert_in_queue_cuda_kernel, ev_graph.cu:169-212) and fill_edges_cuda_kernel (:15-80) searches their in-edges with
// min_index = number of events already in the graph.  Edges point from older to newer events, so the rows of the
// events already in the window do not change: an update only APPENDS level-0 rows.
//
// Here there is no FIFO volume.  The window's events stay where dagr_graph_build_window left them -- the index keyed
// (sample, y, x), newest last inside a pixel's segment -- and the events appended since hang off per-pixel chains,
// newest first:
//   app_head[p]  newest appended event of pixel p (-1: none)           int32[B*H*W]
//   app_next[k]  next older appended event of the same pixel           int32[capacity], k = id - n_static
//   app_xytb[k]  {x, y, t, b} of appended event k                      int32[capacity][4]
// The FIFO column of pixel p, newest first, is then: its chain, followed by its segment read backwards, cut at depth Q -- which is all the reference's walk looks at (ev_graph.cu:58-76).
//   k_async_insert  one workgroup per chunk of <= 1024 new events: denormalise (ev_tgn.py:11-16), link every event to
//                   the previous new event of its pixel (or the old head), publish the new heads.  No atomics: the
//                   order inside a pixel is the event order, as the reference's stable sort gives it.
//   k_async_fill    16 lanes per new event, one spiral position per lane and round (spiral.h): every lane counts the
//                   admissible entries of its column (skip ids >= own, :64; skip dt > delta, :69; stop at depth Q), a
//                   16-lane prefix sum reproduces the sequential "first K in spiral order" cut, then the entries are
//                   written as the new event's row of the engine's neighbour lists (source node, offset code).
// Node rows: a window event's node is its CSR slot, an appended event's node is its event id (rows continue after the
// window's).
#include "common.hpp"

namespace dagr {
// (the builder's index: PixelIndex / graph_ws_index, common.hpp)

namespace {

constexpr int kChunk = 1024;

__host__ __device__ inline void spiral_offset_a(int s, int &sx, int &sy) {
    sx = 0; sy = 0;
    if (s <= 0) return;
    int rho = 1;
    while ((2 * rho + 1) * (2 * rho + 1) <= s) rho++;
    const int k = s - (2 * rho - 1) * (2 * rho - 1);
    if (k < 2 * rho) { sx = rho; sy = -rho + 1 + k; }
    else if (k < 4 * rho) { sx = rho - 1 - (k - 2 * rho); sy = rho; }
    else if (k < 6 * rho) { sx = -rho; sy = rho - 1 - (k - 4 * rho); }
    else { sx = -rho + 1 + (k - 6 * rho); sy = -rho; }
}

__device__ __forceinline__ int scan16a(int v) {
    const int l = threadIdx.x & 15;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        int n = __shfl_up(v, d, 16);
        if (l >= d) v += n;
    }
    return v;
}

template <typename BatchT, bool kIntPos>
__global__ __launch_bounds__(kChunk) void k_async_insert(const void *__restrict__ pos_, const BatchT *__restrict__ batch,
                                                        int n, int first_id, int n_static, int W, int H, int B,
                                                        float fW, float fH, float fT, int32_t *__restrict__ app_head,
                                                        int32_t *__restrict__ app_next, int4 *__restrict__ app_xytb,
                                                        int32_t *__restrict__ status,
                                                        // level-0 inputs of the new rows (row = event id), NULL = not wanted
                                                        const float *__restrict__ feat, float *__restrict__ pos_n,
                                                        int32_t *__restrict__ batch_n, int32_t *__restrict__ batch_ev,
                                                        float *__restrict__ x0, int ldx0, int col_feat, int col_pos) {
    __shared__ int key[kChunk];
    const int i = threadIdx.x;
    int p = -1, x = 0, y = 0, t = 0, b = 0;
    if (i < n) {
        if (kIntPos) {
            const int32_t *pos = static_cast<const int32_t *>(pos_);
            x = pos[3 * (size_t)i]; y = pos[3 * (size_t)i + 1]; t = pos[3 * (size_t)i + 2];
        } else {     // int(pos * [W,H,T] + 1e-3), separately rounded (this library is built with -ffp-contract=off)
            const float *pos = static_cast<const float *>(pos_);
            x = (int)(fW * pos[3 * (size_t)i] + 1e-3f);
            y = (int)(fH * pos[3 * (size_t)i + 1] + 1e-3f);
            t = (int)(fT * pos[3 * (size_t)i + 2] + 1e-3f);
        }
        b = (int)batch[i];
        if (x < 0 || x >= W || y < 0 || y >= H || b < 0 || b >= B) atomicOr(&status[0], 1);   // outside the sensor: dropped
        else p = x + W * (y + H * b);
        app_xytb[first_id - n_static + i] = make_int4(x, y, t, b);
        if (!kIntPos && x0) {        // cf. dagr_graph_gather_inputs: pos, sample index, [polarity | pos_xy] columns
            const float *pos = static_cast<const float *>(pos_);
            const size_t row = (size_t)first_id + i;
            const float px = pos[3 * (size_t)i], py = pos[3 * (size_t)i + 1];
            pos_n[3 * row] = px; pos_n[3 * row + 1] = py; pos_n[3 * row + 2] = pos[3 * (size_t)i + 2];
            batch_n[row] = b;
            batch_ev[row] = b;
            float *xr = x0 + row * ldx0;
            xr[col_feat] = feat[i];
            xr[col_pos] = px;
            xr[col_pos + 1] = py;
        }
    }
    key[i] = p;
    __syncthreads();
    int pred = -1;
    bool last = true;
    if (p >= 0) {
        for (int j = 0; j < n; j++) {
            if (key[j] == p) {
                if (j < i) pred = j;
                if (j > i) last = false;
            }
        }
    }
    // the first new event of a pixel continues the old chain; all old heads are read before any new head is published
    int nxt = -1;
    if (p >= 0) nxt = pred >= 0 ? first_id + pred : app_head[p];
    __syncthreads();
    if (i < n) app_next[first_id - n_static + i] = nxt;
    if (p >= 0 && last) app_head[p] = first_id + i;
}

// The window events of pixel (xn, yb), newest (largest id) first: fn(slot) until it returns false.  A pixel's events are
// one segment of the index with ids ascending along its slots -- sorted timestamps or not -- so the order by id is the
// segment read backwards.
template <typename Fn>
__device__ __forceinline__ void walk_window_pixel(const PixelIndex &ix, int xn, int yb, Fn fn) {
    const int key = ix.segment(xn, yb);
    const int a0 = ix.start[key], a1 = ix.start[key + 1];
    for (int k = a1 - 1; k >= a0; k--)
        if (!fn(k)) return;
}

__global__ __launch_bounds__(kBlock) void k_async_fill(int n, int first_id, int n_static, int W, int H, int B, int K, int Q, int r,
                                                      float delta_t, const PixelIndex ix,
                                                      const int32_t *__restrict__ app_head,
                                                      const int32_t *__restrict__ app_next,
                                                      const int4 *__restrict__ app_xytb, int32_t *__restrict__ nbr_src,
                                                      int16_t *__restrict__ nbr_code, int32_t *__restrict__ deg) {
    const int l = threadIdx.x & 15;
    const int i = (blockIdx.x * kBlock + threadIdx.x) >> 4;
    if (i >= n) return;
    const int2 *__restrict__ slot_it = ix.slot_it;
    const int own = first_id + i;
    const int4 me = app_xytb[own - n_static];
    const int x = me.x, y = me.y, ts = me.z, b = me.w;
    const int side = 2 * r + 1, S = side * side;
    const int64_t row = (int64_t)own * K;
    int total = 1;
    if (l == 0) {       // self loop first (ev_graph.cu:44-46)
        nbr_src[row] = own;
        nbr_code[row] = (int16_t)(r * side + r);
    }
    // an event outside the sensor or the batch range (flagged by k_async_insert) keeps only its self loop
    const bool inside = x >= 0 && x < W && y >= 0 && y < H && b >= 0 && b < B;
    for (int s0 = 0; s0 < S && total < K && inside; s0 += 16) {
        const int s = s0 + l;
        int v = 0, head = -1, xn = 0, yb = 0, code = 0;
        bool probe = false;
        if (s < S) {
            int sx, sy;
            spiral_offset_a(s, sx, sy);
            xn = x + sx;
            const int yn = y + sy;
            code = (sx + r) * side + (sy + r);
            if (xn >= 0 && yn >= 0 && xn < W && yn < H) {          // out of FOV: skip this pixel only
                yb = yn + H * b;
                head = app_head[xn + W * yb];
                probe = true;
            }
        }
        // pass 1: count the admissible entries of the column, newest first, depth Q (ev_graph.cu:58-76)
        if (probe) {
            int depth = 0;
            for (int a = head; a >= 0 && depth < Q && v < K; a = app_next[a - n_static], depth++) {
                if (own > a) {                                                    // :64
                    if ((float)(ts - app_xytb[a - n_static].z) > delta_t) continue;    // :69
                    v++;
                }
            }
            if (depth < Q && v < K)
                walk_window_pixel(ix, xn, yb, [&](int k) {
                    if (!((float)(ts - slot_it[k].y) > delta_t)) v++;             // window events: always older than own
                    depth++;
                    return depth < Q && v < K;
                });
        }
        const int incl = scan16a(v);
        int slot = total + incl - v;
        total += __shfl(incl, 15, 16);
        if (v > 0 && slot < K) {
            int depth = 0;
            for (int a = head; a >= 0 && depth < Q && slot < K; a = app_next[a - n_static], depth++) {
                if (own > a) {
                    if ((float)(ts - app_xytb[a - n_static].z) > delta_t) continue;
                    nbr_src[row + slot] = a;                   // an appended event's node is its id
                    nbr_code[row + slot] = (int16_t)code;
                    slot++;
                }
            }
            if (depth < Q && slot < K)
                walk_window_pixel(ix, xn, yb, [&](int k) {
                    if (!((float)(ts - slot_it[k].y) > delta_t)) {
                        nbr_src[row + slot] = k;               // a window event's node is its CSR slot
                        nbr_code[row + slot] = (int16_t)code;
                        slot++;
                    }
                    depth++;
                    return depth < Q && slot < K;
                });
        }
    }
    if (l == 0) deg[own] = min(total, K);
}

}  // namespace
}  // namespace dagr

using namespace dagr;

extern "C" {

int dagr_async_graph_append(const dagr_graph_desc *desc, void *graph_ws, int64_t n_static, int64_t first_id,
                            int32_t *app_head, int32_t *app_next, int32_t *app_xytb, int64_t capacity,
                            const void *pos, int32_t pos_is_int32, const void *batch, int32_t batch_is_int64,
                            int64_t n_new, int32_t *nbr_src, int16_t *nbr_code, int32_t *deg, int32_t *status,
                            const float *feat, float *pos_nodes, int32_t *batch_nodes, int32_t *batch_events, float *x0,
                            int32_t ldx0, int32_t col_feat, int32_t col_pos, void *stream_) {
    DAGR_CHECK_ARG(desc && graph_ws, "NULL desc / workspace");
    DAGR_CHECK_ARG(n_new >= 0 && n_static >= 0 && first_id >= n_static, "bad event ranges");
    if (n_new == 0) return DAGR_OK;
    DAGR_CHECK_ARG(first_id - n_static + n_new <= capacity, "the appended-event arrays are full");
    DAGR_CHECK_ARG(app_head && app_next && app_xytb && pos && batch && nbr_src && nbr_code && deg && status, "NULL pointer");
    DAGR_CHECK_ARG(desc->max_neighbors <= 16 && desc->radius <= 31, "max_neighbors <= 16 expected");
    if (x0) {
        DAGR_CHECK_ARG(!pos_is_int32 && feat && pos_nodes && batch_nodes && batch_events && ldx0 >= col_pos + 2 &&
                           col_pos >= 0 && col_feat >= 0 && col_feat < ldx0 && col_feat != col_pos && col_feat != col_pos + 1,
                       "level-0 input rows: normalised fp32 pos and all row arrays are needed");
    }
    hipStream_t stream = (hipStream_t)stream_;
    PixelIndex ix;
    graph_ws_index(desc, graph_ws, &ix);
    const int W = desc->width, H = desc->height, B = desc->batch_size;
    for (int64_t c0 = 0; c0 < n_new; c0 += kChunk) {        // chunks in event order: a chunk's heads are in place before the next
        const int n = (int)std::min<int64_t>(kChunk, n_new - c0);
        const int fid = (int)(first_id + c0);
#define DAGR_INS(BT, IP)                                                                                               \
    k_async_insert<BT, IP><<<1, kChunk, 0, stream>>>(                                                                  \
        (const char *)pos + (size_t)c0 * 12, (const BT *)batch + c0, n, fid, (int)n_static, W, H, B, (float)W, (float)H, \
        (float)desc->time_window, app_head, app_next, (int4 *)app_xytb, status, feat ? feat + c0 : nullptr, pos_nodes,     \
        batch_nodes, batch_events, x0, ldx0, col_feat, col_pos)
        if (batch_is_int64) { if (pos_is_int32) DAGR_INS(int64_t, true); else DAGR_INS(int64_t, false); }
        else                { if (pos_is_int32) DAGR_INS(int32_t, true); else DAGR_INS(int32_t, false); }
#undef DAGR_INS
        DAGR_CHECK_LAUNCH();
    }
    // the reference pushes the whole micro-batch into the queue before it searches (ev_graph.py:84-93): all chains first
    k_async_fill<<<(unsigned)ceil_div(n_new * 16, kBlock), kBlock, 0, stream>>>(
        (int)n_new, (int)first_id, (int)n_static, W, H, desc->batch_size, desc->max_neighbors, desc->queue_size, desc->radius,
        (float)desc->delta_t_us, ix, app_head, app_next, (const int4 *)app_xytb, nbr_src, nbr_code, deg);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

int dagr_async_update(const dagr_async_update_args *a, void *stream) {
    DAGR_CHECK_ARG(a && a->gdesc && a->pdesc, "NULL arguments");
    DAGR_CHECK_ARG(a->cin1 >= 1 && a->cin1 <= 4, "events-only input rows expected (polarity | pos_xy)");
    int rc = DAGR_OK;
    if (a->n_new > 0) {
        rc = dagr_async_graph_append(a->gdesc, a->graph_ws, a->n_static, a->first_id, a->app_head, a->app_next, a->app_xytb,
                                     a->capacity, a->pos, 0, a->batch, a->batch_is_int64, a->n_new, a->nbr_src, a->nbr_code,
                                     a->deg, a->status, a->feat, a->pos_nodes, a->batch_nodes, a->batch_events, a->x0, a->ldx0,
                                     a->col_feat, a->col_pos, stream);
        if (rc != DAGR_OK) return rc;
        const int K = a->gdesc->max_neighbors;
        rc = dagr_spline_conv_l0_tiles_rows(0, a->cin1, 0, a->win_x, a->tx, a->win_y, a->ty, a->rx, a->ry, a->den_x, a->den_y,
                                            a->first_id, a->n_new, K, a->nbr_src, a->nbr_code, a->deg, a->x0, a->ldx0, nullptr,
                                            0, a->w1, a->s1, 1, a->h1, a->ldh1, nullptr, stream);
        if (rc != DAGR_OK) return rc;
        rc = dagr_spline_conv_l0_tiles_rows(16, 0, a->cin1, a->win_x, a->tx, a->win_y, a->ty, a->rx, a->ry, a->den_x, a->den_y,
                                            a->first_id, a->n_new, K, a->nbr_src, a->nbr_code, a->deg, a->h1, a->ldh1, a->x0,
                                            a->ldx0, a->w2, a->s2, 1, a->hp0, a->ldhp0, nullptr, stream);
        if (rc != DAGR_OK) return rc;
    }
    return dagr_pool_l0_stream(a->pdesc, a->pool_ws, 0, a->gdesc, a->graph_ws, a->xlo, a->ylo, a->hp0, a->ldhp0, a->pos_nodes,
                               a->batch_events, a->n_static, a->first_id, a->n_new, a->nbr_src, a->nbr_code, a->deg, a->x_out,
                               a->ldo, 0, a->pos_out, a->batch_out, a->n_out, a->rowptr_out, a->col_out, a->code_out, a->e_out,
                               a->e_cap, stream);
}

}  // extern "C"
