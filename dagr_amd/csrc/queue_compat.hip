// queue_compat.hip -- 1:1 replacements of the reference's native entry points
//   ev_graph_cuda.fill_edges_cuda / insert_in_queue_cuda / insert_in_queue_single_cuda
//   (src/dagr/graph/ev_graph.cu:82-128, 241-276, 215-238), operating on the CALLER's state exactly as the
// reference does: event_queue int32[B,Q,H,W] (-1 = empty, q = 0 newest), all_timestamps, a -1-filled
// int64 edge buffer.  Because the FIFO volume persists across calls, this path also covers the
// incremental (reset=False, min_index > 0) use of AsyncGraph / SlidingWindowGraph; the window engine
// (graph_build.hip) is the fast path for reset=True windows.
//
// Parallelisation differs from the reference (same results, bit-exact):
//   * fill: 16 lanes per event, one spiral position per lane and round; each lane walks its FIFO column
//     (stop at the first id < min_index, skip ids >= own, skip dt > delta), a 16-lane prefix sum reproduces
//     the sequential "first max_num_neighbors in spiral order" cut;
//   * insert: one thread per active pixel as in the reference (the column rewrite is inherently serial
//     per pixel), reading the shifted entries before overwriting them.
#include "common.hpp"

namespace dagr {
namespace {

__host__ __device__ inline void spiral_offset_c(int s, int &sx, int &sy) {
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

__device__ __forceinline__ int scan16(int v) {
    const int l = threadIdx.x & 15;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        int n = __shfl_up(v, d, 16);
        if (l >= d) v += n;
    }
    return v;
}

__global__ __launch_bounds__(kBlock) void k_fill_edges(const int32_t *__restrict__ batch, const int32_t *__restrict__ pos,
                                                      const int32_t *__restrict__ all_timestamps,
                                                      const int32_t *__restrict__ indices,
                                                      const int32_t *__restrict__ queue, int64_t *__restrict__ edges,
                                                      int Q, int H, int W, int N, int64_t Kstride, int r,
                                                      float delta_t, int max_nb, int min_index) {
    const int l = threadIdx.x & 15;
    const int e = (blockIdx.x * kBlock + threadIdx.x) >> 4;
    if (e >= N) return;
    const int side = 2 * r + 1, S = side * side;
    const int b = batch[e], x = pos[3 * (size_t)e], y = pos[3 * (size_t)e + 1], ts = pos[3 * (size_t)e + 2];
    const int own = indices[e];
    const int64_t off = (int64_t)e * max_nb;
    int total = 1;
    if (l == 0) {  // self edge first (ev_graph.cu:44-46)
        edges[off] = own - min_index;
        edges[Kstride + off] = own - min_index;
    }
    const int64_t HW = (int64_t)H * W;
    for (int s0 = 0; s0 < S && total < max_nb; s0 += 16) {
        const int s = s0 + l;
        int v = 0;
        const int32_t *colp = nullptr;
        if (s < S) {
            int sx, sy;
            spiral_offset_c(s, sx, sy);
            const int xn = x + sx, yn = y + sy;
            if (xn >= 0 && yn >= 0 && xn < W && yn < H) {
                colp = queue + xn + (int64_t)W * yn + HW * Q * b;
                for (int q = 0; q < Q && v < max_nb; q++) {
                    const int idx = colp[HW * q];
                    if (idx < min_index) break;                                   // ev_graph.cu:62
                    if (own > idx) {                                              // :64
                        const int dt = ts - all_timestamps[idx - min_index];
                        if ((float)dt > delta_t) continue;                        // :69
                        v++;
                    }
                }
            }
        }
        const int incl = scan16(v);
        int slot = total + incl - v;
        total += __shfl(incl, 15, 16);
        if (v > 0 && slot < max_nb) {
            for (int q = 0; q < Q && slot < max_nb; q++) {
                const int idx = colp[HW * q];
                if (idx < min_index) break;
                if (own > idx) {
                    const int dt = ts - all_timestamps[idx - min_index];
                    if ((float)dt > delta_t) continue;
                    edges[off + slot] = idx - min_index;
                    edges[Kstride + off + slot] = own - min_index;
                    slot++;
                }
            }
        }
    }
}

// counts == nullptr: single-event variant (counts = 1, offset = 0, b = 0, x = events[0], y = events[1])
__global__ __launch_bounds__(kBlock) void k_insert_in_queue(const int32_t *__restrict__ indices,
                                                           const int32_t *__restrict__ unique_coords,
                                                           const int32_t *__restrict__ cumsum_counts,
                                                           const int32_t *__restrict__ single_event,
                                                           int32_t *__restrict__ queue, int Q, int H, int W, int Kpix) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= Kpix) return;
    int counts, offset, x, y, b;
    if (single_event) {
        counts = 1; offset = 0; x = single_event[0]; y = single_event[1]; b = 0;   // ev_graph.cu:146-152
    } else {
        offset = i > 0 ? cumsum_counts[i - 1] : 0;
        counts = cumsum_counts[i] - offset;
        const int c = unique_coords[i];
        x = c % W;
        y = ((c - x) / W) % H;
        b = c / (W * H);
    }
    int32_t *col = queue + (int64_t)b * H * W * Q + (int64_t)y * W + x;
    const int64_t HW = (int64_t)H * W;
    for (int q = Q - 1; q >= 0; q--) {   // ev_graph.cu:201-211: reads q - counts (< q) before it is overwritten
        col[HW * q] = (q >= counts) ? col[HW * (q - counts)] : indices[offset + counts - 1 - q];
    }
}

}  // namespace
}  // namespace dagr

using namespace dagr;

extern "C" {

int dagr_fill_edges(const int32_t *batch, const int32_t *pos, const int32_t *all_timestamps, const int32_t *event_queue,
                    const int32_t *indices, int32_t max_num_neighbors, float radius, float delta_t_us, int64_t *edges,
                    int64_t edges_cols, int32_t min_index, int64_t N, int32_t B, int32_t Q, int32_t H, int32_t W,
                    void *stream) {
    DAGR_CHECK_ARG(N >= 0 && B > 0 && Q > 0 && H > 0 && W > 0 && max_num_neighbors >= 1, "bad sizes");
    if (N == 0) return DAGR_OK;
    DAGR_CHECK_ARG(batch && pos && all_timestamps && event_queue && indices && edges, "NULL pointer");
    DAGR_CHECK_ARG(edges_cols >= N * max_num_neighbors, "edge buffer too small");
    k_fill_edges<<<(unsigned)ceil_div(N * 16, kBlock), kBlock, 0, (hipStream_t)stream>>>(
        batch, pos, all_timestamps, indices, event_queue, edges, Q, H, W, (int)N, edges_cols, (int)radius, delta_t_us,
        max_num_neighbors, min_index);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

int dagr_insert_in_queue(const int32_t *indices, const int32_t *unique_coords, const int32_t *cumsum_counts,
                         int64_t num_pixels, int32_t *queue, int32_t B, int32_t Q, int32_t H, int32_t W, void *stream) {
    DAGR_CHECK_ARG(num_pixels >= 0 && B > 0 && Q > 0 && H > 0 && W > 0, "bad sizes");
    if (num_pixels == 0) return DAGR_OK;
    DAGR_CHECK_ARG(indices && unique_coords && cumsum_counts && queue, "NULL pointer");
    k_insert_in_queue<<<(unsigned)ceil_div(num_pixels, kBlock), kBlock, 0, (hipStream_t)stream>>>(
        indices, unique_coords, cumsum_counts, nullptr, queue, Q, H, W, (int)num_pixels);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

int dagr_insert_in_queue_single(const int32_t *indices, const int32_t *event_xy, int32_t *queue, int32_t B, int32_t Q,
                                int32_t H, int32_t W, void *stream) {
    DAGR_CHECK_ARG(B > 0 && Q > 0 && H > 0 && W > 0, "bad sizes");
    DAGR_CHECK_ARG(indices && event_xy && queue, "NULL pointer");
    k_insert_in_queue<<<1, kBlock, 0, (hipStream_t)stream>>>(indices, nullptr, nullptr, event_xy, queue, Q, H, W, 1);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

}  // extern "C"
