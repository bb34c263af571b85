/*
 * oracle/graph_oracle.c -- TEST INFRASTRUCTURE ONLY (never imported by dagr_amd/).
 *
 * Plain-C, single-threaded CPU restatement of the reference's event-graph
 * builder kernels.  Each "GPU thread" of the reference becomes one iteration
 * of a serial loop here; nothing is restructured so that the result is the
 * definition of "bit-exact" for the HIP path.
 *
 * Follows (reference paths relative to /root/reference/src/dagr/graph/):
 *   oracle_spiral_next        <- spiral.h:1-15            (SpiralOut::goNext)
 *   oracle_insert_in_queue    <- ev_graph.cu:169-212      (insert_in_queue_cuda_kernel)
 *   oracle_insert_in_queue_single <- ev_graph.cu:130-166  (single-event variant, b fixed to 0)
 *   oracle_fill_edges         <- ev_graph.cu:15-80        (fill_edges_cuda_kernel)
 *
 * Parity pin: checked on a real MI355X against the reference's own
 * ev_graph.cu compiled from /root/reference by oracle/Makefile (oracle/_ref);
 * the outputs of that run are committed as tests/golden/graph_ref_*.npz.
 */
#include <stdint.h>
#include <math.h>

typedef struct { unsigned layer, leg; int x, y; } spiral_t;

static void spiral_init(spiral_t *s) { s->layer = 1; s->leg = 0; s->x = 0; s->y = 0; }

/* spiral.h:8-15 */
static void spiral_next(spiral_t *s)
{
    switch (s->leg) {
    case 0: ++s->x; if (s->x == (int)s->layer) ++s->leg; break;
    case 1: ++s->y; if (s->y == (int)s->layer) ++s->leg; break;
    case 2: --s->x; if (-s->x == (int)s->layer) ++s->leg; break;
    case 3: --s->y; if (-s->y == (int)s->layer) { s->leg = 0; ++s->layer; } break;
    }
}

/* Export the spiral sequence so tests can pin the visiting order (QUIRK-4). */
void oracle_spiral_offsets(int n, int32_t *dx, int32_t *dy)
{
    spiral_t s; spiral_init(&s);
    for (int i = 0; i < n; i++) { dx[i] = s.x; dy[i] = s.y; spiral_next(&s); }
}

/* ev_graph.cu:169-212 -- one "thread" per unique active pixel.
 * indices: event ids sorted (stably) by linear pixel; unique_coords[K];
 * cumsum_counts[K] inclusive; queue[B,Q,H,W] modified in place. */
void oracle_insert_in_queue(const int32_t *indices, const int32_t *unique_coords,
                            const int32_t *cumsum_counts, int32_t *queue,
                            int B, int Q, int H, int W, int K)
{
    (void)B;
    for (int lin_idx = 0; lin_idx < K; lin_idx++) {
        int counts, offset;
        if (lin_idx > 0) {
            offset = cumsum_counts[lin_idx - 1];
            counts = cumsum_counts[lin_idx] - offset;
        } else {
            offset = 0;
            counts = cumsum_counts[lin_idx];
        }
        int x = unique_coords[lin_idx] % W;
        int y = ((unique_coords[lin_idx] - x) / W) % H;
        int b = unique_coords[lin_idx] / (W * H);
        for (int q = Q - 1; q >= 0; q--) {
            int index = b * H * W * Q + q * H * W + y * W + x;
            if (q >= counts) {
                int shifted = b * H * W * Q + (q - counts) * H * W + y * W + x;
                queue[index] = queue[shifted];
            } else {
                queue[index] = indices[offset + counts - 1 - q];
            }
        }
    }
}

/* ev_graph.cu:130-166 -- len(batch)==1 path: counts=1, offset=0, b=0,
 * x=events[0], y=events[1] (QUIRK-5). */
void oracle_insert_in_queue_single(const int32_t *indices, const int32_t *events,
                                   int32_t *queue, int B, int Q, int H, int W)
{
    (void)B;
    int counts = 1, offset = 0;
    int x = events[0], y = events[1], b = 0;
    for (int q = Q - 1; q >= 0; q--) {
        int index = b * H * W * Q + q * H * W + y * W + x;
        if (q >= counts) {
            int shifted = b * H * W * Q + (q - counts) * H * W + y * W + x;
            queue[index] = queue[shifted];
        } else {
            queue[index] = indices[offset + counts - 1 - q];
        }
    }
}

/* ev_graph.cu:15-80 -- one "thread" per event.  edges is int64[2,K], K =
 * edges.size(1) (>= N*max_num_neighbors); pre-filled with -1 by the caller
 * (ev_graph.py:89). */
void oracle_fill_edges(const int32_t *batch, const int32_t *pos,
                       const int32_t *all_timestamps, const int32_t *indices,
                       const int32_t *event_queue, int64_t *edges,
                       int B, int Q, int H, int W, int N, int64_t K,
                       float radius, float delta_t_us, int max_num_neighbors,
                       int min_index)
{
    (void)B;
    for (int event_idx = 0; event_idx < N; event_idx++) {
        int radius_int = (int)radius;
        int num_neighbors = 0;
        int64_t offset = (int64_t)event_idx * max_num_neighbors;

        int b = batch[event_idx];
        int x = pos[3 * event_idx + 0];
        int y = pos[3 * event_idx + 1];
        int ts_event = pos[3 * event_idx + 2];

        /* self edge first (ev_graph.cu:44-46) */
        edges[offset + num_neighbors + K * 0] = indices[event_idx] - min_index;
        edges[offset + num_neighbors + K * 1] = indices[event_idx] - min_index;
        num_neighbors++;

        spiral_t sp; spiral_init(&sp);
        double npix = pow((double)(2 * radius_int + 1), 2.0);
        for (int i = 0; i < npix; i++) {
            if (num_neighbors >= max_num_neighbors) break;
            for (int q = 0; q < Q; q++) {
                int xn = x + sp.x;
                int yn = y + sp.y;
                if (!((xn >= 0) && (yn >= 0) && (xn < W) && (yn < H))) break;

                int64_t queue_idx = xn + (int64_t)W * yn + (int64_t)H * W * q + (int64_t)H * W * Q * b;
                int idx = event_queue[queue_idx];

                if (idx < min_index) break;

                if (indices[event_idx] > idx) {
                    int32_t ts_neighbor = all_timestamps[idx - min_index];
                    int32_t dt_us = ts_event - ts_neighbor;
                    /* int32 -> float promotion, as in the reference (ev_graph.cu:69) */
                    if ((float)dt_us > delta_t_us) continue;

                    edges[offset + num_neighbors + K * 0] = idx - min_index;
                    edges[offset + num_neighbors + K * 1] = indices[event_idx] - min_index;
                    num_neighbors++;
                    if (num_neighbors >= max_num_neighbors) break;
                }
            }
            spiral_next(&sp);
        }
    }
}
