// graph_build.hip -- window (reset=True) event-graph builder for gfx950.
//
// What the reference does per window (src/dagr/graph/ev_graph.py:52-103, graph/utils.py:6-23,
// ev_graph.cu:15-80,169-212): refill a B x 128 x H x W int32 FIFO volume with -1 (157 MB/sample at
// 640x480), sort events by pixel, rewrite one 128-deep FIFO column per active pixel, then one
// thread per event walks a (2r+1)^2 spiral of FIFO columns, and a boolean-mask pass compacts a
// -1-padded int64 edge buffer.
//
// What this file does instead (same edge set, same order, bit-exact):
//   * events are bucketed by linear pixel p = x + W*(y + H*b) into a CSR-by-pixel array
//     (per-pixel counters -> exclusive scan -> scatter -> in-segment order fix-up).  After all N
//     events of a reset window are inserted, the FIFO column of pixel p holds exactly the newest
//     min(count_p, Q) events of that pixel, newest first -- i.e. the tail of its CSR segment read
//     backwards.  The 157 MB volume and its refill disappear; the search touches a (P+1)-int
//     offset array (1.2 MB/sample, L2-resident) plus {id, t} pairs that sit contiguously per pixel.
//   * the search runs 16 lanes per destination event (4 events per wave64): each lane owns one
//     spiral position of the current 16-position chunk, counts its admissible sources, a 16-lane
//     prefix sum reproduces the reference's sequential "first K in spiral order, newest first
//     inside a pixel" cut exactly, and the chunk loop exits as soon as K slots are filled (dense
//     scenes finish in the first chunk, like the reference's early break).
//   * output is a fixed-stride neighbour list [N, K] (int32 source + int16 offset code) + deg[N]:
//     no -1 fill, no compaction pass, no host sync; the offset code is the SplineConv LUT index.
#include "common.hpp"

#include <type_traits>

namespace dagr {
namespace {

constexpr int kVisBit = (int)0x80000000;  // slot_xyb bit 31: among the newest Q events of its pixel
constexpr int kShortSeg = 64;    // segments up to this length are ordered by per-slot rank counting
constexpr int kMaxQueue = 1024;  // LDS staging bound for the long-segment path
constexpr int kMaxSpiral = 4096; // (2r+1)^2 bound for the LDS spiral table (r <= 31)

struct GraphWs {
    int32_t *cnt;       // [P+1]  per-pixel event counters; all-zero between builds (invariant)
    int32_t *start;     // [P+1]  exclusive scan of cnt
    int32_t *scan_tmp;  // [scan_chained_state_bytes(P+1) / 4]: ticket, tag and per-tile words of the one-launch scan
    int32_t *ev_xyb;    // [Nmax] x | y<<12 | b<<24  (denormalised ints)
    int32_t *ev_t;      // [Nmax] denormalised timestamp (us)
    int32_t *ev_rank;   // [Nmax] arrival rank inside the pixel (arbitrary order)
    int32_t *slot_tmp;  // [Nmax] event id per CSR slot, arrival order
    int2 *slot_it;      // [Nmax] {event id, t} per CSR slot, ascending id inside a pixel
    int32_t *slot_xyb;  // [Nmax] x | y<<12 | b<<24 per CSR slot
    int32_t *ev_slot;   // [Nmax] CSR slot of every event (-1: dropped)
    int32_t *long_list; // [Nmax/kShortSeg + 1] pixels whose segment is longer than kShortSeg
    int32_t *status;    // [8]: 0 n_long, 1 flags, 2..3 num_edges (uint64), 5 / 7 deferral list lengths, 6 unsorted timestamps
    int64_t P;
};

size_t carve(const dagr_graph_desc &d, char *base, GraphWs *ws) {
    const int64_t P = (int64_t)d.width * d.height * d.batch_size;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off = align_up(off + bytes, 256);
        return base ? base + o : nullptr;
    };
    int32_t *cnt = (int32_t *)take((P + 1 + 8) * 4);
    int32_t *start = (int32_t *)take((P + 1 + 8) * 4);
    int32_t *scan_tmp = (int32_t *)take(scan_chained_state_bytes(P + 1));
    int32_t *ev_xyb = (int32_t *)take(d.max_events * 4);
    int32_t *ev_t = (int32_t *)take(d.max_events * 4);
    int32_t *ev_rank = (int32_t *)take(d.max_events * 4);
    int32_t *slot_tmp = (int32_t *)take(d.max_events * 4);
    int2 *slot_it = (int2 *)take(d.max_events * 8);
    int32_t *slot_xyb = (int32_t *)take(d.max_events * 4);
    int32_t *ev_slot = (int32_t *)take(d.max_events * 4);
    int32_t *long_list = (int32_t *)take((d.max_events / kShortSeg + 2) * 4);
    int32_t *status = (int32_t *)take(16 * 4);     // (8, 9: the staging launch's flag words)
    if (ws) *ws = GraphWs{cnt, start, scan_tmp, ev_xyb, ev_t, ev_rank, slot_tmp, slot_it, slot_xyb, ev_slot, long_list, status, P};
    return off;
}

int validate(const dagr_graph_desc *d) {
    DAGR_CHECK_ARG(d != nullptr, "desc is NULL");
    DAGR_CHECK_ARG(d->width > 0 && d->width <= 4096 && d->height > 0 && d->height <= 4096,
                   "width/height must be in 1..4096");
    DAGR_CHECK_ARG(d->batch_size > 0 && d->batch_size <= 127, "batch_size must be in 1..127");
    DAGR_CHECK_ARG((int64_t)d->width * d->height * d->batch_size < (1ll << 31) - 16, "B*H*W overflows int32");
    DAGR_CHECK_ARG(d->max_neighbors >= 1 && d->max_neighbors <= 64, "max_neighbors must be in 1..64");
    DAGR_CHECK_ARG(d->queue_size >= 1 && d->queue_size <= kMaxQueue, "queue_size must be in 1..1024");
    DAGR_CHECK_ARG(d->radius >= 0 && (2 * d->radius + 1) * (2 * d->radius + 1) <= kMaxSpiral, "radius must be in 0..31");
    DAGR_CHECK_ARG(d->time_window > 0, "time_window must be > 0");
    DAGR_CHECK_ARG(d->max_events >= 1 && d->max_events < (1ll << 31) / 64, "max_events out of range");
    return DAGR_OK;
}

// ---------------------------------------------------------------------------------------------
// K1: denormalise (ev_tgn.py:11-16) + per-pixel count.  One thread per event.
//   int(pos * [W,H,T] + 1e-3): fp32 multiply, fp32 add (separately rounded -- this TU is built with
//   -ffp-contract=off), truncation toward zero.
// `flags`: where the two conditions an event can raise are recorded -- the builder's status words (flags[1] |= 1: outside
// the sensor / batch; flags[6] = 1: timestamps not sorted), or the staging launch's own two words (see k_stage_window).
template <typename BatchT, bool kIntPos>
__device__ __forceinline__ void count_event(int e, const void *__restrict__ pos_, const BatchT *__restrict__ batch, int W,
                                            int H, int B, float fW, float fH, float fT, int32_t *__restrict__ cnt,
                                            int32_t *__restrict__ ev_xyb, int32_t *__restrict__ ev_t,
                                            int32_t *__restrict__ ev_rank, int32_t *__restrict__ flag_fov,
                                            int32_t *__restrict__ flag_time) {
    int x, y, t;
    if (kIntPos) {  // already-denormalised int32 [N,3] (SlidingWindowGraph.forward's own input contract)
        const int32_t *pos = static_cast<const int32_t *>(pos_);
        x = pos[3 * (int64_t)e + 0]; y = pos[3 * (int64_t)e + 1]; t = pos[3 * (int64_t)e + 2];
    } else {
        const float *pos = static_cast<const float *>(pos_);
        const float px = pos[3 * (int64_t)e + 0], py = pos[3 * (int64_t)e + 1], pt = pos[3 * (int64_t)e + 2];
        x = (int)(fW * px + 1e-3f);
        y = (int)(fH * py + 1e-3f);
        t = (int)(fT * pt + 1e-3f);
    }
    const int b = (int)batch[e];
    ev_t[e] = t;
    // time flag: set when timestamps are not non-decreasing in event order inside a sample.  Ids then do not order time,
    // and the search kernels fall back from the two binary searches per pixel to the reference's linear FIFO walk.
    if (e > 0 && (int)batch[e - 1] == b) {
        int tp;
        if (kIntPos) tp = static_cast<const int32_t *>(pos_)[3 * (int64_t)(e - 1) + 2];
        else tp = (int)(fT * static_cast<const float *>(pos_)[3 * (int64_t)(e - 1) + 2] + 1e-3f);
        if (tp > t) *flag_time = 1;
    }
    if (x < 0 || x >= W || y < 0 || y >= H || b < 0 || b >= B) {
        // The reference would index its FIFO volume out of bounds here; we flag and drop the
        // event from the pixel index (it keeps its self loop).
        atomicOr(flag_fov, 1);
        ev_xyb[e] = -1;
        ev_rank[e] = 0;
        return;
    }
    ev_xyb[e] = x | (y << 12) | (b << 24);
    const int p = x + W * (y + H * b);
    ev_rank[e] = atomicAdd(&cnt[p], 1);
}

template <typename BatchT, bool kIntPos>
__global__ __launch_bounds__(kBlock) void k_count(const void *__restrict__ pos_, const BatchT *__restrict__ batch,
                                                 int N, int W, int H, int B, float fW,
                                                 float fH, float fT, int32_t *__restrict__ cnt,
                                                 int32_t *__restrict__ ev_xyb, int32_t *__restrict__ ev_t,
                                                 int32_t *__restrict__ ev_rank, int32_t *__restrict__ status) {
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= N) return;
    count_event<BatchT, kIntPos>(e, pos_, batch, W, H, B, fW, fH, fT, cnt, ev_xyb, ev_t, ev_rank, status + 1, status + 6);
}

// K3: scatter event ids into their pixel segment (arrival order).
// n_dev: the window's event count in device memory (launches sized for a capacity N: captured HIP graphs); K1 then ran
// inside the staging launch, whose two flag words (status[8], status[9]) this launch moves into the builder's and re-arms.
__global__ __launch_bounds__(kBlock) void k_scatter(int N, const int32_t *__restrict__ n_dev, int W, int H,
                                                   const int32_t *__restrict__ ev_xyb,
                                                   const int32_t *__restrict__ ev_rank,
                                                   const int32_t *__restrict__ start,
                                                   int32_t *__restrict__ slot_tmp, int32_t *__restrict__ ev_slot,
                                                   int32_t *__restrict__ status) {
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (n_dev && e == 0) {
        if (status[8]) { atomicOr(&status[1], 1); status[8] = 0; }
        if (status[9]) { status[6] = 1; status[9] = 0; }
    }
    if (e >= N || (n_dev && e >= *n_dev)) return;
    const int c = ev_xyb[e];
    if (c < 0) { ev_slot[e] = -1; return; }
    const int p = (c & 4095) + W * (((c >> 12) & 4095) + H * (c >> 24));
    slot_tmp[start[p] + ev_rank[e]] = e;
}

// K4: order each pixel segment by ascending event id (== the reference's stable sort by pixel,
// graph/utils.py:10).  One thread per CSR slot; segments longer than kShortSeg are deferred.
__global__ __launch_bounds__(kBlock) void k_order(int N, int64_t P, int W, int H, int Q, const int32_t *__restrict__ ev_xyb,
                                                 const int32_t *__restrict__ ev_t,
                                                 const int32_t *__restrict__ start,
                                                 const int32_t *__restrict__ slot_tmp, int2 *__restrict__ slot_it,
                                                 int32_t *__restrict__ slot_xyb, int32_t *__restrict__ ev_slot,
                                                 int32_t *__restrict__ long_list, int long_cap,
                                                 int32_t *__restrict__ status) {
    const int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= N || s >= start[P]) return;  // start[P] = number of indexed events (<= N)
    const int e = slot_tmp[s];
    const int c = ev_xyb[e];
    const int p = (c & 4095) + W * (((c >> 12) & 4095) + H * (c >> 24));
    const int a = start[p];
    const int n = start[p + 1] - a;
    if (n <= kShortSeg) {
        int rank = 0;
        for (int k = 0; k < n; k++) rank += (slot_tmp[a + k] < e) ? 1 : 0;
        slot_it[a + rank] = make_int2(e, ev_t[e]);
        slot_xyb[a + rank] = c | ((n - rank <= Q) ? kVisBit : 0);   // FIFO depth (ev_graph.cu:201-211)
        ev_slot[e] = a + rank;
    } else if (s == a) {
        const int i = atomicAdd(&status[0], 1);
        if (i < long_cap) long_list[i] = p; else atomicOr(&status[1], 2);
    }
}

// K5: long segments (> kShortSeg events on one pixel).  Only the newest m = min(n, Q) events of a
// pixel are ever visible to the search (FIFO depth Q, ev_graph.cu:201-211), so: radix-select the
// m-th largest id, gather the m newest into LDS, rank-sort them into the tail of the segment.
__global__ __launch_bounds__(kBlock) void k_order_long(int Q, const int32_t *__restrict__ ev_xyb,
                                                      int32_t *__restrict__ slot_xyb, int32_t *__restrict__ ev_slot,
                                                      const int32_t *__restrict__ ev_t,
                                                      const int32_t *__restrict__ start,
                                                      const int32_t *__restrict__ slot_tmp,
                                                      int2 *__restrict__ slot_it,
                                                      const int32_t *__restrict__ long_list, int long_cap,
                                                      const int32_t *__restrict__ status) {
    __shared__ int hist[256];
    __shared__ int sel[kMaxQueue];
    __shared__ int sh_prefix, sh_remaining, sh_nsel, sh_nrest;
    int n_long = status[0];
    if (n_long > long_cap) n_long = long_cap;
    for (int li = blockIdx.x; li < n_long; li += gridDim.x) {
        const int p = long_list[li];
        const int a = start[p];
        const int n = start[p + 1] - a;
        const int m = n < Q ? n : Q;
        unsigned thr = 0;
        if (n > m) {
            if (threadIdx.x == 0) { sh_prefix = 0; sh_remaining = m; }
            unsigned mask = 0;
            for (int shift = 24; shift >= 0; shift -= 8) {
                hist[threadIdx.x] = 0;
                __syncthreads();
                const unsigned prefix = (unsigned)sh_prefix;
                for (int k = threadIdx.x; k < n; k += kBlock) {
                    const unsigned v = (unsigned)slot_tmp[a + k];
                    if ((v & mask) == prefix) atomicAdd(&hist[(v >> shift) & 255], 1);
                }
                __syncthreads();
                if (threadIdx.x == 0) {
                    int rem = sh_remaining, d = 255;
                    for (; d > 0; d--) {
                        if (hist[d] >= rem) break;
                        rem -= hist[d];
                    }
                    sh_remaining = rem;
                    sh_prefix = (int)(prefix | ((unsigned)d << shift));
                }
                mask |= 255u << shift;
                __syncthreads();
            }
            thr = (unsigned)sh_prefix;
        }
        if (threadIdx.x == 0) { sh_nsel = 0; sh_nrest = 0; }
        __syncthreads();
        for (int k = threadIdx.x; k < n; k += kBlock) {
            const int v = slot_tmp[a + k];
            if ((unsigned)v >= thr) sel[atomicAdd(&sh_nsel, 1)] = v;
            // older events are invisible to the search (beyond the FIFO depth) but remain graph nodes:
            // keep them, in any order, in the head of the segment (voxel pooling walks the segment)
            else {
                const int o = a + atomicAdd(&sh_nrest, 1);
                slot_it[o] = make_int2(v, ev_t[v]);
                slot_xyb[o] = ev_xyb[v];               // not visible: beyond the FIFO depth
                ev_slot[v] = o;
            }
        }
        __syncthreads();
        // sh_nsel == m (ids are unique)
        for (int i = threadIdx.x; i < m; i += kBlock) {
            const int v = sel[i];
            int rank = 0;
            for (int j = 0; j < m; j++) rank += (sel[j] < v) ? 1 : 0;
            slot_it[a + (n - m) + rank] = make_int2(v, ev_t[v]);
            slot_xyb[a + (n - m) + rank] = ev_xyb[v] | kVisBit;
            ev_slot[v] = a + (n - m) + rank;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Closed form of SpiralOut (spiral.h:1-15): position s > 0 lies on the Chebyshev ring rho with
// (2rho-1)^2 <= s < (2rho+1)^2; the ring is walked from (rho, -rho+1) up (+y) to (rho, rho), then
// -x to (-rho, rho), then -y to (-rho, -rho), then +x to (rho, -rho): 2*rho positions per leg.
__host__ __device__ inline void spiral_offset(int s, int &sx, int &sy) {
    sx = 0; sy = 0;
    if (s <= 0) return;
    int rho = 1;
    while ((2 * rho + 1) * (2 * rho + 1) <= s) rho++;
    const int k = s - (2 * rho - 1) * (2 * rho - 1);  // 0 .. 8*rho-1 along the ring
    if (k < 2 * rho) { sx = rho; sy = -rho + 1 + k; }
    else if (k < 4 * rho) { sx = rho - 1 - (k - 2 * rho); sy = rho; }
    else if (k < 6 * rho) { sx = -rho; sy = rho - 1 - (k - 4 * rho); }
    else { sx = -rho + 1 + (k - 6 * rho); sy = -rho; }
}

// K6: spiral radius search, 16 lanes per destination event.
__device__ __forceinline__ int group16_inclusive_scan(int v) {
    const int l = threadIdx.x & 15;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        int n = __shfl_up(v, d, 16);
        if (l >= d) v += n;
    }
    return v;
}

// Level-0 node numbering.  The graph is emitted in *slot space*: node n is CSR slot n, i.e. events
// ordered by (sample, y, x) and by time inside a pixel.  Spatially adjacent destinations then sit next
// to each other in every level-0 array, so the segment-offset rows, the {id,t} candidates and (in the
// SplineConv) the source feature rows they touch are shared through L1/L2 instead of being re-fetched
// per event, and voxel pooling streams its members.  Event-order views (edge_index, permutations) are
// produced on demand by dagr_graph_edge_index / dagr_graph_node_order.
//
// Generic search (any radius): probes go to global memory, 16 lanes per destination, batches of 8 rounds.
__global__ __launch_bounds__(kBlock) void k_search(const int32_t *__restrict__ m_ptr, int W, int H, int K, int Q, int r,
                                                  float delta_t, const int32_t *__restrict__ slot_xyb,
                                                  const int32_t *__restrict__ start,
                                                  const int2 *__restrict__ slot_it,
                                                  int32_t *__restrict__ nbr_src, int16_t *__restrict__ nbr_code,
                                                  int32_t *__restrict__ deg, int32_t *__restrict__ status) {
    __shared__ int16_t sp_tab[kMaxSpiral];  // (sx + 64) | (sy + 64) << 8, spiral.h:1-15 order
    __shared__ int blk_edges;
    const int side = 2 * r + 1;
    const int S = side * side;
    for (int s = threadIdx.x; s < S; s += kBlock) {
        int sx, sy;
        spiral_offset(s, sx, sy);
        sp_tab[s] = (int16_t)((sx + 64) | ((sy + 64) << 8));
    }
    if (threadIdx.x == 0) blk_edges = 0;
    __syncthreads();
    const int M = *m_ptr;
    const int l = threadIdx.x & 15;
    const int n = (blockIdx.x * kBlock + threadIdx.x) >> 4;   // destination slot
    int total = 0;
    if (n < M) {
        const int2 me = slot_it[n];
        const int e = me.x, t = me.y;
        const int c = slot_xyb[n];
        const int64_t row = (int64_t)n * K;
        total = 1;
        if (l == 0) {
            nbr_src[row] = n;  // self loop first (ev_graph.cu:44-46)
            nbr_code[row] = (int16_t)(r * side + r);
        }
        const int x = c & 4095, y = (c >> 12) & 4095, b = (c >> 24) & 127;
        const int plane = W * H * b;
        constexpr int kRounds = 8;
        for (int s0 = 0; s0 < S && total < K;) {
            const int rounds = (s0 == 0) ? 1 : min(kRounds, (S - s0 + 15) >> 4);
            int bnd[kRounds], vis[kRounds], ecode[kRounds];
#pragma unroll
            for (int m = 0; m < kRounds; m++) {
                bnd[m] = 0; vis[m] = 0; ecode[m] = 0;
                const int s = s0 + 16 * m + l;
                if (m < rounds && s < S) {
                    const int code = sp_tab[s];
                    const int sx = (code & 255) - 64, sy = ((code >> 8) & 255) - 64;
                    ecode[m] = (sx + r) * side + (sy + r);
                    const int xn = x + sx, yn = y + sy;
                    if (xn >= 0 && yn >= 0 && xn < W && yn < H) {  // out of FOV: skip this pixel only
                        const int p = plane + yn * W + xn;
                        const int a = start[p];
                        bnd[m] = start[p + 1];
                        vis[m] = min(bnd[m] - a, Q);                // FIFO depth
                    }
                }
            }
            int2 it0[kRounds];
#pragma unroll
            for (int m = 0; m < kRounds; m++) {
                it0[m] = make_int2(0x7fffffff, 0);
                if (vis[m] > 0) it0[m] = slot_it[bnd[m] - 1];
            }
            // admissible sources per position, newest first:
            //   skip ids >= e (newer or self, ev_graph.cu:64); skip dt > delta (continue, :69)
            int v[kRounds];
#pragma unroll
            for (int m = 0; m < kRounds; m++) {
                int cnt = 0;
                if (vis[m] > 0) {
                    cnt = (it0[m].x < e && !((float)(t - it0[m].y) > delta_t)) ? 1 : 0;
                    for (int k = 1; k < vis[m] && cnt < K; k++) {
                        const int2 it = slot_it[bnd[m] - 1 - k];
                        if (it.x >= e) continue;
                        if ((float)(t - it.y) > delta_t) continue;
                        cnt++;
                    }
                }
                v[m] = cnt;
            }
            // sequential cut in spiral order: round by round, lane by lane
#pragma unroll
            for (int m = 0; m < kRounds; m++) {
                if (m < rounds) {   // uniform across the 16-lane group
                    const int incl = group16_inclusive_scan(v[m]);
                    int slot = total + incl - v[m];
                    total += __shfl(incl, 15, 16);
                    if (v[m] > 0 && slot < K) {
                        for (int k = 0; k < vis[m] && slot < K; k++) {
                            const int2 it = (k == 0) ? it0[m] : slot_it[bnd[m] - 1 - k];
                            if (it.x >= e) continue;
                            if ((float)(t - it.y) > delta_t) continue;
                            nbr_src[row + slot] = bnd[m] - 1 - k;   // source node = its CSR slot
                            nbr_code[row + slot] = (int16_t)ecode[m];
                            slot++;
                        }
                    }
                }
            }
            s0 += 16 * rounds;
        }
        if (total > K) total = K;
        if (l == 0) deg[n] = total;
    }
    // window edge count (status[2..3] as uint64)
    if (l == 0 && n < M) atomicAdd(&blk_edges, total);
    __syncthreads();
    if (threadIdx.x == 0 && blk_edges)
        atomicAdd(reinterpret_cast<unsigned long long *>(status + 2), (unsigned long long)blk_edges);
}

// level-0 inputs in node (slot) order -- pos, sample index and the [polarity | ... | pos_xy] feature row -- written by the
// LAST launch of the graph build (dagr_graph_build_window_inputs) instead of a launch of their own
struct GatherArgs {
    const float *pos, *feat;
    float *pos_s;
    int32_t *batch_s;
    float *x0;
    int ldx0, col_feat, col_pos;
};
__device__ __forceinline__ void gather_node(const GatherArgs &g, int n, const int2 *__restrict__ slot_it,
                                            const int32_t *__restrict__ slot_xyb) {
    const int e = slot_it[n].x;
    const float px = g.pos[3 * (size_t)e], py = g.pos[3 * (size_t)e + 1], pt = g.pos[3 * (size_t)e + 2];
    g.pos_s[3 * (size_t)n] = px; g.pos_s[3 * (size_t)n + 1] = py; g.pos_s[3 * (size_t)n + 2] = pt;
    g.batch_s[n] = (slot_xyb[n] >> 24) & 127;
    float *row = g.x0 + (size_t)n * g.ldx0;
    row[g.col_feat] = g.feat[e];
    row[g.col_pos] = px;
    row[g.col_pos + 1] = py;
}

// ---------------------------------------------------------------------------------------------
// K6 (r <= 7): LDS-tiled, persistent variant of k_search<true>.  Same cut semantics, fewer
// instructions per probed pixel:
//   * a 16-lane group loops over events; the spiral position of (round m, lane l) does not depend on
//     the event, so its tile offset and offset code live in registers for the whole kernel;
//   * most pixels hold 0 or 1 visible events: a round in which no lane has more than one takes a
//     ballot/popcount cut (the sequential "first K in spiral order" becomes a prefix popcount);
//     rounds with a multi-event pixel fall back to the exact prefix-sum walk.

// Admissible sources of one pixel for destination (e, t): the visible FIFO entries are the slots [lo_vis, bnd) of the
// pixel's segment, ids ascending.  "Older than the destination" (ev_graph.cu:64) is a prefix [lo_vis, hi); when ids
// order time (status[6] == 0) "dt <= delta" (ev_graph.cu:69) is a suffix [lo, hi) of it: two binary searches instead of
// the reference's newest-first walk over up to Q entries.  Returns [lo, hi); the walk order newest-first is hi-1 .. lo.
__device__ __forceinline__ void admissible_range(const int2 *__restrict__ slot_it, int lo_vis, int bnd, int e, int t,
                                                 float delta_t, int &lo, int &hi) {
    int a = lo_vis, b = bnd;                 // first slot in [a, b) with id >= e
    while (a < b) {
        const int m = (a + b) >> 1;
        if (slot_it[m].x < e) a = m + 1; else b = m;
    }
    hi = a;
    a = lo_vis; b = hi;                      // first slot in [a, b) with dt <= delta
    while (a < b) {
        const int m = (a + b) >> 1;
        if ((float)(t - slot_it[m].y) > delta_t) a = m + 1; else b = m;
    }
    lo = a;
}

constexpr int kTileRounds = 15;  // ceil(15*15 / 16)

__global__ __launch_bounds__(kBlock) void k_search_tiled(const int32_t *__restrict__ m_ptr, int W, int H, int K, int Q,
                                                        int r, float delta_t,
                                                        const int32_t *__restrict__ slot_xyb,
                                                        const int32_t *__restrict__ start,
                                                        const int2 *__restrict__ slot_it,
                                                        int32_t *__restrict__ nbr_src,
                                                        int16_t *__restrict__ nbr_code, int32_t *__restrict__ deg,
                                                        int32_t *__restrict__ status,
                                                        const int32_t *__restrict__ node_list,
                                                        const int32_t *__restrict__ node_list_count, GatherArgs gather) {
    __shared__ int tile[(kBlock / 16) * 16 * 17];
    if (gather.pos) {       // (independent of the search: the pixel index is final since k_order_long)
        const int m = *m_ptr;
        for (int n = blockIdx.x * kBlock + threadIdx.x; n < m; n += gridDim.x * kBlock) gather_node(gather, n, slot_it, slot_xyb);
    }
    // list mode (the usual call): most windows defer nothing -- leave before the per-lane spiral constants are built
    // (an empty sweep of the persistent grid cost 10 us per window)
    if (node_list && *node_list_count <= 0) return;
    const int side = 2 * r + 1;
    const int S = side * side;
    const int l = threadIdx.x & 15;
    const int grp = threadIdx.x >> 4;
    const int gshift = threadIdx.x & 48;  // bit offset of this group inside the wave ballot
    int *my_tile = tile + grp * 16 * 17;
    // per-lane constants: position s = 16*m + l -> (tile offset | offset code << 16), -1 if s >= S
    int pc[kTileRounds];
#pragma unroll
    for (int m = 0; m < kTileRounds; m++) {
        const int s = 16 * m + l;
        int sx, sy;
        spiral_offset(s, sx, sy);
        pc[m] = (s < S) ? (((sy + r) * 17 + (sx + r)) | (((sx + r) * side + (sy + r)) << 16)) : -1;
    }
    long long edges_acc = 0;
    const bool sorted_t = status[6] == 0;    // ids order time: per-pixel binary searches are exact
    // every block sweeps a contiguous range of slots (= a run of pixels along image rows): consecutive
    // destinations share most of their neighbourhood, so offsets and candidates come out of L1/L2
    const int M = node_list ? *node_list_count : *m_ptr;   // list mode: only the nodes the row kernel deferred
    // XCD x = blockIdx % 8 owns the x-th eighth of the slots (one sample for B = 8), its blocks split it
    // into contiguous strips: vertical neighbours of a strip live in the same XCD's L2
    const int G = gridDim.x, nx = (G % 8 == 0) ? 8 : 1;
    const int xcd = blockIdx.x % nx, lb = blockIdx.x / nx, bpx = G / nx;
    const int chunk = (M + nx - 1) / nx;
    const int per_block = (chunk + bpx - 1) / bpx;
    const int n_begin = xcd * chunk + lb * per_block;
    const int n_end = min(min(M, (xcd + 1) * chunk), n_begin + per_block);
    for (int ni = n_begin + grp; ni < n_end; ni += kBlock / 16) {
        const int n = node_list ? node_list[ni] : ni;
        const int2 me = slot_it[n];
        const int e = me.x, t = me.y;
        const int c = slot_xyb[n];
        const int64_t row = (int64_t)n * K;
        int total = 1;
        if (l == 0) {
            nbr_src[row] = n;  // self loop first (ev_graph.cu:44-46)
            nbr_code[row] = (int16_t)(r * side + r);
        }
        {
            const int x = c & 4095, y = (c >> 12) & 4095, b = (c >> 24) & 127;
            const int plane = W * H * b;
            const int lo = max(x - r, 0), hi = min(x + r, W - 1) + 1;
            const int col = plane + min(max(x - r + l, lo), hi);
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int yn = y + i - r;
                int val = 0;
                if (i < side && yn >= 0 && yn < H) val = start[col + yn * W];
                my_tile[i * 17 + l] = val;
            }
            __builtin_amdgcn_wave_barrier();

            auto batch = [&](auto first_c, auto count_c) {
                constexpr int m0 = decltype(first_c)::value, nr = decltype(count_c)::value;
                int bnd[nr], vis[nr];
                int2 it0[nr];
#pragma unroll
                for (int j = 0; j < nr; j++) {
                    const int p = pc[m0 + j];
                    bnd[j] = 0; vis[j] = 0;
                    if (p >= 0) {
                        const int cur = my_tile[p & 0xffff];
                        bnd[j] = my_tile[(p & 0xffff) + 1];
                        vis[j] = min(bnd[j] - cur, Q);   // FIFO depth
                    }
                }
#pragma unroll
                for (int j = 0; j < nr; j++) {
                    it0[j] = make_int2(0x7fffffff, 0);
                    if (vis[j] > 0) it0[j] = slot_it[bnd[j] - 1];
                }
#pragma unroll
                for (int j = 0; j < nr; j++) {
                    if (total >= K) break;                               // group-uniform
                    if (16 * (m0 + j) >= S) break;
                    // newest entry: skip ids >= e (ev_graph.cu:64), skip dt > delta (:69)
                    const bool ok0 = vis[j] > 0 && it0[j].x < e && !((float)(t - it0[j].y) > delta_t);
                    const unsigned multi = (unsigned)(__ballot(vis[j] > 1) >> gshift) & 0xffffu;
                    const int ecode = pc[m0 + j] >> 16;
                    if (multi == 0) {
                        const unsigned bits = (unsigned)(__ballot(ok0) >> gshift) & 0xffffu;
                        const int slot = total + __popc(bits & ((1u << l) - 1u));
                        if (ok0 && slot < K) {
                            nbr_src[row + slot] = bnd[j] - 1;   // source node = its CSR slot
                            nbr_code[row + slot] = (int16_t)ecode;
                        }
                        total += __popc(bits);
                    } else if (sorted_t) {
                        int v = 0, lo = 0, hi = 0;
                        if (vis[j] > 1) {
                            admissible_range(slot_it, bnd[j] - vis[j], bnd[j], e, t, delta_t, lo, hi);
                            v = min(hi - lo, K);
                        } else if (vis[j] == 1) {
                            v = ok0 ? 1 : 0;
                            hi = bnd[j]; lo = hi - v;
                        }
                        const int incl = group16_inclusive_scan(v);
                        int slot = total + incl - v;
                        total += __shfl(incl, 15, 16);
                        for (int k = 0; k < v && slot < K; k++, slot++) {
                            nbr_src[row + slot] = hi - 1 - k;          // newest first
                            nbr_code[row + slot] = (int16_t)ecode;
                        }
                    } else {
                        int v = 0;
                        if (vis[j] > 0) {
                            v = ok0 ? 1 : 0;
                            for (int k = 1; k < vis[j] && v < K; k++) {
                                const int2 it = slot_it[bnd[j] - 1 - k];
                                if (it.x >= e) continue;
                                if ((float)(t - it.y) > delta_t) continue;
                                v++;
                            }
                        }
                        const int incl = group16_inclusive_scan(v);
                        int slot = total + incl - v;
                        total += __shfl(incl, 15, 16);
                        if (v > 0 && slot < K) {
                            for (int k = 0; k < vis[j] && slot < K; k++) {
                                const int2 it = (k == 0) ? it0[j] : slot_it[bnd[j] - 1 - k];
                                if (it.x >= e) continue;
                                if ((float)(t - it.y) > delta_t) continue;
                                nbr_src[row + slot] = bnd[j] - 1 - k;
                                nbr_code[row + slot] = (int16_t)ecode;
                                slot++;
                            }
                        }
                    }
                }
            };
            batch(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
            if (total < K && S > 16) batch(std::integral_constant<int, 1>{}, std::integral_constant<int, 7>{});
            if (total < K && S > 128) batch(std::integral_constant<int, 8>{}, std::integral_constant<int, 7>{});
            if (total > K) total = K;
            __builtin_amdgcn_wave_barrier();  // tile reads of this event precede the next event's writes
        }
        if (l == 0) { deg[n] = total; edges_acc += total; }
    }
    {   // one atomic per wave instead of one per lane group (same single-counter drain as in k_search_rows)
        unsigned long long v = (l == 0) ? (unsigned long long)edges_acc : 0ull;
        v += __shfl_down(v, 32, 64);
        v += __shfl_down(v, 16, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(reinterpret_cast<unsigned long long *>(status + 2), v);
    }
}

// ---------------------------------------------------------------------------------------------
// K6 fast path (r <= 7): candidate-centric search.  In slot order the events of one pixel row of the
// neighbourhood are ONE contiguous slot range [start(row, x-r), start(row, x+r+1)), so
//   1. 15 lanes fetch the 15 row ranges (2 offsets each), a 16-lane scan concatenates them;
//   2. the C candidates are read 16 at a time, coalesced ({id,t} + packed x|y|visible): each lane
//      tests its candidate (older than the destination, dt <= delta, inside the FIFO depth) and
//      keys it with (spiral rank of its pixel, recency inside the pixel);
//   3. the reference's sequential walk "spiral order, newest first, stop at K" is exactly "the K-1
//      smallest keys in key order": valid candidates are compacted into LDS (ballot/popcount) and
//      each takes the slot given by the number of smaller keys.
// Work is proportional to the events actually present in the neighbourhood (~0.33/pixel in the
// benchmark stream) instead of to its (2r+1)^2 pixels.  Neighbourhoods with more than kRowCap
// candidates (dense scenes, where the position-centric kernel exits after the first ring anyway) are
// appended to a list that k_search_tiled processes afterwards.
// kRowCap candidates per neighbourhood: the key list is the kernel's LDS footprint (20 KiB per workgroup); 320 covers
// uniform streams up to ~430 k events per 640x480 window.  Denser neighbourhoods go to the position-centric kernel.
// (LIST_IN re-sweeps a deferral list; kept for experiments.)
constexpr int kRowCap = 320;

// ROUNDS x 16 candidates are requested before the first is examined; WAVES per SIMD = the register budget (512 / WAVES)
template <int CAP, bool LIST_IN, int ROUNDS = 4, int WAVES = 7>
__global__ __launch_bounds__(kBlock, WAVES) void k_search_rows(const int32_t *__restrict__ m_ptr, int W, int H, int K, int r,
                                                       float delta_t, const int32_t *__restrict__ slot_xyb,
                                                       const int32_t *__restrict__ start,
                                                       const int2 *__restrict__ slot_it,
                                                       int32_t *__restrict__ nbr_src, int16_t *__restrict__ nbr_code,
                                                       int32_t *__restrict__ deg, int32_t *__restrict__ status,
                                                       int32_t *__restrict__ node_list,
                                                       int32_t *__restrict__ node_list_count,
                                                       const int32_t *__restrict__ node_in,
                                                       const int32_t *__restrict__ node_in_count) {
    constexpr int G = kBlock / 16;
    __shared__ unsigned char sp_rank[256];      // spiral index of offset (dy + r) * 16 + (dx + r)
    __shared__ unsigned short sp_dec[256];      // spiral index -> offset code (dx + r) * side + (dy + r) | (dy + r) << 12
    __shared__ int row_lo[G][16], row_base[G][17];
    // candidate keys, (spiral rank << 20) | (0xFFFFF - position in its row range): the source slot follows from the key.
    // Dynamic LDS on purpose: with the size visible the compiler's occupancy estimate (made against 64 KiB) drops to 4
    // waves per SIMD at CAP = 320 and it stops holding the kernel to the 72 registers that 7 waves need; the hardware
    // has 160 KiB per CU and runs 6 workgroups of this kernel.
    extern __shared__ int v_key_dyn[];
    __shared__ int def_buf[kBlock / 64][64];
    __shared__ unsigned long long blk_edges;   // the block's edge count: ONE global atomic per workgroup at the end
    int wcnt = 0;    // entries of this wave's deferral buffer (uniform over the wave's active lanes)
    const int side = 2 * r + 1;
    const int S = side * side;
    for (int s = threadIdx.x; s < S; s += kBlock) {
        int sx, sy;
        spiral_offset(s, sx, sy);
        sp_rank[(sy + r) * 16 + (sx + r)] = (unsigned char)s;
        sp_dec[s] = (unsigned short)(((sx + r) * side + (sy + r)) | ((sy + r) << 12));
    }
    if (threadIdx.x == 0) blk_edges = 0ull;
    __syncthreads();
    const int l = threadIdx.x & 15;
    const int grp = threadIdx.x >> 4;
    int *const v_keys = v_key_dyn + grp * CAP;
    const int gshift = threadIdx.x & 48;
    const unsigned lt_mask = (1u << l) - 1u;
    long long edges_acc = 0;
    const int M = LIST_IN ? *node_in_count : *m_ptr;     // list mode: only the nodes the first sweep deferred
    if (M <= 0) return;
    const int Gd = gridDim.x, nx = (Gd % 8 == 0) ? 8 : 1;
    const int xcd = blockIdx.x % nx, lb = blockIdx.x / nx, bpx = Gd / nx;
    const int chunk = (M + nx - 1) / nx;
    const int per_block = (chunk + bpx - 1) / bpx;
    const int n_begin = xcd * chunk + lb * per_block;
    const int n_end = min(min(M, (xcd + 1) * chunk), n_begin + per_block);
    // Software pipeline over this lane group's destinations: the dependent chain {id,t | x,y,b} -> row bounds ->
    // candidates is three HBM latencies; the first two are issued one and two destinations ahead.
    auto node_of = [&](int i) { return LIST_IN ? node_in[min(i, M - 1)] : min(i, M - 1); };
    auto load_node = [&](int i, int2 &me, int &c) {
        const int nn = node_of(i);
        me = slot_it[nn];
        c = slot_xyb[nn];
    };
    auto load_rows = [&](int c, int &lo, int &len) {
        const int x = c & 4095, y = (c >> 12) & 4095, b = (c >> 24) & 127;
        const int yn = y + l - r;
        lo = 0;
        len = 0;
        if (l < side && yn >= 0 && yn < H) {
            const int base = W * (yn + H * b);
            lo = start[base + max(x - r, 0)];
            len = start[base + min(x + r, W - 1) + 1] - lo;
        }
    };
    int2 me, me1;
    int c, c1, lo, len;
    if (n_begin + grp < n_end) {
        load_node(n_begin + grp, me, c);
        load_node(n_begin + grp + G, me1, c1);
        load_rows(c, lo, len);
    }
    for (int ni = n_begin + grp; ni < n_end; ni += G) {
        const int n = node_of(ni);
        // next destinations' loads (results are used one iteration later)
        int2 me2;
        int c2, lo1, len1;
        load_node(ni + 2 * G, me2, c2);
        load_rows(c1, lo1, len1);
        const int e = me.x, t = me.y;
        const int x = c & 4095;
        const int64_t row = (int64_t)n * K;
        // 1. row ranges: 15 lanes hold the 15 row ranges, a 16-lane scan concatenates them
        const int incl = group16_inclusive_scan(len);
        const int C = __shfl(incl, 15, 16);
        const int lo_cur = lo, len_cur = len;
        me = me1; c = c1; me1 = me2; c1 = c2; lo = lo1; len = len1;   // rotate the pipeline
        // Dense neighbourhoods are deferred to the position-centric kernel.  The list append is aggregated per wave
        // (LDS buffer, one global atomic per ~48 entries): one atomicAdd per destination on a single counter
        // serialises at ~350 M/s and was the whole cost of this kernel on dense windows (4.5 ms at 1.6 M deferrals).
        const bool defer = C > CAP;
        {
            const unsigned long long dmask = __ballot(defer && l == 0);
            if (dmask) {
                const int lane64 = threadIdx.x & 63;
                if (defer && l == 0) def_buf[threadIdx.x >> 6][wcnt + __popcll(dmask & ((1ull << lane64) - 1ull))] = n;
                wcnt += __popcll(dmask);
                if (wcnt > 64 - 4) {
                    __builtin_amdgcn_wave_barrier();
                    // written by the wave's first lane group: it runs the most iterations of this loop, so it is
                    // active whenever any group of the wave still is (the others may have left the loop already)
                    int base = 0;
                    if (lane64 == 0) base = atomicAdd(node_list_count, wcnt);
                    base = __shfl(base, 0, 64);
                    if (lane64 < 16)
                        for (int i = lane64; i < wcnt; i += 16) node_list[base + i] = def_buf[threadIdx.x >> 6][i];
                    wcnt = 0;
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        if (defer) continue;
        row_lo[grp][l] = lo_cur;
        row_base[grp][l] = incl - len_cur;
        if (l == 15) row_base[grp][16] = C;
        __builtin_amdgcn_wave_barrier();
        // 2. candidates, 16 per round, 4 rounds of loads in flight.  (Skipping the rounds no lane group of the wave needs
        //    -- a wave-uniform break out of the batch -- was tried in round 3: same time, and the break cost 19 more
        //    spilled registers, i.e. 2.6 x the kernel's HBM traffic in scratch.)
        int V = 0;
        for (int c0 = 0; c0 < C; c0 += 16 * ROUNDS) {
            int2 it[ROUNDS];
            int cxv[ROUNDS], rpv[ROUNDS];     // rpv: position in its row range (20 bits) | row << 20 -- one register, not two:
                                              // the kernel sits exactly at its 72-register budget (7 waves per SIMD)
#pragma unroll
            for (int q = 0; q < ROUNDS; q++) {
                const int ci = c0 + 16 * q + l;
                it[q] = make_int2(0, 0);
                cxv[q] = 0; rpv[q] = 0;
                if (ci < C) {
                    int rr = 0;
                    if (row_base[grp][rr + 8] <= ci) rr += 8;
                    if (row_base[grp][rr + 4] <= ci) rr += 4;
                    if (row_base[grp][rr + 2] <= ci) rr += 2;
                    if (row_base[grp][rr + 1] <= ci) rr += 1;
                    const int rel = ci - row_base[grp][rr];
                    rpv[q] = rel | (rr << 20);
                    const int sv = row_lo[grp][rr] + rel;
                    it[q] = slot_it[sv];
                    cxv[q] = slot_xyb[sv];
                }
            }
#pragma unroll
            for (int q = 0; q < ROUNDS; q++) {
                const int ci = c0 + 16 * q + l;
                bool valid = false;
                int key = 0;
                if (ci < C) {
                    // visible in the FIFO; older than the destination (ev_graph.cu:64); dt <= delta (:69)
                    valid = (cxv[q] < 0) && it[q].x < e && !((float)(t - it[q].y) > delta_t);
                    const int dx = (cxv[q] & 4095) - x;
                    const int rank = sp_rank[(rpv[q] >> 20) * 16 + (dx + r)];
                    // spiral rank first, then newest first inside the pixel (larger slot = newer)
                    key = (rank << 20) | (0xFFFFF - (rpv[q] & 0xFFFFF));
                }
                const unsigned bits = (unsigned)(__ballot(valid) >> gshift) & 0xffffu;
                if (valid) {
                    const int pidx = V + __popc(bits & lt_mask);
                    v_keys[pidx] = key;
                }
                V += __popc(bits);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // 3. the K-1 smallest keys, in key order
        if (l == 0) {
            nbr_src[row] = n;  // self loop first (ev_graph.cu:44-46)
            nbr_code[row] = (int16_t)(r * side + r);
        }
        // Dense neighbourhoods hold far more admissible candidates than the K-1 that survive, and the rank counting
        // below is quadratic in their number: first drop everything beyond the spiral-rank bin (16 positions per bin,
        // one bin per lane) in which the K-1'th candidate falls.
        if (V > 32) {
            int cntb = 0;
            for (int j = 0; j < V; j++) cntb += ((v_keys[j] >> 24) == l) ? 1 : 0;
            const int incl_b = group16_inclusive_scan(cntb);
            const unsigned reach = (unsigned)(__ballot(incl_b >= K - 1) >> gshift) & 0xffffu;
            const int bstar = reach ? (__ffs(reach) - 1) : 15;
            int Wk = 0;
            for (int base = 0; base < V; base += 16) {
                const int vi = base + l;
                const int key = vi < V ? v_keys[vi] : 0x7fffffff;
                const bool keep = vi < V && (key >> 24) <= bstar;
                const unsigned bits = (unsigned)(__ballot(keep) >> gshift) & 0xffffu;
                __builtin_amdgcn_wave_barrier();
                if (keep) v_keys[Wk + __popc(bits & lt_mask)] = key;   // in place: writes stay below this step's reads
                Wk += __popc(bits);
                __builtin_amdgcn_wave_barrier();
            }
            V = Wk;
        }
        for (int vi = l; vi < V; vi += 16) {
            const int mk = v_keys[vi];
            int rk = 0;
            for (int j = 0; j < V; j++) rk += (v_keys[j] < mk) ? 1 : 0;
            if (rk < K - 1) {
                const int dec = sp_dec[mk >> 20];
                nbr_src[row + 1 + rk] = row_lo[grp][dec >> 12] + (0xFFFFF - (mk & 0xFFFFF));   // row range start + position
                nbr_code[row + 1 + rk] = (int16_t)(dec & 0xfff);
            }
        }
        const int total = 1 + min(V, K - 1);
        if (l == 0) { deg[n] = total; edges_acc += total; }
        __builtin_amdgcn_wave_barrier();  // LDS lists are reused by the next destination
    }
    {   // flush the wave's deferral buffer (lane group 0 of a wave runs the most iterations: its count is the wave's)
        const int lane64 = threadIdx.x & 63;
        wcnt = __shfl(wcnt, 0, 64);
        __builtin_amdgcn_wave_barrier();
        if (wcnt > 0) {
            int base = 0;
            if (lane64 == 0) base = atomicAdd(node_list_count, wcnt);
            base = __shfl(base, 0, 64);
            if (lane64 < wcnt) node_list[base + lane64] = def_buf[threadIdx.x >> 6][lane64];
        }
    }
    // A persistent grid ends all of its ~25 k lane groups at about the same time: one global atomic per group on this
    // single counter drained at ~350 M/s, i.e. ~70 us after the last neighbourhood was written (most of the kernel at
    // 25 k events, a quarter of it at 800 k).  Reduced in LDS first.
    if (l == 0 && edges_acc) atomicAdd(&blk_edges, (unsigned long long)edges_acc);
    __syncthreads();
    if (threadIdx.x == 0 && blk_edges)
        atomicAdd(reinterpret_cast<unsigned long long *>(status + 2), blk_edges);
}

// ---------------------------------------------------------------------------------------------
// reference-shaped edge_index (event order, graph/utils.py:22) from the slot-space neighbour lists
__global__ __launch_bounds__(kBlock) void k_deg_by_event(int N, const int32_t *__restrict__ ev_slot,
                                                        const int32_t *__restrict__ deg,
                                                        int32_t *__restrict__ rowptr) {
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e > N) return;
    int d = 0;
    if (e < N) { const int s = ev_slot[e]; d = s >= 0 ? deg[s] : 1; }
    rowptr[e] = d;
}

__global__ __launch_bounds__(kBlock) void k_edge_index(int N, int K, const int32_t *__restrict__ ev_slot,
                                                      const int2 *__restrict__ slot_it,
                                                      const int32_t *__restrict__ nbr_src,
                                                      const int32_t *__restrict__ deg,
                                                      const int32_t *__restrict__ rowptr,
                                                      int64_t *__restrict__ edge_index, int64_t row_stride) {
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int e = (int)(gid / K);
    const int j = (int)(gid % K);
    if (e >= N) return;
    const int s = ev_slot[e];
    const int d = s >= 0 ? deg[s] : 1;
    if (j >= d) return;
    const int64_t o = (int64_t)rowptr[e] + j;
    if (o >= row_stride) return;
    edge_index[o] = s >= 0 ? slot_it[nbr_src[(int64_t)s * K + j]].x : e;
    edge_index[row_stride + o] = e;
}

// the same rows as k_edge_index, as what the convolutions consume: source event id and offset code per edge
// (dx + bias) | (dy + bias) << 16 with (dx, dy) = pixel of the source - pixel of the destination
__global__ __launch_bounds__(kBlock) void k_csr_codes(int N, int K, int side, int r, int bias,
                                                     const int32_t *__restrict__ ev_slot,
                                                     const int2 *__restrict__ slot_it,
                                                     const int32_t *__restrict__ nbr_src,
                                                     const int16_t *__restrict__ nbr_code,
                                                     const int32_t *__restrict__ deg,
                                                     const int32_t *__restrict__ rowptr, int32_t *__restrict__ col,
                                                     int32_t *__restrict__ code, int64_t e_cap) {
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int e = (int)(gid / K);
    const int j = (int)(gid % K);
    if (e >= N) return;
    const int s = ev_slot[e];
    const int d = s >= 0 ? deg[s] : 1;
    if (j >= d) return;
    const int64_t o = (int64_t)rowptr[e] + j;
    if (o >= e_cap) return;
    int src = e, dx = 0, dy = 0;                 // an event outside the sensor keeps its self loop only
    if (s >= 0) {
        src = slot_it[nbr_src[(int64_t)s * K + j]].x;
        const int c = nbr_code[(int64_t)s * K + j];
        dx = c / side - r;
        dy = c % side - r;
    }
    col[o] = src;
    code[o] = (dx + bias) | ((dy + bias) << 16);
}

__global__ __launch_bounds__(kBlock) void k_node_order(int N, const int2 *__restrict__ slot_it,
                                                      const int32_t *__restrict__ ev_slot,
                                                      const int32_t *__restrict__ m_ptr,
                                                      int32_t *__restrict__ slot_event,
                                                      int32_t *__restrict__ event_slot) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    if (slot_event) slot_event[i] = i < *m_ptr ? slot_it[i].x : -1;
    if (event_slot) event_slot[i] = ev_slot[i];
}

__global__ __launch_bounds__(kBlock) void k_gather_inputs(const int32_t *__restrict__ m_ptr, int N,
                                                         const int2 *__restrict__ slot_it,
                                                         const int32_t *__restrict__ slot_xyb, GatherArgs g) {
    const int n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N || n >= *m_ptr) return;
    gather_node(g, n, slot_it, slot_xyb);
}

// the caller's window -> the engine's static input buffers + the event count in device memory (captured-graph mode), and K1
// of the graph build (denormalise + per-pixel count) on the way: the launch in front of the captured window does what the
// window's first launch would (one launch less on the window's dependent chain).  The builder's status words are cleared
// here, so what K1 has to report goes to two words of its own (status[8], status[9]; k_scatter moves them over).
template <typename BatchT>
__global__ __launch_bounds__(kBlock) void k_stage_window(const float *__restrict__ pos, const float *__restrict__ feat,
                                                        const BatchT *__restrict__ batch, int N,
                                                        float *__restrict__ pos_out, float *__restrict__ feat_out,
                                                        int32_t *__restrict__ batch_out, int32_t *__restrict__ n_dev,
                                                        int32_t *__restrict__ status8, int W, int H, int B, float fT,
                                                        int32_t *__restrict__ cnt, int32_t *__restrict__ ev_xyb,
                                                        int32_t *__restrict__ ev_t, int32_t *__restrict__ ev_rank) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i == 0) *n_dev = N;
    if (i < 8) status8[i] = 0;     // the builder's status words start over (dagr_graph_build_window_dev)
    if (i < N) {
        feat_out[i] = feat[i];
        batch_out[i] = (int32_t)batch[i];
        count_event<BatchT, false>(i, pos, batch, W, H, B, (float)W, (float)H, fT, cnt, ev_xyb, ev_t, ev_rank, status8 + 8,
                                   status8 + 9);
    }
    if (i < 3 * N) pos_out[i] = pos[i];
    if (i + gridDim.x * kBlock < 3 * N) pos_out[i + gridDim.x * kBlock] = pos[i + gridDim.x * kBlock];
    if (i + 2 * gridDim.x * kBlock < 3 * N) pos_out[i + 2 * gridDim.x * kBlock] = pos[i + 2 * gridDim.x * kBlock];
}

__global__ void k_format_events(const int16_t *__restrict__ xy, const int32_t *__restrict__ t,
                                const int8_t *__restrict__ p, int64_t N, float fW, float fH, float fT,
                                float *__restrict__ pos, float *__restrict__ feat) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    pos[3 * i + 0] = (float)xy[2 * i + 0] / fW;  // IEEE fp32 division (buffers.py:43)
    pos[3 * i + 1] = (float)xy[2 * i + 1] / fH;
    pos[3 * i + 2] = (float)t[i] / fT;
    feat[i] = (float)p[i];
}

}  // namespace
}  // namespace dagr

using namespace dagr;

namespace dagr {
// views into the builder workspace for the level-0 pooling kernel (pooling.hip)
void graph_ws_views(const dagr_graph_desc *desc, void *workspace, const int32_t **start, const int2 **slot_it) {
    GraphWs ws;
    carve(*desc, (char *)workspace, &ws);
    *start = ws.start;
    *slot_it = ws.slot_it;
}
const int32_t *graph_ws_slot_xyb(const dagr_graph_desc *desc, void *workspace) {
    GraphWs ws;
    carve(*desc, (char *)workspace, &ws);
    return ws.slot_xyb;
}
const int32_t *graph_ws_node_count(const dagr_graph_desc *desc, void *workspace) {
    GraphWs ws;
    carve(*desc, (char *)workspace, &ws);
    return ws.start + ws.P;
}
}  // namespace dagr

extern "C" {

int dagr_format_events(const int16_t *xy, const int32_t *t, const int8_t *p, int64_t N, int32_t width,
                       int32_t height, int32_t time_window, float *pos_out, float *feat_out, void *stream) {
    DAGR_CHECK_ARG(N >= 0, "N < 0");
    if (N == 0) return DAGR_OK;
    DAGR_CHECK_ARG(xy && t && p && pos_out && feat_out, "NULL pointer");
    k_format_events<<<(unsigned)ceil_div(N, kBlock), kBlock, 0, (hipStream_t)stream>>>(
        xy, t, p, N, (float)width, (float)height, (float)time_window, pos_out, feat_out);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

size_t dagr_graph_workspace_bytes(const dagr_graph_desc *desc) {
    if (validate(desc) != DAGR_OK) return 0;
    return carve(*desc, nullptr, nullptr);
}

int dagr_graph_workspace_init(const dagr_graph_desc *desc, void *workspace, size_t workspace_bytes, void *stream) {
    int rc = validate(desc);
    if (rc != DAGR_OK) return rc;
    DAGR_CHECK_ARG(workspace != nullptr, "workspace is NULL");
    GraphWs ws;
    const size_t need = carve(*desc, (char *)workspace, &ws);
    if (workspace_bytes < need) {
        set_error("dagr_graph_workspace_init: workspace too small");
        return DAGR_ERR_WORKSPACE;
    }
    DAGR_CHECK_HIP(hipMemsetAsync(ws.cnt, 0, (ws.P + 1 + 8) * 4, (hipStream_t)stream));
    DAGR_CHECK_HIP(hipMemsetAsync(ws.start, 0, (ws.P + 1 + 8) * 4, (hipStream_t)stream));
    DAGR_CHECK_HIP(hipMemsetAsync(ws.status, 0, 16 * 4, (hipStream_t)stream));
    DAGR_CHECK_HIP(hipMemsetAsync(ws.scan_tmp, 0, scan_chained_state_bytes(ws.P + 1), (hipStream_t)stream));
    return DAGR_OK;
}

static int launch_search(const dagr_graph_desc *desc, const GraphWs &ws, int64_t N, int32_t *nbr_src, int16_t *nbr_code,
                         int32_t *deg, hipStream_t stream, const GatherArgs *gather = nullptr) {
    const GatherArgs ga = gather ? *gather : GatherArgs{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
    const int W = desc->width, H = desc->height;
    const unsigned gS = (unsigned)ceil_div(N * 16, kBlock);
    if (2 * desc->radius + 2 <= 16) {
        // fast path: candidate-centric row kernel; dense neighbourhoods are deferred (list in ev_rank, which
        // is dead after k_scatter; counter in status[5]) to the position-centric tiled kernel
        // (the deferral list lives in ev_rank, dead after k_scatter; its counter is status[5])
        constexpr size_t rows_lds = (size_t)(kBlock / 16) * kRowCap * 4;
        static const unsigned res_tiled = persistent_grid(k_search_tiled, kBlock, 0, 1 << 30);
        // builder knob DAGR_ROWS_VARIANT = 10 * rounds + waves per SIMD (16 candidates per round); default 47
        static const int variant = [] { const char *e = getenv("DAGR_ROWS_VARIANT"); return e ? atoi(e) : 47; }();
        auto launch_rows = [&](auto kern) {
            static thread_local unsigned res_rows = 0;
            if (!res_rows) res_rows = persistent_grid(kern, kBlock, rows_lds, 1 << 30);
            const unsigned gR = round_grid8(std::min<int64_t>(ceil_div(N * 16, kBlock), res_rows));
            kern<<<gR, kBlock, rows_lds, stream>>>(
                ws.start + ws.P, W, H, desc->max_neighbors, desc->radius, (float)desc->delta_t_us, ws.slot_xyb, ws.start,
                ws.slot_it, nbr_src, nbr_code, deg, ws.status, ws.ev_rank, ws.status + 5, nullptr, nullptr);
        };
        switch (variant) {
            case 57: launch_rows(k_search_rows<kRowCap, false, 5, 7>); break;
            case 46: launch_rows(k_search_rows<kRowCap, false, 4, 6>); break;
            case 66: launch_rows(k_search_rows<kRowCap, false, 6, 6>); break;
            case 85: launch_rows(k_search_rows<kRowCap, false, 8, 5>); break;
            default: launch_rows(k_search_rows<kRowCap, false, 4, 7>); break;
        }
        DAGR_CHECK_LAUNCH();
        const unsigned gT = round_grid8(std::min<int64_t>(ceil_div(N * 16, kBlock), res_tiled));
        k_search_tiled<<<gT, kBlock, 0, stream>>>(ws.start + ws.P, W, H, desc->max_neighbors, desc->queue_size,
                                                  desc->radius, (float)desc->delta_t_us, ws.slot_xyb, ws.start,
                                                  ws.slot_it, nbr_src, nbr_code, deg, ws.status, ws.ev_rank,
                                                  ws.status + 5, ga);
    } else {
        k_search<<<gS, kBlock, 0, stream>>>(ws.start + ws.P, W, H, desc->max_neighbors, desc->queue_size, desc->radius,
                                            (float)desc->delta_t_us, ws.slot_xyb, ws.start, ws.slot_it, nbr_src,
                                            nbr_code, deg, ws.status);
        if (gather) {
            DAGR_CHECK_LAUNCH();
            k_gather_inputs<<<(unsigned)ceil_div(N, kBlock), kBlock, 0, stream>>>(ws.start + ws.P, (int)N, ws.slot_it,
                                                                                 ws.slot_xyb, ga);
        }
    }
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

static int build_window(const dagr_graph_desc *desc, void *workspace, const void *pos, int32_t pos_is_int32,
                        const void *batch, int32_t batch_is_int64, int64_t N, const int32_t *n_dev, int32_t *nbr_src,
                        int16_t *nbr_code, int32_t *deg, void *stream_, const GatherArgs *gather = nullptr) {
    int rc = validate(desc);
    if (rc != DAGR_OK) return rc;
    DAGR_CHECK_ARG(workspace != nullptr, "workspace is NULL");
    DAGR_CHECK_ARG(N >= 0 && N <= desc->max_events, "N exceeds desc.max_events");
    hipStream_t stream = (hipStream_t)stream_;
    GraphWs ws;
    carve(*desc, (char *)workspace, &ws);
    if (N == 0) {  // ev_graph.py:70-71: empty window, no edges
        DAGR_CHECK_HIP(hipMemsetAsync(ws.status, 0, 8 * 4, stream));
        return DAGR_OK;
    }
    DAGR_CHECK_ARG(pos && batch && nbr_src && nbr_code && deg, "NULL pointer");
    const int n = (int)N;
    const unsigned gN = (unsigned)ceil_div(N, kBlock);
    const int W = desc->width, H = desc->height, B = desc->batch_size;
    // (device-count form: dagr_stage_window, the launch in front of the captured window, has cleared the status words and
    // run K1 on the window it staged)
    if (!n_dev) {
        DAGR_CHECK_HIP(hipMemsetAsync(ws.status, 0, 8 * 4, stream));
#define DAGR_LAUNCH_COUNT(BT, IP)                                                                          \
    k_count<BT, IP><<<gN, kBlock, 0, stream>>>(pos, (const BT *)batch, n, W, H, B, (float)W, (float)H,        \
                                               (float)desc->time_window, ws.cnt, ws.ev_xyb, ws.ev_t,         \
                                               ws.ev_rank, ws.status)
        if (batch_is_int64) { if (pos_is_int32) DAGR_LAUNCH_COUNT(int64_t, true); else DAGR_LAUNCH_COUNT(int64_t, false); }
        else                { if (pos_is_int32) DAGR_LAUNCH_COUNT(int32_t, true); else DAGR_LAUNCH_COUNT(int32_t, false); }
#undef DAGR_LAUNCH_COUNT
        DAGR_CHECK_LAUNCH();
    }
    // start = exclusive_scan(cnt); cnt is re-zeroed in the same pass (invariant for the next window)
    DAGR_CHECK_HIP(exclusive_scan_i32_chained(ws.cnt, ws.start, ws.P + 1, ws.scan_tmp, true, stream));
    k_scatter<<<gN, kBlock, 0, stream>>>(n, n_dev, W, H, ws.ev_xyb, ws.ev_rank, ws.start, ws.slot_tmp, ws.ev_slot, ws.status);
    DAGR_CHECK_LAUNCH();
    // number of occupied CSR slots M = start[P] <= N (dropped events excluded); slots are a
    // prefix [0, M) so launching N threads with an in-kernel bound read would need M on the host.
    // Out-of-FOV events are an error condition; we order all N slots but guard on start[P].
    const int long_cap = (int)(desc->max_events / kShortSeg + 1);
    k_order<<<gN, kBlock, 0, stream>>>(n, ws.P, W, H, desc->queue_size, ws.ev_xyb, ws.ev_t, ws.start, ws.slot_tmp, ws.slot_it,
                                       ws.slot_xyb, ws.ev_slot, ws.long_list, long_cap, ws.status);
    DAGR_CHECK_LAUNCH();
    k_order_long<<<64, kBlock, 0, stream>>>(desc->queue_size, ws.ev_xyb, ws.slot_xyb, ws.ev_slot, ws.ev_t, ws.start,
                                            ws.slot_tmp, ws.slot_it,
                                            ws.long_list, long_cap, ws.status);
    DAGR_CHECK_LAUNCH();
    return launch_search(desc, ws, N, nbr_src, nbr_code, deg, stream, gather);
}

int dagr_graph_build_window(const dagr_graph_desc *desc, void *workspace, const void *pos, int32_t pos_is_int32,
                            const void *batch, int32_t batch_is_int64, int64_t N, int32_t *nbr_src, int16_t *nbr_code,
                            int32_t *deg, void *stream) {
    return build_window(desc, workspace, pos, pos_is_int32, batch, batch_is_int64, N, nullptr, nbr_src, nbr_code, deg, stream);
}

int dagr_graph_build_window_dev(const dagr_graph_desc *desc, void *workspace, const void *pos, int32_t pos_is_int32,
                                const void *batch, int32_t batch_is_int64, int64_t n_cap, const int32_t *n_dev,
                                int32_t *nbr_src, int16_t *nbr_code, int32_t *deg, void *stream) {
    DAGR_CHECK_ARG(n_dev != nullptr && n_cap > 0, "n_dev is NULL / empty capacity");
    return build_window(desc, workspace, pos, pos_is_int32, batch, batch_is_int64, n_cap, n_dev, nbr_src, nbr_code, deg, stream);
}

int dagr_graph_build_window_inputs(const dagr_graph_desc *desc, void *workspace, const float *pos, const void *batch,
                                   int32_t batch_is_int64, int64_t N, const int32_t *n_dev, int32_t *nbr_src,
                                   int16_t *nbr_code, int32_t *deg, const dagr_l0_inputs *in, void *stream) {
    DAGR_CHECK_ARG(in != nullptr, "inputs is NULL");
    if (N == 0) return build_window(desc, workspace, pos, 0, batch, batch_is_int64, N, n_dev, nbr_src, nbr_code, deg, stream);
    DAGR_CHECK_ARG(pos && in->feat && in->pos_nodes && in->batch_nodes && in->x0 && in->ldx0 >= in->col_pos + 2 &&
                       in->col_pos >= 0 && in->col_feat >= 0 && in->col_feat < in->ldx0 && in->col_feat != in->col_pos &&
                       in->col_feat != in->col_pos + 1, "bad level-0 input description");
    const GatherArgs ga{pos, in->feat, in->pos_nodes, in->batch_nodes, in->x0, in->ldx0, in->col_feat, in->col_pos};
    return build_window(desc, workspace, pos, 0, batch, batch_is_int64, N, n_dev, nbr_src, nbr_code, deg, stream, &ga);
}

const int32_t *dagr_graph_node_count_ptr(const dagr_graph_desc *desc, void *workspace) {
    if (validate(desc) != DAGR_OK || workspace == nullptr) return nullptr;
    return graph_ws_node_count(desc, workspace);
}

int dagr_stage_window(const dagr_graph_desc *desc, void *workspace, const float *pos, const float *feat, const void *batch,
                      int32_t batch_is_int64, int64_t N, float *pos_out, float *feat_out, int32_t *batch_out,
                      int32_t *n_dev, void *stream) {
    int rc = validate(desc);
    if (rc != DAGR_OK) return rc;
    DAGR_CHECK_ARG(workspace && N >= 0 && N <= desc->max_events && pos_out && feat_out && batch_out && n_dev, "bad arguments");
    DAGR_CHECK_ARG(N == 0 || (pos && feat && batch), "NULL input");
    GraphWs ws;
    carve(*desc, (char *)workspace, &ws);
    const unsigned grid = (unsigned)std::max<int64_t>(1, ceil_div(N, kBlock));
    const int W = desc->width, H = desc->height, B = desc->batch_size;
    if (batch_is_int64)
        k_stage_window<int64_t><<<grid, kBlock, 0, (hipStream_t)stream>>>(
            pos, feat, (const int64_t *)batch, (int)N, pos_out, feat_out, batch_out, n_dev, ws.status, W, H, B,
            (float)desc->time_window, ws.cnt, ws.ev_xyb, ws.ev_t, ws.ev_rank);
    else
        k_stage_window<int32_t><<<grid, kBlock, 0, (hipStream_t)stream>>>(
            pos, feat, (const int32_t *)batch, (int)N, pos_out, feat_out, batch_out, n_dev, ws.status, W, H, B,
            (float)desc->time_window, ws.cnt, ws.ev_xyb, ws.ev_t, ws.ev_rank);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

int dagr_graph_search_window(const dagr_graph_desc *desc, void *workspace, int64_t N, int32_t *nbr_src, int16_t *nbr_code,
                             int32_t *deg, void *stream_) {
    int rc = validate(desc);
    if (rc != DAGR_OK) return rc;
    DAGR_CHECK_ARG(workspace != nullptr, "workspace is NULL");
    DAGR_CHECK_ARG(N >= 0 && N <= desc->max_events, "N exceeds desc.max_events");
    if (N == 0) return DAGR_OK;
    DAGR_CHECK_ARG(nbr_src && nbr_code && deg, "NULL pointer");
    hipStream_t stream = (hipStream_t)stream_;
    GraphWs ws;
    carve(*desc, (char *)workspace, &ws);
    // edge counter (status[2..3]) and deferral list length (status[5]) start over; the pixel index stays
    DAGR_CHECK_HIP(hipMemsetAsync(ws.status + 2, 0, 2 * 4, stream));
    DAGR_CHECK_HIP(hipMemsetAsync(ws.status + 5, 0, 4, stream));
    return launch_search(desc, ws, N, nbr_src, nbr_code, deg, stream);
}

int dagr_graph_status(const dagr_graph_desc *desc, void *workspace, int64_t *num_edges, int32_t *flags,
                      void *stream) {
    int rc = validate(desc);
    if (rc != DAGR_OK) return rc;
    DAGR_CHECK_ARG(workspace != nullptr, "workspace is NULL");
    GraphWs ws;
    carve(*desc, (char *)workspace, &ws);
    int32_t h[8];
    DAGR_CHECK_HIP(hipMemcpyAsync(h, ws.status, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream));
    DAGR_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (num_edges) *num_edges = (int64_t)((uint32_t)h[2]) | ((int64_t)h[3] << 32);
    if (flags) *flags = h[1];
    return DAGR_OK;
}

int dagr_graph_counters(const dagr_graph_desc *desc, void *workspace, int32_t *out8_host, void *stream) {
    int rc = validate(desc);
    if (rc != DAGR_OK) return rc;
    DAGR_CHECK_ARG(workspace != nullptr && out8_host != nullptr, "NULL pointer");
    GraphWs ws;
    carve(*desc, (char *)workspace, &ws);
    DAGR_CHECK_HIP(hipMemcpyAsync(out8_host, ws.status, 32, hipMemcpyDeviceToHost, (hipStream_t)stream));
    DAGR_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return DAGR_OK;
}

size_t dagr_scan_scratch_elems(int64_t n) { return scan_scratch_elems(n); }

int dagr_spiral_offsets(int32_t n, int32_t *dx_host, int32_t *dy_host) {
    DAGR_CHECK_ARG(n >= 0 && dx_host && dy_host, "bad arguments");
    for (int s = 0; s < n; s++) {
        int sx, sy;
        spiral_offset(s, sx, sy);
        dx_host[s] = sx;
        dy_host[s] = sy;
    }
    return DAGR_OK;
}

int dagr_graph_edge_index(const dagr_graph_desc *desc, void *workspace, const int32_t *nbr_src, const int32_t *deg,
                          int64_t N, int32_t *rowptr, int32_t *scan_scratch, int64_t *edge_index, int64_t row_stride,
                          void *stream_) {
    int rc = validate(desc);
    if (rc != DAGR_OK) return rc;
    DAGR_CHECK_ARG(workspace && N >= 0 && rowptr && scan_scratch, "bad arguments");
    hipStream_t stream = (hipStream_t)stream_;
    if (N == 0) {
        DAGR_CHECK_HIP(hipMemsetAsync(rowptr, 0, 4, stream));
        return DAGR_OK;
    }
    DAGR_CHECK_ARG(nbr_src && deg, "NULL pointer");
    GraphWs ws;
    carve(*desc, (char *)workspace, &ws);
    const int K = desc->max_neighbors;
    // rowptr[0..N] = exclusive scan of the per-event in-degree (event order)
    k_deg_by_event<<<(unsigned)ceil_div(N + 1, kBlock), kBlock, 0, stream>>>((int)N, ws.ev_slot, deg, rowptr);
    DAGR_CHECK_LAUNCH();
    DAGR_CHECK_HIP(exclusive_scan_i32(rowptr, rowptr, N + 1, scan_scratch, false, stream));
    if (edge_index) {
        k_edge_index<<<(unsigned)ceil_div(N * K, kBlock), kBlock, 0, stream>>>((int)N, K, ws.ev_slot, ws.slot_it,
                                                                              nbr_src, deg, rowptr, edge_index,
                                                                              row_stride);
        DAGR_CHECK_LAUNCH();
    }
    return DAGR_OK;
}

int dagr_graph_csr_codes(const dagr_graph_desc *desc, void *workspace, const int32_t *nbr_src, const int16_t *nbr_code,
                         const int32_t *deg, int64_t N, int32_t code_bias, int32_t *rowptr, int32_t *scan_scratch,
                         int32_t *col, int32_t *code, int64_t e_cap, void *stream_) {
    int rc = validate(desc);
    if (rc != DAGR_OK) return rc;
    DAGR_CHECK_ARG(workspace && N >= 0 && rowptr && scan_scratch && code_bias >= 0 && code_bias < (1 << 15), "bad arguments");
    hipStream_t stream = (hipStream_t)stream_;
    if (N == 0) {
        DAGR_CHECK_HIP(hipMemsetAsync(rowptr, 0, 4, stream));
        return DAGR_OK;
    }
    DAGR_CHECK_ARG(nbr_src && nbr_code && deg && col && code && e_cap >= 0, "NULL pointer");
    GraphWs ws;
    carve(*desc, (char *)workspace, &ws);
    const int K = desc->max_neighbors;
    k_deg_by_event<<<(unsigned)ceil_div(N + 1, kBlock), kBlock, 0, stream>>>((int)N, ws.ev_slot, deg, rowptr);
    DAGR_CHECK_LAUNCH();
    DAGR_CHECK_HIP(exclusive_scan_i32(rowptr, rowptr, N + 1, scan_scratch, false, stream));
    k_csr_codes<<<(unsigned)ceil_div(N * K, kBlock), kBlock, 0, stream>>>((int)N, K, 2 * desc->radius + 1, desc->radius,
                                                                         code_bias, ws.ev_slot, ws.slot_it, nbr_src,
                                                                         nbr_code, deg, rowptr, col, code, e_cap);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

int dagr_graph_node_order(const dagr_graph_desc *desc, void *workspace, int64_t N, int32_t *slot_event,
                          int32_t *event_slot, void *stream) {
    int rc = validate(desc);
    if (rc != DAGR_OK) return rc;
    DAGR_CHECK_ARG(workspace && N >= 0, "bad arguments");
    if (N == 0) return DAGR_OK;
    GraphWs ws;
    carve(*desc, (char *)workspace, &ws);
    k_node_order<<<(unsigned)ceil_div(N, kBlock), kBlock, 0, (hipStream_t)stream>>>((int)N, ws.slot_it, ws.ev_slot,
                                                                                 ws.start + ws.P, slot_event,
                                                                                 event_slot);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

int dagr_graph_gather_inputs(const dagr_graph_desc *desc, void *workspace, const float *pos, const float *feat,
                             int64_t N, float *pos_nodes, int32_t *batch_nodes, float *x0, int32_t ldx0,
                             int32_t col_feat, int32_t col_pos, void *stream) {
    int rc = validate(desc);
    if (rc != DAGR_OK) return rc;
    DAGR_CHECK_ARG(workspace && N >= 0, "bad arguments");
    if (N == 0) return DAGR_OK;
    DAGR_CHECK_ARG(pos && feat && pos_nodes && batch_nodes && x0 && ldx0 >= col_pos + 2 && col_pos >= 0 &&
                       col_feat >= 0 && col_feat < ldx0 && col_feat != col_pos && col_feat != col_pos + 1, "bad arguments");
    GraphWs ws;
    carve(*desc, (char *)workspace, &ws);
    const GatherArgs ga{pos, feat, pos_nodes, batch_nodes, x0, ldx0, col_feat, col_pos};
    k_gather_inputs<<<(unsigned)ceil_div(N, kBlock), kBlock, 0, (hipStream_t)stream>>>(ws.start + ws.P, (int)N, ws.slot_it,
                                                                                     ws.slot_xyb, ga);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

}  // extern "C"
