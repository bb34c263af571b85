// graph_build.hip -- window (reset=True) event-graph builder for gfx950.
//
// What the reference does per window (src/dagr/graph/ev_graph.py:52-103, graph/utils.py:6-23,
// ev_graph.cu:15-80,169-212): refill a B x 128 x H x W int32 FIFO volume with -1 (157 MB/sample at
// 640x480), sort events by pixel, rewrite one 128-deep FIFO column per active pixel, then one
// thread per event walks a (2r+1)^2 spiral of FIFO columns, and a boolean-mask pass compacts a
// -1-padded int64 edge buffer.
//
// What this file does instead (same edge set, same order, bit-exact):
//   * events are bucketed by the key (sample, y, x) into a CSR array (per-key counters -> exclusive scan -> scatter ->
//     in-segment order fix-up).  In slot order the events of one pixel row of the neighbourhood are ONE contiguous range.
//     After all N events of a reset window are inserted, the FIFO column of a pixel holds exactly the newest min(count, Q)
//     events of that pixel, newest first -- its segment read backwards; the 157 MB volume and its refill disappear.
//     (Rounds 5 carried an optional time dimension in the key -- buckets delta_t wide, so that a destination fetched
//     2 x (2r+1) row ranges holding ~0.4 of its candidates.  Measured there, profiles/r5_search_buckets.md: the search is
//     bound by its scattered offset loads, which the second bucket doubles; only dense UNIFORM streams gained, clustered
//     ones -- what recordings look like -- lost.  Removed in round 6.)
//   * the search is candidate-centric: the row ranges' events are tested and keyed by (spiral rank of their pixel,
//     recency); the reference's sequential "first K in spiral order, newest first inside a pixel" cut is "the K-1
//     smallest keys in key order" (k_search_rows, 16 lanes per destination; crowded neighbourhoods in their inner rings
//     first).  Event-dense neighbourhoods (> 320 candidates) take the reference's position-centric walk on the same index
//     (k_search_dense); unsorted timestamps and radii beyond 7 pixels its generic form.
//   * output is a fixed-stride neighbour list [N, K] (int32 source + int16 offset code) + deg[N]:
//     no -1 fill, no compaction pass, no host sync; the offset code is the SplineConv LUT index.
#include "common.hpp"

#include <algorithm>
#include <stdlib.h>
#include <type_traits>

namespace dagr {
namespace {

constexpr int kVisBit = (int)0x80000000;  // slot_xyb bit 31: among the newest Q events of its pixel
constexpr int kShortSeg = 64;    // segments up to this length are ordered by per-slot rank counting
constexpr int kMaxQueue = 1024;  // LDS staging bound for the long-segment path
constexpr int kMaxSpiral = 4096; // (2r+1)^2 bound for the LDS spiral tables (r <= 31)

// 16-lane rows by DPP (a lane group of the search kernels is one DPP row): no LDS round trip, one VALU per step
template <int CTRL>
__device__ __forceinline__ int dpp_row(int v) {      // bound_ctrl: lanes without a source read 0
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ int row16_inclusive_scan(int v) {
    v += dpp_row<0x111>(v);     // row_shr:1
    v += dpp_row<0x112>(v);
    v += dpp_row<0x114>(v);
    v += dpp_row<0x118>(v);
    return v;
}
__device__ __forceinline__ int row16_sum(int v) {    // every lane of the row gets the row's total
    v += dpp_row<0x128>(v);     // row_ror:8
    v += dpp_row<0x124>(v);
    v += dpp_row<0x122>(v);
    v += dpp_row<0x121>(v);
    return v;
}

struct GraphWs {
    int32_t *cnt;       // [PK+1]  per-key event counters; all-zero between builds (invariant)
    int32_t *start;     // [PK+1]  exclusive scan of cnt
    int32_t *scan_tmp;  // [scan_chained_state_bytes(PK+1) / 4]: ticket, tag and per-tile words of the one-launch scan
    int32_t *ev_xyb;    // [Nmax] x | y<<12 | b<<24  (denormalised ints)
    int32_t *ev_t;      // [Nmax] denormalised timestamp (us)
    int32_t *ev_rank;   // [Nmax] arrival rank inside the key's segment (arbitrary order)
    int32_t *slot_tmp;  // [Nmax] event id per CSR slot, arrival order
    int2 *slot_it;      // [Nmax] {event id, t} per CSR slot, ascending id inside a segment
    int32_t *slot_xyb;  // [Nmax] x | y<<12 | b<<24 per CSR slot
    int32_t *ev_slot;   // [Nmax] CSR slot of every event (-1: dropped)
    int32_t *hot_list;  // [Nmax + 1] pixels with more than min(kShortSeg, Q) events (k_fix_pixels)
    int32_t *status;    // [16]: 0 listed pixels, 1 flags, 2..3 num_edges (uint64), 4 pixels beyond the FIFO depth,
                        //       5 deferral list length, 6 unsorted timestamps, 7 destinations answered from their inner
                        //       rings, 8 / 9 the staging launch's flag words
    int64_t P;          // pixels: B * H * W
    int64_t PK;         // keys of the index (= P: one segment per pixel)
};

size_t carve(const dagr_graph_desc &d, char *base, GraphWs *ws) {
    const int64_t P = (int64_t)d.width * d.height * d.batch_size;
    const int64_t PK = P;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off = align_up(off + bytes, 256);
        return base ? base + o : nullptr;
    };
    int32_t *cnt = (int32_t *)take((PK + 1 + 8) * 4);
    int32_t *start = (int32_t *)take((PK + 1 + 8) * 4);
    int32_t *scan_tmp = (int32_t *)take(scan_chained_state_bytes(PK + 1));
    int32_t *ev_xyb = (int32_t *)take(d.max_events * 4);
    int32_t *ev_t = (int32_t *)take(d.max_events * 4);
    int32_t *ev_rank = (int32_t *)take(d.max_events * 4);
    int32_t *slot_tmp = (int32_t *)take(d.max_events * 4);
    int2 *slot_it = (int2 *)take(d.max_events * 8);
    int32_t *slot_xyb = (int32_t *)take(d.max_events * 4);
    int32_t *ev_slot = (int32_t *)take(d.max_events * 4);
    int32_t *hot_list = (int32_t *)take((d.max_events + 2) * 4);
    int32_t *status = (int32_t *)take(16 * 4);
    if (ws) *ws = GraphWs{cnt, start, scan_tmp, ev_xyb, ev_t, ev_rank, slot_tmp, slot_it, slot_xyb, ev_slot, hot_list, status, P, PK};
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
// K1: denormalise (ev_tgn.py:11-16) + per-key count.  One thread per event.
//   int(pos * [W,H,T] + 1e-3): fp32 multiply, fp32 add (separately rounded -- this TU is built with
//   -ffp-contract=off), truncation toward zero.
// `flags`: where the two conditions an event can raise are recorded -- the builder's status words (flags[1] |= 1: outside
// the sensor / batch; flags[6] = 1: timestamps not sorted), or the staging launch's own two words (see k_stage_window).
template <typename BatchT, bool kIntPos>
__device__ __forceinline__ void count_event(int e, const void *__restrict__ pos_, const BatchT *__restrict__ batch, int W,
                                            int H, int B, float fW, float fH, float fT,
                                            int32_t *__restrict__ cnt,
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
    // time flag: set when timestamps are not non-decreasing in event order inside a sample.  Ids then do not order time:
    // the search takes its generic form (every candidate tested on its own, ids instead of positions as recency).
    if (e > 0 && (int)batch[e - 1] == b) {
        int tp;
        if (kIntPos) tp = static_cast<const int32_t *>(pos_)[3 * (int64_t)(e - 1) + 2];
        else tp = (int)(fT * static_cast<const float *>(pos_)[3 * (int64_t)(e - 1) + 2] + 1e-3f);
        if (tp > t) *flag_time = 1;
    }
    if (x < 0 || x >= W || y < 0 || y >= H || b < 0 || b >= B) {
        // The reference would index its FIFO volume out of bounds here; we flag and drop the
        // event from the index (it keeps its self loop).
        atomicOr(flag_fov, 1);
        ev_xyb[e] = -1;
        ev_rank[e] = 0;
        return;
    }
    ev_xyb[e] = x | (y << 12) | (b << 24);
    const int key = x + W * (y + H * b);
    ev_rank[e] = atomicAdd(&cnt[key], 1);
}

template <typename BatchT, bool kIntPos>
__global__ __launch_bounds__(kBlock) void k_count(const void *__restrict__ pos_, const BatchT *__restrict__ batch,
                                                 int N, int W, int H, int B, float fW,
                                                 float fH, float fT, int32_t *__restrict__ cnt,
                                                 int32_t *__restrict__ ev_xyb, int32_t *__restrict__ ev_t,
                                                 int32_t *__restrict__ ev_rank, int32_t *__restrict__ status, int xcd_remap) {
    const int lb = xcd_block((N + kBlock - 1) / kBlock, xcd_remap);
    const int e = lb * kBlock + threadIdx.x;
    if (lb < 0 || e >= N) return;
    count_event<BatchT, kIntPos>(e, pos_, batch, W, H, B, fW, fH, fT, cnt, ev_xyb, ev_t, ev_rank, status + 1, status + 6);
}

__device__ __forceinline__ int key_of_event(int c, int W, int H) {
    return (c & 4095) + W * (((c >> 12) & 4095) + H * (c >> 24));
}

// K3: scatter event ids into their key's segment (arrival order).
// n_dev: the window's event count in device memory (launches sized for a capacity N: captured HIP graphs); K1 then ran
// inside the staging launch, whose two flag words (status[8], status[9]) this launch moves into the builder's and re-arms.
__global__ __launch_bounds__(kBlock) void k_scatter(int N, const int32_t *__restrict__ n_dev, int W, int H,
                                                   const int32_t *__restrict__ ev_xyb,
                                                   const int32_t *__restrict__ ev_rank,
                                                   const int32_t *__restrict__ start,
                                                   int32_t *__restrict__ slot_tmp, int32_t *__restrict__ ev_slot,
                                                   int32_t *__restrict__ status, int xcd_remap) {
    if (n_dev && blockIdx.x == 0 && threadIdx.x == 0) {
        if (status[8]) { atomicOr(&status[1], 1); status[8] = 0; }
        if (status[9]) { status[6] = 1; status[9] = 0; }
    }
    const int Nw = n_dev ? min(N, *n_dev) : N;       // (a captured launch is sized for the capacity)
    const int lb = xcd_block((Nw + kBlock - 1) / kBlock, xcd_remap);
    const int e = lb * kBlock + threadIdx.x;
    if (lb < 0 || e >= Nw) return;
    const int c = ev_xyb[e];
    if (c < 0) { ev_slot[e] = -1; return; }
    slot_tmp[start[key_of_event(c, W, H)] + ev_rank[e]] = e;
}

// K4: order each segment by ascending event id (== the reference's stable sort by pixel, graph/utils.py:10).  One thread
// per CSR slot; every event starts out visible (kVisBit), and the first slot of a segment that holds more than Q events
// (FIFO depth), or that is too long for the rank counting, lists its pixel for k_fix_pixels.
__global__ __launch_bounds__(kBlock) void k_order(int N, int64_t PK, int W, int H, int hot_thr,
                                                 const int32_t *__restrict__ ev_xyb,
                                                 const int32_t *__restrict__ ev_t,
                                                 const int32_t *__restrict__ start,
                                                 const int32_t *__restrict__ slot_tmp, int2 *__restrict__ slot_it,
                                                 int32_t *__restrict__ slot_xyb, int32_t *__restrict__ ev_slot,
                                                 int32_t *__restrict__ hot_list, int hot_cap,
                                                 int32_t *__restrict__ status, int xcd_remap) {
    const int M = min(N, start[PK]);       // start[PK] = number of indexed events (<= N)
    const int lb = xcd_block((M + kBlock - 1) / kBlock, xcd_remap);
    const int s = lb * kBlock + threadIdx.x;
    if (lb < 0 || s >= M) return;  // start[PK] = number of indexed events (<= N)
    const int e = slot_tmp[s];
    const int c = ev_xyb[e];
    const int t = ev_t[e];
    const int key = key_of_event(c, W, H);
    const int a = start[key];
    const int n = start[key + 1] - a;
    if (n <= kShortSeg) {
        int rank = 0;
        for (int k = 0; k < n; k++) rank += (slot_tmp[a + k] < e) ? 1 : 0;
        slot_it[a + rank] = make_int2(e, t);
        slot_xyb[a + rank] = c | kVisBit;
        ev_slot[e] = a + rank;
    }
    if (s == a && n > hot_thr) {
        const int i = atomicAdd(&status[0], 1);
        if (i < hot_cap) hot_list[i] = key;
        else atomicOr(&status[1], 2);
    }
}

// the k-th largest of a set of (distinct) event ids, by 8-bit radix selection; `each(f)` calls f(id) for this thread's
// share of the set.  All threads of the block call it; hist: int[256]; sh: int[4].  The ids of a pixel lie in a narrow
// range (one sample's events), so the digits are taken from id - min, starting at the range's top bits: the first
// histogram is spread over its 256 bins (on the raw ids every element of a pass over the top byte hit ONE bin -- an
// LDS atomic serialised over all lanes, 0.24 ms for one clipped border pixel of an S-edges window) and a 2^20 range takes
// three passes, not four.
template <typename Each>
__device__ __forceinline__ unsigned select_kth_largest(int k, int *hist, int *sh, Each each) {
    if (threadIdx.x == 0) { sh[0] = 0; sh[1] = k; sh[2] = 0x7fffffff; sh[3] = 0; }
    __syncthreads();
    {
        unsigned lo = 0x7fffffffu, hi = 0u;
        each([&](unsigned v) { lo = min(lo, v); hi = max(hi, v); });
        if (lo <= hi) { atomicMin(&sh[2], (int)lo); atomicMax(&sh[3], (int)hi); }
    }
    __syncthreads();
    const unsigned vmin = (unsigned)sh[2];
    const unsigned range = (unsigned)sh[3] - vmin;
    const int bits = 32 - __clz((int)(range | 1u));
    unsigned mask = 0;
    for (int shift = max(bits - 8, 0);; shift = max(shift - 8, 0)) {
        hist[threadIdx.x] = 0;
        __syncthreads();
        const unsigned prefix = (unsigned)sh[0];
        each([&](unsigned v) {
            const unsigned u = v - vmin;
            if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255], 1);
        });
        __syncthreads();
        if (threadIdx.x == 0) {
            int rem = sh[1], d = 255;
            for (; d > 0; d--) {
                if (hist[d] >= rem) break;
                rem -= hist[d];
            }
            sh[1] = rem;
            sh[0] = (int)(prefix | ((unsigned)d << shift));     // (a digit that overlaps the previous one repeats its bits)
        }
        mask |= 255u << shift;
        __syncthreads();
        if (shift == 0) break;
    }
    return (unsigned)sh[0] + vmin;
}

// K5: the listed pixels.  Only the newest Q events of a pixel are visible to the search (FIFO depth Q, ev_graph.cu:201-211;
// "newest" = largest ids: the FIFO is filled in event order), and segments longer than kShortSeg are ordered here: the
// min(n, Q) largest ids (radix selection of the Q-th largest when n > Q) rank-sorted into the segment's tail and marked
// visible, the rest -- invisible whatever their order: they can never be a source -- copied into its head as they lie.
__global__ __launch_bounds__(kBlock) void k_fix_pixels(int Q, const int32_t *__restrict__ ev_xyb,
                                                      int32_t *__restrict__ slot_xyb, int32_t *__restrict__ ev_slot,
                                                      const int32_t *__restrict__ ev_t,
                                                      const int32_t *__restrict__ start,
                                                      const int32_t *__restrict__ slot_tmp,
                                                      int2 *__restrict__ slot_it,
                                                      const int32_t *__restrict__ hot_list, int hot_cap,
                                                      int32_t *__restrict__ status) {
    __shared__ int hist[256];
    __shared__ int sel[kMaxQueue];
    __shared__ int sh_sel[4], sh_nsel, sh_nrest;
    int n_hot = status[0];
    if (n_hot > hot_cap) n_hot = hot_cap;
    for (int li = blockIdx.x; li < n_hot; li += gridDim.x) {
        const int key = hot_list[li];
        const int a = start[key];
        const int n = start[key + 1] - a;
        const int m = n < Q ? n : Q;               // visible entries
        unsigned thr = 0;                          // the m-th largest id of a long segment
        if (n > Q) {
            if (n > kShortSeg)
                thr = select_kth_largest(Q, hist, sh_sel, [&](auto f) {
                    for (int k = threadIdx.x; k < n; k += kBlock) f((unsigned)slot_tmp[a + k]);
                });
            if (threadIdx.x == 0) atomicAdd(&status[4], 1);
        }
        if (n <= kShortSeg) {                      // ordered by k_order: only the visibility may change
            for (int i = threadIdx.x; i < n - m; i += kBlock) slot_xyb[a + i] &= ~kVisBit;
            continue;
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
            for (int q = 0; q < m; q++) rank += (sel[q] < v) ? 1 : 0;
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

__device__ __forceinline__ int group16_inclusive_scan(int v) {
    const int l = threadIdx.x & 15;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        int n = __shfl_up(v, d, 16);
        if (l >= d) v += n;
    }
    return v;
}

// Level-0 node numbering.  The graph is emitted in *slot space*: node n is CSR slot n, i.e. events ordered by
// (sample, y, x) and by time inside a segment.  All events of a band of pixel rows are one contiguous run of
// nodes (voxel pooling streams its members), destinations that follow each other share the row ranges they fetch, and
// the source rows of a tile of nodes sit in a few narrow runs of memory.  Event-order views (edge_index, permutations) are
// produced on demand by dagr_graph_edge_index / dagr_graph_node_order.

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

// Admissible sources of one segment (one pixel; ids and -- timestamps being sorted -- times ascend along its slots) for
// destination (e, t): "older than the destination" (ev_graph.cu:64) is a prefix [a, hi), "dt <= delta" (ev_graph.cu:69) a
// suffix of it, "inside the FIFO depth" a suffix too: binary searches instead of the reference's newest-first walk over up
// to Q entries.  Returns [lo, hi); the walk order newest-first is hi-1 .. lo.  hot: some pixel of the window holds more
// than Q events (status[4]) -- otherwise every slot is visible and the third search is skipped.
__device__ __forceinline__ void admissible_range(const int2 *__restrict__ slot_it, const int32_t *__restrict__ slot_xyb,
                                                 int a, int b, int e, int t, float delta_t, bool hot, int &lo, int &hi) {
    const int a0 = a, b0 = b;
    if (b0 - a0 == 1) {                       // one entry (most segments): tested directly
        const int2 it = slot_it[a0];
        bool ok = it.x < e && !((float)(t - it.y) > delta_t);
        if (ok && hot) ok = slot_xyb[a0] < 0;
        lo = a0;
        hi = ok ? b0 : a0;
        return;
    }
    while (a < b) {                           // first slot in [a, b) with id >= e
        const int m = (a + b) >> 1;
        if (slot_it[m].x < e) a = m + 1; else b = m;
    }
    hi = a;
    a = a0; b = hi;                           // first slot in [a, hi) with dt <= delta
    while (a < b) {
        const int m = (a + b) >> 1;
        if ((float)(t - slot_it[m].y) > delta_t) a = m + 1; else b = m;
    }
    lo = a;
    if (hot && slot_xyb[a0] >= 0) {           // the segment has invisible entries: its first visible slot
        a = a0 + 1; b = b0;
        while (a < b) {
            const int m = (a + b) >> 1;
            if (slot_xyb[m] >= 0) a = m + 1; else b = m;
        }
        lo = max(lo, a);
    }
    if (lo > hi) lo = hi;
}

// ---------------------------------------------------------------------------------------------
// K6 (r <= 7, timestamps sorted): candidate-centric search.  In slot order the events of one pixel row of the neighbourhood
// are ONE contiguous slot range [start(row, x-r), start(row, x+r+1)), so
//   1. 15 lanes fetch the row ranges (2 offsets each) and a 16-lane scan concatenates them;
//   2. the C candidates are read 16 at a time, coalesced ({id,t} + packed x|y|visible): each lane finds its candidate's row
//      by a binary search over the ranges' bases, tests it (older than the destination, dt <= delta, inside the FIFO depth)
//      and keys it with (spiral rank of its pixel, recency inside the pixel = position in the row);
//   3. the reference's sequential walk "spiral order, newest first, stop at K" is exactly "the K-1 smallest keys in key
//      order": valid candidates are compacted into LDS (ballot/popcount) and each takes the slot given by the number of
//      smaller keys.
// Work is proportional to the events present in the neighbourhood instead of to its (2r+1)^2 pixels.  Neighbourhoods with
// more than `defer_cap` (<= kRowCap) candidates (event-dense scenes, where the position-centric walk exits after the first
// rings anyway) are appended to a list (ev_rank, dead after k_scatter; counter in status[5]) that k_search_dense walks
// afterwards; the list is the kernel's LDS footprint (20 KiB per workgroup).  The kernel is bound by VALU issue (a wave64
// instruction holds the SIMD for four cycles: the measured 0.22 wave-instructions per cycle and SIMD are 87 % of that).
// Round 6 tried the two other shapes of the dense class (profiles/r6_search_sweep.md; both bit-exact, both slower): ONE
// destination per wave over an LDS matrix of all row offsets -- ~800 dependent LDS / bpermute round trips per destination
// -- and a separate list of heavy destinations searched in nested ring windows, 16 lanes each, with prefetched inputs:
// 1.16 ns per heavy destination against 0.83 ns for the walk and 0.74 ns for the ring-limited passes below.  What decides
// is instructions per destination: ~40 per 16 candidates here, ~100 per 16 spiral positions in the walk, which is why the
// walk wins from ~1.4 events per pixel (C > 320) and the candidate form below that.
constexpr int kRowCap = 320;
struct alignas(4) Pair { int a, b; };

// ROUNDS x 16 candidates are requested before the first is examined; WAVES per SIMD = the register budget (512 / WAVES)
template <int CAP, int ROUNDS, int WAVES>
__global__ __launch_bounds__(kBlock, WAVES) void k_search_rows(const int32_t *__restrict__ m_ptr, int W, int H, int K, int r,
                                                       float delta_t, int defer_cap, int ring_thr,
                                                       const int32_t *__restrict__ slot_xyb,
                                                       const int32_t *__restrict__ start,
                                                       const int2 *__restrict__ slot_it,
                                                       int32_t *__restrict__ nbr_src, int16_t *__restrict__ nbr_code,
                                                       int32_t *__restrict__ deg, int32_t *__restrict__ status,
                                                       int32_t *__restrict__ node_list,
                                                       int32_t *__restrict__ node_list_count) {
    constexpr int G = kBlock / 16;
    __shared__ unsigned char sp_rank[256];      // spiral index of offset (dy + r) * 16 + (dx + r)
    __shared__ unsigned short sp_dec[256];      // spiral index -> offset code (dx + r) * side + (dy + r) | (dy + r) << 12
    __shared__ int row_lo[G][16];               // first slot of the 15 row ranges
    __shared__ int row_base[G][17];             // the ranges' positions in the concatenation
    __shared__ int row_sub[G][16];              // ... minus the range's offset inside its row: candidate index -> position in the ROW
    // the group's list: the keys of the admissible candidates, (spiral rank << 20) | (0x7FFFF - position in the row): the
    // source slot follows from the key.  Dynamic LDS on purpose: with the size visible the compiler's occupancy estimate
    // (made against 64 KiB) drops and it stops holding the kernel to its register budget; the hardware has 160 KiB per CU.
    extern __shared__ int v_key_dyn[];
    __shared__ int def_buf[kBlock / 64][64];
    __shared__ unsigned long long blk_edges;   // the block's edge count: ONE global atomic per workgroup at the end
    __shared__ int blk_ring;                   // destinations answered from their inner rings (status[7])
    int wcnt = 0;    // entries of this wave's deferral buffer (uniform over the wave's active lanes)
    const int side = 2 * r + 1;
    const int S = side * side;
    for (int s = threadIdx.x; s < S; s += kBlock) {
        int sx, sy;
        spiral_offset(s, sx, sy);
        sp_rank[(sy + r) * 16 + (sx + r)] = (unsigned char)s;
        sp_dec[s] = (unsigned short)(((sx + r) * side + (sy + r)) | ((sy + r) << 12));
    }
    if (threadIdx.x == 0) { blk_edges = 0ull; blk_ring = 0; }
    __syncthreads();
    const int l = threadIdx.x & 15;
    const int grp = threadIdx.x >> 4;
    int *const v_keys = v_key_dyn + grp * (CAP + 4);       // (+ the rank counting's three sentinels)
    const int gshift = threadIdx.x & 48;
    const unsigned lt_mask = (1u << l) - 1u;
    long long edges_acc = 0;
    int ring_acc = 0;
    const int M = *m_ptr;
    // timestamps out of order: ids do not order time -- k_search_dense takes every node in its generic form
    if (M <= 0 || status[6] != 0) return;
    const int Gd = gridDim.x, nx = (Gd % 8 == 0) ? 8 : 1;
    const int xcd = blockIdx.x % nx, lb = blockIdx.x / nx, bpx = Gd / nx;
    const int chunk = (M + nx - 1) / nx;
    const int per_block = (chunk + bpx - 1) / bpx;
    const int n_begin = xcd * chunk + lb * per_block;
    const int n_end = min(min(M, (xcd + 1) * chunk), n_begin + per_block);
    // Software pipeline over this lane group's destinations: the dependent chain {id,t | x,y,b} -> row bounds ->
    // candidates is three memory latencies; the first two are issued one and two destinations ahead.
    auto load_node = [&](int i, int2 &me, int &c) {
        const int nn = min(i, M - 1);
        me = slot_it[nn];
        c = slot_xyb[nn];
    };
    auto load_rows = [&](int c, int &lo0, int &len0) {
        const int x = c & 4095, y = (c >> 12) & 4095, b = (c >> 24) & 127;
        const int yn = y + l - r;
        lo0 = 0; len0 = 0;
        if (l < side && yn >= 0 && yn < H) {
            const int base = W * (yn + H * b);
            lo0 = start[base + max(x - r, 0)];
            len0 = start[base + min(x + r, W - 1) + 1] - lo0;
        }
    };
    int2 me, me1;
    int c, c1, lo0, len0;
    if (n_begin + grp < n_end) {
        load_node(n_begin + grp, me, c);
        load_node(n_begin + grp + G, me1, c1);
        load_rows(c, lo0, len0);
    }
    for (int ni = n_begin + grp; ni < n_end; ni += G) {
        const int n = ni;
        // next destinations' loads (results are used one iteration later)
        int2 me2;
        int c2, nlo0, nlen0;
        load_node(ni + 2 * G, me2, c2);
        load_rows(c1, nlo0, nlen0);
        const int e = me.x, t = me.y;
        const int x = c & 4095;
        const int64_t row = (int64_t)n * K;
        // 1. row ranges: 15 lanes hold them, a 16-lane scan concatenates them
        const int C = row16_sum(len0);
        const int cb0 = row16_inclusive_scan(len0) - len0;
        const int clo0 = lo0, cl0 = len0;
        const int c_dst = c;
        me = me1; c = c1; me1 = me2; c1 = c2; lo0 = nlo0; len0 = nlen0;   // rotate the pipeline
        // Dense neighbourhoods are deferred to the position-centric kernel.  The list append is aggregated per wave
        // (LDS buffer, one global atomic per ~48 entries): one atomicAdd per destination on a single counter
        // serialises at ~350 M/s and was the whole cost of this kernel on dense windows (4.5 ms at 1.6 M deferrals).
        const bool defer = C > defer_cap;
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
        // Ring limit.  The walk keeps the first K - 1 admissible sources in spiral order and the spiral runs
        // ring by ring (Chebyshev distance), so once the rings <= rho hold K - 1 admissible sources nothing outside them is
        // kept: a neighbourhood with many candidates is searched in its inner (2 rho + 1)^2 window first -- exact when that
        // yields K - 1 sources, otherwise the full window is searched as before (the inner pass is then lost work).
        // Numbers (S-edges 100 k events per sample, 15 x 15 window): 34 % of the destinations have 96 < C <= 320
        // candidates, mean 195; the ring that fills their K - 1 holds 60 (tools/ring_stats.py).  ring_thr: low 16 bits = the
        // candidate count from which this is tried, high bits = candidates wanted per source.
        // The row's range is cut into [left | inner window | right] (slots relative to the row's first one); the passes
        // are: the inner windows of all rows; then, only if those fell short of K - 1 sources, the left parts (whole rows
        // outside the window) and the right parts.  Keys carry positions relative to the row's first slot in every pass, so
        // the passes append to one list and nothing is examined twice.
        int in_off = cl0, in_len = 0;          // (a row outside the window: everything is "left")
        int Cu = C, pb = cb0, p_off = 0;
        bool inner = false;
        {
            if ((ring_thr & 0xffff) > 0 && C > (ring_thr & 0xffff)) {
                const int xd = c_dst & 4095, yd = (c_dst >> 12) & 4095, bd = (c_dst >> 24) & 127;
                const int yn = yd + l - r;
                const bool row_ok = l < side && yn >= 0 && yn < H;
                const int rbase = W * (yn + H * bd);
                auto window = [&](int rho, int &wlo, int &wlen) {
                    wlo = 0; wlen = 0;
                    if (row_ok && l >= r - rho && l <= r + rho) {
                        wlo = start[rbase + max(xd - rho, 0)];
                        wlen = start[rbase + min(xd + rho, W - 1) + 1] - wlo;
                    }
                };
                int wlo, wlen, rho = 3;
                window(3, wlo, wlen);
                int Cw = row16_sum(wlen);
                // enough candidates for K - 1 admissible ones?  (about one in four is: dt <= delta_t, older, visible)
                const int want = (ring_thr >> 16) * (K - 1);
                if (Cw < want && r > 5) {
                    // density of the inner window (or of the whole one, if larger) -> the ring that should hold `want`
                    const int d225 = max(Cw * 225 / 49, C);          // candidates per 225 pixels
                    rho = (d225 * 81 >= want * 225) ? 4 : ((d225 * 121 >= want * 225) ? 5 : 0);
                    if (rho > 0) { window(rho, wlo, wlen); Cw = row16_sum(wlen); } else Cw = 0;
                }
                if (Cw >= want && Cw + 48 <= C) {
                    inner = true;
                    if (row_ok && l >= r - rho && l <= r + rho) { in_off = wlo - clo0; in_len = wlen; }
                    Cu = Cw; p_off = in_off;
                    pb = row16_inclusive_scan(in_len) - in_len;
                }
            }
        }
        int V = 0;
        row_lo[grp][l] = clo0;
        for (int pass = 0;; pass++) {       // one pass, or up to three for a ring-limited destination
        row_base[grp][l] = pb;
        row_sub[grp][l] = pb - p_off;
        if (l == 15) row_base[grp][16] = Cu;
        __builtin_amdgcn_wave_barrier();
        // 2. candidates, 16 per round, ROUNDS rounds of loads in flight.  The admissible ones are compacted into the SAME
        //    list: their number never exceeds the number of candidates read so far, and a batch reads all of its list
        //    entries before it writes any key.
        for (int c0 = 0; c0 < Cu; c0 += 16 * ROUNDS) {
            int2 it[ROUNDS];
            int cxv[ROUNDS], rpv[ROUNDS];     // rpv: (range << 16) | position in the range
#pragma unroll
            for (int q = 0; q < ROUNDS; q++) {
                const int ci = c0 + 16 * q + l;
                it[q] = make_int2(0, 0);
                cxv[q] = 0; rpv[q] = 0;
                if (ci < Cu) {
                    int rr = 0;
                    if (row_base[grp][rr + 8] <= ci) rr += 8;
                    if (row_base[grp][rr + 4] <= ci) rr += 4;
                    if (row_base[grp][rr + 2] <= ci) rr += 2;
                    if (row_base[grp][rr + 1] <= ci) rr += 1;
                    rpv[q] = (rr << 16) | (ci - row_sub[grp][rr]);
                    const int sv = row_lo[grp][rpv[q] >> 16] + (rpv[q] & 0xffff);
                    it[q] = slot_it[sv];
                    cxv[q] = slot_xyb[sv];
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < ROUNDS; q++) {
                if (c0 + 16 * q < Cu) {        // (group-uniform: a round no group of the wave needs is skipped as a whole)
                    const int ci = c0 + 16 * q + l;
                    int valid = 0, key = 0;
                    if (ci < Cu) {
                        // visible in the FIFO; older than the destination (ev_graph.cu:64); dt <= delta (:69)
                        valid = ((cxv[q] < 0) && it[q].x < e && !((float)(t - it[q].y) > delta_t)) ? 1 : 0;
                        const int dx = (cxv[q] & 4095) - x;
                        const int rr = rpv[q] >> 16;
                        const int rank = sp_rank[rr * 16 + (dx + r)];
                        // spiral rank first, then newest first inside the pixel: larger slot = newer inside a row's range
                        key = (rank << 20) | (0x7FFFF - (rpv[q] & 0xffff));
                    }
                    const unsigned bits = (unsigned)(__ballot(valid != 0) >> gshift) & 0xffffu;
                    if (valid) v_keys[V + __popc(bits & lt_mask)] = key;
                    V += __popc(bits);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (!inner || pass == 2) break;
        if (pass == 0 && V >= K - 1) { ring_acc += (l == 0) ? 1 : 0; break; }
        // the inner window fell short: the left parts (whole rows outside the window), then the right parts
        const int p_len = pass == 0 ? in_off : cl0 - (in_off + in_len);
        p_off = pass == 0 ? 0 : in_off + in_len;
        Cu = row16_sum(p_len);
        pb = row16_inclusive_scan(p_len) - p_len;
        }
        // 3. the K-1 smallest keys, in key order
        if (l == 0) {
            nbr_src[row] = n;  // self loop first (ev_graph.cu:44-46)
            nbr_code[row] = (int16_t)(r * side + r);
        }
        auto emit = [&](int mk, int rk) {
            const int dec = sp_dec[mk >> 20];
            nbr_src[row + 1 + rk] = row_lo[grp][dec >> 12] + (0x7FFFF - (mk & 0x7FFFF));   // range start + position
            nbr_code[row + 1 + rk] = (int16_t)(dec & 0xfff);
        };
        // Dense neighbourhoods hold far more admissible candidates than the K-1 that survive, and the rank counting
        // below is quadratic in their number: first drop everything beyond the spiral-rank bin (16 positions per bin,
        // one bin per lane) in which the K-1'th candidate falls.
        if (V > 32) {
            int cntb = 0;
            for (int j = 0; j < V; j++) cntb += ((v_keys[j] >> 24) == l) ? 1 : 0;
            const int incl_b = row16_inclusive_scan(cntb);
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
        // rank = number of smaller keys, four keys per LDS read (the list is padded with three sentinels; measured against
        // 15 DPP rotations of one key per lane: the rotations lose as soon as one group of the wave holds more than 16 keys)
        if (l < 3) v_keys[V + l] = 0x7fffffff;
        __builtin_amdgcn_wave_barrier();
        for (int vi = l; vi < V; vi += 16) {
            const int mk = v_keys[vi];
            int rk = 0;
            for (int j = 0; j < V; j += 4) {
                const int k0 = v_keys[j], k1 = v_keys[j + 1], k2 = v_keys[j + 2], k3 = v_keys[j + 3];
                rk += (k0 < mk ? 1 : 0) + (k1 < mk ? 1 : 0) + (k2 < mk ? 1 : 0) + (k3 < mk ? 1 : 0);
            }
            if (rk < K - 1) emit(mk, rk);
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
    if (ring_acc) atomicAdd(&blk_ring, ring_acc);
    __syncthreads();
    if (threadIdx.x == 0 && blk_edges)
        atomicAdd(reinterpret_cast<unsigned long long *>(status + 2), blk_edges);
    if (threadIdx.x == 0 && blk_ring) atomicAdd(&status[7], blk_ring);
}

// ---------------------------------------------------------------------------------------------
// K6, everything the row kernel left.
// LIST mode (the usual call): the position-centric walk of the reference (ev_graph.cu:48-78) for the deferred destinations,
// 16 lanes per destination, one spiral position per lane and round.  A position is one pixel = one segment; timestamps being
// sorted, the admissible entries of a segment are a slot range found by binary searches (admissible_range), a 16-lane
// prefix sum reproduces the sequential "first K in spiral order" cut, and the walk ends with the round that fills K --
// event-dense neighbourhoods (the ones deferred here: more than kRowCap candidates) end in the first rings.
// GENERIC form (every node of the window; chosen on the device when the timestamps are not sorted, status[6], and by the
// host for radii beyond 7 pixels): 16 lanes per destination, candidate-centric over all (2r+1) row ranges, every candidate
// tested on its own, keyed by (spiral rank, id descending) in 64 bits: exact whatever the order of the timestamps, with the
// K - 1 smallest keys kept in a small LDS list that is compacted as it fills.  Slow and rare.
__host__ __device__ inline int dense_gen_cap(int K) { return (K - 1 + 16 + 15) / 16 * 16; }
__host__ __device__ inline size_t dense_lds_bytes(int K, int r) {
    const int S = (2 * r + 1) * (2 * r + 1);
    return (size_t)(kBlock / 16) * 2 * dense_gen_cap(K) * 12 + (size_t)((S + 3) / 4 * 4) * 4;
}

__global__ __launch_bounds__(kBlock) void k_search_dense(const int32_t *__restrict__ m_ptr, int W, int H, int K, int r,
                                                        float delta_t,
                                                        const int32_t *__restrict__ slot_xyb,
                                                        const int32_t *__restrict__ start,
                                                        const int2 *__restrict__ slot_it,
                                                        int32_t *__restrict__ nbr_src,
                                                        int16_t *__restrict__ nbr_code, int32_t *__restrict__ deg,
                                                        int32_t *__restrict__ status,
                                                        const int32_t *__restrict__ node_list,
                                                        const int32_t *__restrict__ node_list_count, int all_nodes,
                                                        GatherArgs gather) {
    constexpr int G = kBlock / 16;
    extern __shared__ __align__(8) unsigned char dense_lds[];
    if (gather.pos) {       // (independent of the search: the index is final since k_fix_pixels)
        const int m = *m_ptr;
        for (int n = blockIdx.x * kBlock + threadIdx.x; n < m; n += gridDim.x * kBlock) gather_node(gather, n, slot_it, slot_xyb);
    }
    const bool unsorted = status[6] != 0;
    const bool generic = all_nodes != 0 || unsorted;
    // list mode (the usual call): most windows defer nothing -- leave before the tables are built
    const int M = generic ? *m_ptr : *node_list_count;
    if (M <= 0) return;
    const int side = 2 * r + 1;
    const int S = side * side;
    const int cap = dense_gen_cap(K);
    unsigned long long *gk = reinterpret_cast<unsigned long long *>(dense_lds);                 // [G][2][cap]
    int *gs = reinterpret_cast<int *>(dense_lds + (size_t)G * 2 * cap * 8);                      // [G][2][cap]
    short *sp_tab = reinterpret_cast<short *>(dense_lds + (size_t)G * 2 * cap * 12);             // spiral index -> offset
    unsigned short *sp_rank = reinterpret_cast<unsigned short *>(sp_tab + (S + 3) / 4 * 4);      // offset -> spiral index
    for (int s = threadIdx.x; s < S; s += kBlock) {
        int sx, sy;
        spiral_offset(s, sx, sy);
        sp_tab[s] = (short)((sx + 64) | ((sy + 64) << 8));
        sp_rank[(sy + r) * side + (sx + r)] = (unsigned short)s;
    }
    __syncthreads();
    const int l = threadIdx.x & 15;
    const int grp = threadIdx.x >> 4;
    const int gshift = threadIdx.x & 48;
    const unsigned lt_mask = (1u << l) - 1u;
    const bool hot = status[4] != 0;         // some pixel holds more than Q events: visibility has to be looked up
    long long edges_acc = 0;
    // every block sweeps a contiguous range of the list (= runs of neighbouring destinations: their probes share cache
    // lines); XCD x = blockIdx % 8 owns the x-th eighth
    const int Gd = gridDim.x, nx = (Gd % 8 == 0) ? 8 : 1;
    const int xcd = blockIdx.x % nx, lb = blockIdx.x / nx, bpx = Gd / nx;
    const int chunk = (M + nx - 1) / nx;
    const int per_block = (chunk + bpx - 1) / bpx;
    const int n_begin = xcd * chunk + lb * per_block;
    const int n_end = min(min(M, (xcd + 1) * chunk), n_begin + per_block);
    if (!generic) {
        for (int ni = n_begin + grp; ni < n_end; ni += G) {
            const int n = node_list[ni];
            const int2 me = slot_it[n];
            const int e = me.x, t = me.y;
            const int c = slot_xyb[n];
            const int x = c & 4095, y = (c >> 12) & 4095, b = (c >> 24) & 127;
            const int64_t row = (int64_t)n * K;
            if (l == 0) {
                nbr_src[row] = n;  // self loop first (ev_graph.cu:44-46)
                nbr_code[row] = (int16_t)(r * side + r);
            }
            int total = 1;
            for (int s0 = 0; s0 < S && total < K; s0 += 16) {
                const int s = s0 + l;
                int code = 0, lo = 0, hi = 0;
                if (s < S) {
                    const int sc = sp_tab[s];
                    const int sx = (sc & 255) - 64, sy = ((sc >> 8) & 255) - 64;
                    code = (sx + r) * side + (sy + r);
                    const int xn = x + sx, yn = y + sy;
                    if (xn >= 0 && yn >= 0 && xn < W && yn < H) {          // out of FOV: skip this pixel only
                        const Pair sN = *reinterpret_cast<const Pair *>(start + xn + W * (yn + H * b));   // one 8-byte load
                        if (sN.b > sN.a) admissible_range(slot_it, slot_xyb, sN.a, sN.b, e, t, delta_t, hot, lo, hi);
                    }
                }
                const int v = min(hi - lo, K);
                int slot = total + row16_inclusive_scan(v) - v;
                total += row16_sum(v);
                for (int k = 0; k < v && slot < K; k++, slot++) {
                    nbr_src[row + slot] = hi - 1 - k;     // newest first
                    nbr_code[row + slot] = (int16_t)code;
                }
            }
            if (total > K) total = K;
            if (l == 0) { deg[n] = total; edges_acc += total; }
        }
    } else
    for (int ni = n_begin + grp; ni < n_end; ni += G) {
        const int n = ni;
        const int2 me = slot_it[n];
        const int e = me.x, t = me.y;
        const int c = slot_xyb[n];
        const int x = c & 4095, y = (c >> 12) & 4095, b = (c >> 24) & 127;
        const int64_t row = (int64_t)n * K;
        if (l == 0) {
            nbr_src[row] = n;  // self loop first (ev_graph.cu:44-46)
            nbr_code[row] = (int16_t)(r * side + r);
        }
        unsigned long long *k0 = gk + (size_t)grp * 2 * cap, *k1 = k0 + cap;
        int *s0p = gs + (size_t)grp * 2 * cap, *s1p = s0p + cap;
        int V = 0;
        // keep the K - 1 smallest keys of the list, in key order (keys are distinct: they carry the event id)
        auto compact = [&]() {
            __builtin_amdgcn_wave_barrier();
            for (int vi = l; vi < V; vi += 16) {
                const unsigned long long mk = k0[vi];
                int rk = 0;
                for (int j = 0; j < V; j++) rk += (k0[j] < mk) ? 1 : 0;
                if (rk < K - 1) { k1[rk] = mk; s1p[rk] = s0p[vi]; }
            }
            __builtin_amdgcn_wave_barrier();
            V = min(V, K - 1);
            unsigned long long *tk_ = k0; k0 = k1; k1 = tk_;
            int *ts_ = s0p; s0p = s1p; s1p = ts_;
        };
        for (int dy = -r; dy <= r; dy++) {
            const int yn = y + dy;
            if (yn < 0 || yn >= H) continue;
            const int base = W * (yn + H * b);
            const int lo = start[base + max(x - r, 0)], hi = start[base + min(x + r, W - 1) + 1];
            for (int c0 = lo; c0 < hi; c0 += 16) {
                const int sv = c0 + l;
                bool valid = false;
                unsigned long long key = 0;
                if (sv < hi) {
                    const int2 it = slot_it[sv];
                    const int cx = slot_xyb[sv];
                    // visible in the FIFO; older than the destination (ev_graph.cu:64); dt <= delta (:69)
                    valid = (cx < 0) && it.x < e && !((float)(t - it.y) > delta_t);
                    const int dx = (cx & 4095) - x;
                    key = ((unsigned long long)sp_rank[(dy + r) * side + (dx + r)] << 32) |
                          (unsigned long long)(0xFFFFFFFFu - (unsigned)it.x);       // newest (largest id) first
                }
                const unsigned bits = (unsigned)(__ballot(valid) >> gshift) & 0xffffu;
                if (valid) {
                    const int pidx = V + __popc(bits & lt_mask);
                    k0[pidx] = key;
                    s0p[pidx] = sv;
                }
                V += __popc(bits);
                if (V > cap - 16) compact();      // (group-uniform)
            }
        }
        compact();
        for (int vi = l; vi < V; vi += 16) {
            int sx, sy;
            spiral_offset((int)(k0[vi] >> 32), sx, sy);
            nbr_src[row + 1 + vi] = s0p[vi];
            nbr_code[row + 1 + vi] = (int16_t)((sx + r) * side + (sy + r));
        }
        const int total = 1 + V;
        __builtin_amdgcn_wave_barrier();  // the lists are reused by the next destination
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
// of the graph build (denormalise + per-key count) on the way: the launch in front of the captured window does what the
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
        count_event<BatchT, false>(i, pos, batch, W, H, B, (float)W, (float)H, fT, cnt, ev_xyb, ev_t, ev_rank,
                                   status8 + 8, status8 + 9);
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
// views into the builder workspace for the level-0 pooling kernel (pooling.hip) and the asynchronous update
// (async_update.hip): the index is keyed (sample, y, x) -- see PixelIndex in common.hpp
void graph_ws_index(const dagr_graph_desc *desc, void *workspace, PixelIndex *out) {
    GraphWs ws;
    carve(*desc, (char *)workspace, &ws);
    out->start = ws.start;
    out->slot_it = ws.slot_it;
    out->slot_xyb = ws.slot_xyb;
    out->n_nodes = ws.start + ws.PK;
    out->unsorted = ws.status + 6;
    out->W = desc->width;
    out->H = desc->height;
}
const int32_t *graph_ws_node_count(const dagr_graph_desc *desc, void *workspace) {
    GraphWs ws;
    carve(*desc, (char *)workspace, &ws);
    return ws.start + ws.PK;
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
    DAGR_CHECK_HIP(hipMemsetAsync(ws.cnt, 0, (ws.PK + 1 + 8) * 4, (hipStream_t)stream));
    DAGR_CHECK_HIP(hipMemsetAsync(ws.start, 0, (ws.PK + 1 + 8) * 4, (hipStream_t)stream));
    DAGR_CHECK_HIP(hipMemsetAsync(ws.status, 0, 16 * 4, (hipStream_t)stream));
    DAGR_CHECK_HIP(hipMemsetAsync(ws.scan_tmp, 0, scan_chained_state_bytes(ws.PK + 1), (hipStream_t)stream));
    return DAGR_OK;
}

static int launch_search(const dagr_graph_desc *desc, const GraphWs &ws, int64_t N, int32_t *nbr_src, int16_t *nbr_code,
                         int32_t *deg, hipStream_t stream, const GatherArgs *gather = nullptr) {
    const GatherArgs ga = gather ? *gather : GatherArgs{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
    const int W = desc->width, H = desc->height, K = desc->max_neighbors, r = desc->radius;
    const bool rows = 2 * r + 2 <= 16;
    if (rows) {
        // fast path: candidate-centric row kernel; dense neighbourhoods are deferred (list in ev_rank, which is dead
        // after k_scatter; counter in status[5]) to the position-centric walk of k_search_dense
        constexpr size_t rows_lds = (size_t)(kBlock / 16) * (kRowCap + 4) * 4;
        // measurement knob DAGR_ROWS_VARIANT = 10 * rounds + waves per SIMD (16 candidates per round)
        static const int variant = (int)knob("DAGR_ROWS_VARIANT", 46);
        // neighbourhoods beyond this many candidates go to the position-centric walk (measurement knob DAGR_DEFER_CAP)
        static const int defer_cap = std::min(kRowCap, std::max(16, (int)knob("DAGR_DEFER_CAP", kRowCap)));
        // candidates from which a neighbourhood is searched in its inner rings first (k_search_rows; 0 = never) and
        // candidates wanted per source.  Measured in round 5 (whole build in us, profiles/r5_ring_sweep.txt): S-edges
        // 8 x 100 k 833 -> 748, 8 x 200 k 1631 -> 1511, S-uniform 8 x 400 k 2761 -> 2449 at (200, 6); wanting 4 per source
        // makes the inner pass fall short three times in four on uniform streams, thresholds below 200 cost sparse windows
        // two loads for nothing
        static const int ring_thr = (int)knob("DAGR_RING_THR", 200) | ((int)knob("DAGR_RING_WANT", 6) << 16);
        auto launch_rows = [&](auto kern) {
            static thread_local unsigned res_rows = 0;
            if (!res_rows) res_rows = persistent_grid(kern, kBlock, rows_lds, 1 << 30);
            const unsigned gR = round_grid8(std::min<int64_t>(ceil_div(N * 16, kBlock), res_rows));
            kern<<<gR, kBlock, rows_lds, stream>>>(ws.start + ws.PK, W, H, K, r, (float)desc->delta_t_us, defer_cap, ring_thr,
                                                   ws.slot_xyb, ws.start, ws.slot_it, nbr_src, nbr_code, deg, ws.status,
                                                   ws.ev_rank, ws.status + 5);
        };
        switch (variant) {
            case 47: launch_rows(k_search_rows<kRowCap, 4, 7>); break;
            case 45: launch_rows(k_search_rows<kRowCap, 4, 5>); break;
            case 36: launch_rows(k_search_rows<kRowCap, 3, 6>); break;
            default: launch_rows(k_search_rows<kRowCap, 4, 6>); break;
        }
        DAGR_CHECK_LAUNCH();
    }
    const size_t dense_lds = dense_lds_bytes(K, r);
    static thread_local unsigned res_dense = 0;
    static thread_local size_t res_dense_lds = 0;
    if (!res_dense || res_dense_lds != dense_lds) {
        res_dense = persistent_grid(k_search_dense, kBlock, dense_lds, 1 << 30);
        res_dense_lds = dense_lds;
    }
    const unsigned gT = round_grid8(std::min<int64_t>(ceil_div(N * 16, kBlock), res_dense));
    k_search_dense<<<gT, kBlock, dense_lds, stream>>>(ws.start + ws.PK, W, H, K, r, (float)desc->delta_t_us, ws.slot_xyb,
                                                     ws.start, ws.slot_it, nbr_src, nbr_code, deg, ws.status, ws.ev_rank,
                                                     ws.status + 5, rows ? 0 : 1, ga);
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
    const unsigned gN = xcd_grid(ceil_div(N, kBlock));
    const int xr = xcd_remap_on();
    const int W = desc->width, H = desc->height, B = desc->batch_size;
    // (device-count form: dagr_stage_window, the launch in front of the captured window, has cleared the status words and
    // run K1 on the window it staged)
    if (!n_dev) {
        DAGR_CHECK_HIP(hipMemsetAsync(ws.status, 0, 8 * 4, stream));
#define DAGR_LAUNCH_COUNT(BT, IP)                                                                          \
    k_count<BT, IP><<<gN, kBlock, 0, stream>>>(pos, (const BT *)batch, n, W, H, B, (float)W, (float)H,        \
                                               (float)desc->time_window, ws.cnt, ws.ev_xyb, ws.ev_t,         \
                                               ws.ev_rank, ws.status, xr)
        if (batch_is_int64) { if (pos_is_int32) DAGR_LAUNCH_COUNT(int64_t, true); else DAGR_LAUNCH_COUNT(int64_t, false); }
        else                { if (pos_is_int32) DAGR_LAUNCH_COUNT(int32_t, true); else DAGR_LAUNCH_COUNT(int32_t, false); }
#undef DAGR_LAUNCH_COUNT
        DAGR_CHECK_LAUNCH();
    }
    // start = exclusive_scan(cnt); cnt is re-zeroed in the same pass (invariant for the next window)
    DAGR_CHECK_HIP(exclusive_scan_i32_chained(ws.cnt, ws.start, ws.PK + 1, ws.scan_tmp, true, stream));
    k_scatter<<<gN, kBlock, 0, stream>>>(n, n_dev, W, H, ws.ev_xyb, ws.ev_rank, ws.start, ws.slot_tmp,
                                         ws.ev_slot, ws.status, xr);
    DAGR_CHECK_LAUNCH();
    // number of occupied CSR slots M = start[PK] <= N (dropped events excluded); slots are a
    // prefix [0, M) so launching N threads with an in-kernel bound read would need M on the host.
    // Out-of-FOV events are an error condition; we order all N slots but guard on start[PK].
    const int hot_cap = (int)(desc->max_events + 1);
    const int hot_thr = std::min(kShortSeg, desc->queue_size);
    k_order<<<gN, kBlock, 0, stream>>>(n, ws.PK, W, H, hot_thr, ws.ev_xyb, ws.ev_t, ws.start, ws.slot_tmp, ws.slot_it,
                                       ws.slot_xyb, ws.ev_slot, ws.hot_list, hot_cap, ws.status, xr);
    DAGR_CHECK_LAUNCH();
    k_fix_pixels<<<1024, kBlock, 0, stream>>>(desc->queue_size, ws.ev_xyb, ws.slot_xyb, ws.ev_slot, ws.ev_t, ws.start,
                                            ws.slot_tmp, ws.slot_it, ws.hot_list, hot_cap, ws.status);
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
    DAGR_CHECK_HIP(hipMemsetAsync(ws.status + 7, 0, 4, stream));
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
                                                                                 ws.start + ws.PK, slot_event,
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
    k_gather_inputs<<<(unsigned)ceil_div(N, kBlock), kBlock, 0, (hipStream_t)stream>>>(ws.start + ws.PK, (int)N, ws.slot_it,
                                                                                     ws.slot_xyb, ga);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

}  // extern "C"
