// common.hpp -- shared host/device helpers for libdagr_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>

#include "dagr_hip.h"

namespace dagr {

void set_error(const std::string &msg);

#define DAGR_CHECK_ARG(cond, msg)                                             \
    do {                                                                      \
        if (!(cond)) {                                                        \
            ::dagr::set_error(std::string(__func__) + ": " + (msg));          \
            return DAGR_ERR_INVALID_ARG;                                      \
        }                                                                     \
    } while (0)

#define DAGR_CHECK_HIP(expr)                                                  \
    do {                                                                      \
        hipError_t _e = (expr);                                               \
        if (_e != hipSuccess) {                                               \
            ::dagr::set_error(std::string(__func__) + ": " #expr " -> " +     \
                              hipGetErrorString(_e));                         \
            return DAGR_ERR_HIP;                                              \
        }                                                                     \
    } while (0)

#define DAGR_CHECK_LAUNCH() DAGR_CHECK_HIP(hipGetLastError())

constexpr int kBlock = 256;  // 4 waves of 64
constexpr int kWave = 64;

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Grid of a persistent (grid-stride) kernel: exactly the number of blocks that are co-resident, so
// that every block gets the same share of the work (an over-sized grid runs in uneven waves).
inline int device_cu_count() {
    static int cus = [] {
        int dev = 0, n = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        return n;
    }();
    return cus;
}
template <typename Kern>
inline unsigned persistent_grid(Kern kernel, int block, size_t dyn_lds, int64_t max_useful_blocks) {
    // the occupancy query is cached per (kernel, dynamic LDS size): this runs on every launch.  Kernels of
    // one signature share this instantiation, so the key includes the function address.
    struct Entry { const void *fn; size_t lds; int per_cu; };
    static thread_local Entry cache[32];
    static thread_local int n_cached = 0;
    const void *fn = reinterpret_cast<const void *>(kernel);
    int cached_per_cu = 0;
    for (int i = 0; i < n_cached; i++)
        if (cache[i].fn == fn && cache[i].lds == dyn_lds) cached_per_cu = cache[i].per_cu;
    if (cached_per_cu < 1) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, dyn_lds) != hipSuccess || per_cu < 1)
            per_cu = 1;
        cached_per_cu = per_cu;
        if (n_cached < 32) cache[n_cached++] = Entry{fn, dyn_lds, per_cu};
    }
    int64_t g = (int64_t)cached_per_cu * device_cu_count();
    if (g > max_useful_blocks) g = max_useful_blocks;
    return (unsigned)(g < 1 ? 1 : g);
}
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- degree-1 open B-spline basis at an integer LUT coordinate (spline_conv.py:27-34 builds the
// pseudo-coordinate; torch_spline_conv basis: v = pseudo*(5-1), frac = v-floor(v),
// factor(k_mod) = 1 - frac - k_mod + 2*frac*k_mod, tap = (floor(v)+k_mod) % 5, x fastest).
struct Axis {
    int k0, k1;     // kernel taps
    float b0, b1;   // their weights
};
__host__ __device__ inline Axis spline_axis(int idx, int r, float den) {
    const float pseudo = (float)(idx - r) / den + 0.5f;
    const float v = pseudo * 4.0f;
    const float fl = floorf(v);
    const float frac = v - fl;
    Axis a;
    const int f = (int)fl;
    a.k0 = f % 5;
    a.k1 = (f + 1) % 5;
    a.b0 = ((1.0f - frac) - 0.0f) + (2.0f * frac) * 0.0f;
    a.b1 = ((1.0f - frac) - 1.0f) + (2.0f * frac) * 1.0f;
    return a;
}

// ---- device-side wave / block primitives (wave = 64 lanes) -------------------------------
__device__ __forceinline__ int wave_inclusive_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int n = __shfl_up(v, d, 64);
        if (lane >= d) v += n;
    }
    return v;
}

// Exclusive scan over a 256-thread block. `smem` must hold >= 4 ints. Returns the exclusive
// prefix of `v` for this thread and the block total in `total`.
__device__ __forceinline__ int block_exclusive_scan(int v, int *smem, int &total) {
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    int incl = wave_inclusive_scan(v);
    if (lane == 63) smem[wid] = incl;
    __syncthreads();
    int w0 = smem[0], w1 = smem[1], w2 = smem[2], w3 = smem[3];
    __syncthreads();
    int base = (wid > 0 ? w0 : 0) + (wid > 1 ? w1 : 0) + (wid > 2 ? w2 : 0);
    total = w0 + w1 + w2 + w3;
    return base + incl - v;
}

// ---- XCD-aware work split.  Workgroup b is (observed, not contractual) placed on XCD b % 8 and each
// XCD has a private 4 MiB L2.  Event ids are time-ordered per sample and samples are concatenated, so
// giving every XCD one contiguous eighth of the items keeps an item's neighbours (earlier events of the
// same sample / the same pixel plane) in that XCD's L2 instead of pulling the whole array into all
// eight.  Pure performance mapping: any placement yields the same result.
struct XcdSplit {
    int first, end, stride;
};
__device__ __forceinline__ XcdSplit xcd_split(int n_items, int items_per_block, int item_in_block) {
    const int G = gridDim.x;
    const int nx = (G % 8 == 0) ? 8 : 1;
    const int xcd = blockIdx.x % nx, lb = blockIdx.x / nx, bpx = G / nx;
    const int chunk = (n_items + nx - 1) / nx;
    const int begin = xcd * chunk;
    XcdSplit s;
    s.end = min(n_items, begin + chunk);
    s.first = begin + lb * items_per_block + item_in_block;
    s.stride = bpx * items_per_block;
    return s;
}
// One-item-per-thread launches over E items in natural order (events: sample after sample; CSR slots: sample, row, pixel):
// workgroup b of the grid sits on XCD b % 8 (see xcd_split), so the natural block index deals every sample's items to all
// eight L2s.  Remapped, XCD x takes the x-th contiguous eighth of the blocks -- with B = 8 samples one sample's per-pixel
// counters, offsets and slots (1.2 + 1.2 + 1.2 MB at 640x480) stay in ONE 4-MiB L2.  Returns the logical block, or -1 for
// the padding blocks of a grid rounded up to a multiple of 8.  xcd_remap = 0: identity (measurement knob).
__device__ __forceinline__ int xcd_block(int n_blocks, int xcd_remap) {
    if (!xcd_remap) return (int)blockIdx.x < n_blocks ? (int)blockIdx.x : -1;
    const int bpx = (n_blocks + 7) >> 3;
    const int lb = (int)(blockIdx.x >> 3) + (int)(blockIdx.x & 7) * bpx;
    return ((int)(blockIdx.x >> 3) < bpx && lb < n_blocks) ? lb : -1;
}
// Measurement knobs (A/B switches read from the environment) exist only in the measurement build of the library
// (`make measure`: -DDAGR_MEASURE -> dagr_amd/lib/libdagr_hip_measure.so, loaded with DAGR_LIB=measure by the probes
// under tools/).  The product library reads no environment variable: every knob is its default.
inline long long knob(const char *name, long long dflt) {
#ifdef DAGR_MEASURE
    const char *e = getenv(name);
    return e ? atoll(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}
inline int xcd_remap_on() {      // DAGR_XCD_REMAP=0: natural block order (A/B measurements)
    static const int v = (int)knob("DAGR_XCD_REMAP", 1);
    return v;
}
inline unsigned xcd_grid(int64_t n_blocks) { return (unsigned)((n_blocks + 7) / 8 * 8); }   // whole rounds of the eight XCDs
inline unsigned round_grid8(int64_t g) { return (unsigned)(g <= 8 ? (g < 1 ? 1 : g) : (g + 7) / 8 * 8); }

// ---- generic int32 exclusive scan over n elements (3 launches) --------------------------
constexpr int kScanTile = 2048;  // elements per block: 256 threads x 8
size_t scan_scratch_elems(int64_t n);
// out[i] = sum_{j<i} in[j]; if zero_input, in[] is cleared after being consumed. in may alias out
// only if !zero_input.  scratch: scan_scratch_elems(n) ints.
hipError_t exclusive_scan_i32(int32_t *in, int32_t *out, int64_t n, int32_t *scratch, bool zero_input,
                              hipStream_t stream);
// the same in one launch (decoupled look-back); state: scan_chained_state_bytes(n) bytes, zero before the first launch
size_t scan_chained_state_bytes(int64_t n);
hipError_t exclusive_scan_i32_chained(int32_t *in, int32_t *out, int64_t n, void *state, bool zero_input,
                                      hipStream_t stream);


hipError_t launch_gemm_mfma(const int32_t *m_ptr, int m_max, const float *A, int lda, const float *Wm, int ldw,
                            const float *bias, float *C, int ldc, int K, int N, int relu, hipStream_t stream);
const int32_t *graph_ws_node_count(const dagr_graph_desc *desc, void *workspace);
// The window builder's event index as the other kernels see it (graph_build.hip).  Events are CSR slots in
// (sample, y, x) order, ids ascending inside a segment (one pixel); key of a segment: x + W * (y + H * sample);
// start[key] .. start[key + 1] are its slots.  All events of a range of pixel rows of one sample are ONE contiguous run of
// slots (rows(y0) .. rows(y1)); a pixel's events, oldest (smallest id) first, are its segment.
struct PixelIndex {
    const int32_t *start;
    const int2 *slot_it;        // {event id, t} per slot
    const int32_t *slot_xyb;    // x | y << 12 | sample << 24 | visible << 31 per slot
    const int32_t *n_nodes;     // number of indexed events (device)
    const int32_t *unsorted;    // != 0: timestamps were not non-decreasing in event order inside a sample (device)
    int W, H;
    __host__ __device__ int segment(int x, int yb) const { return x + W * yb; }   // yb = y + H * sample
    __host__ __device__ int row_begin(int yb) const { return W * yb; }            // key of the first segment of pixel row yb
};
void graph_ws_index(const dagr_graph_desc *desc, void *workspace, PixelIndex *out);

}  // namespace dagr
