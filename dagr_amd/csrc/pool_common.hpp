// pool_common.hpp -- the voxel-pooling workspace and the per-node merge shared by pooling.hip (launch (A) of a pooling
// step) and gemm.hip (the same merge fused into the epilogue of the SplineConv that produces the pooled features).
#pragma once
#include "common.hpp"

namespace dagr {
namespace {

constexpr int kRowSlots = 64;           // coarse in-degree bound per cluster (flagged if exceeded)
constexpr int kPoolScanTile = 2048;     // table slots per workgroup of the chained scan (256 threads x 8)
constexpr double kPosScale = 1099511627776.0;  // 2^40
constexpr double kFeatScale = 4294967296.0;    // 2^32

struct PoolWs {
    int32_t *occupied;   // [T+1] flags, zero between calls
    int32_t *newid;      // [T+1] exclusive scan (newid[T] = number of clusters)
    int32_t *scan_tmp;
    long long *possum;   // [2][T][3] fixed point 2^-40 (pair `epoch & 1` is the one being filled)
    int32_t *cnt;        // [2][T]
    int32_t *perm;       // [T] max member index (consecutive_cluster's perm on CPU)
    long long *xacc;     // [T][C]: ordered-int max (low 32 bits) or fixed-point sum
    int32_t *rows;       // [T][64] source-cluster sets, -1 = empty (raw ids on the 3-launch path)
    int32_t *rowcnt;     // [T+1]
    int32_t *status;     // [8]: 0 flags (sticky); 4 = epoch; 5 = level-0 nodes merged through the global path (sticky, cumulative);
                         //      6 = launch tag of the chained scan; 7 = its tile ticket counter (zero between launches)
    unsigned long long *tile_state;   // [ceil((T + 1) / kPoolScanTile) + 8] chained scan: tag | flag | clusters | edges
    unsigned long long *nbmask;  // [T] level 0: 5x5 bitmaps of source cells, zero between calls: bits 0-24 cells of the
                                 // slot's own sample plane, bits 32-56 cells of the plane below (sources of the slot's
                                 // t == 1.0 members, QUIRK-1)
    int T;
};

__host__ __device__ inline size_t pool_carve(const dagr_pool_desc &d, char *base, PoolWs *ws) {
    const int64_t T = (int64_t)d.gx * d.gy * (d.batch_size + 1);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off = (off + bytes + 255) / 256 * 256;
        return base ? base + o : nullptr;
    };
    PoolWs w;
    w.occupied = (int32_t *)take((T + 32) * 4);
    w.newid = (int32_t *)take((T + 32) * 4);
    w.scan_tmp = (int32_t *)take(((T + 1 + kScanTile - 1) / kScanTile + 8) * 4);
    w.possum = (long long *)take(2 * T * 3 * 8);
    w.cnt = (int32_t *)take(2 * T * 4);
    w.perm = (int32_t *)take(T * 4);
    w.xacc = (long long *)take(T * (size_t)d.channels * 8);
    w.rows = (int32_t *)take(T * (size_t)kRowSlots * 4);
    w.rowcnt = (int32_t *)take((T + 32) * 4);
    w.status = (int32_t *)take(32);
    w.tile_state = (unsigned long long *)take(((T + 1 + kPoolScanTile - 1) / kPoolScanTile + 8) * 8);
    w.nbmask = (unsigned long long *)take((T + 9) * 8);
    w.T = (int)T;
    if (ws) *ws = w;
    return off;
}

// the position / count accumulators of the pair being filled (before the scan of this call) ...
__device__ __forceinline__ int ws_pair(const PoolWs &ws) { return ws.status[4] & 1; }
__device__ __forceinline__ long long *ws_possum(const PoolWs &ws, int pair) { return ws.possum + (size_t)pair * ws.T * 3; }
__device__ __forceinline__ int32_t *ws_cnt(const PoolWs &ws, int pair) { return ws.cnt + (size_t)pair * ws.T; }

__device__ __forceinline__ int enc_f(float f) {
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float dec_f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }
constexpr int kEncMin = (int)0x80000000;

// grid_cluster (torch_cluster): trunc((pos - 0) / size) per dimension; time voxel = 1, batch voxel = 1.
__device__ __forceinline__ int cluster_raw(float px, float py, float pt, int b, const dagr_pool_desc &d, bool &ok) {
    const int cx = (int)(px / d.vx);
    const int cy = (int)(py / d.vy);
    const int ct = (int)(pt / 1.0f);
    ok = (cx >= 0 && cx < d.gx && cy >= 0 && cy < d.gy && ct >= 0 && ct <= 1 && b >= 0 && b < d.batch_size);
    return cx + d.gx * (cy + d.gy * (ct + b));
}

// coarse edges: insert source cluster `cs` into the slot set of destination row `cd`; true = this call added it
__device__ __forceinline__ bool row_insert(int32_t *__restrict__ rows, int cd, int cs, int32_t *status) {
    int32_t *row = rows + (size_t)cd * kRowSlots;
    unsigned h = ((unsigned)cs * 2654435761u) >> 26;  // 6 bits
    for (int probe = 0; probe < kRowSlots; probe++) {
        // L2-coherent read (sc1): neighbouring nodes insert the same few sources over and over; a stale
        // L1 line would send every one of them to the atomic
        const int cur = __hip_atomic_load(&row[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == cs) return false;
        if (cur == -1) {
            const int old = atomicCAS(&row[h], -1, cs);
            if (old == -1) return true;
            if (old == cs) return false;
        }
        h = (h + 1) & (kRowSlots - 1);
    }
    atomicOr(status, 2);  // more than 64 distinct sources
    return false;
}


// One node's share of launch (A) that does not depend on its features: count, largest member index, fixed-point position
// sums, the occupancy flag, and the node's in-edges as (source cluster -> this cluster) entries of the destination's slot
// set.  `lane` / `n_lanes`: the lanes of the calling group walk the in-edges together; lane 0 does the bookkeeping.
__device__ __forceinline__ void pool_merge_node(const dagr_pool_desc &d, const PoolWs &ws, int n, int raw,
                                                const float *__restrict__ pos, const int32_t *__restrict__ batch,
                                                const int32_t *__restrict__ col, int e0, int e1, int lane, int n_lanes,
                                                int32_t *__restrict__ cluster_raw_out) {
    if (lane == 0) {
        const int pair = ws_pair(ws);
        cluster_raw_out[n] = raw;
        ws.occupied[raw] = 1;
        atomicAdd(&ws_cnt(ws, pair)[raw], 1);
        atomicMax(&ws.perm[raw], n);
#pragma unroll
        for (int k = 0; k < 3; k++)
            atomicAdd(reinterpret_cast<unsigned long long *>(ws_possum(ws, pair) + (size_t)raw * 3 + k),
                      (unsigned long long)(long long)llrint((double)pos[3 * n + k] * kPosScale));
    }
    for (int e = e0 + lane; e < e1; e += n_lanes) {
        const int src = col[e];
        bool oks;
        const int rs = cluster_raw(pos[3 * src], pos[3 * src + 1], pos[3 * src + 2], batch[src], d, oks);
        if (!oks || rs == raw) continue;
        if (row_insert(ws.rows, raw, rs, ws.status)) atomicAdd(&ws.rowcnt[raw], 1);
    }
}

}  // namespace
}  // namespace dagr
