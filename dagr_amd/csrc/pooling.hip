// pooling.hip -- voxel-grid pooling of the event graph (fp32 features, exact integer decisions).
//
// Reference: Pooling.forward (src/dagr/model/layers/pooling.py:51-97) =
//   torch_cluster.grid_cluster -> torch.unique relabel (consecutive_cluster :12-16, host sync) ->
//   cluster[edge_index], drop self loops, unique(dim=-1) (host sync, lexicographic sort) ->
//   scatter-mean of pos (pool_pos), scatter-max / mean of x, round_to_pixel (:47-49), T.Cartesian.
//
// Here (no host sync, nothing sorted):
//   * cluster id = cx + gx*(cy + gy*(ct + b)) from the same fp32 divisions (IEEE, this TU is built with
//     -ffp-contract=off; SURVEY QUIRK-2), including the t == 1.0 leak into the next sample's id range
//     (QUIRK-1).  Ids index a dense table of T = gx*gy*(B+1) slots; an exclusive scan over the
//     occupancy flags gives the consecutive ids torch.unique(sorted=True) would give.
//   * reductions are order-independent and therefore deterministic: max through an order-preserving
//     int mapping, means through exact 64-bit fixed-point sums (2^-40 for positions, 2^-32 for
//     features) divided once at the end.
//   * level 0 (N events): nodes are the graph builder's CSR slots, so the voxels of one sample / voxel row own one
//     contiguous run of them: equal runs of slots are streamed into LDS windows of table slots (k_pool_l0_slots below).
//     Coarser levels (<= 2240*(B+1) nodes): atomics into the same accumulators.
//   * coarse edges: per destination cluster a 64-slot open-addressing set of source clusters
//     (atomicCAS) -- or, from level 0, two 5x5 bitmaps of source cells -- then one wave per cluster sorts
//     its row; rows -> CSR; each edge gets the integer LUT coordinate the next SplineConv needs, computed
//     with the reference's float formula (T.Cartesian then spline_conv.py:41-42).
//   * three launches per pooling step: (A) gather -- members into the accumulators AND the coarse edges, keyed by
//     the raw (table) ids so that nothing waits for the relabelling; (S) one workgroup scans the occupancy flags and
//     the row sizes together (new ids + row pointers); (C) one wave per table slot writes the pooled node, sorts its
//     row and emits the CSR edges with their LUT coordinates.  (C) reads the positions of the source clusters
//     from their accumulators while other waves finalize theirs, so the position/count accumulators exist twice
//     and alternate between calls (device-side epoch): (C) of call n re-arms the pair call n-1 used.  The chain was
//     8 (pooled levels) / 12 (level 0) dependent launches of ~4.5 us each before.
#include <climits>

#include "common.hpp"
#include "pool_common.hpp"

namespace dagr {
namespace {

constexpr int kMaxL0Channels = 512;     // feature channels of a level-0 pooling (LDS window: at least one voxel row)

// torch.div(a, b, rounding_mode='floor') for floats (c10::div_floor_floating)
__device__ __forceinline__ float div_floor(float a, float b) {
    if (b == 0.0f) return a / b;
    const float mod = fmodf(a, b);
    float div = (a - mod) / b;
    if ((mod != 0.0f) && ((b < 0.0f) != (mod < 0.0f))) div -= 1.0f;
    float floordiv;
    if (div != 0.0f) {
        floordiv = floorf(div);
        if (div - floordiv > 0.5f) floordiv += 1.0f;
    } else {
        floordiv = copysignf(0.0f, a / b);
    }
    return floordiv;
}

// ---------------------------------------------------------------------------------------------
// pooled levels, launch (A): one thread per (node, channel) merges the node into its cluster's accumulators; the
// same threads then walk the node's in-edges (thread `ch` takes edges ch, ch + C, ...) and insert
// (source cluster -> this cluster) into the destination's slot set -- by raw ids, the source's recomputed from its
// position, so the edges wait neither for the relabelling nor for a cluster-id pass over the nodes.
__global__ __launch_bounds__(kBlock) void k_pool_accumulate(dagr_pool_desc d, const int32_t *__restrict__ n_ptr,
                                                           int n_max, const float *__restrict__ x, int ldx,
                                                           const float *__restrict__ pos,
                                                           const int32_t *__restrict__ batch,
                                                           const int32_t *__restrict__ rowptr,
                                                           const int32_t *__restrict__ col, PoolWs ws,
                                                           int32_t *__restrict__ cluster_raw_out) {
    const int n_nodes = n_ptr ? min(*n_ptr, n_max) : n_max;
    const int C = d.channels;
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int n = (int)(gid / C), ch = (int)(gid % C);
    if (n >= n_nodes) return;
    bool ok;
    const int raw = cluster_raw(pos[3 * n], pos[3 * n + 1], pos[3 * n + 2], batch[n], d, ok);
    if (!ok) {
        if (ch == 0) { atomicOr(&ws.status[0], 1); cluster_raw_out[n] = -1; }
        return;
    }
    const float v = x[(size_t)n * ldx + ch];
    if (d.aggr == 0)
        atomicMax(reinterpret_cast<int *>(ws.xacc + (size_t)raw * C + ch), enc_f(v));
    else
        atomicAdd(reinterpret_cast<unsigned long long *>(ws.xacc + (size_t)raw * C + ch),
                  (unsigned long long)(long long)llrint((double)v * kFeatScale));
    // the node's own bookkeeping (thread ch == 0) and its in-edges (all C threads of the node)
    pool_merge_node(d, ws, n, raw, pos, batch, col, rowptr ? rowptr[n] : 0, rowptr ? rowptr[n + 1] : 0, ch, C, cluster_raw_out);
}

// ---------------------------------------------------------------------------------------------
// level 0, streaming form.  Nodes are CSR slots in (sample, y, x, time) order, so all voxels of one sample / voxel row
// (a "band") own ONE contiguous run of slots, and their table ids are contiguous too (raw = cx + gx * (cy + gy * b)).
// The kernel therefore never looks a member up: the slots are cut into EQUAL contiguous runs, one per workgroup (balanced
// whatever the event density: S-edges puts thousands of members into a few voxels), and every workgroup streams its run
// -- feature rows, positions, ids, degrees and neighbour codes as fully coalesced reads -- into an LDS window of VW table
// slots that starts at the band of its first slot:
//   phase B  one lane per node (64 nodes per wave step): voxel from two per-pixel LDS tables, count / largest id /
//            64-bit fixed-point position sums as LDS atomics, the 5x5 source-cell bitmap from the node's neighbour codes;
//   phase A  the step's 64 feature rows as one flat run of 16-byte pieces across the wave (lane f: node f / PPN, piece
//            f % PPN), each piece merged by LDS atomics (ordered-int max / 64-bit fixed-point add).  The first pieces are
//            requested BEFORE phase B's arithmetic and every further batch before the previous one is merged: a wave
//            keeps ~8 KB in flight.
// Workgroups are large (16 waves, one or two per CU): a workgroup merges every voxel it touched into the global
// accumulators with one atomic per word, and a band cut into fewer runs means fewer of those.  A run longer than the
// window's bands is walked in segments (the window is merged, re-armed and moved: bands ascend along the slots);
// t == 1.0 nodes (QUIRK-1: their cluster is the same cell one sample plane up) go straight to the global
// accumulators; status[5] counts them (sticky).  Every reduction is
// order-free (integer max / integer sums), so the result does not depend on how slots are cut into runs, steps or waves.
// Before (round 3): one wave per voxel walking its members through a row table in LDS -- a chain of dependent loads per
// member, 0.18 - 0.23 of HBM peak, and a second launch for the tails of event-dense voxels.
constexpr int kPoolL0Block = 1024;

template <int AGGR>
__host__ __device__ inline size_t pool_l0_lds_bytes(int VW, int C, int W, int H) {
    size_t b = ((size_t)VW * C * (AGGR == 0 ? 4 : 8) + 7) / 8 * 8;   // feature accumulators
    b += (size_t)VW * (24 + 4 + 4 + 4);                                 // position sums, count, largest id, bitmap
    b += (kPoolL0Block / 64) * 64 * 2;                                  // per wave: window slot of the step's nodes
    b += ((size_t)(W + H) * 2 + 7) / 8 * 8;                             // pixel -> voxel column / row
    return b + 64;
}

template <int AGGR, int VEC>   // 0 = max, 1 = mean; VEC = floats per piece (4: rows are 16-byte aligned, 1: any layout)
__global__ __launch_bounds__(kPoolL0Block) void k_pool_l0_slots(dagr_pool_desc d, int W, int H, int n_cap, int VW,
                                                               const int32_t *__restrict__ n_ptr,
                                                               const int32_t *__restrict__ xlo,  // [gx+1] pixel bounds
                                                               const int32_t *__restrict__ ylo,  // [gy+1]
                                                               const int32_t *__restrict__ start, int row_keys,
                                                               const int2 *__restrict__ slot_it,
                                                               const int32_t *__restrict__ slot_xyb,
                                                               const float *__restrict__ x, int ldx,
                                                               const float *__restrict__ pos, PoolWs ws,
                                                               // coarse-edge fast path (NULL = off): offset codes
                                                               const int16_t *__restrict__ nbr_code,
                                                               const int32_t *__restrict__ nbr_src,
                                                               const int32_t *__restrict__ deg, int K, int r) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    constexpr int NW = kPoolL0Block / 64;
    const int n = min(*n_ptr, n_cap);
    // every workgroup takes the same share of the nodes that are there (n comes from the device: the launch is sized
    // for the capacity, so that a captured graph serves windows of any size)
    const int per_block = max(64 * NW, (int)(((long long)n + gridDim.x - 1) / gridDim.x + 63) / 64 * 64);
    const int s_begin = blockIdx.x * per_block;
    if (s_begin >= n) return;
    const int s_end = min(n, s_begin + per_block);
    const int C = d.channels;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int *l_acc; unsigned long long *l_ps; int *l_cnt, *l_perm; unsigned *l_nbm; short *l_sv; unsigned short *l_xlut, *l_ylut;
    {
        unsigned char *p = lds_raw;
        l_acc = reinterpret_cast<int *>(p); p += ((size_t)VW * C * (AGGR == 0 ? 4 : 8) + 7) / 8 * 8;
        l_ps = reinterpret_cast<unsigned long long *>(p); p += (size_t)VW * 24;
        l_cnt = reinterpret_cast<int *>(p); p += (size_t)VW * 4;
        l_perm = reinterpret_cast<int *>(p); p += (size_t)VW * 4;
        l_nbm = reinterpret_cast<unsigned *>(p); p += (size_t)VW * 4;
        l_sv = reinterpret_cast<short *>(p); p += NW * 64 * 2;
        l_xlut = reinterpret_cast<unsigned short *>(p); p += (size_t)W * 2;
        l_ylut = reinterpret_cast<unsigned short *>(p);
    }
    auto arm_window = [&]() {
        for (int i = threadIdx.x; i < VW * C; i += kPoolL0Block) {
            if (AGGR == 0) l_acc[i] = kEncMin;
            else reinterpret_cast<unsigned long long *>(l_acc)[i] = 0ull;
        }
        for (int i = threadIdx.x; i < VW; i += kPoolL0Block) {
            l_cnt[i] = 0; l_perm[i] = -1; l_nbm[i] = 0u;
            l_ps[3 * i] = 0ull; l_ps[3 * i + 1] = 0ull; l_ps[3 * i + 2] = 0ull;
        }
    };
    arm_window();
    // pixel -> voxel column / row (cell c owns the pixels [lo[c], lo[c+1]): the host's fp32 division, tabulated)
    for (int c = threadIdx.x; c < d.gx; c += kPoolL0Block)
        for (int px = xlo[c]; px < min(xlo[c + 1], W); px++) l_xlut[px] = (unsigned short)c;
    for (int c = threadIdx.x; c < d.gy; c += kPoolL0Block)
        for (int py = ylo[c]; py < min(ylo[c + 1], H); py++) l_ylut[py] = (unsigned short)c;
    __syncthreads();
    const int cells = d.gx * d.gy;
    const int pair = ws_pair(ws);
    long long *w_possum = ws_possum(ws, pair);
    int32_t *w_cnt = ws_cnt(ws, pair);
    const int side = 2 * r + 1;
    const int side_magic = (65536 + side - 1) / side;   // code / side == (code * magic) >> 16 for code * side < 65536
    const int PPN = C / VEC;                            // pieces per node
    // f / PPN == umulhi(f, ceil(2^32 / PPN)) for f * PPN < 2^32
    const unsigned ppn_magic = PPN > 1 ? (unsigned)((0x100000000ull + (unsigned)PPN - 1) / (unsigned)PPN) : 0u;
    constexpr int kBig = 1 << 28;
    int n_slow = 0;
    const int T = ws.T;
    const int WB = VW / d.gx;          // whole bands the window holds (>= 1)
    // The run is walked in segments: the window covers the band of the segment's first slot and the WB - 1 bands after it;
    // bands ascend along the slots, so the segment ends where band q0 + WB starts -- one entry of the builder's per-pixel
    // offsets.  A run of a full-size window is one segment (sometimes two); sparse windows move the window often, on few
    // nodes.
    for (int seg_begin = s_begin; seg_begin < s_end;) {
    int raw0, seg_end = s_end;
    {
        const int c0 = slot_xyb[seg_begin];
        const int q0 = (int)l_ylut[(c0 >> 12) & 4095] + d.gy * ((c0 >> 24) & 127);
        raw0 = d.gx * q0;
        const int qe = q0 + WB;
        if (qe < d.gy * d.batch_size) {
            const int be = qe / d.gy, cye = qe - be * d.gy;
            seg_end = min(s_end, max(seg_begin + 1, start[row_keys * (min(ylo[cye], H) + H * be)]));
        }
    }
    for (int cs = seg_begin + 64 * wv; cs < seg_end; cs += 64 * NW) {
        // ---- phase B loads: one lane per node
        const int s = cs + lane;
        const bool live = s < seg_end;
        int c = 0, id = 0, dg = 0;
        float px = 0.f, py = 0.f, pt = 0.f;
        int4 lo4 = make_int4(0, 0, 0, 0), hi4 = lo4;
        if (live) {
            c = slot_xyb[s];
            px = pos[3 * (size_t)s]; py = pos[3 * (size_t)s + 1]; pt = pos[3 * (size_t)s + 2];
            id = slot_it[s].x;       // event id: consecutive_cluster's `perm`
            if (nbr_code) {
                dg = min(deg[s], K);
                if (K == 16) {
                    lo4 = *reinterpret_cast<const int4 *>(nbr_code + (size_t)s * 16);
                    hi4 = *reinterpret_cast<const int4 *>(nbr_code + (size_t)s * 16 + 8);
                }
            }
        }
        // ---- phase A, first batch of pieces: requested before phase B's arithmetic (rows cs .. cs + 63 are contiguous)
        const float *xrow = x + (size_t)cs * ldx;
        const int n_live = min(64, seg_end - cs);
        constexpr int U = 4;
        float val[U][VEC];
        int who[U];       // node << 16 | piece, or -1
        auto request = [&](int f0) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int f = f0 + 64 * u + lane;
                who[u] = -1;
                if (f < n_live * PPN) {
                    const int k = PPN > 1 ? (int)__umulhi((unsigned)f, ppn_magic) : f, p = f - k * PPN;
                    who[u] = (k << 16) | p;
                    if constexpr (VEC == 4) {
                        const float4 qv = *reinterpret_cast<const float4 *>(xrow + (size_t)k * ldx + 4 * p);
                        val[u][0] = qv.x; val[u][1] = qv.y; val[u][2] = qv.z; val[u][3] = qv.w;
                    } else {
                        val[u][0] = xrow[(size_t)k * ldx + p];
                    }
                }
            }
        };
        request(0);
        // ---- phase B arithmetic
        int vloc = -1;
        if (live) {
            const int xp = c & 4095, yp = (c >> 12) & 4095, b = (c >> 24) & 127;
            const int cx = l_xlut[xp], cy = l_ylut[yp];
            const bool leak = pt >= 1.0f;
            const int raw = cx + d.gx * (cy + d.gy * b);
            const int rel = raw - raw0;
            const bool inwin = !leak && rel >= 0 && rel < VW;
            const unsigned long long q0 = (unsigned long long)(long long)llrint((double)px * kPosScale);
            const unsigned long long q1 = (unsigned long long)(long long)llrint((double)py * kPosScale);
            const unsigned long long q2 = (unsigned long long)(long long)llrint((double)pt * kPosScale);
            // source cells of the node's in-edges: a source lies within r pixels of its destination, r <= 2 cells (checked
            // by the caller), so they form a 5x5 bitmap around the node's voxel.  Lower pixel bounds of the cells
            // cx-1 .. cx+2 (and rows): the source's cell = cx-2 + #(bounds <= its pixel).
            unsigned nbm = 0, nbm_up = 0, nbm_low = 0;
            if (nbr_code) {
                int tx[4], ty[4];      // thresholds on the offset code's (ox, oy): xs >= bound  <=>  ox >= bound - xp + r
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int qx = cx - 1 + k, qy = cy - 1 + k;
                    tx[k] = (qx <= 0 ? -kBig : (qx >= d.gx ? kBig : xlo[qx])) - xp + r;
                    ty[k] = (qy <= 0 ? -kBig : (qy >= d.gy ? kBig : ylo[qy])) - yp + r;
                }
                int codes[16];
                if (K == 16) {
                    const int w8[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
                    for (int j = 0; j < 8; j++) { codes[2 * j] = (short)(w8[j] & 0xffff); codes[2 * j + 1] = w8[j] >> 16; }
                } else {
#pragma unroll
                    for (int j = 0; j < 16; j++) codes[j] = j < dg ? (int)nbr_code[(size_t)s * K + j] : 0;
                }
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    if (j < dg) {
                        const int code = codes[j];
                        const int ox = (code * side_magic) >> 16, oy = code - ox * side;
                        const int dcx = (ox >= tx[0]) + (ox >= tx[1]) + (ox >= tx[2]) + (ox >= tx[3]);   // 0..4, 2 = own cell
                        const int dcy = (oy >= ty[0]) + (oy >= ty[1]) + (oy >= ty[2]) + (oy >= ty[3]);
                        const unsigned bit = 1u << (dcy * 5 + dcx);
                        if (!leak) {
                            nbm |= bit;
                        } else {
                            // QUIRK-1: a t == 1.0 node belongs to cluster raw + gx*gy (the same cell one sample plane up),
                            // so its in-edges are that slot's: from this plane's cells, or -- sources that are t == 1.0
                            // nodes themselves -- from the slot's own plane
                            const int src = nbr_src[(size_t)s * K + j];
                            if (pos[3 * (size_t)src + 2] >= 1.0f) nbm_up |= bit;
                            else nbm_low |= bit;
                        }
                    }
                }
                nbm &= ~(1u << 12);      // own cell = self loops
                nbm_up &= ~(1u << 12);
            }
            if (inwin) {
                vloc = rel;
                atomicAdd(&l_cnt[rel], 1);
                atomicMax(&l_perm[rel], id);
                atomicAdd(&l_ps[3 * rel], q0);
                atomicAdd(&l_ps[3 * rel + 1], q1);
                atomicAdd(&l_ps[3 * rel + 2], q2);
                if (nbm) atomicOr(&l_nbm[rel], nbm);
            } else {
                // outside the window, or a t == 1.0 node: this lane merges its node into the global accumulators
                const int rl = leak ? raw + cells : raw;
                n_slow++;
                for (int ch = 0; ch < C; ch++) {
                    const float v = x[(size_t)s * ldx + ch];
                    if (AGGR == 0)
                        atomicMax(reinterpret_cast<int *>(ws.xacc + (size_t)rl * C + ch), enc_f(v));
                    else
                        atomicAdd(reinterpret_cast<unsigned long long *>(ws.xacc + (size_t)rl * C + ch),
                                  (unsigned long long)(long long)llrint((double)v * kFeatScale));
                }
                ws.occupied[rl] = 1;
                atomicAdd(&w_cnt[rl], 1);
                atomicMax(&ws.perm[rl], id);
                atomicAdd(reinterpret_cast<unsigned long long *>(w_possum + (size_t)rl * 3 + 0), q0);
                atomicAdd(reinterpret_cast<unsigned long long *>(w_possum + (size_t)rl * 3 + 1), q1);
                atomicAdd(reinterpret_cast<unsigned long long *>(w_possum + (size_t)rl * 3 + 2), q2);
                if (!leak) {
                    if (nbm) atomicOr(&ws.nbmask[rl], (unsigned long long)nbm);
                } else if (nbm_up | nbm_low) {
                    atomicOr(&ws.nbmask[rl], ((unsigned long long)nbm_low << 32) | nbm_up);
                }
            }
        }
        l_sv[wv * 64 + lane] = (short)vloc;
        __builtin_amdgcn_wave_barrier();
        // ---- phase A: merge a batch while the next one is in flight
        for (int f0 = 0; f0 < n_live * PPN; f0 += 64 * U) {
            float cur[U][VEC];
            int tgt[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                tgt[u] = -1;
                if (who[u] >= 0) {
                    const int v = l_sv[wv * 64 + (who[u] >> 16)];
                    if (v >= 0) tgt[u] = v * C + (who[u] & 0xffff) * VEC;
                }
#pragma unroll
                for (int j = 0; j < VEC; j++) cur[u][j] = val[u][j];
            }
            if (f0 + 64 * U < n_live * PPN) request(f0 + 64 * U);
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (tgt[u] >= 0) {
#pragma unroll
                    for (int j = 0; j < VEC; j++) {
                        if (AGGR == 0)
                            atomicMax(&l_acc[tgt[u] + j], enc_f(cur[u][j]));
                        else
                            atomicAdd(reinterpret_cast<unsigned long long *>(l_acc) + tgt[u] + j,
                                      (unsigned long long)(long long)llrint((double)cur[u][j] * kFeatScale));
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();   // the step's window slots are re-used by the next step
    }
    __syncthreads();
    // ---- merge the touched voxels into the global accumulators (t == 1.0 nodes of the sample below and other workgroups
    //      hit the same slots concurrently: atomics, exactly once per word)
    for (int i = threadIdx.x; i < VW * C; i += kPoolL0Block) {
        const int v = i / C;
        if (l_cnt[v] > 0 && raw0 + v < T) {
            const size_t o = (size_t)(raw0 + v) * C + (i - v * C);
            if (AGGR == 0) atomicMax(reinterpret_cast<int *>(ws.xacc + o), l_acc[i]);
            else atomicAdd(reinterpret_cast<unsigned long long *>(ws.xacc + o),
                           reinterpret_cast<unsigned long long *>(l_acc)[i]);
        }
    }
    for (int v = threadIdx.x; v < VW; v += kPoolL0Block) {
        const int cn = l_cnt[v];
        if (cn > 0 && raw0 + v < T) {
            const int raw = raw0 + v;
            ws.occupied[raw] = 1;
            atomicAdd(&w_cnt[raw], cn);
            atomicMax(&ws.perm[raw], l_perm[v]);
            atomicAdd(reinterpret_cast<unsigned long long *>(w_possum + (size_t)raw * 3 + 0), l_ps[3 * v]);
            atomicAdd(reinterpret_cast<unsigned long long *>(w_possum + (size_t)raw * 3 + 1), l_ps[3 * v + 1]);
            atomicAdd(reinterpret_cast<unsigned long long *>(w_possum + (size_t)raw * 3 + 2), l_ps[3 * v + 2]);
            if (l_nbm[v]) atomicOr(&ws.nbmask[raw], (unsigned long long)l_nbm[v]);
        }
    }
    seg_begin = seg_end;
    if (seg_begin < s_end) {            // the window moves on: start it over
        __syncthreads();
        arm_window();
        __syncthreads();
    }
    }   // segments
    // sticky counter of the nodes that took the global path (tests assert that it ran where it should)
    {
        int tot = n_slow;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) tot += __shfl_xor(tot, off, 64);
        if (lane == 0 && tot) atomicAdd(&ws.status[5], tot);
    }
}

// level 0: cluster of every event (for the coarse edges)
__global__ __launch_bounds__(kBlock) void k_pool_l0_event_cluster(dagr_pool_desc d, int N,
                                                                 const float *__restrict__ pos,
                                                                 const int32_t *__restrict__ batch32,
                                                                 const int64_t *__restrict__ batch64,
                                                                 int32_t *__restrict__ cluster_raw_out,
                                                                 int32_t *__restrict__ status) {
    const int n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    bool ok;
    const int b = batch32 ? batch32[n] : (int)batch64[n];
    const int raw = cluster_raw(pos[3 * (size_t)n], pos[3 * (size_t)n + 1], pos[3 * (size_t)n + 2], b, d, ok);
    if (!ok) atomicOr(status, 1);
    cluster_raw_out[n] = ok ? raw : -1;
}

// ---------------------------------------------------------------------------------------------
// pooled position of table slot `raw` from the accumulators of `pair`: mean, then round_to_pixel
// (pooling.py:47-49): floor((pos + 1e-5) / wh_inv) * wh_inv on x, y.  Every reader of a cluster's position goes
// through this one function (the node itself and, for the LUT coordinates, each of its out-edges).
__device__ __forceinline__ void cluster_pos(const PoolWs &ws, int pair, int raw, int cnt, const dagr_pool_desc &d,
                                            float (&p)[3]) {
    const long long *ps = ws_possum(ws, pair) + (size_t)raw * 3;
#pragma unroll
    for (int k = 0; k < 3; k++) p[k] = (float)(((double)ps[k] / kPosScale) / (double)cnt);
    p[0] = div_floor(p[0] + 1e-5f, d.inv_w) * d.inv_w;
    p[1] = div_floor(p[1] + 1e-5f, d.inv_h) * d.inv_h;
}

// finalize of the generic level-0 path (no neighbour codes / wide radius): one thread per (table slot, channel).
// Writes the pooled node and re-arms the slot.
__global__ __launch_bounds__(kBlock) void k_pool_finalize(dagr_pool_desc d, PoolWs ws,
                                                         const int32_t *__restrict__ batch32,
                                                         const int64_t *__restrict__ batch64,
                                                         float *__restrict__ x_out, int ldo, int xoff,
                                                         float *__restrict__ pos_out, int32_t *__restrict__ batch_out,
                                                         int32_t *__restrict__ n_out) {
    const int T = d.gx * d.gy * (d.batch_size + 1);
    const int C = d.channels;
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int raw = (int)(gid / C), ch = (int)(gid % C);
    if (raw >= T) return;
    if (gid == 0) *n_out = ws.newid[T];
    const int pair = ws_pair(ws);
    const int cnt = ws_cnt(ws, pair)[raw];
    long long *acc = ws.xacc + (size_t)raw * C + ch;
    if (cnt > 0) {
        const int c = ws.newid[raw];
        float v;
        if (d.aggr == 0) v = dec_f((int)(*acc));
        else v = (float)(((double)(*acc) / kFeatScale) / (double)cnt);
        x_out[(size_t)c * ldo + xoff + ch] = v;
        if (ch == 0) {
            float p[3];
            cluster_pos(ws, pair, raw, cnt, d, p);
            pos_out[3 * c] = p[0]; pos_out[3 * c + 1] = p[1]; pos_out[3 * c + 2] = p[2];
            const int pm = ws.perm[raw];
            batch_out[c] = batch32 ? batch32[pm] : (int)batch64[pm];
            // Net.forward concatenates pos[:, :2] to x before the next Layer (net.py:137-138): the
            // caller asks for it by reserving the two columns after the C channels.
            if (d.append_pos) {
                x_out[(size_t)c * ldo + xoff + C] = p[0];
                x_out[(size_t)c * ldo + xoff + C + 1] = p[1];
            }
        }
    }
    // re-arm this thread's own accumulator words for the next window
    *acc = 0ll;
    if (d.aggr == 0) *reinterpret_cast<int *>(acc) = kEncMin;
    if (ch == 0) {
        long long *ps = ws_possum(ws, pair) + (size_t)raw * 3;
        ps[0] = 0; ps[1] = 0; ps[2] = 0;
    }
}

__global__ void k_fill_enc_min(long long *p, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { p[i] = 0ll; *reinterpret_cast<int *>(p + i) = kEncMin; }
}

// second half of the re-arm (cnt / perm are read by every channel thread above, so clear them later)
__global__ __launch_bounds__(kBlock) void k_pool_rearm(int T, PoolWs ws) {
    const int raw = blockIdx.x * kBlock + threadIdx.x;
    if (raw >= T) return;
    ws_cnt(ws, ws_pair(ws))[raw] = 0;
    ws.perm[raw] = -1;
    ws.nbmask[raw] = 0ull;
}

// ---------------------------------------------------------------------------------------------
// level 0: fixed-stride neighbour lists.  Nodes are in pixel order, so a workgroup sweeping a contiguous
// run of nodes sees the same few (source cluster -> destination cluster) pairs over and over: a small
// LDS cache of recently inserted pairs filters them, each distinct pair of a wave is then inserted once
// by one elected lane.  (Insertion is idempotent, so cache races only cost a redundant insert.)
__global__ __launch_bounds__(kBlock) void k_coarse_edges_ell(int N, int K, const int32_t *__restrict__ nbr_src,
                                                            const int32_t *__restrict__ deg,
                                                            const int32_t *__restrict__ cluster_raw_in,
                                                            const int32_t *__restrict__ newid, int32_t *rows,
                                                            int32_t *status) {
    __shared__ unsigned long long seen[256];
    seen[threadIdx.x] = ~0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int per_iter = kBlock / K;                         // nodes per block iteration (K divides 256 for K=16)
    const int G = gridDim.x, nx = (G % 8 == 0) ? 8 : 1;
    const int xcd = blockIdx.x % nx, lb = blockIdx.x / nx, bpx = G / nx;
    const int chunk = (N + nx - 1) / nx;
    const int per_block = (chunk + bpx - 1) / bpx;
    const int n_begin = xcd * chunk + lb * per_block;
    const int n_end = min(min(N, (xcd + 1) * chunk), n_begin + per_block);
    const int sub = threadIdx.x / K, j = threadIdx.x % K;
    for (int n0 = n_begin; n0 < n_end; n0 += per_iter) {
        const int n = n0 + sub;
        int rd = -1, rs = -1;
        if (sub < per_iter && n < n_end && j < deg[n]) {
            rd = cluster_raw_in[n];
            rs = cluster_raw_in[nbr_src[(size_t)n * K + j]];
        }
        bool has = (rd != rs) && rd >= 0 && rs >= 0;
        const unsigned long long key = ((unsigned long long)(unsigned)rd << 32) | (unsigned)rs;
        const unsigned h = (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 56);
        if (has && seen[h] == key) has = false;
        unsigned long long pending = __ballot(has);
        while (pending) {
            const int leader = __ffsll((long long)pending) - 1;
            const int prd = __shfl(rd, leader, 64), prs = __shfl(rs, leader, 64);
            if (lane == leader) {
                row_insert(rows, newid[prd], newid[prs], status);
                seen[h] = key;
            }
            const bool same = has && rd == prd && rs == prs;
            pending &= ~__ballot(same);
            if (same) has = false;
        }
    }
}

// one wave per cluster row: sort the occupied slots ascending, count them
__global__ __launch_bounds__(kBlock) void k_rows_sort(int T, const int32_t *__restrict__ newid, int32_t *rows,
                                                     int32_t *__restrict__ rowcnt) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (c > T) return;
    const int nc = newid[T];
    if (c >= nc) {
        if (lane == 0 && c <= T) rowcnt[c] = 0;
        return;
    }
    int32_t *row = rows + (size_t)c * kRowSlots;
    const int v = row[lane];
    const bool valid = v >= 0;
    int rank = 0;
    for (int k = 0; k < 64; k++) {
        const int o = __shfl(v, k, 64);
        rank += (o >= 0 && o < v) ? 1 : 0;
    }
    const int count = __popcll(__ballot(valid));
    row[lane] = -1;
    __builtin_amdgcn_wave_barrier();
    if (valid) row[rank] = v;   // all reads happened before (shuffles above), same-wave ordering
    if (lane == 0) rowcnt[c] = count;
}

// rows -> CSR + LUT coordinates; re-arms the slot sets
__global__ __launch_bounds__(kBlock) void k_rows_emit(dagr_pool_desc d, int T, const int32_t *__restrict__ newid,
                                                     int32_t *rows, const int32_t *__restrict__ rowptr,
                                                     const float *__restrict__ pos_out, int32_t *__restrict__ col,
                                                     int32_t *__restrict__ code, int32_t *__restrict__ e_out,
                                                     int e_cap, int32_t *status) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int nc = newid[T];
    if (c == 0 && lane == 0) *e_out = rowptr[nc];
    if (c >= nc) return;
    int32_t *row = rows + (size_t)c * kRowSlots;
    const int cnt = rowptr[c + 1] - rowptr[c];
    const int v = row[lane];
    row[lane] = -1;
    if (lane >= cnt) return;
    const int o = rowptr[c] + lane;
    if (o >= e_cap) { atomicOr(status, 4); return; }
    // T.Cartesian(norm=True, max_value=M): (pos[src] - pos[dst]) / (2M) + 0.5 ; then the LUT index
    // of message_lut (spline_conv.py:41-42): trunc(attr * R00 + R02 + 1e-3)
    const float ax = (pos_out[3 * v] - pos_out[3 * c]) / d.two_max + 0.5f;
    const float ay = (pos_out[3 * v + 1] - pos_out[3 * c + 1]) / d.two_max + 0.5f;
    const int ix = (int)((ax * d.r00 + d.r02) + 1e-3f);
    const int iy = (int)((ay * d.r11 + d.r12) + 1e-3f);
    if (ix < 0 || ix > 2 * d.rx || iy < 0 || iy > 2 * d.ry) atomicOr(status, 8);
    col[o] = v;
    code[o] = (ix & 0xffff) | (iy << 16);
}

// LUT coordinates of an existing CSR level for a consumer whose table was built for ANOTHER domain
// (DAGR.cache_luts, dagr.py:52-62: with num_scales = 1 head "1" consumes out4 but keeps the pool3 table):
// the edge attribute is this level's T.Cartesian value, the index is message_lut's with the consumer's
// attr_remapping_matrix (spline_conv.py:41-42).  One wave per destination row.
__global__ __launch_bounds__(kBlock) void k_recode(const int32_t *__restrict__ n_ptr, int n_max,
                                                  const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                  const float *__restrict__ pos, float two_max, float r00, float r02,
                                                  float r11, float r12, int rx, int ry, int32_t *__restrict__ code,
                                                  int e_cap, int32_t *status) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int nc = min(*n_ptr, n_max);
    if (c >= nc) return;
    const int e0 = rowptr[c], e1 = min(rowptr[c + 1], e_cap);
    for (int o = e0 + lane; o < e1; o += 64) {
        const int v = col[o];
        const float ax = (pos[3 * v] - pos[3 * c]) / two_max + 0.5f;
        const float ay = (pos[3 * v + 1] - pos[3 * c + 1]) / two_max + 0.5f;
        const int ix = (int)((ax * r00 + r02) + 1e-3f);
        const int iy = (int)((ay * r11 + r12) + 1e-3f);
        if (ix < 0 || ix > 2 * rx || iy < 0 || iy > 2 * ry) atomicOr(status, 8);
        code[o] = (ix & 0xffff) | (iy << 16);
    }
}

// ---------------------------------------------------------------------------------------------
// launch (S): scans the occupancy flags (-> consecutive ids, torch.unique's order) and the row sizes (-> CSR row pointers of
// the relabelled clusters) of the T + 1 table slots together; clears both inputs, closes the epoch.  Row sizes: the insert
// counters (pooled levels) or the population of the cell bitmaps (level 0).
// KEEP (asynchronous updates, dagr_pool_l0_stream): the accumulators stay as they are -- nothing is cleared and the epoch
// stays open, so that later micro-batches keep adding to the same voxels.
// ceil((T + 1) / 2048) workgroups in ONE launch (until round 4 one workgroup walked the table in rounds: 28 us for the
// 20 k slots of a B = 8 level-0 table, a single CU issuing every load, clear and store; now ~7 us).  Decoupled look-back: a workgroup takes the next tile (ticket
// counter: tiles are handed out in the order workgroups start, so a tile's predecessors are always running or done),
// scans it, publishes its totals in one 64-bit word -- launch tag (12 bits) | flag (1 = tile totals, 2 = totals of all
// tiles up to here) | clusters (22 bits) | edges (28 bits) -- and a wave looks back over its predecessors' words until
// it meets an inclusive one.  The tile that takes the last ticket writes the counts, the empty tail rows, closes the
// epoch and re-arms ticket counter and tag.  Integers throughout: the result is the single-workgroup kernel's, bit for
// bit.
template <bool MASKS, bool KEEP = false>
__global__ __launch_bounds__(kBlock) void k_pool_scan_chained(PoolWs ws, int32_t *__restrict__ n_out,
                                                             int32_t *__restrict__ rowptr_out,
                                                             int32_t *__restrict__ e_out) {
    constexpr int PER = kPoolScanTile / kBlock;   // 8
    __shared__ __align__(16) unsigned char st_occ[kPoolScanTile], st_cnt[kPoolScanTile];
    __shared__ int sm_scan[8];
    __shared__ int sh_tile, sh_base_occ, sh_base_cnt;
    __shared__ unsigned sh_tag;
    const int n = ws.T + 1;
    const int ntiles = (n + kPoolScanTile - 1) / kPoolScanTile;
    if (threadIdx.x == 0) {
        sh_tile = atomicAdd(&ws.status[7], 1);
        sh_tag = (unsigned)__hip_atomic_load(&ws.status[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xfffu;
    }
    __syncthreads();
    const int tile = sh_tile;
    const unsigned long long tag = (unsigned long long)sh_tag << 52;
    const int base = tile * kPoolScanTile;
    // load phase, coalesced: thread t takes elements base + 256 j + t; parked as bytes in LDS (a flag is 0/1, a row size or
    // bitmap population is <= 64), then every thread scans 8 CONSECUTIVE slots
    {
        int ov[PER], cv[PER];
        unsigned long long mv[PER];
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const int e = base + j * kBlock + (int)threadIdx.x;
            ov[j] = (e < n) ? ws.occupied[e] : 0;
            if (MASKS) mv[j] = (e < ws.T) ? ws.nbmask[e] : 0ull;
            else cv[j] = (e < n) ? ws.rowcnt[e] : 0;
        }
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const int e = base + j * kBlock + (int)threadIdx.x;
            if (e < n && !KEEP) {
                ws.occupied[e] = 0;
                if (!MASKS) ws.rowcnt[e] = 0;
            }
            const int c = MASKS ? __popcll(mv[j]) : cv[j];
            st_occ[j * kBlock + threadIdx.x] = (unsigned char)ov[j];
            st_cnt[j * kBlock + threadIdx.x] = (unsigned char)min(c, 255);
        }
    }
    __syncthreads();
    int occ[PER], cnt[PER];
    {
        const uint2 po = *reinterpret_cast<const uint2 *>(st_occ + threadIdx.x * PER);
        const uint2 pc = *reinterpret_cast<const uint2 *>(st_cnt + threadIdx.x * PER);
        const unsigned wo[2] = {po.x, po.y}, wc[2] = {pc.x, pc.y};
#pragma unroll
        for (int k = 0; k < PER; k++) {
            occ[k] = (wo[k >> 2] >> (8 * (k & 3))) & 0xff;
            cnt[k] = (wc[k >> 2] >> (8 * (k & 3))) & 0xff;
        }
    }
    int s_occ = 0, s_cnt = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) { s_occ += occ[k]; s_cnt += cnt[k]; }
    int t_occ, t_cnt;
    const int x_occ = block_exclusive_scan(s_occ, sm_scan, t_occ);
    const int x_cnt = block_exclusive_scan(s_cnt, sm_scan + 4, t_cnt);
    // publish this tile's totals, then look back (wave 0): lane l reads the word of tile - 1 - l
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        auto pack = [&](int flag, int o, int c) {
            return tag | ((unsigned long long)flag << 50) | ((unsigned long long)(unsigned)o << 28) | (unsigned long long)(unsigned)c;
        };
        if (lane == 0)
            __hip_atomic_store(&ws.tile_state[tile], pack(tile == 0 ? 2 : 1, t_occ, t_cnt), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        int b_occ = 0, b_cnt = 0;
        for (int hi = tile - 1; hi >= 0;) {        // window of predecessors hi, hi - 1, ..., hi - 63
            const int pidx = hi - lane;
            unsigned long long wd = 0;
            bool ready = pidx < 0;
            while (!__all(ready)) {                // spin until every predecessor of the window has published
                if (!ready) {
                    wd = __hip_atomic_load(&ws.tile_state[pidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ready = (wd >> 52) == (unsigned long long)sh_tag && ((wd >> 50) & 3ull) != 0ull;
                }
            }
            const bool incl = pidx >= 0 && ((wd >> 50) & 3ull) == 2ull;
            const unsigned long long im = __ballot(incl);
            const int stop = im ? (__ffsll((long long)im) - 1) : 63;     // nearest predecessor with inclusive totals
            int vo = (pidx >= 0 && lane <= stop) ? (int)((wd >> 28) & 0x3fffffull) : 0;
            int vc = (pidx >= 0 && lane <= stop) ? (int)(wd & 0xfffffffull) : 0;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { vo += __shfl_xor(vo, off, 64); vc += __shfl_xor(vc, off, 64); }
            b_occ += vo; b_cnt += vc;
            if (im) break;
            hi -= 64;
        }
        if (lane == 0) {
            sh_base_occ = b_occ; sh_base_cnt = b_cnt;
            if (tile > 0)
                __hip_atomic_store(&ws.tile_state[tile], pack(2, b_occ + t_occ, b_cnt + t_cnt), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    int e_occ = sh_base_occ + x_occ, e_cnt = sh_base_cnt + x_cnt;
    const int i0 = base + threadIdx.x * PER;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        if (i0 + k < n) {
            ws.newid[i0 + k] = e_occ;
            if (occ[k]) rowptr_out[e_occ] = e_cnt;
        }
        e_occ += occ[k]; e_cnt += cnt[k];
    }
    if (tile == ntiles - 1) {
        // the last ticket: every other tile has published its totals before this one could finish its look-back
        const int nc = sh_base_occ + t_occ, ne = sh_base_cnt + t_cnt;
        for (int c = nc + threadIdx.x; c <= ws.T; c += kBlock) rowptr_out[c] = ne;   // rows past the last cluster are empty
        if (threadIdx.x == 0) {
            *n_out = nc;
            *e_out = ne;
            if (!KEEP) ws.status[4] ^= 1;   // the next call fills the other accumulator pair; launch (C) reads the one just filled
            ws.status[7] = 0;
            ws.status[6] = (int)((sh_tag + 1u) & 0xfffu);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// launch (C): one wave per table slot.  Writes the pooled node (features: lanes over channels), turns the slot's
// source set -- 64 hashed raw ids (pooled levels) or the two cell bitmaps (level 0) -- into its sorted CSR row with the
// LUT coordinate of every edge, and re-arms what the slot owns: its feature accumulators, perm, its set, and the
// position/count accumulators of the OTHER pair (the one the previous call left behind).
template <bool MASKS, bool KEEP = false>
__global__ __launch_bounds__(kBlock) void k_pool_emit(dagr_pool_desc d, PoolWs ws, const int32_t *__restrict__ batch32,
                                                     const int64_t *__restrict__ batch64, float *__restrict__ x_out,
                                                     int ldo, int xoff, float *__restrict__ pos_out,
                                                     int32_t *__restrict__ batch_out,
                                                     const int32_t *__restrict__ rowptr_out, int32_t *__restrict__ col,
                                                     int32_t *__restrict__ code, int e_cap) {
    const int lane = threadIdx.x & 63;
    const int raw = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (raw >= ws.T) return;
    const int pair = KEEP ? ws_pair(ws) : ws_pair(ws) ^ 1;     // the scan has closed the epoch (KEEP: it stays open)
    const int cnt = ws_cnt(ws, pair)[raw];
    if (!KEEP) {
        if (lane == 0) ws_cnt(ws, pair ^ 1)[raw] = 0;
        if (lane < 3) ws_possum(ws, pair ^ 1)[(size_t)raw * 3 + lane] = 0ll;
    }
    if (cnt == 0) return;                      // empty slot: nothing was accumulated, no set entries
    const int C = d.channels;
    const int c = ws.newid[raw];
    for (int ch = lane; ch < C; ch += 64) {
        long long *acc = ws.xacc + (size_t)raw * C + ch;
        float v;
        if (d.aggr == 0) v = dec_f((int)(*acc));
        else v = (float)(((double)(*acc) / kFeatScale) / (double)cnt);
        x_out[(size_t)c * ldo + xoff + ch] = v;
        if (!KEEP) {
            *acc = 0ll;
            if (d.aggr == 0) *reinterpret_cast<int *>(acc) = kEncMin;
        }
    }
    float p[3];
    cluster_pos(ws, pair, raw, cnt, d, p);
    if (lane == 0) {
        pos_out[3 * c] = p[0]; pos_out[3 * c + 1] = p[1]; pos_out[3 * c + 2] = p[2];
        const int pm = ws.perm[raw];
        // the sample of the member with the largest index (consecutive_cluster's perm); with resident accumulators
        // (KEEP) the slot holds (sample + 1) << 26 | event id, so that "largest" does not depend on the order in which
        // the samples' events arrived: a cluster shared by a sample's t == 1.0 nodes and the next sample's nodes
        // (QUIRK-1) takes the next sample, as it does in a window, where samples are concatenated in order
        batch_out[c] = KEEP ? (pm >> 26) - 1 : (batch32 ? batch32[pm] : (int)batch64[pm]);
        if (d.append_pos) {       // net.py:137-138, see k_pool_finalize
            x_out[(size_t)c * ldo + xoff + C] = p[0];
            x_out[(size_t)c * ldo + xoff + C + 1] = p[1];
        }
        if (!KEEP) ws.perm[raw] = -1;
    }
    // the slot's sources in ascending raw id (= ascending new id), one per lane
    int v = -1, rank = 0;
    if (MASKS) {
        const unsigned long long m = ws.nbmask[raw];
        if (lane == 0 && !KEEP) ws.nbmask[raw] = 0ull;
        // bit j of `ord` = j-th candidate in ascending id: the 25 cells of the plane below, then the 25 of this plane
        const unsigned long long ord = ((m >> 32) & 0x1ffffffull) | ((m & 0x1ffffffull) << 25);
        if (lane < 50 && ((ord >> lane) & 1ull)) {
            const int k = lane < 25 ? lane : lane - 25;
            v = raw + (k % 5 - 2) + d.gx * (k / 5 - 2) - (lane < 25 ? d.gx * d.gy : 0);
            rank = __popcll(ord & ((1ull << lane) - 1ull));
        }
    } else {
        int32_t *row = ws.rows + (size_t)raw * kRowSlots;
        v = row[lane];
        row[lane] = -1;
        for (int k = 0; k < 64; k++) {
            const int o = __shfl(v, k, 64);
            rank += (o >= 0 && o < v) ? 1 : 0;
        }
    }
    if (v < 0) return;
    const int o = rowptr_out[c] + rank;
    if (o >= e_cap) { atomicOr(ws.status, 4); return; }
    float q[3];
    cluster_pos(ws, pair, v, ws_cnt(ws, pair)[v], d, q);
    // T.Cartesian(norm=True, max_value=M): (pos[src] - pos[dst]) / (2M) + 0.5 ; then the LUT index
    // of message_lut (spline_conv.py:41-42): trunc(attr * R00 + R02 + 1e-3)
    const float ax = (q[0] - p[0]) / d.two_max + 0.5f;
    const float ay = (q[1] - p[1]) / d.two_max + 0.5f;
    const int ix = (int)((ax * d.r00 + d.r02) + 1e-3f);
    const int iy = (int)((ay * d.r11 + d.r12) + 1e-3f);
    if (ix < 0 || ix > 2 * d.rx || iy < 0 || iy > 2 * d.ry) atomicOr(ws.status, 8);
    col[o] = ws.newid[v];
    code[o] = (ix & 0xffff) | (iy << 16);
}

// ---------------------------------------------------------------------------------------------
// Asynchronous updates: level-0 rows appended to a resident window (async_update.hip) join the accumulators of their
// voxels.  16 lanes per new node: lanes over channels (features) and over its <= 16 in-edges (the source-cell bitmaps);
// same quantities as pool_l0_cell, by atomics (a micro-batch is small).  The new node's pixel comes from its own
// position (denormalised as the graph builder does it); its sources' pixels from the offset codes of its row.
__global__ __launch_bounds__(kBlock) void k_pool_l0_add_rows(dagr_pool_desc d, int W, int H, int first_row, int n_rows,
                                                            const int32_t *__restrict__ xlo,
                                                            const int32_t *__restrict__ ylo,
                                                            const float *__restrict__ x, int ldx,
                                                            const float *__restrict__ pos,
                                                            const int32_t *__restrict__ batch_events, PoolWs ws,
                                                            const int16_t *__restrict__ nbr_code,
                                                            const int32_t *__restrict__ nbr_src,
                                                            const int32_t *__restrict__ deg, int K, int r) {
    const int l = threadIdx.x & 15;
    const int i = (blockIdx.x * kBlock + threadIdx.x) >> 4;
    if (i >= n_rows) return;
    const int s = first_row + i;          // an appended event's node row is its event id
    const float px = pos[3 * (size_t)s], py = pos[3 * (size_t)s + 1], pt = pos[3 * (size_t)s + 2];
    const int b = batch_events[s];
    bool ok;
    const int raw = cluster_raw(px, py, pt, b, d, ok);      // a t == 1.0 node lands one sample plane up (QUIRK-1)
    if (!ok) {
        if (l == 0) atomicOr(&ws.status[0], 1);
        return;
    }
    const bool leak = pt >= 1.0f;
    const int cells = d.gx * d.gy;
    const int C = d.channels;
    const int pair = ws_pair(ws);
    for (int ch = l; ch < C; ch += 16) {
        const float v = x[(size_t)s * ldx + ch];
        if (d.aggr == 0) atomicMax(reinterpret_cast<int *>(ws.xacc + (size_t)raw * C + ch), enc_f(v));
        else atomicAdd(reinterpret_cast<unsigned long long *>(ws.xacc + (size_t)raw * C + ch),
                       (unsigned long long)(long long)llrint((double)v * kFeatScale));
    }
    if (l == 0) {
        ws.occupied[raw] = 1;
        atomicAdd(&ws_cnt(ws, pair)[raw], 1);
        atomicMax(&ws.perm[raw], ((b + 1) << 26) | s);
        atomicAdd(reinterpret_cast<unsigned long long *>(ws_possum(ws, pair) + (size_t)raw * 3 + 0),
                  (unsigned long long)(long long)llrint((double)px * kPosScale));
        atomicAdd(reinterpret_cast<unsigned long long *>(ws_possum(ws, pair) + (size_t)raw * 3 + 1),
                  (unsigned long long)(long long)llrint((double)py * kPosScale));
        atomicAdd(reinterpret_cast<unsigned long long *>(ws_possum(ws, pair) + (size_t)raw * 3 + 2),
                  (unsigned long long)(long long)llrint((double)pt * kPosScale));
    }
    if (nbr_code && l < deg[s]) {
        const int xpix = (int)((float)W * px + 1e-3f), ypix = (int)((float)H * py + 1e-3f);   // ev_tgn.py:11-16
        const int cx = (raw % cells) % d.gx, cy = (raw % cells) / d.gx;
        const int side = 2 * r + 1;
        const int code = nbr_code[(size_t)s * K + l];
        const int ox = code / side, oy = code - ox * side;
        const int xs = xpix + ox - r, ys = ypix + oy - r;
        int dcx = 0, dcy = 0;          // source cell = cx - 2 + #(lower bounds of cells cx-1 .. cx+2 that are <= xs)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int qx = cx - 1 + q, qy = cy - 1 + q;
            const int bxq = qx <= 0 ? INT_MIN : (qx >= d.gx ? INT_MAX : xlo[qx]);
            const int byq = qy <= 0 ? INT_MIN : (qy >= d.gy ? INT_MAX : ylo[qy]);
            dcx += xs >= bxq;
            dcy += ys >= byq;
        }
        const int bit = dcy * 5 + dcx;
        if (!leak) {
            if (bit != 12) atomicOr(&ws.nbmask[raw], 1ull << bit);                 // own cell = self loops
        } else {
            // raw already is the slot one plane up: sources that are t == 1.0 nodes themselves sit in its own plane
            // (bits 0..24; own cell = self loops), the others in the plane below (bits 32..56, any of the 25 cells)
            const int src = nbr_src[(size_t)s * K + l];
            if (pos[3 * (size_t)src + 2] >= 1.0f) {
                if (bit != 12) atomicOr(&ws.nbmask[raw], 1ull << bit);
            } else {
                atomicOr(&ws.nbmask[raw], 1ull << (32 + bit));
            }
        }
    }
}

// resident accumulators: perm (largest member event id, as the window kernels leave it) -> (sample + 1) << 26 | id
__global__ __launch_bounds__(kBlock) void k_pool_perm_keys(int T, const int32_t *__restrict__ batch_events, PoolWs ws) {
    const int raw = blockIdx.x * kBlock + threadIdx.x;
    if (raw >= T) return;
    const int pm = ws.perm[raw];
    if (pm >= 0) ws.perm[raw] = ((batch_events[pm] + 1) << 26) | pm;
}

}  // namespace
}  // namespace dagr

using namespace dagr;

namespace {
int validate_pool(const dagr_pool_desc *d) {
    DAGR_CHECK_ARG(d != nullptr, "desc is NULL");
    DAGR_CHECK_ARG(d->gx > 0 && d->gy > 0 && d->batch_size > 0 && d->channels > 0, "bad sizes");
    DAGR_CHECK_ARG((int64_t)d->gx * d->gy * (d->batch_size + 1) < (1 << 22), "voxel table too large");
    DAGR_CHECK_ARG(d->aggr == 0 || d->aggr == 1, "aggr must be 0 (max) or 1 (mean)");
    DAGR_CHECK_ARG(d->vx > 0 && d->vy > 0 && d->two_max > 0, "bad voxel size / cartesian max");
    return DAGR_OK;
}

// level-0 accumulation: k_pool_l0_slots over the graph builder's slot arrays.  The grid is sized for `n_cap` nodes; the
// kernel reads the node count from the builder's workspace (start[P]) and shares the nodes that are there among all
// workgroups, so a launch captured in a HIP graph serves windows of any size up to n_cap.
int launch_pool_l0_slots(const dagr_pool_desc *desc, const dagr_graph_desc *gdesc, void *graph_ws, const int32_t *xlo,
                         const int32_t *ylo, const float *x, int ldx, const float *pos, const PoolWs &ws,
                         const int16_t *nbr_code, const int32_t *nbr_src, const int32_t *deg, int64_t n_cap,
                         hipStream_t stream) {
    PixelIndex ix;
    graph_ws_index(gdesc, graph_ws, &ix);
    const int32_t *start = ix.start; const int2 *slot_it = ix.slot_it;
    const int32_t *slot_xyb = ix.slot_xyb;
    const int32_t *n_ptr = ix.n_nodes;
    const int row_keys = ix.W;                // keys per pixel row: the band of a voxel row starts at a row's first key
    const int C = desc->channels, W = gdesc->width, H = gdesc->height, K = gdesc->max_neighbors;
    const bool vec4 = C % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)x & 15) == 0;
    // LDS window: up to 60 KB of accumulators per (16-wave) workgroup, at least one voxel row, at most 1024 table slots
    const size_t per_slot = (size_t)C * (desc->aggr == 0 ? 4 : 8) + 36;
    const size_t budget = (size_t)60 * 1024 - (size_t)(W + H) * 2 - 2048 - 1024;
    int VW = (int)std::min<size_t>(1024, budget / per_slot);
    VW = std::max(VW, desc->gx) / 2 * 2 + 2;
    const size_t lds = desc->aggr == 0 ? pool_l0_lds_bytes<0>(VW, C, W, H) : pool_l0_lds_bytes<1>(VW, C, W, H);
    DAGR_CHECK_ARG(lds <= 64 * 1024, "level-0 pooling: one voxel row of accumulators does not fit the LDS window");
    DAGR_CHECK_ARG(desc->gx < 65536 && desc->gy < 65536 && VW < 32768, "voxel grid too large for the level-0 pooling kernel");
    // one 16-wave workgroup per CU is resident at a time (114 registers); wide rows run two rounds of shorter runs, which
    // overlaps one round's merge with the other's streaming (measured: 112 vs 127 us at 80 channels, 47 vs 44 us at 16)
    static const int gmult_env = (int)knob("DAGR_POOL_GRID_MULT", 0);
    const int gmult = gmult_env > 0 ? gmult_env : (C > 32 ? 2 : 1);
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((int64_t)gmult * device_cu_count(),
                                                                          ceil_div(n_cap, kPoolL0Block)));
#define DAGR_POOL_L0_LAUNCH(AG, VEC)                                                                                  \
    k_pool_l0_slots<AG, VEC><<<grid, kPoolL0Block, lds, stream>>>(*desc, W, H, (int)n_cap, VW, n_ptr, xlo, ylo, start, row_keys, slot_it, \
                                                                   slot_xyb, x, ldx, pos, ws, nbr_code, nbr_src, deg, K, \
                                                                   gdesc->radius)
    if (desc->aggr == 0) { if (vec4) DAGR_POOL_L0_LAUNCH(0, 4); else DAGR_POOL_L0_LAUNCH(0, 1); }
    else                 { if (vec4) DAGR_POOL_L0_LAUNCH(1, 4); else DAGR_POOL_L0_LAUNCH(1, 1); }
#undef DAGR_POOL_L0_LAUNCH
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}
}  // namespace

extern "C" {

size_t dagr_pool_workspace_bytes(const dagr_pool_desc *desc) {
    if (validate_pool(desc) != DAGR_OK) return 0;
    return pool_carve(*desc, nullptr, nullptr);
}

int dagr_pool_workspace_init(const dagr_pool_desc *desc, void *workspace, size_t bytes, void *stream_) {
    int rc = validate_pool(desc);
    if (rc != DAGR_OK) return rc;
    DAGR_CHECK_ARG(workspace, "workspace is NULL");
    PoolWs ws;
    if (pool_carve(*desc, (char *)workspace, &ws) > bytes) {
        set_error("dagr_pool_workspace_init: workspace too small");
        return DAGR_ERR_WORKSPACE;
    }
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t T = (int64_t)desc->gx * desc->gy * (desc->batch_size + 1);
    DAGR_CHECK_HIP(hipMemsetAsync(ws.occupied, 0, (T + 32) * 4, stream));
    DAGR_CHECK_HIP(hipMemsetAsync(ws.newid, 0, (T + 32) * 4, stream));
    DAGR_CHECK_HIP(hipMemsetAsync(ws.possum, 0, 2 * T * 3 * 8, stream));
    DAGR_CHECK_HIP(hipMemsetAsync(ws.cnt, 0, 2 * T * 4, stream));
    DAGR_CHECK_HIP(hipMemsetAsync(ws.perm, 0xff, T * 4, stream));
    DAGR_CHECK_HIP(hipMemsetAsync(ws.rows, 0xff, T * (size_t)kRowSlots * 4, stream));
    DAGR_CHECK_HIP(hipMemsetAsync(ws.rowcnt, 0, (T + 32) * 4, stream));
    DAGR_CHECK_HIP(hipMemsetAsync(ws.status, 0, 32, stream));
    DAGR_CHECK_HIP(hipMemsetAsync(ws.tile_state, 0, ((T + 1 + kPoolScanTile - 1) / kPoolScanTile + 8) * 8, stream));
    DAGR_CHECK_HIP(hipMemsetAsync(ws.nbmask, 0, (T + 9) * 8, stream));
    // feature accumulators: ordered-int minimum for max, 0 for mean
    {
        const size_t n = T * (size_t)desc->channels;
        if (desc->aggr == 0) {
            k_fill_enc_min<<<(unsigned)ceil_div(n, 256), 256, 0, stream>>>(ws.xacc, n);
            DAGR_CHECK_LAUNCH();
        } else {
            DAGR_CHECK_HIP(hipMemsetAsync(ws.xacc, 0, n * 8, stream));
        }
    }
    return DAGR_OK;
}


// launch (S): one workgroup per 2048 table slots
#define DAGR_POOL_SCAN(MASKS, KEEP)                                                                                   \
    k_pool_scan_chained<MASKS, KEEP><<<(unsigned)ceil_div(T + 1, kPoolScanTile), kBlock, 0, stream>>>(ws, n_out, rowptr_out, e_out)

static int pool_tail(const dagr_pool_desc *d, PoolWs &ws, const int32_t *batch32, const int64_t *batch64,
                     float *x_out, int ldo, int xoff, float *pos_out, int32_t *batch_out, int32_t *n_out,
                     int32_t *rowptr_out, int32_t *col_out, int32_t *code_out, int32_t *e_out, int e_cap,
                     hipStream_t stream) {
    const int T = d->gx * d->gy * (d->batch_size + 1);
    const int wpb = kBlock / 64;
    k_rows_sort<<<(unsigned)ceil_div(T + 1, wpb), kBlock, 0, stream>>>(T, ws.newid, ws.rows, ws.rowcnt);
    DAGR_CHECK_LAUNCH();
    DAGR_CHECK_HIP(exclusive_scan_i32(ws.rowcnt, rowptr_out, T + 1, ws.scan_tmp, false, stream));
    k_rows_emit<<<(unsigned)ceil_div(T, wpb), kBlock, 0, stream>>>(*d, T, ws.newid, ws.rows, rowptr_out, pos_out,
                                                                  col_out, code_out, e_out, e_cap, ws.status);
    DAGR_CHECK_LAUNCH();
    k_pool_rearm<<<(unsigned)ceil_div(T, kBlock), kBlock, 0, stream>>>(T, ws);
    DAGR_CHECK_LAUNCH();
    (void)batch32; (void)batch64; (void)x_out; (void)ldo; (void)xoff; (void)batch_out; (void)n_out;
    return DAGR_OK;
}

int dagr_pool_l0(const dagr_pool_desc *desc, void *pool_ws, const dagr_graph_desc *gdesc, void *graph_ws,
                 const int32_t *xlo, const int32_t *ylo, const float *x, int32_t ldx, const float *pos,
                 const int32_t *batch_nodes, const void *batch, int32_t batch_is_int64, int64_t N,
                 const int32_t *nbr_src, const int16_t *nbr_code, const int32_t *deg,
                 int32_t *cluster_scratch, float *x_out, int32_t ldo, int32_t xoff, float *pos_out,
                 int32_t *batch_out, int32_t *n_out, int32_t *rowptr_out, int32_t *col_out, int32_t *code_out,
                 int32_t *e_out, int32_t e_cap, void *stream_) {
    int rc = validate_pool(desc);
    if (rc != DAGR_OK) return rc;
    DAGR_CHECK_ARG(pool_ws && gdesc && graph_ws, "NULL workspace/desc");
    DAGR_CHECK_ARG(desc->channels <= kMaxL0Channels, "too many channels for the level-0 pooling kernel");
    DAGR_CHECK_ARG(desc->batch_size == gdesc->batch_size, "batch_size mismatch");
    DAGR_CHECK_ARG(n_out && rowptr_out && e_out, "NULL output");
    hipStream_t stream = (hipStream_t)stream_;
    PoolWs ws;
    pool_carve(*desc, (char *)pool_ws, &ws);
    const int T = desc->gx * desc->gy * (desc->batch_size + 1);
    const int32_t *b32 = batch_is_int64 ? nullptr : (const int32_t *)batch;
    const int64_t *b64 = batch_is_int64 ? (const int64_t *)batch : nullptr;
    // coarse edges from per-cell bitmaps when a source can be at most two cells away from its destination
    // (cells are floor or ceil of W*vx pixels wide) and the neighbour lists carry their offset codes
    const int K = gdesc->max_neighbors;
    const int cell_w = (int)floorf(desc->vx * (float)gdesc->width), cell_h = (int)floorf(desc->vy * (float)gdesc->height);
    const bool fast_edges = nbr_code != nullptr && K <= 16 && gdesc->radius <= 2 * std::min(cell_w, cell_h) &&
                            gdesc->width <= 4096;
    if (N > 0) {
        DAGR_CHECK_ARG(xlo && ylo && x && pos && batch_nodes && batch && nbr_src && deg && cluster_scratch, "NULL input");
        rc = launch_pool_l0_slots(desc, gdesc, graph_ws, xlo, ylo, x, ldx, pos, ws, fast_edges ? nbr_code : nullptr, nbr_src,
                                  deg, N, stream);
        if (rc != DAGR_OK) return rc;
    }
    if (fast_edges) {
        // (S) ids + row pointers from the occupancy flags and the bitmap populations, (C) nodes + CSR rows
        DAGR_POOL_SCAN(true, false);
        DAGR_CHECK_LAUNCH();
        k_pool_emit<true><<<(unsigned)ceil_div(T, kBlock / 64), kBlock, 0, stream>>>(
            *desc, ws, b32, b64, x_out, ldo, xoff, pos_out, batch_out, rowptr_out, col_out, code_out, e_cap);
        DAGR_CHECK_LAUNCH();
        return DAGR_OK;
    }
    // generic path (no neighbour codes, or sources more than two cells away): relabel, finalize, hash the edges
    DAGR_CHECK_HIP(exclusive_scan_i32(ws.occupied, ws.newid, T + 1, ws.scan_tmp, true, stream));
    k_pool_finalize<<<(unsigned)ceil_div((int64_t)T * desc->channels, kBlock), kBlock, 0, stream>>>(
        *desc, ws, b32, b64, x_out, ldo, xoff, pos_out, batch_out, n_out);
    DAGR_CHECK_LAUNCH();
    if (N > 0) {
        k_pool_l0_event_cluster<<<(unsigned)ceil_div(N, kBlock), kBlock, 0, stream>>>(
            *desc, (int)N, pos, batch_nodes, nullptr, cluster_scratch, ws.status);
        DAGR_CHECK_LAUNCH();
        DAGR_CHECK_ARG(K <= kBlock, "max_neighbors too large");
        const unsigned gE = round_grid8(std::min<int64_t>(ceil_div(N, kBlock / K), 256 * 8));
        k_coarse_edges_ell<<<gE, kBlock, 0, stream>>>((int)N, K, nbr_src, deg, cluster_scratch, ws.newid, ws.rows,
                                                      ws.status);
        DAGR_CHECK_LAUNCH();
    }
    return pool_tail(desc, ws, b32, b64, x_out, ldo, xoff, pos_out, batch_out, n_out, rowptr_out, col_out, code_out,
                     e_out, e_cap, stream);
}

int dagr_pool_l0_accumulate(const dagr_pool_desc *desc, void *pool_ws, const dagr_graph_desc *gdesc, void *graph_ws,
                            const int32_t *xlo, const int32_t *ylo, const float *x, int32_t ldx, const float *pos,
                            int64_t N, const int32_t *nbr_src, const int16_t *nbr_code, const int32_t *deg,
                            void *stream_) {
    int rc = validate_pool(desc);
    if (rc != DAGR_OK) return rc;
    DAGR_CHECK_ARG(pool_ws && gdesc && graph_ws && xlo && ylo && x && pos && nbr_src && deg && N > 0, "bad arguments");
    DAGR_CHECK_ARG(desc->channels <= kMaxL0Channels && desc->batch_size == gdesc->batch_size, "bad descriptors");
    PoolWs ws;
    pool_carve(*desc, (char *)pool_ws, &ws);
    const int K = gdesc->max_neighbors;
    const int cell_w = (int)floorf(desc->vx * (float)gdesc->width), cell_h = (int)floorf(desc->vy * (float)gdesc->height);
    const bool fast_edges = nbr_code != nullptr && K <= 16 && gdesc->radius <= 2 * std::min(cell_w, cell_h) &&
                            gdesc->width <= 4096;
    return launch_pool_l0_slots(desc, gdesc, graph_ws, xlo, ylo, x, ldx, pos, ws, fast_edges ? nbr_code : nullptr, nbr_src,
                                deg, N, (hipStream_t)stream_);
}

int dagr_pool_l0_stream(const dagr_pool_desc *desc, void *pool_ws, int32_t rebuild, const dagr_graph_desc *gdesc,
                        void *graph_ws, const int32_t *xlo, const int32_t *ylo, const float *x, int32_t ldx, const float *pos,
                        const int32_t *batch_events, int64_t n_window, int64_t first_row, int64_t n_rows,
                        const int32_t *nbr_src, const int16_t *nbr_code, const int32_t *deg, float *x_out, int32_t ldo,
                        int32_t xoff, float *pos_out, int32_t *batch_out, int32_t *n_out, int32_t *rowptr_out,
                        int32_t *col_out, int32_t *code_out, int32_t *e_out, int32_t e_cap, void *stream_) {
    int rc = validate_pool(desc);
    if (rc != DAGR_OK) return rc;
    DAGR_CHECK_ARG(pool_ws && gdesc && graph_ws, "NULL workspace/desc");
    DAGR_CHECK_ARG(desc->channels <= kMaxL0Channels, "too many channels for the level-0 pooling kernel");
    DAGR_CHECK_ARG(desc->batch_size == gdesc->batch_size, "batch_size mismatch");
    DAGR_CHECK_ARG(n_out && rowptr_out && e_out && xlo && ylo && x && pos && batch_events && nbr_src && nbr_code && deg,
                   "NULL pointer");
    DAGR_CHECK_ARG(n_window >= 0 && first_row >= n_window && n_rows >= 0 && first_row + n_rows < (1 << 26) &&
                       desc->batch_size < 30, "bad row ranges");
    hipStream_t stream = (hipStream_t)stream_;
    PoolWs ws;
    pool_carve(*desc, (char *)pool_ws, &ws);
    const int T = desc->gx * desc->gy * (desc->batch_size + 1);
    const int K = gdesc->max_neighbors;
    const int cell_w = (int)floorf(desc->vx * (float)gdesc->width), cell_h = (int)floorf(desc->vy * (float)gdesc->height);
    DAGR_CHECK_ARG(K <= 16 && gdesc->radius <= 2 * std::min(cell_w, cell_h) && gdesc->width <= 4096,
                   "asynchronous pooling keeps its coarse edges as cell bitmaps: the search radius must span at most two cells");
    if (rebuild) {
        // the accumulators of the resident window, from scratch (this workspace is the asynchronous mode's own)
        rc = dagr_pool_workspace_init(desc, pool_ws, pool_carve(*desc, nullptr, nullptr), stream_);
        if (rc != DAGR_OK) return rc;
        if (n_window > 0) {
            rc = launch_pool_l0_slots(desc, gdesc, graph_ws, xlo, ylo, x, ldx, pos, ws, nbr_code, nbr_src, deg, n_window,
                                      stream);
            if (rc != DAGR_OK) return rc;
            k_pool_perm_keys<<<(unsigned)ceil_div(T, kBlock), kBlock, 0, stream>>>(T, batch_events, ws);
            DAGR_CHECK_LAUNCH();
        }
    }
    if (n_rows > 0) {
        k_pool_l0_add_rows<<<(unsigned)ceil_div(n_rows * 16, kBlock), kBlock, 0, stream>>>(
            *desc, gdesc->width, gdesc->height, (int)first_row, (int)n_rows, xlo, ylo, x, ldx, pos, batch_events, ws, nbr_code,
            nbr_src, deg, K, gdesc->radius);
        DAGR_CHECK_LAUNCH();
    }
    DAGR_POOL_SCAN(true, true);
    DAGR_CHECK_LAUNCH();
    k_pool_emit<true, true><<<(unsigned)ceil_div(T, kBlock / 64), kBlock, 0, stream>>>(
        *desc, ws, batch_events, nullptr, x_out, ldo, xoff, pos_out, batch_out, rowptr_out, col_out, code_out, e_cap);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

int dagr_pool_csr(const dagr_pool_desc *desc, void *pool_ws, const int32_t *n_ptr, int32_t n_max, const float *x,
                  int32_t ldx, const float *pos, const int32_t *batch, const int32_t *rowptr, const int32_t *col,
                  int32_t *cluster_scratch, float *x_out, int32_t ldo, int32_t xoff, float *pos_out,
                  int32_t *batch_out, int32_t *n_out, int32_t *rowptr_out, int32_t *col_out, int32_t *code_out,
                  int32_t *e_out, int32_t e_cap, void *stream_) {
    int rc = validate_pool(desc);
    if (rc != DAGR_OK) return rc;
    DAGR_CHECK_ARG(pool_ws, "NULL workspace");
    DAGR_CHECK_ARG(n_out && rowptr_out && e_out, "NULL output");
    hipStream_t stream = (hipStream_t)stream_;
    PoolWs ws;
    pool_carve(*desc, (char *)pool_ws, &ws);
    const int T = desc->gx * desc->gy * (desc->batch_size + 1);
    if (n_max > 0) {
        DAGR_CHECK_ARG(x && pos && batch && rowptr && col && cluster_scratch, "NULL input");
        k_pool_accumulate<<<(unsigned)ceil_div((int64_t)n_max * desc->channels, kBlock), kBlock, 0, stream>>>(
            *desc, n_ptr, n_max, x, ldx, pos, batch, rowptr, col, ws, cluster_scratch);
        DAGR_CHECK_LAUNCH();
    }
    DAGR_POOL_SCAN(false, false);
    DAGR_CHECK_LAUNCH();
    k_pool_emit<false><<<(unsigned)ceil_div(T, kBlock / 64), kBlock, 0, stream>>>(
        *desc, ws, batch, nullptr, x_out, ldo, xoff, pos_out, batch_out, rowptr_out, col_out, code_out, e_cap);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

int dagr_pool_status(const dagr_pool_desc *desc, void *pool_ws, int32_t *flags_host, void *stream) {
    int rc = validate_pool(desc);
    if (rc != DAGR_OK) return rc;
    DAGR_CHECK_ARG(pool_ws && flags_host, "NULL pointer");
    PoolWs ws;
    pool_carve(*desc, (char *)pool_ws, &ws);
    DAGR_CHECK_HIP(hipMemcpyAsync(flags_host, ws.status, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    DAGR_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return DAGR_OK;
}

const int32_t *dagr_pool_status_ptr(const dagr_pool_desc *desc, void *pool_ws) {
    if (validate_pool(desc) != DAGR_OK || !pool_ws) return nullptr;
    PoolWs ws;
    pool_carve(*desc, (char *)pool_ws, &ws);
    return ws.status;
}

int dagr_pool_counters(const dagr_pool_desc *desc, void *pool_ws, int32_t *out8_host, void *stream) {
    int rc = validate_pool(desc);
    if (rc != DAGR_OK) return rc;
    DAGR_CHECK_ARG(pool_ws && out8_host, "NULL pointer");
    PoolWs ws;
    pool_carve(*desc, (char *)pool_ws, &ws);
    DAGR_CHECK_HIP(hipMemcpyAsync(out8_host, ws.status, 32, hipMemcpyDeviceToHost, (hipStream_t)stream));
    DAGR_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return DAGR_OK;
}

int dagr_pool_recode(const int32_t *n_ptr, int32_t n_max, const int32_t *rowptr, const int32_t *col, const float *pos,
                     float two_max, float r00, float r02, float r11, float r12, int32_t rx, int32_t ry,
                     int32_t *code_out, int32_t e_cap, int32_t *status, void *stream) {
    DAGR_CHECK_ARG(n_max >= 0 && e_cap >= 0, "bad sizes");
    if (n_max == 0) return DAGR_OK;
    DAGR_CHECK_ARG(n_ptr && rowptr && col && pos && code_out && status, "NULL pointer");
    DAGR_CHECK_ARG(two_max > 0 && rx >= 0 && ry >= 0, "bad domain");
    k_recode<<<(unsigned)ceil_div(n_max, kBlock / 64), kBlock, 0, (hipStream_t)stream>>>(
        n_ptr, n_max, rowptr, col, pos, two_max, r00, r02, r11, r12, rx, ry, code_out, e_cap, status);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

}  // extern "C"
