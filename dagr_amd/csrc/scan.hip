// scan.hip -- int32 exclusive prefix sum (tile reduce -> scan of tile sums -> tile rescan).
// Used for the per-pixel event offsets (P = B*H*W bins), the neighbour-list row pointers and the
// occupied-voxel relabelling.  HBM traffic: 2 reads + 1 write of n ints (+ optional re-zero).
#include "common.hpp"

#include <stdlib.h>

namespace dagr {

size_t scan_scratch_elems(int64_t n) { return (size_t)ceil_div(n, kScanTile) + 8; }

namespace {

__global__ __launch_bounds__(kBlock) void scan_tile_reduce(const int32_t *__restrict__ in, int64_t n,
                                                          int32_t *__restrict__ tile_sums) {
    __shared__ int smem[4];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + threadIdx.x * 8;
    int s = 0;
    if (base + 8 <= n) {
        const int4 a = *reinterpret_cast<const int4 *>(in + base);
        const int4 b = *reinterpret_cast<const int4 *>(in + base + 4);
        s = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
    } else {
        for (int k = 0; k < 8; k++)
            if (base + k < n) s += in[base + k];
    }
    int total;
    block_exclusive_scan(s, smem, total);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// single block: in-place exclusive scan of tile_sums[0..m)
__global__ __launch_bounds__(kBlock) void scan_tile_sums(int32_t *__restrict__ tile_sums, int m) {
    __shared__ int smem[4];
    int carry = 0;
    for (int base = 0; base < m; base += kBlock) {
        int i = base + threadIdx.x;
        int v = i < m ? tile_sums[i] : 0;
        int total;
        int ex = block_exclusive_scan(v, smem, total);
        if (i < m) tile_sums[i] = carry + ex;
        carry += total;
    }
}

__global__ __launch_bounds__(kBlock) void scan_tile_apply(int32_t *__restrict__ in, int32_t *__restrict__ out,
                                                         int64_t n, const int32_t *__restrict__ tile_sums,
                                                         int zero_input) {
    __shared__ int smem[4];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + threadIdx.x * 8;
    int v[8];
    const bool full = base + 8 <= n;
    if (full) {
        const int4 a = *reinterpret_cast<const int4 *>(in + base);
        const int4 b = *reinterpret_cast<const int4 *>(in + base + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = (base + k < n) ? in[base + k] : 0;
    }
    int s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s += v[k];
    int total;
    int ex = block_exclusive_scan(s, smem, total) + tile_sums[blockIdx.x];
    int o[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { o[k] = ex; ex += v[k]; }
    if (full) {
        *reinterpret_cast<int4 *>(out + base) = make_int4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<int4 *>(out + base + 4) = make_int4(o[4], o[5], o[6], o[7]);
        if (zero_input) {
            *reinterpret_cast<int4 *>(in + base) = make_int4(0, 0, 0, 0);
            *reinterpret_cast<int4 *>(in + base + 4) = make_int4(0, 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (base + k < n) {
                out[base + k] = o[k];
                if (zero_input) in[base + k] = 0;
            }
    }
}

// whole scan in one workgroup (n <= 64 Ki): the voxel tables and cluster rows of the pooled levels
// are a few thousand entries, where three dependent launches cost more than the work
__global__ __launch_bounds__(1024) void scan_single_block(int32_t *__restrict__ in, int32_t *__restrict__ out,
                                                         int n, int zero_input) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024 * 4) {
        const int i0 = base + threadIdx.x * 4;
        int v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = (i0 + k < n) ? in[i0 + k] : 0;
        const int s = v[0] + v[1] + v[2] + v[3];
        const int incl = wave_inclusive_scan(s);
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        int wbase = 0, total = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int ws = wsum[k];
            if (k < wid) wbase += ws;
            total += ws;
        }
        int ex = carry_s + wbase + incl - s;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (i0 + k < n) {
                out[i0 + k] = ex;
                if (zero_input) in[i0 + k] = 0;
            }
            ex += v[k];
        }
        __syncthreads();
        if (threadIdx.x == 0) carry_s += total;
        __syncthreads();
    }
}

// The same scan in ONE launch (decoupled look-back, as k_pool_scan_chained in pooling.hip): a workgroup takes the next tile
// (ticket counter: a tile's predecessors are always running or done), scans it, publishes its total in a 64-bit word --
// launch tag (24 bits) | flag (1 = tile total, 2 = total of all tiles up to here) | value -- and one wave looks back over
// the predecessors' words until it meets an inclusive one.  The last ticket re-arms the counter and bumps the tag.  The
// per-pixel offsets of a window (P + 1 = 307 k entries per VGA sample) were three dependent launches; in a captured window
// every launch costs ~4.6 us before it does anything.
__global__ __launch_bounds__(kBlock) void scan_chained(int32_t *__restrict__ in, int32_t *__restrict__ out, int64_t n,
                                                      unsigned long long *__restrict__ state, int32_t *__restrict__ ctrl,
                                                      int zero_input) {
    __shared__ int smem[4];
    __shared__ int sh_tile, sh_base;
    __shared__ unsigned sh_tag;
    if (threadIdx.x == 0) {
        sh_tile = atomicAdd(&ctrl[0], 1);
        sh_tag = (unsigned)__hip_atomic_load(&ctrl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xffffffu;
    }
    __syncthreads();
    const int tile = sh_tile;
    const int ntiles = (int)((n + kScanTile - 1) / kScanTile);
    const unsigned long long tag = (unsigned long long)sh_tag << 40;
    const int64_t base = (int64_t)tile * kScanTile + threadIdx.x * 8;
    int v[8];
    const bool full = base + 8 <= n;
    if (full) {
        const int4 a = *reinterpret_cast<const int4 *>(in + base);
        const int4 b = *reinterpret_cast<const int4 *>(in + base + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = (base + k < n) ? in[base + k] : 0;
    }
    int s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s += v[k];
    int total;
    int ex = block_exclusive_scan(s, smem, total);
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        auto pack = [&](int flag, int val) {
            return tag | ((unsigned long long)flag << 32) | (unsigned long long)(unsigned)val;
        };
        if (lane == 0)
            __hip_atomic_store(&state[tile], pack(tile == 0 ? 2 : 1, total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int b_sum = 0;
        for (int hi = tile - 1; hi >= 0;) {        // window of predecessors hi, hi - 1, ..., hi - 63
            const int pidx = hi - lane;
            unsigned long long wd = 0;
            bool ready = pidx < 0;
            while (!__all(ready)) {
                if (!ready) {
                    wd = __hip_atomic_load(&state[pidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ready = (wd >> 40) == (unsigned long long)sh_tag && ((wd >> 32) & 3ull) != 0ull;
                }
            }
            const bool incl = pidx >= 0 && ((wd >> 32) & 3ull) == 2ull;
            const unsigned long long im = __ballot(incl);
            const int stop = im ? (__ffsll((long long)im) - 1) : 63;     // nearest predecessor with an inclusive total
            int val = (pidx >= 0 && lane <= stop) ? (int)(unsigned)(wd & 0xffffffffull) : 0;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) val += __shfl_xor(val, off, 64);
            b_sum += val;
            if (im) break;
            hi -= 64;
        }
        if (lane == 0) {
            sh_base = b_sum;
            if (tile > 0)
                __hip_atomic_store(&state[tile], pack(2, b_sum + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    ex += sh_base;
    int o[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { o[k] = ex; ex += v[k]; }
    if (full) {
        *reinterpret_cast<int4 *>(out + base) = make_int4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<int4 *>(out + base + 4) = make_int4(o[4], o[5], o[6], o[7]);
        if (zero_input) {
            *reinterpret_cast<int4 *>(in + base) = make_int4(0, 0, 0, 0);
            *reinterpret_cast<int4 *>(in + base + 4) = make_int4(0, 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (base + k < n) {
                out[base + k] = o[k];
                if (zero_input) in[base + k] = 0;
            }
    }
    if (tile == ntiles - 1 && threadIdx.x == 0) {     // every ticket has been handed out: re-arm for the next launch
        ctrl[0] = 0;
        ctrl[1] = (int)((sh_tag + 1u) & 0xffffffu);
    }
}

// The one-launch scan on WIDE tiles (THREADS x EPT elements): the window builder's key table (B * H * W keys:
// 2.5 M entries for a B = 8 batch of VGA samples) stays within a few hundred tiles, where the look-back of one wave per
// tile is short, and every element is read ONCE (the three-launch form reads the input twice).  A wave owns a contiguous
// sub-tile of 64 * EPT elements and reads it as EPT / 4 fully coalesced 1-KiB pieces; per piece a wave scan, a running
// carry across the pieces, one LDS word per wave for the tile's total.
template <int THREADS, int EPT>
__global__ __launch_bounds__(THREADS) void scan_chained_wide(int32_t *__restrict__ in, int32_t *__restrict__ out, int64_t n,
                                                            unsigned long long *__restrict__ state,
                                                            int32_t *__restrict__ ctrl, int zero_input) {
    constexpr int NW = THREADS / 64, CH = EPT / 4, TILE = THREADS * EPT;
    __shared__ int wsum[NW];
    __shared__ int sh_tile, sh_base;
    __shared__ unsigned sh_tag;
    if (threadIdx.x == 0) {
        sh_tile = atomicAdd(&ctrl[0], 1);
        sh_tag = (unsigned)__hip_atomic_load(&ctrl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xffffffu;
    }
    __syncthreads();
    const int tile = sh_tile;
    const int ntiles = (int)((n + TILE - 1) / TILE);
    const unsigned long long tag = (unsigned long long)sh_tag << 40;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t wbase = (int64_t)tile * TILE + (int64_t)wid * (64 * EPT) + lane * 4;
    // (one base pointer per array + compile-time piece offsets: 64-bit indices per piece cost the registers of a piece)
    const int32_t *in_w = in + wbase;
    const int rem = (int)min((int64_t)(64 * EPT + 4), max((int64_t)0, n - wbase));      // elements of the array from wbase on
    int4 v[CH];
#pragma unroll
    for (int k = 0; k < CH; k++) {
        if (k * 256 + 4 <= rem) v[k] = *reinterpret_cast<const int4 *>(in_w + k * 256);
        else {
            v[k].x = k * 256 < rem ? in_w[k * 256] : 0;
            v[k].y = k * 256 + 1 < rem ? in_w[k * 256 + 1] : 0;
            v[k].z = k * 256 + 2 < rem ? in_w[k * 256 + 2] : 0;
            v[k].w = 0;
        }
    }
    int ex[CH];          // exclusive prefix, inside the wave's sub-tile, of this lane's four elements of piece k
    int carry = 0;
#pragma unroll
    for (int k = 0; k < CH; k++) {
        const int sum4 = v[k].x + v[k].y + v[k].z + v[k].w;
        const int incl = wave_inclusive_scan(sum4);
        ex[k] = carry + incl - sum4;
        carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) wsum[wid] = carry;
    __syncthreads();
    int wprefix = 0, total = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        const int t = wsum[w];
        if (w < wid) wprefix += t;
        total += t;
    }
    if (threadIdx.x < 64) {
        auto pack = [&](int flag, int val) {
            return tag | ((unsigned long long)flag << 32) | (unsigned long long)(unsigned)val;
        };
        if (lane == 0)
            __hip_atomic_store(&state[tile], pack(tile == 0 ? 2 : 1, total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int b_sum = 0;
        for (int hi = tile - 1; hi >= 0;) {        // window of predecessors hi, hi - 1, ..., hi - 63
            const int pidx = hi - lane;
            unsigned long long wd = 0;
            bool ready = pidx < 0;
            while (!__all(ready)) {
                if (!ready) {
                    wd = __hip_atomic_load(&state[pidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ready = (wd >> 40) == (unsigned long long)sh_tag && ((wd >> 32) & 3ull) != 0ull;
                }
            }
            const bool incl = pidx >= 0 && ((wd >> 32) & 3ull) == 2ull;
            const unsigned long long im = __ballot(incl);
            const int stop = im ? (__ffsll((long long)im) - 1) : 63;     // nearest predecessor with an inclusive total
            int val = (pidx >= 0 && lane <= stop) ? (int)(unsigned)(wd & 0xffffffffull) : 0;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) val += __shfl_xor(val, off, 64);
            b_sum += val;
            if (im) break;
            hi -= 64;
        }
        if (lane == 0) {
            sh_base = b_sum;
            if (tile > 0)
                __hip_atomic_store(&state[tile], pack(2, b_sum + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    const int base = sh_base + wprefix;
    int32_t *out_w = out + wbase;
    int32_t *zin_w = in + wbase;
#pragma unroll
    for (int k = 0; k < CH; k++) {
        const int o0 = base + ex[k], o1 = o0 + v[k].x, o2 = o1 + v[k].y, o3 = o2 + v[k].z;
        if (k * 256 + 4 <= rem) {
            *reinterpret_cast<int4 *>(out_w + k * 256) = make_int4(o0, o1, o2, o3);
            if (zero_input) *reinterpret_cast<int4 *>(zin_w + k * 256) = make_int4(0, 0, 0, 0);
        } else {
            if (k * 256 < rem) { out_w[k * 256] = o0; if (zero_input) zin_w[k * 256] = 0; }
            if (k * 256 + 1 < rem) { out_w[k * 256 + 1] = o1; if (zero_input) zin_w[k * 256 + 1] = 0; }
            if (k * 256 + 2 < rem) { out_w[k * 256 + 2] = o2; if (zero_input) zin_w[k * 256 + 2] = 0; }
        }
    }
    if (tile == ntiles - 1 && threadIdx.x == 0) {     // every ticket has been handed out: re-arm for the next launch
        ctrl[0] = 0;
        ctrl[1] = (int)((sh_tag + 1u) & 0xffffffu);
    }
}

}  // namespace

size_t scan_chained_state_bytes(int64_t n) { return ((size_t)ceil_div(n, kScanTile) + 8) * 8 + 64; }

hipError_t exclusive_scan_i32_chained(int32_t *in, int32_t *out, int64_t n, void *state, bool zero_input,
                                      hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    // the look-back advances 64 tiles per round trip: beyond a few hundred tiles (a B = 8 batch of VGA samples is 1 200)
    // the three-launch form is faster (measured 36 vs ~25 us), below it the single launch wins (8.5 vs 14 us at 150 tiles)
    // state: [ticket, tag, pad...] (64 bytes) then one 64-bit word per tile; all zero before the first launch
    int32_t *ctrl = (int32_t *)state;
    unsigned long long *words = (unsigned long long *)((char *)state + 64);
    if (ceil_div(n, kScanTile) > 320) {
        // wider tiles keep the tile count (= the look-back's length) down: the narrowest tile that gives <= 320 of them
        static const int force = (int)knob("DAGR_SCAN_EPT", 0);   // measurement knob
        int ept = n <= 320ll * 1024 * 8 ? 8 : n <= 320ll * 1024 * 16 ? 16 : n <= 320ll * 1024 * 32 ? 32 : 48;
        if (force == 8 || force == 16 || force == 32 || force == 48) ept = force;
        if (force < 0 || ceil_div(n, 1024 * 48) > 2048)     // (far beyond any window: the three-launch form)
            return exclusive_scan_i32(in, out, n, (int32_t *)((char *)state + 64), zero_input, stream);
        const unsigned tiles = (unsigned)ceil_div(n, 1024ll * ept);
        const int z = zero_input ? 1 : 0;
        if (ept == 8) scan_chained_wide<1024, 8><<<tiles, 1024, 0, stream>>>(in, out, n, words, ctrl, z);
        else if (ept == 16) scan_chained_wide<1024, 16><<<tiles, 1024, 0, stream>>>(in, out, n, words, ctrl, z);
        else if (ept == 32) scan_chained_wide<1024, 32><<<tiles, 1024, 0, stream>>>(in, out, n, words, ctrl, z);
        else scan_chained_wide<1024, 48><<<tiles, 1024, 0, stream>>>(in, out, n, words, ctrl, z);
        return hipGetLastError();
    }
    scan_chained<<<(unsigned)ceil_div(n, kScanTile), kBlock, 0, stream>>>(in, out, n, words, ctrl, zero_input ? 1 : 0);
    return hipGetLastError();
}

hipError_t exclusive_scan_i32(int32_t *in, int32_t *out, int64_t n, int32_t *scratch, bool zero_input,
                              hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    if (n <= 65536) {
        scan_single_block<<<1, 1024, 0, stream>>>(in, out, (int)n, zero_input ? 1 : 0);
        return hipGetLastError();
    }
    const int tiles = (int)ceil_div(n, kScanTile);
    scan_tile_reduce<<<tiles, kBlock, 0, stream>>>(in, n, scratch);
    scan_tile_sums<<<1, kBlock, 0, stream>>>(scratch, tiles);
    scan_tile_apply<<<tiles, kBlock, 0, stream>>>(in, out, n, scratch, zero_input ? 1 : 0);
    return hipGetLastError();
}

}  // namespace dagr
