// conv_pooled_tiles.hip -- pooled-level SplineConv (CSR graphs of <= 2240*(B+1) nodes) as 16-node wave tiles, gfx950.
//
// Reference op: MySplineConv.forward/_forward/message_lut (src/dagr/model/layers/spline_conv.py:39-78) + BatchNorm(eval)
// + ReLU (model/layers/conv.py:23-28) + the skip Linear + BN of ConvBlockWithSkip (conv.py:47-56), as one contraction
//     out[n] = [ A[n][tap][ch] (25*cin) | x[n] (cin) | xskip[n] (cskip) ] . Wm + bias,   A = sum_edges basis * x[src].
//
// The pooled levels are small (a few hundred to ~18 k nodes) and K = 26*cin + cskip is long (up to 3.5 k), so a conv is
// bound by latency chains, not by bytes or flops.  k_conv_fused (gemm.hip) spends them serially in one 1024-thread
// workgroup per 16 nodes: zero a [16][K] LDS tile, walk the edges with LDS read-modify-writes, barrier, contract with
// weights streamed from L2 (26 us per launch, 16 launches per window).  Here the layout of conv_l0_tiles.hip is used:
//   * workgroup = (16-node tile, 16-column tile[s]), 4 waves; lane (c = l & 15, q = l >> 4) works for NODE c on a channel
//     quad; wave w takes the 16-channel groups w, w+4, ...: it walks its nodes' CSR rows itself and keeps
//     A[25 taps][4 channels] in registers -- which already are the A operands of v_mfma_f32_16x16x4_f32;
//   * the weights are packed on the host in exactly the order the waves consume them ([column tile][k-step][lane]), one
//     coalesced 256-byte wave load per MFMA, no LDS staging, no zeroing, no barrier before the contraction;
//   * the four waves' partial tiles (a 4-way split of K) are summed through 4 KiB of LDS in a fixed order.
// One workgroup per (tile, column tile), so that >= 256 workgroups exist even for ~1 k nodes.
// Exact fp32; results are run-to-run identical (fixed summation order).
//
// STATUS (round 2, measured on MI355X): correct (tests/test_spline_fused_gpu.py) and independent of K -- it also covers
// the convs whose aggregated row does not fit k_conv_fused's LDS tile -- but NOT faster: with ~230 live registers a
// workgroup is one wave per SIMD, every column tile repeats the edge walk, and the per-launch time came out at ~33 us
// against ~26 us for k_conv_fused (tail of a B = 8 window 0.53 vs 0.42 ms; B = 1: 0.26 vs 0.22 ms).  The engine keeps
// k_conv_fused (DAGR_POOLED_TILES=1 switches this kernel in for A/B runs).
#include "common.hpp"

namespace dagr {
namespace {

using f32x4_t = __attribute__((ext_vector_type(4))) float;

struct PooledShape {   // how K is cut into per-wave units; shared by the host packer (dagr_spline_conv_tiles_pack)
    int cin, cskip, G, CE, SG, SE, n_units, n_steps;
    __host__ __device__ PooledShape(int cin_, int cskip_) : cin(cin_), cskip(cskip_) {
        G = cin / 16; CE = cin % 16; SG = cskip / 16; SE = cskip % 16;
        n_units = G + (CE ? 1 : 0) + SG + (SE ? 1 : 0);
        n_steps = G * 104 + (CE ? 28 : 0) + SG * 4 + (SE ? 4 : 0);     // every unit a multiple of 4 k-steps (zero padded)
    }
    // first k-step of unit u and its kind: 0 main group, 1 extras, 2 skip group, 3 skip extras
    __host__ __device__ void unit(int u, int &kind, int &idx, int &step0) const {
        if (u < G) { kind = 0; idx = u; step0 = u * 104; return; }
        u -= G;
        int base = G * 104;
        if (CE) { if (u == 0) { kind = 1; idx = 0; step0 = base; return; } u--; base += 28; }
        if (u < SG) { kind = 2; idx = u; step0 = base + 4 * u; return; }
        kind = 3; idx = 0; step0 = base + 4 * SG;
    }
    // row of the [K, N] weight matrix behind k-step s for lane quad q, or -1 (zero operand)
    __host__ __device__ int row(int s, int q) const {
        if (s < G * 104) {
            const int cg = s / 104, r = s % 104;
            if (r < 100) return (r >> 2) * cin + 16 * cg + 4 * q + (r & 3);      // tap r/4, channel 16cg + 4q + r%4
            return 25 * cin + 16 * cg + 4 * q + (r - 100);                      // root
        }
        s -= G * 104;
        if (CE) {
            if (s < 25) return q < CE ? s * cin + 16 * G + q : -1;
            if (s == 25) return q < CE ? 25 * cin + 16 * G + q : -1;
            if (s < 28) return -1;
            s -= 28;
        }
        if (s < 4 * SG) return 26 * cin + 16 * (s >> 2) + 4 * q + (s & 3);
        s -= 4 * SG;
        return (s == 0 && q < SE) ? 26 * cin + 16 * SG + q : -1;
    }
};

template <int NCT>   // column tiles (16 outputs each) per workgroup
__global__ __launch_bounds__(kBlock, 1) void k_conv_pooled_tiles(
    const int32_t *__restrict__ n_ptr, int n_max, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const int32_t *__restrict__ code, const float *__restrict__ x, int ldx, int cin, const float *__restrict__ xskip,
    int ldskip, int cskip, int rx, int ry, float den_x, float den_y, const float *__restrict__ Wt,
    const float *__restrict__ bias, float *__restrict__ C, int ldc, int N, int relu) {
    extern __shared__ __align__(16) float lds[];
    float *ax_l = lds;                              // [2rx+1][8]  per-axis basis weights of the 5 taps
    float *ay_l = ax_l + (2 * rx + 1) * 8;          // [2ry+1][8]
    float *red = ay_l + (2 * ry + 1) * 8;           // [4 waves][NCT][16][16]
    const PooledShape S(cin, cskip);
    for (int i = threadIdx.x; i < (2 * rx + 1) * 8; i += kBlock) {
        const Axis a = spline_axis(i >> 3, rx, den_x);
        const int t = i & 7;
        ax_l[i] = (t == a.k0 ? a.b0 : 0.0f) + (t == a.k1 ? a.b1 : 0.0f);
    }
    for (int i = threadIdx.x; i < (2 * ry + 1) * 8; i += kBlock) {
        const Axis a = spline_axis(i >> 3, ry, den_y);
        const int t = i & 7;
        ay_l[i] = (t == a.k0 ? a.b0 : 0.0f) + (t == a.k1 ? a.b1 : 0.0f);
    }
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = l & 15, q = l >> 4;
    const int m0 = blockIdx.x * 16;
    const int n = m0 + c;
    // rowptr has n_max + 1 entries: read it before the device-side bound is known (one latency, not two)
    const int nn = min(n, n_max - 1);
    int e0 = rowptr[nn], e1 = rowptr[nn + 1];
    const int M = n_ptr ? min(*n_ptr, n_max) : n_max;
    if (m0 >= M) return;                                                   // uniform over the workgroup
    const bool valid = n < M;
    if (!valid) e1 = e0;
    const int d = e1 - e0;
    int dmax = d;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) dmax = max(dmax, __shfl_xor(dmax, off, 16));
    __syncthreads();                                                       // axis tables

    const int ct0 = blockIdx.y * NCT;
    const int n_ct = (N + 15) / 16;
    const int n_q4 = S.n_steps >> 2;                 // float4 groups of 4 k-steps per column tile
    f32x4_t acc[NCT][2];
#pragma unroll
    for (int t = 0; t < NCT; t++) acc[t][0] = acc[t][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // B operands: Wt as float4[(ct * n_q4 + step / 4) * 64 + lane] = the weights of 4 consecutive k-steps for this lane.
    // A conv of these small levels is a chain of memory latencies, so they are requested in batches of kWB float4
    // (4 kWB k-steps) per column tile, the first batch BEFORE the edge walk of the unit, and consumed by MFMAs on two
    // accumulator chains (40-cycle dependent latency, 32 to issue).
    constexpr int kWB = 8;
    const float4 *wq = reinterpret_cast<const float4 *>(Wt) + l;
    auto load_w = [&](float4(&dst)[NCT][kWB], int q4_0, int n_q) {
#pragma unroll
        for (int t = 0; t < NCT; t++)
#pragma unroll
            for (int k = 0; k < kWB; k++)
                dst[t][k] = (k < n_q && ct0 + t < n_ct) ? wq[((size_t)(ct0 + t) * n_q4 + q4_0 + k) * 64]
                                                         : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto mac4 = [&](const float4(&w)[NCT][kWB], int k, float a0, float a1, float a2, float a3) {
#pragma unroll
        for (int t = 0; t < NCT; t++) {
            acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, w[t][k].x, acc[t][0], 0, 0, 0);
            acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, w[t][k].y, acc[t][1], 0, 0, 0);
            acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, w[t][k].z, acc[t][0], 0, 0, 0);
            acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, w[t][k].w, acc[t][1], 0, 0, 0);
        }
    };

    for (int u = wv; u < S.n_units; u += 4) {                              // wave-uniform
        int kind, idx, step0;
        S.unit(u, kind, idx, step0);
        const int q4 = step0 >> 2;
        float4 wa[NCT][kWB], wb[NCT][kWB];
        if (kind >= 2) {   // skip-input rows: straight from memory, one float4 group of weights
            load_w(wa, q4, 1);
            if (kind == 2) {
                const float4 xs = valid ? *reinterpret_cast<const float4 *>(xskip + (size_t)n * ldskip + 16 * idx + 4 * q)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
                mac4(wa, 0, xs.x, xs.y, xs.z, xs.w);
            } else {
                const float xe = (valid && q < S.SE) ? xskip[(size_t)n * ldskip + 16 * S.SG + q] : 0.f;
                mac4(wa, 0, xe, 0.f, 0.f, 0.f);
            }
            continue;
        }
        const bool main_grp = kind == 0;
        load_w(wa, q4, kWB);                                               // lands while the edges are walked
        // ---- phase 1: this lane's node, its in-edges, channels [ch0, ch0+4) (main) / ch0 + q (extras)
        const int ch0 = main_grp ? 16 * idx + 4 * q : 16 * S.G + q;
        const bool ex_ok = !main_grp && q < S.CE;
        float a4[25][4];
#pragma unroll
        for (int t = 0; t < 25; t++) a4[t][0] = a4[t][1] = a4[t][2] = a4[t][3] = 0.f;
        constexpr int EC = 8;                                              // edges per chunk: all their loads in flight
#pragma unroll 1
        for (int j0 = 0; j0 < dmax; j0 += EC) {
            int srcs[EC], cds[EC];
            float4 xv[EC];
#pragma unroll
            for (int uu = 0; uu < EC; uu++) {
                const bool ok = j0 + uu < d;
                srcs[uu] = ok ? col[e0 + j0 + uu] : nn;     // predicated-off lanes re-read their own (existing) row
                cds[uu] = ok ? code[e0 + j0 + uu] : 0;
            }
#pragma unroll
            for (int uu = 0; uu < EC; uu++) {
                if (main_grp) xv[uu] = *reinterpret_cast<const float4 *>(x + (size_t)srcs[uu] * ldx + ch0);
                else xv[uu] = make_float4(ex_ok ? x[(size_t)srcs[uu] * ldx + ch0] : 0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int uu = 0; uu < EC; uu++) {
                if (j0 + uu < dmax) {                                      // group-uniform: skip the chunk's empty tail
                    const bool ok = j0 + uu < d;
                    const int ix = cds[uu] & 0xffff, iy = (cds[uu] >> 16) & 0xffff;
                    const float4 wx0 = *reinterpret_cast<const float4 *>(ax_l + 8 * ix);
                    const float wx4 = ax_l[8 * ix + 4];
                    const float4 wy0 = *reinterpret_cast<const float4 *>(ay_l + 8 * iy);
                    const float wy4 = ay_l[8 * iy + 4];
                    const float wx[5] = {wx0.x, wx0.y, wx0.z, wx0.w, wx4};
                    float wy[5] = {wy0.x, wy0.y, wy0.z, wy0.w, wy4};
#pragma unroll
                    for (int b = 0; b < 5; b++) wy[b] = ok ? wy[b] : 0.f;
#pragma unroll
                    for (int b = 0; b < 5; b++)
#pragma unroll
                        for (int a = 0; a < 5; a++) {
                            const float w = wx[a] * wy[b];                  // basis product of tap a + 5 b
                            const int t = a + 5 * b;
                            a4[t][0] = fmaf(w, xv[uu].x, a4[t][0]);
                            if (main_grp) {
                                a4[t][1] = fmaf(w, xv[uu].y, a4[t][1]);
                                a4[t][2] = fmaf(w, xv[uu].z, a4[t][2]);
                                a4[t][3] = fmaf(w, xv[uu].w, a4[t][3]);
                            }
                        }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- phase 2 for this unit
        if (main_grp) {
            const float4 xr = valid ? *reinterpret_cast<const float4 *>(x + (size_t)n * ldx + ch0)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
            // 26 float4 groups: taps 0..24, then the root row; batches of 8 / 8 / 8 / 2
            load_w(wb, q4 + 8, kWB);
#pragma unroll
            for (int k = 0; k < 8; k++) mac4(wa, k, a4[k][0], a4[k][1], a4[k][2], a4[k][3]);
            load_w(wa, q4 + 16, kWB);
#pragma unroll
            for (int k = 0; k < 8; k++) mac4(wb, k, a4[8 + k][0], a4[8 + k][1], a4[8 + k][2], a4[8 + k][3]);
            load_w(wb, q4 + 24, 2);
#pragma unroll
            for (int k = 0; k < 8; k++) mac4(wa, k, a4[16 + k][0], a4[16 + k][1], a4[16 + k][2], a4[16 + k][3]);
            mac4(wb, 0, a4[24][0], a4[24][1], a4[24][2], a4[24][3]);
            mac4(wb, 1, xr.x, xr.y, xr.z, xr.w);
        } else {
            // 28 k-steps = 7 float4 groups: taps 0..24 of the extra channel, its root row, two zero steps
            const float xr = (valid && ex_ok) ? x[(size_t)n * ldx + ch0] : 0.f;
#pragma unroll
            for (int k = 0; k < 6; k++) mac4(wa, k, a4[4 * k][0], a4[4 * k + 1][0], a4[4 * k + 2][0], a4[4 * k + 3][0]);
            mac4(wa, 6, a4[24][0], xr, 0.f, 0.f);
        }
    }
    // ---- 4-way split-K reduction through LDS (fixed order), bias, activation
#pragma unroll
    for (int t = 0; t < NCT; t++)
#pragma unroll
        for (int r = 0; r < 4; r++)
            red[((wv * NCT + t) * 16 + 4 * q + r) * 16 + c] = acc[t][0][r] + acc[t][1][r];   // [node 4q + r][column c]
    __syncthreads();
    for (int i = threadIdx.x; i < NCT * 256; i += kBlock) {
        const int t = i >> 8, row = (i >> 4) & 15, cc = i & 15;
        const int orow = m0 + row, ocol = (ct0 + t) * 16 + cc;
        if (orow < M && ocol < N) {
            float v = red[((0 * NCT + t) * 16 + row) * 16 + cc];
            v += red[((1 * NCT + t) * 16 + row) * 16 + cc];
            v += red[((2 * NCT + t) * 16 + row) * 16 + cc];
            v += red[((3 * NCT + t) * 16 + row) * 16 + cc];
            v += bias ? bias[ocol] : 0.f;
            if (relu) v = fmaxf(v, 0.f);
            C[(size_t)orow * ldc + ocol] = v;
        }
    }
}

}  // namespace
}  // namespace dagr

using namespace dagr;

extern "C" {

// number of floats of the packed weight image for (cin, cskip, N)
size_t dagr_spline_conv_tiles_pack_elems(int32_t cin, int32_t cskip, int32_t N) {
    if (cin < 1 || cskip < 0 || N < 1) return 0;
    const PooledShape S(cin, cskip);
    return (size_t)((N + 15) / 16) * S.n_steps * 64;
}

// host helper: Wt[((ct * n_steps / 4 + s / 4) * 64 + lane) * 4 + s % 4] = Wm[row(s, lane >> 4)][16 ct + (lane & 15)]
// (0 outside [K, N]): per lane a float4 of the weights of 4 consecutive k-steps;
// Wm is the [K = 26 cin + cskip, N] matrix of dagr_gemm_bias_act, row stride ldw.  Both HOST arrays.
int dagr_spline_conv_tiles_pack(const float *Wm_host, int32_t ldw, int32_t cin, int32_t cskip, int32_t N,
                                float *Wt_host) {
    DAGR_CHECK_ARG(Wm_host && Wt_host && cin >= 1 && cskip >= 0 && N >= 1 && ldw >= N, "bad arguments");
    const PooledShape S(cin, cskip);
    const int n_ct = (N + 15) / 16;
    for (int ct = 0; ct < n_ct; ct++)
        for (int s = 0; s < S.n_steps; s++)
            for (int lane = 0; lane < 64; lane++) {
                const int r = S.row(s, lane >> 4), cc = 16 * ct + (lane & 15);
                Wt_host[(((size_t)ct * (S.n_steps >> 2) + (s >> 2)) * 64 + lane) * 4 + (s & 3)] =
                    (r >= 0 && cc < N) ? Wm_host[(size_t)r * ldw + cc] : 0.0f;
            }
    return DAGR_OK;
}

int dagr_spline_conv_tiles(const int32_t *n_nodes_ptr, int32_t n_nodes_max, const int32_t *rowptr, const int32_t *col,
                           const int32_t *code, const float *x, int32_t ldx, int32_t cin, const float *xskip,
                           int32_t ldskip, int32_t cskip, int32_t rx, int32_t ry, float den_x, float den_y,
                           const float *Wt, const float *bias, float *C, int32_t ldc, int32_t N, int32_t relu,
                           void *stream) {
    DAGR_CHECK_ARG(n_nodes_max >= 0 && cin >= 1 && N >= 1, "bad sizes");
    if (n_nodes_max == 0) return DAGR_OK;
    DAGR_CHECK_ARG(rowptr && col && code && x && Wt && C, "NULL pointer");
    DAGR_CHECK_ARG(((uintptr_t)Wt % 16) == 0, "packed weights must be 16-byte aligned");
    DAGR_CHECK_ARG(cskip == 0 || xskip, "xskip is NULL");
    DAGR_CHECK_ARG(cin % 16 <= 4 && cskip % 16 <= 4, "channel counts must be 16 k + (0..4)");
    DAGR_CHECK_ARG(cin < 16 || (ldx % 4 == 0 && ((uintptr_t)x % 16) == 0), "x rows must be 16-byte aligned");
    DAGR_CHECK_ARG(cskip < 16 || (ldskip % 4 == 0 && ((uintptr_t)xskip % 16) == 0), "xskip rows must be 16-byte aligned");
    DAGR_CHECK_ARG(rx >= 0 && ry >= 0 && rx < 32768 && ry < 32768 && den_x > 0 && den_y > 0, "bad offset domain");
    const int tiles = (int)ceil_div(n_nodes_max, 16);
    const int n_ct = (N + 15) / 16;
    // one workgroup per (tile, column tile): meant for levels of up to a few thousand nodes, where that many workgroups
    // are what covers the chip (a workgroup keeps ~230 registers per lane: one wave per SIMD).  Larger levels are better
    // served by dagr_spline_conv_fused, which walks the edges once per tile for all columns; the caller chooses.
    const size_t lds_bytes = ((size_t)(2 * rx + 1) + (size_t)(2 * ry + 1)) * 8 * 4 + (size_t)4 * 256 * 4;
    DAGR_CHECK_ARG(lds_bytes <= 64 * 1024, "offset domain too large for the axis tables");
    static thread_local size_t set_for = 0;
    if (set_for < lds_bytes) {
        DAGR_CHECK_HIP(hipFuncSetAttribute((const void *)k_conv_pooled_tiles<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes));
        set_for = lds_bytes;
    }
    const dim3 grid((unsigned)tiles, (unsigned)n_ct);
    k_conv_pooled_tiles<1><<<grid, kBlock, lds_bytes, (hipStream_t)stream>>>(
        n_nodes_ptr, n_nodes_max, rowptr, col, code, x, ldx, cin, xskip, ldskip, cskip, rx, ry, den_x, den_y, Wt, bias, C,
        ldc, N, relu);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

}  // extern "C"
