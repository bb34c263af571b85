// spline_conv.hip -- SplineConv message passing for gfx950 (fp32).
//
// Reference op (src/dagr/model/layers/spline_conv.py:39-78 over PyG SplineConv / torch_spline_conv /
// torch_scatter):   out[n] = sum_{(j->n)} x_j . What(dx,dy)  +  x_n . Wroot^T (+ bias)
// with What(dx,dy) = sum_{s<4} basis_s(u) W[tap_s(u)], u = integer pixel offset mapped to [0,1]
// (degree-1 open B-spline over a 5x5 kernel).  The reference materialises What for every integer
// offset (init_lut, GBs) and per edge gathers a Cin x Cout slab of it.
//
// Here the sum is re-associated per destination node:
//     A[n][k][:] = sum_j basis_{j,k} x_j           (k = kernel tap, "tap aggregation": gather-bound)
//     out[n]     = sum_k A[n][k] . W[k] + x_n . Wroot^T (+ skip input . Wskip^T) + shift ; ReLU
// so the weights are touched once per node (from LDS / registers), never per edge, and no LUT
// exists.  BatchNorm(eval) is folded into the weights/shift on the host; the ConvBlockWithSkip
// branch (conv.py:47-56) is two extra weight blocks of the same contraction.
//
//   * level 0 (event graph, fixed-stride neighbour lists, Cout=16): k_conv_l0 -- fully fused, one
//     16-lane group per node (lane = input channel), 3x3 active taps, weights in LDS.
//   * pooled levels (<= 2240 cells/sample, CSR): k_tap_aggregate writes A[n] = [25*Cin | Cin | Cskip]
//     and k_gemm_bias_act contracts it with the packed weight matrix.
#include "common.hpp"

#include <cstdlib>

namespace dagr {
namespace {

// ---------------------------------------------------------------------------------------------
// Generic path, step 1: tap aggregation over a CSR-by-destination graph.
// One wave per destination node; LDS accumulators [25][Cin] per wave; lanes stride the channels.
// Row layout of A: [25*Cin taps | Cin root copy | Cskip skip-input copy], row stride lda.
constexpr int kAggWaves = 4;
// LPN = lanes per node: 64 (one wave per node, the pooled levels: 18 ... 258 input channels) or 16 (four nodes per wave,
// rows of <= 16 channels: the event level of the training path, where 48 of 64 lanes used to idle).  A node's edges are
// walked in the same order by either form: same sums, bit for bit.
template <int LPN>
__global__ __launch_bounds__(kBlock) void k_tap_aggregate(
    const int32_t *__restrict__ n_nodes_ptr, int n_nodes_max, const int32_t *__restrict__ rowptr,
    const int32_t *__restrict__ col, const int32_t *__restrict__ code, const float *__restrict__ x, int ldx,
    int cin, const float *__restrict__ xskip, int ldskip, int cskip, int rx, int ry, float den_x, float den_y,
    float *__restrict__ A, int lda) {
    extern __shared__ float lds[];
    constexpr int NPW = 64 / LPN;                    // nodes per wave
    const int lane = threadIdx.x & (LPN - 1);
    const int slot = threadIdx.x / LPN;              // node slot inside the workgroup
    const int n = blockIdx.x * (kAggWaves * NPW) + slot;
    const int n_nodes = n_nodes_ptr ? min(*n_nodes_ptr, n_nodes_max) : n_nodes_max;
    if (n >= n_nodes) return;
    float *acc = lds + (size_t)slot * 25 * cin;
    for (int i = lane; i < 25 * cin; i += LPN) acc[i] = 0.0f;
    const int e0 = rowptr[n], e1 = rowptr[n + 1];
    for (int e = e0; e < e1; e++) {
        const int src = col[e];
        const int c = code[e];
        const Axis ax = spline_axis(c & 0xffff, rx, den_x);
        const Axis ay = spline_axis(c >> 16, ry, den_y);
        // basis[s] = (1 * fx) * fy ; tap = kx + 5*ky  (s = ax + 2*ay)
        const float b00 = ax.b0 * ay.b0, b10 = ax.b1 * ay.b0, b01 = ax.b0 * ay.b1, b11 = ax.b1 * ay.b1;
        float *a00 = acc + (ax.k0 + 5 * ay.k0) * cin;
        float *a10 = acc + (ax.k1 + 5 * ay.k0) * cin;
        float *a01 = acc + (ax.k0 + 5 * ay.k1) * cin;
        float *a11 = acc + (ax.k1 + 5 * ay.k1) * cin;
        const float *xs = x + (size_t)src * ldx;
        for (int i = lane; i < cin; i += LPN) {
            const float v = xs[i];
            a00[i] += b00 * v;
            a10[i] += b10 * v;
            a01[i] += b01 * v;
            a11[i] += b11 * v;
        }
    }
    float *row = A + (size_t)n * lda;
    for (int i = lane; i < 25 * cin; i += LPN) row[i] = acc[i];
    const float *xn = x + (size_t)n * ldx;
    for (int i = lane; i < cin; i += LPN) row[25 * cin + i] = xn[i];
    if (cskip > 0) {
        const float *sn = xskip + (size_t)n * ldskip;
        for (int i = lane; i < cskip; i += LPN) row[26 * cin + i] = sn[i];
    }
}


// ---------------------------------------------------------------------------------------------
// Backward of the tap aggregation (training path): A[n] = [sum_edges basis * x[src] per tap | x[n] | ..] is linear in x,
// so with gA = dL/dA (from the GEMM's backward, a plain library GEMM) the input gradient is its transpose:
// gx[src] += sum_{taps of the edge} basis * gA[dst][tap], plus the root copy gx[n] += gA[n][25 cin + .].
// One wave per destination node, lanes stride the channels.  The scatter is DETERMINISTIC: contributions are added as
// 64-bit fixed-point integers (integer addition is associative, so the order in which the atomics land does not
// matter), scaled by 2^50 / max|gA| -- every contribution is a convex combination of gA entries, so |term| <= max|gA|,
// a node has far fewer than 2^12 out-edges, and the resolution max|gA| * 2^-50 is 2^-26 of an fp32 ulp at the
// tensor's own scale.  k_fixed_to_float turns the sums into fp32.  acc must be zero-initialised by the caller.
constexpr double kFixedOne = 1125899906842624.0;     // 2^50

template <int LPN>      // lanes per node, as k_tap_aggregate
__global__ __launch_bounds__(kBlock) void k_tap_scatter_grad(const int32_t *__restrict__ n_nodes_ptr, int n_nodes_max,
                                                            const int32_t *__restrict__ rowptr,
                                                            const int32_t *__restrict__ col,
                                                            const int32_t *__restrict__ code,
                                                            const float *__restrict__ gA, int lda, int cin, int rx,
                                                            int ry, float den_x, float den_y,
                                                            const float *__restrict__ amax,
                                                            long long *__restrict__ acc) {
    constexpr int NPW = 64 / LPN;
    const int lane = threadIdx.x & (LPN - 1);
    const int n = blockIdx.x * (kAggWaves * NPW) + threadIdx.x / LPN;
    const int n_nodes = n_nodes_ptr ? min(*n_nodes_ptr, n_nodes_max) : n_nodes_max;
    if (n >= n_nodes) return;
    const float m = *amax;
    const double scale = m > 0.0f ? kFixedOne / (double)m : 0.0;
    auto add = [&](size_t at, float v) {
        atomicAdd(reinterpret_cast<unsigned long long *>(acc + at), (unsigned long long)__double2ll_rn((double)v * scale));
    };
    const float *row = gA + (size_t)n * lda;
    for (int i = lane; i < cin; i += LPN) add((size_t)n * cin + i, row[25 * cin + i]);
    const int e0 = rowptr[n], e1 = rowptr[n + 1];
    for (int e = e0; e < e1; e++) {
        const int src = col[e];
        const int c = code[e];
        const Axis ax = spline_axis(c & 0xffff, rx, den_x);
        const Axis ay = spline_axis(c >> 16, ry, den_y);
        const float b00 = ax.b0 * ay.b0, b10 = ax.b1 * ay.b0, b01 = ax.b0 * ay.b1, b11 = ax.b1 * ay.b1;
        const float *a00 = row + (ax.k0 + 5 * ay.k0) * cin;
        const float *a10 = row + (ax.k1 + 5 * ay.k0) * cin;
        const float *a01 = row + (ax.k0 + 5 * ay.k1) * cin;
        const float *a11 = row + (ax.k1 + 5 * ay.k1) * cin;
        for (int i = lane; i < cin; i += LPN)
            add((size_t)src * cin + i, b00 * a00[i] + b10 * a10[i] + b01 * a01[i] + b11 * a11[i]);
    }
}

// The same scatter for the narrow convs of the event level (cin, cout <= 16; 400 k rows in a training step) WITHOUT the
// gA matrix: gA[n] = g[n] . Wm^T is 26 cin values per node -- 16 lanes rebuild the row in LDS from g[n] (16 values) and the
// weights (<= 26 KB, staged once per workgroup), then walk the node's edges exactly as k_tap_scatter_grad<16> does.  The
// [n, 26 cin] matrix (0.67 GB for the 16 -> 16 conv) was written by a library GEMM, read by a max-norm reduction and read
// again here.  `amax` must bound |gA| (the caller passes max|g| * max_k sum_co |Wm[k, co]|).
__global__ __launch_bounds__(kBlock) void k_tap_scatter_grad_w(const int32_t *__restrict__ n_nodes_ptr, int n_nodes_max,
                                                              const int32_t *__restrict__ rowptr,
                                                              const int32_t *__restrict__ col,
                                                              const int32_t *__restrict__ code,
                                                              const float *__restrict__ g, int ldg, int cout,
                                                              const float *__restrict__ Wm, int ldw, int cin, int rx,
                                                              int ry, float den_x, float den_y,
                                                              const float *__restrict__ amax,
                                                              long long *__restrict__ acc) {
    extern __shared__ __align__(16) float lds[];
    constexpr int LPN = 16, NPB = kBlock / LPN;
    const int K = 26 * cin;
    float *Ws = lds;                                  // [K][16], columns >= cout zero
    float *rows = lds + (size_t)K * 16;               // [NPB][K]
    for (int i = threadIdx.x; i < K * 16; i += kBlock) {
        const int k = i >> 4, co = i & 15;
        Ws[i] = co < cout ? Wm[(size_t)k * ldw + co] : 0.0f;
    }
    const int lane = threadIdx.x & (LPN - 1), slot = threadIdx.x / LPN;
    const int n = blockIdx.x * NPB + slot;
    const int n_nodes = n_nodes_ptr ? min(*n_nodes_ptr, n_nodes_max) : n_nodes_max;
    const bool active = n < n_nodes;
    const float gl = (active && lane < cout) ? g[(size_t)n * ldg + lane] : 0.0f;
    float gv[16];
#pragma unroll
    for (int co = 0; co < 16; co++) gv[co] = __shfl(gl, co, LPN);
    __syncthreads();
    float *row = rows + (size_t)slot * K;
    for (int k = lane; k < K; k += LPN) {
        const float4 *w = reinterpret_cast<const float4 *>(Ws + (size_t)k * 16);
        float a = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 wq = w[q];
            a += gv[4 * q] * wq.x; a += gv[4 * q + 1] * wq.y; a += gv[4 * q + 2] * wq.z; a += gv[4 * q + 3] * wq.w;
        }
        row[k] = a;
    }
    __syncthreads();
    if (!active) return;
    const float m = *amax;
    const double scale = m > 0.0f ? kFixedOne / (double)m : 0.0;
    auto add = [&](size_t at, float v) {
        atomicAdd(reinterpret_cast<unsigned long long *>(acc + at), (unsigned long long)__double2ll_rn((double)v * scale));
    };
    for (int i = lane; i < cin; i += LPN) add((size_t)n * cin + i, row[25 * cin + i]);
    const int e0 = rowptr[n], e1 = rowptr[n + 1];
    for (int e = e0; e < e1; e++) {
        const int src = col[e];
        const int c = code[e];
        const Axis ax = spline_axis(c & 0xffff, rx, den_x);
        const Axis ay = spline_axis(c >> 16, ry, den_y);
        const float b00 = ax.b0 * ay.b0, b10 = ax.b1 * ay.b0, b01 = ax.b0 * ay.b1, b11 = ax.b1 * ay.b1;
        const float *a00 = row + (ax.k0 + 5 * ay.k0) * cin;
        const float *a10 = row + (ax.k1 + 5 * ay.k0) * cin;
        const float *a01 = row + (ax.k0 + 5 * ay.k1) * cin;
        const float *a11 = row + (ax.k1 + 5 * ay.k1) * cin;
        for (int i = lane; i < cin; i += LPN)
            add((size_t)src * cin + i, b00 * a00[i] + b10 * a10[i] + b01 * a01[i] + b11 * a11[i]);
    }
}

__global__ __launch_bounds__(kBlock) void k_fixed_to_float(const int32_t *__restrict__ n_nodes_ptr, int n_nodes_max, int cin,
                                                          const float *__restrict__ amax,
                                                          const long long *__restrict__ acc, float *__restrict__ gx,
                                                          int ldg) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int n_nodes = n_nodes_ptr ? min(*n_nodes_ptr, n_nodes_max) : n_nodes_max;
    if (i >= (int64_t)n_nodes * cin) return;
    const int n = (int)(i / cin), c = (int)(i - (int64_t)n * cin);
    gx[(size_t)n * ldg + c] = (float)((double)acc[i] * ((double)*amax / kFixedOne));
}

// ---------------------------------------------------------------------------------------------
// Generic path, step 2: C[M,N] = act(A[M,K] . Wm[K,N] + bias[N]).  fp32, LDS-tiled 64x64x16,
// 256 threads x (4x4) outputs.  M is bounded on the device (*m_ptr) so no host sync is needed.
constexpr int GM = 64, GN = 64, GK = 16;
__global__ __launch_bounds__(kBlock) void k_gemm_bias_act(const int32_t *__restrict__ m_ptr, int m_max,
                                                         const float *__restrict__ A, int lda,
                                                         const float *__restrict__ Wm, int ldw,
                                                         const float *__restrict__ bias, float *__restrict__ C,
                                                         int ldc, int K, int N, int relu) {
    __shared__ float As[GK][GM + 4];
    __shared__ float Ws[GK][GN + 4];
    const int M = m_ptr ? min(*m_ptr, m_max) : m_max;
    const int m0 = blockIdx.x * GM, n0 = blockIdx.y * GN;
    if (m0 >= M) return;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // 16 x 16 threads
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += GK) {
        // A tile: 64 rows x 16 k  (256 threads x 4 elements, k fastest in memory)
        {
            const int r = threadIdx.x >> 2, kk = (threadIdx.x & 3) * 4;
            const int gm = m0 + r;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int gk = k0 + kk + q;
                As[kk + q][r] = (gm < M && gk < K) ? A[(size_t)gm * lda + gk] : 0.0f;
            }
        }
        // W tile: 16 k x 64 n
        {
            const int kk = threadIdx.x >> 4, c = (threadIdx.x & 15) * 4;
            const int gk = k0 + kk;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int gn = n0 + c + q;
                Ws[kk][c + q] = (gk < K && gn < N) ? Wm[(size_t)gk * ldw + gn] : 0.0f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < GK; kk++) {
            float a[4], w[4];
#pragma unroll
            for (int i = 0; i < 4; i++) a[i] = As[kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; j++) w[j] = Ws[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = fmaf(a[i], w[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int gm = m0 + ty * 4 + i;
        if (gm >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int gn = n0 + tx * 4 + j;
            if (gn >= N) continue;
            float v = acc[i][j] + (bias ? bias[gn] : 0.0f);
            if (relu) v = fmaxf(v, 0.0f);
            C[(size_t)gm * ldc + gn] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Level-0 fused SplineConv (event graph).  Offsets |dx|,|dy| <= r map to pseudo-coordinates whose
// taps fall in a 3-wide window per axis (v = 4*(d/(2MW)+0.5) in (1,3) because r < M*W), so only
// 3x3 of the 25 taps are active: `tab` holds, per offset code, the 9 products bx[a]*by[b] laid out
// [code][12] (window-relative taps, a fastest), built by k_build_l0_table with spline_axis().
//
// Weights (packed on the host, BN folded): rows [9*CIN taps | CIN root | CSKIP skip] x 16 outputs,
// staged in LDS with a row stride of 20 floats (80 B): lane i reads row (k*CIN+i) as 4 x b128,
// and rows i, i+1, ... land on disjoint bank quads (20*i mod 64), conflict-free.
constexpr int kL0Out = 16;
constexpr int kL0RowStride = 20;

template <int CIN, int CSKIP, int NT>
__global__ __launch_bounds__(kBlock, 4) void k_conv_l0(int N, int K, int ncodes, const int32_t *__restrict__ nbr_src,
                                                   const int16_t *__restrict__ nbr_code,
                                                   const int32_t *__restrict__ deg, const float *__restrict__ x,
                                                   int ldx, const float *__restrict__ xskip, int ldskip,
                                                   const float *__restrict__ tab,    // [ncodes][NTP]
                                                   const float *__restrict__ wpack,  // [((NT+1)*CIN+CSKIP)][16]
                                                   const float *__restrict__ shift,  // [16]
                                                   int relu, float *__restrict__ out, int ldo) {
    constexpr int NCH = (CIN + 15) / 16;   // channel slots per lane
    constexpr int NTP = (NT + 3) / 4 * 4;  // table row stride (floats), 16-byte aligned rows
    constexpr int NROWS = (NT + 1) * CIN + CSKIP;
    extern __shared__ __align__(16) float lds[];
    float *w_s = lds;                                   // NROWS * 20
    float *tab_s = lds + NROWS * kL0RowStride;          // ncodes * NTP
    for (int i = threadIdx.x; i < NROWS * kL0Out; i += kBlock)
        w_s[(i >> 4) * kL0RowStride + (i & 15)] = wpack[i];
    for (int i = threadIdx.x; i < ncodes * NTP; i += kBlock) tab_s[i] = tab[i];
    __syncthreads();

    const int l = threadIdx.x & 15;
    const int groups_per_block = kBlock / 16;
    const float my_shift = shift[l];
    const XcdSplit xs = xcd_split(N, groups_per_block, threadIdx.x >> 4);
    for (int n = xs.first; n < xs.end; n += xs.stride) {
        const int d = deg[n];
        const int64_t row = (int64_t)n * K;
        float A[NCH][NTP];
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
            for (int k = 0; k < NTP; k++) A[c][k] = 0.0f;
        // neighbour slots are read 16 at a time by the 16 lanes; all (up to 16) source rows are
        // requested before the first one is consumed (the gathers are the latency that matters here)
        for (int j0 = 0; j0 < d; j0 += 16) {
            int my_src = 0, my_code = 0;
            if (j0 + l < d) { my_src = nbr_src[row + j0 + l]; my_code = nbr_code[row + j0 + l]; }
            const int cnt = min(16, d - j0);
            float v[NCH][16];
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int src = __shfl(my_src, j, 16);
                const float *xs = x + (size_t)src * ldx;
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    const int ch = c * 16 + l;
                    v[c][j] = (j < cnt && ch < CIN) ? xs[ch] : 0.0f;
                }
            }
#pragma unroll
            for (int j = 0; j < 16; j++) {
                if (j < cnt) {
                    const int code = __shfl(my_code, j, 16);
                    float t[NTP];
#pragma unroll
                    for (int q = 0; q < NTP / 4; q++) {
                        const float4 tq = *reinterpret_cast<const float4 *>(tab_s + code * NTP + 4 * q);
                        t[4 * q] = tq.x; t[4 * q + 1] = tq.y; t[4 * q + 2] = tq.z; t[4 * q + 3] = tq.w;
                    }
#pragma unroll
                    for (int c = 0; c < NCH; c++)
#pragma unroll
                        for (int k = 0; k < NT; k++) A[c][k] = fmaf(t[k], v[c][j], A[c][k]);
                }
                __builtin_amdgcn_sched_barrier(0);  // keep the table reads of later slots from piling up in VGPRs
            }
        }
        // contraction: lane i owns rows (k*CIN + i); 16 partial outputs per lane
        float p[16];
#pragma unroll
        for (int o = 0; o < 16; o++) p[o] = 0.0f;
        auto fma_row = [&](float a, int r) {
            const float4 *w4 = reinterpret_cast<const float4 *>(w_s + r * kL0RowStride);
            const float4 w0 = w4[0], w1 = w4[1], w2 = w4[2], w3 = w4[3];
            p[0] = fmaf(a, w0.x, p[0]); p[1] = fmaf(a, w0.y, p[1]); p[2] = fmaf(a, w0.z, p[2]); p[3] = fmaf(a, w0.w, p[3]);
            p[4] = fmaf(a, w1.x, p[4]); p[5] = fmaf(a, w1.y, p[5]); p[6] = fmaf(a, w1.z, p[6]); p[7] = fmaf(a, w1.w, p[7]);
            p[8] = fmaf(a, w2.x, p[8]); p[9] = fmaf(a, w2.y, p[9]); p[10] = fmaf(a, w2.z, p[10]); p[11] = fmaf(a, w2.w, p[11]);
            p[12] = fmaf(a, w3.x, p[12]); p[13] = fmaf(a, w3.y, p[13]); p[14] = fmaf(a, w3.z, p[14]); p[15] = fmaf(a, w3.w, p[15]);
        };
        const float *xn = x + (size_t)n * ldx;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const int ch = c * 16 + l;
            if (ch < CIN) {
#pragma unroll
                for (int k = 0; k < NT; k++) fma_row(A[c][k], k * CIN + ch);
                fma_row(xn[ch], NT * CIN + ch);  // root weight
            }
        }
        if (CSKIP > 0) {
            const float *sn = xskip + (size_t)n * ldskip;
#pragma unroll
            for (int c = 0; c < (CSKIP + 15) / 16; c++) {
                const int ch = c * 16 + l;
                if (ch < CSKIP) fma_row(sn[ch], (NT + 1) * CIN + ch);
            }
        }
        // transpose-reduce over the 16 lanes: lane o ends with sum_i p_i[o]
        float q8[8];
        {
            const bool hi = (l & 8) != 0;
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const float keep = hi ? p[m + 8] : p[m];
                const float send = hi ? p[m] : p[m + 8];
                q8[m] = keep + __shfl_xor(send, 8, 16);
            }
        }
        float q4[4];
        {
            const bool hi = (l & 4) != 0;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const float keep = hi ? q8[m + 4] : q8[m];
                const float send = hi ? q8[m] : q8[m + 4];
                q4[m] = keep + __shfl_xor(send, 4, 16);
            }
        }
        float q2[2];
        {
            const bool hi = (l & 2) != 0;
#pragma unroll
            for (int m = 0; m < 2; m++) {
                const float keep = hi ? q4[m + 2] : q4[m];
                const float send = hi ? q4[m] : q4[m + 2];
                q2[m] = keep + __shfl_xor(send, 2, 16);
            }
        }
        float r;
        {
            const bool hi = (l & 1) != 0;
            const float keep = hi ? q2[1] : q2[0];
            const float send = hi ? q2[0] : q2[1];
            r = keep + __shfl_xor(send, 1, 16);
        }
        r += my_shift;
        if (relu) r = fmaxf(r, 0.0f);
        out[(size_t)n * ldo + l] = r;
    }
}

// One thread per offset code: the window products bx[a]*by[b] at [a + tx*b], row stride ntp.
__global__ void k_build_l0_table(int rx, int ry, float den_x, float den_y, int win_x, int tx, int win_y, int ty,
                                 int ntp, float *__restrict__ tab, int32_t *__restrict__ bad) {
    const int code = blockIdx.x * blockDim.x + threadIdx.x;
    const int sx = 2 * rx + 1, sy = 2 * ry + 1;
    if (code >= sx * sy) return;
    const Axis ax = spline_axis(code / sy, rx, den_x);
    const Axis ay = spline_axis(code % sy, ry, den_y);
    float bx[5] = {0, 0, 0, 0, 0}, by[5] = {0, 0, 0, 0, 0};
    // a tap outside the window is only acceptable with zero weight (pseudo exactly on a knot)
    const int ax0 = ax.k0 - win_x, ax1 = ax.k1 - win_x, ay0 = ay.k0 - win_y, ay1 = ay.k1 - win_y;
    if (ax0 >= 0 && ax0 < tx) bx[ax0] = ax.b0; else if (ax.b0 != 0.0f) atomicOr(bad, 1);
    if (ax1 >= 0 && ax1 < tx) bx[ax1] = ax.b1; else if (ax.b1 != 0.0f) atomicOr(bad, 1);
    if (ay0 >= 0 && ay0 < ty) by[ay0] = ay.b0; else if (ay.b0 != 0.0f) atomicOr(bad, 1);
    if (ay1 >= 0 && ay1 < ty) by[ay1] = ay.b1; else if (ay.b1 != 0.0f) atomicOr(bad, 1);
    for (int k = 0; k < ntp; k++) tab[code * ntp + k] = 0.0f;
    for (int b = 0; b < ty; b++)
        for (int a = 0; a < tx; a++) tab[code * ntp + a + tx * b] = bx[a] * by[b];
}

}  // namespace
}  // namespace dagr

using namespace dagr;

extern "C" {

int dagr_spline_tap_window(int32_t r, float den, int32_t *first_tap_host, int32_t *num_taps_host) {
    DAGR_CHECK_ARG(r >= 0 && den > 0 && first_tap_host && num_taps_host, "bad arguments");
    int lo = 5, hi = -1;
    for (int idx = 0; idx <= 2 * r; idx++) {
        const Axis a = spline_axis(idx, r, den);
        if (a.b0 != 0.0f) { lo = a.k0 < lo ? a.k0 : lo; hi = a.k0 > hi ? a.k0 : hi; }
        if (a.b1 != 0.0f) { lo = a.k1 < lo ? a.k1 : lo; hi = a.k1 > hi ? a.k1 : hi; }
    }
    *first_tap_host = lo;
    *num_taps_host = hi - lo + 1;
    return DAGR_OK;
}

int dagr_spline_l0_table(int32_t rx, int32_t ry, float den_x, float den_y, int32_t win_x, int32_t tx, int32_t win_y,
                         int32_t ty, float *tab, int32_t *bad_flag, void *stream) {
    DAGR_CHECK_ARG(tab && bad_flag && rx >= 0 && ry >= 0, "bad arguments");
    DAGR_CHECK_ARG(tx >= 1 && tx <= 5 && ty >= 1 && ty <= 5 && win_x >= 0 && win_y >= 0 && win_x + tx <= 5 &&
                       win_y + ty <= 5, "bad tap window");
    const int n = (2 * rx + 1) * (2 * ry + 1);
    const int ntp = (tx * ty + 3) / 4 * 4;
    DAGR_CHECK_HIP(hipMemsetAsync(bad_flag, 0, 4, (hipStream_t)stream));
    k_build_l0_table<<<(unsigned)ceil_div(n, 64), 64, 0, (hipStream_t)stream>>>(rx, ry, den_x, den_y, win_x, tx, win_y,
                                                                             ty, ntp, tab, bad_flag);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

int dagr_spline_conv_l0(int32_t cin, int32_t cskip, int32_t ntaps, int64_t N, int32_t K, int32_t ncodes,
                        const int32_t *nbr_src,
                        const int16_t *nbr_code, const int32_t *deg, const float *x, int32_t ldx,
                        const float *xskip, int32_t ldskip, const float *tab, const float *wpack, const float *shift,
                        int32_t relu, float *out, int32_t ldo, void *stream_) {
    DAGR_CHECK_ARG(N >= 0, "N < 0");
    if (N == 0) return DAGR_OK;
    DAGR_CHECK_ARG(nbr_src && nbr_code && deg && x && tab && wpack && shift && out, "NULL pointer");
    DAGR_CHECK_ARG(cskip == 0 || xskip, "xskip is NULL");
    hipStream_t stream = (hipStream_t)stream_;
    const int nrows = (ntaps + 1) * cin + cskip;
    const int ntp = (ntaps + 3) / 4 * 4;
    const size_t lds_bytes = ((size_t)nrows * kL0RowStride + (size_t)ncodes * ntp) * 4;
    DAGR_CHECK_ARG(lds_bytes <= 160 * 1024, "weights + offset table exceed LDS");
    const int groups = kBlock / 16;
    const int64_t useful = ceil_div(N, groups);
    unsigned grid = 1;
#define DAGR_L0_CASE(CI, CS, NTAPS)                                                                                \
    if (cin == CI && cskip == CS && ntaps == NTAPS) {                                                              \
        {                                                                                                          \
            static thread_local size_t set_for = 0;                                                                \
            if (set_for != lds_bytes) {                                                                            \
                DAGR_CHECK_HIP(hipFuncSetAttribute((const void *)k_conv_l0<CI, CS, NTAPS>,                         \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));    \
                set_for = lds_bytes;                                                                               \
            }                                                                                                      \
        }                                                                                                          \
        grid = round_grid8(persistent_grid(k_conv_l0<CI, CS, NTAPS>, kBlock, lds_bytes, useful));                               \
        k_conv_l0<CI, CS, NTAPS><<<grid, kBlock, lds_bytes, stream>>>((int)N, K, ncodes, nbr_src, nbr_code, deg, x, \
                                                                      ldx, xskip, ldskip, tab, wpack, shift, relu,  \
                                                                      out, ldo);                                    \
        DAGR_CHECK_LAUNCH();                                                                                       \
        return DAGR_OK;                                                                                            \
    }
#define DAGR_L0_CASES(NTAPS)                                                                   \
    DAGR_L0_CASE(3, 0, NTAPS)   /* events-only conv_block1.conv_block1 (net.py:75) */          \
    DAGR_L0_CASE(16, 3, NTAPS)  /* events-only conv_block1.conv_block2 + skip Linear 3->16 */  \
    DAGR_L0_CASE(19, 0, NTAPS)  /* --use_image: 1 + 16 image channels + 2 */                   \
    DAGR_L0_CASE(16, 19, NTAPS)
    DAGR_L0_CASES(9)    // 3x3 taps (square sensors)
    DAGR_L0_CASES(15)   // 3x5 taps (the y extent is normalised by the width-derived radius)
    DAGR_L0_CASES(25)   // full 5x5
#undef DAGR_L0_CASES
#undef DAGR_L0_CASE
    set_error("dagr_spline_conv_l0: unsupported (cin, cskip, ntaps) combination");
    return DAGR_ERR_UNSUPPORTED;
}

int dagr_spline_tap_aggregate(const int32_t *n_nodes_ptr, int32_t n_nodes_max, const int32_t *rowptr,
                              const int32_t *col, const int32_t *code, const float *x, int32_t ldx, int32_t cin,
                              const float *xskip, int32_t ldskip, int32_t cskip, int32_t rx, int32_t ry, float den_x,
                              float den_y, float *A, int32_t lda, void *stream) {
    DAGR_CHECK_ARG(n_nodes_max >= 0, "n_nodes_max < 0");
    if (n_nodes_max == 0) return DAGR_OK;
    DAGR_CHECK_ARG(rowptr && col && code && x && A, "NULL pointer");
    DAGR_CHECK_ARG(cin >= 1 && lda >= 26 * cin + cskip, "lda too small");
    const bool narrow = cin <= 16;                    // four nodes per wave
    const int nodes_per_block = kAggWaves * (narrow ? 4 : 1);
    const size_t lds_bytes = (size_t)nodes_per_block * 25 * cin * 4;
    DAGR_CHECK_ARG(lds_bytes <= 160 * 1024, "cin too large for the LDS accumulators");
    {
        static thread_local size_t set_max[2] = {0, 0};   // the attribute is a maximum: raise it only when needed
        if (lds_bytes > set_max[narrow]) {
            DAGR_CHECK_HIP(hipFuncSetAttribute(narrow ? (const void *)k_tap_aggregate<16> : (const void *)k_tap_aggregate<64>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            set_max[narrow] = lds_bytes;
        }
    }
    const unsigned grid = (unsigned)ceil_div(n_nodes_max, nodes_per_block);
    if (narrow)
        k_tap_aggregate<16><<<grid, kBlock, lds_bytes, (hipStream_t)stream>>>(
            n_nodes_ptr, n_nodes_max, rowptr, col, code, x, ldx, cin, xskip, ldskip, cskip, rx, ry, den_x, den_y, A, lda);
    else
        k_tap_aggregate<64><<<grid, kBlock, lds_bytes, (hipStream_t)stream>>>(
            n_nodes_ptr, n_nodes_max, rowptr, col, code, x, ldx, cin, xskip, ldskip, cskip, rx, ry, den_x, den_y, A, lda);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

int dagr_spline_tap_scatter_grad(const int32_t *n_nodes_ptr, int32_t n_nodes_max, const int32_t *rowptr,
                                 const int32_t *col, const int32_t *code, const float *grad_A, int32_t lda, int32_t cin,
                                 int32_t rx, int32_t ry, float den_x, float den_y, const float *grad_A_absmax,
                                 int64_t *acc, float *grad_x, int32_t ldg, void *stream) {
    DAGR_CHECK_ARG(n_nodes_max >= 0, "n_nodes_max < 0");
    if (n_nodes_max == 0) return DAGR_OK;
    DAGR_CHECK_ARG(rowptr && col && code && grad_A && grad_x && grad_A_absmax && acc, "NULL pointer");
    DAGR_CHECK_ARG(cin >= 1 && lda >= 26 * cin && ldg >= cin, "bad strides");
    if (cin <= 16)
        k_tap_scatter_grad<16><<<(unsigned)ceil_div(n_nodes_max, kAggWaves * 4), kBlock, 0, (hipStream_t)stream>>>(
            n_nodes_ptr, n_nodes_max, rowptr, col, code, grad_A, lda, cin, rx, ry, den_x, den_y, grad_A_absmax,
            (long long *)acc);
    else
        k_tap_scatter_grad<64><<<(unsigned)ceil_div(n_nodes_max, kAggWaves), kBlock, 0, (hipStream_t)stream>>>(
            n_nodes_ptr, n_nodes_max, rowptr, col, code, grad_A, lda, cin, rx, ry, den_x, den_y, grad_A_absmax,
            (long long *)acc);
    DAGR_CHECK_LAUNCH();
    k_fixed_to_float<<<(unsigned)ceil_div((int64_t)n_nodes_max * cin, kBlock), kBlock, 0, (hipStream_t)stream>>>(
        n_nodes_ptr, n_nodes_max, cin, grad_A_absmax, (const long long *)acc, grad_x, ldg);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

int dagr_spline_tap_scatter_grad_w(const int32_t *n_nodes_ptr, int32_t n_nodes_max, const int32_t *rowptr,
                                   const int32_t *col, const int32_t *code, const float *grad_out, int32_t ldg, int32_t cout,
                                   const float *Wm, int32_t ldw, int32_t cin, int32_t rx, int32_t ry, float den_x,
                                   float den_y, const float *grad_A_bound, int64_t *acc, float *grad_x, int32_t ldgx,
                                   void *stream) {
    DAGR_CHECK_ARG(n_nodes_max >= 0, "n_nodes_max < 0");
    if (n_nodes_max == 0) return DAGR_OK;
    DAGR_CHECK_ARG(rowptr && col && code && grad_out && Wm && grad_x && grad_A_bound && acc, "NULL pointer");
    DAGR_CHECK_ARG(cin >= 1 && cin <= 16 && cout >= 1 && cout <= 16 && ldg >= cout && ldw >= cout && ldgx >= cin,
                   "the fused form covers cin, cout <= 16 (use dagr_spline_tap_scatter_grad)");
    const size_t lds_bytes = ((size_t)26 * cin * 16 + (size_t)(kBlock / 16) * 26 * cin) * 4;     // <= 53 KB
    k_tap_scatter_grad_w<<<(unsigned)ceil_div(n_nodes_max, kBlock / 16), kBlock, lds_bytes, (hipStream_t)stream>>>(
        n_nodes_ptr, n_nodes_max, rowptr, col, code, grad_out, ldg, cout, Wm, ldw, cin, rx, ry, den_x, den_y, grad_A_bound,
        (long long *)acc);
    DAGR_CHECK_LAUNCH();
    k_fixed_to_float<<<(unsigned)ceil_div((int64_t)n_nodes_max * cin, kBlock), kBlock, 0, (hipStream_t)stream>>>(
        n_nodes_ptr, n_nodes_max, cin, grad_A_bound, (const long long *)acc, grad_x, ldgx);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

int dagr_gemm_bias_act(const int32_t *m_ptr, int32_t m_max, const float *A, int32_t lda, const float *Wm,
                       int32_t ldw, const float *bias, float *C, int32_t ldc, int32_t K, int32_t N, int32_t relu,
                       void *stream) {
    DAGR_CHECK_ARG(m_max >= 0 && K >= 1 && N >= 1, "bad sizes");
    if (m_max == 0) return DAGR_OK;
    DAGR_CHECK_ARG(A && Wm && C, "NULL pointer");
    // wide outputs with 16-byte aligned rows go to the exact-fp32 MFMA kernel (gemm.hip)
    if (ldw % 8 == 0 && ldw >= N && lda % 4 == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)Wm % 16) == 0) {
        DAGR_CHECK_HIP(launch_gemm_mfma(m_ptr, m_max, A, lda, Wm, ldw, bias, C, ldc, K, N, relu, (hipStream_t)stream));
        return DAGR_OK;
    }
    dim3 grid((unsigned)ceil_div(m_max, GM), (unsigned)ceil_div(N, GN));
    k_gemm_bias_act<<<grid, kBlock, 0, (hipStream_t)stream>>>(m_ptr, m_max, A, lda, Wm, ldw, bias, C, ldc, K, N, relu);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

}  // extern "C"
