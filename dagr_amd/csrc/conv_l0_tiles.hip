// conv_l0_tiles.hip -- level-0 SplineConv (event graph) as 16-node wave tiles, gfx950, fp32.
//
// Reference op: MySplineConv.forward/_forward/message_lut (src/dagr/model/layers/spline_conv.py:39-78) fused with
// BatchNorm(eval)+ReLU (model/layers/conv.py:23-28) and the skip Linear+BN of ConvBlockWithSkip (conv.py:47-56);
// same re-association as spline_conv.hip:   A[n][tap][ch] = sum_edges basis[edge][tap] * x[src][ch]   (phase 1)
//                                           out[n][:]      = [A[n] | x[n] | xskip[n]] . Wpack + shift   (phase 2)
//
// What limited the previous level-0 kernels was LDS bandwidth (per-edge offset-table rows and per-node weight rows as
// ds_read_b128, PMC: profiles/r1_final_image_pmc_sq.csv) and, in the MFMA variant, the staging of phase-1 results through
// LDS to turn "one node per 16-lane group" into the MFMA A-operand layout.  This kernel is laid out so that neither
// exists:
//   * one wave owns a tile of 16 consecutive nodes (node = CSR slot, so a tile is a run of events of neighbouring
//     pixels); lane (c = l & 15, q = l >> 4) works for NODE c on channel quad q: it walks the node's neighbour list
//     itself (<= 16 entries) and keeps A[tap][4q..4q+3] in registers -- TX*TY*4 accumulators (60 for the 3x5 tap window
//     of a 640x480 sensor).  A source row is 64 bytes read as four 16-byte pieces by the node's four lanes.
//   * those registers already ARE the A operand of v_mfma_f32_16x16x4_f32 (A[i = l & 15][k = l >> 4]: row = node,
//     k-slice = this lane's channel quad): phase 2 is TX*TY*4 (+ root/extras/skip) MFMAs straight from registers; the
//     B operands (packed weights, re-laid at kernel start into [k-step][lane] order) are one conflict-free
//     ds_read_b32 each.  No staging tile, no per-edge LDS table: the per-axis basis weights (2r+1 rows of 4 / 8 floats)
//     sit in LDS and are combined in registers.
//   * channels beyond the 16 "main" ones (the first conv of the --use_image model has 19 inputs, the events-only one 3)
//     are "extras": lane q < CE keeps A[tap] for channel CM + q and feeds one extra k-step per tap.
// Exact fp32 throughout (the f32 MFMA is a k-ordered fmaf chain); only the summation order differs from the oracle.
#include "common.hpp"

#include <stdlib.h>

namespace dagr {
namespace {

using f32x4_t = __attribute__((ext_vector_type(4))) float;
using f32x2_t = __attribute__((ext_vector_type(2))) float;

template <int CM, int CE, int CS, int TX, int TY>
struct L0Steps {
    static constexpr int NT = TX * TY;
    static constexpr int CIN = CM + CE;
    static constexpr int SKF = CS >= 16 ? 1 : 0;                  // skip columns 0..15 as a float4 per lane
    static constexpr int SKE = CS - 16 * SKF;                     // remaining skip columns (one k-step, lane q < SKE)
    static constexpr int kMain = CM ? NT * 4 : 0;                 // step s = tap*4 + r : row tap*CIN + 4q + r
    static constexpr int kExtra = CE ? NT : 0;                    // step tap          : row tap*CIN + CM + q (q < CE)
    static constexpr int kRootM = CM ? 4 : 0;                     // step r            : row NT*CIN + 4q + r
    static constexpr int kRootE = CE ? 1 : 0;                     //                     row NT*CIN + CM + q
    static constexpr int kSkipF = SKF ? 4 : 0;                    // step r            : row (NT+1)*CIN + 4q + r
    static constexpr int kSkipE = SKE ? 1 : 0;                    //                     row (NT+1)*CIN + 16*SKF + q
    static constexpr int N = kMain + kExtra + kRootM + kRootE + kSkipF + kSkipE;
    static_assert(CM == 0 || CM == 16, "main channel block is 0 or 16 wide");
    static_assert(CE >= 0 && CE <= 4 && SKE >= 0 && SKE <= 4, "extras / skip remainder ride on the 4 lanes of a node");
    // packed-weight row behind k-step s for lane quad q, or -1 (zero operand)
    __host__ __device__ static int row(int s, int q) {
        if (s < kMain) return (s >> 2) * CIN + 4 * q + (s & 3);
        s -= kMain;
        if (s < kExtra) return q < CE ? s * CIN + CM + q : -1;
        s -= kExtra;
        if (s < kRootM) return NT * CIN + 4 * q + s;
        s -= kRootM;
        if (s < kRootE) return q < CE ? NT * CIN + CM + q : -1;
        s -= kRootE;
        if (s < kSkipF) return (NT + 1) * CIN + 4 * q + s;
        s -= kSkipF;
        return q < SKE ? (NT + 1) * CIN + 16 * SKF + q : -1;
    }
};

constexpr int kTileWaves = 4;   // waves per workgroup (each owns its tiles; they share the weight image in LDS)

template <int CM, int CE, int CS, int TX, int TY, bool LEAN>
__global__ __launch_bounds__(kTileWaves * 64, LEAN ? 3 : 2) void k_conv_l0_tiles(
    int n_first, int N, const int32_t *__restrict__ n_ptr, int rx, int ry, float den_x, float den_y, int win_x, int win_y, const int32_t *__restrict__ nbr_src,
    const int16_t *__restrict__ nbr_code, const int32_t *__restrict__ deg, const float *__restrict__ x, int ldx,
    const float *__restrict__ xskip, int ldskip, const float *__restrict__ wpack, const float *__restrict__ shift,
    int relu, float *__restrict__ out, int ldo, int ablate) {
    using S = L0Steps<CM, CE, CS, TX, TY>;
    constexpr int NT = S::NT;
    extern __shared__ __align__(16) float lds[];
    float *w_l = lds;                            // [S::N][64]  B operands in [k-step][lane] order
    // [(2rx+1)(2ry+1) + 1][WS]: the NT tap weights bx[a] * by[b] of every offset code (row = the code the graph builder
    // stores: ix * (2ry+1) + iy), the last row all zero (absent edges).  Until round 4 the kernel kept the two per-axis
    // tables and rebuilt the products per edge: code decode + three table rows + 8 packed multiplies + 5 selects = 17 of
    // the 51 VALU instructions an edge cost on a kernel that is bound by instruction issue; now an edge reads its row.
    // Row stride = NT rounded up to 4, + 4: consecutive rows start 5 (or 3) 16-byte slots apart, so the 16 rows a
    // ds_read_b128 group touches spread over the bank slots instead of piling onto four.
    // Measured (800 k nodes): 16 -> 16 + skip 0.173 -> 0.162 ms (image), 0.168 -> 0.155 ms (events-only); 19 -> 16 conv
    // 0.226 -> 0.222 ms (with the freed registers it fits three waves per SIMD without scratch); the 3 -> 16 conv, whose
    // edge is 15 scalar FMAs, becomes LDS-bound on the four row reads (0.085 -> 0.150 ms) and keeps the per-axis tables.
    constexpr bool PRE = CM > 0;
    constexpr int WS = (NT + 3) / 4 * 4 + 4;
    float *wt_l = w_l + S::N * 64;
    float *ax_l = wt_l;                          // !PRE: [2rx+1][4]  per-axis basis weights of the TX-wide tap window
    float *ay_l = ax_l + (2 * rx + 1) * 4;       //       [2ry+1][8]
    const int sy = 2 * ry + 1;
    const int n_codes = (2 * rx + 1) * sy;
    for (int i = threadIdx.x; i < S::N * 64; i += blockDim.x) {
        const int s = i >> 6, lq = (i >> 4) & 3, c = i & 15;
        const int r = S::row(s, lq);
        w_l[i] = r >= 0 ? wpack[r * 16 + c] : 0.0f;
    }
    for (int i = threadIdx.x; PRE && i < (n_codes + 1) * WS; i += blockDim.x) {
        const int o = i / WS, t = i - o * WS;
        float w = 0.0f;
        if (o < n_codes && t < NT) {
            const Axis ax = spline_axis(o / sy, rx, den_x);
            const Axis ay = spline_axis(o % sy, ry, den_y);
            const int ta = t % TX + win_x, tb = t / TX + win_y;     // taps of the 5-tap kernel this column stands for
            const float wx = (ta == ax.k0 ? ax.b0 : 0.0f) + (ta == ax.k1 ? ax.b1 : 0.0f);
            const float wy = (tb == ay.k0 ? ay.b0 : 0.0f) + (tb == ay.k1 ? ay.b1 : 0.0f);
            w = wx * wy;                                            // == the level-0 offset table entry (bx[a]*by[b])
        }
        wt_l[i] = w;
    }
    for (int i = threadIdx.x; !PRE && i < (2 * rx + 1) * 4; i += blockDim.x) {
        const Axis a = spline_axis(i >> 2, rx, den_x);
        const int t = (i & 3) + win_x;           // tap of the 5-tap kernel this column stands for
        ax_l[i] = (i & 3) < TX ? ((t == a.k0 ? a.b0 : 0.0f) + (t == a.k1 ? a.b1 : 0.0f)) : 0.0f;
    }
    for (int i = threadIdx.x; !PRE && i < (2 * ry + 1) * 8; i += blockDim.x) {
        const Axis a = spline_axis(i >> 3, ry, den_y);
        const int t = (i & 7) + win_y;
        ay_l[i] = (i & 7) < TY ? ((t == a.k0 ? a.b0 : 0.0f) + (t == a.k1 ? a.b1 : 0.0f)) : 0.0f;
    }
    __syncthreads();

    // n_ptr: the level's node count in device memory (launches sized for a capacity N: captured HIP graphs)
    if (n_ptr) N = max(0, min(N, *n_ptr - n_first));
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = l & 15, q = l >> 4;
    const float my_shift = shift[c];
    const float inv_sy = 1.0f / (float)(2 * ry + 1);
    // XCD x = blockIdx % 8 owns the x-th eighth of the nodes (one sample of a B = 8 batch); its workgroups sweep that range
    // TOGETHER: workgroup lb takes the 64-node groups lb, lb + bpx, lb + 2 bpx ... (one 16-node tile per wave).  The source
    // rows a tile gathers lie within +-r pixel rows of it, so the rows all workgroups of the XCD need at one time form a band
    // of a few hundred KB that stays in the XCD's 4-MB L2.  (Contiguous strips per workgroup -- rounds 1-2 -- put 64 strips
    // in flight per XCD, together the whole 8 MB of its rows: every source row came from HBM ~3 times, PMC.)
    const int G = gridDim.x, nx = (G % 8 == 0) ? 8 : 1;
    const int xcd = blockIdx.x % nx, lb = blockIdx.x / nx, bpx = G / nx;
    const int chunk = ((N + nx - 1) / nx + 15) / 16 * 16;
    const int stride = 16 * kTileWaves * bpx;
    // nodes [n_first, n_first + N): the whole level, or the rows an asynchronous update appended (a node's result does
    // not depend on the tile it shares with its neighbours in memory)
    const int n_begin = n_first + xcd * chunk + lb * 16 * kTileWaves;
    const int n_end = n_first + min(N, (xcd + 1) * chunk);

    // Software pipeline over this wave's tiles.  Per tile the dependent chain is {degree, neighbour row} -> source rows ->
    // FMAs; with ~200 registers per lane only two waves share a SIMD, so the chain is shortened instead of hidden: the
    // neighbour row of the NEXT tile is requested while this tile's source rows are in flight, and all (<= 16) source
    // rows of a tile are requested before the first one is consumed (two batches of 8).
    auto load_meta = [&](int n0_, int &d_, int4(&s_)[4], int2(&c_)[4]) {
        const int n_ = n0_ + c;
        const bool ok_ = n0_ < n_end && n_ < n_end;
        const int nn_ = ok_ ? n_ : min(n0_, n_end - 1);
        d_ = ok_ ? deg[nn_] : 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            s_[k] = *reinterpret_cast<const int4 *>(nbr_src + (size_t)nn_ * 16 + 4 * k);
            c_[k] = *reinterpret_cast<const int2 *>(nbr_code + (size_t)nn_ * 16 + 4 * k);
        }
    };
    int d_nx;
    int4 s_nx[4];
    int2 c_nx[4];
    // LEAN: three waves per SIMD instead of two (<= 170 registers): the next tile's neighbour row is not held across the tile
    // and the source rows come in two batches of 8 through the same registers -- the waits are hidden by the third wave
    // instead of being shortened
    if (!LEAN && n_begin + 16 * wv < n_end) load_meta(n_begin + 16 * wv, d_nx, s_nx, c_nx);
    for (int n0 = n_begin + 16 * wv; n0 < n_end; n0 += stride) {
        if (LEAN) load_meta(n0, d_nx, s_nx, c_nx);
        const int n = n0 + c;
        const bool valid = n < n_end;
        const int nn = valid ? n : n0;                  // a row that exists, for the predicated-off lanes
        const int d = d_nx;
        // offset codes: with a main channel block (register-bound instantiations) they stay packed two per register until
        // their edge; the 3 -> 16 conv has registers to spare and unpacks them once (measured: 0.085 vs 0.090 ms)
        int srcs[16], cpk[8], cun[CM ? 1 : 16];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            srcs[4 * k] = s_nx[k].x; srcs[4 * k + 1] = s_nx[k].y; srcs[4 * k + 2] = s_nx[k].z; srcs[4 * k + 3] = s_nx[k].w;
            cpk[2 * k] = c_nx[k].x; cpk[2 * k + 1] = c_nx[k].y;
            if (!CM) {
                cun[(4 * k) % (CM ? 1 : 16)] = c_nx[k].x & 0xffff; cun[(4 * k + 1) % (CM ? 1 : 16)] = (c_nx[k].x >> 16) & 0xffff;
                cun[(4 * k + 2) % (CM ? 1 : 16)] = c_nx[k].y & 0xffff; cun[(4 * k + 3) % (CM ? 1 : 16)] = (c_nx[k].y >> 16) & 0xffff;
            }
        }
        auto code_of = [&](int u) { return CM ? ((cpk[u >> 1] >> (16 * (u & 1))) & 0xffff) : cun[u % (CM ? 1 : 16)]; };
        int dmax = d;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) dmax = max(dmax, __shfl_xor(dmax, off, 16));
        // source rows: first batch of 8, and the second one when any node of the tile has more than 8 in-edges
        float4 xv[LEAN ? 8 : 16];
        float xe[LEAN ? 8 : 16];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int src = (u < d) ? srcs[u] : nn;
            if (CM) xv[u] = *reinterpret_cast<const float4 *>(x + (size_t)src * ldx + 4 * q);
            if (CE) xe[u] = (q < CE) ? x[(size_t)src * ldx + CM + q] : 0.f;
        }
        if (!LEAN && dmax > 8) {
#pragma unroll
            for (int u = 8; u < 16; u++) {
                const int src = (u < d) ? srcs[u] : nn;
                if (CM) xv[u % (LEAN ? 8 : 16)] = *reinterpret_cast<const float4 *>(x + (size_t)src * ldx + 4 * q);
                if (CE) xe[u % (LEAN ? 8 : 16)] = (q < CE) ? x[(size_t)src * ldx + CM + q] : 0.f;
            }
        }
        // root / skip operands of phase 2 and the next tile's neighbour row: requested now, consumed later
        float4 xr = make_float4(0.f, 0.f, 0.f, 0.f), xs = make_float4(0.f, 0.f, 0.f, 0.f);
        float xre = 0.f, xse = 0.f;
        if (CM) xr = *reinterpret_cast<const float4 *>(x + (size_t)nn * ldx + 4 * q);
        if (CE) xre = (q < CE) ? x[(size_t)nn * ldx + CM + q] : 0.f;
        if (S::SKF) xs = *reinterpret_cast<const float4 *>(xskip + (size_t)nn * ldskip + 4 * q);
        if (S::SKE) xse = (q < S::SKE) ? xskip[(size_t)nn * ldskip + 16 * S::SKF + q] : 0.f;
        if (!LEAN) load_meta(n0 + stride, d_nx, s_nx, c_nx);

        // A[tap][4 channels] as two packed pairs: the 60 FMAs of an edge issue as 30 v_pk_fma_f32 (two fp32 FMAs per lane
        // and instruction -- the rate the 157 TFLOP/s fp32 peak is quoted for; scalar v_fma_f32 tops out at half of it).
        // Same fused multiply-adds, same order per accumulator: bit-identical results.
        f32x2_t acc[CM ? NT : 1][2];
        float acce[CE ? NT : 1];
#pragma unroll
        for (int t = 0; t < (CM ? NT : 1); t++) { acc[t][0] = f32x2_t{0.f, 0.f}; acc[t][1] = f32x2_t{0.f, 0.f}; }
#pragma unroll
        for (int t = 0; t < (CE ? NT : 1); t++) acce[t] = 0.f;

        // ---- phase 1: this lane's node, its <= 16 in-edges.  (Requesting the basis rows of edge u+1 before the FMAs of edge
        // u was measured: 3-10 % slower -- more live registers, no shorter chain.)
        auto edge = [&](int u) {
            constexpr int NX = LEAN ? 8 : 16;
            auto tap = [&](int t, float w) {
                if (CM) {
                    const f32x2_t w2 = {w, w};
                    acc[t][0] = __builtin_elementwise_fma(w2, f32x2_t{xv[u % NX].x, xv[u % NX].y}, acc[t][0]);
                    acc[t][1] = __builtin_elementwise_fma(w2, f32x2_t{xv[u % NX].z, xv[u % NX].w}, acc[t][1]);
                }
                if (CE) acce[t] = fmaf(w, xe[u % NX], acce[t]);
            };
            if constexpr (PRE) {
                const int code = (u < d) ? code_of(u) : n_codes;      // absent edge: the all-zero row
                const float *wr = wt_l + code * WS;
#pragma unroll
                for (int k = 0; k < (NT + 3) / 4; k++) {
                    const float4 w4 = *reinterpret_cast<const float4 *>(wr + 4 * k);
                    const float w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (4 * k + j < NT) tap(4 * k + j, w[j]);
                }
            } else {
                const bool ok = u < d;
                const int code = ok ? code_of(u) : 0;
                const int ix = (int)(((float)code + 0.5f) * inv_sy);
                const int iy = code - ix * sy;
                const float4 wx4 = *reinterpret_cast<const float4 *>(ax_l + 4 * ix);
                const float4 wy4 = *reinterpret_cast<const float4 *>(ay_l + 8 * iy);
                const float4 wy5 = *reinterpret_cast<const float4 *>(ay_l + 8 * iy + 4);
                const float wx[4] = {wx4.x, wx4.y, wx4.z, wx4.w};
                float wy[8] = {wy4.x, wy4.y, wy4.z, wy4.w, wy5.x, wy5.y, wy5.z, wy5.w};
#pragma unroll
                for (int b = 0; b < TY; b++) wy[b] = ok ? wy[b] : 0.f;
#pragma unroll
                for (int b = 0; b < TY; b++)
#pragma unroll
                    for (int a = 0; a < TX; a++) tap(a + TX * b, wx[a] * wy[b]);   // == the level-0 offset table entry (bx[a]*by[b])
            }
            __builtin_amdgcn_sched_barrier(0);   // one edge's weights live at a time (register pressure)
        };
        if (ablate & 1) dmax = min(dmax, 1);     // (measurement only, DAGR_L0_ABLATE: phase 1 cut to one edge)
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (u < dmax) edge(u);               // group-uniform
        if (dmax > 8) {
            if (LEAN) {       // second batch of source rows, through the registers of the first
#pragma unroll
                for (int u = 8; u < 16; u++) {
                    const int src = (u < d) ? srcs[u] : nn;
                    if (CM) xv[u - 8] = *reinterpret_cast<const float4 *>(x + (size_t)src * ldx + 4 * q);
                    if (CE) xe[u - 8] = (q < CE) ? x[(size_t)src * ldx + CM + q] : 0.f;
                }
            }
#pragma unroll
            for (int u = 8; u < 16; u++)
                if (u < dmax) edge(u);
        }

        // ---- phase 2: out[16 nodes][16] = [A | root | skip] . Wpack on the matrix pipe, operands from registers
        f32x4_t o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
        const float *wb = w_l + l;
        int s = 0;
#define DAGR_STEP(aval)                                                                        \
    do {                                                                                       \
        if (s & 1) o1 = __builtin_amdgcn_mfma_f32_16x16x4f32((aval), wb[(s) * 64], o1, 0, 0, 0); \
        else o0 = __builtin_amdgcn_mfma_f32_16x16x4f32((aval), wb[(s) * 64], o0, 0, 0, 0);       \
        s++;                                                                                   \
    } while (0)
        if (ablate & 2) {                        // (measurement only: phase 2 cut to the root / skip steps)
            if (CM) {
#pragma unroll
                for (int t = 0; t < NT; t++) { o0[0] += acc[t][0][0] + acc[t][0][1]; o1[0] += acc[t][1][0] + acc[t][1][1]; }
            }
            if (CE) {
#pragma unroll
                for (int t = 0; t < NT; t++) o0[1] += acce[t];
            }
            s = S::kMain + S::kExtra;
        } else {
        if (CM) {
#pragma unroll
            for (int t = 0; t < NT; t++) {
                DAGR_STEP(acc[t][0][0]); DAGR_STEP(acc[t][0][1]); DAGR_STEP(acc[t][1][0]); DAGR_STEP(acc[t][1][1]);
            }
        }
        if (CE) {
#pragma unroll
            for (int t = 0; t < NT; t++) DAGR_STEP(acce[t]);
        }
        }
        if (CM) { DAGR_STEP(xr.x); DAGR_STEP(xr.y); DAGR_STEP(xr.z); DAGR_STEP(xr.w); }
        if (CE) DAGR_STEP(xre);
        if (S::SKF) { DAGR_STEP(xs.x); DAGR_STEP(xs.y); DAGR_STEP(xs.z); DAGR_STEP(xs.w); }
        if (S::SKE) DAGR_STEP(xse);
#undef DAGR_STEP
        // o[r] = out[node 4q + r][channel c]
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int m = n0 + 4 * q + r;
            if (m < n_end) {
                float v = (o0[r] + o1[r]) + my_shift;
                if (relu) v = fmaxf(v, 0.f);
                out[(size_t)m * ldo + c] = v;
            }
        }
    }
}

template <int CM, int CE, int CS, int TX, int TY, bool LEAN>
int launch_tiles_v(int64_t n_first, int64_t N, const int32_t *n_ptr, int rx, int ry, float den_x, float den_y, int win_x, int win_y, const int32_t *nbr_src,
                 const int16_t *nbr_code, const int32_t *deg, const float *x, int ldx, const float *xskip, int ldskip,
                 const float *wpack, const float *shift, int relu, float *out, int ldo, hipStream_t stream) {
    using S = L0Steps<CM, CE, CS, TX, TY>;
    constexpr int WS = (S::NT + 3) / 4 * 4 + 4;
    constexpr bool PRE = CM > 0;     // per-offset weight rows / per-axis tables (see the kernel)
    const size_t lds_bytes = ((size_t)S::N * 64 + (PRE ? ((size_t)(2 * rx + 1) * (2 * ry + 1) + 1) * WS
                                                       : (size_t)(2 * rx + 1) * 4 + (size_t)(2 * ry + 1) * 8)) * 4;
    DAGR_CHECK_ARG(lds_bytes <= 64 * 1024, "offset domain too large for the per-offset weight table");
    auto kern = k_conv_l0_tiles<CM, CE, CS, TX, TY, LEAN>;
    {
        static thread_local size_t set_for = 0;
        if (set_for < lds_bytes) {
            DAGR_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)lds_bytes));
            set_for = lds_bytes;
        }
    }
    const int64_t tiles = ceil_div(N, 16);
    const unsigned grid = round_grid8(persistent_grid(kern, kTileWaves * 64, lds_bytes, ceil_div(tiles, kTileWaves)));
    static const int ablate = (int)knob("DAGR_L0_ABLATE", 0);   // measurement build only (results are wrong when set)
    kern<<<grid, kTileWaves * 64, lds_bytes, stream>>>((int)n_first, (int)N, n_ptr, rx, ry, den_x, den_y, win_x, win_y, nbr_src, nbr_code, deg,
                                                       x, ldx, xskip, ldskip, wpack, shift, relu, out, ldo, ablate);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

template <int CM, int CE, int CS, int TX, int TY>
int launch_tiles(int64_t n_first, int64_t N, const int32_t *n_ptr, int rx, int ry, float den_x, float den_y, int win_x, int win_y,
                 const int32_t *nbr_src, const int16_t *nbr_code, const int32_t *deg, const float *x, int ldx,
                 const float *xskip, int ldskip, const float *wpack, const float *shift, int relu, float *out, int ldo,
                 hipStream_t stream) {
    // three waves per SIMD (LEAN: <= 168 registers, source rows in two batches of 8): -4 % on the 16 -> 16 + skip and
    // 3 -> 16 convs (0.179 -> 0.171, 0.090 -> 0.087 ms at 800 k nodes).  The 19 -> 16 conv (main block + extras) used to
    // spill at that budget (+7 %) and ran at two waves until the per-offset weight rows and the packed offset codes freed
    // the registers (166, no scratch).  PMC (profiles/r3_*_pmc_sq.csv): the SIMDs' issue slots are ~77 % busy at two waves
    // -- the kernel is bound by instruction issue (packed FMAs already), which is why a third wave buys so little.
    constexpr bool kLean = true;
    return launch_tiles_v<CM, CE, CS, TX, TY, kLean>(n_first, N, n_ptr, rx, ry, den_x, den_y, win_x, win_y, nbr_src, nbr_code,
                                                     deg, x, ldx, xskip, ldskip, wpack, shift, relu, out, ldo, stream);
}

}  // namespace
}  // namespace dagr

using namespace dagr;

extern "C" int dagr_spline_conv_l0_tiles_rows(int32_t cmain, int32_t cextra, int32_t cskip, int32_t win_x, int32_t tx,
                                              int32_t win_y, int32_t ty, int32_t rx, int32_t ry, float den_x, float den_y,
                                              int64_t first_node, int64_t N, int32_t K, const int32_t *nbr_src,
                                              const int16_t *nbr_code, const int32_t *deg, const float *x, int32_t ldx,
                                              const float *xskip, int32_t ldskip, const float *wpack, const float *shift,
                                              int32_t relu, float *out, int32_t ldo, const int32_t *n_ptr, void *stream_);

extern "C" int dagr_spline_conv_l0_tiles(int32_t cmain, int32_t cextra, int32_t cskip, int32_t win_x, int32_t tx,
                                         int32_t win_y, int32_t ty, int32_t rx, int32_t ry, float den_x, float den_y,
                                         int64_t N, int32_t K, const int32_t *nbr_src, const int16_t *nbr_code,
                                         const int32_t *deg, const float *x, int32_t ldx, const float *xskip,
                                         int32_t ldskip, const float *wpack, const float *shift, int32_t relu,
                                         float *out, int32_t ldo, const int32_t *n_ptr, void *stream_) {
    return dagr_spline_conv_l0_tiles_rows(cmain, cextra, cskip, win_x, tx, win_y, ty, rx, ry, den_x, den_y, 0, N, K, nbr_src,
                                          nbr_code, deg, x, ldx, xskip, ldskip, wpack, shift, relu, out, ldo, n_ptr, stream_);
}

extern "C" int dagr_spline_conv_l0_tiles_rows(int32_t cmain, int32_t cextra, int32_t cskip, int32_t win_x, int32_t tx,
                                              int32_t win_y, int32_t ty, int32_t rx, int32_t ry, float den_x, float den_y,
                                              int64_t first_node, int64_t N, int32_t K, const int32_t *nbr_src,
                                              const int16_t *nbr_code, const int32_t *deg, const float *x, int32_t ldx,
                                              const float *xskip, int32_t ldskip, const float *wpack, const float *shift,
                                              int32_t relu, float *out, int32_t ldo, const int32_t *n_ptr, void *stream_) {
    DAGR_CHECK_ARG(N >= 0 && first_node >= 0 && first_node + N < (1ll << 31), "bad node range");
    if (N == 0) return DAGR_OK;
    DAGR_CHECK_ARG(nbr_src && nbr_code && deg && x && wpack && shift && out, "NULL pointer");
    DAGR_CHECK_ARG(K == 16, "the tiled level-0 conv walks 16-entry neighbour lists");
    DAGR_CHECK_ARG(cskip == 0 || xskip, "xskip is NULL");
    DAGR_CHECK_ARG(rx >= 0 && ry >= 0 && den_x > 0 && den_y > 0 && win_x >= 0 && win_y >= 0 && win_x + tx <= 5 &&
                       win_y + ty <= 5, "bad offset domain / tap window");
    DAGR_CHECK_ARG(cmain == 0 || (ldx % 4 == 0 && ((uintptr_t)x % 16) == 0), "x rows must be 16-byte aligned");
    DAGR_CHECK_ARG(cskip < 16 || (ldskip % 4 == 0 && ((uintptr_t)xskip % 16) == 0), "xskip rows must be 16-byte aligned");
    DAGR_CHECK_ARG(((uintptr_t)nbr_src % 16) == 0 && ((uintptr_t)nbr_code % 8) == 0, "neighbour lists must be aligned");
    hipStream_t stream = (hipStream_t)stream_;
#define DAGR_TILES(CM_, CE_, CS_, TX_, TY_)                                                                        \
    if (cmain == CM_ && cextra == CE_ && cskip == CS_ && tx == TX_ && ty == TY_)                                   \
        return launch_tiles<CM_, CE_, CS_, TX_, TY_>(first_node, N, n_ptr, rx, ry, den_x, den_y, win_x, win_y, nbr_src, nbr_code, deg, x, \
                                                    ldx, xskip, ldskip, wpack, shift, relu, out, ldo, stream);
#define DAGR_TILES_WIN(TX_, TY_)                                                                   \
    DAGR_TILES(0, 3, 0, TX_, TY_)    /* events-only conv_block1.conv_block1: 3 -> 16 (net.py:75) */  \
    DAGR_TILES(16, 0, 3, TX_, TY_)   /* events-only conv_block1.conv_block2: 16 -> 16 + skip 3 */    \
    DAGR_TILES(16, 3, 0, TX_, TY_)   /* --use_image first conv: 16 image + 3 channels -> 16 */       \
    DAGR_TILES(16, 0, 19, TX_, TY_)  /* --use_image second conv: 16 -> 16 + skip 19 */
    DAGR_TILES_WIN(3, 3)
    DAGR_TILES_WIN(3, 5)
    DAGR_TILES_WIN(5, 3)
#undef DAGR_TILES_WIN
#undef DAGR_TILES
    set_error("dagr_spline_conv_l0_tiles: unsupported (cmain, cextra, cskip, tx, ty) combination");
    return DAGR_ERR_UNSUPPORTED;
}
