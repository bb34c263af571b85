// gemm.hip -- C[M,N] = act(A[M,K] . Wm[K,N] + bias[N]) with exact-fp32 MFMA (v_mfma_f32_16x16x4_f32).
//
// The pooled-level SplineConvs are contractions of the tap-aggregated rows A[n] (K = 26*Cin (+Cskip),
// up to 1682) with a packed weight matrix (N = 64 / 128): GEMM-shaped, so they run on the matrix
// cores.  The f32-input MFMA is bit-exact fp32 FMA at the vector rate (MI355X_MICROARCH.md), which keeps
// the 1e-4 parity bar with no precision trade.
//
// Shape of the problem: M is small (<= 2240*(B+1) rows, often a few hundred) and K is long, so what
// matters is (i) enough workgroups to occupy 256 CUs and (ii) a short dependent K-chain per wave.
// Block tile BM x 64 with BM = 16 (one accumulator per wave) or 32 (two accumulators, picked when
// M/32 already yields >= 2 blocks per CU); 16 waves: wave w of a quad owns output columns
// [16w, 16w+16) and the 4 quads split K (chain = K/16 MFMAs per wave, four load streams in flight).  Operands are staged in LDS as [k][m] / [k][n] with row strides = 16 mod 32
// dwords so that a lane's MFMA operand (i = lane&15, k = lane>>4) is a conflict-free read; the next
// 32-wide K stage is prefetched into registers while the current one is multiplied.  M is bounded by a
// device-side count: no host sync.  Summation order is fixed: results are run-to-run identical.
#include <algorithm>

#include "common.hpp"
#include "pool_common.hpp"

#ifndef DAGR_TRACE          // tools/microbench/conv_trace.hip defines it to time the stages of k_conv_fused
#define DAGR_TRACE(i)
#endif

namespace dagr {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int NB = 64, KB = 32, KSPLIT = 4;
constexpr int WS_STRIDE = NB + 16;  // 80 = 16 (mod 32)
constexpr int kGemmThreads = kBlock * KSPLIT;

template <int BM>
__global__ __launch_bounds__(kGemmThreads) void k_gemm_mfma(const int32_t *__restrict__ m_ptr, int m_max,
                                                     const float *__restrict__ A, int lda,
                                                     const float *__restrict__ Wm, int ldw,
                                                     const float *__restrict__ bias, float *__restrict__ C, int ldc,
                                                     int K, int N, int relu) {
    constexpr int RT = BM / 16;            // 16-row tiles per wave
    constexpr int AS_STRIDE = BM + 16;     // 32 or 48 = 0/16 (mod 32) alternating with k
    constexpr int A_THREADS = BM * (KB / 4);
    // the 4 wave quads of a block each own a 32-wide slice of every 128-wide K step (their K-chains and
    // their global-load latencies overlap); partial tiles are summed through LDS in a fixed order
    constexpr int kStage = KSPLIT * KB * (AS_STRIDE + WS_STRIDE);
    constexpr int kReduce = KSPLIT * BM * NB;
    __shared__ __align__(16) float lds[kStage > kReduce ? kStage : kReduce];
    const int M = m_ptr ? min(*m_ptr, m_max) : m_max;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * NB;
    if (m0 >= M) return;
    const int ks = threadIdx.x >> 8;
    float *As = lds + ks * KB * AS_STRIDE;
    float *Ws = lds + KSPLIT * KB * AS_STRIDE + ks * KB * WS_STRIDE;
    const int t = threadIdx.x & 255, l = t & 63, w = t >> 6;
    // staging roles
    const int a_row = t >> 3, a_kq = (t & 7) * 4;    // A: BM rows x 32 k, 4 consecutive k per thread
    const bool a_thr = t < A_THREADS;
    const bool a_ok = a_thr && (m0 + a_row) < M;
    const float *a_src = A + (size_t)(m0 + (a_thr ? a_row : 0)) * lda + a_kq;
    const int w_k = t >> 4, w_nq = (t & 15) * 4;     // W: 32 k x 64 n, two float4 per thread (k, k+16)
    const bool w_ok = (n0 + w_nq + 4) <= ldw;        // weight rows are zero-padded to a multiple of 8 columns
    float4 ra, rw0, rw1;
    auto load_tile = [&](int k0) {
        ra = rw0 = rw1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a_ok) {
            if (k0 + a_kq + 4 <= K) {
                ra = *reinterpret_cast<const float4 *>(a_src + k0);
            } else if (k0 + a_kq < K) {
                float tmp[4];
#pragma unroll
                for (int j = 0; j < 4; j++) tmp[j] = (k0 + a_kq + j < K) ? a_src[k0 + j] : 0.f;
                ra = make_float4(tmp[0], tmp[1], tmp[2], tmp[3]);
            }
        }
        if (w_ok) {
            if (k0 + w_k < K) rw0 = *reinterpret_cast<const float4 *>(Wm + (size_t)(k0 + w_k) * ldw + n0 + w_nq);
            if (k0 + w_k + 16 < K)
                rw1 = *reinterpret_cast<const float4 *>(Wm + (size_t)(k0 + w_k + 16) * ldw + n0 + w_nq);
        }
    };
    auto store_tile = [&]() {
        if (a_thr) {
            float *as = As + a_kq * AS_STRIDE + a_row;
            as[0 * AS_STRIDE] = ra.x; as[1 * AS_STRIDE] = ra.y; as[2 * AS_STRIDE] = ra.z; as[3 * AS_STRIDE] = ra.w;
        }
        *reinterpret_cast<float4 *>(Ws + w_k * WS_STRIDE + w_nq) = rw0;
        *reinterpret_cast<float4 *>(Ws + (w_k + 16) * WS_STRIDE + w_nq) = rw1;
    };
    f32x4 acc[RT];
#pragma unroll
    for (int r = 0; r < RT; r++) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int KSTEP = KB * KSPLIT;
    load_tile(ks * KB);
    store_tile();
    __syncthreads();
    const float *a_rd = As + (l >> 4) * AS_STRIDE + (l & 15);
    const float *w_rd = Ws + (l >> 4) * WS_STRIDE + w * 16 + (l & 15);
    for (int k0 = 0; k0 < K; k0 += KSTEP) {
        const bool more = (k0 + KSTEP) < K;
        if (more) load_tile(k0 + KSTEP + ks * KB);
#pragma unroll
        for (int kk = 0; kk < KB; kk += 4) {
            const float b = w_rd[kk * WS_STRIDE];
#pragma unroll
            for (int r = 0; r < RT; r++) {
                const float a = a_rd[kk * AS_STRIDE + r * 16];
                acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[r], 0, 0, 0);
            }
        }
        __syncthreads();
        if (more) {
            store_tile();
            __syncthreads();
        }
    }
    // split-K reduction (staging buffers are dead after the last barrier above)
    float *red = lds + ks * BM * NB;
#pragma unroll
    for (int r = 0; r < RT; r++)
#pragma unroll
        for (int q = 0; q < 4; q++) red[(r * 16 + (l >> 4) * 4 + q) * NB + w * 16 + (l & 15)] = acc[r][q];
    __syncthreads();
    for (int idx = threadIdx.x; idx < BM * NB; idx += kGemmThreads) {
        const int row = m0 + idx / NB, col = n0 + idx % NB;
        if (row < M && col < N) {
            float v = ((lds[idx] + lds[BM * NB + idx]) + lds[2 * BM * NB + idx]) + lds[3 * BM * NB + idx];
            v += bias ? bias[col] : 0.f;
            if (relu) v = fmaxf(v, 0.f);
            C[(size_t)row * ldc + col] = v;
        }
    }
}

}  // namespace

hipError_t launch_gemm_mfma(const int32_t *m_ptr, int m_max, const float *A, int lda, const float *Wm, int ldw,
                            const float *bias, float *C, int ldc, int K, int N, int relu, hipStream_t stream) {
    const unsigned gy = (unsigned)ceil_div(N, NB);
    if (ceil_div(m_max, 32) * gy >= 512) {
        dim3 grid((unsigned)ceil_div(m_max, 32), gy);
        k_gemm_mfma<32><<<grid, kGemmThreads, 0, stream>>>(m_ptr, m_max, A, lda, Wm, ldw, bias, C, ldc, K, N, relu);
    } else {
        dim3 grid((unsigned)ceil_div(m_max, 16), gy);
        k_gemm_mfma<16><<<grid, kGemmThreads, 0, stream>>>(m_ptr, m_max, A, lda, Wm, ldw, bias, C, ldc, K, N, relu);
    }
    return hipGetLastError();
}

}  // namespace dagr

// ---------------------------------------------------------------------------------------------
// Fused pooled-level SplineConv: tap aggregation + contraction in one launch.
// A workgroup owns 16 destination nodes.  Phase A: wave w aggregates node w's taps straight into the block's
// LDS A-tile [16][K] (same arithmetic and order as k_tap_aggregate, spline_conv.hip): the [T, 26*Cin] matrix
// never goes to HBM and 16 of the 32 launches per step disappear.  Phase B: the 16 x K x N contraction with
// v_mfma_f32_16x16x4_f32, A operands read from the tile (row stride = 2 mod 32: conflict-free), weights read
// straight from L2 (no reuse inside a block), 4 wave quads split K, fixed-order reduction.  Column blocks
// of 64 are looped inside the same workgroup.
//
// Measured on MI355X (B = 8 x 100k events, 16 fused convs per step, 415 us): removing the gathers leaves 238 us,
// removing the MFMA loop 292 us, both 114 us -- the phases simply add up because the 113-KiB tile allows one
// workgroup per CU.  Alternatives tried and dropped: (i) K cut into two passes over a 59-KiB tile with two
// workgroups per CU: 505 us (the scatter is VALU/LDS work that doubles with the passes); (ii) a persistent
// workgroup with a 3-deep register pipeline for the rowptr -> col/code -> x load chain: 439 us; (iii) the spline
// basis evaluated once per node with one edge per lane + branch-free tap redirection: 440 us.
namespace dagr {
namespace {

struct AxisF {
    int k0, k1;
    float b0, b1;
};
// launch (A) of the pooling step that consumes this conv's output, fused into the conv (dagr_spline_conv_fused_pool): the
// epilogue merges every output element into its cluster's accumulator, wave w does the bookkeeping and the in-edges of
// node w -- exactly k_pool_accumulate's work (pooling.hip), on the values the epilogue holds in registers.
struct PoolFuse {
    int on;
    dagr_pool_desc d;
    PoolWs ws;
    const float *pos;
    const int32_t *batch;
    int32_t *cluster_raw_out;
};
// Extra convs of a multi-job launch (dagr_spline_conv_fused_multi / _pair): blockIdx.z = 1 + index.  Jobs are independent
// convs -- other graphs, row shapes and widths allowed -- that become ready at the same point of a window (a head scale's
// convs beside the next level's): on levels of <= 1 260 rows a launch costs what a conv costs, and the chip has the CUs.
constexpr int kMaxExtraJobs = 3;
struct ConvJob {
    const int32_t *n_ptr, *rowptr, *col, *code;
    const float *x, *xskip, *Wq, *bias;
    float *C;
    int n_max, ldx, cin, ldskip, cskip, rx, ry, ldc, N, relu, KP, NC, gy, tp;
    float den_x, den_y;
};
struct ConvJobs {
    int extra;
    ConvJob j[kMaxExtraJobs];
};
// the parameters of job blockIdx.z replace the kernel's own (job 0); false: this workgroup lies outside the job's grid
#define DAGR_CONV_PICK_JOB(TP_STMT)                                                                                      \
    if (blockIdx.z > 0) {                                                                                                \
        const ConvJob &jb = jobs.j[blockIdx.z - 1];                                                                      \
        n_ptr = jb.n_ptr; n_max = jb.n_max; rowptr = jb.rowptr; col = jb.col; code = jb.code; x = jb.x; ldx = jb.ldx;    \
        cin = jb.cin; xskip = jb.xskip; ldskip = jb.ldskip; cskip = jb.cskip; rx = jb.rx; ry = jb.ry; den_x = jb.den_x;  \
        den_y = jb.den_y; Wq = jb.Wq; bias = jb.bias; C = jb.C; ldc = jb.ldc; N = jb.N; relu = jb.relu; KP = jb.KP;      \
        NC = jb.NC; TP_STMT;                                                                                             \
        if ((int)blockIdx.y >= jb.gy || (int)blockIdx.x * 16 >= jb.n_max) return;                                        \
    } else if ((int)blockIdx.y >= gy0 || (int)blockIdx.x * 16 >= n_max) return;
__device__ __forceinline__ AxisF spline_axis_f(int idx, int r, float den) {   // == spline_axis (spline_conv.hip)
    const float pseudo = (float)(idx - r) / den + 0.5f;
    const float v = pseudo * 4.0f;
    const float fl = floorf(v);
    const float frac = v - fl;
    AxisF a;
    const int f = (int)fl;
    a.k0 = f % 5;
    a.k1 = (f + 1) % 5;
    a.b0 = ((1.0f - frac) - 0.0f) + (2.0f * frac) * 0.0f;
    a.b1 = ((1.0f - frac) - 1.0f) + (2.0f * frac) * 1.0f;
    return a;
}

// U = weight groups in flight per wave.  4: the form tuned for the small levels (72 registers, one 16-wave workgroup per
// CU).  3: 62 registers = 8 waves per SIMD, so that TWO workgroups share a CU when their tiles fit (<= 80 KB each): on
// levels that take several rounds of workgroups (level 1 - 2 of a B = 8 batch) one's edge walk overlaps the other's
// contraction, which a single resident workgroup runs back to back.
template <int U>
__global__ __launch_bounds__(kGemmThreads) __attribute__((amdgpu_waves_per_eu(U == 4 ? 4 : 8, U == 4 ? 7 : 8))) void k_conv_fused(
    const int32_t *__restrict__ n_ptr, int n_max, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const int32_t *__restrict__ code, const float *__restrict__ x, int ldx, int cin, const float *__restrict__ xskip,
    int ldskip, int cskip, int rx, int ry, float den_x, float den_y, const float *__restrict__ Wq,
    const float *__restrict__ bias, float *__restrict__ C, int ldc, int N, int relu, int KP, int NC, int gy0, ConvJobs jobs,
    PoolFuse pf) {
    DAGR_CONV_PICK_JOB((void)0)
    if (blockIdx.z > 0) pf.on = 0;                   // the fused pooling merge belongs to job 0
    extern __shared__ __align__(16) float fl[];
    __shared__ int s_raw[16];                        // fused pooling: cluster (table slot) of the tile's nodes, -1 = none
    const int K = 26 * cin + cskip;
    float *At = fl;                                  // [16][KP], columns K..KP-1 zero
    float *red = fl + 16 * KP;                       // [KSPLIT][16][NB] split-K partials
    const int m0 = blockIdx.x * 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: loop bounds below stay in SGPRs
    DAGR_TRACE(0);
    // rowptr has n_max + 1 entries: read it before the device-side bound is known (one latency, not two)
    const int n_spec = min(m0 + wv, n_max - 1);
    const int e0 = rowptr[n_spec], e1s = rowptr[n_spec + 1];
    const int M = n_ptr ? min(*n_ptr, n_max) : n_max;
    if (m0 >= M) return;
    int my_raw = -1;
    if (pf.on) {
        const int n = m0 + wv;
        if (n < M) {
            bool ok;
            my_raw = cluster_raw(pf.pos[3 * n], pf.pos[3 * n + 1], pf.pos[3 * n + 2], pf.batch[n], pf.d, ok);
            if (!ok) {
                my_raw = -1;
                if (lane == 0 && blockIdx.y == 0) { atomicOr(&pf.ws.status[0], 1); pf.cluster_raw_out[n] = -1; }
            }
        }
        if (lane == 0) s_raw[wv] = my_raw;
    }
    DAGR_TRACE(1);
    // ---- phase A: wave wv aggregates node m0 + wv (edge order and arithmetic of k_tap_aggregate)
    {
        const int n = m0 + wv;
        float *row = At + wv * KP;
        for (int i = lane; i < KP; i += 64) row[i] = 0.0f;
        DAGR_TRACE(7);
        if (n < M) {
            const int ne = e1s - e0;
            const float *xn = x + (size_t)n * ldx;
            for (int i = lane; i < cin; i += 64) row[25 * cin + i] = xn[i];
            if (cskip > 0) {
                const float *sn = xskip + (size_t)n * ldskip;
                for (int i = lane; i < cskip; i += 64) row[26 * cin + i] = sn[i];
            }
            DAGR_TRACE(8);
            for (int base = 0; base < ne; base += 64) {
                const int cnt = min(64, ne - base);
                int my_src = 0, my_cd = 0;
                if (lane < cnt) {
                    my_src = col[e0 + base + lane];
                    my_cd = code[e0 + base + lane];
                }
                DAGR_TRACE(2);
                for (int c0 = 0; c0 < cin; c0 += 64) {
                    const int ch = c0 + lane;
                    const bool ch_ok = ch < cin;
                    // A remainder of a few channels beyond this chunk (every pooled level carries C + 2 = 66 / 130 inputs:
                    // the features and pos_xy) rides along on the first lanes instead of costing a second walk over the
                    // edges with two active lanes -- same per-channel arithmetic and order, half the dependent chain.
                    const int rem = cin - (c0 + 64);
                    const bool ride = rem > 0 && rem <= 8;       // wave-uniform
                    const int ch2 = c0 + 64 + lane;
                    const bool ch2_ok = ride && lane < rem;
                    constexpr int UA = 8;
                    for (int j = 0; j < cnt; j += UA) {
                        float v[UA], v2[UA];
                        int cd[UA];
#pragma unroll
                        for (int u = 0; u < UA; u++) {   // UA gathers in flight
                            const int jj = min(j + u, cnt - 1);
                            const int src = __builtin_amdgcn_readlane(my_src, jj);
                            cd[u] = __builtin_amdgcn_readlane(my_cd, jj);
                            v[u] = ch_ok ? x[(size_t)src * ldx + ch] : 0.0f;
                            v2[u] = ch2_ok ? x[(size_t)src * ldx + ch2] : 0.0f;
                        }
                        DAGR_TRACE(9);
#pragma unroll
                        for (int u = 0; u < UA; u++) {
                            if (j + u < cnt && ch_ok) {
                                const AxisF ax = spline_axis_f(cd[u] & 0xffff, rx, den_x);
                                const AxisF ay = spline_axis_f(cd[u] >> 16, ry, den_y);
                                const float b00 = ax.b0 * ay.b0, b10 = ax.b1 * ay.b0, b01 = ax.b0 * ay.b1,
                                            b11 = ax.b1 * ay.b1;
                                float *a00 = row + (ax.k0 + 5 * ay.k0) * cin + ch;
                                float *a10 = row + (ax.k1 + 5 * ay.k0) * cin + ch;
                                float *a01 = row + (ax.k0 + 5 * ay.k1) * cin + ch;
                                float *a11 = row + (ax.k1 + 5 * ay.k1) * cin + ch;
                                // the four taps of one edge are distinct (k0 != k1 on both axes): read all, then write
                                const float o00 = *a00, o10 = *a10, o01 = *a01, o11 = *a11;
                                *a00 = o00 + b00 * v[u];
                                *a10 = o10 + b10 * v[u];
                                *a01 = o01 + b01 * v[u];
                                *a11 = o11 + b11 * v[u];
                                if (ch2_ok) {
                                    const float p00 = a00[64], p10 = a10[64], p01 = a01[64], p11 = a11[64];
                                    a00[64] = p00 + b00 * v2[u];
                                    a10[64] = p10 + b10 * v2[u];
                                    a01[64] = p01 + b01 * v2[u];
                                    a11[64] = p11 + b11 * v2[u];
                                }
                            }
                        }
                    }
                    if (ride) break;
                }
            }
        }
    }
    // fused pooling: the node's bookkeeping and in-edges (one column workgroup per node tile does it)
    if (pf.on && my_raw >= 0 && blockIdx.y == 0)
        pool_merge_node(pf.d, pf.ws, m0 + wv, my_raw, pf.pos, pf.batch, col, e0, e1s, lane, 64, pf.cluster_raw_out);
    DAGR_TRACE(3);
    __syncthreads();
    DAGR_TRACE(4);
    // ---- phase B: [16 x K] . [K x N].  Wave (ks, w): K quarter ks, 16-column tile w of a 64-column block.
    // No weight staging: every weight element is used by exactly one wave of the block, so the B operands
    // come straight from L2 in the host-packed MFMA operand order Wq[col tile][k group of 16][lane][4]
    // (element j of lane l = W[16 g + 4 j + (l >> 4)][16 c + (l & 15)]): one 1-KiB fully coalesced wave
    // load feeds 4 MFMAs, U of them are in flight per wave, and the loop has no barrier.
    // NC = column tiles per workgroup: 4 (one workgroup per 16 nodes, K split 4 ways) or, for levels with too few
    // nodes to fill the chip, 1 (gridDim.y workgroups per 16 nodes, K split 16 ways).
    const int ksplit = 16 / NC, NBc = 16 * NC;
    const int ks = wv / NC, w = wv % NC;
    const int kk = lane >> 4, nn = lane & 15;
    const int G = (K + 15) / 16, Gq = (G + ksplit - 1) / ksplit;
    const int gbeg = ks * Gq, gend = min(G, gbeg + Gq);
    const int nfull = max(0, gend - gbeg) / U;              // wave-uniform (wv is scalar)
    const float *a_rd = At + nn * KP + kk;
    const int n_first = (NC == 4) ? 0 : (int)blockIdx.y * NBc;
    const int n_last = (NC == 4) ? N : min(N, n_first + NBc);
    for (int n0 = n_first; n0 < n_last; n0 += NBc) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        if (n0 + w * 16 < N) {
            const float4 *wq = reinterpret_cast<const float4 *>(Wq) + ((size_t)(n0 / 16 + w) * G) * 64 + lane;
            auto load_b = [&](float4(&dst)[U], int g0) {
#pragma unroll
                for (int u = 0; u < U; u++) dst[u] = wq[(size_t)(g0 + u) * 64];
            };
            // U groups = 16 k-steps: all A operands are read from the tile first, then the MFMAs run back to back
            auto mac = [&](const float4(&b)[U], int g0) {
                float a[U][4];
                const float *ap = a_rd + 16 * g0;
#pragma unroll
                for (int u = 0; u < U; u++)
#pragma unroll
                    for (int j = 0; j < 4; j++) a[u][j] = ap[16 * u + 4 * j];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][0], b[u].x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][1], b[u].y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][2], b[u].z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][3], b[u].w, acc1, 0, 0, 0);
                }
            };
            float4 b0[U], b1[U];
            int g = gbeg;
            if (nfull > 0) load_b(b0, g);
            for (int it = 0; it < nfull; it += 2) {     // ping-pong: no register copies between batches
                if (it + 1 < nfull) load_b(b1, g + U);
                mac(b0, g);
                if (it + 1 < nfull) {
                    if (it + 2 < nfull) load_b(b0, g + 2 * U);
                    mac(b1, g + U);
                }
                g += 2 * U;
            }
            DAGR_TRACE(10);
            {   // < U left-over groups: all of their operands are requested before the first is used (on the small
                // levels K / 16 splits into 7 groups per wave: one full batch + 3 left-overs, which used to be three
                // dependent load -> MFMA round trips)
                const int g0 = gbeg + nfull * U, rem = gend - g0;
                float4 bt[U - 1];
#pragma unroll
                for (int r = 0; r < U - 1; r++)
                    bt[r] = r < rem ? wq[(size_t)(g0 + r) * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int r = 0; r < U - 1; r++) {
                    if (r < rem) {
                        const float *ap = a_rd + 16 * (g0 + r);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[0], bt[r].x, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4], bt[r].y, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[8], bt[r].z, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[12], bt[r].w, acc1, 0, 0, 0);
                    }
                }
            }
        }
        const f32x4 acc = {acc0[0] + acc1[0], acc0[1] + acc1[1], acc0[2] + acc1[2], acc0[3] + acc1[3]};
        DAGR_TRACE(5);
        float *rd = red + ks * 16 * NBc;
#pragma unroll
        for (int q = 0; q < 4; q++) rd[(kk * 4 + q) * NBc + w * 16 + nn] = acc[q];
        __syncthreads();
        DAGR_TRACE(11);
        if (tid < 16 * NBc) {
            const int idx = tid;
            const int orow = m0 + idx / NBc, ocol = n0 + idx % NBc;
            if (orow < M && ocol < N) {
                float v = red[idx];
                for (int s = 1; s < ksplit; s++) v += red[s * 16 * NBc + idx];   // fixed order
                v += bias ? bias[ocol] : 0.f;
                if (relu) v = fmaxf(v, 0.f);
                C[(size_t)orow * ldc + ocol] = v;
                if (pf.on) {
                    const int raw = s_raw[idx / NBc];
                    if (raw >= 0 && ocol < pf.d.channels) {
                        long long *acc = pf.ws.xacc + (size_t)raw * pf.d.channels + ocol;
                        if (pf.d.aggr == 0) atomicMax(reinterpret_cast<int *>(acc), enc_f(v));
                        else atomicAdd(reinterpret_cast<unsigned long long *>(acc),
                                       (unsigned long long)(long long)llrint((double)v * kFeatScale));
                    }
                }
            }
        }
        __syncthreads();
        DAGR_TRACE(6);
    }
}

// ---------------------------------------------------------------------------------------------
// Narrow inputs on large levels: level 1's first conv of a B = 8 batch is 18 -> 64 on ~18 k nodes.  In k_conv_fused a wave
// aggregates ONE node with one lane per input channel -- 18 of 64 lanes -- and every 16-node workgroup reads the whole
// weight matrix from L2.  Here a workgroup owns 48 nodes (three 16-row MFMA tiles): phase A puts THREE nodes on a wave
// (21 lanes each, every lane walks its own node's edge list: the loads of a node's 21 lanes coalesce into one request), and
// phase B uses every weight fragment for three row tiles.  K = 26 cin <= 546 keeps the 48-row A tile (+ the split-K
// partials) at 142 KB.  Same arithmetic and order per output element as k_conv_fused (edge order, tap order, k-split of 4,
// fixed-order reduction): bit-identical results.
// RT = 16-row tiles per workgroup.  3: the form described above.  5 (N <= 64): the split-K partials alias the A tile (it is
// dead once every wave has read it), so an 80-node tile fits the LDS (154 KB) and the level of a B = 8 batch is ONE round
// of workgroups (252 on 256 CUs) instead of one and a half; a wave then aggregates its five nodes in two passes of three.
// Measured (L1c1 18 -> 64, 17 922 nodes): 16-node tiles 43.0 us, RT = 3 34.4, RT = 2 with two workgroups per CU 35.9.
constexpr int kNarrowSub = 21;        // lanes per node in phase A
template <int kNarrowRT, bool ALIAS>
__global__ __launch_bounds__(kGemmThreads) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_conv_fused_narrow(
    const int32_t *__restrict__ n_ptr, int n_max, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const int32_t *__restrict__ code, const float *__restrict__ x, int ldx, int cin, int rx, int ry, float den_x, float den_y,
    const float *__restrict__ Wq, const float *__restrict__ bias, float *__restrict__ C, int ldc, int N, int relu, int KP) {
    constexpr int ROWS = 16 * kNarrowRT;
    extern __shared__ __align__(16) float fl[];
    const int K = 26 * cin;
    float *At = fl;                                  // [ROWS][KP], columns K..KP-1 zero
    float *red = ALIAS ? fl : fl + ROWS * KP;        // [KSPLIT][ROWS][NB] split-K partials
    const int m0 = blockIdx.x * ROWS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = n_ptr ? min(*n_ptr, n_max) : n_max;
    if (m0 >= M) return;
    // ---- phase A: wave wv aggregates nodes m0 + RT wv + {0 .. RT-1}, three at a time; lane = (node of the pass, input channel)
    {
        const int sub = lane / kNarrowSub, ch = lane - kNarrowSub * sub;
#pragma unroll
        for (int r3 = 0; r3 < kNarrowRT; r3++) {
            float *zr = At + (kNarrowRT * wv + r3) * KP;
            for (int i = lane; i < KP; i += 64) zr[i] = 0.0f;
        }
        __builtin_amdgcn_wave_barrier();
        for (int p0 = 0; p0 < kNarrowRT; p0 += 3) {
            const int slot = p0 + sub;                   // this lane's node among the wave's RT
            const int n = m0 + kNarrowRT * wv + slot;
            const bool node_ok = sub < 3 && slot < kNarrowRT && n < M;
            const bool act = node_ok && ch < cin;
            float *row = At + (kNarrowRT * wv + min(slot, kNarrowRT - 1)) * KP;
            int e0 = 0, ne = 0;
            if (node_ok) {
                e0 = rowptr[n];
                ne = rowptr[n + 1] - e0;
            }
            if (act) row[25 * cin + ch] = x[(size_t)n * ldx + ch];
            constexpr int UA = 4;
            for (int j = 0; __any(j < ne); j += UA) {
                int src[UA], cd[UA];
                float v[UA];
#pragma unroll
                for (int u = 0; u < UA; u++) {           // a node's lanes read the same two words: one request each
                    const bool ok = j + u < ne;
                    src[u] = ok ? col[e0 + j + u] : 0;
                    cd[u] = ok ? code[e0 + j + u] : 0;
                }
#pragma unroll
                for (int u = 0; u < UA; u++) v[u] = (act && j + u < ne) ? x[(size_t)src[u] * ldx + ch] : 0.0f;
#pragma unroll
                for (int u = 0; u < UA; u++) {
                    if (act && j + u < ne) {
                        const AxisF ax = spline_axis_f(cd[u] & 0xffff, rx, den_x);
                        const AxisF ay = spline_axis_f(cd[u] >> 16, ry, den_y);
                        const float b00 = ax.b0 * ay.b0, b10 = ax.b1 * ay.b0, b01 = ax.b0 * ay.b1, b11 = ax.b1 * ay.b1;
                        float *a00 = row + (ax.k0 + 5 * ay.k0) * cin + ch;
                        float *a10 = row + (ax.k1 + 5 * ay.k0) * cin + ch;
                        float *a01 = row + (ax.k0 + 5 * ay.k1) * cin + ch;
                        float *a11 = row + (ax.k1 + 5 * ay.k1) * cin + ch;
                        // the four taps of one edge are distinct (k0 != k1 on both axes): read all, then write
                        const float o00 = *a00, o10 = *a10, o01 = *a01, o11 = *a11;
                        *a00 = o00 + b00 * v[u];
                        *a10 = o10 + b10 * v[u];
                        *a01 = o01 + b01 * v[u];
                        *a11 = o11 + b11 * v[u];
                    }
                }
            }
        }
    }
    __syncthreads();
    // ---- phase B: [48 x K] . [K x N]; wave (ks, w): K quarter ks, 16-column tile w of a 64-column block, three row tiles
    const int ks = wv >> 2, w = wv & 3;
    const int kk = lane >> 4, nn = lane & 15;
    const int G = (K + 15) / 16, Gq = (G + KSPLIT - 1) / KSPLIT;
    const int gbeg = ks * Gq, gend = min(G, gbeg + Gq);
    for (int n0 = 0; n0 < N; n0 += NB) {
        f32x4 acc[kNarrowRT][2];
#pragma unroll
        for (int rt = 0; rt < kNarrowRT; rt++) { acc[rt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[rt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        if (n0 + w * 16 < N) {
            const float4 *wq = reinterpret_cast<const float4 *>(Wq) + ((size_t)(n0 / 16 + w) * G) * 64 + lane;
            constexpr int U = 2;                     // weight groups in flight
            for (int g0 = gbeg; g0 < gend; g0 += U) {
                float4 b[U];
#pragma unroll
                for (int u = 0; u < U; u++)
                    b[u] = g0 + u < gend ? wq[(size_t)(g0 + u) * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < U; u++) {
                    if (g0 + u < gend) {             // (wave-uniform)
#pragma unroll
                        for (int rt = 0; rt < kNarrowRT; rt++) {
                            const float *ap = At + (16 * rt + nn) * KP + kk + 16 * (g0 + u);
                            // (two accumulators per row tile, alternating as in k_conv_fused: same k-order per output)
                            acc[rt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[0], b[u].x, acc[rt][0], 0, 0, 0);
                            acc[rt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4], b[u].y, acc[rt][1], 0, 0, 0);
                            acc[rt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[8], b[u].z, acc[rt][0], 0, 0, 0);
                            acc[rt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[12], b[u].w, acc[rt][1], 0, 0, 0);
                        }
                    }
                }
            }
        }
        if (ALIAS) __syncthreads();              // every wave is done with the A tile before the partials overwrite it
        float *rd = red + ks * ROWS * NB;
#pragma unroll
        for (int rt = 0; rt < kNarrowRT; rt++)
#pragma unroll
            for (int q = 0; q < 4; q++) rd[(16 * rt + kk * 4 + q) * NB + w * 16 + nn] = acc[rt][0][q] + acc[rt][1][q];
        __syncthreads();
        for (int idx = tid; idx < ROWS * NB; idx += kGemmThreads) {
            const int orow = m0 + idx / NB, ocol = n0 + idx % NB;
            if (orow < M && ocol < N) {
                float v = red[idx];
                for (int s = 1; s < KSPLIT; s++) v += red[s * ROWS * NB + idx];   // fixed order
                v += bias ? bias[ocol] : 0.f;
                if (relu) v = fmaxf(v, 0.f);
                C[(size_t)orow * ldc + ocol] = v;
            }
        }
        __syncthreads();
    }
}

// Tap range of a pass: K is cut at tap boundaries into P = ceil(25 / tp) passes of tp taps (tp * cin a multiple of 16,
// so that a pass starts on a weight group); the pass that holds tap 24 also holds the root and skip columns.  tp >= 25:
// the whole row in one pass (every level of dagr-s / dagr-n).  Wider rows (26 * cin + cskip > ~2270 floats: dagr-m's
// levels, every head conv) used to go through HBM as a [T, K] matrix and a second launch.
__global__ __launch_bounds__(kGemmThreads) void k_conv_fused_mp(
    const int32_t *__restrict__ n_ptr, int n_max, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const int32_t *__restrict__ code, const float *__restrict__ x, int ldx, int cin, const float *__restrict__ xskip,
    int ldskip, int cskip, int rx, int ry, float den_x, float den_y, const float *__restrict__ Wq,
    const float *__restrict__ bias, float *__restrict__ C, int ldc, int N, int relu, int KP, int NC, int tp, int gy0,
    ConvJobs jobs) {
    DAGR_CONV_PICK_JOB(tp = jb.tp)
    extern __shared__ __align__(16) float fl[];
    const int K = 26 * cin + cskip;
    float *At = fl;                                  // [16][KP]: the pass's columns, zero behind them
    float *red = fl + 16 * KP;                       // [KSPLIT][16][NB] split-K partials
    const int m0 = blockIdx.x * 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: loop bounds below stay in SGPRs
    constexpr int U = 4;
    // rowptr has n_max + 1 entries: read it before the device-side bound is known (one latency, not two)
    const int n_spec = min(m0 + wv, n_max - 1);
    const int e0 = rowptr[n_spec], e1s = rowptr[n_spec + 1];
    const int M = n_ptr ? min(*n_ptr, n_max) : n_max;
    // Phase-B roles are fixed by the thread id, so the first weight batch of this wave and the bias of this thread's
    // output column are requested now: they arrive while phase A runs (a small level is one latency chain per launch,
    // and this takes the weight round trip and the bias round trip off it).
    // NC = column tiles per workgroup: 4 (K split 4 ways) or, for levels with too few nodes to fill the chip, fewer
    // (more workgroups per 16 nodes, K split 16 / NC ways).  loop_cols: one workgroup per 16 nodes walks all 64-column
    // blocks over the same tile (single pass only); else blockIdx.y picks the block.
    const int ksplit = 16 / NC, NBc = 16 * NC;
    const int ks = wv / NC, w = wv % NC;
    const int kk = lane >> 4, nn = lane & 15;
    const int G = (K + 15) / 16;
    const int P = (25 + tp - 1) / tp;
    const int n_first = (int)blockIdx.y * NBc;
    const int n_last = min(N, n_first + NBc);
    // groups [g_lo, g_hi) of pass p, this wave's share [gbeg, gend) of them
    auto pass_groups = [&](int p, int &g_lo, int &gbeg, int &gend) {
        g_lo = (p * tp * cin) >> 4;
        const int g_hi = (p == P - 1) ? G : ((p + 1) * tp * cin) >> 4;
        const int Gq = (g_hi - g_lo + ksplit - 1) / ksplit;
        gbeg = g_lo + ks * Gq;
        gend = min(g_hi, gbeg + Gq);
    };
    float4 b0[U], b1[U];
    {
        int g_lo, gbeg, gend;
        pass_groups(0, g_lo, gbeg, gend);
        const float4 *wq = reinterpret_cast<const float4 *>(Wq) + ((size_t)(n_first / 16 + w) * G) * 64 + lane;
        if (n_first + w * 16 < N && gend - gbeg >= U) {
#pragma unroll
            for (int u = 0; u < U; u++) b0[u] = wq[(size_t)(gbeg + u) * 64];
        }
    }
    float bias_first = 0.0f;
    if (bias && tid < 16 * NBc && n_first + tid % NBc < N) bias_first = bias[n_first + tid % NBc];
    if (m0 >= M) return;
    const int n = m0 + wv;
    float *row = At + wv * KP;
    const bool active = n < M;                       // wave-uniform
    const int ne = active ? e1s - e0 : 0;
    // every global load of the prologue is in flight before the first one is waited for: the first 64 edges, the
    // node's own row and its skip row (<= 2 registers each; longer rows loop)
    int first_src = 0, first_cd = 0;
    if (lane < min(64, ne)) {
        first_src = col[e0 + lane];
        first_cd = code[e0 + lane];
    }
    float rv[2] = {0.0f, 0.0f}, sv[2] = {0.0f, 0.0f};
    if (active) {
        const float *xn = x + (size_t)n * ldx;
#pragma unroll
        for (int r = 0; r < 2; r++)
            if (lane + 64 * r < cin) rv[r] = xn[lane + 64 * r];
        if (cskip > 0) {
            const float *sn = xskip + (size_t)n * ldskip;
#pragma unroll
            for (int r = 0; r < 2; r++)
                if (lane + 64 * r < cskip) sv[r] = sn[lane + 64 * r];
        }
    }
    // ---- phase A of pass p: wave wv aggregates the pass's taps of node m0 + wv (edge order and arithmetic of
    // k_tap_aggregate) into its row of the tile
    auto phase_a = [&](int p) {
        const int k_lo = p * tp * cin;                                  // first column of the pass
        const int k_taps = (min(25, (p + 1) * tp) - p * tp) * cin;       // its tap columns
        for (int i = lane; i < KP; i += 64) row[i] = 0.0f;
        if (!active) return;
        if (p == P - 1) {
            float *rt = row + (25 * cin - k_lo);
#pragma unroll
            for (int r = 0; r < 2; r++) {
                if (lane + 64 * r < cin) rt[lane + 64 * r] = rv[r];
                if (lane + 64 * r < cskip) rt[cin + lane + 64 * r] = sv[r];
            }
            const float *xn = x + (size_t)n * ldx;
            for (int i = lane + 128; i < cin; i += 64) rt[i] = xn[i];
            if (cskip > 128) {
                const float *sn = xskip + (size_t)n * ldskip;
                for (int i = lane + 128; i < cskip; i += 64) rt[cin + i] = sn[i];
            }
        }
        for (int base = 0; base < ne; base += 64) {
            const int cnt = min(64, ne - base);
            int my_src = first_src, my_cd = first_cd;
            if (base > 0) {
                my_src = 0; my_cd = 0;
                if (lane < cnt) {
                    my_src = col[e0 + base + lane];
                    my_cd = code[e0 + base + lane];
                }
            }
                    // The spline basis of an edge is the same for all channels: lane e evaluates edge e once (4 tap columns,
            // packed 2 x 16 bits, and 4 weights -- same expressions as k_tap_aggregate), the channel lanes below
            // pick them up with readlane.  Evaluating it per edge on all 64 lanes was most of this phase's issue time.
            int o01, o23;
            float w00, w10, w01, w11;
            {
                const AxisF ax = spline_axis_f(my_cd & 0xffff, rx, den_x);
                const AxisF ay = spline_axis_f(my_cd >> 16, ry, den_y);
                w00 = ax.b0 * ay.b0; w10 = ax.b1 * ay.b0; w01 = ax.b0 * ay.b1; w11 = ax.b1 * ay.b1;
                o01 = ((ax.k0 + 5 * ay.k0) * cin) | (((ax.k1 + 5 * ay.k0) * cin) << 16);
                o23 = ((ax.k0 + 5 * ay.k1) * cin) | (((ax.k1 + 5 * ay.k1) * cin) << 16);
            }
            for (int c0 = 0; c0 < cin; c0 += 64) {
                const int ch = c0 + lane;
                const bool ch_ok = ch < cin;
                // A remainder of a few channels beyond this chunk (every pooled level carries C + 2 = 66 / 130 inputs:
                // the features and pos_xy) rides along on the first lanes instead of costing a second walk over the
                // edges with two active lanes -- same per-channel arithmetic and order, half the dependent chain.
                const int rem = cin - (c0 + 64);
                const bool ride = rem > 0 && rem <= 8;       // wave-uniform
                const int ch2 = c0 + 64 + lane;
                const bool ch2_ok = ride && lane < rem;
                constexpr int UA = 8;
                for (int j = 0; j < cnt; j += UA) {
                    float v[UA], v2[UA];
#pragma unroll
                    for (int u = 0; u < UA; u++) {   // UA gathers in flight
                        const int jj = min(j + u, cnt - 1);
                        const int src = __builtin_amdgcn_readlane(my_src, jj);
                        v[u] = ch_ok ? x[(size_t)src * ldx + ch] : 0.0f;
                        v2[u] = ch2_ok ? x[(size_t)src * ldx + ch2] : 0.0f;
                    }
#pragma unroll
                    for (int u = 0; u < UA; u++) {
                        if (j + u < cnt) {                   // wave-uniform
                            const int p01 = __builtin_amdgcn_readlane(o01, j + u);
                            const int p23 = __builtin_amdgcn_readlane(o23, j + u);
                            const float bq[4] = {__int_as_float(__builtin_amdgcn_readlane(__float_as_int(w00), j + u)),
                                                 __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w10), j + u)),
                                                 __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w01), j + u)),
                                                 __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w11), j + u))};
                            // columns of the four taps inside this pass's tile; a tap of another pass is skipped (scalar)
                            const int tq[4] = {(p01 & 0xffff) - k_lo, (p01 >> 16) - k_lo, (p23 & 0xffff) - k_lo,
                                               (p23 >> 16) - k_lo};
                            // the four taps of one edge are distinct (k0 != k1 on both axes): read all, then write
                            float o[4], q[4];
#pragma unroll
                            for (int t = 0; t < 4; t++) {
                                if ((unsigned)tq[t] < (unsigned)k_taps) {
                                    if (ch_ok) o[t] = row[tq[t] + ch];
                                    if (ch2_ok) q[t] = row[tq[t] + ch + 64];
                                }
                            }
#pragma unroll
                            for (int t = 0; t < 4; t++) {
                                if ((unsigned)tq[t] < (unsigned)k_taps) {
                                    if (ch_ok) row[tq[t] + ch] = o[t] + bq[t] * v[u];
                                    if (ch2_ok) row[tq[t] + ch + 64] = q[t] + bq[t] * v2[u];
                                }
                            }
                        }
                    }
                }
                if (ride) break;
            }
        }
    };
    // ---- phase B: [16 x K] . [K x N].  Wave (ks, w): K share ks, 16-column tile w of the column block.
    // No weight staging: every weight element is used by exactly one wave of the block, so the B operands
    // come straight from L2 in the host-packed MFMA operand order Wq[col tile][k group of 16][lane][4]
    // (element j of lane l = W[16 g + 4 j + (l >> 4)][16 c + (l & 15)]): one 1-KiB fully coalesced wave
    // load feeds 4 MFMAs, U of them are in flight per wave, and the loop has no barrier.
    phase_a(0);
    __syncthreads();
    for (int n0 = n_first; n0 < n_last; n0 += NBc) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        // the pass's share of the contraction for column block n0
        auto contract = [&](int p, bool first_batch_loaded) {
            if (n0 + w * 16 >= N) return;
            int g_lo, gbeg, gend;
            pass_groups(p, g_lo, gbeg, gend);
            const int nfull = max(0, gend - gbeg) / U;              // wave-uniform (wv is scalar)
            const float *a_rd = At + nn * KP + kk - 16 * g_lo;
            const float4 *wq = reinterpret_cast<const float4 *>(Wq) + ((size_t)(n0 / 16 + w) * G) * 64 + lane;
            auto load_b = [&](float4(&dst)[U], int g0) {
#pragma unroll
                for (int u = 0; u < U; u++) dst[u] = wq[(size_t)(g0 + u) * 64];
            };
            // U groups = 16 k-steps: all A operands are read from the tile first, then the MFMAs run back to back
            auto mac = [&](const float4(&b)[U], int g0) {
                float a[U][4];
                const float *ap = a_rd + 16 * g0;
#pragma unroll
                for (int u = 0; u < U; u++)
#pragma unroll
                    for (int j = 0; j < 4; j++) a[u][j] = ap[16 * u + 4 * j];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][0], b[u].x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][1], b[u].y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][2], b[u].z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][3], b[u].w, acc1, 0, 0, 0);
                }
            };
            int g = gbeg;
            if (nfull > 0 && !first_batch_loaded) load_b(b0, g);   // the very first batch was requested at kernel entry
            for (int it = 0; it < nfull; it += 2) {     // ping-pong: no register copies between batches
                if (it + 1 < nfull) load_b(b1, g + U);
                mac(b0, g);
                if (it + 1 < nfull) {
                    if (it + 2 < nfull) load_b(b0, g + 2 * U);
                    mac(b1, g + U);
                }
                g += 2 * U;
            }
            {   // < U left-over groups: all of their operands are requested before the first is used (on the small
                // levels K / 16 splits into 7 groups per wave: one full batch + 3 left-overs, which used to be three
                // dependent load -> MFMA round trips)
                const int g0 = gbeg + nfull * U, rem = gend - g0;
                float4 bt[U - 1];
#pragma unroll
                for (int r = 0; r < U - 1; r++)
                    bt[r] = r < rem ? wq[(size_t)(g0 + r) * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int r = 0; r < U - 1; r++) {
                    if (r < rem) {
                        const float *ap = a_rd + 16 * (g0 + r);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[0], bt[r].x, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4], bt[r].y, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[8], bt[r].z, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[12], bt[r].w, acc1, 0, 0, 0);
                    }
                }
            }
        };
        contract(0, n0 == n_first);
        for (int p = 1; p < P; p++) {           // (one column block per workgroup: the tile is rebuilt per pass)
            __syncthreads();
            phase_a(p);
            __syncthreads();
            contract(p, false);
        }
        const f32x4 acc = {acc0[0] + acc1[0], acc0[1] + acc1[1], acc0[2] + acc1[2], acc0[3] + acc1[3]};
            float *rd = red + ks * 16 * NBc;
#pragma unroll
        for (int q = 0; q < 4; q++) rd[(kk * 4 + q) * NBc + w * 16 + nn] = acc[q];
        __syncthreads();
        if (tid < 16 * NBc) {
            const int idx = tid;
            const int orow = m0 + idx / NBc, ocol = n0 + idx % NBc;
            if (orow < M && ocol < N) {
                float v = red[idx];
                for (int s = 1; s < ksplit; s++) v += red[s * 16 * NBc + idx];   // fixed order
                v += (n0 == n_first) ? bias_first : (bias ? bias[ocol] : 0.f);
                if (relu) v = fmaxf(v, 0.f);
                C[(size_t)orow * ldc + ocol] = v;
            }
        }
        __syncthreads();
        }
}

// Pass scheme of a fused conv: taps per pass and the tile's row stride; false when even one tap does not fit.
bool fused_plan(int cin, int cskip, int *tp_out, int *kp_out) {
    auto kp_of = [](int k) { return (k + 1 + 31) / 32 * 32 + 2; };   // >= k + 3 and = 2 (mod 32): bank = 2*row + k, conflict-free per half-wave
    const size_t red_bytes = (size_t)KSPLIT * 16 * NB * 4;
    auto fits = [&](int k) { return (size_t)16 * kp_of(k) * 4 + red_bytes <= 160 * 1024; };
    const int K = 26 * cin + cskip;
    if (fits(K)) { *tp_out = 25; *kp_out = kp_of(K); return true; }
    int gcd = 16;
    while (cin % gcd) gcd >>= 1;
    const int m = 16 / gcd;                     // tp * cin is a multiple of 16
    // fewest passes; among those the smallest tile
    for (int P = 2; P <= 25; P++) {
        int best_tp = 0, best_k = 0;
        for (int tp = m; tp < 25; tp += m) {
            if ((25 + tp - 1) / tp != P) continue;
            const int last = 25 - (P - 1) * tp;
            const int k = std::max(tp * cin, last * cin + cin + cskip);
            if (fits(k) && (best_tp == 0 || k < best_k)) { best_tp = tp; best_k = k; }
        }
        if (best_tp) { *tp_out = best_tp; *kp_out = kp_of(best_k); return true; }
    }
    return false;
}

}  // namespace
}  // namespace dagr

namespace {
struct HostConvJob {        // one fused conv as the C entry points describe it
    const int32_t *n_ptr;
    int32_t n_max;
    const int32_t *rowptr, *col, *code;
    const float *x;
    int32_t ldx, cin;
    const float *xskip;
    int32_t ldskip, cskip, rx, ry;
    float den_x, den_y;
    const float *Wq, *bias;
    float *C;
    int32_t ldc, N, relu;
};
}  // namespace

// 1 .. 4 independent fused convs in ONE launch (blockIdx.z = job).  All of them in the same form (single pass, or passes).
static int launch_conv_jobs(const HostConvJob *hj, int count, const dagr::PoolFuse *pool, void *stream) {
    using namespace dagr;
    DAGR_CHECK_ARG(count >= 1 && count <= 1 + kMaxExtraJobs, "1 .. 4 convs per launch");
    ConvJob dev[1 + kMaxExtraJobs];
    size_t lds_max = 0;
    int mp_all = -1, live = 0;
    unsigned gx = 1, gy = 1;
    for (int i = 0; i < count; i++) {
        const HostConvJob &h = hj[i];
        DAGR_CHECK_ARG(h.n_max >= 0 && h.cin >= 1 && h.N >= 1, "bad sizes");
        ConvJob &d = dev[i];
        d = ConvJob{h.n_ptr, h.rowptr, h.col, h.code, h.x, h.xskip, h.Wq, h.bias, h.C, h.n_max, h.ldx, h.cin, h.ldskip,
                    h.cskip, h.rx, h.ry, h.ldc, h.N, h.relu, 0, 4, 0, 25, h.den_x, h.den_y};
        if (h.n_max == 0) continue;             // (gy = 0: every workgroup of the job leaves at once)
        live++;
        DAGR_CHECK_ARG(h.rowptr && h.col && h.code && h.x && h.Wq && h.C, "NULL pointer");
        DAGR_CHECK_ARG(h.cskip == 0 || h.xskip, "xskip is NULL");
        DAGR_CHECK_ARG(((uintptr_t)h.Wq % 16) == 0, "packed weights must be 16-byte aligned");
        int tp = 25, KP = 0;
        if (!fused_plan(h.cin, h.cskip, &tp, &KP)) {
            set_error("dagr_spline_conv_fused: one tap of the input row does not fit the LDS tile (use tap_aggregate + gemm)");
            return DAGR_ERR_UNSUPPORTED;
        }
        const int mp = tp < 25 ? 1 : 0;
        if (mp_all >= 0 && mp != mp_all) {
            set_error("dagr_spline_conv_fused_multi: single-pass and multi-pass convs cannot share a launch");
            return DAGR_ERR_UNSUPPORTED;
        }
        mp_all = mp;
        lds_max = std::max(lds_max, ((size_t)16 * KP + (size_t)KSPLIT * 16 * NB) * 4);
        // Column tiles per workgroup (4, 2 or 1): the tile leaves room for one workgroup per CU, so the launch runs in
        // ceil(workgroups / CUs) rounds.  A round costs the edge walk + prologue (~5 us, repeated by every workgroup of a
        // node tile) plus ~1.5 us per column tile (measured, tools/microbench/conv_trace.hip); few nodes -> spread the
        // columns over workgroups, but never into an extra round: 71 node tiles x 4 column workgroups ran 284 workgroups
        // on 256 CUs, 18.5 us where 141 node tiles took 16.9.
        const int row_blocks = ceil_div(h.n_max, 16);
        const int passes = tp >= 25 ? 1 : (25 + tp - 1) / tp;
        int nc = 4, loop_cols = passes == 1 ? 1 : 0;
        float best = 0.0f;
        for (int c = 4; c >= 1; c >>= 1) {
            // single pass, 4 tiles: one workgroup per node tile walks all 64-column blocks over its tile; else gridDim.y
            // workgroups of c tiles each
            const int loop = (c == 4 && passes == 1) ? 1 : 0;
            // (the job's own workgroups: the split -- and with it the summation order of the K partials -- must not
            // depend on which other convs share the launch)
            const int wgs = loop ? row_blocks : row_blocks * (int)ceil_div(h.N, 16 * c);
            const float tiles = loop ? 4.0f * (float)ceil_div(h.N, 64) : (float)c;
            // n_max is a capacity: a level's table has one sample plane more than the batch fills (QUIRK-1), and
            // workgroups past the device-side count leave at once -- count 9 in 10 as live
            const float cost = (float)ceil_div((int64_t)wgs * 9 / 10, device_cu_count()) * (5.0f * (float)passes + 1.5f * tiles);
            if (best == 0.0f || cost < best) { best = cost; nc = c; loop_cols = loop; }
        }
        d.KP = KP; d.NC = nc; d.tp = tp;
        d.gy = loop_cols ? 1 : (int)ceil_div(h.N, 16 * nc);
        gx = std::max(gx, (unsigned)row_blocks);
        gy = std::max(gy, (unsigned)d.gy);
    }
    if (live == 0) return DAGR_OK;
    const bool mp = mp_all == 1;
    DAGR_CHECK_ARG(!(pool && (mp || dev[0].gy == 0)), "the fused pooling merge needs the single-pass form of the conv");
    // two workgroups per CU (the 62-register form) when the launch takes more than one round of workgroups and two tiles
    // fit the LDS; builder knob DAGR_CONV_DENSE=0: never
    static const bool dense_ok = knob("DAGR_CONV_DENSE", 1) != 0;
    const bool dense = dense_ok && !mp && lds_max <= 80 * 1024 && (int64_t)gx * gy * count > device_cu_count();
    const int which = mp ? 1 : (dense ? 2 : 0);
    static thread_local size_t set_max[3] = {0, 0, 0};
    if (lds_max > set_max[which]) {
        const void *fn = mp ? (const void *)k_conv_fused_mp : (dense ? (const void *)k_conv_fused<3> : (const void *)k_conv_fused<4>);
        DAGR_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
        set_max[which] = lds_max;
    }
    ConvJobs jobs{};
    jobs.extra = count - 1;
    for (int i = 1; i < count; i++) jobs.j[i - 1] = dev[i];
    const ConvJob &a = dev[0];
    const dim3 grid(gx, gy, (unsigned)count);
    PoolFuse pf{};
    if (pool) pf = *pool;
    if (mp)
        k_conv_fused_mp<<<grid, kGemmThreads, lds_max, (hipStream_t)stream>>>(
            a.n_ptr, a.n_max, a.rowptr, a.col, a.code, a.x, a.ldx, a.cin, a.xskip, a.ldskip, a.cskip, a.rx, a.ry, a.den_x,
            a.den_y, a.Wq, a.bias, a.C, a.ldc, a.N, a.relu, a.KP, a.NC, a.tp, a.gy, jobs);
    else if (dense)
        k_conv_fused<3><<<grid, kGemmThreads, lds_max, (hipStream_t)stream>>>(
            a.n_ptr, a.n_max, a.rowptr, a.col, a.code, a.x, a.ldx, a.cin, a.xskip, a.ldskip, a.cskip, a.rx, a.ry, a.den_x,
            a.den_y, a.Wq, a.bias, a.C, a.ldc, a.N, a.relu, a.KP, a.NC, a.gy, jobs, pf);
    else
        k_conv_fused<4><<<grid, kGemmThreads, lds_max, (hipStream_t)stream>>>(
            a.n_ptr, a.n_max, a.rowptr, a.col, a.code, a.x, a.ldx, a.cin, a.xskip, a.ldskip, a.cskip, a.rx, a.ry, a.den_x,
            a.den_y, a.Wq, a.bias, a.C, a.ldc, a.N, a.relu, a.KP, a.NC, a.gy, jobs, pf);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}

extern "C" int dagr_spline_conv_fused(const int32_t *n_nodes_ptr, int32_t n_nodes_max, const int32_t *rowptr,
                                      const int32_t *col, const int32_t *code, const float *x, int32_t ldx, int32_t cin,
                                      const float *xskip, int32_t ldskip, int32_t cskip, int32_t rx, int32_t ry,
                                      float den_x, float den_y, const float *Wq, const float *bias, float *C,
                                      int32_t ldc, int32_t N, int32_t relu, void *stream) {
    using namespace dagr;
    // narrow inputs on a level of several workgroup rounds: 48- / 80-node tiles, three nodes per wave and pass
    // (k_conv_fused_narrow).  Measurement knob DAGR_CONV_NARROW: 0 = never, 3 / 5 = force that tile form
    static const int narrow = (int)knob("DAGR_CONV_NARROW", 1);
    if (narrow && cin >= 1 && cin <= kNarrowSub && cskip == 0 && N >= 1 &&
        (int64_t)n_nodes_max >= (int64_t)48 * device_cu_count()) {
        DAGR_CHECK_ARG(rowptr && col && code && x && Wq && C, "NULL pointer");
        DAGR_CHECK_ARG(((uintptr_t)Wq % 16) == 0, "packed weights must be 16-byte aligned");
        int tp = 25, KP = 0;
        if (fused_plan(cin, 0, &tp, &KP) && tp >= 25) {
            // 80-node tiles when that makes the level one round of workgroups and the tile fits (the aliased partials serve
            // ONE 64-column block)
            const size_t tile5 = (size_t)16 * 5 * KP * 4, part5 = (size_t)KSPLIT * 16 * 5 * NB * 4;
            const bool two = narrow != 3 && N <= NB && std::max(tile5, part5) <= 160 * 1024 &&
                             (narrow == 5 || ceil_div(n_nodes_max, 80) <= device_cu_count());
            const int rt = two ? 5 : 3;
            const size_t tile = (size_t)16 * rt * KP * 4, part = (size_t)KSPLIT * 16 * rt * NB * 4;
            const size_t lds = two ? std::max(tile, part) : tile + part;
            const void *fn = two ? (const void *)k_conv_fused_narrow<5, true> : (const void *)k_conv_fused_narrow<3, false>;
            static thread_local size_t set_lds[2] = {0, 0};
            if (lds > set_lds[two ? 1 : 0]) {
                DAGR_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                set_lds[two ? 1 : 0] = lds;
            }
            const unsigned grid = (unsigned)ceil_div(n_nodes_max, 16 * rt);
            if (two)
                k_conv_fused_narrow<5, true><<<grid, kGemmThreads, lds, (hipStream_t)stream>>>(
                    n_nodes_ptr, n_nodes_max, rowptr, col, code, x, ldx, cin, rx, ry, den_x, den_y, Wq, bias, C, ldc, N, relu, KP);
            else
                k_conv_fused_narrow<3, false><<<grid, kGemmThreads, lds, (hipStream_t)stream>>>(
                    n_nodes_ptr, n_nodes_max, rowptr, col, code, x, ldx, cin, rx, ry, den_x, den_y, Wq, bias, C, ldc, N, relu, KP);
            DAGR_CHECK_LAUNCH();
            return DAGR_OK;
        }
    }
    const HostConvJob j{n_nodes_ptr, n_nodes_max, rowptr, col, code, x, ldx, cin, xskip, ldskip, cskip, rx, ry, den_x, den_y,
                        Wq, bias, C, ldc, N, relu};
    return launch_conv_jobs(&j, 1, nullptr, stream);
}

extern "C" int dagr_spline_conv_fused_pool(const int32_t *n_nodes_ptr, int32_t n_nodes_max, const int32_t *rowptr,
                                           const int32_t *col, const int32_t *code, const float *x, int32_t ldx, int32_t cin,
                                           const float *xskip, int32_t ldskip, int32_t cskip, int32_t rx, int32_t ry,
                                           float den_x, float den_y, const float *Wq, const float *bias, float *C,
                                           int32_t ldc, int32_t N, int32_t relu, const dagr_pool_desc *pdesc, void *pool_ws,
                                           const float *pos, const int32_t *batch, int32_t *cluster_scratch, void *stream) {
    using namespace dagr;
    DAGR_CHECK_ARG(pdesc && pool_ws && pos && batch && cluster_scratch, "NULL pooling arguments");
    DAGR_CHECK_ARG(pdesc->channels == N && pdesc->gx > 0 && pdesc->gy > 0 && pdesc->batch_size > 0 &&
                       (pdesc->aggr == 0 || pdesc->aggr == 1), "the pooling consumes exactly this conv's N output columns");
    if (n_nodes_max == 0) return DAGR_OK;
    PoolFuse pf{};
    pf.on = 1;
    pf.d = *pdesc;
    pool_carve(*pdesc, (char *)pool_ws, &pf.ws);
    pf.pos = pos; pf.batch = batch; pf.cluster_raw_out = cluster_scratch;
    const HostConvJob j{n_nodes_ptr, n_nodes_max, rowptr, col, code, x, ldx, cin, xskip, ldskip, cskip, rx, ry, den_x, den_y,
                        Wq, bias, C, ldc, N, relu};
    return launch_conv_jobs(&j, 1, &pf, stream);
}

extern "C" int dagr_spline_conv_fused_pair(const int32_t *n_nodes_ptr, int32_t n_nodes_max, const int32_t *rowptr,
                                           const int32_t *col, const int32_t *code, int32_t ldx, int32_t cin, int32_t rx,
                                           int32_t ry, float den_x, float den_y, int32_t ldc, int32_t relu, const float *x_a,
                                           const float *Wq_a, const float *bias_a, float *C_a, int32_t N_a,
                                           const float *x_b, const float *Wq_b, const float *bias_b, float *C_b,
                                           int32_t N_b, void *stream) {
    DAGR_CHECK_ARG(x_b && Wq_b && C_b && N_b >= 1 && ((uintptr_t)Wq_b % 16) == 0, "bad second conv");
    const HostConvJob j[2] = {
        {n_nodes_ptr, n_nodes_max, rowptr, col, code, x_a, ldx, cin, nullptr, 0, 0, rx, ry, den_x, den_y, Wq_a, bias_a, C_a,
         ldc, N_a, relu},
        {n_nodes_ptr, n_nodes_max, rowptr, col, code, x_b, ldx, cin, nullptr, 0, 0, rx, ry, den_x, den_y, Wq_b, bias_b, C_b,
         ldc, N_b, relu}};
    return launch_conv_jobs(j, 2, nullptr, stream);
}

extern "C" int dagr_spline_conv_fused_multi(const dagr_conv_job *jobs, int32_t count, void *stream) {
    DAGR_CHECK_ARG(jobs && count >= 1 && count <= 4, "1 .. 4 jobs");
    HostConvJob j[4];
    for (int i = 0; i < count; i++) {
        const dagr_conv_job &q = jobs[i];
        j[i] = HostConvJob{q.n_nodes_ptr, q.n_nodes_max, q.rowptr, q.col, q.code, q.x, q.ldx, q.cin, q.xskip, q.ldskip,
                           q.cskip, q.rx, q.ry, q.den_x, q.den_y, q.Wq, q.bias, q.C, q.ldc, q.N, q.relu};
    }
    return launch_conv_jobs(j, count, nullptr, stream);
}

extern "C" size_t dagr_spline_conv_fused_lds_bytes(int32_t cin, int32_t cskip) {
    using namespace dagr;
    int tp = 25, KP = 0;
    if (!fused_plan(cin, cskip, &tp, &KP)) return (size_t)1 << 30;     // no pass scheme: callers compare against 160 KiB
    return ((size_t)16 * KP + (size_t)KSPLIT * 16 * NB) * 4;
}

extern "C" int32_t dagr_spline_conv_fused_passes(int32_t cin, int32_t cskip) {
    using namespace dagr;
    int tp = 25, KP = 0;
    if (cin < 1 || cskip < 0 || !fused_plan(cin, cskip, &tp, &KP)) return 0;
    return tp >= 25 ? 1 : (25 + tp - 1) / tp;
}
