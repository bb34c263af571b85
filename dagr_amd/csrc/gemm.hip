// gemm.hip -- C[M,N] = act(A[M,K] . Wm[K,N] + bias[N]) with exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// The pooled-level SplineConvs are contractions of the tap-aggregated rows A[n] (K = 26*Cin (+Cskip),
// up to 1682) with a packed weight matrix (N = 64 / 128): GEMM-shaped, so they run on the matrix
// cores.  The f32-input MFMA is bit-exact fp32 FMA at the vector rate (MI355X_MICROARCH.md), which keeps
// the 1e-4 parity bar with no precision trade.
//
// Shape of the problem: M is small (<= 2240*(B+1) rows, often a few hundred) and K is long, so the
// critical path of a 64x64 output tile is its K-chain (K/2 dependent 64-cycle MFMAs).  A block is
// therefore 16 waves: 2x2 waves tile the 64x64 output and 4 wave-quads split K (each quad owns a
// 32-wide slice of every 128-wide K step); the four partial tiles are summed through LDS in a fixed
// order (deterministic).  Operands are staged in LDS as [k][m] / [k][n] so that a lane's MFMA operand
// (i = lane&31, k = lane>>5) is a conflict-free row read; the next K step is prefetched into registers
// while the current one is multiplied.  M is bounded by a device-side count: no host sync.
#include "common.hpp"

namespace dagr {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int MB = 64, NB = 64, KB = 32, KSPLIT = 4;
constexpr int AS_STRIDE = MB + 1;   // [k][m]: reads are 32 consecutive m; odd stride spreads the transposed stores
constexpr int WS_STRIDE = NB + 4;   // [k][n]: 16-byte aligned rows for b128 stores
constexpr int kGemmThreads = 256 * KSPLIT;
constexpr int kStageFloats = KSPLIT * KB * (AS_STRIDE + WS_STRIDE);
constexpr int kReduceFloats = KSPLIT * MB * NB;
constexpr int kLdsFloats = kStageFloats > kReduceFloats ? kStageFloats : kReduceFloats;

__global__ __launch_bounds__(kGemmThreads) void k_gemm_mfma(const int32_t *__restrict__ m_ptr, int m_max,
                                                           const float *__restrict__ A, int lda,
                                                           const float *__restrict__ Wm, int ldw,
                                                           const float *__restrict__ bias, float *__restrict__ C,
                                                           int ldc, int K, int N, int relu) {
    __shared__ __align__(16) float lds[kLdsFloats];
    const int M = m_ptr ? min(*m_ptr, m_max) : m_max;
    const int m0 = blockIdx.x * MB, n0 = blockIdx.y * NB;
    if (m0 >= M) return;
    const int tid = threadIdx.x;
    const int ks = tid >> 8;            // K slice of this wave quad
    const int t = tid & 255, l = t & 63, w = t >> 6;
    const int wm = w & 1, wn = w >> 1;
    float *As = lds + ks * KB * AS_STRIDE;
    float *Ws = lds + KSPLIT * KB * AS_STRIDE + ks * KB * WS_STRIDE;
    // staging roles inside the slice
    const int a_row = t >> 2, a_kq = (t & 3) * 8;   // A: 64 rows x 32 k, 8 consecutive k per thread
    const int w_k = t >> 3, w_nq = (t & 7) * 8;     // W: 32 k x 64 n, 8 consecutive n per thread
    const bool a_ok = (m0 + a_row) < M;
    const float *a_src = A + (size_t)(m0 + a_row) * lda + a_kq;
    const bool w_ok = (n0 + w_nq + 8) <= ldw;       // weight rows are zero-padded to a multiple of 8 columns
    float4 ra0, ra1, rw0, rw1;
    auto load_tile = [&](int k0) {   // k0 = first k of this slice in this step
        ra0 = ra1 = rw0 = rw1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a_ok) {
            if (k0 + a_kq + 8 <= K) {
                ra0 = *reinterpret_cast<const float4 *>(a_src + k0);
                ra1 = *reinterpret_cast<const float4 *>(a_src + k0 + 4);
            } else if (k0 + a_kq < K) {
                float tmp[8];
#pragma unroll
                for (int j = 0; j < 8; j++) tmp[j] = (k0 + a_kq + j < K) ? a_src[k0 + j] : 0.f;
                ra0 = make_float4(tmp[0], tmp[1], tmp[2], tmp[3]);
                ra1 = make_float4(tmp[4], tmp[5], tmp[6], tmp[7]);
            }
        }
        if (w_ok && k0 + w_k < K) {
            const float *ws = Wm + (size_t)(k0 + w_k) * ldw + n0 + w_nq;
            rw0 = *reinterpret_cast<const float4 *>(ws);
            rw1 = *reinterpret_cast<const float4 *>(ws + 4);
        }
    };
    auto store_tile = [&]() {
        float *as = As + a_kq * AS_STRIDE + a_row;
        as[0 * AS_STRIDE] = ra0.x; as[1 * AS_STRIDE] = ra0.y; as[2 * AS_STRIDE] = ra0.z; as[3 * AS_STRIDE] = ra0.w;
        as[4 * AS_STRIDE] = ra1.x; as[5 * AS_STRIDE] = ra1.y; as[6 * AS_STRIDE] = ra1.z; as[7 * AS_STRIDE] = ra1.w;
        float4 *wsd = reinterpret_cast<float4 *>(Ws + w_k * WS_STRIDE + w_nq);
        wsd[0] = rw0;
        wsd[1] = rw1;
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    constexpr int KSTEP = KB * KSPLIT;
    load_tile(ks * KB);
    store_tile();
    __syncthreads();
    const float *a_rd = As + (l >> 5) * AS_STRIDE + wm * 32 + (l & 31);
    const float *w_rd = Ws + (l >> 5) * WS_STRIDE + wn * 32 + (l & 31);
    for (int k0 = 0; k0 < K; k0 += KSTEP) {
        const bool more = (k0 + KSTEP) < K;
        if (more) load_tile(k0 + KSTEP + ks * KB);
#pragma unroll
        for (int kk = 0; kk < KB; kk += 2) {
            const float a = a_rd[kk * AS_STRIDE];
            const float b = w_rd[kk * WS_STRIDE];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            store_tile();
            __syncthreads();
        }
    }
    // split-K reduction through LDS (staging buffers are dead after the last barrier above)
    float *red = lds + ks * MB * NB;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        red[row * NB + wn * 32 + (l & 31)] = acc[r];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < (MB * NB) / kGemmThreads; q++) {
        const int idx = q * kGemmThreads + tid;
        const int row = idx / NB, col = idx % NB;
        float v = ((lds[idx] + lds[MB * NB + idx]) + lds[2 * MB * NB + idx]) + lds[3 * MB * NB + idx];
        const int gm = m0 + row, gn = n0 + col;
        if (gm < M && gn < N) {
            v += bias ? bias[gn] : 0.f;
            if (relu) v = fmaxf(v, 0.f);
            C[(size_t)gm * ldc + gn] = v;
        }
    }
}

}  // namespace

hipError_t launch_gemm_mfma(const int32_t *m_ptr, int m_max, const float *A, int lda, const float *Wm, int ldw,
                            const float *bias, float *C, int ldc, int K, int N, int relu, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)k_gemm_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, 0);
        (void)e;
        attr_set = true;
    }
    dim3 grid((unsigned)ceil_div(m_max, MB), (unsigned)ceil_div(N, NB));
    k_gemm_mfma<<<grid, kGemmThreads, 0, stream>>>(m_ptr, m_max, A, lda, Wm, ldw, bias, C, ldc, K, N, relu);
    return hipGetLastError();
}

}  // namespace dagr
