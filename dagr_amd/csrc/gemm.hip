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
#include "common.hpp"

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
