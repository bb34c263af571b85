// gemm_lt.hip -- the 1x1 convolutions of the channels-last image branch as library GEMMs with their whole epilogue fused.
//
// Reference: the ResNet bottleneck of the image branch (src/dagr/model/networks/net_img.py:42-48): out = relu(bn3(conv3(.))
// + identity) and relu(bn1(conv1(.))).  With BatchNorm folded (eval) a 1x1 convolution on a channels-last map IS a
// row-major GEMM [B*H*W, Cin] x [Cin, Cout] + bias; hipBLASLt's epilogue adds the bias, an optional residual matrix
// (beta = 1 on C) and the ReLU inside the same kernel -- the residual join of every bottleneck used to be a pass of its
// own over the block's largest map (16 launches, 0.5 ms of a B = 8 step).  Plain library GEMM, fp32 in / fp32 accumulate.
#include <hipblaslt/hipblaslt.h>

#include <map>
#include <tuple>

#include "common.hpp"

namespace dagr {
namespace {

struct LtPlan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t a = nullptr, b = nullptr, c = nullptr, d = nullptr;
    hipblasLtMatmulAlgo_t algo;
    size_t workspace = 0;
};

hipblasLtHandle_t lt_handle() {
    static hipblasLtHandle_t h = [] {
        hipblasLtHandle_t hh = nullptr;
        if (hipblasLtCreate(&hh) != HIPBLAS_STATUS_SUCCESS) hh = nullptr;
        return hh;
    }();
    return h;
}

}  // namespace
}  // namespace dagr

using namespace dagr;

// D[M, N] = act(A[M, K] . Wt[K, N] + bias[N] (+ R[M, N])), row-major, fp32.  act: 0 none, 1 ReLU.  R may alias nothing else;
// D may not alias A.  workspace: device scratch for the library (>= dagr_gemm_epilogue_workspace_bytes()).
extern "C" size_t dagr_gemm_epilogue_workspace_bytes(void) { return (size_t)64 << 20; }

extern "C" int dagr_gemm_epilogue(const float *A, int64_t M, int32_t K, int64_t lda, const float *Wt, int32_t N,
                                  const float *bias, const float *R, int64_t ldr, int32_t act, float *D, int64_t ldd,
                                  void *workspace, size_t workspace_bytes, void *stream) {
    DAGR_CHECK_ARG(M >= 0 && K >= 1 && N >= 1 && lda >= K && ldd >= N && (!R || ldr >= N) && (act == 0 || act == 1), "bad sizes");
    if (M == 0) return DAGR_OK;
    DAGR_CHECK_ARG(A && Wt && D, "NULL pointer");
    hipblasLtHandle_t h = lt_handle();
    if (!h) { set_error("dagr_gemm_epilogue: hipblasLtCreate failed"); return DAGR_ERR_HIP; }
    // column-major view: D^T[N x M] = Wt^T[N x K] . A^T[K x M]; the bias runs along the rows of D^T (= output channels)
    using Key = std::tuple<int64_t, int, int, int64_t, int64_t, int64_t, int, int, int>;
    static thread_local std::map<Key, LtPlan> plans;
    const Key key{M, K, N, lda, R ? ldr : -1, ldd, act, bias ? 1 : 0, R ? 1 : 0};
    auto it = plans.find(key);
    if (it == plans.end()) {
        LtPlan p;
#define LT_OK(expr)                                                                                        \
    do {                                                                                                   \
        if ((expr) != HIPBLAS_STATUS_SUCCESS) { set_error("dagr_gemm_epilogue: " #expr " failed"); return DAGR_ERR_HIP; } \
    } while (0)
        LT_OK(hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
        const hipblasOperation_t op_n = HIPBLAS_OP_N;
        LT_OK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &op_n, sizeof(op_n)));
        LT_OK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &op_n, sizeof(op_n)));
        hipblasLtEpilogue_t epi = act ? (bias ? HIPBLASLT_EPILOGUE_RELU_BIAS : HIPBLASLT_EPILOGUE_RELU)
                                      : (bias ? HIPBLASLT_EPILOGUE_BIAS : HIPBLASLT_EPILOGUE_DEFAULT);
        LT_OK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)));
        if (bias) {
            const hipDataType bt = HIP_R_32F;
            LT_OK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)));
            LT_OK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)));
        }
        LT_OK(hipblasLtMatrixLayoutCreate(&p.a, HIP_R_32F, (uint64_t)N, (uint64_t)K, (int64_t)N));      // Wt^T
        LT_OK(hipblasLtMatrixLayoutCreate(&p.b, HIP_R_32F, (uint64_t)K, (uint64_t)M, lda));             // A^T
        LT_OK(hipblasLtMatrixLayoutCreate(&p.c, HIP_R_32F, (uint64_t)N, (uint64_t)M, R ? ldr : ldd));   // R^T (or D^T, beta = 0)
        LT_OK(hipblasLtMatrixLayoutCreate(&p.d, HIP_R_32F, (uint64_t)N, (uint64_t)M, ldd));             // D^T
        hipblasLtMatmulPreference_t pref = nullptr;
        LT_OK(hipblasLtMatmulPreferenceCreate(&pref));
        const uint64_t max_ws = workspace ? (uint64_t)workspace_bytes : 0;
        LT_OK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &max_ws, sizeof(max_ws)));
        constexpr int kCand = 16;
        hipblasLtMatmulHeuristicResult_t res[kCand];
        int found = 0;
        const hipblasStatus_t st = hipblasLtMatmulAlgoGetHeuristic(h, p.desc, p.a, p.b, p.c, p.d, pref, kCand, res, &found);
        hipblasLtMatmulPreferenceDestroy(pref);
        if (st != HIPBLAS_STATUS_SUCCESS || found < 1) {
            set_error("dagr_gemm_epilogue: the library offers no kernel for this GEMM + epilogue");
            return DAGR_ERR_UNSUPPORTED;
        }
        int best = 0;
        // The library ranks its kernels by a model; the first time a shape is seen (a warm-up call, never inside a stream
        // capture) the candidates are timed on the caller's own operands and the fastest is kept for the shape.
        static const bool tune = knob("DAGR_LT_TUNE", 1) != 0;
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing((hipStream_t)stream, &cap);
        if (tune && found > 1 && cap == hipStreamCaptureStatusNone) {
            hipEvent_t e0, e1;
            if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
                const float alpha = 1.0f, beta = R ? 1.0f : 0.0f;
                float best_ms = 0.f;
                for (int c = 0; c < found; c++) {
                    if (res[c].workspaceSize > workspace_bytes) continue;
                    bool ok = true;
                    float ms = 0.f;
                    for (int rep = 0; rep < 3 && ok; rep++) {       // first repetition = warm-up (code load), last two timed
                        if (rep == 1) (void)hipEventRecord(e0, (hipStream_t)stream);
                        ok = hipblasLtMatmul(h, p.desc, &alpha, Wt, p.a, A, p.b, &beta, R ? (const void *)R : (const void *)D,
                                             p.c, D, p.d, &res[c].algo, workspace, workspace_bytes,
                                             (hipStream_t)stream) == HIPBLAS_STATUS_SUCCESS;
                    }
                    (void)hipEventRecord(e1, (hipStream_t)stream);
                    if (hipEventSynchronize(e1) != hipSuccess || !ok) continue;
                    (void)hipEventElapsedTime(&ms, e0, e1);
                    if (best_ms == 0.f || ms < best_ms) { best_ms = ms; best = c; }
                }
                (void)hipEventDestroy(e0);
                (void)hipEventDestroy(e1);
            }
        }
        p.algo = res[best].algo;
        p.workspace = res[best].workspaceSize;
        it = plans.emplace(key, p).first;
#undef LT_OK
    }
    LtPlan &p = it->second;
    if (bias && hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)) !=
                    HIPBLAS_STATUS_SUCCESS) {
        set_error("dagr_gemm_epilogue: setting the bias pointer failed");
        return DAGR_ERR_HIP;
    }
    DAGR_CHECK_ARG(p.workspace <= workspace_bytes, "workspace too small for the selected kernel");
    const float alpha = 1.0f, beta = R ? 1.0f : 0.0f;
    const hipblasStatus_t st = hipblasLtMatmul(h, p.desc, &alpha, Wt, p.a, A, p.b, &beta, R ? (const void *)R : (const void *)D,
                                               p.c, D, p.d, &p.algo, workspace, workspace_bytes, (hipStream_t)stream);
    if (st != HIPBLAS_STATUS_SUCCESS) {
        set_error("dagr_gemm_epilogue: hipblasLtMatmul failed");
        return DAGR_ERR_HIP;
    }
    return DAGR_OK;
}
