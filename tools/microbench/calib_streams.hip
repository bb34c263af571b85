// calib_streams.hip (was csrc/debug.hip until round 4; no longer part of libdagr_hip) -- known-byte-count streaming kernels used to calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE on
// this access pattern (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own
// access pattern before trusting an absolute").  Not on the product path.
#include "common.hpp"

namespace dagr {
namespace {
__global__ __launch_bounds__(kBlock) void k_calib_read4(const float *__restrict__ src, size_t n, float *sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) acc += src[i];
    if (acc == 123456.789f) *sink = acc;
}
__global__ __launch_bounds__(kBlock) void k_calib_read16(const float4 *__restrict__ src, size_t n4, float *sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (size_t)gridDim.x * kBlock) {
        const float4 v = src[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123456.789f) *sink = acc;
}
// 64-byte rows gathered by 16-lane groups (the level-0 SplineConv's pattern): row index pseudo-random
__global__ __launch_bounds__(kBlock) void k_calib_gather64(const float *__restrict__ src, size_t rows, size_t n_gathers,
                                                          float *sink) {
    float acc = 0.f;
    const int l = threadIdx.x & 15;
    for (size_t g = ((size_t)blockIdx.x * kBlock + threadIdx.x) >> 4; g < n_gathers;
         g += ((size_t)gridDim.x * kBlock) >> 4) {
        const size_t r = (g * 2654435761ull + 12345ull) % rows;
        acc += src[r * 16 + l];
    }
    if (acc == 123456.789f) *sink = acc;
}
__global__ __launch_bounds__(kBlock) void k_calib_write4(float *__restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) dst[i] = 1.0f;
}
}  // namespace
}  // namespace dagr

using namespace dagr;

extern "C" int dagr_debug_calibrate(int32_t mode, float *buf, size_t n_floats, size_t n_gathers, float *sink,
                                    void *stream_) {
    DAGR_CHECK_ARG(buf && sink && n_floats >= 1024, "bad arguments");
    hipStream_t stream = (hipStream_t)stream_;
    const unsigned grid = 256 * 8;
    switch (mode) {
    case 0: k_calib_read4<<<grid, kBlock, 0, stream>>>(buf, n_floats, sink); break;
    case 1: k_calib_read16<<<grid, kBlock, 0, stream>>>((const float4 *)buf, n_floats / 4, sink); break;
    case 2: k_calib_gather64<<<grid, kBlock, 0, stream>>>(buf, n_floats / 16, n_gathers, sink); break;
    case 3: k_calib_write4<<<grid, kBlock, 0, stream>>>(buf, n_floats); break;
    default: DAGR_CHECK_ARG(false, "mode must be 0..3");
    }
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}
