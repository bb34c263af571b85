// downsample.hip -- event-stream downsampling on the device (f5: the on-GPU half of the event I/O).
// Reference: scripts/downsample_events.py:91-124 (`downsample_events` / `_filter_events_resize`, a numba loop over all
// events): every output cell of fx x fy input pixels integrates polarity / (fx*fy); when |state| reaches 1 the event
// passes (at the cell's coordinates) and its polarity is subtracted.  The loop is sequential in time but independent
// between cells, so: events grouped by output cell in time order (stable sort by cell, done by the caller with one
// torch.sort -- the same preparation the reference's graph builder asks for, graph/utils.py:9-13), then one thread per
// occupied cell walks its run.  `change_map` carries the integrator state from chunk to chunk as in the reference.
#include "common.hpp"

namespace dagr {
namespace {

__global__ __launch_bounds__(kBlock) void k_downsample_cells(const int32_t *__restrict__ order,       // event ids by (cell, time)
                                                            const int32_t *__restrict__ cell_of,     // [n_runs] cell id
                                                            const int32_t *__restrict__ run_end,     // [n_runs] cumulative
                                                            int n_runs, const int8_t *__restrict__ p, float inc,
                                                            float *__restrict__ change_map, uint8_t *__restrict__ keep) {
    const int r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= n_runs) return;
    const int a = r == 0 ? 0 : run_end[r - 1], b = run_end[r];
    const int cell = cell_of[r];
    float state = change_map[cell];
    for (int k = a; k < b; k++) {
        const int e = order[k];
        const float pol = (float)p[e];
        // downsample_events.py:117: change_map += p * 1.0 / (fx * fy) -- a float64 product rounded into the fp32 map
        state = (float)((double)state + (double)pol * (double)inc);
        bool pass = fabsf(state) >= 1.0f;                                  // :119
        keep[e] = pass ? 1 : 0;
        if (pass) state -= pol;                                            // :121
    }
    change_map[cell] = state;
}

}  // namespace
}  // namespace dagr

using namespace dagr;

extern "C" int dagr_downsample_events(const int32_t *order, const int32_t *run_cell, const int32_t *run_end,
                                      int32_t n_runs, const int8_t *polarity, int32_t fx, int32_t fy, float *change_map,
                                      uint8_t *keep, void *stream) {
    DAGR_CHECK_ARG(n_runs >= 0 && fx >= 1 && fy >= 1, "bad sizes");
    if (n_runs == 0) return DAGR_OK;
    DAGR_CHECK_ARG(order && run_cell && run_end && polarity && change_map && keep, "NULL pointer");
    const float inc = (float)(1.0 / (double)(fx * fy));
    k_downsample_cells<<<(unsigned)ceil_div(n_runs, kBlock), kBlock, 0, (hipStream_t)stream>>>(
        order, run_cell, run_end, n_runs, polarity, inc, change_map, keep);
    DAGR_CHECK_LAUNCH();
    return DAGR_OK;
}
