// oracle/ref_driver.hip -- TEST INFRASTRUCTURE ONLY.
// Pulls in the reference's own graph-builder translation unit (src/dagr/graph/ev_graph.cu, path
// injected by oracle/Makefile; the file is compiled where it lies, never copied) and exposes its
// three host entry points (ev_graph.cu:279-283) through a C ABI over raw device pointers, so that
// tests can run the reference's real kernels on the MI355X box and compare them with the CPU
// oracle and with libdagr_hip.
#include DAGR_REF_EV_GRAPH_CU

static torch::Tensor view(const void *p, std::initializer_list<int64_t> shape) {
    torch::Tensor t;
    t.ptr = const_cast<void *>(p);
    t.shape = shape;
    return t;
}

extern "C" {

// ev_graph.cu:82-128
int ref_fill_edges(const int32_t *batch, const int32_t *pos, const int32_t *all_timestamps, int64_t n_ts,
                   const int32_t *event_queue, const int32_t *indices, int max_num_neighbors, float radius,
                   float delta_t_us, int64_t *edges, int64_t K, int min_index, int N, int B, int Q, int H, int W) {
    auto tb = view(batch, {N});
    auto tp = view(pos, {N, 3});
    auto tt = view(all_timestamps, {n_ts});
    auto tq = view(event_queue, {B, Q, H, W});
    auto ti = view(indices, {N});
    auto te = view(edges, {2, K});
    fill_edges_cuda(tb, tp, tt, tq, ti, max_num_neighbors, radius, delta_t_us, te, min_index);
    return (int)hipDeviceSynchronize();
}

// ev_graph.cu:241-276
int ref_insert_in_queue(const int32_t *indices, int N, const int32_t *unique_coords, const int32_t *cumsum_counts,
                        int Kpix, int32_t *queue, int B, int Q, int H, int W) {
    auto ti = view(indices, {N});
    auto tu = view(unique_coords, {Kpix});
    auto tc = view(cumsum_counts, {Kpix});
    auto tq = view(queue, {B, Q, H, W});
    insert_in_queue_cuda(ti, tu, tc, tq);
    return (int)hipDeviceSynchronize();
}

// ev_graph.cu:215-238
int ref_insert_in_queue_single(const int32_t *indices, const int32_t *events, int32_t *queue, int B, int Q, int H,
                               int W) {
    auto ti = view(indices, {1});
    auto te = view(events, {1, 3});
    auto tq = view(queue, {B, Q, H, W});
    insert_in_queue_single_cuda(ti, te, tq);
    return (int)hipDeviceSynchronize();
}
}
