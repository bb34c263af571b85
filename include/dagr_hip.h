/*
 * dagr_hip.h -- C ABI of libdagr_hip.so, the MI355X (gfx950) engine for DAGR's
 * event-graph hot path.
 *
 * Conventions (all entry points)
 *   - plain C: pointers + sizes, no torch / HIP types.  `stream` is a
 *     hipStream_t passed as void* (NULL = the null stream).
 *   - every pointer is a DEVICE pointer owned by the caller unless marked
 *     "host".  The library never allocates device memory: callers query
 *     *_workspace_bytes() and hand in a workspace.
 *   - returns 0 on success, a negative dagr_status otherwise;
 *     dagr_last_error() returns a thread-local message (the Python binding
 *     turns it into RuntimeError, mirroring the reference's AT_ASSERTM ->
 *     RuntimeError convention, ev_graph.cu:9-12).
 *   - kernels are enqueued asynchronously on `stream`; nothing synchronises
 *     unless the entry point says so.
 *
 * Reference interfaces replaced (paths relative to uzh-rpg/dagr @ 2025-02-02):
 *   src/dagr/graph/ev_graph.cu:279-283   pybind module `ev_graph_cuda`
 *   src/dagr/graph/utils.py:6-23         host prep around it (sort/unique/cumsum, mask compaction)
 *   src/dagr/graph/ev_graph.py:18-166    AsyncGraph / SlidingWindowGraph state
 *   src/dagr/model/layers/ev_tgn.py:11-16 denormalize_pos
 *   src/dagr/utils/buffers.py:33-44      format_data
 *   third-party ops named at each section below.
 */
#ifndef DAGR_HIP_H
#define DAGR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    DAGR_OK = 0,
    DAGR_ERR_INVALID_ARG = -1,
    DAGR_ERR_HIP = -2,
    DAGR_ERR_WORKSPACE = -3,
    DAGR_ERR_UNSUPPORTED = -4
} dagr_status;

const char *dagr_last_error(void);
/* library / ABI version: major*10000 + minor*100 + patch */
int dagr_version(void);
/* number of visible HIP devices (<0 on error) -- lets bindings fail loudly early */
int dagr_device_count(void);

/* ------------------------------------------------------------------------ *
 * (a1) format_data  -- src/dagr/utils/buffers.py:33-44
 *   pos_out[n] = { x/W, y/H, t/T } as fp32 true division of fp32-converted ints,
 *   feat_out[n] = (float)p.
 * ------------------------------------------------------------------------ */
int dagr_format_events(const int16_t *xy /*[N,2]*/, const int32_t *t /*[N]*/, const int8_t *p /*[N]*/,
                       int64_t N, int32_t width, int32_t height, int32_t time_window,
                       float *pos_out /*[N,3]*/, float *feat_out /*[N]*/, void *stream);

/* ------------------------------------------------------------------------ *
 * Window graph builder (reset=True semantics)
 *   replaces, for one self-contained window, the sequence
 *     SlidingWindowGraph.reset()                       ev_graph.py:52-60
 *     denormalize_pos                                  ev_tgn.py:11-16
 *     _insert_events_into_queue + insert_in_queue_*    graph/utils.py:6-18, ev_graph.cu:130-212
 *     _search_for_edges + fill_edges_cuda_kernel       graph/utils.py:20-23, ev_graph.cu:15-80
 *   Output is the reference's edge set in the reference's order (self loop
 *   first, then spiral discovery order, <= max_neighbors per destination),
 *   stored as fixed-stride neighbour lists instead of a -1-padded int64 buffer.
 * ------------------------------------------------------------------------ */
typedef struct {
    int32_t width;          /* W, pixels                                  */
    int32_t height;         /* H, pixels                                  */
    int32_t batch_size;     /* B, samples per window batch                */
    int32_t max_neighbors;  /* K, incl. the self loop (reference: 16)     */
    int32_t queue_size;     /* Q, per-pixel FIFO depth (reference: 128)   */
    int32_t radius;         /* r, pixels: int(radius*W+1), ev_tgn.py:29   */
    int32_t delta_t_us;     /* int(radius*time_window),  ev_tgn.py:28     */
    int32_t time_window;    /* T, us (normaliser of pos[:,2])             */
    int64_t max_events;     /* capacity N_max of the workspace            */
} dagr_graph_desc;

size_t dagr_graph_workspace_bytes(const dagr_graph_desc *desc);
/* must be called once on a fresh workspace (zeroes the per-pixel counters) */
int dagr_graph_workspace_init(const dagr_graph_desc *desc, void *workspace, size_t workspace_bytes, void *stream);

/* pos: pos_is_int32 == 0: normalised fp32 [N,3] exactly as format_data produces it (denormalised
 *      on the fly, ev_tgn.py:11-16); pos_is_int32 == 1: int32 [N,3] = (x, y, t_us), the input
 *      contract of SlidingWindowGraph.forward (ev_graph.py:139).
 * batch: int32[N] or int64[N] (sample index, non-decreasing not required).
 * nbr_src[N,K] int32 : source event of slot j of destination n (slot 0 = n itself)
 * nbr_code[N,K] int16: (dx+r)*(2r+1) + (dy+r) with (dx,dy) = pixel offset source - destination
 * deg[N] int32       : number of valid slots (1..K)
 * Slots >= deg[n] are left untouched. */
int dagr_graph_build_window(const dagr_graph_desc *desc, void *workspace,
                            const void *pos, int32_t pos_is_int32,
                            const void *batch, int32_t batch_is_int64, int64_t N,
                            int32_t *nbr_src, int16_t *nbr_code, int32_t *deg, void *stream);

/* Device-side status words written by the last build on this workspace.
 * Synchronises `stream`.  flags bit0: event outside [0,W)x[0,H)x[0,B); bit1: internal list overflow.
 * num_edges = sum(deg). */
int dagr_graph_status(const dagr_graph_desc *desc, void *workspace, int64_t *num_edges /*host*/,
                      int32_t *flags /*host*/, void *stream);

/* Reference-shaped output: edge_index int64[2,E] in the order of
 * `edges[:, edges[1] >= 0]` (graph/utils.py:22).  `row_stride` = allocated columns (>= E).
 * rowptr int32[N+1] receives the exclusive scan of deg (also useful as CSR-by-destination).
 * Asynchronous; E = rowptr[N]. scratch: int32[dagr_scan_scratch_elems(N+1)]. */
size_t dagr_scan_scratch_elems(int64_t n);
int dagr_graph_edge_index(const int32_t *nbr_src, const int32_t *deg, int64_t N, int32_t K,
                          int32_t *rowptr, int32_t *scan_scratch,
                          int64_t *edge_index, int64_t row_stride, void *stream);

/* Host-side helper: first n offsets of the search spiral (spiral.h:1-15), the closed form the
 * search kernel uses.  dx/dy are HOST arrays.  Lets CPU-only tests pin the visiting order. */
int dagr_spiral_offsets(int32_t n, int32_t *dx_host, int32_t *dy_host);

#ifdef __cplusplus
}
#endif
#endif /* DAGR_HIP_H */
