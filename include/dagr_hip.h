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
 * 1:1 replacements of the reference's native module `ev_graph_cuda` (src/dagr/graph/ev_graph.cu:279-283),
 * same arguments in the same order, on the caller's own state (FIFO volume int32[B,Q,H,W] with -1 =
 * empty and q = 0 newest, timestamp log, -1-filled int64 edge buffer), so the Python classes of
 * src/dagr/graph/ev_graph.py work unchanged on top of them -- including reset=False / min_index > 0.
 *   dagr_fill_edges             <- fill_edges_cuda(batch, pos, all_timestamps, event_queue, indices,
 *                                  max_num_neighbors, radius, delta_t_us, edges, min_index)   ev_graph.cu:82-128
 *   dagr_insert_in_queue        <- insert_in_queue_cuda(indices, unique_coords, cumsum_counts, queue)   :241-276
 *   dagr_insert_in_queue_single <- insert_in_queue_single_cuda(indices, events, queue)                  :215-238
 * Shapes ride along as integers (N = len(batch), edges_cols = edges.size(1), queue dims).
 * ------------------------------------------------------------------------ */
int dagr_fill_edges(const int32_t *batch /*[N]*/, const int32_t *pos /*[N,3] x,y,t_us*/,
                    const int32_t *all_timestamps, const int32_t *event_queue /*[B,Q,H,W]*/,
                    const int32_t *indices /*[N]*/, int32_t max_num_neighbors, float radius, float delta_t_us,
                    int64_t *edges /*[2,edges_cols], in/out*/, int64_t edges_cols, int32_t min_index,
                    int64_t N, int32_t B, int32_t Q, int32_t H, int32_t W, void *stream);
int dagr_insert_in_queue(const int32_t *indices /*[N] pixel-sorted*/, const int32_t *unique_coords /*[num_pixels]*/,
                         const int32_t *cumsum_counts /*[num_pixels]*/, int64_t num_pixels, int32_t *queue,
                         int32_t B, int32_t Q, int32_t H, int32_t W, void *stream);
int dagr_insert_in_queue_single(const int32_t *indices /*[1]*/, const int32_t *event_xy /*>= [2]: x, y*/,
                                int32_t *queue, int32_t B, int32_t Q, int32_t H, int32_t W, void *stream);

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
/* must be called once on a fresh workspace (zeroes the per-key counters) */
int dagr_graph_workspace_init(const dagr_graph_desc *desc, void *workspace, size_t workspace_bytes, void *stream);

/* Node numbering: the graph comes out in *slot space* -- node n is the n-th event in (sample, y, x,
 * time) order (the builder's CSR-by-pixel order), so that spatial neighbours are memory
 * neighbours in every level-0 array and the events of a range of pixel rows are one run of nodes.  nbr_src holds node numbers.  dagr_graph_node_order exports the permutation,
 * dagr_graph_gather_inputs brings the per-event inputs into node order, dagr_graph_edge_index emits the
 * reference-shaped, event-ordered int64 edge_index.
 *
 * pos: pos_is_int32 == 0: normalised fp32 [N,3] exactly as format_data produces it (denormalised
 *      on the fly, ev_tgn.py:11-16); pos_is_int32 == 1: int32 [N,3] = (x, y, t_us), the input
 *      contract of SlidingWindowGraph.forward (ev_graph.py:139).
 * batch: int32[N] or int64[N] (sample index, non-decreasing not required).
 * nbr_src[N,K] int32 : source node of slot j of destination node n (slot 0 = n itself)
 * nbr_code[N,K] int16: (dx+r)*(2r+1) + (dy+r) with (dx,dy) = pixel offset source - destination
 * deg[N] int32       : number of valid slots (1..K)
 * Slots >= deg[n] are left untouched. */
int dagr_graph_build_window(const dagr_graph_desc *desc, void *workspace,
                            const void *pos, int32_t pos_is_int32,
                            const void *batch, int32_t batch_is_int64, int64_t N,
                            int32_t *nbr_src, int16_t *nbr_code, int32_t *deg, void *stream);

/* dagr_graph_build_window with the event count in DEVICE memory: every launch is sized for `n_cap` events and bounded by
 * *n_dev (<= n_cap) on the device, so the call can be captured in a HIP graph once and replayed for windows of any size.
 * pos / batch: the static buffers dagr_stage_window filled -- that launch has already run the build's first step
 * (denormalise + per-key count) on them and cleared the status words; this call continues from there. */
int dagr_graph_build_window_dev(const dagr_graph_desc *desc, void *workspace, const void *pos, int32_t pos_is_int32,
                                const void *batch, int32_t batch_is_int64, int64_t n_cap, const int32_t *n_dev,
                                int32_t *nbr_src, int16_t *nbr_code, int32_t *deg, void *stream);
/* device address of the number of nodes (indexed events) of the last build on this workspace: the `n_ptr` of the kernels
 * that follow it in a captured window */
const int32_t *dagr_graph_node_count_ptr(const dagr_graph_desc *desc, void *workspace);
/* One launch that copies a caller's window (format_data output: pos fp32[N,3], feat fp32[N], batch int32/int64[N]) into
 * static buffers (batch as int32), writes N to *n_dev, clears the builder's status words and runs the build's first step
 * (denormalise_pos + the per-key count, ev_tgn.py:11-16): the only per-window launch in front of a captured window graph
 * (dagr_graph_build_window_dev continues from its results). */
int dagr_stage_window(const dagr_graph_desc *desc, void *workspace, const float *pos, const float *feat, const void *batch,
                      int32_t batch_is_int64, int64_t N, float *pos_out, float *feat_out, int32_t *batch_out,
                      int32_t *n_dev, void *stream);

/* The neighbour search alone, again, on the pixel index the last dagr_graph_build_window left in `workspace` (same N):
 * rewrites nbr_src / nbr_code / deg and the edge count.  For measurement (bench.py times the search kernels on their own
 * stream with HIP events); a product caller has no use for it. */
int dagr_graph_search_window(const dagr_graph_desc *desc, void *workspace, int64_t N,
                             int32_t *nbr_src, int16_t *nbr_code, int32_t *deg, void *stream);

/* Device-side status words written by the last build on this workspace.
 * Synchronises `stream`.  flags bit0: event outside [0,W)x[0,H)x[0,B); bit1: internal list overflow.
 * num_edges = sum(deg). */
int dagr_graph_status(const dagr_graph_desc *desc, void *workspace, int64_t *num_edges /*host*/,
                      int32_t *flags /*host*/, void *stream);
/* All eight status words of the last build (synchronises `stream`): [0] pixels with more than min(64, Q) events, [1] flags,
 * [2..3] num_edges (uint64), [4] pixels beyond the FIFO depth, [5] destinations the row kernel deferred to the
 * position-centric walk (neighbourhoods of more than 320 candidates), [6] 1 when timestamps were not non-decreasing inside
 * a sample (every destination then takes the generic search), [7] destinations the row kernel answered from the inner rings
 * of their neighbourhood.  Tests use it to show which paths a window took. */
int dagr_graph_counters(const dagr_graph_desc *desc, void *workspace, int32_t *out8_host, void *stream);

/* Reference-shaped output: edge_index int64[2,E] in the order of
 * `edges[:, edges[1] >= 0]` (graph/utils.py:22), i.e. event ids, destinations ascending.
 * `row_stride` = allocated columns (>= E).  rowptr int32[N+1] receives the exclusive scan of the
 * per-event in-degree.
 * Asynchronous; E = rowptr[N]. scratch: int32[dagr_scan_scratch_elems(N+1)]. */
size_t dagr_scan_scratch_elems(int64_t n);
int dagr_graph_edge_index(const dagr_graph_desc *desc, void *workspace, const int32_t *nbr_src, const int32_t *deg,
                          int64_t N, int32_t *rowptr, int32_t *scan_scratch,
                          int64_t *edge_index, int64_t row_stride, void *stream);
/* The same edges as dagr_graph_edge_index, in the form the convolutions consume, without a host round trip for E:
 * rowptr int32[N+1] by destination EVENT, col int32[e_cap] source event ids, code int32[e_cap] = (dx + code_bias) |
 * (dy + code_bias) << 16 with (dx, dy) = source pixel - destination pixel (what T.Cartesian's attribute of the edge
 * encodes, transforms/cartesian in ev_tgn.py:39-58 + net.py:118-121); row e holds deg(e) entries, self loop first.
 * e_cap >= N * max_neighbors always suffices.  Asynchronous. */
int dagr_graph_csr_codes(const dagr_graph_desc *desc, void *workspace, const int32_t *nbr_src, const int16_t *nbr_code,
                         const int32_t *deg, int64_t N, int32_t code_bias, int32_t *rowptr, int32_t *scan_scratch,
                         int32_t *col, int32_t *code, int64_t e_cap, void *stream);
/* slot_event[n] = event id of node n, event_slot[e] = node of event e (either may be NULL) */
int dagr_graph_node_order(const dagr_graph_desc *desc, void *workspace, int64_t N, int32_t *slot_event,
                          int32_t *event_slot, void *stream);
/* node-ordered level-0 inputs: pos_nodes[N,3], batch_nodes[N] (int32), and per node the feature row
 * x0[n, col_feat] = feat[e], x0[n, col_pos..col_pos+1] = pos_xy[e]  (Net.forward's cat(x, pos[:, :2]), net.py:124-125;
 * the other columns are left for the sampled image features -- the column order of x0 is the caller's choice as long
 * as the packed weights follow it) */
int dagr_graph_gather_inputs(const dagr_graph_desc *desc, void *workspace, const float *pos, const float *feat,
                             int64_t N, float *pos_nodes, int32_t *batch_nodes, float *x0, int32_t ldx0,
                             int32_t col_feat, int32_t col_pos, void *stream);

/* dagr_graph_build_window (n_dev NULL) / dagr_graph_build_window_dev (n_dev given) on normalised fp32 positions that ALSO
 * writes the node-ordered level-0 inputs of dagr_graph_gather_inputs, from its last launch: one launch less per window. */
typedef struct dagr_l0_inputs {
    const float *feat;          /* fp32[N] per event (polarity)                       */
    float *pos_nodes;           /* fp32[N,3]                                          */
    int32_t *batch_nodes;       /* int32[N]                                           */
    float *x0;                  /* level-0 feature rows, row stride ldx0              */
    int32_t ldx0, col_feat, col_pos;
} dagr_l0_inputs;
int dagr_graph_build_window_inputs(const dagr_graph_desc *desc, void *workspace, const float *pos, const void *batch,
                                   int32_t batch_is_int64, int64_t N, const int32_t *n_dev, int32_t *nbr_src,
                                   int16_t *nbr_code, int32_t *deg, const dagr_l0_inputs *inputs, void *stream);

/* ------------------------------------------------------------------------ *
 * SplineConv (degree-1 open B-spline, 5x5 kernel, sum aggregation)
 *   replaces MySplineConv.forward/_forward/message_lut          model/layers/spline_conv.py:39-78
 *   over  torch_geometric SplineConv + ToSparseTensor, torch_spline_conv spline_basis /
 *   spline_weighting, torch_scatter segment_csr  (third-party, un-vendored: SURVEY.md 2.3),
 *   fused with BatchNorm(eval)+ReLU (model/layers/conv.py:23-28) and the skip branch (:47-56).
 *
 *   Edge offsets are integer LUT coordinates (ix, iy) in [0,2rx] x [0,2ry], exactly the index the
 *   reference's message_lut derives (spline_conv.py:41-42); den_x = fp32(2*Mx*width),
 *   den_y = fp32(2*My*height) as in init_lut (spline_conv.py:29-30).
 *   Weight packing (host side, BN folded: W*scale[o], shift[o]):
 *     level 0  : wpack[(tx*ty*cin taps | cin root | cskip skip)][16], taps = tx x ty window starting
 *                at (win_x, win_y) of the 5x5 kernel, tap row = (a + tx*b)*cin + i
 *     generic  : Wm[(25*cin taps | cin root | cskip skip)][cout], tap row = (kx + 5*ky)*cin + i
 * ------------------------------------------------------------------------ */
/* first kernel tap (0..4) and number of taps touched by offsets -r..r on one axis (host) */
int dagr_spline_tap_window(int32_t r, float den, int32_t *first_tap_host, int32_t *num_taps_host);
/* tab[(2rx+1)*(2ry+1)][ntp], ntp = round_up(tx*ty, 4): per offset code = ix*(2ry+1)+iy the window
 * products bx[a]*by[b] at [a + tx*b]; window = taps [win_x, win_x+tx) x [win_y, win_y+ty);
 * bad_flag (device int32) is set if an offset needs a tap outside the window */
int dagr_spline_l0_table(int32_t rx, int32_t ry, float den_x, float den_y, int32_t win_x, int32_t tx,
                         int32_t win_y, int32_t ty, float *tab, int32_t *bad_flag, void *stream);
/* fused level-0 conv on the builder's neighbour lists; cout is 16 (Net: int(base_width*32)).
 * out[n,0:16] = act(sum_j x[src_j] . What(code_j) + x[n] . root + xskip[n] . skip + shift) */
int dagr_spline_conv_l0(int32_t cin, int32_t cskip, int32_t ntaps /* tx*ty: 9, 15 or 25 */,
                        int64_t N, int32_t K, int32_t ncodes,
                        const int32_t *nbr_src, const int16_t *nbr_code, const int32_t *deg,
                        const float *x, int32_t ldx, const float *xskip, int32_t ldskip,
                        const float *tab, const float *wpack, const float *shift, int32_t relu,
                        float *out, int32_t ldo, void *stream);
/* level-0 conv as 16-node wave tiles (csrc/conv_l0_tiles.hip): same function as dagr_spline_conv_l0, but the basis is
 * evaluated from the offset domain (rx, ry, den_x, den_y; tap window [win_x, win_x+tx) x [win_y, win_y+ty)) instead of
 * a code table, and the input row is [cmain (0 or 16) channels | cextra (<= 4) channels], 16-byte aligned when cmain = 16.
 * wpack rows: [(a + tx*b)*cin + i | cin root | cskip skip] x 16 with i in the column order of x, cin = cmain + cextra.
 * K (neighbour-list stride) must be 16; (tx, ty) in {(3,3), (3,5), (5,3)}. */
int dagr_spline_conv_l0_tiles(int32_t cmain, int32_t cextra, int32_t cskip, int32_t win_x, int32_t tx, int32_t win_y,
                              int32_t ty, int32_t rx, int32_t ry, float den_x, float den_y, int64_t N, int32_t K,
                              const int32_t *nbr_src, const int16_t *nbr_code, const int32_t *deg, const float *x,
                              int32_t ldx, const float *xskip, int32_t ldskip, const float *wpack, const float *shift,
                              int32_t relu, float *out, int32_t ldo, const int32_t *n_ptr, void *stream);
/* n_ptr (device, may be NULL): the level's node count; the rows processed are min(N, *n_ptr - first_node) -- a launch
 * sized for a capacity N then serves windows of any size (captured HIP graphs).
 * the same on the nodes [first_node, first_node + N) only (all arrays are the full level's: an asynchronous update
 * computes the rows it appended; a node's result does not depend on which nodes share its tile) */
int dagr_spline_conv_l0_tiles_rows(int32_t cmain, int32_t cextra, int32_t cskip, int32_t win_x, int32_t tx, int32_t win_y,
                                   int32_t ty, int32_t rx, int32_t ry, float den_x, float den_y, int64_t first_node,
                                   int64_t N, int32_t K, const int32_t *nbr_src, const int16_t *nbr_code,
                                   const int32_t *deg, const float *x, int32_t ldx, const float *xskip, int32_t ldskip,
                                   const float *wpack, const float *shift, int32_t relu, float *out, int32_t ldo,
                                   const int32_t *n_ptr, void *stream);
/* generic step 1: A[n] = [sum_j basis*x_j per tap (25*cin) | x[n] (cin) | xskip[n] (cskip)] over a
 * CSR-by-destination graph; code[e] = ix | iy<<16.  n_nodes_ptr (device, may be NULL) bounds the
 * rows actually processed (<= n_nodes_max) without a host sync. */
int dagr_spline_tap_aggregate(const int32_t *n_nodes_ptr, int32_t n_nodes_max, const int32_t *rowptr,
                              const int32_t *col, const int32_t *code, const float *x, int32_t ldx,
                              int32_t cin, const float *xskip, int32_t ldskip, int32_t cskip,
                              int32_t rx, int32_t ry, float den_x, float den_y,
                              float *A, int32_t lda, void *stream);
/* backward of step 1 w.r.t. x (training path): grad_x[src] = sum over its out-edges of basis * grad_A[dst][tap], plus
 * grad_A[n][25 cin ..] (root copy).  Deterministic: the scatter adds 64-bit fixed-point integers (scale 2^50 /
 * *grad_A_absmax, a device scalar >= max |grad_A|; acc int64[n_nodes_max * cin], zeroed by the caller), so the result
 * does not depend on the order in which the atomics land; grad_x is overwritten.  The weight gradient of the
 * contraction is a plain GEMM (A^T . grad_out) on the A matrix of dagr_spline_tap_aggregate. */
int dagr_spline_tap_scatter_grad(const int32_t *n_nodes_ptr, int32_t n_nodes_max, const int32_t *rowptr,
                                 const int32_t *col, const int32_t *code, const float *grad_A, int32_t lda, int32_t cin,
                                 int32_t rx, int32_t ry, float den_x, float den_y, const float *grad_A_absmax,
                                 int64_t *acc, float *grad_x, int32_t ldg, void *stream);
/* The same gradient for narrow convs (cin, cout <= 16: the event level, 400 k rows per training step) straight from
 * grad_out[n, cout] and the weight matrix Wm[26 cin, cout] (rows: 25 taps x cin, then the root rows): grad_A's row of a
 * node is rebuilt in LDS instead of being written by a GEMM (0.67 GB for the 16 -> 16 conv), reduced for its maximum and
 * read back.  *grad_A_bound: a device scalar >= max |grad_out . Wm^T|, e.g. max|grad_out| * max_k sum_co |Wm[k, co]|. */
int dagr_spline_tap_scatter_grad_w(const int32_t *n_nodes_ptr, int32_t n_nodes_max, const int32_t *rowptr,
                                   const int32_t *col, const int32_t *code, const float *grad_out, int32_t ldg, int32_t cout,
                                   const float *Wm, int32_t ldw, int32_t cin, int32_t rx, int32_t ry, float den_x,
                                   float den_y, const float *grad_A_bound, int64_t *acc, float *grad_x, int32_t ldgx,
                                   void *stream);
/* fused steps 1+2: 16 aggregated rows live in LDS, no A matrix in HBM, one launch.  K = 26*cin + cskip beyond the tile
 * (~2270 floats) is cut at tap boundaries into passes over the same edges, the accumulators staying in registers;
 * dagr_spline_conv_fused_lds_bytes(cin, cskip) = the tile of the scheme chosen, > 160 KiB when a single tap
 * (cin floats, + cin + cskip in the last pass) does not fit.
 * Wq = the [K, N] matrix of dagr_gemm_bias_act re-packed on the host into MFMA operand order:
 *   Wq[c][g][l][j] = W[16 g + 4 j + (l >> 4)][16 c + (l & 15)],  c < ceil(N/16), g < ceil(K/16), l < 64, j < 4,
 * zero outside [K, N]; 16-byte aligned. */
size_t dagr_spline_conv_fused_lds_bytes(int32_t cin, int32_t cskip);
/* passes of the scheme (1 = the whole row in the tile; 0 = unsupported).  Every pass walks the edges again: measured on
 * MI355X the multi-pass form beats tap_aggregate + gemm up to ~1.5 k node slots (one launch instead of two, no A
 * matrix) and loses beyond (tools/microbench/head_ab.hip). */
int32_t dagr_spline_conv_fused_passes(int32_t cin, int32_t cskip);
int dagr_spline_conv_fused(const int32_t *n_nodes_ptr, int32_t n_nodes_max, const int32_t *rowptr,
                           const int32_t *col, const int32_t *code, const float *x, int32_t ldx, int32_t cin,
                           const float *xskip, int32_t ldskip, int32_t cskip, int32_t rx, int32_t ry,
                           float den_x, float den_y, const float *Wq, const float *bias, float *C, int32_t ldc,
                           int32_t N, int32_t relu, void *stream);
/* Two fused convs over the SAME graph in one launch (gridDim.z = 2): same row shape (cin channels, no skip input, row
 * stride ldx), each with its own input pointer, packed weights, bias, output pointer (row stride ldc) and width.  The
 * detection head's predictors: cls_pred on the cls_conv half of the fused row and reg_pred | obj_pred on the reg_conv half
 * (model/networks/dagr.py:183-189) -- on <= 1 260-row levels a launch is what a conv costs. */
int dagr_spline_conv_fused_pair(const int32_t *n_nodes_ptr, int32_t n_nodes_max, const int32_t *rowptr,
                                const int32_t *col, const int32_t *code, int32_t ldx, int32_t cin, int32_t rx, int32_t ry,
                                float den_x, float den_y, int32_t ldc, int32_t relu, const float *x_a, const float *Wq_a,
                                const float *bias_a, float *C_a, int32_t N_a, const float *x_b, const float *Wq_b,
                                const float *bias_b, float *C_b, int32_t N_b, void *stream);
/* 1 .. 4 INDEPENDENT fused convs in one launch (gridDim.z = job): each job is a full dagr_spline_conv_fused argument set
 * (its own graph, row shape, weights, output).  What a window offers at one point of its dependency graph -- head scale 1's
 * stem beside the first conv of layer5, its cls_conv | reg_conv beside the second, its predictors beside head scale 2's
 * stem (model/networks/net.py:166-186, dagr.py:213-236) -- costs one launch instead of a stream fork + join.  All jobs must
 * take the same form (dagr_spline_conv_fused_passes == 1 for all, or > 1 for all); DAGR_ERR_UNSUPPORTED otherwise. */
typedef struct dagr_conv_job {
    const int32_t *n_nodes_ptr;
    int32_t n_nodes_max;
    const int32_t *rowptr, *col, *code;
    const float *x;
    int32_t ldx, cin;
    const float *xskip;
    int32_t ldskip, cskip, rx, ry;
    float den_x, den_y;
    const float *Wq, *bias;
    float *C;
    int32_t ldc, N, relu;
} dagr_conv_job;
int dagr_spline_conv_fused_multi(const dagr_conv_job *jobs, int32_t count, void *stream);
/* generic step 2: C[M,N] = act(A[M,K] . Wm[K,N] + bias[N]); M = min(*m_ptr, m_max) */
int dagr_gemm_bias_act(const int32_t *m_ptr, int32_t m_max, const float *A, int32_t lda,
                       const float *Wm, int32_t ldw, const float *bias, float *C, int32_t ldc,
                       int32_t K, int32_t N, int32_t relu, void *stream);

/* ------------------------------------------------------------------------ *
 * Voxel-grid pooling
 *   replaces Pooling.forward / consecutive_cluster / round_to_pixel   model/layers/pooling.py:12-97
 *   over torch_cluster.grid_cluster, torch_scatter.scatter_max, PyG pool_pos / _avg_pool_x,
 *   T.Cartesian (third-party, un-vendored), plus the LUT index of the consuming SplineConvs
 *   (spline_conv.py:41-42).  Output graph: CSR by destination cluster, sources ascending,
 *   code[e] = ix | iy<<16.  All counts stay on the device (n_out, e_out; rowptr_out[n_out] = e_out).
 * ------------------------------------------------------------------------ */
typedef struct {
    int32_t batch_size;   /* B                                                          */
    int32_t channels;     /* C: feature channels pooled                                 */
    int32_t gx, gy;       /* voxels per axis = trunc(0.9999999/size)+1 (grid_cluster)   */
    float vx, vy;         /* voxel_size[0], voxel_size[1]  (pooling.py:24)              */
    float inv_w, inv_h;   /* wh_inv = 1/W, 1/H in fp32      (pooling.py:32)             */
    float two_max;        /* fp32(2*max_value) of this pooling's Cartesian transform    */
    float r00, r02, r11, r12; /* attr_remapping_matrix of the consuming convs (spline_conv.py:23-24) */
    int32_t rx, ry;       /* their LUT offset domain                                    */
    int32_t aggr;         /* 0 = max, 1 = mean                                          */
    int32_t append_pos;   /* also write pos[:, :2] into columns C, C+1 of each output row (net.py:137-138) */
} dagr_pool_desc;

size_t dagr_pool_workspace_bytes(const dagr_pool_desc *desc);
int dagr_pool_workspace_init(const dagr_pool_desc *desc, void *workspace, size_t workspace_bytes, void *stream);
/* level 0: pools the events of the window whose graph was just built on (gdesc, graph_ws).
 * xlo[gx+1] / ylo[gy+1]: first pixel column/row of each voxel (device int32; from the same fp32
 * division as grid_cluster).  x_out rows start at column xoff; outputs sized for
 * T = gx*gy*(B+1) clusters; rowptr_out has T+2 entries; e_cap = capacity of col_out/code_out.
 * nbr_code (the offset codes dagr_graph_build_window wrote next to nbr_src) enables the coarse-edge fast
 * path: with radius <= 2 voxels the source voxels of a voxel's in-edges are a 5x5 bitmap collected while it
 * is pooled; windows containing t == 1.0 events (whose cluster ids leave the pixel grid) and NULL fall back
 * to the generic per-edge set insertion.  Same result either way. */
int dagr_pool_l0(const dagr_pool_desc *desc, void *pool_ws, const dagr_graph_desc *gdesc, void *graph_ws,
                 const int32_t *xlo, const int32_t *ylo, const float *x, int32_t ldx,
                 const float *pos_nodes, const int32_t *batch_nodes /* node order */,
                 const void *batch, int32_t batch_is_int64 /* event order */, int64_t N,
                 const int32_t *nbr_src, const int16_t *nbr_code /* may be NULL */, const int32_t *deg,
                 int32_t *cluster_scratch /*[N]*/, float *x_out, int32_t ldo, int32_t xoff, float *pos_out,
                 int32_t *batch_out, int32_t *n_out, int32_t *rowptr_out, int32_t *col_out, int32_t *code_out,
                 int32_t *e_out, int32_t e_cap, void *stream);
/* ------------------------------------------------------------------------ *
 * Asynchronous operation (reset=False): a micro-batch of events joins the resident window
 *   replaces, for the events-to-level-1 part, the reference's incremental update
 *     EV_TGN.forward(reset=False) -> AsyncGraph.forward            model/layers/ev_tgn.py:45-56, graph/ev_graph.py:63-103
 *     insert_in_queue_cuda_kernel / fill_edges_cuda_kernel (min_index > 0)   graph/ev_graph.cu:15-80,169-212
 *     asynchronous conv on the new nodes (graph_new_nodes branch)  asynchronous/conv.py:107-125,196-207
 *     asynchronous pooling, new-node branch (cluster caches)       asynchronous/max_pool.py:126-158
 *   Edges point from older to newer events: the rows of the window's events never change, an update appends rows.
 *   State owned by the caller (device): app_head int32[B*H*W] (-1 = empty), app_next int32[capacity],
 *   app_xytb int32[capacity][4]; event ids continue the window's (first_id = n_static + number appended so far); an
 *   appended event's node row is its id.  status int32[>=1]: bit 0 = event outside the sensor (self loop only).
 * ------------------------------------------------------------------------ */
int dagr_async_graph_append(const dagr_graph_desc *desc, void *graph_ws, int64_t n_static, int64_t first_id,
                            int32_t *app_head, int32_t *app_next, int32_t *app_xytb, int64_t capacity,
                            const void *pos /* fp32 normalised or int32 [n,3] */, int32_t pos_is_int32,
                            const void *batch, int32_t batch_is_int64, int64_t n_new,
                            int32_t *nbr_src, int16_t *nbr_code, int32_t *deg /* the level's full arrays */,
                            int32_t *status,
                            /* optional (x0 != NULL, fp32 pos): the node-ordered level-0 inputs of the new rows, as
                             * dagr_graph_gather_inputs writes them for a window, and the sample index by event id */
                            const float *feat, float *pos_nodes, int32_t *batch_nodes, int32_t *batch_events, float *x0,
                            int32_t ldx0, int32_t col_feat, int32_t col_pos, void *stream);
/* One asynchronous update as ONE call: dagr_async_graph_append (with the node-ordered inputs), conv_block1 on the n new rows
 * (dagr_spline_conv_l0_tiles_rows twice) and dagr_pool_l0_stream, issued back to back from native code.  Same kernels, same
 * arguments as the four calls; what it removes is the host time between their launches (an update of a few events is
 * ~8 launches of a few microseconds each: issued one by one from the caller's language the GPU waits for the host).
 * Events-only rows (polarity | pos_xy: cin1 = 3). */
typedef struct dagr_async_update_args {
    const dagr_graph_desc *gdesc; void *graph_ws; int64_t n_static, first_id;
    int32_t *app_head, *app_next, *app_xytb; int64_t capacity;
    const float *pos; const void *batch; int32_t batch_is_int64; int64_t n_new;
    int32_t *nbr_src; int16_t *nbr_code; int32_t *deg; int32_t *status;
    const float *feat; float *pos_nodes; int32_t *batch_nodes, *batch_events; float *x0; int32_t ldx0, col_feat, col_pos;
    int32_t win_x, tx, win_y, ty, rx, ry; float den_x, den_y;
    int32_t cin1; const float *w1, *s1; float *h1; int32_t ldh1;
    const float *w2, *s2; float *hp0; int32_t ldhp0;
    const dagr_pool_desc *pdesc; void *pool_ws; const int32_t *xlo, *ylo;
    float *x_out; int32_t ldo; float *pos_out; int32_t *batch_out, *n_out, *rowptr_out, *col_out, *code_out, *e_out;
    int32_t e_cap;
} dagr_async_update_args;
int dagr_async_update(const dagr_async_update_args *args, void *stream);
/* Launch (A) of dagr_pool_l0 alone: the level-0 accumulation kernel over the window last built on `graph_ws`, into the
 * accumulators of `pool_ws` (max / fixed-point sums: repeating it leaves max accumulators unchanged and scales the sums;
 * the next dagr_pool_l0 call re-arms everything).  For measurement (bench.py times the kernel on its own with HIP
 * events); a product caller has no use for it. */
int dagr_pool_l0_accumulate(const dagr_pool_desc *desc, void *pool_ws, const dagr_graph_desc *gdesc, void *graph_ws,
                            const int32_t *xlo, const int32_t *ylo, const float *x, int32_t ldx, const float *pos,
                            int64_t N, const int32_t *nbr_src, const int16_t *nbr_code, const int32_t *deg, void *stream);
/* pool1 with RESIDENT accumulators (its own workspace, dagr_pool_workspace_bytes): rebuild != 0 recomputes them from the
 * window (nodes [0, n_window) through the builder's pixel index); rows [first_row, first_row + n_rows) -- appended
 * nodes, first_row >= n_window -- are added; then level 1 is emitted exactly as dagr_pool_l0 emits it, and the
 * accumulators stay.  batch_events: sample index by EVENT id (window events in event order, then the appended ones).
 * Needs the cell-bitmap form of the coarse edges (search radius <= two cells). */
int dagr_pool_l0_stream(const dagr_pool_desc *desc, void *pool_ws, int32_t rebuild, const dagr_graph_desc *gdesc,
                        void *graph_ws, const int32_t *xlo, const int32_t *ylo, const float *x, int32_t ldx, const float *pos,
                        const int32_t *batch_events, int64_t n_window, int64_t first_row, int64_t n_rows,
                        const int32_t *nbr_src, const int16_t *nbr_code, const int32_t *deg, float *x_out, int32_t ldo,
                        int32_t xoff, float *pos_out, int32_t *batch_out, int32_t *n_out, int32_t *rowptr_out,
                        int32_t *col_out, int32_t *code_out, int32_t *e_out, int32_t e_cap, void *stream);

/* dagr_spline_conv_fused + launch (A) of the pooling step that consumes its output (dagr_pool_csr's accumulation) in one
 * launch: the epilogue merges every output element into its cluster's accumulator (the values it holds in registers),
 * wave w does the bookkeeping and the in-edges of node w.  Single-pass form only (dagr_spline_conv_fused_passes == 1);
 * pdesc->channels must equal N.  Follow with dagr_pool_csr(..., n_max = 0, ...): scan + emit only.  pos / batch: the
 * level's node positions and samples; cluster_scratch int32[n_nodes_max]. */
int dagr_spline_conv_fused_pool(const int32_t *n_nodes_ptr, int32_t n_nodes_max, const int32_t *rowptr,
                                const int32_t *col, const int32_t *code, const float *x, int32_t ldx, int32_t cin,
                                const float *xskip, int32_t ldskip, int32_t cskip, int32_t rx, int32_t ry,
                                float den_x, float den_y, const float *Wq, const float *bias, float *C,
                                int32_t ldc, int32_t N, int32_t relu, const dagr_pool_desc *pdesc, void *pool_ws,
                                const float *pos, const int32_t *batch, int32_t *cluster_scratch, void *stream);
/* coarser levels: input graph in CSR; n_ptr (device) = number of valid input nodes (<= n_max) */
int dagr_pool_csr(const dagr_pool_desc *desc, void *pool_ws, const int32_t *n_ptr, int32_t n_max, const float *x,
                  int32_t ldx, const float *pos, const int32_t *batch, const int32_t *rowptr, const int32_t *col,
                  int32_t *cluster_scratch /*[n_max]*/, float *x_out, int32_t ldo, int32_t xoff, float *pos_out,
                  int32_t *batch_out, int32_t *n_out, int32_t *rowptr_out, int32_t *col_out, int32_t *code_out,
                  int32_t *e_out, int32_t e_cap, void *stream);
/* LUT coordinates of an existing pooled level (CSR + pos as written by dagr_pool_*) for a consumer whose table
 * covers ANOTHER domain: DAGR.cache_luts (dagr.py:52-62) gives head "1" the pool3 table even when num_scales = 1
 * makes it consume out4.  code_out[e] = ix | iy<<16 with ix = trunc(attr_x*r00 + r02 + 1e-3) (message_lut,
 * spline_conv.py:41-42), attr = (pos[src]-pos[dst])/two_max + 0.5 of THIS level's Cartesian transform.
 * status (device int32): bit3 set when a coordinate leaves [0,2rx] x [0,2ry]. */
int dagr_pool_recode(const int32_t *n_ptr, int32_t n_max, const int32_t *rowptr, const int32_t *col, const float *pos,
                     float two_max, float r00, float r02, float r11, float r12, int32_t rx, int32_t ry,
                     int32_t *code_out, int32_t e_cap, int32_t *status, void *stream);
/* flags bit0: node outside the voxel grid; bit1: > 64 distinct sources for one cluster;
 * bit2: edge capacity exceeded; bit3: LUT coordinate out of range.  Synchronises `stream`. */
int dagr_pool_status(const dagr_pool_desc *desc, void *pool_ws, int32_t *flags_host, void *stream);
/* device address of that status word (inside pool_ws): a caller that reads other results back anyway (the training path
 * reads the two output counts of dagr_pool_csr) fetches it in the same transfer instead of a synchronisation of its own */
const int32_t *dagr_pool_status_ptr(const dagr_pool_desc *desc, void *pool_ws);
/* All eight status words (synchronises `stream`): [0] the flags above, [4] accumulator epoch, [5] level-0 nodes merged
 * through the global path so far (outside the streaming kernel's LDS window, or t == 1.0 nodes; cumulative). */
int dagr_pool_counters(const dagr_pool_desc *desc, void *pool_ws, int32_t *out8_host, void *stream);

/* ------------------------------------------------------------------------ *
 * to_dense  -- model/layers/spline_conv.py:80-107 (SplineConvToDense tail)
 *   dense[B,C,Hc,Wc] (fully written, zeros where no node); cell = trunc(pos_xy / voxel_xy);
 *   winner_scratch int32[B*Hc*Wc]; status (device int32) bit0 = node outside the map.
 * ------------------------------------------------------------------------ */
int dagr_to_dense(const int32_t *n_ptr, int32_t n_max, const float *x, int32_t ldx, int32_t channels,
                  const float *pos, const int32_t *batch, float vx, float vy, int32_t batch_size,
                  int32_t Hc, int32_t Wc, int32_t *winner_scratch, float *dense, int32_t *status, void *stream);

/* ------------------------------------------------------------------------ *
 * training path (scripts/train_ncaltech101.py:41-74): backward halves the reference gets from
 * torch_scatter's / torch's autograd.  The forward entry points above are reused unchanged.
 * ------------------------------------------------------------------------ */
/* Pooling's feature aggregation, max (pooling.py:74-75, torch_scatter.scatter_max): arg[c, ch] = lowest node index
 * among cluster c's members whose x equals x_pooled[c, ch] (the CPU reducer's first maximum).  cluster int32[n] =
 * consecutive cluster id of each node (-1: outside the grid); arg int32[n_clusters, channels]. */
int dagr_pool_argmax(const int32_t *cluster, int32_t n, const float *x, int32_t ldx, int32_t channels,
                     const float *x_pooled, int32_t ldp, int32_t n_clusters, int32_t *arg, void *stream);
/* grad_x[n, ch] = grad_pooled[cluster[n], ch] where arg[cluster[n], ch] == n, else 0 (aggr 0, max), or
 * grad_pooled[cluster[n], ch] / count[cluster[n]] (aggr 1, mean: _avg_pool_x, pooling.py:77).  Fully written. */
int dagr_pool_grad(const int32_t *cluster, int32_t n, int32_t channels, int32_t aggr, const int32_t *arg,
                   const int32_t *count, const float *grad_pooled, int32_t ldg, float *grad_x, int32_t ldgx,
                   void *stream);
/* to_dense (spline_conv.py:80-107) backward: grad_x[n, :] = grad_dense[batch[n], :, cy, cx] for every node inside the
 * map (0 outside) -- INCLUDING nodes whose row a later node of the same cell overwrote in the forward: torch's
 * index_put backward gathers grad[indices] for all written rows, which is what the reference back-propagates. */
int dagr_to_dense_grad(int32_t n, int32_t channels, const float *pos, const int32_t *batch, float vx, float vy,
                       int32_t batch_size, int32_t Hc, int32_t Wc, const float *grad_dense,
                       float *grad_x, int32_t ldgx, void *stream);

/* y = relu(y + z) in place over n floats (16-byte aligned buffers of identical layout): the residual join of the
 * image branch's ResNet blocks (reference: torchvision Bottleneck/BasicBlock.forward, used by
 * src/dagr/model/networks/net_img.py:42-86) in one pass instead of add + clamp. */
int dagr_add_relu(float *y, const float *z, int64_t n, void *stream);
/* y = relu(y + bias[c]) in place over a channels-last map of n floats, C channels (C % 4 == 0): bias + ReLU after a
 * bias-free convolution in one pass (torchvision block forward: bnX folded into convX, then relu). */
int dagr_bias_relu(float *y_nhwc, const float *bias, int64_t n, int32_t C, void *stream);
/* same with SiLU (v / (1 + exp(-v))): yolox BaseConv (conv -> BN -> SiLU) of the CNN head (dagr.py:106-122). */
int dagr_bias_silu(float *y_nhwc, const float *bias, int64_t n, int32_t C, void *stream);
/* ResNet stem tail in one pass over channels-last maps: y[B, OH, OW, C] = maxpool 3x3 / stride 2 / pad 1 of
 * relu(x[B, H, W, C] * scale[c] + shift[c]), OH = (H-1)/2 + 1 (torchvision ResNet.forward bn1 -> relu -> maxpool,
 * net_img.py:80-84; eval-mode BatchNorm as an affine pair).  C % 4 == 0, 16-byte aligned buffers. */
int dagr_bn_relu_maxpool(const float *x_nhwc, int32_t B, int32_t H, int32_t W, int32_t C, const float *scale,
                         const float *shift, float *y_nhwc, void *stream);

/* ------------------------------------------------------------------------ *
 * sample_features (--use_image) -- model/networks/net.py:15-17,193-221
 *   out[n, coff:coff+C] = trilinear grid_sample (align_corners=True) of feat at node n's position;
 *   feat is channels-last fp32 [B,h,w,C]; width/height = sensor size used for normalisation.
 * ------------------------------------------------------------------------ */
int dagr_sample_features(const int32_t *n_ptr, int32_t n_max, const float *pos, const void *batch,
                         int32_t batch_is_int64, const float *feat_nhwc, int32_t B, int32_t h, int32_t w, int32_t C,
                         int32_t width, int32_t height, float *out, int32_t ldo, int32_t coff, void *stream);

/* ------------------------------------------------------------------------ *
 * Detection post-processing -- model/utils.py:25-33,61-110 (batched_nms_coordinate_trick over
 * torchvision.ops.nms, called per image from a Python loop)
 *   boxes[B,A,4] (x1,y1,x2,y2), scores[B,A], cls[B,A] (class id; boxes are shifted by cls*class_offset
 *   before the IoU test), valid[B,A] (uint8).  order_out[B,A]: anchor indices by descending score
 *   (invalid last), keep_out[B,A]: 1 where order_out[i] survives greedy NMS (IoU > thr suppressed),
 *   n_keep[B].  A <= 1024.
 * ------------------------------------------------------------------------ */
int dagr_nms_batched(const float *boxes, const float *scores, const int32_t *cls, const uint8_t *valid,
                     int32_t B, int32_t A, float iou_threshold, float class_offset,
                     int32_t *order_out, int32_t *keep_out, int32_t *n_keep, void *stream);

/* to_dense of the head scales (model/layers/spline_conv.py:80-107: cell = trunc(pos_xy / voxel_xy), highest node index
 * wins a shared cell) + the image branch's logits (model/networks/dagr.py:219-222,230-234) + collect_outputs /
 * decode_outputs (dagr.py:283-312) in ONE launch.  Per scale: the predictor rows pred[n, ld] (reg 4 | obj 1 | cls C),
 * node positions / samples, the map geometry, optionally the CNN head's reg / obj / cls maps (element strides for
 * [b, c, y, x]; NULL = events only) and optionally a dense[B, 5 + C, Hc, Wc] buffer that receives the fused logit maps.
 * out[B, A, 5 + C] as dagr_decode_heads.  scale1 may be NULL (num_scales = 1).  status bit0: node outside the map. */
typedef struct dagr_head_scale {
    const int32_t *n_ptr; int32_t n_max;
    const float *pred; int32_t ld;
    const float *pos; const int32_t *batch;
    float vx, vy, stride; int32_t Hc, Wc;
    const float *cnn[3]; int32_t cnn_stride[3][4];
    float *dense;
} dagr_head_scale;
int dagr_heads_finish(const dagr_head_scale *scale0, const dagr_head_scale *scale1, int32_t batch_size, int32_t channels,
                      float *out, int32_t *status, void *stream);
/* The same launch followed, inside it, by dagr_postprocess of every image (model/utils.py:61-110; dagr.py:94-95: the eval
 * forward post-processes its decoded outputs at once): one workgroup per image decodes the image's rows and runs its
 * confidence mask + class-offset NMS on them.  det[B, A, 6] / n_keep[B] as dagr_postprocess; `out` is still written. */
int dagr_heads_finish_detect(const dagr_head_scale *scale0, const dagr_head_scale *scale1, int32_t batch_size,
                             int32_t channels, float *out, int32_t *status, float conf_threshold, float iou_threshold,
                             float class_offset, float *det, int32_t *n_keep, void *stream);
/* collect_outputs + decode_outputs of the eval head (model/networks/dagr.py:283-312; grid/stride cache of
 * model/utils.py:119-134) for one or two scales in one launch: dense logit maps [B, channels = 5+C, Hs, Ws] (reg | obj |
 * cls) -> out[B, A, channels], A = H0*W0 (+ H1*W1), xy = (logit + cell) * stride, wh = exp(logit) * stride, the rest
 * sigmoid.  dense1 may be NULL (num_scales = 1). */
int dagr_decode_heads(const float *dense0, int32_t H0, int32_t W0, float stride0, const float *dense1, int32_t H1,
                      int32_t W1, float stride1, int32_t B, int32_t channels, float *out, void *stream);

/* The whole of postprocess_network_output (model/utils.py:61-110, filtering=True) for a window batch in ONE launch:
 * pred[B,A,5+C] = decoded head outputs (cx, cy, w, h, obj, cls...) -> cxcywh->xyxy in the reference's op order,
 * class max / argmax, score = obj*class_conf, mask (score*class_conf >= conf_threshold), class-offset greedy NMS.
 * det[B,A,6]: the survivors of image b are rows 0..n_keep[b]-1 = (x1, y1, x2, y2, score, label) in descending score
 * (ties: ascending anchor); rows beyond are unspecified.  A <= 1024. */
int dagr_postprocess(const float *pred, int32_t B, int32_t A, int32_t num_classes, float conf_threshold,
                     float iou_threshold, float class_offset, float *det, int32_t *n_keep, void *stream);

/* ------------------------------------------------------------------------ *
 * 1:1 replacements of the reference's native module `asy_tools` (src/dagr/asynchronous/asy_tools/main.cu:239-244): the
 * masked row operators of the asynchronous per-event network update.  indices int64[K] = rows (nodes) that changed;
 * feature matrices are [num_nodes, C] fp32 row-major; only the selected rows are read / written, in place.
 *   dagr_masked_lin          <- masked_lin(indices, x_in, x_out, weight[Cout,Cin], bias[Cout], add)      main.cu:220-236
 *   dagr_masked_lin_no_bias  <- masked_lin_no_bias(indices, x_in, x_out, weight, add)                    main.cu:198-216
 *   dagr_masked_isdiff       <- masked_isdiff(indices, x_old, x_new, atol, rtol): marks unchanged rows with -1 in
 *                               `indices`; the caller compacts (`indices[indices > -1]`, main.cu:138)     main.cu:112-139
 *   dagr_masked_inplace_BN   <- masked_inplace_BN(indices, x, x_out, mean, var, weight, bias, eps)       main.cu:69-96
 * Same argument order as the reference, then the shapes (K, Cin, Cout / C) and the stream.  add != 0: accumulate onto
 * x_out instead of overwriting it.
 * ------------------------------------------------------------------------ */
int dagr_masked_lin(const int64_t *indices, const float *x_in, float *x_out, const float *weight, const float *bias,
                    int32_t add, int64_t K, int32_t Cin, int32_t Cout, void *stream);
int dagr_masked_lin_no_bias(const int64_t *indices, const float *x_in, float *x_out, const float *weight, int32_t add,
                            int64_t K, int32_t Cin, int32_t Cout, void *stream);
int dagr_masked_isdiff(int64_t *indices, const float *x_old, const float *x_new, float atol, float rtol, int64_t K,
                       int32_t C, void *stream);
int dagr_masked_inplace_BN(const int64_t *indices, const float *x, float *x_out, const float *running_mean,
                           const float *running_var, const float *weight, const float *bias, float eps, int64_t K,
                           int32_t C, void *stream);

/* ------------------------------------------------------------------------ *
 * Event-stream downsampling -- scripts/downsample_events.py:91-124 (downsample_events / _filter_events_resize)
 *   Events grouped by output cell (x / fx, y / fy) in time order: order[N] = event ids sorted (stably) by cell,
 *   run_cell[n_runs] / run_end[n_runs] = the cells that occur and the cumulative end of their runs in `order`
 *   (torch.sort + unique_consecutive + cumsum, as graph/utils.py:9-13 prepares pixels).  polarity int8[N] in {-1,+1};
 *   change_map fp32[Ho*Wo] = the integrator state, carried from chunk to chunk; keep[N] (uint8) = 1 where the event
 *   passes.  The surviving events keep their order; their coordinates are x / fx, y / fy.
 * ------------------------------------------------------------------------ */
int dagr_downsample_events(const int32_t *order, const int32_t *run_cell, const int32_t *run_end, int32_t n_runs,
                           const int8_t *polarity, int32_t fx, int32_t fy, float *change_map, uint8_t *keep,
                           void *stream);

/* The 1x1 convolutions of the channels-last image branch (src/dagr/model/networks/net_img.py:42-48, BatchNorm folded) as
 * library GEMMs (hipBLASLt, fp32 in / fp32 accumulate) with the whole epilogue in the kernel:
 *   D[M, N] = act(A[M, K] . Wt[K, N] + bias[N] (+ R[M, N])),  row-major, row strides lda / ldr / ldd; act 0 = none, 1 = ReLU;
 * bias and R may be NULL.  The residual join + ReLU of a bottleneck (relu(bn3(conv3(x)) + identity)) is one launch.
 * workspace: device scratch of at least dagr_gemm_epilogue_workspace_bytes() bytes (caller-owned, no hidden allocation). */
size_t dagr_gemm_epilogue_workspace_bytes(void);
int dagr_gemm_epilogue(const float *A, int64_t M, int32_t K, int64_t lda, const float *Wt, int32_t N, const float *bias,
                       const float *R, int64_t ldr, int32_t act, float *D, int64_t ldd, void *workspace,
                       size_t workspace_bytes, void *stream);

/* Host-side helper: first n offsets of the search spiral (spiral.h:1-15), the closed form the
 * search kernel uses.  dx/dy are HOST arrays.  Lets CPU-only tests pin the visiting order. */
int dagr_spiral_offsets(int32_t n, int32_t *dx_host, int32_t *dy_host);

#ifdef __cplusplus
}
#endif
#endif /* DAGR_HIP_H */
