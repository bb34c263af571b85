"""ctypes binding of libdagr_hip.so (the C ABI declared in include/dagr_hip.h).

The product path has no CPU fallback: if the shared library is missing or a call fails, a
RuntimeError is raised (the reference raises RuntimeError through AT_ASSERTM for bad inputs,
``src/dagr/graph/ev_graph.cu:9-12``).  PyTorch only supplies device memory and streams here; every
argument crossing this boundary is a raw pointer or an integer.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# builder knob: DAGR_HIP_LIB=<path> loads an experimental build of the same library (A/B runs of kernel variants)
LIB_PATH = os.environ.get("DAGR_HIP_LIB") or os.path.join(_HERE, "lib", "libdagr_hip.so")

c_void_p = ctypes.c_void_p
c_i32 = ctypes.c_int32
c_i64 = ctypes.c_int64
c_size_t = ctypes.c_size_t
c_float = ctypes.c_float


class PoolDesc(ctypes.Structure):
    """``dagr_pool_desc`` (include/dagr_hip.h)."""
    _fields_ = [("batch_size", c_i32), ("channels", c_i32), ("gx", c_i32), ("gy", c_i32), ("vx", c_float),
                ("vy", c_float), ("inv_w", c_float), ("inv_h", c_float), ("two_max", c_float), ("r00", c_float),
                ("r02", c_float), ("r11", c_float), ("r12", c_float), ("rx", c_i32), ("ry", c_i32), ("aggr", c_i32),
                ("append_pos", c_i32)]


class GraphDesc(ctypes.Structure):
    """``dagr_graph_desc`` (include/dagr_hip.h)."""
    _fields_ = [("width", c_i32), ("height", c_i32), ("batch_size", c_i32), ("max_neighbors", c_i32),
                ("queue_size", c_i32), ("radius", c_i32), ("delta_t_us", c_i32), ("time_window", c_i32),
                ("max_events", c_i64)]


class HeadScale(ctypes.Structure):
    """``dagr_head_scale`` (include/dagr_hip.h)."""
    _fields_ = [("n_ptr", c_void_p), ("n_max", c_i32), ("pred", c_void_p), ("ld", c_i32), ("pos", c_void_p),
                ("batch", c_void_p), ("vx", c_float), ("vy", c_float), ("stride", c_float), ("Hc", c_i32), ("Wc", c_i32),
                ("cnn", c_void_p * 3), ("cnn_stride", (c_i32 * 4) * 3), ("dense", c_void_p)]


class ConvJob(ctypes.Structure):
    """``dagr_conv_job`` (include/dagr_hip.h)."""
    _fields_ = [("n_nodes_ptr", c_void_p), ("n_nodes_max", c_i32), ("rowptr", c_void_p), ("col", c_void_p),
                ("code", c_void_p), ("x", c_void_p), ("ldx", c_i32), ("cin", c_i32), ("xskip", c_void_p),
                ("ldskip", c_i32), ("cskip", c_i32), ("rx", c_i32), ("ry", c_i32), ("den_x", c_float), ("den_y", c_float),
                ("Wq", c_void_p), ("bias", c_void_p), ("C", c_void_p), ("ldc", c_i32), ("N", c_i32), ("relu", c_i32)]


class L0Inputs(ctypes.Structure):
    """``dagr_l0_inputs`` (include/dagr_hip.h)."""
    _fields_ = [("feat", c_void_p), ("pos_nodes", c_void_p), ("batch_nodes", c_void_p), ("x0", c_void_p), ("ldx0", c_i32),
                ("col_feat", c_i32), ("col_pos", c_i32)]


class AsyncUpdateArgs(ctypes.Structure):
    """``dagr_async_update_args`` (include/dagr_hip.h)."""
    _fields_ = [("gdesc", ctypes.POINTER(GraphDesc)), ("graph_ws", c_void_p), ("n_static", c_i64), ("first_id", c_i64),
                ("app_head", c_void_p), ("app_next", c_void_p), ("app_xytb", c_void_p), ("capacity", c_i64),
                ("pos", c_void_p), ("batch", c_void_p), ("batch_is_int64", c_i32), ("n_new", c_i64),
                ("nbr_src", c_void_p), ("nbr_code", c_void_p), ("deg", c_void_p), ("status", c_void_p),
                ("feat", c_void_p), ("pos_nodes", c_void_p), ("batch_nodes", c_void_p), ("batch_events", c_void_p),
                ("x0", c_void_p), ("ldx0", c_i32), ("col_feat", c_i32), ("col_pos", c_i32),
                ("win_x", c_i32), ("tx", c_i32), ("win_y", c_i32), ("ty", c_i32), ("rx", c_i32), ("ry", c_i32),
                ("den_x", c_float), ("den_y", c_float),
                ("cin1", c_i32), ("w1", c_void_p), ("s1", c_void_p), ("h1", c_void_p), ("ldh1", c_i32),
                ("w2", c_void_p), ("s2", c_void_p), ("hp0", c_void_p), ("ldhp0", c_i32),
                ("pdesc", ctypes.POINTER(PoolDesc)), ("pool_ws", c_void_p), ("xlo", c_void_p), ("ylo", c_void_p),
                ("x_out", c_void_p), ("ldo", c_i32), ("pos_out", c_void_p), ("batch_out", c_void_p), ("n_out", c_void_p),
                ("rowptr_out", c_void_p), ("col_out", c_void_p), ("code_out", c_void_p), ("e_out", c_void_p),
                ("e_cap", c_i32)]


# name -> (restype, argtypes); the single source of truth for the symbols we bind.  The CPU-only
# test-suite checks that every function declared in include/dagr_hip.h appears here and resolves.
SIGNATURES = {
    "dagr_last_error": (ctypes.c_char_p, []),
    "dagr_version": (ctypes.c_int, []),
    "dagr_device_count": (ctypes.c_int, []),
    "dagr_format_events": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i32, c_i32, c_i32,
                                          c_void_p, c_void_p, c_void_p]),
    "dagr_fill_edges": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i32, c_float, c_float, c_void_p,
                                       c_i64, c_i32, c_i64, c_i32, c_i32, c_i32, c_i32, c_void_p]),
    "dagr_insert_in_queue": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_i32, c_i32, c_i32, c_i32,
                                            c_void_p]),
    "dagr_insert_in_queue_single": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_i32, c_i32, c_i32, c_i32, c_void_p]),
    "dagr_graph_workspace_bytes": (c_size_t, [ctypes.POINTER(GraphDesc)]),
    "dagr_graph_workspace_init": (ctypes.c_int, [ctypes.POINTER(GraphDesc), c_void_p, c_size_t, c_void_p]),
    "dagr_graph_build_window": (ctypes.c_int, [ctypes.POINTER(GraphDesc), c_void_p, c_void_p, c_i32, c_void_p,
                                               c_i32, c_i64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dagr_graph_search_window": (ctypes.c_int, [ctypes.POINTER(GraphDesc), c_void_p, c_i64, c_void_p, c_void_p, c_void_p,
                                                c_void_p]),
    "dagr_graph_status": (ctypes.c_int, [ctypes.POINTER(GraphDesc), c_void_p, ctypes.POINTER(c_i64),
                                         ctypes.POINTER(c_i32), c_void_p]),
    "dagr_graph_counters": (ctypes.c_int, [ctypes.POINTER(GraphDesc), c_void_p, c_void_p, c_void_p]),
    "dagr_scan_scratch_elems": (c_size_t, [c_i64]),
    "dagr_graph_edge_index": (ctypes.c_int, [ctypes.POINTER(GraphDesc), c_void_p, c_void_p, c_void_p, c_i64, c_void_p,
                                             c_void_p, c_void_p, c_i64, c_void_p]),
    "dagr_graph_node_order": (ctypes.c_int, [ctypes.POINTER(GraphDesc), c_void_p, c_i64, c_void_p, c_void_p, c_void_p]),
    "dagr_graph_gather_inputs": (ctypes.c_int, [ctypes.POINTER(GraphDesc), c_void_p, c_void_p, c_void_p, c_i64,
                                                c_void_p, c_void_p, c_void_p, c_i32, c_i32, c_i32, c_void_p]),
    "dagr_spline_conv_l0_tiles": (ctypes.c_int, [c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_float,
                                                 c_float, c_i64, c_i32, c_void_p, c_void_p, c_void_p, c_void_p, c_i32,
                                                 c_void_p, c_i32, c_void_p, c_void_p, c_i32, c_void_p, c_i32, c_void_p,
                                                 c_void_p]),
    "dagr_graph_build_window_dev": (ctypes.c_int, [ctypes.POINTER(GraphDesc), c_void_p, c_void_p, c_i32, c_void_p,
                                                   c_i32, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dagr_graph_build_window_inputs": (ctypes.c_int, [ctypes.POINTER(GraphDesc), c_void_p, c_void_p, c_void_p, c_i32, c_i64,
                                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dagr_graph_node_count_ptr": (c_void_p, [ctypes.POINTER(GraphDesc), c_void_p]),
    "dagr_graph_csr_codes": (ctypes.c_int, [ctypes.POINTER(GraphDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i32,
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_void_p]),
    "dagr_stage_window": (ctypes.c_int, [ctypes.POINTER(GraphDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_i32, c_i64,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dagr_spiral_offsets": (ctypes.c_int, [c_i32, c_void_p, c_void_p]),
    "dagr_spline_tap_window": (ctypes.c_int, [c_i32, c_float, ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)]),
    "dagr_spline_l0_table": (ctypes.c_int, [c_i32, c_i32, c_float, c_float, c_i32, c_i32, c_i32, c_i32, c_void_p,
                                            c_void_p, c_void_p]),
    "dagr_spline_conv_l0": (ctypes.c_int, [c_i32, c_i32, c_i32, c_i64, c_i32, c_i32, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_i32, c_void_p, c_i32, c_void_p, c_void_p, c_void_p, c_i32,
                                           c_void_p, c_i32, c_void_p]),
    "dagr_spline_tap_aggregate": (ctypes.c_int, [c_void_p, c_i32, c_void_p, c_void_p, c_void_p, c_void_p, c_i32,
                                                 c_i32, c_void_p, c_i32, c_i32, c_i32, c_i32, c_float, c_float,
                                                 c_void_p, c_i32, c_void_p]),
    "dagr_spline_tap_scatter_grad": (ctypes.c_int, [c_void_p, c_i32, c_void_p, c_void_p, c_void_p, c_void_p, c_i32, c_i32,
                                                    c_i32, c_i32, c_float, c_float, c_void_p, c_void_p, c_void_p, c_i32,
                                                    c_void_p]),
    "dagr_spline_tap_scatter_grad_w": (ctypes.c_int, [c_void_p, c_i32, c_void_p, c_void_p, c_void_p, c_void_p, c_i32, c_i32,
                                                      c_void_p, c_i32, c_i32, c_i32, c_i32, c_float, c_float, c_void_p,
                                                      c_void_p, c_void_p, c_i32, c_void_p]),
    "dagr_pool_workspace_bytes": (c_size_t, [ctypes.POINTER(PoolDesc)]),
    "dagr_pool_workspace_init": (ctypes.c_int, [ctypes.POINTER(PoolDesc), c_void_p, c_size_t, c_void_p]),
    "dagr_pool_l0": (ctypes.c_int, [ctypes.POINTER(PoolDesc), c_void_p, ctypes.POINTER(GraphDesc), c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_i32, c_void_p, c_void_p, c_void_p, c_i32, c_i64,
                                    c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_i32, c_i32, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_i32, c_void_p]),
    "dagr_pool_l0_accumulate": (ctypes.c_int, [ctypes.POINTER(PoolDesc), c_void_p, ctypes.POINTER(GraphDesc), c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_i32, c_void_p, c_i64, c_void_p, c_void_p,
                                               c_void_p, c_void_p]),
    "dagr_pool_csr": (ctypes.c_int, [ctypes.POINTER(PoolDesc), c_void_p, c_void_p, c_i32, c_void_p, c_i32, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i32, c_i32, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i32, c_void_p]),
    "dagr_pool_recode": (ctypes.c_int, [c_void_p, c_i32, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float,
                                        c_float, c_i32, c_i32, c_void_p, c_i32, c_void_p, c_void_p]),
    "dagr_pool_status": (ctypes.c_int, [ctypes.POINTER(PoolDesc), c_void_p, ctypes.POINTER(c_i32), c_void_p]),
    "dagr_pool_status_ptr": (c_void_p, [ctypes.POINTER(PoolDesc), c_void_p]),
    "dagr_pool_counters": (ctypes.c_int, [ctypes.POINTER(PoolDesc), c_void_p, c_void_p, c_void_p]),
    "dagr_to_dense": (ctypes.c_int, [c_void_p, c_i32, c_void_p, c_i32, c_i32, c_void_p, c_void_p, c_float, c_float,
                                     c_i32, c_i32, c_i32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dagr_pool_argmax": (ctypes.c_int, [c_void_p, c_i32, c_void_p, c_i32, c_i32, c_void_p, c_i32, c_i32, c_void_p, c_void_p]),
    "dagr_pool_grad": (ctypes.c_int, [c_void_p, c_i32, c_i32, c_i32, c_void_p, c_void_p, c_void_p, c_i32, c_void_p, c_i32,
                                      c_void_p]),
    "dagr_to_dense_grad": (ctypes.c_int, [c_i32, c_i32, c_void_p, c_void_p, c_float, c_float, c_i32, c_i32, c_i32, c_void_p,
                                          c_void_p, c_i32, c_void_p]),
    "dagr_sample_features": (ctypes.c_int, [c_void_p, c_i32, c_void_p, c_void_p, c_i32, c_void_p, c_i32, c_i32, c_i32,
                                            c_i32, c_i32, c_i32, c_void_p, c_i32, c_i32, c_void_p]),
    "dagr_nms_batched": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i32, c_i32, c_float, c_float, c_void_p,
                                        c_void_p, c_void_p, c_void_p]),
    "dagr_heads_finish": (ctypes.c_int, [ctypes.POINTER(HeadScale), ctypes.POINTER(HeadScale), c_i32, c_i32, c_void_p,
                                         c_void_p, c_void_p]),
    "dagr_heads_finish_detect": (ctypes.c_int, [ctypes.POINTER(HeadScale), ctypes.POINTER(HeadScale), c_i32, c_i32, c_void_p,
                                                c_void_p, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "dagr_decode_heads": (ctypes.c_int, [c_void_p, c_i32, c_i32, c_float, c_void_p, c_i32, c_i32, c_float, c_i32, c_i32,
                                         c_void_p, c_void_p]),
    "dagr_postprocess": (ctypes.c_int, [c_void_p, c_i32, c_i32, c_i32, c_float, c_float, c_float, c_void_p, c_void_p,
                                        c_void_p]),
    "dagr_masked_lin": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i32, c_i64, c_i32, c_i32, c_void_p]),
    "dagr_masked_lin_no_bias": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i32, c_i64, c_i32, c_i32, c_void_p]),
    "dagr_masked_isdiff": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_float, c_float, c_i64, c_i32, c_void_p]),
    "dagr_masked_inplace_BN": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                              c_i64, c_i32, c_void_p]),
    "dagr_downsample_events": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_i32, c_void_p, c_i32, c_i32, c_void_p,
                                              c_void_p, c_void_p]),
    "dagr_spline_conv_l0_tiles_rows": (ctypes.c_int, [c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_float,
                                                      c_float, c_i64, c_i64, c_i32, c_void_p, c_void_p, c_void_p, c_void_p,
                                                      c_i32, c_void_p, c_i32, c_void_p, c_void_p, c_i32, c_void_p, c_i32,
                                                      c_void_p, c_void_p]),
    "dagr_async_graph_append": (ctypes.c_int, [ctypes.POINTER(GraphDesc), c_void_p, c_i64, c_i64, c_void_p, c_void_p,
                                               c_void_p, c_i64, c_void_p, c_i32, c_void_p, c_i32, c_i64, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i32,
                                               c_i32, c_i32, c_void_p]),
    "dagr_async_update": (ctypes.c_int, [ctypes.POINTER(AsyncUpdateArgs), c_void_p]),
    "dagr_pool_l0_stream": (ctypes.c_int, [ctypes.POINTER(PoolDesc), c_void_p, c_i32, ctypes.POINTER(GraphDesc), c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_i32, c_void_p, c_void_p, c_i64, c_i64, c_i64,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_i32, c_i32, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i32, c_void_p]),
    "dagr_spline_conv_fused_lds_bytes": (c_size_t, [c_i32, c_i32]),
    "dagr_spline_conv_fused_passes": (c_i32, [c_i32, c_i32]),
    "dagr_spline_conv_fused": (ctypes.c_int, [c_void_p, c_i32, c_void_p, c_void_p, c_void_p, c_void_p, c_i32, c_i32,
                                              c_void_p, c_i32, c_i32, c_i32, c_i32, c_float, c_float, c_void_p,
                                              c_void_p, c_void_p, c_i32, c_i32, c_i32, c_void_p]),
    "dagr_spline_conv_fused_pool": (ctypes.c_int, [c_void_p, c_i32, c_void_p, c_void_p, c_void_p, c_void_p, c_i32, c_i32,
                                                   c_void_p, c_i32, c_i32, c_i32, c_i32, c_float, c_float, c_void_p,
                                                   c_void_p, c_void_p, c_i32, c_i32, c_i32, ctypes.POINTER(PoolDesc),
                                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dagr_spline_conv_fused_multi": (ctypes.c_int, [c_void_p, c_i32, c_void_p]),
    "dagr_spline_conv_fused_pair": (ctypes.c_int, [c_void_p, c_i32, c_void_p, c_void_p, c_void_p, c_i32, c_i32, c_i32, c_i32,
                                                   c_float, c_float, c_i32, c_i32, c_void_p, c_void_p, c_void_p, c_void_p,
                                                   c_i32, c_void_p, c_void_p, c_void_p, c_void_p, c_i32, c_void_p]),
    "dagr_gemm_epilogue_workspace_bytes": (c_size_t, []),
    "dagr_gemm_epilogue": (ctypes.c_int, [c_void_p, c_i64, c_i32, c_i64, c_void_p, c_i32, c_void_p, c_void_p, c_i64, c_i32,
                                          c_void_p, c_i64, c_void_p, c_size_t, c_void_p]),
    "dagr_add_relu": (ctypes.c_int, [c_void_p, c_void_p, c_i64, c_void_p]),
    "dagr_bias_relu": (ctypes.c_int, [c_void_p, c_void_p, c_i64, c_i32, c_void_p]),
    "dagr_bias_silu": (ctypes.c_int, [c_void_p, c_void_p, c_i64, c_i32, c_void_p]),
    "dagr_bn_relu_maxpool": (ctypes.c_int, [c_void_p, c_i32, c_i32, c_i32, c_i32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dagr_gemm_bias_act": (ctypes.c_int, [c_void_p, c_i32, c_void_p, c_i32, c_void_p, c_i32, c_void_p, c_void_p,
                                          c_i32, c_i32, c_i32, c_i32, c_void_p]),
}

_lib = None


def lib():
    """Load libdagr_hip.so once; fail loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"dagr_amd: {LIB_PATH} not found -- build it with `make` (or __graft_entry__.build()). "
                "There is no CPU fallback for the event-graph hot path.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().dagr_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libdagr_hip {what} failed (status {rc}): {msg}")


def ptr(t):
    """Raw device/host pointer of a torch tensor (or None)."""
    return None if t is None else c_void_p(t.data_ptr())


def cur_stream(device=None):
    """hipStream_t of torch's current stream, as void*."""
    import torch
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)
