"""GPU parity of the HIP window graph builder against the CPU oracle (bit-exact, integers)."""
import numpy as np
import pytest
import torch

from oracle import graph as og
from dagr_amd.utils import synthetic as syn
from tests.graph_cases import heavy_cases, small_cases, medium_cases, wide_radius_cases

pytestmark = pytest.mark.gpu


def _run_hip(case, use_float_pos=False):
    from dagr_amd.graph.ev_graph import WindowGraphBuilder
    from dagr_amd.utils.synthetic import format_data_np
    dev = torch.device("cuda:0")
    N = len(case["x"])
    g = WindowGraphBuilder(case["W"], case["H"], case["B"], case["K"], case["Q"], case["r"], case["dt"],
                           max_events=max(N, 16), device=dev)
    if use_float_pos:
        pos = torch.from_numpy(format_data_np(case["x"], case["y"], case["t"], case["W"], case["H"])).to(dev)
    else:
        pos = torch.from_numpy(np.stack([case["x"], case["y"], case["t"]], -1).astype(np.int32).reshape(N, 3)).to(dev)
    batch = torch.from_numpy(case["b"].astype(np.int64)).to(dev)
    outs = []
    for rep in range(2):  # second build checks the "counters are zero again" invariant
        nbr_src, nbr_code, deg = g.build(pos, batch)
        ei, rowptr = g.edge_index(nbr_src, deg)
        ne, flags = g.status()
        # neighbour lists are in node (slot) order: bring them to event order for the checks below
        slot_event, event_slot = [v.cpu().numpy() for v in g.node_order(N)] if N else (np.zeros(0, int), np.zeros(0, int))
        ok = event_slot >= 0
        es = np.maximum(event_slot, 0)
        deg_ev = np.where(ok, deg.cpu().numpy()[es], 1) if N else deg.cpu().numpy()
        code_ev = nbr_code.cpu().numpy()[es] if N else nbr_code.cpu().numpy()
        if N:
            src_n = nbr_src.cpu().numpy()[es]
            valid = (np.arange(src_n.shape[1])[None, :] < deg_ev[:, None]) & ok[:, None]
            src_ev = np.where(valid, slot_event[np.where(valid, src_n, 0)], 0)
        else:
            src_ev = nbr_src.cpu().numpy()
        outs.append((ei.cpu().numpy(), src_ev, code_ev, deg_ev, ne, flags))
    assert (outs[0][0] == outs[1][0]).all()
    return outs[1]


def _oracle(case):
    return og.build_window_graph(case["x"], case["y"], case["t"], case["b"], case["W"], case["H"], case["B"],
                                 case["r"], case["dt"], K=case["K"], Q=case["Q"])


def _check_codes(case, nbr_src, nbr_code, deg):
    r = case["r"]
    side = 2 * r + 1
    for e in range(0, len(deg), max(1, len(deg) // 200)):
        for j in range(deg[e]):
            s = nbr_src[e, j]
            dx, dy = case["x"][s] - case["x"][e], case["y"][s] - case["y"][e]
            assert nbr_code[e, j] == (dx + r) * side + (dy + r)


@pytest.mark.parametrize("case", small_cases(), ids=lambda c: c["name"])
def test_small_cases_bit_exact(case):
    ei, nbr_src, nbr_code, deg, ne, flags = _run_hip(case)
    ref = _oracle(case)
    assert flags == 0
    assert ei.shape == ref.shape, (ei.shape, ref.shape)
    assert (ei == ref).all()
    assert ne == ref.shape[1]
    _check_codes(case, nbr_src, nbr_code, deg)


@pytest.mark.parametrize("case", wide_radius_cases(), ids=lambda c: c["name"])
def test_wide_radius_cases_bit_exact(case):
    """r > 7: every node goes through the generic form of k_search_dense (sorted and unsorted timestamps, K = 24,
    FIFO depths 8 / 16 so that pixels exceed them)."""
    ei, nbr_src, nbr_code, deg, ne, flags = _run_hip(case)
    ref = _oracle(case)
    assert flags == 0
    assert ei.shape == ref.shape, (ei.shape, ref.shape)
    assert (ei == ref).all()
    assert ne == ref.shape[1] and deg.max() <= case["K"]
    _check_codes(case, nbr_src, nbr_code, deg)


@pytest.mark.parametrize("case", medium_cases(), ids=lambda c: c["name"])
def test_medium_cases_bit_exact_float_pos(case):
    """normalised fp32 pos in (format_data -> denormalize_pos round trip fused in the kernel)."""
    ei, nbr_src, nbr_code, deg, ne, flags = _run_hip(case, use_float_pos=True)
    pos = og.denormalize_pos(__import__("dagr_amd.utils.synthetic", fromlist=["x"]).format_data_np(
        case["x"], case["y"], case["t"], case["W"], case["H"]), case["W"], case["H"], 1000000)
    ref = og.build_window_graph(pos[:, 0], pos[:, 1], pos[:, 2], case["b"], case["W"], case["H"], case["B"],
                                case["r"], case["dt"], K=case["K"], Q=case["Q"])
    assert flags == 0
    assert ei.shape == ref.shape
    assert (ei == ref).all()
    # invariants stated by the reference (ev_tgn.py:52-54): src <= dst, dst non-decreasing
    assert (ei[0] <= ei[1]).all() and (np.diff(ei[1]) >= 0).all()
    assert deg.max() <= case["K"] and deg.min() >= 1


def test_out_of_range_event_is_flagged():
    case = small_cases()[3].copy()
    case["x"] = case["x"].copy(); case["x"][0] = case["W"] + 3
    ei, nbr_src, nbr_code, deg, ne, flags = _run_hip(case)
    assert flags & 1


def test_format_events_bit_exact_vs_reference_golden():
    """a1 `format_data` (utils/buffers.py:33-44): `dagr_format_events` on the dataset's raw dtypes against the outputs
    of the reference's own function (tests/golden/ref_py_functions.npz: fmt_*), bit for bit; plus a 640x480 window
    against the fp32 true division written out in numpy."""
    import os
    import types
    from dagr_amd.utils.buffers import format_data
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_py_functions.npz"))
    d = types.SimpleNamespace(width=torch.tensor([320]), height=torch.tensor([215]), time_window=torch.tensor([1000000]),
                              pos=torch.from_numpy(G["fmt_pos"]).cuda(), t=torch.from_numpy(G["fmt_t"]).cuda(),
                              x=torch.from_numpy(G["fmt_x"]).cuda())
    assert d.pos.dtype == torch.int16 and d.t.dtype == torch.int32 and d.x.dtype == torch.int8   # -> the HIP kernel
    d = format_data(d)
    assert d.t is None
    assert np.array_equal(d.pos.cpu().numpy(), G["fmt_out_pos"]) and np.array_equal(d.x.cpu().numpy(), G["fmt_out_x"])
    x, y, t, p = syn.uniform_window(100000, 640, 480, seed=3)
    d = types.SimpleNamespace(width=torch.tensor([640]), height=torch.tensor([480]), time_window=torch.tensor([1000000]),
                              pos=torch.from_numpy(np.stack([x, y], -1)).cuda(), t=torch.from_numpy(t).cuda(),
                              x=torch.from_numpy(p.reshape(-1, 1)).cuda())
    d = format_data(d)
    assert np.array_equal(d.pos.cpu().numpy(), syn.format_data_np(x, y, t, 640, 480))
    assert np.array_equal(d.x.cpu().numpy(), p.astype(np.float32).reshape(-1, 1))


def _float_case(case):
    from dagr_amd.utils.synthetic import format_data_np
    dev = torch.device("cuda:0")
    pos = torch.from_numpy(format_data_np(case["x"], case["y"], case["t"], case["W"], case["H"])).to(dev)
    batch = torch.from_numpy(case["b"].astype(np.int64)).to(dev)
    feat = torch.from_numpy(np.where(np.arange(len(case["x"])) % 3 == 0, -1.0, 1.0).astype(np.float32)).to(dev)
    return pos, batch, feat


def _edges(g, lists, n):
    """edge_index (event ids; slot numbering inside a long pixel segment is not specified, event ids are)."""
    return g.edge_index(lists[0][:n], lists[2][:n])[0]


def _builder(case, N):
    from dagr_amd.graph.ev_graph import WindowGraphBuilder
    return WindowGraphBuilder(case["W"], case["H"], case["B"], case["K"], case["Q"], case["r"], case["dt"],
                              max_events=max(N, 16), device=torch.device("cuda:0"))


@pytest.mark.parametrize("case", medium_cases()[:2] + small_cases()[:4], ids=lambda c: c["name"])
def test_build_with_level0_inputs_equals_build_then_gather(case):
    """dagr_graph_build_window_inputs (the node-ordered level-0 inputs written by the build's last launch) ==
    dagr_graph_build_window + dagr_graph_gather_inputs, bit for bit: neighbour lists, pos / sample of every node, the
    feature row's own columns; the columns left for image features stay untouched."""
    import ctypes
    from dagr_amd import _lib
    L, P = _lib.lib(), _lib.ptr
    dev = torch.device("cuda:0")
    pos, batch, feat = _float_case(case)
    N = int(pos.shape[0])
    if N == 0:
        pytest.skip("empty case")
    g = _builder(case, N)
    ld, col_feat, col_pos = 6, 4, 1
    a = g.build(pos, batch)
    ei_a, st_a = _edges(g, a, N), g.status()
    pos_b = torch.full((N, 3), -5.0, device=dev); b_b = torch.full((N,), -5, dtype=torch.int32, device=dev)
    x_b = torch.full((N, ld), -5.0, device=dev)
    inputs = _lib.L0Inputs(feat=feat.data_ptr(), pos_nodes=pos_b.data_ptr(), batch_nodes=b_b.data_ptr(), x0=x_b.data_ptr(),
                           ldx0=ld, col_feat=col_feat, col_pos=col_pos)
    b = g.build(pos, batch, inputs=inputs)
    assert g.status() == st_a and torch.equal(_edges(g, b, N), ei_a)
    # every node carries its own event's inputs
    slot_event, event_slot = g.node_order(N)
    assert bool((event_slot >= 0).all())
    s_ = event_slot.long()
    assert torch.equal(pos_b[s_], pos) and torch.equal(b_b[s_].long(), batch) and torch.equal(x_b[s_, col_feat], feat)
    assert torch.equal(x_b[s_][:, col_pos:col_pos + 2], pos[:, :2])
    assert bool((x_b[:, [0, 3, 5]] == -5.0).all())
    # ... which is what the separate launch writes
    pos_a = torch.full((N, 3), -5.0, device=dev); b_a = torch.full((N,), -5, dtype=torch.int32, device=dev)
    x_a = torch.full((N, ld), -5.0, device=dev)
    _lib.check(L.dagr_graph_gather_inputs(ctypes.byref(g.desc), P(g.workspace), P(pos), P(feat), N, P(pos_a), P(b_a), P(x_a), ld,
                                          col_feat, col_pos, _lib.cur_stream(dev)), "gather")
    assert torch.equal(pos_a, pos_b) and torch.equal(b_a, b_b) and torch.equal(x_a, x_b)


@pytest.mark.parametrize("mutate", ["plain", "out_of_range", "unsorted_time"])
def test_staged_device_count_build_equals_host_count_build(mutate):
    """The captured-window form: dagr_stage_window (copy + event count to device memory + the build's first step) followed by
    dagr_graph_build_window_dev on capacity-sized buffers == dagr_graph_build_window on the window itself -- neighbour
    lists, edge count and the status flags (an event outside the sensor, timestamps out of order), also when the same
    buffers then serve a smaller window."""
    import ctypes
    from dagr_amd import _lib
    L, P = _lib.lib(), _lib.ptr
    dev = torch.device("cuda:0")
    case = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in medium_cases()[0].items()}
    if mutate == "out_of_range":
        case["x"][5] = case["W"] + 2
    if mutate == "unsorted_time":
        i = len(case["t"]) // 2
        case["t"][i], case["t"][i + 1] = case["t"][i + 1] + 7, case["t"][i]
    pos, batch, feat = _float_case(case)
    N = int(pos.shape[0])
    cap = N + 1000
    K = case["K"]
    for n in (N, N // 3):
        g1 = _builder(case, cap)
        want = g1.build(pos[:n].contiguous(), batch[:n].contiguous())
        want_status = g1.status()
        g2 = _builder(case, cap)
        in_pos = torch.zeros((cap, 3), device=dev); in_feat = torch.zeros((cap,), device=dev)
        in_batch = torch.zeros((cap,), dtype=torch.int32, device=dev); n_dev = torch.zeros((1,), dtype=torch.int32, device=dev)
        out = (torch.zeros((cap, K), dtype=torch.int32, device=dev), torch.zeros((cap, K), dtype=torch.int16, device=dev),
               torch.zeros((cap,), dtype=torch.int32, device=dev))
        for rep in range(2):        # twice: the staging launch re-arms what the previous window left
            _lib.check(L.dagr_stage_window(ctypes.byref(g2.desc), P(g2.workspace), P(pos), P(feat), P(batch), 1, n, P(in_pos),
                                           P(in_feat), P(in_batch), P(n_dev), _lib.cur_stream(dev)), "stage_window")
            g2.build(in_pos, in_batch, out=out, n_dev=n_dev)
        assert g2.status() == want_status, (mutate, n)
        if mutate == "out_of_range":
            assert want_status[1] & 1
        assert torch.equal(_edges(g1, want, n), _edges(g2, out, n)), (mutate, n)


def _search_counters(case):
    """{deferred to the position-centric walk, answered from the inner rings, unsorted} of one more build of the case."""
    import ctypes
    from dagr_amd import _lib
    N = len(case["x"])
    g = _builder(case, max(N, 16))
    dev = torch.device("cuda:0")
    pos = torch.from_numpy(np.stack([case["x"], case["y"], case["t"]], -1).astype(np.int32).reshape(N, 3)).to(dev)
    g.build(pos, torch.from_numpy(case["b"].astype(np.int64)).to(dev))
    out = (ctypes.c_int32 * 8)()
    _lib.check(_lib.lib().dagr_graph_counters(ctypes.byref(g.desc), _lib.ptr(g.workspace), ctypes.cast(out, ctypes.c_void_p),
                                              _lib.cur_stream(dev)), "counters")
    return dict(walked=int(out[5]), inner=int(out[7]), unsorted=int(out[6]))


@pytest.mark.parametrize("case", heavy_cases(), ids=lambda c: c["name"])
def test_heavy_neighbourhoods_bit_exact(case):
    """Event-dense neighbourhoods -- the row kernel's ring-limited passes (inner window first, then the parts left and right
    of it) and the position-centric walk of the deferred ones -- == the oracle == (where oracle/_ref is built) the
    reference's own insert_in_queue + fill_edges kernels run on this GPU, edge for edge; the counters show that the
    intended path ran."""
    ei, nbr_src, nbr_code, deg, ne, flags = _run_hip(case)
    ref = _oracle(case)
    assert flags == 0
    assert ei.shape == ref.shape, (ei.shape, ref.shape)
    assert (ei == ref).all()
    assert ne == ref.shape[1] and deg.max() <= case["K"]
    _check_codes(case, nbr_src, nbr_code, deg)
    from oracle import ref_harness
    if ref_harness.available():
        own = ref_harness.reference_window_graph(case["x"], case["y"], case["t"], case["b"], case["W"], case["H"], case["B"],
                                                 case["r"], case["dt"], K=case["K"], Q=case["Q"])
        assert own.shape == ei.shape and (own == ei).all()
    cnt = _search_counters(case)
    assert cnt["unsorted"] == 0
    if case["name"].startswith(("walk", "blob", "uniform_dense", "stale")):
        assert cnt["walked"] > 0, cnt
    if case["name"].startswith("edges_r7"):
        assert cnt["inner"] > 0, cnt
