"""GPU parity of the asynchronous (reset=False) update -- f3, SURVEY.md section 8f rank 3.

The reference's guarantee (``asynchronous/evaluate_flops.py:139-147``): the asynchronous model's output after feeding events
incrementally equals the synchronous forward on all events so far (it checks 1e-3).  Here, piece by piece and end to end:
  * the in-edges of appended events (per-pixel chains + the window's pixel index, csrc/async_update.hip) equal, bit for
    bit, the reference's FIFO search over ALL events (the C oracle / the reference's own kernels) -- incl. hot pixels beyond
    the FIFO depth, borders, identical timestamps, single-event updates;
  * level 1 after an update (pool1's resident accumulators, dagr_pool_l0_stream) equals level 1 of a window built from
    all events, array by array, bit for bit;
  * ``DAGR.forward(x, reset=False)`` equals one ``reset=True`` call on all events and equals the re-evaluating form
    (``make_model_synchronous``), bit for bit, for B = 1 and B = 2.
"""
import numpy as np
import pytest
import torch

from oracle import graph as og
from oracle import model as om
from dagr_amd.utils import synthetic as syn
from dagr_amd.utils.testing_weights import randomize_
from tests.graph_cases import small_cases

pytestmark = pytest.mark.gpu


def _model(W, H, B, seed=0, **over):
    from dagr_amd.model.networks.dagr import DAGR
    torch.manual_seed(seed)
    args = om.default_args(batch_size=B, **over)
    model = randomize_(DAGR(args, height=H, width=W), seed=seed).eval().cuda()
    model.cache_luts(width=W, height=H, radius=args.radius)
    return args, model


def _dev(x, y, t, p, b, W, H):
    dev = torch.device("cuda:0")
    pos = torch.from_numpy(syn.format_data_np(x, y, t, W, H)).to(dev)
    feat = torch.from_numpy(np.asarray(p, np.float32)).view(-1, 1).to(dev)
    return pos, feat, torch.from_numpy(np.asarray(b, np.int64)).to(dev)


def _edges_by_event(eng, n_total):
    """event-ordered (src id, dst id) pairs of the engine's level-0 rows: window rows are CSR slots, appended rows ids."""
    n0 = eng._N
    slot_event, _ = eng.graph.node_order(n0)
    row_event = torch.cat([slot_event.long(), torch.arange(n0, n_total, device=slot_event.device)])
    deg = eng.deg[:n_total].long()
    src = eng.nbr_src[:n_total].long()
    valid = torch.arange(src.shape[1], device=src.device)[None, :] < deg[:, None]
    dst_e = row_event[:, None].expand_as(src)[valid]
    src_e = row_event[src[valid]]
    order = torch.argsort(dst_e, stable=True)          # rows are already in spiral order; group by destination id
    return torch.stack([src_e[order], dst_e[order]]).cpu().numpy()


def _hot_pixel_320():
    """320 x 215 (r = 4): 300 events on one pixel (beyond the FIFO depth of 128) in a cloud of neighbours, identical
    timestamps in places, events on the border."""
    rng = np.random.default_rng(5)
    n = 300
    t = np.sort(rng.integers(985000, 1000001, n + 260))
    x = np.concatenate([np.full(n, 100), rng.integers(94, 107, 200), rng.integers(0, 3, 30), rng.integers(317, 320, 30)])
    y = np.concatenate([np.full(n, 80), rng.integers(74, 87, 200), rng.integers(0, 215, 30), rng.integers(0, 215, 30)])
    perm = rng.permutation(n + 260)
    t[50:60] = t[50]
    return dict(name="hot_pixel_320", x=x[perm].astype(np.int32), y=y[perm].astype(np.int32), t=t.astype(np.int32),
                b=np.zeros(n + 260, np.int32), W=320, H=215, B=1, r=4, dt=10000, K=16, Q=128)


@pytest.mark.parametrize("case", [c for c in small_cases() if len(c["x"]) >= 40 and c["K"] == 16 and c["W"] >= 64
                                  and c["B"] == 1] + [_hot_pixel_320()],
                         ids=lambda c: c["name"])
def test_appended_rows_have_the_reference_edges(case):
    """Window of the first part of the events, the rest appended in three micro-batches (one of them a single event):
    every event's in-edges == the reference's graph object fed the same sequence of calls (reset, then attaching calls:
    ev_tgn.py:45-56) -- radius / delta_t as the model derives them from the sensor width (ev_tgn.py:28-29).  (With more
    than Q events on a pixel this is NOT what one call on all events gives: the reference searches the FIFO as it stands
    after the call's own insertions.)"""
    W, H, B = case["W"], case["H"], case["B"]
    args, model = _model(W, H, B)
    eng = model.engine()
    x, y, t, b = case["x"], case["y"], case["t"], case["b"]
    p = np.ones(len(x), np.int8)
    N = len(x)
    # (multi-sample micro-batches: test_level1_and_outputs_after_updates_equal_a_window_on_all_events, test_dagr_forward_reset_false)
    cuts = [N // 2, N // 2 + (N - N // 2) // 2, N - 1, N]
    pos, feat, batch = _dev(x, y, t, p, b, W, H)
    eng.forward_raw(pos[:cuts[0]], feat[:cuts[0]], batch[:cuts[0]])
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        eng.forward_append(pos[lo:hi], feat[lo:hi], batch[lo:hi])
    eng.check_status()
    got = _edges_by_event(eng, N)
    gp = eng.graph.params
    assert gp["queue_size"] == 128
    g = og.SlidingWindowGraph(width=W, height=H, batch_size=B, max_num_neighbors=16, max_queue_size=128,
                              radius=gp["radius"], delta_t_us=gp["delta_t_us"])
    ipos = np.stack([x, y, t], -1).astype(np.int32)
    parts = [g.forward(np.ascontiguousarray(b[lo:hi].astype(np.int32)), ipos[lo:hi], delete_nodes=False, collect_edges=True)
             for lo, hi in zip([0] + cuts[:-1], cuts)]
    ref = np.concatenate([np.asarray(e, np.int64).reshape(2, -1) for e in parts], axis=1)
    assert got.shape == ref.shape
    assert (got == ref).all()


def _level1(eng):
    l1 = eng.levels[0]
    n, e = [int(v) for v in l1.counts.tolist()]
    return dict(n=n, e=e, x=l1.x[:n].clone(), pos=l1.pos[:n].clone(), batch=l1.batch[:n].clone(),
                rowptr=l1.rowptr[:n + 1].clone(), col=l1.col[:e].clone(), code=l1.code[:e].clone())


@pytest.mark.parametrize("B,stream", [(1, "uniform"), (2, "edges")])
def test_level1_and_outputs_after_updates_equal_a_window_on_all_events(B, stream):
    W, H = 320, 215
    args, model = _model(W, H, B, seed=3)
    eng = model.engine()
    gen = syn.uniform_window if stream == "uniform" else syn.edges_window
    raw = [gen(6000, W, H, seed=90 + s) for s in range(B)]
    # the last events of the dataset's windows sit at t == time_window (pos t == 1.0, QUIRK-1): they arrive in the updates
    cuts = [0, 4000, 5200, 5990, 5999, 6000]

    def part(lo, hi):
        xs = [np.concatenate([r[k][lo:hi] for r in raw]) for k in range(4)]
        b = np.concatenate([np.full(hi - lo, s, np.int64) for s in range(B)])
        return _dev(xs[0], xs[1], xs[2], xs[3], b, W, H)

    with torch.no_grad():
        eng.set_low_latency(False)
        eng.forward_raw(*part(cuts[0], cuts[1]))
        for lo, hi in zip(cuts[1:-1], cuts[2:]):
            out_async = eng.forward_append(*part(lo, hi)).clone()
        eng.check_status()
        lvl_async = _level1(eng)
        out_full = eng.forward_raw(*part(0, 6000)).clone()
        eng.check_status()
        lvl_full = _level1(eng)
    assert lvl_async["n"] == lvl_full["n"] and lvl_async["e"] == lvl_full["e"]
    for k in ("x", "pos", "batch", "rowptr", "col", "code"):
        assert torch.equal(lvl_async[k], lvl_full[k]), k
    assert torch.equal(out_async, out_full)


@pytest.mark.parametrize("B", [1, 2])
def test_dagr_forward_reset_false(B):
    """The consistency check of evaluate_flops.py:139-147 in this stack's terms: events_initial with reset=True, then
    micro-batches with reset=False (incremental) == the same calls on a model switched to the re-evaluating form
    (make_model_synchronous) == one reset=True call on everything."""
    from dagr_amd.asynchronous import make_model_asynchronous, make_model_synchronous
    from dagr_amd.data import Batch, Data
    from dagr_amd.utils.buffers import format_data
    W, H = 320, 215
    args, model = _model(W, H, B, seed=5)
    raw = [syn.edges_window(5000, W, H, seed=40 + s) for s in range(B)]

    def batch_of(lo, hi):
        samples = []
        for s in range(B):
            x, y, t, p = (a[lo:hi] for a in raw[s])
            samples.append(Data(x=torch.from_numpy(p.reshape(-1, 1)), pos=torch.from_numpy(np.stack([x, y], -1)),
                                t=torch.from_numpy(t), width=W, height=H, time_window=1000000))
        return format_data(Batch.from_data_list(samples).cuda())

    cuts = [0, 3000, 4000, 4900, 4990, 4999, 5000]
    outs = {}
    with torch.no_grad():
        for mode, convert in (("asynchronous", make_model_asynchronous), ("synchronous", make_model_synchronous)):
            convert(model)
            for k in range(len(cuts) - 1):
                det, = model(batch_of(cuts[k], cuts[k + 1]), reset=(k == 0), return_targets=False)
            outs[mode] = [{k: v.clone() for k, v in d.items()} for d in det]
        make_model_asynchronous(model)
        full, = model(batch_of(0, 5000), reset=True, return_targets=False)
    for a, s_, f in zip(outs["asynchronous"], outs["synchronous"], full):
        for key in ("boxes", "scores", "labels"):
            assert torch.equal(a[key], s_[key]), key
            assert torch.equal(a[key], f[key]), key
    assert sum(len(d["boxes"]) for d in full) > 0


def test_update_grows_the_row_arrays_and_a_reset_starts_over():
    W, H, B = 320, 215, 1
    args, model = _model(W, H, B, seed=7)
    eng = model.engine()
    x, y, t, p = syn.uniform_window(9000, W, H, seed=11)
    b = np.zeros(9000, np.int64)
    pos, feat, batch = _dev(x, y, t, p, b, W, H)
    with torch.no_grad():
        eng.forward_raw(pos[:1500], feat[:1500], batch[:1500])
        cap0 = eng.rows_cap
        out = None
        for lo in range(1500, 9000, 1500):           # far beyond the reserve of the first update
            out = eng.forward_append(pos[lo:lo + 1500], feat[lo:lo + 1500], batch[lo:lo + 1500]).clone()
        eng.check_status()
        full = eng.forward_raw(pos, feat, batch).clone()
        assert torch.equal(out, full)
        # a reset=True window drops the asynchronous state; the next update starts from the new window
        eng.forward_raw(pos[:3000], feat[:3000], batch[:3000])
        out2 = eng.forward_append(pos[3000:3100], feat[3000:3100], batch[3000:3100]).clone()
        full2 = eng.forward_raw(pos[:3100], feat[:3100], batch[:3100]).clone()
        assert torch.equal(out2, full2)
    assert eng.rows_cap >= 9000 and cap0 <= eng.rows_cap


def test_reset_false_falls_back_where_the_incremental_path_does_not_apply():
    """configurations outside the incremental path (here max_neighbors = 8: the tiled level-0 conv and the
    chain search are built for 16) used to raise on reset=False; they now re-evaluate the running window, as the
    synchronous form does.  Same for an engine rebuilt between calls (weights edited in place)."""
    from dagr_amd.data import Batch, Data
    from dagr_amd.utils.buffers import format_data
    W, H, B = 320, 215, 1
    raw = syn.edges_window(4000, W, H, seed=77)

    def batch_of(lo, hi):
        x, y, t, p = (a[lo:hi] for a in raw)
        d = Data(x=torch.from_numpy(p.reshape(-1, 1)), pos=torch.from_numpy(np.stack([x, y], -1)), t=torch.from_numpy(t),
                 width=W, height=H, time_window=1000000)
        return format_data(Batch.from_data_list([d]).cuda())

    with torch.no_grad():
        args, model = _model(W, H, B, seed=9, max_neighbors=8)
        assert not model.engine().l0_tiles and not model.engine().can_append()
        model(batch_of(0, 3000), reset=True, return_targets=False)
        det, = model(batch_of(3000, 4000), reset=False, return_targets=False)
        full, = model(batch_of(0, 4000), reset=True, return_targets=False)
        for key in ("boxes", "scores", "labels"):
            assert torch.equal(det[0][key], full[0][key]), key
        # an engine rebuilt between the calls has no resident window: the update re-evaluates instead of failing
        args, model = _model(W, H, B, seed=9)
        model(batch_of(0, 3000), reset=True, return_targets=False)
        assert model.engine().can_append()
        with torch.no_grad():
            model.backbone.conv_block1.conv_block1.norm.module.bias.add_(0.0)      # bumps the version: new engine
        det, = model(batch_of(3000, 4000), reset=False, return_targets=False)
        full, = model(batch_of(0, 4000), reset=True, return_targets=False)
        for key in ("boxes", "scores", "labels"):
            assert torch.equal(det[0][key], full[0][key]), key
