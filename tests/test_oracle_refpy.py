"""Pins the oracle's restatements (and the host mirror's twins) of the reference's own plain-torch functions to
outputs of those functions themselves: tests/golden/ref_py_functions.npz was produced by
tests/make_golden_refpy.py, which imports /root/reference/src with the absent third-party packages stubbed and
calls the reference code on CPU.  Exact equality unless stated."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import model as om
from oracle import ops as oo
from oracle import postprocess as opost

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_py_functions.npz"))


def T(name):
    return torch.from_numpy(G[name])


@pytest.mark.parametrize("k", [0, 1, 2])
def test_sample_features(k):
    W, H = [int(v) for v in G[f"samp{k}_WH"]]
    pos = torch.stack([T(f"samp{k}_x") / W, T(f"samp{k}_y") / H, torch.zeros(len(G[f"samp{k}_x"]))], 1)
    got = om.sample_features(pos, T(f"samp{k}_b"), T(f"samp{k}_feat"), W, H)
    # pos * W re-creates the pixel coordinate up to one fp32 rounding of x / W * W
    assert (got - T(f"samp{k}_out")).abs().max().item() <= 2e-5 * max(1.0, float(T(f"samp{k}_out").abs().max()))


def test_pooling_sizes_and_lut_params():
    from dagr_amd.model.networks import net as mirror_net
    from dagr_amd.model import utils as mirror_utils
    for k, spec in enumerate(["5x7", "4x5"]):
        assert torch.equal(om.compute_pooling_at_each_layer(spec, 4), T(f"pool_sizes{k}"))
        assert torch.equal(mirror_net.compute_pooling_at_each_layer(spec, 4), T(f"pool_sizes{k}"))
    ps = T("pool_sizes0")
    for i in range(4):
        cart_max = float(2 * ps[i][0])
        want = G["voxel_params"][i]
        got = om.voxel_size_to_params(types.SimpleNamespace(voxel_size=ps[i], cart_max=cart_max), 215, 320)
        assert list(got) == [int(want[0]), int(want[1]), want[2]]
        layer = types.SimpleNamespace(voxel_size=ps[i], transform=types.SimpleNamespace(max=cart_max))
        got = mirror_utils.voxel_size_to_params(layer, 215, 320)
        assert list(got) == [int(want[0]), int(want[1]), want[2]]


def test_consecutive_cluster_and_round_to_pixel():
    u, inv, perm, cnt = oo.consecutive_cluster(T("cc_src"))
    assert torch.equal(u, T("cc_unique")) and torch.equal(inv, T("cc_inv"))
    assert torch.equal(perm, T("cc_perm")) and torch.equal(cnt, T("cc_counts"))
    assert torch.equal(oo.round_to_pixel(T("rtp_in").clone(), T("rtp_whinv")), T("rtp_out"))


def test_to_dense():
    got = oo.to_dense(T("dense_x"), T("dense_pos"), T("dense_pooling"), T("dense_batch"), 2)
    assert torch.equal(got, T("dense_out"))


def test_to_dense_shared_cells_keep_the_highest_node_whatever_the_thread_count():
    """Duplicate (batch, cell) targets: torch's index_put order is undefined (and not sequential on CPU from ~10^4
    written elements on); the oracle pins "highest node index survives" = the reference on one thread."""
    g = torch.Generator().manual_seed(3)
    n, C, B = 400, 101, 3
    size = torch.tensor([1 / 7, 1 / 5, 1.0])
    pos = torch.rand((n, 3), generator=g) * 0.999
    pos[200:] = pos[:200]                                    # every cell hit at least twice
    batch = torch.randint(0, B, (200,), generator=g).repeat(2)
    x = torch.randn((n, C), generator=g)
    want = torch.zeros((B, C, 5, 7))
    ex, ey = (pos[:, :2] / size[:2]).t().long()
    for i in range(n):
        want[int(batch[i]), :, int(ey[i]), int(ex[i])] = x[i]
    assert torch.equal(oo.to_dense(x, pos, size, batch, B), want)


def test_head_decode():
    """decode_outputs + init_grid_and_stride as restated at the end of oracle.model.head_forward."""
    hw, strides = [tuple(int(v) for v in r) for r in G["dec_hw"]], [int(s) for s in G["dec_strides"]]
    grids, strs = [], []
    for (hs, ws), stride in zip(hw, strides):
        yv, xv = torch.meshgrid(torch.arange(hs), torch.arange(ws), indexing="ij")
        grids.append(torch.stack((xv, yv), 2).view(1, -1, 2))
        strs.append(torch.full((1, hs * ws, 1), stride))
    grid, stride = torch.cat(grids, 1).float(), torch.cat(strs, 1).float()
    assert torch.equal(grid, T("dec_grid")) and torch.equal(stride, T("dec_stride"))
    out = T("dec_raw").clone()
    out[..., :2] = (out[..., :2] + grid) * stride
    out[..., 2:4] = torch.exp(out[..., 2:4]) * stride
    assert torch.equal(out, T("dec_out"))
    # the very lines of the oracle, so that an edit there cannot drift away from this check unnoticed
    import inspect
    src = inspect.getsource(om.head_forward)
    assert "outputs[..., :2] = (outputs[..., :2] + grid_cache) * stride_cache" in src
    assert "outputs[..., 2:4] = torch.exp(outputs[..., 2:4]) * stride_cache" in src


def test_postprocess_network_output():
    res = opost.postprocess_network_output(T("post_pred"), 3, conf_thre=0.2, nms_thre=0.5, height=215, width=320)
    for i, r in enumerate(res):
        assert torch.equal(r["boxes"], T(f"post{i}_boxes"))
        assert torch.equal(r["scores"], T(f"post{i}_scores"))
        assert torch.equal(r["labels"], T(f"post{i}_labels"))
    assert sum(len(r["boxes"]) for r in res) > 10


def test_format_data_and_denormalize_pos():
    from dagr_amd.utils import buffers as mirror_buf
    from dagr_amd.utils import synthetic as syn
    d = types.SimpleNamespace(width=torch.tensor([320]), height=torch.tensor([215]), time_window=torch.tensor([1000000]),
                              pos=T("fmt_pos"), t=T("fmt_t"), x=T("fmt_x"))
    d = mirror_buf.format_data(d)       # CPU tensors: the torch branch of the mirror
    assert torch.equal(d.pos, T("fmt_out_pos")) and torch.equal(d.x, T("fmt_out_x")) and d.t is None
    pos_np = syn.format_data_np(G["fmt_pos"][:, 0], G["fmt_pos"][:, 1], G["fmt_t"], 320, 215)
    assert np.array_equal(pos_np, G["fmt_out_pos"])
    # ev_tgn.py:11-16 -- the integer pixel / microsecond coordinates the graph builder works on
    den = torch.tensor([320, 215, 1000000])
    got = (den.view(1, -1) * d.pos + 1e-3).int()
    assert torch.equal(got, T("denorm_out"))
    assert torch.equal(got[:, :2], T("fmt_pos").int()) and torch.equal(got[:, 2], T("fmt_t"))


def test_lut_construction_and_lookup():
    """MySplineConv.init_lut + message_lut (spline_conv.py:16-47), run with the oracle's spline_basis in place of
    torch_spline_conv's: the reference's table, remapping matrix and messages vs the oracle's SplineConvParams."""
    Himg, Wimg, rx, ry, Mx, My = G["lut_params"]
    p = oo.SplineConvParams(T("lut_weight"), root_weight=None)
    p.init_lut(int(Himg), int(Wimg), int(rx), float(Mx), int(ry), float(My))
    assert torch.equal(p.remap, T("lut_remap"))
    assert torch.equal(p.full_lut(), T("lut_table"))
    dx, dy = p.lut_index(T("lut_edge_attr"))
    assert int(dx.min()) >= 0 and int(dx.max()) <= 2 * int(rx) and int(dy.max()) <= 2 * int(ry)
    assert torch.equal(p.message(T("lut_xj"), T("lut_edge_attr")), T("lut_msg"))


@pytest.mark.parametrize("k,aggr", [(0, "max"), (1, "mean")])
def test_pooling_forward_glue(k, aggr):
    """Pooling.forward (pooling.py:51-97) executed from the reference with grid_cluster / scatter_max / pool_pos served
    by the oracle's primitives: cluster relabelling, coarse-edge filtering + unique, batch[perm], round_to_pixel."""
    pp = oo.PoolingParams(T(f"pool{k}_size"), 320, 215, 2, cart_max=1.0, aggr=aggr)
    x, pos, batch, ei, _ = oo.pooling(pp, T(f"pool{k}_x"), T(f"pool{k}_pos"), T(f"pool{k}_batch"), T(f"pool{k}_ei"))
    assert torch.equal(x, T(f"pool{k}_out_x")) and torch.equal(pos, T(f"pool{k}_out_pos"))
    assert torch.equal(batch, T(f"pool{k}_out_batch")) and torch.equal(ei, T(f"pool{k}_out_ei"))
    assert ei.shape[1] > 500 and x.shape[0] > 50


def test_pooling_forward_keep_temporal_ordering():
    """pooling.py:69-72 (``--keep_temporal_ordering``) executed from the reference: only coarse edges towards clusters whose
    newest member is strictly newer than the source cluster's survive."""
    pp = oo.PoolingParams(T("poolt_size"), 320, 215, 2, cart_max=1.0, aggr="max")
    x, pos, batch, ei, _ = oo.pooling(pp, T("poolt_x"), T("poolt_pos"), T("poolt_batch"), T("poolt_ei"),
                                      keep_temporal_ordering=True)
    assert torch.equal(x, T("poolt_out_x")) and torch.equal(pos, T("poolt_out_pos"))
    assert torch.equal(batch, T("poolt_out_batch")) and torch.equal(ei, T("poolt_out_ei"))
    plain = oo.pooling(pp, T("poolt_x"), T("poolt_pos"), T("poolt_batch"), T("poolt_ei"))[3]
    assert 100 < ei.shape[1] < plain.shape[1]


def test_sliding_window_graph_host_state_machine():
    """AsyncGraph / SlidingWindowGraph (graph/ev_graph.py:18-166) and graph/utils.py, run from the reference over five
    consecutive windows (400, 1, 350, 0, 300 events: batched insert, single-event insert, empty window) with
    ev_graph_cuda served by the C emulation: new edges, deleted edges, running edge list and node counts."""
    from oracle import graph as og
    Wg, Hg, Bg, Kg, Qg, rg, dtg = [int(v) for v in G["swg_params"]]
    swg = og.SlidingWindowGraph(width=Wg, height=Hg, batch_size=Bg, max_num_neighbors=Kg, max_queue_size=Qg, radius=rg,
                                delta_t_us=dtg)
    n_edges = 0
    for w in range(5):
        ret = swg.forward(G[f"swg{w}_batch"], G[f"swg{w}_pos"], return_node_counts=True, return_total_edges=True,
                          delete_nodes=True, collect_edges=True)
        edges, deleted, total, counts = ret
        assert np.array_equal(edges, G[f"swg{w}_edges"]), f"window {w}: new edges"
        want_del = G[f"swg{w}_deleted"]
        if deleted is None:
            assert want_del.shape[1] == 0
        else:
            assert np.array_equal(deleted, want_del), f"window {w}: deleted edges"
        assert np.array_equal(total, G[f"swg{w}_total"]), f"window {w}: running edge list"
        assert list(counts) == [int(v) for v in G[f"swg{w}_counts"]]
        n_edges += edges.shape[1]
    assert n_edges > 2000


def test_model_ema_mirror():
    """networks/ema.py:6-51 run from the reference (three updates of a toy module) vs the host mirror's ModelEMA."""
    from dagr_amd.model.networks.ema import ModelEMA

    def toy(seed):
        torch.manual_seed(seed)
        m = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3))
        with torch.no_grad():
            m[1].running_mean.normal_()
            m[1].running_var.uniform_(0.5, 1.5)
        return m
    ema = ModelEMA(toy(0))
    for s in (1, 2, 3):
        ema.update(toy(s))
    assert ema.updates == int(G["ema_updates"])
    for k, v in ema.ema.state_dict().items():
        want = T("ema_" + k)
        if v.dtype.is_floating_point:
            assert (v - want).abs().max().item() <= 1e-6 * (1 + want.abs().max().item()), k
        else:
            assert torch.equal(v, want), k
    assert all(not p.requires_grad for p in ema.ema.parameters()) and not ema.ema.training


def test_model_args_match_the_reference_flags_on_its_configs():
    """utils/args.py FLAGS() run from the reference on config/dagr-{n,s,m,l}-dsec.yaml (with --batch_size 8) vs the
    mirror's model_args(): every field the model constructors read has the same value."""
    import json
    from dagr_amd.utils.args import model_args
    flags = json.loads(bytes(G["flags_json"]).decode())
    read_by_model = ["radius", "max_neighbors", "edge_attr_dim", "aggr", "kernel_size", "activation", "pooling_aggr",
                     "base_width", "after_pool_width", "net_stem_width", "yolo_stem_width", "num_scales",
                     "pooling_dim_at_output", "use_image", "img_net", "no_events", "pretrain_cnn",
                     "keep_temporal_ordering", "batch_size", "dataset", "time_window_us"]
    for name, ref in flags.items():
        mine = vars(model_args(name, batch_size=8))
        for k in read_by_model:
            assert k in ref and k in mine, (name, k)
            assert mine[k] == ref[k], (name, k, mine[k], ref[k])


GM = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_py_model.npz"))
MODEL_CASES = [  # must mirror tests/make_golden_refpy_model.py:CASES
    ("s_b2", 320, 215, 2, 11, {}),
    ("s_b1_edges", 240, 180, 1, 12, {}),
    ("l_b2", 320, 215, 2, 13, dict(net_stem_width=1.0, yolo_stem_width=1.0)),
    ("s_img18_b2", 320, 215, 2, 14, dict(use_image=True, img_net="resnet18")),
    ("l_ncaltech_b2", 240, 180, 2, 15, dict(net_stem_width=1.0, yolo_stem_width=1.0, num_scales=1,
                                            dataset="ncaltech101")),
]


@pytest.mark.parametrize("name,W,H,B,seed,over", MODEL_CASES)
def test_whole_model_wiring_matches_the_reference_code(name, W, H, B, seed, over):
    """oracle.model.forward_events (what every GPU parity test compares the HIP path with) vs the decoded outputs of
    the reference's own Net / Layer / MySplineConv(LUT) / Pooling / EV_TGN / GNNHead code run on CPU over the oracle's
    primitives (tests/make_golden_refpy_model.py): same weights, same events."""
    from dagr_amd.model.networks.dagr import DAGR
    from dagr_amd.utils.testing_weights import randomize_
    args = om.default_args(batch_size=B, **over)
    torch.manual_seed(seed)
    model = randomize_(DAGR(args, height=H, width=W), seed=seed).eval()
    sd = model.state_dict()
    image_feat = cnn_out = None
    if over.get("use_image"):
        # image branch = the mirror's torch modules on CPU (HookModule + CNNHead re-declarations); the reference run
        # used its own HookModule / CNNHead code over the same ResNet class, so this also holds those two to it
        img = torch.randint(0, 256, (B, 3, H, W), generator=torch.Generator().manual_seed(seed)).float() / 255.0
        nc = om.NetConstants(args, H, W)
        with torch.no_grad():
            image_feat, outs = model.backbone.net(img)
            resized = [torch.nn.functional.interpolate(f, o) for f, o in zip(outs[-args.num_scales:], nc.output_sizes)]
            cnn_out = model.head.cnn_head(resized)
    out, _ = om.forward_events(sd, args, H, W, GM[f"{name}_x"], GM[f"{name}_y"], GM[f"{name}_t"], GM[f"{name}_p"],
                               GM[f"{name}_b"], B, image_feat=image_feat, cnn_out=cnn_out)
    want = torch.from_numpy(GM[f"{name}_out"])
    assert out.shape == want.shape
    rel = ((out - want).abs() / (1 + want.abs())).max().item()
    assert rel <= 1e-6, f"decoded outputs differ from the reference code's by {rel}"


def test_ev_tgn_reset_and_incremental_calls():
    """EV_TGN.forward (ev_tgn.py:39-58) from the reference over three calls -- reset=True, reset=False (the new nodes
    attach to the running graph: indices keep growing, the FIFO volume is kept), reset=True -- vs the oracle's
    SlidingWindowGraph driven the same way."""
    from oracle import graph as og
    from dagr_amd.utils import synthetic as syn
    W, H, B = [int(v) for v in GM["tgn_params"]]
    radius, delta_t = og.graph_params(0.05, W, 1000000)
    g = None
    for k in range(3):
        reset = bool(GM[f"tgn{k}_reset"])
        x, y, t, b = GM[f"tgn{k}_x"], GM[f"tgn{k}_y"], GM[f"tgn{k}_t"], GM[f"tgn{k}_b"]
        pos = og.denormalize_pos(syn.format_data_np(x, y, t, W, H), W, H, 1000000)
        assert np.array_equal(pos[:, 0], x) and np.array_equal(pos[:, 1], y) and np.array_equal(pos[:, 2], t)
        if g is None:
            g = og.SlidingWindowGraph(width=W, height=H, batch_size=B, max_num_neighbors=16, max_queue_size=128,
                                      radius=radius, delta_t_us=delta_t)
        elif reset:
            g.reset()
        e = g.forward(np.ascontiguousarray(b.astype(np.int32)), pos, delete_nodes=False, collect_edges=reset)
        assert np.array_equal(e.astype(np.int64), GM[f"tgn{k}_edges"]), f"call {k} (reset={reset})"


def test_in_repo_restatements_of_third_party_primitives():
    """The reference restates two of the absent third-party primitives itself (asynchronous/cartesian.py:6-15 for
    T.Cartesian, asynchronous/max_pool.py:245-252 for the 2-D voxel index of grid_cluster): the oracle's restatements
    agree with those on the model's pooling sizes."""
    pos, ei = torch.from_numpy(GM["cart_pos"]), torch.from_numpy(GM["cart_ei"])
    assert torch.equal(oo.cartesian(pos, ei, 0.0625), torch.from_numpy(GM["cart_out"]))
    ps = torch.from_numpy(GM["vox_sizes"])
    for i in range(4):
        p2 = torch.from_numpy(GM[f"vox{i}_pos"])
        pp = oo.PoolingParams(ps[i], 320, 215, 1, cart_max=1.0)
        pos4 = torch.cat([p2, torch.zeros(len(p2), 2)], 1)          # t = 0, sample 0
        got = oo.grid_cluster(pos4, pp.voxel_size, pp.start, pp.end)
        assert torch.equal(got, torch.from_numpy(GM[f"vox{i}_idx"]))


@pytest.mark.parametrize("name,W,H,B,seed,over", [
    ("train_s_b2", 240, 180, 2, 21, {}),
    ("train_l_ncaltech_b3", 240, 180, 3, 22, dict(net_stem_width=1.0, yolo_stem_width=1.0, num_scales=1,
                                                  dataset="ncaltech101")),
    ("train_s_img18_b2", 240, 180, 2, 23, dict(use_image=True, img_net="resnet18")),
])
def test_training_forward_and_gradients_match_the_reference_code(name, W, H, B, seed, over):
    """oracle.train.training_losses + torch autograd (what the GPU training tests compare the HIP path with) vs the
    reference's OWN training branch -- DAGR.forward (dagr.py:78-88) -> YOLOX.forward -> GNNHead.forward (:238-282) over
    Net / Layer / MySplineConv (spline-basis message) / Pooling / BatchNorm on batch statistics -- run on CPU by
    tests/make_golden_refpy_model.py: the six outputs and the gradients of parameters along the depth of the network."""
    from oracle import train as otr
    from dagr_amd.model.networks.dagr import DAGR
    from dagr_amd.model.utils import convert_to_training_format
    from dagr_amd.utils.testing_weights import randomize_
    args = om.default_args(batch_size=B, **over)
    torch.manual_seed(seed)
    model = randomize_(DAGR(args, height=H, width=W), seed=seed)
    sd = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() and "running" not in k
              else v.detach().clone()) for k, v in model.state_dict().items()}
    bbox, bbox_batch = torch.from_numpy(GM[f"{name}_bbox"]), torch.from_numpy(GM[f"{name}_bbox_batch"])
    # the mirror's target formatting == the reference's convert_to_training_format (run inside the golden forward)
    assert torch.equal(convert_to_training_format(bbox, bbox_batch, B), otr.convert_to_training_format(bbox, bbox_batch, B))
    extra = {}
    if over.get("use_image"):
        # the image branch = the mirror's torch modules (HookModule over the ResNet, CNNHead) in training mode, their
        # parameters re-bound to the leaf tensors of sd so that the gradients land there
        img = torch.randint(0, 256, (B, 3, H, W), generator=torch.Generator().manual_seed(seed)).float() / 255.0
        model.train()
        image_feat, cnn_out = _image_branch_functional(model, sd, img, om.NetConstants(args, H, W), args.num_scales)
        extra = dict(image_feat=image_feat, cnn_out=cnn_out, bbox0=torch.from_numpy(GM[f"{name}_bbox0"]),
                     bbox0_batch=bbox_batch)
    out = otr.training_losses(sd, args, H, W, GM[f"{name}_x"], GM[f"{name}_y"], GM[f"{name}_t"], GM[f"{name}_p"],
                              GM[f"{name}_b"], B, bbox, bbox_batch, **extra)
    want = GM[f"{name}_losses"]          # total, iou, conf, cls, l1, num_fg
    got = [float(out[0]), float(out[1]), float(out[2]), float(out[3]), float(out[4]), float(out[5])]
    assert np.allclose(got, want, rtol=2e-6, atol=1e-6), (got, want.tolist())
    out[0].backward()
    n_grads = sum(1 for v in sd.values() if v.requires_grad and v.grad is not None)
    assert n_grads == int(GM[f"{name}_n_grads"])
    for k in GM[f"{name}_grad_keys"]:
        g_ref = torch.from_numpy(GM[f"{name}_grad:{k}"])
        g = sd[str(k)].grad
        tol = 2e-5
        if f"{name}_gradnorm:{k}" in GM:       # strided sample + norm of a big tensor
            assert float(g.double().norm()) == pytest.approx(float(GM[f"{name}_gradnorm:{k}"]), rel=1e-4), k
            g = g.reshape(-1)[::g.numel() // 4096]
            tol = 2e-4                          # conv weight gradients: long fp32 reductions, thread-order dependent
        assert float((g - g_ref).abs().max()) <= tol * max(1e-6, float(g_ref.abs().max())), k


def _image_branch_functional(model, sd, img, nc, num_scales):
    """``Net.forward``'s image part (net.py:109-110) and ``GNNHead.forward``'s CNN head (dagr.py:197-206) evaluated with
    the parameters taken from ``sd`` (``torch.func.functional_call``), so that autograd reaches those leaf tensors."""
    from torch.func import functional_call
    net, head = model.backbone.net, model.head.cnn_head
    pn = {k[len("backbone.net."):]: v for k, v in sd.items() if k.startswith("backbone.net.")}
    ph = {k[len("head.cnn_head."):]: v for k, v in sd.items() if k.startswith("head.cnn_head.")}
    image_feat, outs = functional_call(net, pn, (img,))
    resized = [torch.nn.functional.interpolate(f, o) for f, o in zip(outs[-num_scales:], nc.output_sizes)]
    return image_feat, functional_call(head, ph, (resized,))
