"""GPU parity of the module-level operator API: every layer module is a ``Data -> Data`` callable like its reference twin
(model/layers/*.py), evaluated by the same C entry points the window engine uses.  Checked against the oracle's
restatements, module by module and end to end (``DAGR.forward_modules`` == the engine == the oracle)."""
import copy

import numpy as np
import pytest
import torch

from oracle import graph as og
from oracle import model as om
from oracle import ops as oo
from dagr_amd.data import Batch, Data
from dagr_amd.utils import synthetic as syn
from dagr_amd.utils.buffers import format_data
from dagr_amd.utils.testing_weights import randomize_

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _err(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    if not a.numel():
        return 0.0
    unit = max(1.0, float(b.pow(2).mean().sqrt()))
    return ((a - b).abs() / (unit + b.abs())).max().item()


def _model(W, H, B, seed=0, **over):
    from dagr_amd.model.networks.dagr import DAGR
    torch.manual_seed(seed)
    args = om.default_args(batch_size=B, **over)
    model = randomize_(DAGR(args, height=H, width=W), seed=seed).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.cuda()
    model.cache_luts(width=W, height=H, radius=args.radius)
    return args, model, sd


def _batch(W, H, B, n, seed, gen=syn.edges_window, with_image=False):
    raw, samples = [], []
    for s in range(B):
        x, y, t, p = gen(n, W, H, seed=seed + s)
        raw.append((x, y, t, p))
        d = Data(x=torch.from_numpy(p.reshape(-1, 1)), pos=torch.from_numpy(np.stack([x, y], -1)), t=torch.from_numpy(t),
                 width=W, height=H, time_window=1000000)
        if with_image:
            d.image = torch.randint(0, 256, (1, 3, H, W), generator=torch.Generator().manual_seed(seed + s), dtype=torch.uint8)
        samples.append(d)
    cat = lambda i: np.concatenate([r[i] for r in raw])
    b = np.concatenate([np.full(len(r[0]), i, np.int64) for i, r in enumerate(raw)])
    return format_data(Batch.from_data_list(samples).cuda()), (cat(0), cat(1), cat(2), cat(3), b)


def _oracle_level0(args, W, H, B, ev):
    x, y, t, p, b = ev
    nc = om.NetConstants(args, H, W)
    pos = torch.from_numpy(syn.format_data_np(x, y, t, W, H))
    r, dt = og.graph_params(args.radius, W, 1000000)
    dpos = og.denormalize_pos(pos.numpy(), W, H, 1000000)
    ei = torch.from_numpy(og.build_window_graph(dpos[:, 0], dpos[:, 1], dpos[:, 2], b.astype(np.int32), W, H, B, r, dt,
                                                K=args.max_neighbors, Q=128))
    attr = torch.clamp(oo.cartesian(pos, ei, nc.effective_radius), min=0, max=1)
    feat = torch.from_numpy(p.astype(np.float32)).view(-1, 1)
    return nc, om.Graph(torch.cat((feat, pos[:, :2]), 1), pos, torch.from_numpy(b), ei, attr)


def test_layer_pooling_and_dense_modules_match_the_oracle():
    W, H, B = 320, 215, 2
    args, model, sd = _model(W, H, B, seed=3)
    bb = model.backbone
    data, ev = _batch(W, H, B, 3000, seed=21)
    nc, g = _oracle_level0(args, W, H, B, ev)
    with torch.no_grad():
        data = bb.events_to_graph(data, reset=True)                       # EV_TGN.forward
        assert torch.equal(data.edge_index.cpu(), g.edge_index)
        data = bb.edge_attrs(data)                                        # Cartesian
        data.edge_attr = torch.clamp(data.edge_attr, min=0, max=1)
        assert torch.equal(data.edge_attr.cpu(), g.edge_attr)
        data.x = torch.cat((data.x, data.pos[:, :2]), dim=1)
        data = bb.conv_block1(data)                                       # Layer
        luts = om.level_lut_params(args, nc)
        g = om.layer(sd, "backbone.conv_block1.", g, (luts[0][0], luts[0][1], luts[0][2], H, W))
        assert _err(data.x, g.x) < TOL
        data = bb.pool1(data)                                             # Pooling
        res = oo.pooling(nc.pools[0], g.x, g.pos, g.batch, g.edge_index, exact_mean=True)
        assert data.x.shape == res[0].shape and torch.equal(data.edge_index.cpu(), res[3])
        assert torch.equal(data.pos.cpu()[:, :2], res[1][:, :2]) and _err(data.pos, res[1]) < 1e-6
        assert torch.equal(data.batch.cpu(), res[2]) and _err(data.x, res[0]) < TOL
        assert _err(data.edge_attr, res[4]) < 1e-6
        # SplineConvToDense on this level: a fresh 16 -> 5 predictor with a bias, the level's own LUT domain
        from dagr_amd.model.layers.spline_conv import SplineConvToDense
        torch.manual_seed(1)
        pred = SplineConvToDense(data.x.shape[1], 5, bias=True, args=args)
        with torch.no_grad():
            pred.bias.uniform_(-0.5, 0.5)
        pred = pred.cuda()
        rx, ry, M = luts[1]
        pred.init_lut(height=H, width=W, Mx=M, rx=rx, ry=ry)
        sd2 = {"p." + k: v.detach().cpu() for k, v in pred.state_dict().items()}
        data.pooling = bb.pool1.voxel_size[:3]
        dense = pred(data, batch_size=B)
        og_ = om.Graph(*res)
        og_.pooling = nc.pools[0].voxel_size[:3]
        want = om._pred_to_dense(sd2, "p.", og_, (rx, ry, M, H, W), B)
        bad = ((dense.cpu() - want).abs() > 1e-3).nonzero()
        assert dense.shape == want.shape and _err(dense, want) < TOL, (len(bad), bad[:6].tolist(),
                                                                       dense.cpu()[tuple(bad[0])].item() if len(bad) else None,
                                                                       want[tuple(bad[0])].item() if len(bad) else None)


@pytest.mark.parametrize("over", [{}, dict(use_image=True, img_net="resnet18")])
def test_module_by_module_forward_equals_engine_and_oracle(over):
    W, H, B = 320, 215, 2
    args, model, sd = _model(W, H, B, seed=5, **over)
    data, ev = _batch(W, H, B, 2500, seed=31, with_image=bool(over))
    with torch.no_grad():
        out_eng = model.engine().forward_data(data).clone()
        out_mod = model.forward_modules(copy.copy(data), reset=True)
    assert out_mod.shape == out_eng.shape
    grid, stride = model.engine().grid_cache, model.engine().stride_cache
    un = lambda o: torch.cat([o[..., :2] / stride - grid, torch.log(o[..., 2:4] / stride), o[..., 4:]], -1)
    # with --use_image the module path runs the plain eval-mode image modules, the engine its folded inference copy
    # (BN folded, 1x1 convs as GEMMs): PyTorch convolutions on both sides, re-associated -- 2e-4 of the map scale
    # (test_image_branch_inference_copy_matches_the_modules), which the GNN layers then carry along
    assert _err(un(out_mod), un(out_eng)) < (20 * TOL if over else TOL)
    if not over:
        out_o, _ = om.forward_events(sd, args, H, W, *ev, B, exact_pos_mean=True)
        assert _err(un(out_mod), un(out_o.cuda())) < TOL


def test_keep_temporal_ordering_runs_module_by_module_and_matches_the_oracle():
    """``--keep_temporal_ordering`` (pooling.py:69-72): a coarse edge survives only if the destination cluster's newest member
    is strictly newer than the source cluster's.  The filter lives in the Pooling modules (the oracle's restatement of it is
    pinned to the reference's code by tests/test_oracle_refpy.py); ``DAGR.forward`` of such a model evaluates module by
    module instead of through the window engine, whose fused pooling does not filter."""
    W, H, B = 320, 215, 2
    args, model, sd = _model(W, H, B, seed=8, keep_temporal_ordering=True)
    assert model.module_path_only and all(p.keep_temporal_ordering for p in
                                           (model.backbone.pool1, model.backbone.pool2, model.backbone.pool3, model.backbone.pool4))
    data, ev = _batch(W, H, B, 2500, seed=37)
    with torch.no_grad():
        out_mod = model.forward_modules(copy.copy(data), reset=True)
        det, = model(copy.copy(data), return_targets=False)
    grid, stride = model.engine().grid_cache, model.engine().stride_cache
    un = lambda o: torch.cat([o[..., :2] / stride - grid, torch.log(o[..., 2:4] / stride), o[..., 4:]], -1)
    out_o, _ = om.forward_events(sd, args, H, W, *ev, B, exact_pos_mean=True)
    assert _err(un(out_mod), un(out_o.cuda())) < TOL
    plain = copy.copy(args)
    plain.keep_temporal_ordering = False
    out_plain, _ = om.forward_events(sd, plain, H, W, *ev, B, exact_pos_mean=True)
    assert float((out_plain - out_o).abs().max()) > 1e-3            # the filter changes the result
    assert len(det) == B and all(set(d) >= {"boxes", "scores", "labels"} for d in det)
