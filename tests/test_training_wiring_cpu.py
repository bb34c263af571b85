"""CPU suite: the WIRING of the product's training forward -- ``DAGR.forward`` in training mode -> ``Net.forward`` ->
``ConvBlock`` / ``ConvBlockWithSkip`` / ``Layer`` (batch-statistics BatchNorm) -> ``GNNHead.forward`` losses -- with the
four entry points that need a GPU (graph build, SplineConv, voxel pooling, to_dense) replaced by the oracle's primitives.
What remains is exactly the host code a GPU run executes around the kernels; it must reproduce ``oracle.train`` (itself
pinned to the reference's own training branch) in losses and gradients.  The kernels' side is tests/test_training_gpu.py."""
import numpy as np
import pytest
import torch

from oracle import graph as og
from oracle import model as om
from oracle import ops as oo
from oracle import train as otr
from dagr_amd.data import Batch, Data
from dagr_amd.utils import synthetic as syn
from dagr_amd.utils.testing_weights import randomize_


@pytest.fixture
def oracle_kernels(monkeypatch):
    from dagr_amd.model.layers import _ops
    from dagr_amd.model.layers.ev_tgn import EV_TGN, denormalize_pos

    def tgn_forward(self, events, reset=True):
        W, H, T = int(events.width[0]), int(events.height[0]), int(events.time_window[0])
        r, dt = og.graph_params(self.radius, W, T)
        d = denormalize_pos(events).numpy()
        ei = og.build_window_graph(d[:, 0], d[:, 1], d[:, 2], events.batch.numpy().astype(np.int32), W, H,
                                   events.num_graphs, r, dt, K=self.max_neighbors, Q=128)
        events.edge_index = torch.from_numpy(ei)
        return events

    def conv_on_data(conv, data, norm=None, skip=None, xskip=None, relu=False):
        assert norm is None and skip is None and not relu, "training mode must call the plain conv"
        p = oo.SplineConvParams(conv.weight, conv.lin.weight, conv.bias)
        adj = oo.to_sparse(data.edge_index, data.edge_attr[:, :2], data.x.shape[0])
        return oo.spline_conv(p, data.x, adj)

    def voxel_pool(pool, data):
        pp = oo.PoolingParams(pool.voxel_size[:3], 1.0 / float(pool.wh_inv[0, 0]), 1.0 / float(pool.wh_inv[0, 1]),
                              pool.batch_size, pool.transform.max, aggr=pool.aggr)
        pp.wh_inv = pool.wh_inv
        x, pos, batch, ei, ea = oo.pooling(pp, data.x, data.pos, data.batch, data.edge_index)
        out = data.__class__()
        out.__dict__.update({k: v for k, v in data.__dict__.items() if not k.startswith("_dagr")})
        out.x, out.pos, out.batch, out.edge_index, out.edge_attr = x, pos, batch, ei, ea
        return out

    monkeypatch.setattr(EV_TGN, "forward", tgn_forward)
    monkeypatch.setattr(_ops, "conv_on_data", conv_on_data)
    monkeypatch.setattr(_ops, "voxel_pool", voxel_pool)
    monkeypatch.setattr(_ops, "to_dense", lambda x, pos, pooling, batch, batch_size: oo.to_dense(x, pos, pooling, batch, batch_size))
    monkeypatch.setattr(_ops, "sample_features",
                        lambda data, feat, width, height: om.sample_features(data.pos, data.batch, feat, width, height))


@pytest.mark.parametrize("over", [{}, dict(num_scales=1, dataset="ncaltech101")], ids=["two_scales", "ncaltech_one_scale"])
def test_training_forward_wiring_matches_the_oracle(oracle_kernels, over):
    from dagr_amd.model.networks.dagr import DAGR
    from dagr_amd.utils.buffers import format_data
    W, H, B, seed = 240, 180, 3, 7
    torch.manual_seed(seed)
    args = om.default_args(batch_size=B, **over)
    model = randomize_(DAGR(args, height=H, width=W), seed=seed).train()
    sd = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() and "running" not in k
              else v.detach().clone()) for k, v in model.state_dict().items()}
    samples, raw = [], []
    for s in range(B):
        x, y, t, p = syn.edges_window(1500, W, H, seed=70 + s)
        raw.append((x, y, t, p))
        boxes = np.array([[20.0 + 30 * s, 30.0, 80.0, 60.0, s % 2, 1, 0], [100.0, 50.0 + 10 * s, 50.0, 70.0, 1, 1, 0]],
                         dtype=np.float32)[:1 + s % 2]
        if s == 2:
            boxes = boxes[:0]                                  # a sample without boxes
        samples.append(Data(x=torch.from_numpy(p.reshape(-1, 1)), pos=torch.from_numpy(np.stack([x, y], -1)),
                            t=torch.from_numpy(t), width=W, height=H, time_window=1000000,
                            bbox=torch.from_numpy(boxes).reshape(-1, 7), sequence=f"s{s}"))
    batch = Batch.from_data_list(samples, follow_batch=["bbox"])
    ev = [np.concatenate([r[k] for r in raw]) for k in range(4)]
    b = np.concatenate([np.full(len(r[0]), i, np.int64) for i, r in enumerate(raw)])
    ref = otr.training_losses(sd, args, H, W, ev[0], ev[1], ev[2], ev[3], b, B, batch.bbox, batch.bbox_batch)
    ref[0].backward()
    out = model(format_data(batch))
    assert set(out) == {"total_loss", "iou_loss", "l1_loss", "conf_loss", "cls_loss", "num_fg"}
    got = [float(out[k]) for k in ("total_loss", "iou_loss", "conf_loss", "cls_loss", "l1_loss", "num_fg")]
    assert np.allclose(got, [float(v) for v in ref], rtol=1e-5, atol=1e-6), (got, [float(v) for v in ref])
    out["total_loss"].backward()
    params = dict(model.named_parameters())
    n = 0
    for k, v in sd.items():
        if v.requires_grad and v.grad is not None:
            g = params[k].grad
            assert g is not None, k
            assert float((g - v.grad).abs().max()) <= 1e-4 * max(1e-6, float(v.grad.abs().max())), k
            n += 1
        elif k in params:
            assert params[k].grad is None, f"{k} got a gradient but has none in the oracle"
    assert n >= 60
    bn = model.backbone.conv_block1.conv_block1.norm.module
    assert int(bn.num_batches_tracked) == 1


def test_exact_codes_recover_the_pixel_offsets_at_every_level():
    """Training-mode convs evaluate the basis at offset / den + 0.5; the offsets come back out of the float Cartesian
    attributes (pixel-grid positions at every level: events are pixels, pooled positions are floored to pixels)."""
    from dagr_amd.model.layers import _ops
    from dagr_amd.model.networks.net import Net
    W, H = 240, 180
    net = Net(om.default_args(batch_size=2), height=H, width=W)
    g = torch.Generator().manual_seed(0)
    maxima = [net.edge_attrs.max] + [getattr(net, f"pool{k}").transform.max for k in (1, 2, 3, 4)]
    reach = [3, 12, 30, 60, 110]
    for M, r in zip(maxima, reach):
        n = 4000
        px = torch.stack([torch.randint(0, W, (n,), generator=g), torch.randint(0, H, (n,), generator=g)], 1)
        src = torch.randint(0, n, (20000,), generator=g)
        dst = torch.randint(0, n, (20000,), generator=g)
        d = px[src] - px[dst]
        ok = (d[:, 0].abs() <= r) & (d[:, 1].abs() <= r)
        src, dst, d = src[ok], dst[ok], d[ok]
        pos = torch.cat([px.float() / torch.tensor([W, H]), torch.rand(n, 1, generator=g)], 1)
        attr = _ops.cartesian(pos, torch.stack([src, dst]), M)
        code, den_x, den_y = _ops.exact_codes(attr, M, W, H)
        assert torch.equal((code & 0xffff) - _ops.EXACT_R, d[:, 0].int())
        assert torch.equal((code >> 16) - _ops.EXACT_R, d[:, 1].int())
        pseudo = d[:, 0].float() / den_x + 0.5                      # what the kernels evaluate
        assert float((pseudo - attr[:, 0]).abs().max()) < 2e-6


def test_training_forward_wiring_with_the_image_branch(oracle_kernels):
    """``--use_image`` training wiring on CPU (train_dsec.py): detached feature sampling, detached CNN-head logits in the
    hybrid sum, the image branch's own loss against ``bbox0``, the element-wise sum of the two loss tuples."""
    from tests.test_oracle_refpy import _image_branch_functional
    from dagr_amd.model.networks.dagr import DAGR
    from dagr_amd.utils.buffers import format_data
    W, H, B, seed = 240, 180, 2, 9
    torch.manual_seed(seed)
    args = om.default_args(batch_size=B, use_image=True, img_net="resnet18")
    model = randomize_(DAGR(args, height=H, width=W), seed=seed).train()
    sd = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() and "running" not in k
              else v.detach().clone()) for k, v in model.state_dict().items()}
    samples, raw = [], []
    img = torch.randint(0, 256, (B, 3, H, W), generator=torch.Generator().manual_seed(seed), dtype=torch.uint8)
    for s_ in range(B):
        x, y, t, p = syn.edges_window(1200, W, H, seed=90 + s_)
        raw.append((x, y, t, p))
        boxes = torch.tensor([[30.0 + 40 * s_, 30.0, 80.0, 60.0, float(s_ % 2), 1, 0]])
        samples.append(Data(x=torch.from_numpy(p.reshape(-1, 1)), pos=torch.from_numpy(np.stack([x, y], -1)),
                            t=torch.from_numpy(t), width=W, height=H, time_window=1000000, bbox=boxes,
                            bbox0=boxes - torch.tensor([[3.0, 3.0, 0, 0, 0, 0, 0]]), image=img[s_:s_ + 1], sequence=f"s{s_}"))
    batch = Batch.from_data_list(samples, follow_batch=["bbox", "bbox0"])
    ev = [np.concatenate([r[k] for r in raw]) for k in range(4)]
    b = np.concatenate([np.full(len(r[0]), i, np.int64) for i, r in enumerate(raw)])
    ref_model = DAGR(args, height=H, width=W).train()
    image_feat, cnn_out = _image_branch_functional(ref_model, sd, img.float() / 255.0, om.NetConstants(args, H, W),
                                                   args.num_scales)
    ref = otr.training_losses(sd, args, H, W, ev[0], ev[1], ev[2], ev[3], b, B, batch.bbox, batch.bbox_batch,
                              image_feat=image_feat, cnn_out=cnn_out, bbox0=batch.bbox0, bbox0_batch=batch.bbox0_batch)
    ref[0].backward()
    out = model(format_data(batch))
    got = [float(out[k]) for k in ("total_loss", "iou_loss", "conf_loss", "cls_loss", "l1_loss", "num_fg")]
    assert np.allclose(got, [float(v) for v in ref], rtol=1e-5, atol=1e-6), (got, [float(v) for v in ref])
    out["total_loss"].backward()
    params = dict(model.named_parameters())
    n_img = 0
    for k, v in sd.items():
        if v.requires_grad and v.grad is not None:
            g = params[k].grad
            assert g is not None, k
            # (+ 5e-7: a conv bias in front of a BatchNorm has a gradient that is zero in exact arithmetic -- both sides hold
            # 1e-8-sized rounding noise there, and a relative bar on noise compares nothing)
            assert float((g - v.grad).abs().max()) <= 2e-4 * float(v.grad.abs().max()) + 5e-7, k
            n_img += k.startswith("backbone.net.") or "cnn_head" in k
    assert n_img >= 60
