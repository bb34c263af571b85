"""CPU oracle for the network wiring of the hot path (events-only and image-fused), torch fp32 CPU.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Consumes a ``state_dict`` with the reference's
key layout and restates, op for op:
  * ``Net.__init__`` constants and ``Net.forward``           model/networks/net.py:19-28,31-106,108-190
  * ``sample_features``                                       model/networks/net.py:193-221
  * ``Layer`` / ``ConvBlock`` / ``ConvBlockWithSkip``         model/layers/conv.py:10-72
  * ``DAGR.cache_luts`` / ``voxel_size_to_params``            model/networks/dagr.py:37-72, model/utils.py:112-116
  * ``GNNHead.process_feature`` / eval ``forward`` / decode   model/networks/dagr.py:179-236,283-312
The graph itself comes from oracle/graph.py.
"""
import types

import numpy as np
import torch

from . import graph as og
from . import ops


def default_args(**over):
    """config/dagr-s-dsec.yaml (network part) + flags of utils/args.py."""
    a = dict(radius=0.01, time_window_us=1000000, max_neighbors=16, batch_size=1, activation="relu",
             edge_attr_dim=2, aggr="sum", kernel_size=5, pooling_aggr="max", base_width=0.5, after_pool_width=1,
             net_stem_width=0.5, yolo_stem_width=0.5, num_scales=2, pooling_dim_at_output="5x7", dataset="dsec",
             use_image=False, no_events=False, pretrain_cnn=False, keep_temporal_ordering=False, img_net="resnet50")
    a.update(over)
    return types.SimpleNamespace(**a)


def compute_pooling_at_each_layer(pooling_dim_at_output, num_layers):  # net.py:19-28
    py, px = map(int, pooling_dim_at_output.split("x"))
    pooling_base = torch.tensor([1.0 / px, 1.0 / py, 1.0 / 1])
    poolings = []
    for i in range(num_layers):
        pooling = pooling_base / 2 ** (3 - i)
        pooling[-1] = 1
        poolings.append(pooling)
    return torch.stack(poolings)


def net_channels(args):  # net.py:35-38
    return [1, int(args.base_width * 32), int(args.after_pool_width * 64), int(args.net_stem_width * 128),
            int(args.net_stem_width * 128), int(args.net_stem_width * 128)]


class NetConstants:
    """Everything ``Net.__init__`` derives from (args, height, width) (net.py:31-106)."""

    def __init__(self, args, height, width):
        self.height, self.width = height, width
        ch = net_channels(args)
        self.channels = ch
        self.output_channels = ch[1:]
        self.out_channels = ch[1:][-2:]
        self.input_channels = ch[:-1]
        self.feature_channels = ch[1:] if args.use_image else []  # HookModule feature_channels (net.py:49)
        if args.use_image:
            self.input_channels = [self.input_channels[i] + self.feature_channels[i] for i in range(5)]
        self.num_classes = dict(dsec=2, ncaltech101=100).get(args.dataset, 2)
        self.poolings = compute_pooling_at_each_layer(args.pooling_dim_at_output, num_layers=4)
        self.max_vals_for_cartesian = 2 * self.poolings[:, :2].max(-1).values
        strides = torch.ceil(self.poolings[-2:, 1] * height).numpy().astype("int32").tolist()
        self.strides = strides[-args.num_scales:]
        self.effective_radius = 2 * float(int(args.radius * width + 2) / width)
        # (cart max of the transform of pool k) -- net.py:77,83,89,95
        self.cart_max = [2 * self.effective_radius, self.max_vals_for_cartesian[1], self.max_vals_for_cartesian[2],
                         self.max_vals_for_cartesian[3]]
        self.pool_aggr = [args.pooling_aggr, args.pooling_aggr, args.pooling_aggr, "mean"]  # net.py:96-97
        self.pools = [ops.PoolingParams(self.poolings[i], width, height, args.batch_size, self.cart_max[i],
                                        self.pool_aggr[i]) for i in range(4)]
        # get_output_sizes, net.py:103-106
        self.output_sizes = [(1 / p.voxel_size[:2] + 1e-3).int().numpy().tolist()[::-1] for p in self.pools[2:]]
        self.num_scales = args.num_scales


def voxel_size_to_params(pool, height, width):  # model/utils.py:112-116
    rx = int(np.ceil(2 * pool.voxel_size[0].cpu().numpy() * width))
    ry = int(np.ceil(2 * pool.voxel_size[1].cpu().numpy() * height))
    return rx, ry, pool.cart_max


def level_lut_params(args, nc):
    """(rx, ry, M) for graph levels 0..4 as ``DAGR.cache_luts`` assigns them (dagr.py:37-72).
    Level 0: the event graph; level k>=1: after pool k.  Head "1" uses level 3's, head "2" level 4's -- whatever
    level they consume (see head_forward)."""
    M0 = 2 * float(int(args.radius * nc.width + 2) / nc.width)
    r0 = int(args.radius * nc.width + 1)
    levels = [(r0, r0, M0)]
    for k in range(4):
        levels.append(voxel_size_to_params(nc.pools[k], nc.height, nc.width))
    return levels


# --------------------------------------------------------------------------- parameter access
def conv_params(sd, prefix):
    bias = sd.get(prefix + "bias")
    return ops.SplineConvParams(sd[prefix + "weight"], sd[prefix + "lin.weight"], bias)


def bn_params(sd, prefix):
    return {k: sd[prefix + "module." + k] for k in ("weight", "bias", "running_mean", "running_var")}


class Graph:
    """Just enough of PyG ``Data`` for the oracle."""

    def __init__(self, x, pos, batch, edge_index, edge_attr):
        self.x, self.pos, self.batch, self.edge_index, self.edge_attr = x, pos, batch, edge_index, edge_attr
        self.adj = None
        self.pooling = None

    def shallow_copy(self):  # model/utils.py:158-166 (adj_t is NOT carried over)
        g = Graph(self.x.clone(), self.pos, self.batch, self.edge_index, self.edge_attr)
        g.pooling = self.pooling
        return g


def _conv(sd, prefix, g, lut):
    """``MySplineConv.forward`` (spline_conv.py:49-62)."""
    p = conv_params(sd, prefix)
    if lut is not None:
        rx, ry, M, H, W = lut
        p.init_lut(height=H, width=W, Mx=M, rx=rx, ry=ry)
    if g.adj is None:
        g.edge_attr = g.edge_attr[:, :ops.DIM]
        g.adj = ops.to_sparse(g.edge_index, g.edge_attr, g.x.shape[0])
    g.x = ops.spline_conv(p, g.x, g.adj)
    return g


def conv_block(sd, prefix, g, lut):  # conv.py:23-28
    g = _conv(sd, prefix + "conv.", g, lut)
    g.x = ops.batch_norm_eval(g.x, bn_params(sd, prefix + "norm."))
    g.x = torch.relu(g.x)
    return g


def conv_block_with_skip(sd, prefix, g, g_skip, lut):  # conv.py:47-56
    g = _conv(sd, prefix + "conv.", g, lut)
    skip = g_skip.x @ sd[prefix + "lin.mlp.weight"].t()
    skip = ops.batch_norm_eval(skip, bn_params(sd, prefix + "norm_skip."))
    g.x = ops.batch_norm_eval(g.x, bn_params(sd, prefix + "norm."))
    g.x = torch.relu(g.x + skip)
    return g


def layer(sd, prefix, g, lut):  # conv.py:68-72
    g_skip = g.shallow_copy()
    g = conv_block(sd, prefix + "conv_block1.", g, lut)
    return conv_block_with_skip(sd, prefix + "conv_block2.", g, g_skip, lut)


def sample_features(pos, batch, image_feat, width, height):
    """``sample_features`` / ``_sample_features`` (net.py:193-221): 3-D grid_sample, align_corners=True."""
    x = pos[:, 0] * width
    y = pos[:, 1] * height
    b = batch.float()
    x = 2 * x / (width - 1) - 1
    y = 2 * y / (height - 1) - 1
    bs = image_feat.shape[0]
    bs = bs if bs > 1 else 2
    b = 2 * b / (bs - 1) - 1
    grid = torch.stack((x, y, b), dim=-1).view(1, 1, 1, -1, 3)
    feat = image_feat.permute(1, 0, 2, 3).unsqueeze(0)
    s = torch.nn.functional.grid_sample(feat, grid=grid, mode="bilinear", align_corners=True)
    return s.view(feat.shape[1], -1).t()


def net_forward(sd, args, nc, pos, feat, batch, edge_index, use_lut=True, image_feat=None, trace=None,
                exact_pos_mean=False):
    """``Net.forward`` (net.py:108-190) after ``events_to_graph``.  pos fp32[N,3] normalised,
    feat fp32[N,1], batch int64[N], edge_index int64[2,E].  Returns [out3, out4][-num_scales:].
    ``trace`` (dict) receives per-stage tensors for layer-by-layer parity tests."""
    H, W = nc.height, nc.width
    luts = level_lut_params(args, nc) if use_lut else [None] * 5

    def lut(k):
        return None if not use_lut else (luts[k][0], luts[k][1], luts[k][2], H, W)

    def rec(name, g):
        if trace is not None:
            trace[name] = dict(x=g.x.clone(), pos=g.pos.clone(), batch=g.batch.clone(),
                               edge_index=g.edge_index.clone())

    x = feat
    if image_feat is not None:
        x = torch.cat((x, sample_features(pos, batch, image_feat[0], W, H)), dim=1)
        if trace is not None:
            trace["x0_image"] = x.clone()
    edge_attr = ops.cartesian(pos, edge_index, nc.effective_radius)  # net.py:122
    edge_attr = torch.clamp(edge_attr, min=0, max=1)                 # net.py:123
    g = Graph(torch.cat((x, pos[:, :2]), dim=1), pos, batch, edge_index, edge_attr)
    g = layer(sd, "backbone.conv_block1.", g, lut(0))
    rec("layer1", g)
    names = ["backbone.layer2.", "backbone.layer3.", "backbone.layer4.", "backbone.layer5."]
    outs = []
    for k in range(4):
        if image_feat is not None:
            g.x = torch.cat((g.x, sample_features(g.pos, g.batch, image_feat[k + 1], W, H)), dim=1)
        res = ops.pooling(nc.pools[k], g.x, g.pos, g.batch, g.edge_index, exact_mean=exact_pos_mean,
                          keep_temporal_ordering=bool(getattr(args, "keep_temporal_ordering", False)))
        if res is not None:  # pooling.py:52-53 returns the input untouched on an empty graph
            g = Graph(*res)
        rec(f"pool{k + 1}", g)
        g.x = torch.cat((g.x, g.pos[:, :2]), dim=1)
        g = layer(sd, names[k], g, lut(k + 1))
        rec(f"layer{k + 2}", g)
        if k == 2:
            out3 = g.shallow_copy()
            out3.pooling = nc.pools[2].voxel_size[:3]
            outs.append(out3)
        if k == 3:
            g.pooling = nc.pools[3].voxel_size[:3]
            outs.append(g)
    return outs[-nc.num_scales:]


def _pred_to_dense(sd, prefix, g, lut, batch_size):
    """``SplineConvToDense.forward`` (spline_conv.py:110-118)."""
    g = _conv(sd, prefix, g, lut)
    return ops.to_dense(g.x, g.pos, g.pooling, g.batch, batch_size)


def head_process_feature(sd, scale, g, lut, batch_size):
    """``GNNHead.process_feature`` (dagr.py:179-190)."""
    s = str(scale)
    g = conv_block(sd, f"head.stem{s}.", g, lut)
    cls_feat = conv_block(sd, f"head.cls_conv{s}.", g.shallow_copy(), lut)
    reg_feat = conv_block(sd, f"head.reg_conv{s}.", g, lut)
    cls_output = _pred_to_dense(sd, f"head.cls_pred{s}.", cls_feat, lut, batch_size)
    reg_output = _pred_to_dense(sd, f"head.reg_pred{s}.", reg_feat.shallow_copy(), lut, batch_size)
    obj_output = _pred_to_dense(sd, f"head.obj_pred{s}.", reg_feat, lut, batch_size)
    return cls_output, reg_output, obj_output


def head_forward(sd, args, nc, outs, batch_size, use_lut=True, cnn_out=None, trace=None):
    """Eval branch of ``GNNHead.forward`` (dagr.py:192-236,283-312): returns decoded
    ``[B, n_anchors_all, 5+num_classes]`` and the raw (cls, reg, obj) maps per scale."""
    H, W = nc.height, nc.width
    luts = level_lut_params(args, nc) if use_lut else None
    hybrid = []
    raw = []
    for k, g in enumerate(outs):
        # DAGR.cache_luts (dagr.py:52-72) ties the LUT to the head's NAME, not to the level it consumes: stem1 /
        # cls_conv1 / reg_conv1 / *_pred1 always get the pool3 domain, *2 the pool4 domain.  With num_scales = 1
        # (config/dagr-l-ncaltech.yaml) head "1" consumes out4, whose edge attributes were normalised by pool4's
        # Cartesian maximum, and looks them up in the pool3 table: message_lut's index becomes a
        # half-resolution one (trunc(dx_pix * M3/M4 + rx3 + 1e-3)).
        lvl = 3 + k
        lut = None if not use_lut else (luts[lvl][0], luts[lvl][1], luts[lvl][2], H, W)
        cls_o, reg_o, obj_o = head_process_feature(sd, k + 1, g, lut, batch_size)
        if cnn_out is not None:  # dagr.py:219-222,230-234
            cls_o[:batch_size] += cnn_out["cls_output"][k]
            reg_o[:batch_size] += cnn_out["reg_output"][k]
            obj_o[:batch_size] += cnn_out["obj_output"][k]
        raw.append((cls_o.clone(), reg_o.clone(), obj_o.clone()))
        hybrid.append(torch.cat([reg_o, obj_o.sigmoid(), cls_o.sigmoid()], 1))  # collect_outputs, dagr.py:300-302
    hw = [o.shape[-2:] for o in hybrid]
    outputs = torch.cat([o.flatten(start_dim=2) for o in hybrid], dim=2).permute(0, 2, 1).contiguous()
    # decode_outputs (dagr.py:306-312) + init_grid_and_stride (model/utils.py:119-134)
    grids, strides = [], []
    for (hs, ws), stride in zip(hw, nc.strides):
        yv, xv = torch.meshgrid(torch.arange(hs), torch.arange(ws), indexing="ij")
        grid = torch.stack((xv, yv), 2).view(1, -1, 2)
        grids.append(grid)
        strides.append(torch.full((1, grid.shape[1], 1), stride))
    grid_cache = torch.cat(grids, dim=1).float()
    stride_cache = torch.cat(strides, dim=1).float()
    outputs[..., :2] = (outputs[..., :2] + grid_cache) * stride_cache
    outputs[..., 2:4] = torch.exp(outputs[..., 2:4]) * stride_cache
    if trace is not None:
        trace["head_raw"] = raw
    return outputs, raw


def forward_events(sd, args, height, width, x, y, t, p, b, batch_size, use_lut=True, trace=None,
                   time_window=1000000, image_feat=None, cnn_out=None, exact_pos_mean=False, edge_index=None):
    """Whole hot path for one window batch from raw events (int arrays): format_data
    (utils/buffers.py:33-44) -> EV_TGN (layers/ev_tgn.py:39-58) -> Net -> GNNHead eval."""
    nc = NetConstants(args, height, width)
    pos = torch.from_numpy(np.stack([x.astype(np.float32) / np.float32(width),
                                     y.astype(np.float32) / np.float32(height),
                                     t.astype(np.float32) / np.float32(time_window)], -1).astype(np.float32))
    feat = torch.from_numpy(p.astype(np.float32)).view(-1, 1)
    batch = torch.from_numpy(b.astype(np.int64))
    if edge_index is None:
        r, dt = og.graph_params(args.radius, width, time_window)
        dpos = og.denormalize_pos(pos.numpy(), width, height, time_window)
        ei = og.build_window_graph(dpos[:, 0], dpos[:, 1], dpos[:, 2], b.astype(np.int32), width, height, batch_size,
                                   r, dt, K=args.max_neighbors, Q=128)
        ei = torch.from_numpy(ei)
    else:       # the caller built the graph (bench.py's CPU leg: the same C builder, one thread per sample)
        ei = edge_index
    if trace is not None:
        trace["edge_index"] = ei.clone()
    outs = net_forward(sd, args, nc, pos, feat, batch, ei, use_lut=use_lut, image_feat=image_feat, trace=trace,
                       exact_pos_mean=exact_pos_mean)
    return head_forward(sd, args, nc, outs, batch_size, use_lut=use_lut, cnn_out=cnn_out, trace=trace)
