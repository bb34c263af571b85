"""WindowEngine -- sequences the HIP kernels of libdagr_hip for one batch of event windows.

Mirrors the control flow of ``Net.forward`` (src/dagr/model/networks/net.py:108-190) and the eval
branch of ``GNNHead.forward`` (src/dagr/model/networks/dagr.py:192-236,283-312), but every graph
level lives in pre-sized device buffers with device-side node/edge counts, so a window runs without
host synchronisation: graph build -> fused level-0 convs -> voxel pooling -> (tap aggregation +
GEMM) per pooled conv -> dense head maps.  PyTorch supplies memory and the stream only.
"""
import contextlib
import ctypes
import gc
import os
import types

import numpy as np
import torch

from . import _lib
from .graph.ev_graph import WindowGraphBuilder


def _f32(v):
    """Value of ``v`` (python float or 0-dim tensor) as the fp32 torch would compute with."""
    return float(torch.as_tensor(v, dtype=torch.float32))


def _same_domain(a, b):
    return (a["rx"], a["ry"], a["den_x"], a["den_y"]) == (b["rx"], b["ry"], b["den_x"], b["den_y"]) \
        and torch.equal(a["remap"], b["remap"])


def _bn_affine(bn):
    m = bn.module
    scale = (m.weight / torch.sqrt(m.running_var + m.eps)).detach().float()
    shift = (m.bias - m.running_mean * scale).detach().float()
    return scale, shift



@contextlib.contextmanager
def _capture(graph):
    """``torch.cuda.graph`` with Python's cyclic collector paused for the length of the capture.  Entering the context
    collects everything that is garbage already; a collection that starts in the middle of the capture would run
    finalizers (of objects a previous engine left in a cycle, of temporaries of the body) that may call into the HIP runtime
    while the stream is capturing -- one ad-hoc ordering of the GPU test files aborted inside a captured tail that way."""
    was_enabled = gc.isenabled()
    with torch.cuda.graph(graph):
        gc.disable()
        try:
            yield
        finally:
            if was_enabled:
                gc.enable()

class _ConvPack:
    """Packed weights of one fused contraction: Wm[K, N], bias[N] (BN folded), plus shape info."""

    def __init__(self, cin, cskip, Wm, bias, relu):
        self.K, self.N = Wm.shape
        self.ldw = (self.N + 7) // 8 * 8      # zero-padded columns: rows stay 32-byte aligned for the MFMA GEMM
        if self.ldw != self.N:
            Wm = torch.cat([Wm, torch.zeros((self.K, self.ldw - self.N), dtype=Wm.dtype, device=Wm.device)], 1)
        self.cin, self.cskip, self.Wm, self.bias, self.relu = cin, cskip, Wm.contiguous(), bias.contiguous(), relu
        # MFMA operand order for dagr_spline_conv_fused: Wq[c][g][l][j] = W[16g + 4j + (l>>4)][16c + (l&15)]
        K16, N16 = (self.K + 15) // 16 * 16, (self.N + 15) // 16 * 16
        Wp = torch.zeros((K16, N16), dtype=Wm.dtype, device=Wm.device)
        Wp[: self.K, : self.N] = Wm[:, : self.N]
        # [g, j, kk, c, nn] -> [c, g, kk, nn, j]  (lane l = 16 kk + nn)
        self.Wq = Wp.view(K16 // 16, 4, 4, N16 // 16, 16).permute(3, 0, 2, 4, 1).contiguous()


def _pack_generic(convs, norms, skip=None, relu=True, device="cuda"):
    """Columns of several convs that read the same input are concatenated (N = sum of couts).
    rows: [25*cin taps (kx + 5*ky major, then input channel) | cin root | cskip skip]."""
    cols, biases = [], []
    cin = convs[0].in_channels
    for conv, norm in zip(convs, norms):
        W = conv.weight.detach().float()            # [25, cin, cout]
        root = conv.lin.weight.detach().float()     # [cout, cin]
        cout = W.shape[2]
        if norm is not None:
            scale, shift = _bn_affine(norm)
        else:
            scale, shift = torch.ones(cout, device=W.device), torch.zeros(cout, device=W.device)
        if conv.bias is not None:
            shift = shift + conv.bias.detach().float() * scale
        block = torch.cat([W.reshape(25 * cin, cout), root.t()], 0) * scale.view(1, -1)
        if skip is not None:
            lin, norm_skip = skip
            s_scale, s_shift = _bn_affine(norm_skip)
            block = torch.cat([block, lin.mlp.weight.detach().float().t() * s_scale.view(1, -1)], 0)
            shift = shift + s_shift
        cols.append(block)
        biases.append(shift)
    cskip = skip[0].mlp.in_features if skip is not None else 0
    return _ConvPack(cin, cskip, torch.cat(cols, 1).to(device), torch.cat(biases).to(device), relu)


def _pack_l0(conv, norm, win, skip=None, device="cuda", cols_in=None, cols_skip=None):
    """Level-0 packing: rows [(a + tx*b)*cin + i | root | skip] x 16 over the tx x ty tap window.
    ``cols_in`` / ``cols_skip``: reference channel behind every column of the input / skip-input rows as the engine
    lays them out (identity when None)."""
    win_x, tx, win_y, ty = win
    W = conv.weight.detach().float()
    cin, cout = W.shape[1], W.shape[2]
    if cout != 16:
        raise NotImplementedError("the level-0 kernels are specialised for 16 output channels (base_width = 0.5)")
    cols_in = list(range(cin)) if cols_in is None else list(cols_in)
    scale, shift = _bn_affine(norm)
    rows = []
    for b in range(ty):
        for a in range(tx):
            rows.append(W[(win_x + a) + 5 * (win_y + b)][cols_in] * scale.view(1, -1))
    rows.append(conv.lin.weight.detach().float().t()[cols_in] * scale.view(1, -1))
    cskip = 0
    if skip is not None:
        lin, norm_skip = skip
        s_scale, s_shift = _bn_affine(norm_skip)
        ws = lin.mlp.weight.detach().float().t()
        if cols_skip is not None:
            ws = ws[list(cols_skip)]
        rows.append(ws * s_scale.view(1, -1))
        shift = shift + s_shift
        cskip = lin.mlp.in_features
    return cin, cskip, torch.cat(rows, 0).contiguous().to(device), shift.contiguous().to(device)


class _Conv1x1Gemm(torch.nn.Module):
    """A folded 1x1 Conv2d of the channels-last image branch as one GEMM ([B*H*W, Cin] x [Cin, Cout] + bias):
    hipBLASLt's fp32 GEMM is ~25 % faster than MIOpen's implicit-GEMM kernels on these shapes
    (tools/conv1x1_bench.py).  Output stays channels-last (a permuted view of the NHWC result)."""

    def __init__(self, conv, relu=False):
        super().__init__()
        cout, cin = conv.weight.shape[:2]
        self.stride = conv.stride[0]
        self.relu = relu       # ReLU in the GEMM epilogue (hipBLASLt) instead of a separate pass
        self.register_buffer("wt", conv.weight.detach().reshape(cout, cin).t().contiguous())
        self.register_buffer("b", conv.bias.detach().clone() if conv.bias is not None else torch.zeros(cout, device=conv.weight.device))

    def forward(self, x, residual=None):
        """``residual`` (a channels-last map of the output's shape): relu(conv(x) + bias + residual) in ONE library GEMM
        (dagr_gemm_epilogue: hipBLASLt with beta = 1 on the residual, bias and ReLU in the epilogue) -- the join of a
        bottleneck (net_img.py:47-48) without a pass of its own."""
        if self.stride != 1:
            x = x[:, :, ::self.stride, ::self.stride]
        B, C, H, W = x.shape
        a = x.permute(0, 2, 3, 1).reshape(-1, C)
        N = self.wt.shape[1]
        r = None if residual is None else residual.permute(0, 2, 3, 1)
        if _LT["ok"] and a.is_cuda and a.dtype == torch.float32 and a.stride(1) == 1 \
                and (r is None or (r.is_contiguous() and tuple(r.shape) == (B, H, W, N))):
            # every 1x1 conv of the branch through the same entry: bias (+ residual) (+ ReLU) in the GEMM's epilogue, the
            # library kernel picked by timing its candidates on this shape once (dagr_gemm_epilogue)
            y = torch.empty((a.shape[0], N), dtype=torch.float32, device=a.device)
            ws = _lt_workspace(a.device)
            act = 1 if (self.relu or r is not None) else 0
            rc = _lib.lib().dagr_gemm_epilogue(_lib.ptr(a), a.shape[0], C, a.stride(0), _lib.ptr(self.wt), N,
                                               _lib.ptr(self.b), _lib.ptr(r), N, act, _lib.ptr(y), N, _lib.ptr(ws),
                                               ws.numel(), _lib.cur_stream(a.device))
            if rc == 0:
                return y.view(B, H, W, -1).permute(0, 3, 1, 2)
            _LT["ok"] = False      # the library has no kernel for this epilogue here: torch's GEMM (+ a join pass) from now on
        if residual is not None:
            y = torch.addmm(self.b, a, self.wt).view(B, H, W, -1).permute(0, 3, 1, 2)
            return _add_relu_(y, residual)
        y = torch._addmm_activation(self.b, a, self.wt) if self.relu else torch.addmm(self.b, a, self.wt)
        return y.view(B, H, W, -1).permute(0, 3, 1, 2)


_LT = {"ok": os.environ.get("DAGR_LT_RESIDUAL", "1") != "0", "ws": {}}


def _lt_workspace(device):
    """Scratch for the library GEMMs of ``dagr_gemm_epilogue`` (allocated once per device, before any capture)."""
    key = str(device)
    if key not in _LT["ws"]:
        _LT["ws"][key] = torch.empty(int(_lib.lib().dagr_gemm_epilogue_workspace_bytes()), dtype=torch.uint8, device=device)
    return _LT["ws"][key]


def _gemmify_1x1(module):
    for name, child in list(module.named_children()):
        if isinstance(child, torch.nn.Conv2d) and child.kernel_size == (1, 1) and child.groups == 1 \
                and child.padding == (0, 0) and child.stride[0] == child.stride[1]:
            setattr(module, name, _Conv1x1Gemm(child))
        else:
            _gemmify_1x1(child)


def _add_relu_(y, z):
    """y = relu(y + z) in place, one pass (dagr_add_relu) when both maps share one dense layout."""
    if y.is_cuda and y.dtype == torch.float32 and z.dtype == torch.float32 and y.shape == z.shape \
            and y.stride() == z.stride() and y.data_ptr() % 16 == 0 and z.data_ptr() % 16 == 0 \
            and (y.is_contiguous() or y.is_contiguous(memory_format=torch.channels_last)):
        _lib.check(_lib.lib().dagr_add_relu(_lib.ptr(y), _lib.ptr(z), y.numel(), _lib.cur_stream(y.device)), "add_relu")
        return y
    return torch.relu_(y.add_(z))


def _bias_relu_(y, bias):
    """y = relu(y + bias[c]) in place, one pass (dagr_bias_relu) on a channels-last map."""
    C = y.shape[1]
    if y.is_cuda and y.dtype == torch.float32 and C % 4 == 0 and y.is_contiguous(memory_format=torch.channels_last) \
            and y.data_ptr() % 16 == 0 and bias.data_ptr() % 16 == 0:
        _lib.check(_lib.lib().dagr_bias_relu(_lib.ptr(y), _lib.ptr(bias), y.numel(), C, _lib.cur_stream(y.device)),
                   "bias_relu")
        return y
    return torch.relu_(y.add_(bias.view(1, -1, 1, 1)))


def _baseconv_forward(m, x):
    """yolox BaseConv.forward of the inference copy: bias-free conv, then folded-BN bias + SiLU in one pass."""
    y = m.conv(x)
    b = m._bias
    C = y.shape[1]
    if y.is_cuda and y.dtype == torch.float32 and C % 4 == 0 and y.is_contiguous(memory_format=torch.channels_last) \
            and y.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0:
        _lib.check(_lib.lib().dagr_bias_silu(_lib.ptr(y), _lib.ptr(b), y.numel(), C, _lib.cur_stream(y.device)),
                   "bias_silu")
        return y
    return torch.nn.functional.silu(y.add_(b.view(1, -1, 1, 1)), inplace=True)


def _conv_bias_relu(blk, name, x):
    """relu(conv(x) + bias): spatial convs of the inference copy run bias-free (MIOpen would add the bias in a pass of
    its own) and get bias + ReLU in one pass; the 1x1 GEMMs keep their epilogue.  (MIOpen's own conv + bias + ReLU fusion
    plan, aten::miopen_convolution_relu, was measured in round 5: it falls to a naive kernel on these fp32 NHWC shapes --
    260 ms per B = 8 forward instead of 4.7; profiles/r5_conv_phases.md.)"""
    conv = getattr(blk, name)
    b = getattr(blk, "_" + name + "_bias", None)
    out = conv(x)
    if b is not None:
        return _bias_relu_(out, b)
    if getattr(conv, "relu", False):
        return out
    return blk.relu(out)


def _bottleneck_forward(blk, x):
    """Bottleneck.forward (net_img.py:43-48) of the folded inference copy: conv1's ReLU rides in its GEMM
    epilogue, the residual join is one pass."""
    identity = x if blk.downsample is None else blk.downsample(x)
    out = _conv_bias_relu(blk, "conv1", x)
    out = _conv_bias_relu(blk, "conv2", out)
    if isinstance(blk.conv3, _Conv1x1Gemm):
        return blk.conv3(out, residual=identity)      # relu(conv3 + bias + identity): one GEMM
    return _add_relu_(blk.conv3(out), identity)


def _resnet_features_forward(net, x, emit=None):
    """ResNet.forward_features (net_img.py:80-88) of the inference copy: bn1 -> relu -> maxpool after the raw conv1
    output (which the reference taps) in one pass over the largest activation map of the network.  ``emit(name, map)`` is
    called as soon as a stage's output exists (the engine's pipelined window starts a graph level when ITS map is ready)."""
    emit = emit or (lambda name, t: None)
    c1 = net.conv1(x)
    emit("conv1", c1)
    mp = net.maxpool
    ok = (c1.is_cuda and c1.dtype == torch.float32 and c1.is_contiguous(memory_format=torch.channels_last)
          and c1.shape[1] % 4 == 0 and mp.kernel_size == 3 and mp.stride == 2 and mp.padding == 1
          and mp.dilation == 1 and not mp.ceil_mode and hasattr(net, "_stem_affine"))
    if ok:
        B, C, H, W = c1.shape
        scale, shift = net._stem_affine
        y = torch.empty((B, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=c1.dtype, device=c1.device,
                        memory_format=torch.channels_last)
        _lib.check(_lib.lib().dagr_bn_relu_maxpool(_lib.ptr(c1), B, H, W, C, _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(y),
                                                   _lib.cur_stream(c1.device)), "bn_relu_maxpool")
    else:
        y = net.maxpool(net.relu(net.bn1(c1)))
    l1 = net.layer1(y)
    emit("layer1", l1)
    l2 = net.layer2(l1)
    emit("layer2", l2)
    l3 = net.layer3(l2)
    emit("layer3", l3)
    l4 = net.layer4(l3)
    emit("layer4", l4)
    return dict(conv1=c1, layer1=l1, layer2=l2, layer3=l3, layer4=l4)


def _basicblock_forward(blk, x):
    identity = x if blk.downsample is None else blk.downsample(x)
    out = _conv_bias_relu(blk, "conv1", x)
    return _add_relu_(blk.conv2(out), identity)


class _Level:
    """Device buffers of one pooled graph level (capacity T = gx*gy*(B+1) nodes)."""

    def __init__(self, T, cin, cout, device, cf_next=0):
        self.T = T
        self.e_cap = T * 64
        i32 = dict(dtype=torch.int32, device=device)
        f32 = dict(dtype=torch.float32, device=device)
        # pooled features + pos[:, :2]; row stride padded to 16 bytes (the tiled conv reads 16-byte channel quads)
        self.x = torch.zeros((T, (cin + 3) // 4 * 4), **f32)
        self.pos = torch.zeros((T, 3), **f32)
        self.batch = torch.zeros((T,), **i32)
        self.rowptr = torch.zeros((T + 2,), **i32)
        self.col = torch.zeros((self.e_cap,), **i32)
        self.code = torch.zeros((self.e_cap,), **i32)
        self.counts = torch.zeros((2,), **i32)       # [n_nodes, n_edges]
        self.cluster = torch.zeros((T,), **i32)      # scratch for the next pooling
        self.h1 = torch.zeros((T, cout), **f32)
        # Layer output h2 = hp[:, :cout]; with --use_image the image features sampled at this level's
        # nodes are written into hp[:, cout:] (sampling_skip, net.py:142,155,171) and hp is what gets pooled
        self.hp = torch.zeros((T, cout + cf_next), **f32)
        self.cout = cout


class WindowEngine:
    def __init__(self, model, max_events=1 << 17, device="cuda"):
        self.model = model
        self.device = torch.device(device)
        args = model.args
        bb, head = model.backbone, model.head
        self.args = args
        self.B = int(args.batch_size)
        self.W, self.H = int(model.width), int(model.height)
        self.time_window = int(getattr(args, "time_window_us", 1000000))
        self.num_classes = bb.num_classes
        self.num_scales = int(args.num_scales)
        self.use_image = bool(args.use_image)
        self.no_events = bool(getattr(args, "no_events", False))
        # channels of the image features concatenated before Layer k (net.py:47-50): [16,64,s,s,s]
        self.feat_ch = list(bb.net.feature_channels) if self.use_image else [0] * 5
        if self.use_image:   # channels-last image branch: its feature maps are then read as [B,h,w,C] rows
            bb.net.to(memory_format=torch.channels_last)
            head.cnn_head.to(memory_format=torch.channels_last)
        if bb.conv_block1.conv_block1.conv.lut_domain is None:
            model.cache_luts(width=self.W, height=self.H, radius=args.radius)
            model._engine = self
        self.L = _lib.lib()
        if self.L.dagr_device_count() < 1:
            raise RuntimeError("dagr_amd: no HIP device visible; the event-graph path has no CPU fallback")
        self._keep = []
        self._cnn_out = None
        self._img_stream = None
        self._feat_ready = None                 # pipelined window: event per feature map (keyed by the map's id)
        # detection post-processing captured with the window / tail graph (forward_detections): thresholds it was captured
        # with, its static outputs, and whether the last forward ran it
        self._post_key = None
        self._det = self._n_keep = None
        self._wg_post = self._graph_post = (None, None)
        self._post_fresh = False
        self._cnn_ready = None
        # captured --use_image windows: graph levels start when THEIR feature map exists (builder knob to A/B)
        self.pipeline_image = os.environ.get("DAGR_PIPELINE_IMAGE", "1") != "0"
        self._net_f = self._cnn_f = None
        self.fuse_convs = os.environ.get("DAGR_FUSE_CONVS", "1") != "0"
        # head scale 1's convs ride in the launches of layer5 / head scale 2 (dagr_spline_conv_fused_multi)
        self.merge_heads = os.environ.get("DAGR_MERGE_HEADS", "1") != "0"
        self._tail_jobs = None
        self._inputs_gathered = False
        self.fast_coarse_edges = os.environ.get("DAGR_FAST_COARSE_EDGES", "1") != "0"
        self.fuse_pool_accumulate = os.environ.get("DAGR_FUSE_POOL", "1") != "0"
        self._pool_accumulated = [False] * 4
        self.fuse_image_epilogues = os.environ.get("DAGR_IMG_EPILOGUES", "1") != "0"
        # Latency mode (one window batch at a time, e.g. DAGR.forward): head scale 1 runs beside pool4 / layer5 / head scale
        # 2 and everything after pool1 is replayed as one HIP graph -- the host issues one launch instead of ~70.  When
        # several engines keep the GPU full on their own streams (bench.py's throughput rigs) both cost throughput
        # (measured, events-only, 3 engines: 696 M events/s plain, 620 M with the side stream, 594 M with graph replay),
        # so such callers switch it off with set_low_latency(False).
        self.set_low_latency(os.environ.get("DAGR_LOW_LATENCY", "1") != "0")
        self._head_stream = self._head_join = self._graph = self._graph_out = None
        self._graph_warm = 0
        self._wg = self._wg_out = None       # the whole window as one captured HIP graph (latency mode)
        self._wg_warm = 0
        self._dev_mode = False               # stages bound by the device-side event / node counts (inside the capture)
        self.in_image = None
        self._async_on = False
        self._app = None
        self._n_rows = 0
        self._N = 0                  # events of the resident window (0: none yet)
        self._prepare(bb, head)
        self.max_events = 0
        self._alloc_events(int(max_events))

    # ------------------------------------------------------------------------------------ plan
    def _prepare(self, bb, head):
        dev = self.device
        L = self.L
        stream = _lib.cur_stream(dev)
        layers = [bb.conv_block1, bb.layer2, bb.layer3, bb.layer4, bb.layer5]
        pools = [bb.pool1, bb.pool2, bb.pool3, bb.pool4]
        self.dom = [layer.conv_block1.conv.lut_domain for layer in layers]
        # ---- level 0
        d0 = self.dom[0]
        if d0["rx"] != d0["ry"]:
            raise NotImplementedError("level 0 expects a square search radius (ev_tgn.py:29 derives it from the width)")
        win = []
        for r, den in ((d0["rx"], d0["den_x"]), (d0["ry"], d0["den_y"])):
            lo, cnt = ctypes.c_int32(0), ctypes.c_int32(0)
            _lib.check(L.dagr_spline_tap_window(r, den, ctypes.byref(lo), ctypes.byref(cnt)), "tap_window")
            if cnt.value <= 3:      # 3-tap window (clamped into the 5-tap kernel), else all 5 taps
                win += [min(lo.value, 2), 3]
            else:
                win += [0, 5]
        self.win0 = tuple(win)      # (win_x, tx, win_y, ty)
        self.ntaps0 = win[1] * win[3]
        ntp = (self.ntaps0 + 3) // 4 * 4
        self.ncodes0 = (2 * d0["rx"] + 1) * (2 * d0["ry"] + 1)
        self.tab0 = torch.zeros((self.ncodes0, ntp), dtype=torch.float32, device=dev)
        bad = torch.zeros((1,), dtype=torch.int32, device=dev)
        _lib.check(L.dagr_spline_l0_table(d0["rx"], d0["ry"], d0["den_x"], d0["den_y"], win[0], win[1], win[2], win[3],
                                          _lib.ptr(self.tab0), _lib.ptr(bad), stream), "l0_table")
        if int(bad.item()) != 0:
            raise RuntimeError("level-0 offset table: an offset needs a kernel tap outside the chosen window")
        l0 = layers[0]
        # Input row of level 0.  Reference channel order (net.py:118,124-125): [polarity | image feats | pos_xy].  The
        # tiled kernel (csrc/conv_l0_tiles.hip) reads a 16-channel main block as 16-byte pieces, so with --use_image the
        # row is laid out [16 image feats | polarity | pos_xy | pad] (80 B); events-only [polarity | pos_xy | pad] (16 B).
        c0 = 1 + self.feat_ch[0] + 2
        self.l0_tiles = (os.environ.get("DAGR_L0_TILES", "1") != "0" and (win[1], win[3]) in ((3, 3), (3, 5), (5, 3))
                         and int(self.args.max_neighbors) == 16 and self.feat_ch[0] in (0, 16))
        if self.l0_tiles:
            nf = self.feat_ch[0]
            self.x0_cols = list(range(1, 1 + nf)) + [0, 1 + nf, 2 + nf]      # reference channel of every x0 column
            self.x0_ld = (c0 + 3) // 4 * 4
            self.x0_feat_col, self.x0_img_col, self.x0_pos_col = nf, 0, nf + 1
        else:
            self.x0_cols = list(range(c0))
            self.x0_ld = c0
            self.x0_feat_col, self.x0_img_col, self.x0_pos_col = 0, 1, c0 - 2
        self.l0_conv1 = _pack_l0(l0.conv_block1.conv, l0.conv_block1.norm, self.win0, device=dev, cols_in=self.x0_cols)
        self.l0_conv2 = _pack_l0(l0.conv_block2.conv, l0.conv_block2.norm, self.win0,
                                 skip=(l0.conv_block2.lin, l0.conv_block2.norm_skip), device=dev,
                                 cols_skip=self.x0_cols)
        # ---- pooled levels 1..4
        self.packs = []
        for layer in layers[1:]:
            c1 = _pack_generic([layer.conv_block1.conv], [layer.conv_block1.norm], device=dev)
            c2 = _pack_generic([layer.conv_block2.conv], [layer.conv_block2.norm],
                               skip=(layer.conv_block2.lin, layer.conv_block2.norm_skip), device=dev)
            self.packs.append((c1, c2))
        # ---- poolings
        self.pool_desc, self.pool_ws, self.levels = [], [], []
        B = self.B
        chans = [l.out_channel for l in layers]
        fch = self.feat_ch
        for k, pool in enumerate(pools):
            vs = pool.voxel_size.detach().float().cpu()
            end = torch.Tensor([0.9999999, 0.9999999])
            g = ((end - 0) / vs[:2]).to(torch.int64) + 1            # grid_cluster num_voxels
            nd = self.dom[k + 1]
            remap = nd["remap"]
            desc = _lib.PoolDesc(batch_size=B, channels=chans[k] + fch[k + 1], gx=int(g[0]), gy=int(g[1]), vx=float(vs[0]),
                                 vy=float(vs[1]), inv_w=float(pool.wh_inv[0, 0]), inv_h=float(pool.wh_inv[0, 1]),
                                 two_max=_f32(2 * pool.transform.max), r00=float(remap[0, 0]),
                                 r02=float(remap[0, 2]), r11=float(remap[1, 1]), r12=float(remap[1, 2]),
                                 rx=nd["rx"], ry=nd["ry"], aggr=0 if pool.aggr == "max" else 1, append_pos=1)
            nbytes = L.dagr_pool_workspace_bytes(ctypes.byref(desc))
            if nbytes == 0:
                raise RuntimeError("libdagr_hip: " + L.dagr_last_error().decode())
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _lib.check(L.dagr_pool_workspace_init(ctypes.byref(desc), _lib.ptr(ws), nbytes, stream), "pool_ws_init")
            self.pool_desc.append(desc)
            self.pool_ws.append(ws)
            T = int(g[0]) * int(g[1]) * (B + 1)
            self.levels.append(_Level(T, chans[k] + fch[k + 1] + 2, chans[k + 1], dev,
                                      cf_next=fch[k + 2] if k + 2 < 5 else 0))
        # voxel -> first pixel tables for the level-0 pooling (same fp32 division as grid_cluster)
        vs = pools[0].voxel_size.detach().float().cpu()
        d = self.pool_desc[0]
        cellx = ((torch.arange(self.W).float() / self.W) / vs[0]).to(torch.int64)
        celly = ((torch.arange(self.H).float() / self.H) / vs[1]).to(torch.int64)
        self.xlo = torch.searchsorted(cellx, torch.arange(d.gx + 1)).to(torch.int32).to(dev)
        self.ylo = torch.searchsorted(celly, torch.arange(d.gy + 1)).to(torch.int32).to(dev)
        # ---- head
        self.head_packs, self.head_dom = [], []
        first_level = 5 - self.num_scales   # levels feeding scale 1..num_scales (3,4 or 4)
        self.head_levels = list(range(first_level, 5))
        for s in range(1, self.num_scales + 1):
            g = lambda n: getattr(head, n + str(s))
            stem = _pack_generic([g("stem").conv], [g("stem").norm], device=dev)
            cr = _pack_generic([g("cls_conv").conv, g("reg_conv").conv], [g("cls_conv").norm, g("reg_conv").norm],
                               device=dev)
            cls = _pack_generic([g("cls_pred")], [None], relu=False, device=dev)
            ro = _pack_generic([g("reg_pred"), g("obj_pred")], [None, None], relu=False, device=dev)
            self.head_packs.append((stem, cr, cls, ro))
            # DAGR.cache_luts ties the table to the head's name ("1" -> pool3, "2" -> pool4), whatever level it
            # consumes (dagr.py:52-72): the convs of head s are evaluated on THEIR domain
            hd = g("stem").conv.lut_domain
            for n in ("cls_conv", "reg_conv"):
                assert _same_domain(getattr(head, n + str(s)).conv.lut_domain, hd)
            for n in ("cls_pred", "reg_pred", "obj_pred"):
                assert _same_domain(getattr(head, n + str(s)).lut_domain, hd)
            self.head_dom.append(hd)
        self.n_reg = head.stem1.conv.out_channels
        osz = bb.get_output_sizes()[-self.num_scales:]
        self.out_sizes = osz                                   # [[H,W], ...]
        self.strides = list(bb.strides)
        self.head_vox = [pools[2].voxel_size[:2].detach().float().cpu(), pools[3].voxel_size[:2].detach().float().cpu()][-self.num_scales:]
        # scratch for the head / aggregation
        lda_max = 0
        for k, (c1, c2) in enumerate(self.packs):
            lda_max = max(lda_max, self.levels[k].T * (max(c1.K, c2.K) + 3))
        for i, lvl in enumerate(self.head_levels):
            for p in self.head_packs[i]:
                lda_max = max(lda_max, self.levels[lvl - 1].T * (p.K + 3))
        self.A = torch.zeros((lda_max,), dtype=torch.float32, device=dev)
        # second scratch for head scale 1 when it overlaps layer5 on a side stream (only the unfused convs touch it)
        lda2 = max([self.levels[self.head_levels[0] - 1].T * (p.K + 3) for p in self.head_packs[0]])
        self.A2 = torch.zeros((lda2 if self.num_scales > 1 else 1,), dtype=torch.float32, device=dev)
        self.head_buf = []
        for i, lvl in enumerate(self.head_levels):
            T = self.levels[lvl - 1].T
            Hc, Wc = self.out_sizes[i]
            self.head_buf.append(dict(
                stem=torch.zeros((T, self.n_reg), dtype=torch.float32, device=dev),
                cr=torch.zeros((T, 2 * self.n_reg), dtype=torch.float32, device=dev),
                pred=torch.zeros((T, 5 + self.num_classes), dtype=torch.float32, device=dev),
                dense=torch.zeros((self.B, 5 + self.num_classes, Hc, Wc), dtype=torch.float32, device=dev)))
        self.status = torch.zeros((4,), dtype=torch.int32, device=dev)
        self.fused_passes_max_nodes = int(os.environ.get("DAGR_FUSED_PASSES_MAX_NODES", "1600"))
        # builder knob: levels with more node slots than this take tap aggregation + GEMM as two launches
        self.fuse_max_nodes = int(os.environ.get("DAGR_FUSE_MAX_NODES", str(10 ** 9)))
        # a head whose table domain is not its input level's (num_scales = 1: head "1" on out4 with the pool3
        # table) gets its own LUT coordinates (dagr_pool_recode)
        self.head_code = []
        for i, lvl in enumerate(self.head_levels):
            same = _same_domain(self.head_dom[i], self.dom[lvl])
            self.head_code.append(None if same else torch.zeros_like(self.levels[lvl - 1].code))
        # grid / stride cache of decode_outputs (model/utils.py:119-134)
        grids, strides = [], []
        for (hs, ws_), stride in zip(self.out_sizes, self.strides):
            yv, xv = torch.meshgrid(torch.arange(hs), torch.arange(ws_), indexing="ij")
            grid = torch.stack((xv, yv), 2).view(1, -1, 2)
            grids.append(grid)
            strides.append(torch.full((1, grid.shape[1], 1), stride))
        self.grid_cache = torch.cat(grids, dim=1).float().to(dev)
        self.stride_cache = torch.cat(strides, dim=1).float().to(dev)

    def set_low_latency(self, on):
        self.overlap_heads = bool(on) and os.environ.get("DAGR_OVERLAP_HEADS", "1") != "0"
        self.tail_graph = bool(on) and os.environ.get("DAGR_TAIL_GRAPH", "1") != "0"
        # the WHOLE window (image branch, graph build, level 0, tail, heads, decode) as one HIP graph: every launch is sized
        # for the engine's event capacity and bounded by counts that live in device memory
        self.window_graph = bool(on) and os.environ.get("DAGR_WINDOW_GRAPH", "1") != "0"
        return self

    def _alloc_events(self, n):
        if n <= self.max_events:
            return
        dev = self.device
        n = max(n, 1024)
        self.max_events = n
        bb = self.model.backbone
        tgn = bb.events_to_graph
        self.graph = tgn.window_builder(self.W, self.H, self.time_window, self.B, dev, max_events=n)
        K = self.graph.K
        self.nbr_src = torch.zeros((n, K), dtype=torch.int32, device=dev)
        self.nbr_code = torch.zeros((n, K), dtype=torch.int16, device=dev)
        self.deg = torch.zeros((n,), dtype=torch.int32, device=dev)
        self.h1 = torch.zeros((n, 16), dtype=torch.float32, device=dev)
        self.hp0 = torch.zeros((n, 16 + self.feat_ch[1]), dtype=torch.float32, device=dev)  # [h2 | image feats]
        self.x0buf = torch.zeros((n, self.x0_ld), dtype=torch.float32, device=dev)
        self.cluster0 = torch.zeros((n,), dtype=torch.int32, device=dev)
        self.pos_n = torch.zeros((n, 3), dtype=torch.float32, device=dev)     # node (slot) order
        self.batch_n = torch.zeros((n,), dtype=torch.int32, device=dev)
        # static inputs of the captured window (latency mode): the caller's events are staged here by one launch
        self.in_pos = torch.zeros((n, 3), dtype=torch.float32, device=dev)
        self.in_feat = torch.zeros((n,), dtype=torch.float32, device=dev)
        self.in_batch = torch.zeros((n,), dtype=torch.int32, device=dev)
        self.n_dev = torch.zeros((1,), dtype=torch.int32, device=dev)
        self._wg = self._wg_out = None       # captured on the old buffers
        self._wg_warm = 0
        self.rows_cap = n
        self._async_on = False
        self._app = None

    # ---------------------------------------------------------------------- asynchronous operation
    _ROW_ARRAYS = ("nbr_src", "nbr_code", "deg", "h1", "hp0", "x0buf", "cluster0", "pos_n", "batch_n")

    def _grow_rows(self, rows):
        """More level-0 rows WITHOUT losing the resident window (the asynchronous state lives in these arrays)."""
        if rows <= self.rows_cap:
            return
        cap = max(rows, self.rows_cap + self.rows_cap // 2)
        for name in self._ROW_ARRAYS:
            old = getattr(self, name)
            new = torch.zeros((cap,) + tuple(old.shape[1:]), dtype=old.dtype, device=old.device)
            new[:old.shape[0]] = old
            setattr(self, name, new)
        if self._app is not None:
            for name in ("next", "xytb", "batch_ev"):
                old = self._app[name]
                new = torch.zeros((cap,) + tuple(old.shape[1:]), dtype=old.dtype, device=old.device)
                new[:old.shape[0]] = old
                self._app[name] = new
        self.rows_cap = cap

    def can_append(self):
        """Whether ``forward_append`` can attach events to the resident window: the incremental path runs on the tiled
        level-0 conv and keeps pool1's coarse edges as cell bitmaps (search radius <= two voxels, dagr_pool_l0_stream),
        and it needs a window of THIS engine to attach to.  ``DAGR.forward(reset=False)`` re-evaluates the running
        window otherwise (``make_model_synchronous``'s path)."""
        if not self.l0_tiles or self.no_events or self._N <= 0 or self.B >= 30:
            return False
        d, g = self.pool_desc[0], self.graph.params
        cell_w, cell_h = int(np.floor(np.float32(d.vx) * np.float32(g["width"]))), int(np.floor(np.float32(d.vy) * np.float32(g["height"])))
        return g["max_neighbors"] <= 16 and g["radius"] <= 2 * min(cell_w, cell_h) and g["width"] <= 4096

    def async_begin(self):
        """Turn the resident window (the last ``forward_raw``) into the state of an asynchronous run: per-pixel chains for
        the events to come (empty), the sample index by event id, and pool1's accumulators resident in their own
        workspace (dagr_pool_l0_stream).  Called by the first ``forward_append`` after a window."""
        L, P = self.L, _lib.ptr
        dev = self.device
        n0 = self._N
        if n0 == 0:
            raise RuntimeError("asynchronous update without a window: call forward_raw (reset=True) first")
        self._grow_rows(n0 + max(4096, n0 // 4))
        npix = self.B * self.H * self.W
        if self._app is None or self._app["head"].shape[0] != npix:
            nbytes = L.dagr_pool_workspace_bytes(ctypes.byref(self.pool_desc[0]))
            self._app = dict(head=torch.empty((npix,), dtype=torch.int32, device=dev),
                             next=torch.zeros((self.rows_cap,), dtype=torch.int32, device=dev),
                             xytb=torch.zeros((self.rows_cap, 4), dtype=torch.int32, device=dev),
                             batch_ev=torch.zeros((self.rows_cap,), dtype=torch.int32, device=dev),
                             status=torch.zeros((4,), dtype=torch.int32, device=dev),
                             pool_ws=torch.empty(nbytes, dtype=torch.uint8, device=dev))
        a = self._app
        for name in ("next", "xytb", "batch_ev"):
            if a[name].shape[0] < self.rows_cap:
                a[name] = torch.zeros((self.rows_cap,) + tuple(a[name].shape[1:]), dtype=a[name].dtype, device=dev)
        a.pop("args", None)                  # the argument block of dagr_async_update is rebuilt for this window
        a["head"].fill_(-1)
        a["status"].zero_()
        a["batch_ev"][:n0] = self._batch.to(torch.int32)
        self._n_rows = n0
        self._async_on = True
        self._pool1_stream(rebuild=True, first=n0, n=0)

    def _pool1_stream(self, rebuild, first, n):
        L, P = self.L, _lib.ptr
        g, a, l1, d = self.graph, self._app, self.levels[0], self.pool_desc[0]
        _lib.check(L.dagr_pool_l0_stream(ctypes.byref(d), P(a["pool_ws"]), 1 if rebuild else 0, ctypes.byref(g.desc),
                                         P(g.workspace), P(self.xlo), P(self.ylo), P(self.hp0), self.hp0.shape[1],
                                         P(self.pos_n), P(a["batch_ev"]), self._N, first, n, P(self.nbr_src),
                                         P(self.nbr_code), P(self.deg), P(l1.x), l1.x.shape[1], 0, P(l1.pos), P(l1.batch),
                                         P(l1.counts), P(l1.rowptr), P(l1.col), P(l1.code),
                                         ctypes.c_void_p(l1.counts.data_ptr() + 4), l1.e_cap,
                                         _lib.cur_stream(self.device)), "pool_l0_stream")

    def _conv_l0_rows(self, pack, first, n, x, ldx, xskip, ldskip, out, ldo):
        L, P = self.L, _lib.ptr
        cin, cskip, w, s = pack
        d0 = self.dom[0]
        wx, tx, wy, ty = self.win0
        cm = 16 if cin >= 16 else 0
        _lib.check(L.dagr_spline_conv_l0_tiles_rows(cm, cin - cm, cskip, wx, tx, wy, ty, d0["rx"], d0["ry"], d0["den_x"],
                                                    d0["den_y"], first, n, self.graph.K, P(self.nbr_src), P(self.nbr_code),
                                                    P(self.deg), x, ldx, xskip, ldskip, P(w), P(s), 1, out, ldo, None,
                                                    _lib.cur_stream(self.device)), "conv_l0_tiles_rows")

    def _async_call(self, a, first, n, pos, feat, batch, stream):
        """Fill / refresh the argument block of ``dagr_async_update`` and issue the update."""
        P = lambda t: None if t is None else t.data_ptr()
        u = a.get("args")
        g, l1, d0 = self.graph, self.levels[0], self.dom[0]
        if u is None or a.get("args_rows_cap") != self.rows_cap:
            u = _lib.AsyncUpdateArgs()
            u.gdesc, u.graph_ws = ctypes.pointer(g.desc), P(g.workspace)
            u.app_head, u.app_next, u.app_xytb = P(a["head"]), P(a["next"]), P(a["xytb"])
            u.capacity = a["next"].shape[0]
            u.nbr_src, u.nbr_code, u.deg, u.status = P(self.nbr_src), P(self.nbr_code), P(self.deg), P(a["status"])
            u.pos_nodes, u.batch_nodes, u.batch_events = P(self.pos_n), P(self.batch_n), P(a["batch_ev"])
            u.x0, u.ldx0, u.col_feat, u.col_pos = P(self.x0buf), self.x0_ld, self.x0_feat_col, self.x0_pos_col
            u.win_x, u.tx, u.win_y, u.ty = self.win0
            u.rx, u.ry, u.den_x, u.den_y = d0["rx"], d0["ry"], d0["den_x"], d0["den_y"]
            cin1, _, w1, s1 = self.l0_conv1
            _, _, w2, s2 = self.l0_conv2
            u.cin1, u.w1, u.s1, u.h1, u.ldh1 = cin1, P(w1), P(s1), P(self.h1), 16
            u.w2, u.s2, u.hp0, u.ldhp0 = P(w2), P(s2), P(self.hp0), self.hp0.shape[1]
            u.pdesc, u.pool_ws, u.xlo, u.ylo = ctypes.pointer(self.pool_desc[0]), P(a["pool_ws"]), P(self.xlo), P(self.ylo)
            u.x_out, u.ldo, u.pos_out, u.batch_out = P(l1.x), l1.x.shape[1], P(l1.pos), P(l1.batch)
            u.n_out, u.rowptr_out, u.col_out, u.code_out = P(l1.counts), P(l1.rowptr), P(l1.col), P(l1.code)
            u.e_out, u.e_cap = l1.counts.data_ptr() + 4, l1.e_cap
            a["args"], a["args_rows_cap"] = u, self.rows_cap
        u.n_static, u.first_id, u.n_new = self._N, first, n
        u.pos, u.feat, u.batch = P(pos), P(feat), P(batch)
        u.batch_is_int64 = 1 if (batch is not None and batch.dtype == torch.int64) else 0
        _lib.check(self.L.dagr_async_update(ctypes.byref(u), stream), "async_update")

    def forward_append(self, pos, feat, batch, static_out=False):
        """``reset=False``: the n events of a micro-batch attach to the resident window (EV_TGN.forward, ev_tgn.py:45-56).
        Edges point from older to newer events, so the window's level-0 rows stand; the update
          1. links the events into their pixels' chains and searches their in-edges (dagr_async_graph_append),
          2. computes conv_block1 on the n new rows only (dagr_spline_conv_l0_tiles_rows),
          3. adds the rows to pool1's resident accumulators and re-emits level 1 (dagr_pool_l0_stream),
          4. runs the fixed-size part (levels 1-4, heads, decode) as a window does.
        Output = what ``forward_raw`` gives on all events so far, bit for bit (same kernels per row; order-free
        accumulators).  pos fp32[n,3] normalised as format_data does, feat fp32[n,1], batch int32/int64[n]."""
        if not self.l0_tiles:
            raise NotImplementedError("asynchronous updates run on the tiled level-0 conv (16 neighbours, 3x3/3x5 tap window)")
        L, P = self.L, _lib.ptr
        if not self._async_on:
            self.async_begin()
        n = int(pos.shape[0])
        a = self._app
        first = self._n_rows
        stream = _lib.cur_stream(self.device)
        if n:
            self._grow_rows(first + n)
            a = self._app
            pos = pos.float().contiguous()
            feat = feat.float().reshape(-1).contiguous()
            batch = batch.contiguous()
        if not self.use_image:
            # events-only: graph append + both level-0 convs on the new rows + pool1's resident accumulators as ONE native
            # call (dagr_async_update): same kernels, no host time between their launches
            self._async_call(a, first, n, pos if n else None, feat if n else None, batch if n else None, stream)
            self._n_rows = first + n
        else:
            if n:
                b64 = 1 if batch.dtype == torch.int64 else 0
                g = self.graph
                _lib.check(L.dagr_async_graph_append(ctypes.byref(g.desc), P(g.workspace), self._N, first, P(a["head"]),
                                                     P(a["next"]), P(a["xytb"]), a["next"].shape[0], P(pos), 0, P(batch), b64,
                                                     n, P(self.nbr_src), P(self.nbr_code), P(self.deg), P(a["status"]), P(feat),
                                                     P(self.pos_n), P(self.batch_n), P(a["batch_ev"]), P(self.x0buf),
                                                     self.x0_ld, self.x0_feat_col, self.x0_pos_col, stream),
                           "async_graph_append")
                rows = slice(first, first + n)
                self._sample(None, n, self.pos_n[rows], self.batch_n[rows], 0, self._img_feats[0], self.x0buf[rows],
                             self.x0_img_col)
                self._conv_l0_rows(self.l0_conv1, first, n, P(self.x0buf), self.x0_ld, None, 0, P(self.h1), 16)
                self._conv_l0_rows(self.l0_conv2, first, n, P(self.h1), 16, P(self.x0buf), self.x0_ld, P(self.hp0),
                                   self.hp0.shape[1])
                self._sample(None, n, self.pos_n[rows], self.batch_n[rows], 0, self._img_feats[1], self.hp0[rows], 16)
                self._n_rows = first + n
            self._pool1_stream(rebuild=False, first=first, n=n)
        if self.tail_graph and not self.use_image:
            return self._replay_tail(static_out)
        return self._tail_and_head()

    # ------------------------------------------------------------------------------- kernels
    def _conv_generic(self, lvl, pack, x, ldx, xskip, ldskip, out, ldo, dom, stream, code=None, scratch=None):
        L = self.L
        P = _lib.ptr
        code = lvl.code if code is None else code
        scratch = self.A if scratch is None else scratch
        passes = L.dagr_spline_conv_fused_passes(pack.cin, pack.cskip)
        if self.fuse_convs and lvl.T <= self.fuse_max_nodes and \
                (passes == 1 or (passes > 1 and lvl.T <= self.fused_passes_max_nodes)):
            # tap aggregation + contraction in one launch (A tile lives in LDS; rows wider than the tile in passes over
            # the edges, which pays on small levels only: tools/microbench/head_ab.hip)
            _lib.check(L.dagr_spline_conv_fused(P(lvl.counts), lvl.T, P(lvl.rowptr), P(lvl.col), P(code), x, ldx,
                                                pack.cin, xskip, ldskip, pack.cskip, dom["rx"], dom["ry"], dom["den_x"],
                                                dom["den_y"], P(pack.Wq), P(pack.bias), out, ldo, pack.N,
                                                1 if pack.relu else 0, stream), "spline_conv_fused")
            return
        lda = (pack.K + 3) // 4 * 4    # 16-byte aligned rows for the MFMA GEMM's float4 loads
        _lib.check(L.dagr_spline_tap_aggregate(P(lvl.counts), lvl.T, P(lvl.rowptr), P(lvl.col), P(code),
                                               x, ldx, pack.cin, xskip, ldskip, pack.cskip, dom["rx"], dom["ry"],
                                               dom["den_x"], dom["den_y"], P(scratch), lda, stream), "tap_aggregate")
        _lib.check(L.dagr_gemm_bias_act(P(lvl.counts), lvl.T, P(scratch), lda, P(pack.Wm), pack.ldw, P(pack.bias),
                                        out, ldo, pack.K, pack.N, 1 if pack.relu else 0, stream), "gemm")

    # -------------------------------------------------------------------------------- stages
    def stage_graph(self, pos, batch, feat=None):
        """events -> neighbour lists (EV_TGN.forward, layers/ev_tgn.py:39-58).  With ``feat`` (the events' features) the
        build's last launch also writes the node-ordered level-0 inputs (``stage_l0_input`` then only samples the image
        features): one launch less per window."""
        N = int(pos.shape[0])
        self._alloc_events(N)
        self._N = N
        self._pos, self._batch = pos, batch
        self._nbr = (self.nbr_src[:N], self.nbr_code[:N], self.deg[:N])
        inputs = None
        if feat is not None and pos.dtype == torch.float32:
            self._feat_flat = feat.float().reshape(-1).contiguous()      # (kept alive until the launch has run)
            inputs = _lib.L0Inputs(feat=self._feat_flat.data_ptr(), pos_nodes=self.pos_n.data_ptr(),
                                   batch_nodes=self.batch_n.data_ptr(), x0=self.x0buf.data_ptr(), ldx0=self.x0_ld,
                                   col_feat=self.x0_feat_col, col_pos=self.x0_pos_col)
        self.graph.build(pos, batch, out=self._nbr, n_dev=self.n_dev if self._dev_mode else None, inputs=inputs)
        self._inputs_gathered = inputs is not None
        self._n_rows = N              # a new window: the asynchronous state of the previous one is gone
        self._async_on = False

    def _nptr(self):
        """``n_ptr`` of the level-0 kernels: the builder's device-side node count inside a captured window, else NULL (the
        host passes the exact count)."""
        return self.graph.node_count_ptr() if self._dev_mode else None

    def _sample(self, n_ptr, n_max, pos, batch, b64, fmap, out, coff):
        """sample_features (net.py:193-221) of one channels-last feature map into out[:, coff:coff+C]."""
        if self._feat_ready is not None:            # pipelined window: the map comes from the image branch's stream
            ready = self._feat_ready.get(id(fmap))
            if ready is None:
                raise RuntimeError("pipelined window: a sampled feature map has no ready-event (the image branch did not "
                                   "announce it) -- reading it would race with the image stream")
            torch.cuda.current_stream(self.device).wait_event(ready)
        Bf, C, h, w = fmap.shape
        nhwc = fmap.permute(0, 2, 3, 1)
        if not nhwc.is_contiguous():
            nhwc = nhwc.contiguous()
        self._keep.append(nhwc)
        _lib.check(self.L.dagr_sample_features(n_ptr, n_max, _lib.ptr(pos), _lib.ptr(batch), b64, _lib.ptr(nhwc), Bf, h,
                                               w, C, self.W, self.H, _lib.ptr(out), out.shape[1], coff,
                                               _lib.cur_stream(self.device)), "sample_features")

    def _fold_image_branch(self):
        """Inference copy of the image branch with every Conv2d+BatchNorm2d(eval) pair folded into one
        conv (torch.nn.utils.fusion.fuse_conv_bn_eval): the BN passes over the 640x480 activations are
        pure HBM traffic.  The stem conv1 stays unfolded: the reference taps its raw output
        (feature_layers=["conv1", ...], net.py:47).  Parameters of the original modules are untouched."""
        import copy
        from torch.nn.utils.fusion import fuse_conv_bn_eval
        bb, head = self.model.backbone, self.model.head
        net = copy.deepcopy(bb.net).eval()
        cnn = copy.deepcopy(head.cnn_head).eval()

        def fold_block(blk):
            for ci, bi in (("conv1", "bn1"), ("conv2", "bn2"), ("conv3", "bn3")):
                if hasattr(blk, ci):
                    setattr(blk, ci, fuse_conv_bn_eval(getattr(blk, ci), getattr(blk, bi)))
                    setattr(blk, bi, torch.nn.Identity())
            if blk.downsample is not None:
                blk.downsample = torch.nn.Sequential(fuse_conv_bn_eval(blk.downsample[0], blk.downsample[1]))
        for name in ("layer1", "layer2", "layer3", "layer4"):
            for blk in getattr(net.module, name):
                fold_block(blk)
        for m in cnn.modules():
            if hasattr(m, "conv") and hasattr(m, "bn") and isinstance(m.bn, torch.nn.BatchNorm2d):
                m.conv = fuse_conv_bn_eval(m.conv, m.bn)
                m.bn = torch.nn.Identity()
                if self.fuse_image_epilogues and isinstance(getattr(m, "act", None), torch.nn.SiLU) \
                        and m.conv.bias is not None:
                    m._bias = m.conv.bias.detach().clone().contiguous()
                    m.conv.bias = None
                    m.forward = types.MethodType(_baseconv_forward, m)
        net = net.to(memory_format=torch.channels_last)
        cnn = cnn.to(memory_format=torch.channels_last)
        for blkname in ("layer1", "layer2", "layer3", "layer4"):
            _gemmify_1x1(getattr(net.module, blkname))
            for blk in getattr(net.module, blkname):
                relu_convs = ("conv1", "conv2") if hasattr(blk, "conv3") else ("conv1",)
                if self.fuse_image_epilogues:
                    for cname in relu_convs:
                        conv = getattr(blk, cname)
                        if isinstance(conv, _Conv1x1Gemm):
                            conv.relu = True                      # ReLU in the hipBLASLt epilogue
                        elif isinstance(conv, torch.nn.Conv2d) and conv.bias is not None:
                            setattr(blk, "_" + cname + "_bias", conv.bias.detach().clone().contiguous())
                            conv.bias = None                      # bias + ReLU in one pass after the conv
                blk.forward = types.MethodType(_bottleneck_forward if hasattr(blk, "conv3") else _basicblock_forward,
                                               blk)
        if self.fuse_image_epilogues:
            bn = net.module.bn1
            scale = (bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)).contiguous()
            shift = (bn.bias.detach().float() - bn.running_mean.detach().float() * scale).contiguous()
            net.module._stem_affine = (scale, shift)
            net.module.forward_features = types.MethodType(_resnet_features_forward, net.module)
        _gemmify_1x1(net.feature_dconv)
        _gemmify_1x1(net.output_dconv)
        self._net_f, self._cnn_f = net, cnn

    def _image_branch(self, image, on_feature=None):
        """``on_feature(j, map)``: called when feature map j (feature_layers[j] through its 1x1 dconv) has been issued --
        HookModule.forward (net_img.py:137-145) applies the dconvs after the whole trunk; here each one follows its layer."""
        if self._net_f is None:
            self._fold_image_branch()
        bb, head = types.SimpleNamespace(net=self._net_f), types.SimpleNamespace(cnn_head=self._cnn_f)
        x = image.contiguous(memory_format=torch.channels_last)
        net = self._net_f
        if on_feature is not None and hasattr(net.module, "_stem_affine"):
            feats = [None] * len(net.feature_layers)

            def emit(name, t):
                if name in net.feature_layers:
                    j = net.feature_layers.index(name)
                    feats[j] = net.feature_dconv[j](t) if len(net.feature_dconv) > 0 else t
                    on_feature(j, feats[j])
            d = _resnet_features_forward(net.module, x, emit)
            outs = [d[l] for l in net.output_layers]
            if len(net.output_dconv) > 0:
                outs = [dconv(o) for o, dconv in zip(outs, net.output_dconv)]
        else:
            feats, outs = bb.net(x)
            if on_feature is not None:              # unfolded trunk (DAGR_IMG_EPILOGUES=0): all maps exist only now
                for j, f in enumerate(feats):
                    on_feature(j, f)
        outs = outs[-self.num_scales:]
        resized = [torch.nn.functional.interpolate(f, o) for f, o in zip(outs, self.out_sizes)]
        return feats, head.cnn_head(resized)

    def stage_image(self, image):
        """HookModule + CNNHead on PyTorch-ROCm (net.py:110, dagr.py:205-206)."""
        self._keep = []
        self._img_feats, self._cnn_out = self._image_branch(image)

    def image_async(self, image, stream=None):
        """Start the image branch of a *future* window batch on a side stream and return a handle for
        ``forward_raw(..., image_handle=h)``.  Windows are independent, and a frame is available before
        the 50 ms of events that follow it, so the dense CNN of batch i+1 can share the GPU with the
        gather-bound event path of batch i."""
        cur = torch.cuda.current_stream(self.device)
        if stream is not None:
            self._img_stream = stream
        if self._img_stream is None:
            self._img_stream = torch.cuda.Stream(self.device)
        s = self._img_stream
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            feats, cnn_out = self._image_branch(image)
            ev = torch.cuda.Event()
            ev.record(s)
        return feats, cnn_out, ev

    def _consume_image_handle(self, handle):
        feats, cnn_out, ev = handle
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        for t in list(feats) + [o for v in cnn_out.values() for o in v]:
            t.record_stream(cur)   # allocated on the side stream, read on this one
        self._keep = []
        self._img_feats, self._cnn_out = feats, cnn_out

    def stage_l0_input(self, feat):
        """Level-0 inputs in node order: x = cat(x, [image feats,] pos[:, :2]) (net.py:118,124-125)."""
        N = self._N
        x0 = self.x0buf[:N]
        g = self.graph
        if not self._inputs_gathered:
            f = feat.float().reshape(-1).contiguous()
            _lib.check(self.L.dagr_graph_gather_inputs(ctypes.byref(g.desc), _lib.ptr(g.workspace), _lib.ptr(self._pos),
                                                       _lib.ptr(f), N, _lib.ptr(self.pos_n), _lib.ptr(self.batch_n),
                                                       _lib.ptr(x0), self.x0_ld, self.x0_feat_col, self.x0_pos_col,
                                                       _lib.cur_stream(self.device)), "graph_gather_inputs")
        if self.use_image:
            self._sample(self._nptr(), N, self.pos_n, self.batch_n, 0, self._img_feats[0], x0, self.x0_img_col)
        self._x0 = x0

    def _conv_l0(self, pack, x, ldx, xskip, ldskip, out, ldo):
        L, P = self.L, _lib.ptr
        nbr_src, nbr_code, deg = self._nbr
        cin, cskip, w, s = pack
        d0 = self.dom[0]
        wx, tx, wy, ty = self.win0
        stream = _lib.cur_stream(self.device)
        if self.l0_tiles:
            cm = 16 if cin >= 16 else 0
            _lib.check(L.dagr_spline_conv_l0_tiles(cm, cin - cm, cskip, wx, tx, wy, ty, d0["rx"], d0["ry"], d0["den_x"],
                                                   d0["den_y"], self._N, self.graph.K, P(nbr_src), P(nbr_code), P(deg),
                                                   x, ldx, xskip, ldskip, P(w), P(s), 1, out, ldo, self._nptr(), stream),
                       "conv_l0_tiles")
        else:
            _lib.check(L.dagr_spline_conv_l0(cin, cskip, self.ntaps0, self._N, self.graph.K, self.ncodes0, P(nbr_src),
                                             P(nbr_code), P(deg), x, ldx, xskip, ldskip, P(self.tab0), P(w), P(s), 1,
                                             out, ldo, stream), "conv_l0")

    def stage_l0_conv1(self):
        """conv_block1.conv_block1: SplineConv(3|19 -> 16)+BN+ReLU (conv.py:23-28)."""
        P = _lib.ptr
        assert self.l0_conv1[0] == len(self.x0_cols)
        self._conv_l0(self.l0_conv1, P(self._x0), self.x0_ld, None, 0, P(self.h1), 16)

    def stage_l0_conv2(self, sample=True):
        """conv_block1.conv_block2: SplineConv(16->16)+BN + skip Linear+BN, ReLU (conv.py:47-56);
        with --use_image followed by sampling_skip(image_feat[1]) (net.py:129)."""
        P = _lib.ptr
        self._conv_l0(self.l0_conv2, P(self.h1), 16, P(self._x0), self.x0_ld, P(self.hp0), self.hp0.shape[1])
        if self.use_image and sample:
            self._sample(self._nptr(), self._N, self.pos_n, self.batch_n, 0, self._img_feats[1], self.hp0[:self._N], 16)

    def stage_l0_sample1(self):
        """sampling_skip(image_feat[1]) (net.py:129) alone: the level-0 nodes' features of the second map into hp0[:, 16:]."""
        self._sample(None, self._N, self.pos_n, self.batch_n, 0, self._img_feats[1], self.hp0[:self._N], 16)

    def stage_pool1(self):
        """pool1 (net.py:131) on the event graph."""
        L, P = self.L, _lib.ptr
        g = self.graph
        nbr_src, nbr_code, deg = self._nbr
        l1 = self.levels[0]
        d = self.pool_desc[0]
        b64 = 1 if self._batch.dtype == torch.int64 else 0
        _lib.check(L.dagr_pool_l0(ctypes.byref(d), P(self.pool_ws[0]), ctypes.byref(g.desc), P(g.workspace),
                                  P(self.xlo), P(self.ylo), P(self.hp0), self.hp0.shape[1], P(self.pos_n),
                                  P(self.batch_n), P(self._batch), b64, self._N, P(nbr_src),
                                  P(nbr_code) if self.fast_coarse_edges else None, P(deg), P(self.cluster0), P(l1.x),
                                  l1.x.shape[1], 0, P(l1.pos), P(l1.batch), P(l1.counts), P(l1.rowptr), P(l1.col),
                                  P(l1.code), ctypes.c_void_p(l1.counts.data_ptr() + 4), l1.e_cap,
                                  _lib.cur_stream(self.device)), "pool_l0")

    def pool1_accumulate_again(self):
        """pool1's accumulation kernel alone on the resident window (measurement: bench.py); follow with ``stage_pool1``."""
        L, P = self.L, _lib.ptr
        g = self.graph
        nbr_src, nbr_code, deg = self._nbr
        _lib.check(L.dagr_pool_l0_accumulate(ctypes.byref(self.pool_desc[0]), P(self.pool_ws[0]), ctypes.byref(g.desc),
                                             P(g.workspace), P(self.xlo), P(self.ylo), P(self.hp0), self.hp0.shape[1],
                                             P(self.pos_n), self._N, P(nbr_src),
                                             P(nbr_code) if self.fast_coarse_edges else None, P(deg),
                                             _lib.cur_stream(self.device)), "pool_l0_accumulate")

    def _stage_level(self, k, trace=None):
        """Layer k+2 on pooled level k+1 (conv pair), then -- for k < 3 -- [sampling_skip] + pool k+2 (net.py:137-184)."""
        L, P = self.L, _lib.ptr
        stream = _lib.cur_stream(self.device)
        lvl = self.levels[k]
        c1, c2 = self.packs[k]
        dom = self.dom[k + 1]
        ldx = lvl.x.shape[1]
        self._conv_generic(lvl, c1, P(lvl.x), ldx, None, 0, P(lvl.h1), c1.N, dom, stream)
        ldh = lvl.hp.shape[1]
        # events-only levels that are pooled next: launch (A) of that pooling (merge of every node into its cluster's
        # accumulators, coarse-edge sets) rides in this conv's epilogue -- one launch less per level
        fuse_pool = (k < 3 and not self.use_image and self.fuse_convs and self.fuse_pool_accumulate
                     and L.dagr_spline_conv_fused_passes(c2.cin, c2.cskip) == 1
                     and self.pool_desc[k + 1].channels == c2.N)
        self._pool_accumulated[k] = fuse_pool
        if fuse_pool:
            d = self.pool_desc[k + 1]
            _lib.check(L.dagr_spline_conv_fused_pool(P(lvl.counts), lvl.T, P(lvl.rowptr), P(lvl.col), P(lvl.code), P(lvl.h1),
                                                     c1.N, c2.cin, P(lvl.x), ldx, c2.cskip, dom["rx"], dom["ry"],
                                                     dom["den_x"], dom["den_y"], P(c2.Wq), P(c2.bias), P(lvl.hp), ldh, c2.N,
                                                     1 if c2.relu else 0, ctypes.byref(d), P(self.pool_ws[k + 1]), P(lvl.pos),
                                                     P(lvl.batch), P(lvl.cluster), stream), "spline_conv_fused_pool")
        else:
            self._conv_generic(lvl, c2, P(lvl.h1), c1.N, P(lvl.x), ldx, P(lvl.hp), ldh, dom, stream)
        if trace is not None:
            trace[f"pool{k + 1}"] = self._level_snapshot(lvl, lvl.x)
            trace[f"layer{k + 2}"] = self._level_snapshot(lvl, lvl.hp[:, :lvl.cout])

    def _stage_pool(self, k):
        """[sampling_skip(image_feat[k+2])] + pool k+2: level k+1 -> level k+2 (net.py:142-146,155-159,171-175)."""
        L, P = self.L, _lib.ptr
        stream = _lib.cur_stream(self.device)
        lvl, nxt = self.levels[k], self.levels[k + 1]
        ldh = lvl.hp.shape[1]
        if self.use_image:
            self._sample(P(lvl.counts), lvl.T, lvl.pos, lvl.batch, 0, self._img_feats[k + 2], lvl.hp, lvl.cout)
        d = self.pool_desc[k + 1]
        # (n_max = 0: the accumulation already happened in the conv's epilogue -- scan + emit only)
        _lib.check(L.dagr_pool_csr(ctypes.byref(d), P(self.pool_ws[k + 1]), P(lvl.counts),
                                   0 if self._pool_accumulated[k] else lvl.T, P(lvl.hp),
                                   ldh, P(lvl.pos), P(lvl.batch), P(lvl.rowptr), P(lvl.col),
                                   P(lvl.cluster), P(nxt.x), nxt.x.shape[1], 0, P(nxt.pos), P(nxt.batch),
                                   P(nxt.counts), P(nxt.rowptr), P(nxt.col), P(nxt.code),
                                   ctypes.c_void_p(nxt.counts.data_ptr() + 4), nxt.e_cap, stream), "pool_csr")

    def stage_tail(self, trace=None):
        """layer2..layer5 with pool2..pool4 (net.py:137-184)."""
        for k in range(4):
            self._stage_level(k, trace)
            if k < 3:
                self._stage_pool(k)

    def _head_recode(self, i):
        """Edge codes of head scale i's graph in the head convs' own offset domain, when that differs from the level's
        (num_scales = 1: the head's table covers pool3's domain, dagr.py:52-62)."""
        code = self.head_code[i]
        if code is None:
            return
        L, P = self.L, _lib.ptr
        lvln = self.head_levels[i]
        lvl = self.levels[lvln - 1]
        dom = self.head_dom[i]
        rm = dom["remap"]
        _lib.check(L.dagr_pool_recode(P(lvl.counts), lvl.T, P(lvl.rowptr), P(lvl.col), P(lvl.pos),
                                      self.pool_desc[lvln - 1].two_max, float(rm[0, 0]), float(rm[0, 2]),
                                      float(rm[1, 1]), float(rm[1, 2]), dom["rx"], dom["ry"], P(code),
                                      lvl.e_cap, ctypes.c_void_p(self.status.data_ptr() + 4), _lib.cur_stream(self.device)),
                   "pool_recode")

    def _conv_job(self, lvl, pack, x, ldx, xskip, ldskip, out, ldo, dom, code=None):
        """One entry of a multi-job launch: the arguments ``_conv_generic`` would pass (device addresses as ints)."""
        code = lvl.code if code is None else code
        return _lib.ConvJob(n_nodes_ptr=lvl.counts.data_ptr(), n_nodes_max=lvl.T, rowptr=lvl.rowptr.data_ptr(),
                            col=lvl.col.data_ptr(), code=code.data_ptr(), x=x, ldx=ldx, cin=pack.cin, xskip=xskip,
                            ldskip=ldskip, cskip=pack.cskip, rx=dom["rx"], ry=dom["ry"], den_x=dom["den_x"],
                            den_y=dom["den_y"], Wq=pack.Wq.data_ptr(), bias=pack.bias.data_ptr(), C=out, ldc=ldo, N=pack.N,
                            relu=1 if pack.relu else 0)

    def _build_tail_jobs(self):
        """Launch plan of layer5 + both head scales when both exist (net.py:166-186, dagr.py:179-236): head scale 1 needs
        level 3 only, so its convs share the launches of pool4's consumers --
            [layer5.conv1 | stem_1]  [layer5.conv2 | cls_conv,reg_conv_1]  [stem_2 | reg,obj_pred_1 | cls_pred_1]
            [cls_conv,reg_conv_2]  [reg,obj_pred_2 | cls_pred_2]
        five launches where the stream fork ran eight (three of them beside the others, + a fork and a join).  None when
        a conv of the plan needs the pass form (dagr-m / dagr-l heads) or a single scale exists."""
        L = self.L
        if not (self.fuse_convs and self.merge_heads and len(self.head_levels) == 2 and self.head_levels[0] == 3):
            return None
        c1, c2 = self.packs[3]
        packs = [c1, c2] + [p for hp in self.head_packs for p in hp]
        if any(L.dagr_spline_conv_fused_passes(p.cin, p.cskip) != 1 for p in packs):
            return None
        nr = self.n_reg
        lvl4 = self.levels[3]
        dom4 = self.dom[4]
        ldx4, ldh4 = lvl4.x.shape[1], lvl4.hp.shape[1]
        layer5 = [self._conv_job(lvl4, c1, lvl4.x.data_ptr(), ldx4, None, 0, lvl4.h1.data_ptr(), c1.N, dom4),
                  self._conv_job(lvl4, c2, lvl4.h1.data_ptr(), c1.N, lvl4.x.data_ptr(), ldx4, lvl4.hp.data_ptr(), ldh4, dom4)]
        heads = []
        for i in range(2):
            lvl = self.levels[self.head_levels[i] - 1]
            stem, cr, cls, ro = self.head_packs[i]
            hb, dom, code = self.head_buf[i], self.head_dom[i], self.head_code[i]
            pred = hb["pred"]
            npred = pred.shape[1]
            heads.append([
                [self._conv_job(lvl, stem, lvl.hp.data_ptr(), lvl.hp.shape[1], None, 0, hb["stem"].data_ptr(), nr, dom, code)],
                [self._conv_job(lvl, cr, hb["stem"].data_ptr(), nr, None, 0, hb["cr"].data_ptr(), 2 * nr, dom, code)],
                # pred columns: [reg(4) | obj(1) | cls(num_classes)] = order of collect_outputs (dagr.py:300-302)
                [self._conv_job(lvl, ro, hb["cr"].data_ptr() + 4 * nr, 2 * nr, None, 0, pred.data_ptr(), npred, dom, code),
                 self._conv_job(lvl, cls, hb["cr"].data_ptr(), 2 * nr, None, 0, pred.data_ptr() + 4 * 5, npred, dom, code)]])
        plan = [[layer5[0]] + heads[0][0], [layer5[1]] + heads[0][1], heads[1][0] + heads[0][2], heads[1][1], heads[1][2]]
        return [((_lib.ConvJob * len(jobs))(*jobs), len(jobs)) for jobs in plan]

    def _stage_level5_and_heads(self):
        """layer5 and both head scales by the launch plan of ``_build_tail_jobs``."""
        stream = _lib.cur_stream(self.device)
        self._pool_accumulated[3] = False
        for i in range(2):
            self._head_recode(i)
        for arr, count in self._tail_jobs:
            _lib.check(self.L.dagr_spline_conv_fused_multi(arr, count, stream), "spline_conv_fused_multi")

    def _stage_head_scale(self, i, scratch=None):
        """GNNHead.process_feature of scale i (dagr.py:179-190): stem, cls_conv | reg_conv (one launch, shared input),
        then reg_pred | obj_pred on the reg half and cls_pred on the cls half of that row as the two jobs of ONE paired
        launch (dagr_spline_conv_fused_pair) -> the scale's predictor rows."""
        L, P = self.L, _lib.ptr
        stream = _lib.cur_stream(self.device)
        lvln = self.head_levels[i]
        lvl = self.levels[lvln - 1]
        stem, cr, cls, ro = self.head_packs[i]
        hb = self.head_buf[i]
        dom = self.head_dom[i]
        code = self.head_code[i]
        self._head_recode(i)
        nr = self.n_reg
        self._conv_generic(lvl, stem, P(lvl.hp), lvl.hp.shape[1], None, 0, P(hb["stem"]), nr, dom, stream, code, scratch)
        self._conv_generic(lvl, cr, P(hb["stem"]), nr, None, 0, P(hb["cr"]), 2 * nr, dom, stream, code, scratch)
        pred = hb["pred"]
        npred = pred.shape[1]
        # pred columns: [reg(4) | obj(1) | cls(num_classes)] = order of collect_outputs (dagr.py:300-302)
        x_reg, x_cls = ctypes.c_void_p(hb["cr"].data_ptr() + 4 * nr), P(hb["cr"])
        o_reg, o_cls = P(pred), ctypes.c_void_p(pred.data_ptr() + 4 * 5)
        if self.fuse_convs and L.dagr_spline_conv_fused_passes(ro.cin, 0) >= 1 and \
                (L.dagr_spline_conv_fused_passes(ro.cin, 0) == 1 or lvl.T <= self.fused_passes_max_nodes):
            kcode = lvl.code if code is None else code
            _lib.check(L.dagr_spline_conv_fused_pair(P(lvl.counts), lvl.T, P(lvl.rowptr), P(lvl.col), P(kcode), 2 * nr, ro.cin,
                                                     dom["rx"], dom["ry"], dom["den_x"], dom["den_y"], npred, 0, x_reg,
                                                     P(ro.Wq), P(ro.bias), o_reg, ro.N, x_cls, P(cls.Wq), P(cls.bias), o_cls,
                                                     cls.N, stream), "spline_conv_fused_pair")
        else:
            self._conv_generic(lvl, ro, x_reg, 2 * nr, None, 0, o_reg, npred, dom, stream, code, scratch)
            self._conv_generic(lvl, cls, x_cls, 2 * nr, None, 0, o_cls, npred, dom, stream, code, scratch)
        return pred

    def _heads_finish(self):
        """to_dense of every scale (spline_conv.py:80-107) + the CNN head's logits (dagr.py:219-222,230-234) +
        collect_outputs / decode_outputs (dagr.py:283-312): one launch (dagr_heads_finish).  The fused logit maps land in
        ``head_buf[i]["dense"]`` as a by-product (traces, tests)."""
        P = _lib.ptr
        if self._cnn_ready is not None:             # pipelined window: the CNN head ran beside the graph levels
            torch.cuda.current_stream(self.device).wait_event(self._cnn_ready)
            self._cnn_ready = None
            self._feat_ready = None
        scales = []
        for i, lvln in enumerate(self.head_levels):
            lvl, hb = self.levels[lvln - 1], self.head_buf[i]
            Hc, Wc = self.out_sizes[i]
            vox = self.head_vox[i]
            hs = _lib.HeadScale(n_ptr=lvl.counts.data_ptr(), n_max=lvl.T, pred=hb["pred"].data_ptr(), ld=hb["pred"].shape[1],
                                pos=lvl.pos.data_ptr(), batch=lvl.batch.data_ptr(), vx=float(vox[0]), vy=float(vox[1]),
                                stride=float(self.strides[i]), Hc=int(Hc), Wc=int(Wc), dense=hb["dense"].data_ptr())
            if self._cnn_out is not None:
                for k, name in enumerate(("reg_output", "obj_output", "cls_output")):
                    t = self._cnn_out[name][i]
                    hs.cnn[k] = t.data_ptr()
                    for j in range(4):
                        hs.cnn_stride[k][j] = int(t.stride(j))
            scales.append(hs)
        A = sum(int(h) * int(w) for h, w in self.out_sizes)
        CH = 5 + self.num_classes
        out = torch.empty((self.B, A, CH), dtype=torch.float32, device=self.device)
        s0, s1 = ctypes.byref(scales[0]), ctypes.byref(scales[1]) if len(scales) > 1 else None
        if self._post_key is not None:
            # a caller wants detections (forward_detections): every image's post-processing runs inside the same launch,
            # right behind its decode (dagr_heads_finish_detect) -- fresh buffers per call (inside a capture they belong
            # to the graph and are remembered with it)
            self._det = torch.empty((self.B, A, 6), dtype=torch.float32, device=self.device)
            self._n_keep = torch.empty((self.B,), dtype=torch.int32, device=self.device)
            conf, nms = self._post_key
            _lib.check(self.L.dagr_heads_finish_detect(s0, s1, self.B, CH, P(out), P(self.status), conf, nms,
                                                       float(max(self.W, self.H) + 1), P(self._det), P(self._n_keep),
                                                       _lib.cur_stream(self.device)), "heads_finish_detect")
            self._post_fresh = True
        else:
            _lib.check(self.L.dagr_heads_finish(s0, s1, self.B, CH, P(out), P(self.status), _lib.cur_stream(self.device)),
                       "heads_finish")
        self._fused_dense = [hb["dense"] for hb in self.head_buf]
        return out

    def stage_head(self):
        """GNNHead.process_feature per scale, then to_dense + fusion + decode in one launch (dagr.py:179-236,283-312)."""
        for i in range(len(self.head_levels)):
            self._stage_head_scale(i)
        return self._heads_finish()

    def _tail_and_head(self, trace=None):
        """Levels 1..4 and both head scales with the dependency structure the graph has: head scale 1 only needs level 3
        (out3), so it runs on a side stream next to pool4 -> layer5 -> head scale 2 (net.py:166-186, dagr.py:213-236);
        then to_dense + image-logit fusion + decode of both scales in one launch.  Returns the decoded outputs."""
        first = self.head_levels[0]                  # 3 when both scales exist, 4 with num_scales = 1
        if trace is None:
            if self._tail_jobs is None:
                self._tail_jobs = self._build_tail_jobs() or False
            if self._tail_jobs:
                for k in range(3):
                    self._stage_level(k)
                    self._stage_pool(k)
                self._stage_level5_and_heads()
                return self._heads_finish()
        done = [False] * len(self.head_levels)
        cur = torch.cuda.current_stream(self.device)
        forked = False
        for k in range(4):
            self._stage_level(k, trace)
            if k + 1 == first and first == 3 and trace is None and self.overlap_heads:
                if self._head_stream is None:
                    self._head_stream = torch.cuda.Stream(self.device)
                ev = torch.cuda.Event()
                ev.record(cur)
                self._head_stream.wait_event(ev)
                with torch.cuda.stream(self._head_stream):
                    self._stage_head_scale(0, scratch=self.A2)
                    done[0] = True
                    self._head_join = torch.cuda.Event()
                    self._head_join.record(self._head_stream)
                forked = True
            if k < 3:
                self._stage_pool(k)
        for i in range(len(self.head_levels)):
            if not done[i]:
                self._stage_head_scale(i)
        if forked:
            cur.wait_event(self._head_join)
        return self._heads_finish()

    def _post_launch(self, out):
        """``postprocess_network_output`` (model/utils.py:61-110; dagr.py:94-95) of the decoded outputs for the paths that do
        not end in ``_heads_finish`` (``--no_events``: the image branch's own maps); everywhere else the heads' launch has
        already run it (dagr_heads_finish_detect).  Only when a caller asked for detections (forward_detections)."""
        if self._post_key is None or self._post_fresh:     # (nobody asked / the heads' own launch already did it)
            return
        B, A, C = out.shape
        self._det = torch.empty((B, A, 6), dtype=torch.float32, device=self.device)
        self._n_keep = torch.empty((B,), dtype=torch.int32, device=self.device)
        conf, nms = self._post_key
        _lib.check(self.L.dagr_postprocess(_lib.ptr(out), B, A, int(self.num_classes), conf, nms,
                                           float(max(self.W, self.H) + 1), _lib.ptr(self._det), _lib.ptr(self._n_keep),
                                           _lib.cur_stream(self.device)), "postprocess")
        self._post_fresh = True

    def forward_detections(self, pos, feat, batch, image=None, append=False):
        """One window (``append``: one asynchronous update) through forward + post-processing: ``(det[B, A, 6], n_keep[B])``
        as ``model.utils.postprocess_device`` returns them, valid until the engine's next call.  In latency mode the
        post-processing is part of the captured graph: no launch, and no host time, between the heads and the NMS."""
        from .model.utils import postprocess_device
        key = (float(self.model.conf_threshold), float(self.model.nms_threshold))
        if key != self._post_key:                   # (re)capture with these thresholds
            self._post_key = key
            self._wg = None
            self._wg_warm = 0
            self._graph = None
            self._graph_warm = 0
        self._post_fresh = False
        if append:
            out = self.forward_append(pos, feat, batch, static_out=True)
        else:
            out = self.forward_raw(pos, feat, batch, image=image, static_out=True)
        if self._post_fresh:
            return self._det, self._n_keep
        return postprocess_device(out, self.num_classes, key[0], key[1], self.H, self.W)

    def forward_raw(self, pos, feat, batch, image=None, trace=None, image_handle=None, static_out=False):
        """pos fp32[N,3] normalised (format_data), feat fp32[N,1], batch int32/int64[N] on the device;
        image fp32[B,3,H,W] in [0,1] when the model was built with --use_image.
        Returns decoded head outputs [B, n_anchors, 5+num_classes] (GNNHead.forward eval).
        ``static_out``: the caller consumes the outputs before the engine's next window (``DAGR.forward`` post-processes
        them at once), so a captured window may hand out its own output buffer instead of a copy."""
        if trace is None and image_handle is None and self.window_graph and self.l0_tiles and not self.no_events \
                and (image is not None or not self.use_image):
            return self._forward_window_graph(pos, feat, batch, image, static_out)
        self._cnn_out = None
        if self.use_image:
            if image_handle is not None:
                self._consume_image_handle(image_handle)
            elif image is None:
                raise RuntimeError("model was built with --use_image: an image batch is required")
            else:
                self.stage_image(image)
        if self.no_events:
            # GNNHead.forward eval, dagr.py:283-284: the image branch's own outputs (dagr.py:207-212); the reference
            # still runs the event path and throws its maps away -- here it is skipped
            c = self._cnn_out
            return self._decode_maps([torch.cat([c["reg_output"][k], c["obj_output"][k], c["cls_output"][k]], 1)
                                      for k in range(self.num_scales)])
        self.stage_graph(pos.contiguous(), batch.contiguous(), feat)
        self.stage_l0_input(feat)
        self.stage_l0_conv1()
        self.stage_l0_conv2()
        if trace is not None:
            trace["nbr"] = tuple(t.clone() for t in self._nbr)
            _, ev_slot = self.graph.node_order(self._N)     # traces are reported in event order
            ev_slot = ev_slot.long()
            trace["layer1"] = self.hp0[:self._N, :16][ev_slot].clone()
            if self.use_image:
                back = [self.x0_cols.index(k) for k in range(len(self.x0_cols))]   # reference channel order
                trace["x0"] = self._x0[ev_slot][:, back].clone()
        self.stage_pool1()
        if trace is None and self.tail_graph and not self.use_image:
            return self._replay_tail(static_out)
        out = self._tail_and_head(trace)
        if trace is not None:
            trace["head_dense"] = [o.clone() for o in self._fused_dense]
        return out

    def _forward_static(self):
        """One window on the engine's static input buffers with every launch sized for the event capacity and bounded by
        the device-side counts: the body of the captured window graph (and of its eager warm-up runs)."""
        self._dev_mode = True
        try:
            self._cnn_out = None
            if self.use_image:
                # the graph build needs nothing from the frame: it runs beside the image branch (a fork inside the captured
                # window; the level-0 input rows are the first stage that needs both)
                cur = torch.cuda.current_stream(self.device)
                if self._head_stream is None:
                    self._head_stream = torch.cuda.Stream(self.device)
                fork = torch.cuda.Event()
                fork.record(cur)
                self._head_stream.wait_event(fork)
                with torch.cuda.stream(self._head_stream):
                    self.stage_graph(self.in_pos, self.in_batch, self.in_feat)
                    join = torch.cuda.Event()
                    join.record(self._head_stream)
                if self.pipeline_image:
                    # The graph levels do not wait for the whole image branch: level k samples feature map k (+ 1), which
                    # exists as soon as ResNet stage k has run (net.py:110-184: the reference calls the CNN first, but its
                    # outputs are consumed level by level).  The branch gets a stream of its own and records an event per
                    # map; a B = 1 window is two chains of small dependent kernels, and the shorter one (the graph levels,
                    # ~0.4 ms) now runs UNDER the longer one (the CNN, ~1.3 ms) instead of after it.
                    if self._img_stream is None:
                        self._img_stream = torch.cuda.Stream(self.device)
                    s_img = self._img_stream
                    s_img.wait_event(fork)
                    self._feat_ready = {}
                    self._keep = []
                    with torch.cuda.stream(s_img):
                        def on_feature(j, fmap):
                            ev = torch.cuda.Event()
                            ev.record(s_img)
                            self._feat_ready[id(fmap)] = ev
                            fmap.record_stream(cur)
                        self._img_feats, self._cnn_out = self._image_branch(self.in_image, on_feature)
                        for v in self._cnn_out.values():
                            for o in v:
                                o.record_stream(cur)
                        self._cnn_ready = torch.cuda.Event()
                        self._cnn_ready.record(s_img)
                else:
                    self.stage_image(self.in_image)
                cur.wait_event(join)
            else:
                self.stage_graph(self.in_pos, self.in_batch, self.in_feat)
            self.stage_l0_input(self.in_feat)
            self.stage_l0_conv1()
            self.stage_l0_conv2()
            self.stage_pool1()
            out = self._tail_and_head()
            self._post_launch(out)
            return out
        finally:
            self._dev_mode = False

    def _forward_window_graph(self, pos, feat, batch, image, static_out=False):
        """Latency mode: the caller's window is staged into the static buffers by ONE launch (which also writes the event
        count to device memory); everything else -- image branch, graph build, level 0, pooled levels, heads, decode: ~45
        launches events-only, several hundred with the ResNet-50 branch -- is one replayed HIP graph.  Until round 4 only the
        part after pool1 was captured and the image model not at all (the host issued every MIOpen launch of a window)."""
        L, P = self.L, _lib.ptr
        N = int(pos.shape[0])
        if N > self.max_events:
            self._alloc_events(max(N, 2 * self.max_events))
        pos = pos.float().contiguous()
        feat = feat.float().reshape(-1).contiguous()
        batch = batch.contiguous()
        g = self.graph
        _lib.check(L.dagr_stage_window(ctypes.byref(g.desc), P(g.workspace), P(pos), P(feat), P(batch),
                                       1 if batch.dtype == torch.int64 else 0, N, P(self.in_pos), P(self.in_feat),
                                       P(self.in_batch), P(self.n_dev), _lib.cur_stream(self.device)), "stage_window")
        if self.use_image:
            if self.in_image is None or self.in_image.shape != image.shape:
                self.in_image = torch.empty(tuple(image.shape), dtype=torch.float32, device=self.device)
                self._wg = None
                self._wg_warm = 0
            self.in_image.copy_(image)
        if self._wg is None:
            if self._wg_warm < 2:                    # lazy one-time work (kernel attributes, MIOpen's search) stays eager
                self._wg_warm += 1
                out = self._forward_static()
            else:
                g = torch.cuda.CUDAGraph()
                with _capture(g):
                    out = self._forward_static()
                self._wg, self._wg_out, self._wg_post = g, out, (self._det, self._n_keep)
                g.replay()
        else:
            self._wg.replay()
            out = self._wg_out
            self._det, self._n_keep = self._wg_post
            self._post_fresh = self._post_key is not None
        # the resident window (what check_status / an asynchronous update / the probes look at): the actual count
        self._N = N
        self._n_rows = N
        self._async_on = False
        self._pos, self._batch = self.in_pos[:N], self.in_batch[:N]
        self._nbr = (self.nbr_src[:N], self.nbr_code[:N], self.deg[:N])
        self._x0 = self.x0buf[:N]
        return out if static_out else out.clone()   # the graph's output buffer is rewritten by the next window

    def _replay_tail(self, static_out=False):
        """Everything after pool1 has launch shapes that do not depend on the window (node / edge counts stay on the
        device): ~70 small dependent launches, captured once as a HIP graph and replayed -- the host then issues ONE
        launch for them, which is what bounds single-window latency at small N.  (Events-only: with --use_image the tail
        samples this window's freshly allocated feature maps.)"""
        if self._graph is None:
            if self._graph_warm < 2:                 # lazy one-time work (function attributes, allocator) stays eager
                self._graph_warm += 1
                out = self._tail_and_head()
                self._post_launch(out)
                return out
            g = torch.cuda.CUDAGraph()
            with _capture(g):
                out = self._tail_and_head()
                self._post_launch(out)
            self._graph, self._graph_out, self._graph_post = g, out, (self._det, self._n_keep)
        self._graph.replay()
        self._det, self._n_keep = self._graph_post
        self._post_fresh = self._post_key is not None
        return self._graph_out if static_out else self._graph_out.clone()   # rewritten by the next window

    def _decode_maps(self, dense_maps):
        """collect_outputs + decode_outputs (dagr.py:283-312) in one launch (dagr_decode_heads)."""
        P = _lib.ptr
        d = [m.contiguous() for m in dense_maps]
        self._keep_maps = d
        CH = d[0].shape[1]
        A = sum(m.shape[2] * m.shape[3] for m in d)
        out = torch.empty((self.B, A, CH), dtype=torch.float32, device=self.device)
        second = d[1] if len(d) > 1 else None
        _lib.check(self.L.dagr_decode_heads(P(d[0]), d[0].shape[2], d[0].shape[3], float(self.strides[0]), P(second),
                                            second.shape[2] if second is not None else 0,
                                            second.shape[3] if second is not None else 0,
                                            float(self.strides[1]) if second is not None else 0.0, self.B, CH, P(out),
                                            _lib.cur_stream(self.device)), "decode_heads")
        return out

    def _level_snapshot(self, lvl, x):
        n, e = [int(v) for v in lvl.counts.tolist()]
        return dict(x=x[:n].clone(), pos=lvl.pos[:n].clone(), batch=lvl.batch[:n].clone(),
                    rowptr=lvl.rowptr[:n + 1].clone(), col=lvl.col[:e].clone(), code=lvl.code[:e].clone())

    def l0_kernel_names(self):
        """Kernel (as rocprofv3 prints it) behind each level-0 conv stage, for bench.py's roofline line."""
        c0 = 3 + self.feat_ch[0]
        nt = self.ntaps0
        if self.l0_tiles:
            tx, ty = self.win0[1], self.win0[3]
            cm = 16 if c0 >= 16 else 0
            return {"l0_conv1": f"k_conv_l0_tiles<{cm}, {c0 - cm}, 0, {tx}, {ty}, true>",     # (LEAN: csrc/conv_l0_tiles.hip:launch_tiles)
                    "l0_conv2": f"k_conv_l0_tiles<16, 0, {c0}, {tx}, {ty}, true>"}
        return {"l0_conv1": f"k_conv_l0<{c0}, 0, {nt}>", "l0_conv2": f"k_conv_l0<16, {c0}, {nt}>"}

    def check_status(self):
        """Raise if any kernel flagged an inconsistency (synchronises)."""
        ne, fl = self.graph.status()
        if fl:
            raise RuntimeError(f"graph builder flagged {fl:#x} (events outside the sensor / batch range)")
        for k, (d, ws) in enumerate(zip(self.pool_desc, self.pool_ws)):
            f = ctypes.c_int32(0)
            _lib.check(self.L.dagr_pool_status(ctypes.byref(d), _lib.ptr(ws), ctypes.byref(f),
                                               _lib.cur_stream(self.device)), "pool_status")
            if f.value:
                raise RuntimeError(f"pool{k + 1} flagged {f.value:#x}")
        if self._async_on:
            if int(self._app["status"][0]):
                raise RuntimeError("asynchronous update: event outside the sensor / batch range")
            f = ctypes.c_int32(0)
            _lib.check(self.L.dagr_pool_status(ctypes.byref(self.pool_desc[0]), _lib.ptr(self._app["pool_ws"]),
                                               ctypes.byref(f), _lib.cur_stream(self.device)), "pool_status")
            if f.value:
                raise RuntimeError(f"pool1 (asynchronous accumulators) flagged {f.value:#x}")
        st = self.status.tolist()
        if st[0]:
            raise RuntimeError("to_dense: node outside the output map")
        if st[1]:
            raise RuntimeError("head: LUT coordinate outside the head's table (dagr_pool_recode)")

    def check_batch(self, data):
        """Host-side validation of a batch against the engine's constants (no device read-back)."""
        ng = getattr(data, "num_graphs", None)
        if ng is not None and int(ng) > self.B:
            raise RuntimeError(f"batch of {int(ng)} windows, but the model was built with batch_size = {self.B}")
        geo = getattr(data, "_geometry", None)          # python ints stashed by the collation (data/__init__.py)
        for k, (name, want) in enumerate((("width", self.W), ("height", self.H), ("time_window", self.time_window))):
            v = geo[k] if geo is not None else getattr(data, name, None)
            if v is not None and not (torch.is_tensor(v) and v.is_cuda):   # no device read-back on the hot path
                v = int(v[0]) if hasattr(v, "__len__") else int(v)
                if v != want:
                    raise RuntimeError(f"data.{name} = {v}, but the model was built for {want}")

    def forward_detections_data(self, data):
        """``forward_detections`` on ``DAGR.forward``'s input contract."""
        batch = data.batch if getattr(data, "batch", None) is not None else \
            torch.zeros(data.pos.shape[0], dtype=torch.int64, device=data.pos.device)
        self.check_batch(data)
        return self.forward_detections(data.pos.float(), data.x.float(), batch, image=getattr(data, "image", None))

    def forward_data(self, data, static_out=False):
        """``DAGR.forward`` input contract: ``data`` after ``format_data`` (pos fp32[N,3] normalised,
        x fp32[N,1], batch)."""
        batch = data.batch if getattr(data, "batch", None) is not None else \
            torch.zeros(data.pos.shape[0], dtype=torch.int64, device=data.pos.device)
        self.check_batch(data)
        return self.forward_raw(data.pos.float(), data.x.float(), batch, image=getattr(data, "image", None),
                                static_out=static_out)
