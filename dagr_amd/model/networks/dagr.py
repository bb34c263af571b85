"""Mirror of ``src/dagr/model/networks/dagr.py``: ``DAGR`` (:14-103, a YOLOX with ``backbone`` = Net
and ``head`` = GNNHead), ``CNNHead`` (:106-122), ``GNNHead`` (:125-312, eval branch).  Sub-module
names and parameter shapes follow the reference so ``ModelEMA(model).ema.load_state_dict(ckpt['ema'])``
(scripts/run_test.py:57-58) works.  The forward pass is executed by ``dagr_amd.engine.WindowEngine``
(hand-written HIP kernels behind include/dagr_hip.h) in eval mode; in training mode the layers run module by module
with gradients (``model/layers/autograd.py``) and the head returns the YOLOX losses (``yolox_loss.py``)."""
import torch

from ..layers.conv import ConvBlock
from ..layers.spline_conv import SplineConvToDense
from ..utils import (voxel_size_to_params, postprocess_network_output, convert_to_evaluation_format,
                     detections_from_device)
from .net import Net
from .yolox_min import YOLOXHeadParams


class CNNHead(YOLOXHeadParams):
    """dagr.py:106-122 -- dense YOLOX head on the (resized) image features; PyTorch-ROCm."""

    def forward(self, xin):
        outputs = dict(cls_output=[], reg_output=[], obj_output=[])
        for k, (cls_conv, reg_conv, x) in enumerate(zip(self.cls_convs, self.reg_convs, xin)):
            x = self.stems[k](x)
            cls_feat = cls_conv(x)
            reg_feat = reg_conv(x)
            outputs["cls_output"].append(self.cls_preds[k](cls_feat))
            outputs["reg_output"].append(self.reg_preds[k](reg_feat))
            outputs["obj_output"].append(self.obj_preds[k](reg_feat))
        return outputs


class GNNHead(YOLOXHeadParams):
    def __init__(self, num_classes, strides=(8, 16, 32), in_channels=(256, 512, 1024),
                 in_channels_cnn=(256, 512, 1024), act="silu", depthwise=False, pretrain_cnn=False, args=None):
        YOLOXHeadParams.__init__(self, num_classes, args.yolo_stem_width, strides, in_channels, act)
        self.pretrain_cnn = pretrain_cnn
        self.num_scales = args.num_scales
        self.use_image = bool(args.use_image)
        self.batch_size = args.batch_size
        self.no_events = bool(args.no_events)
        self.in_channels = list(in_channels)
        self.n_anchors = 1
        self.num_classes = num_classes
        n_reg = max(in_channels)
        self.stem1 = ConvBlock(in_channels=in_channels[0], out_channels=n_reg, args=args)
        self.cls_conv1 = ConvBlock(in_channels=n_reg, out_channels=n_reg, args=args)
        self.cls_pred1 = SplineConvToDense(n_reg, self.n_anchors * self.num_classes, bias=True, args=args)
        self.reg_conv1 = ConvBlock(in_channels=n_reg, out_channels=n_reg, args=args)
        self.reg_pred1 = SplineConvToDense(n_reg, 4, bias=True, args=args)
        self.obj_pred1 = SplineConvToDense(n_reg, self.n_anchors, bias=True, args=args)
        if self.num_scales > 1:
            self.stem2 = ConvBlock(in_channels=in_channels[1], out_channels=n_reg, args=args)
            self.cls_conv2 = ConvBlock(in_channels=n_reg, out_channels=n_reg, args=args)
            self.cls_pred2 = SplineConvToDense(n_reg, self.n_anchors * self.num_classes, bias=True, args=args)
            self.reg_conv2 = ConvBlock(in_channels=n_reg, out_channels=n_reg, args=args)
            self.reg_pred2 = SplineConvToDense(n_reg, 4, bias=True, args=args)
            self.obj_pred2 = SplineConvToDense(n_reg, self.n_anchors, bias=True, args=args)
        if self.use_image:
            self.cnn_head = CNNHead(num_classes=num_classes, strides=strides, in_channels=in_channels_cnn)
        self.strides = list(strides)

    # -- dagr.py:179-236,283-312 (eval branch), module by module ---------------------------------------------
    def process_feature(self, x, stem, cls_conv, reg_conv, cls_pred, reg_pred, obj_pred, batch_size):
        from ..utils import shallow_copy
        x = stem(x)
        cls_feat = cls_conv(shallow_copy(x))
        reg_feat = reg_conv(x)
        cls_output = cls_pred(cls_feat, batch_size=batch_size)
        reg_output = reg_pred(shallow_copy(reg_feat), batch_size=batch_size)
        obj_output = obj_pred(reg_feat, batch_size=batch_size)
        return cls_output, reg_output, obj_output

    def _graphed_losses(self, fn, lab, flat, tag):
        """The YOLOX loss (SimOTA assignment + three terms: ~150 small torch launches forward, ~200 backward, all on
        shapes that depend only on the configuration) as two replayed HIP graphs (``torch.cuda.make_graphed_callables``);
        the training step is host-bound, and this is a quarter of its launches.  Captured once per (shapes, device);
        ``DAGR_GRAPH_LOSS=0`` or a capture that fails falls back to the launch-by-launch form (same kernels, same bits).
        A replayed graph owns its input / output buffers, so a call site's graph serves ONE forward at a time: ``losses``
        below keeps a "backward pending" mark per call site (set by the forward, cleared when the backward has run through
        the graph or the loss tensor has been dropped), and a forward that arrives while the mark is set -- gradient
        accumulation, several micro-batches summed before one backward -- takes the launch-by-launch form instead of
        overwriting the first forward's buffers.  Returns (callable, key); key is None for the launch-by-launch form."""
        import os
        if not lab.is_cuda or os.environ.get("DAGR_GRAPH_LOSS", "1") == "0" or not torch.is_grad_enabled() \
                or not all(t.requires_grad for t in flat):
            return fn, None
        # (one capture per call site: a graphed callable owns its input / output buffers, and the image branch's loss and
        # the hybrid loss of a --use_image step are both alive until backward)
        key = (tag, str(lab.device), tuple(lab.shape), tuple(tuple(t.shape) for t in flat), flat[0].dtype)
        cache = self.__dict__.setdefault("_loss_graphs", {})
        if key not in cache:
            try:
                sample = (torch.zeros_like(lab),) + tuple(torch.randn_like(t).requires_grad_(True) for t in flat)
                sample[0][:, 0, 1:] = torch.tensor([40.0, 40.0, 30.0, 30.0], device=lab.device)      # one box per image
                import gc
                gc_was_on = gc.isenabled()
                gc.disable()                            # (no finalizers into the HIP runtime while the stream captures)
                try:
                    cache[key] = torch.cuda.make_graphed_callables(fn, sample)
                finally:
                    if gc_was_on:
                        gc.enable()
            except Exception as exc:                    # capture not possible here: keep the eager form
                import warnings
                warnings.warn(f"loss graph capture failed ({exc}); using the launch-by-launch form")
                cache[key] = None                       # (a sentinel: the fallback is the caller's own eager closure)
        if cache[key] is None:
            return fn, None
        pending = self.__dict__.setdefault("_loss_graph_pending", {})
        mark = pending.get(key)
        if mark is not None and not mark[1] and mark[0]() is not None:
            return fn, None                             # the previous forward of this call site still awaits its backward
        return cache[key], key

    def forward(self, xin, labels=None, imgs=None, output_sizes=None):
        """Eval: decoded ``[B, n_anchors, 5 + num_classes]`` from the backbone outputs (and the image outputs with
        ``--use_image``); ``--no_events`` returns the image branch's own detections (dagr.py:283-284).
        Training (dagr.py:238-282): the YOLOX losses of the hybrid outputs (+ those of the image branch's own outputs
        with ``--use_image``; only those with ``--pretrain_cnn``) as the 6-tuple ``get_losses`` returns."""
        if output_sizes is None:
            output_sizes = getattr(self, "output_sizes", None)
        out_cnn = None
        image_labels = None
        if self.use_image:
            xin, image_feat = xin
            if labels is not None:
                labels, image_labels = labels
            image_feat = [torch.nn.functional.interpolate(f, o) for f, o in zip(image_feat, output_sizes)]
            out_cnn = self.cnn_head(image_feat)
        batch_size = len(out_cnn["cls_output"][0]) if self.use_image else self.batch_size
        raw, image_raw = [], []          # per scale: [reg | obj | cls] logits
        for k, g in enumerate(xin):
            s = str(k + 1)
            cls_o, reg_o, obj_o = self.process_feature(g, *(getattr(self, n + s) for n in
                                                            ("stem", "cls_conv", "reg_conv", "cls_pred", "reg_pred",
                                                             "obj_pred")), batch_size=batch_size)
            if out_cnn is not None:
                # dagr.py:219-222,230-234: the image logits enter the hybrid sum detached
                cls_o = cls_o + out_cnn["cls_output"][k].detach()
                reg_o = reg_o + out_cnn["reg_output"][k].detach()
                obj_o = obj_o + out_cnn["obj_output"][k].detach()
                image_raw.append((out_cnn["reg_output"][k], out_cnn["obj_output"][k], out_cnn["cls_output"][k]))
            raw.append((reg_o, obj_o, cls_o))
        if self.training:
            from .yolox_loss import detection_losses, output_and_grid

            def losses_eager(lab, *flat):
                maps = [flat[3 * k:3 * k + 3] for k in range(len(flat) // 3)]
                outs, grids = zip(*(output_and_grid(torch.cat(m, 1), st) for m, st in zip(maps, self.strides)))
                r = detection_losses(lab, torch.cat(outs, 1), list(grids), self.strides[:len(maps)], self.num_classes)
                return r[0], r[1], r[2], r[3], r[5]          # (r[4]: the L1 term, 0.0 with use_l1 = False, dagr.py:168)

            def losses(maps, lab, tag="events"):
                flat = [t for m in maps for t in m]
                fn, key = self._graphed_losses(losses_eager, lab, flat, tag)
                total, iou, obj, cls, ratio = fn(lab, *flat)
                if key is not None:
                    # a replayed graph hands out ITS buffers, rewritten by the next step: the caller gets values of its own
                    # (one small launch for the five scalars; `total` keeps its link to the graph's backward)
                    total, iou, obj, cls, ratio = torch.stack((total, iou, obj, cls, ratio)).unbind(0)
                    # the mark lives as long as the AUTOGRAD NODE of this loss, not the Python tensor: the hybrid loss of a
                    # --use_image step (`loss_image[i] + loss_events[i]`) and callers that sum losses drop `total` while its
                    # backward is still pending; a sentinel in the node's metadata dies only when the graph itself is freed
                    import weakref
                    alive = _Sentinel()
                    total.grad_fn.metadata["dagr_loss_graph_pending"] = alive
                    mark = [weakref.ref(alive), False]           # [this forward's graph is alive, its backward has run]
                    del alive
                    self._loss_graph_pending[key] = mark

                    def _done(grad, mark=mark):
                        mark[1] = True
                        return grad
                    total.register_hook(_done)
                return total, iou, obj, cls, 0.0, ratio
            if self.use_image:
                # dagr.py:241-268: CNNHead always yields both scales; the image branch learns to detect on its own
                both = [(out_cnn["reg_output"][k], out_cnn["obj_output"][k], out_cnn["cls_output"][k]) for k in (0, 1)]
                loss_image = list(losses(both, image_labels, "image"))
                if not self.pretrain_cnn:
                    loss_events = losses(raw, labels)
                    for i in range(5):
                        loss_image[i] = loss_image[i] + loss_events[i]
                return tuple(loss_image)
            return losses(raw, labels)
        maps = [torch.cat([r, o.sigmoid(), c.sigmoid()], 1) for r, o, c in raw]
        image_maps = [torch.cat([r, o.sigmoid(), c.sigmoid()], 1) for r, o, c in image_raw]
        out = image_maps if self.no_events else maps
        outputs = torch.cat([o.flatten(start_dim=2) for o in out], dim=2).permute(0, 2, 1).contiguous()
        from ..utils import init_grid_and_stride
        grid, stride = init_grid_and_stride([o.shape[-2:] for o in out], self.strides, outputs)
        outputs[..., :2] = (outputs[..., :2] + grid) * stride
        outputs[..., 2:4] = torch.exp(outputs[..., 2:4]) * stride
        return outputs


class _Sentinel:
    """Weak-referenceable token kept in an autograd node's metadata (see ``GNNHead.forward``: the loss-graph guard)."""
    __slots__ = ("__weakref__",)


def _window_part(d):
    """What a running window remembers of one call: the events (pos, polarity, sample index) -- not the Data object."""
    batch = d.batch if getattr(d, "batch", None) is not None else \
        torch.zeros(d.pos.shape[0], dtype=torch.int64, device=d.pos.device)
    return d.pos.float(), d.x.float().view(-1, 1), batch.long()


def _concat_window(parts):
    """All events of the running window ordered by sample, arrival order inside a sample: the layout one reset=True call
    on the same events has (collation concatenates sample after sample).  The list is compacted to one part."""
    if len(parts) > 1:
        parts[:] = [tuple(torch.cat([p[k] for p in parts]) for k in range(3))]
    pos, feat, batch = parts[0]
    order = torch.argsort(batch, stable=True)
    return pos[order].contiguous(), feat[order].contiguous(), batch[order].contiguous()


class DAGR(torch.nn.Module):
    def __init__(self, args, height, width):
        super().__init__()
        self.conf_threshold = 0.001
        self.nms_threshold = 0.65
        self.check_device_status = True    # set False to skip the per-call status read-back
        self.height = height
        self.width = width
        self.args = args
        self.backbone = Net(args, height=height, width=width)
        self.head = GNNHead(num_classes=self.backbone.num_classes, in_channels=self.backbone.out_channels,
                            in_channels_cnn=self.backbone.out_channels_cnn, strides=self.backbone.strides,
                            pretrain_cnn=args.pretrain_cnn, args=args)
        self._engine = None
        self._engine_stamp = None
        self._stamp_tensors = None
        self._window = None          # the events since the last reset=True call (DAGR.forward(reset=False))
        self._window_image = None
        self.asynchronous = True     # reset=False calls update incrementally (asynchronous.make_model_synchronous: off)
        if bool(args.no_events) and not bool(args.use_image):
            raise ValueError("--no_events returns the image branch's detections (dagr.py:283-284): it needs --use_image")
        # --keep_temporal_ordering (pooling.py:69-72): the coarse-edge filter lives in the Pooling modules; the window
        # engine's fused pooling does not apply it, so eval forwards of such a model run module by module
        self.module_path_only = bool(getattr(args, "keep_temporal_ordering", False))
        if "img_net_checkpoint" in vars(args):
            from ..utils import init_subnetwork
            state_dict = torch.load(args.img_net_checkpoint)
            init_subnetwork(self, state_dict["ema"], "backbone.net.", freeze=True)
            init_subnetwork(self, state_dict["ema"], "head.cnn_head.")

    # -- dagr.py:37-72 -------------------------------------------------------------------------
    def cache_luts(self, width, height, radius):
        M = 2 * float(int(radius * width + 2) / width)
        r = int(radius * width + 1)
        b, h = self.backbone, self.head
        for blk in (b.conv_block1.conv_block1, b.conv_block1.conv_block2):
            blk.conv.init_lut(height=height, width=width, Mx=M, rx=r)
        levels = [(b.pool1, [b.layer2]), (b.pool2, [b.layer3]), (b.pool3, [b.layer4]), (b.pool4, [b.layer5])]
        for k, (pool, layers) in enumerate(levels):
            rx, ry, M = voxel_size_to_params(pool, height, width)
            for layer in layers:
                layer.conv_block1.conv.init_lut(height=height, width=width, Mx=M, rx=rx, ry=ry)
                layer.conv_block2.conv.init_lut(height=height, width=width, Mx=M, rx=rx, ry=ry)
            s = {2: "1", 3: "2"}.get(k)
            if s is not None and (s == "1" or h.num_scales > 1):
                for name in ("stem", "cls_conv", "reg_conv"):
                    getattr(h, name + s).conv.init_lut(height=height, width=width, Mx=M, rx=rx, ry=ry)
                for name in ("cls_pred", "reg_pred", "obj_pred"):
                    getattr(h, name + s).init_lut(height=height, width=width, Mx=M, rx=rx, ry=ry)
        self._engine = None  # parameters / domains changed: rebuild the device-side plan lazily

    def forward_modules(self, x, reset=True):
        """The eval forward as the reference wires it (``YOLOX.forward``: head(backbone(x))), every layer through its own
        module-level operator instead of the window engine.  Same decoded outputs; used to check the operator API and
        for ``reset=False`` calls on the layers that support them."""
        x.reset = reset
        return self.head(self.backbone(x), output_sizes=self.backbone.get_output_sizes()[-self.head.num_scales:])

    def forward_training(self, x):
        """dagr.py:78-88 + ``YOLOX.forward`` (training branch): targets in (class, cx, cy, w, h) rows, the layers module
        by module with gradients (differentiable SplineConv / pooling / to_dense over libdagr_hip, batch-statistics
        BatchNorm), the YOLOX losses as a dict with the reference's keys."""
        from ..utils import convert_to_training_format
        targets = convert_to_training_format(x.bbox, x.bbox_batch, x.num_graphs)
        if self.backbone.use_image:
            targets = (targets, convert_to_training_format(x.bbox0, x.bbox0_batch, x.num_graphs))
        self.head.output_sizes = self.backbone.get_output_sizes()[-self.head.num_scales:]
        x.reset = True
        loss, iou_loss, conf_loss, cls_loss, l1_loss, num_fg = self.head(self.backbone(x), targets, x)
        return {"total_loss": loss, "iou_loss": iou_loss, "l1_loss": l1_loss, "conf_loss": conf_loss,
                "cls_loss": cls_loss, "num_fg": num_fg}

    def _weights_stamp(self):
        """Cheap fingerprint of everything the engine snapshots (packed, BN-folded weights; the folded copy of the
        image branch): in-place edits bump ``_version``, ``.to()`` / ``load_state_dict`` on sub-modules
        (``init_subnetwork``) change storage or version."""
        # the tensor list is walked once per engine (module traversal is the expensive part of this check, and it
        # sits on the single-window latency path); `.to()` / `load_state_dict` / `train()` drop the engine themselves
        ts = self._stamp_tensors
        if ts is None:
            ts = self._stamp_tensors = list(self.parameters()) + list(self.buffers())
        ver = 0
        for t in ts:
            ver += t._version
        return ver, (ts[0].data_ptr(), str(ts[0].device)) if ts else None

    def invalidate_engine(self):
        """Drop the device-side plan; the next forward re-packs the weights."""
        self._engine = None
        self._stamp_tensors = None

    def _apply(self, fn, *a, **kw):          # .to() / .cuda() / .float(): parameters are replaced
        self._engine = None
        self._stamp_tensors = None
        return super()._apply(fn, *a, **kw)

    def train(self, mode=True):
        if mode:
            self._stamp_tensors = None
        return super().train(mode)

    def engine(self):
        stamp = self._weights_stamp()
        if self._engine is None or self._engine_stamp != stamp:
            from ...engine import WindowEngine
            self._engine = None
            self._stamp_tensors = None
            self._engine = WindowEngine(self)
            self._engine_stamp = self._weights_stamp()
        return self._engine

    def load_state_dict(self, *a, **kw):
        r = super().load_state_dict(*a, **kw)
        self._engine = None
        self._stamp_tensors = None
        return r

    # -- dagr.py:74-103 (eval branch) ----------------------------------------------------------
    def forward(self, x, reset=True, return_targets=True, filtering=True):
        if self.training:
            return self.forward_training(x)
        if self.module_path_only:
            if not reset:
                raise NotImplementedError("reset=False with --keep_temporal_ordering: the module path evaluates whole windows")
            outputs = self.forward_modules(x, reset=True)
            detections = postprocess_network_output(outputs, self.backbone.num_classes, self.conf_threshold,
                                                    self.nms_threshold, filtering=filtering, height=self.height,
                                                    width=self.width)
            ret = [detections]
            if return_targets and hasattr(x, "bbox"):
                ret.append(convert_to_evaluation_format(x))
            return ret
        eng = self.engine()
        eng.check_batch(x)
        if reset:
            self._window = None                  # a new window: the running one is gone
        det_dev = None           # (det, n_keep) when forward + post-processing ran as one captured graph
        if reset or self._window is None:
            # a window of its own (every evaluation script's call), or the first call of an asynchronous run
            if filtering:
                det_dev = eng.forward_detections_data(x)     # every image's post-processing inside the heads' last launch
            else:
                outputs = eng.forward_data(x, static_out=filtering)  # post-processed below, before the next window
            # only remembered: a later reset=False call continues from it.  A running window keeps the FRAME of the call
            # that opened it (DSEC: the image at the start of the window, dsec_data.py:141-184); later micro-batches
            # bring events only, in the incremental and in the re-evaluating mode alike.
            self._window = [_window_part(x)]
            self._window_image = getattr(x, "image", None)
        elif self.asynchronous and eng.can_append():
            # dagr.py:90 `x.reset = reset` -> ev_tgn.py:45-56: the new events attach to the running graph.  Incremental:
            # only their level-0 rows are computed, pool1's resident accumulators are extended, the fixed-size part of
            # the network runs as for a window (engine.forward_append).  Equal to one reset=True call on all events so
            # far -- the guarantee the reference's asynchronous model gives for its update (evaluate_flops.py:139-147).
            batch = x.batch if getattr(x, "batch", None) is not None else \
                torch.zeros(x.pos.shape[0], dtype=torch.int64, device=x.pos.device)
            if filtering:
                det_dev = eng.forward_detections(x.pos, x.x, batch, append=True)
            else:
                outputs = eng.forward_append(x.pos, x.x, batch, static_out=filtering)
            self._window.append(_window_part(x))
            if len(self._window) > 64:           # bounded bookkeeping on long streams: one concatenated part
                _concat_window(self._window)
        else:
            # make_model_synchronous -- and every configuration the incremental path does not cover (max_neighbors != 16,
            # a 5x5 tap window, --no_events, a search radius beyond two voxels, an engine rebuilt after a weight edit):
            # the whole running window again (the reference's synchronous forward on all events)
            self._window.append(_window_part(x))
            pos, feat, batch = _concat_window(self._window)
            outputs = eng.forward_raw(pos, feat, batch, image=self._window_image)
            eng._async_on = False
        if det_dev is not None:
            detections = detections_from_device(*det_dev)
        else:
            detections = postprocess_network_output(outputs, self.backbone.num_classes, self.conf_threshold,
                                                    self.nms_threshold, filtering=filtering, height=self.height,
                                                    width=self.width)
        if self.check_device_status:
            # sticky device-side flags (events outside the sensor / batch range, pooled-level capacity overflows,
            # to_dense cells outside the map): a window that tripped one was computed on a truncated graph.  The
            # stream was just drained by postprocess, so this is a handful of 4-byte reads.
            eng.check_status()
        ret = [detections]
        if return_targets and hasattr(x, "bbox"):
            ret.append(convert_to_evaluation_format(x))
        return ret
