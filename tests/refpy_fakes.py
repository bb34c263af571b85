"""Functional stand-ins for the third-party packages the reference imports, built on the oracle's primitives.

TEST INFRASTRUCTURE for tests/make_golden_refpy.py only (it runs in the build container, where /root/reference
exists).  With these in ``sys.modules`` the reference's OWN model code -- ``Net``, ``Layer`` / ``ConvBlock``,
``MySplineConv`` (incl. ``init_lut`` / ``message_lut``), ``Pooling``, ``EV_TGN`` + ``SlidingWindowGraph``,
``GNNHead``, ``DAGR.cache_luts`` -- executes unmodified on CPU; what the stand-ins compute is exactly what the
oracle's restatements of the third-party primitives compute (oracle/ops.py, oracle/graph_oracle.c), so the golden
outputs pin the reference's *wiring* (which op sees which tensor, in which order), not the third-party arithmetic.

Parameter / buffer names follow the real packages (PyG ``SplineConv``: weight, lin.weight, bias, buffers
kernel_size / is_open_spline; PyG ``BatchNorm``: module.*), so a reference-layout ``state_dict`` loads strictly."""
import sys
import types

import numpy as np
import torch

from oracle import graph as og
from oracle import ops as oo
from oracle import postprocess as opost


class _Placeholder(types.ModuleType):
    """Any attribute is an empty class: enough for imports of names that are never called."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (), {"__init__": lambda self, *a, **k: None})
        setattr(self, name, cls)
        return cls


def _module(name, **attrs):
    m = sys.modules.get(name)
    if m is None:
        m = _Placeholder(name)
        m.__path__ = []
        sys.modules[name] = m
        if "." in name:                      # `import a.b as c` resolves b as an attribute of a
            parent, child = name.rsplit(".", 1)
            setattr(_module(parent), child, m)
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


# ------------------------------------------------------------------------------------------ torch_geometric.data
class Data:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __contains__(self, key):
        return key in self.__dict__


class Batch(Data):
    @staticmethod
    def from_data_list(lst):
        d = lst[0]
        d.batch = torch.zeros(len(d.x), dtype=torch.long)
        return d


# ------------------------------------------------------------------------------------------ transforms
class Cartesian:
    """T.Cartesian(norm=True, max_value, cat=False): edge_attr = (pos[src] - pos[dst]) / (2 max) + 0.5."""

    def __init__(self, norm=True, max_value=None, cat=True):
        self.norm, self.max, self.cat = norm, max_value, cat

    def __call__(self, data):
        assert self.norm and not self.cat
        data.edge_attr = oo.cartesian(data.pos, data.edge_index, self.max)
        return data


class _Adj:
    """What the reference needs of a torch_sparse.SparseTensor: the (destination-sorted) CSR and numel()."""

    def __init__(self, rowptr, src, val):
        self.rowptr, self.src, self.val = rowptr, src, val

    def numel(self):
        return self.src.numel()


class ToSparseTensor:
    def __init__(self, attr="edge_attr", remove_edge_index=True, **kw):
        assert attr == "edge_attr" and not remove_edge_index

    def __call__(self, data):
        rowptr, src, val, _ = oo.to_sparse(data.edge_index, data.edge_attr, data.x.shape[0])
        data.adj_t = _Adj(rowptr, src, val)
        return data


# ------------------------------------------------------------------------------------------ nn
class _PygLinear(torch.nn.Module):
    def __init__(self, ic, oc):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.zeros(oc, ic))

    def forward(self, x):
        return x @ self.weight.t()


class SplineConv(torch.nn.Module):
    """PyG SplineConv's construction and the part of MessagePassing.propagate the reference reaches (SparseTensor
    path, aggr = sum): message(x_j, edge_attr) in CSR order, then segment_csr sum per destination."""

    def __init__(self, in_channels, out_channels, dim, kernel_size, is_open_spline=True, degree=1, aggr="mean",
                 root_weight=True, bias=True, **kw):
        super().__init__()
        assert aggr == "sum", "reference configs: aggr: sum"
        self.in_channels, self.out_channels, self.dim, self.degree, self.aggr = in_channels, out_channels, dim, degree, aggr
        self.root_weight = root_weight
        self.register_buffer("kernel_size", torch.tensor([int(kernel_size)] * dim, dtype=torch.long))
        self.register_buffer("is_open_spline", torch.tensor([1] * dim, dtype=torch.uint8))
        self.weight = torch.nn.Parameter(torch.zeros(int(kernel_size) ** dim, in_channels, out_channels))
        if root_weight:
            self.lin = _PygLinear(in_channels, out_channels)
        if bias:
            self.bias = torch.nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter("bias", None)

    def message(self, x_j, edge_attr):      # PyG SplineConv.message; replaced by message_lut after init_lut
        basis, index = oo.spline_basis(edge_attr)
        return oo.spline_weighting(x_j, self.weight, basis, index)

    def propagate(self, edge_index, x, edge_attr=None, size=None):
        adj = edge_index
        assert isinstance(adj, _Adj) and edge_attr is None
        msg = self.message(x[0][adj.src], adj.val)
        return oo.segment_csr_sum(msg, adj.rowptr)


class BatchNorm(torch.nn.Module):
    def __init__(self, in_channels, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, **kw):
        super().__init__()
        self.module = torch.nn.BatchNorm1d(in_channels, eps, momentum, affine, track_running_stats)

    def forward(self, x):
        return self.module(x)


# ------------------------------------------------------------------------------------------ yolox
class YOLOX(torch.nn.Module):
    """yolox.models.YOLOX: ``forward(x, targets)`` -- eval: head(backbone(x)); training: the head's six loss terms as a
    dict (what ``DAGR.forward`` returns in training mode, dagr.py:86-88)."""

    def __init__(self, backbone=None, head=None):
        super().__init__()
        self.backbone, self.head = backbone, head

    def forward(self, x, targets=None):
        fpn_outs = self.backbone(x)
        if self.training:
            assert targets is not None
            loss, iou_loss, conf_loss, cls_loss, l1_loss, num_fg = self.head(fpn_outs, targets, x)
            return {"total_loss": loss, "iou_loss": iou_loss, "l1_loss": l1_loss, "conf_loss": conf_loss,
                    "cls_loss": cls_loss, "num_fg": num_fg}
        return self.head(fpn_outs)


def _yolox_head_base():
    """yolox.models.YOLOXHead as far as the reference uses it: the dense conv towers (stems, cls/reg convs and
    preds) that CNNHead.forward runs and every checkpoint carries -- the host mirror's parameter-compatible
    re-declaration (dagr_amd/model/networks/yolox_min.py) -- and the training losses ``GNNHead`` inherits
    (oracle/yolox_loss.py: the published algorithm, restated)."""
    from dagr_amd.model.networks.yolox_min import YOLOXHeadParams
    from oracle.yolox_loss import YOLOXLossMixin

    class YOLOXHead(YOLOXHeadParams, YOLOXLossMixin):
        def __init__(self, num_classes, width=1.0, strides=(8, 16, 32), in_channels=(256, 512, 1024), act="silu",
                     depthwise=False):
            assert not depthwise
            super().__init__(num_classes, width, strides, in_channels, act)
    return YOLOXHead


def _tv_resnet(name):
    """torchvision.models.resnetXX(pretrained=...) -> the mirror's torchvision-compatible ResNet; forward() runs the
    trunk through module calls, so the reference's forward hooks see conv1 / layer1..4 in order."""
    from dagr_amd.model.networks.net_img import make_img_net

    def ctor(pretrained=False, **kw):
        net = make_img_net(name)
        net.forward = lambda x: net.forward_features(x)["layer4"]
        return net
    return ctor


from oracle.yolox_loss import IOUloss  # noqa: E402  (yolox.models.IOUloss, restated)


# ------------------------------------------------------------------------------------------ ev_graph_cuda
def _np32(t):
    a = t.numpy()
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a


def insert_in_queue_cuda(sorted_indices, unique_coords, cumsum_counter, queue):
    B, Q, H, W = queue.shape
    uc = np.ascontiguousarray(unique_coords.numpy().astype(np.int32))
    og.lib().oracle_insert_in_queue(og._p32(_np32(sorted_indices.contiguous())), og._p32(uc),
                                    og._p32(_np32(cumsum_counter.contiguous())), og._p32(_np32(queue)), B, Q, H, W, len(uc))
    return queue


def insert_in_queue_single_cuda(indices, pos, queue):
    B, Q, H, W = queue.shape
    og.lib().oracle_insert_in_queue_single(og._p32(_np32(indices.contiguous())), og._p32(_np32(pos.contiguous())),
                                           og._p32(_np32(queue)), B, Q, H, W)
    return queue


def fill_edges_cuda(batch, pos, all_timestamps, queue, indices, K, radius, delta_t_us, edges, min_index):
    B, Q, H, W = queue.shape
    e = edges.numpy()
    og.lib().oracle_fill_edges(og._p32(_np32(batch.contiguous())), og._p32(_np32(pos.contiguous())),
                               og._p32(_np32(all_timestamps)), og._p32(_np32(indices.contiguous())), og._p32(_np32(queue)),
                               e.ctypes.data_as(og._i64p), B, Q, H, W, len(batch), e.shape[1], float(radius),
                               float(delta_t_us), int(K), int(min_index))


# ------------------------------------------------------------------------------------------ install
def install():
    """Register the stand-ins (and placeholders for what is imported but never run) in sys.modules."""
    _module("torch_geometric")
    _module("torch_geometric.data", Data=Data, Batch=Batch)
    _module("torch_geometric.transforms", Cartesian=Cartesian)
    _module("torch_geometric.transforms.to_sparse_tensor", ToSparseTensor=ToSparseTensor)
    _module("torch_geometric.nn", BatchNorm=BatchNorm)
    _module("torch_geometric.nn.conv", SplineConv=SplineConv)
    _module("torch_geometric.nn.pool")
    _module("torch_geometric.nn.pool.avg_pool",
            _avg_pool_x=lambda cluster, x: oo.scatter_mean(x, cluster, int(cluster.max()) + 1))
    _module("torch_geometric.nn.pool.pool",
            pool_pos=lambda cluster, pos: oo.scatter_mean(pos, cluster, int(cluster.max()) + 1))
    _module("torch_scatter",
            scatter_max=lambda src, index, dim=0: (oo.scatter_max(src, index, int(index.max()) + 1), None))
    _module("torch_cluster", grid_cluster=lambda pos, size, start=None, end=None: oo.grid_cluster(pos, size, start, end))
    _module("torch_spline_conv", spline_basis=lambda pseudo, kernel_size, is_open_spline, degree: oo.spline_basis(
        pseudo, int(kernel_size[0]), int(is_open_spline[0]), int(degree)))
    tv = _module("torchvision")
    tv.ops = types.SimpleNamespace(nms=opost.nms)
    _module("torchvision.models", resnet18=_tv_resnet("resnet18"), resnet34=_tv_resnet("resnet34"),
            resnet50=_tv_resnet("resnet50"))
    _module("yolox")
    _module("yolox.models", YOLOX=YOLOX, YOLOXHead=_yolox_head_base(), IOUloss=IOUloss)
    _module("ev_graph_cuda", insert_in_queue_cuda=insert_in_queue_cuda,
            insert_in_queue_single_cuda=insert_in_queue_single_cuda, fill_edges_cuda=fill_edges_cuda)
    for name in ("detectron2", "detectron2.evaluation", "detectron2.evaluation.fast_eval_api", "pycocotools",
                 "pycocotools.coco"):
        _module(name)


def use_reference_package(src="/root/reference/src"):
    """Make ``import dagr.<x>`` resolve to the REFERENCE's sources.  The repository ships its own top-level ``dagr`` package
    (an alias of ``dagr_amd``), which as a regular package would win over the reference's ``src/dagr`` (a namespace
    package: no ``__init__.py``) whatever the order of ``sys.path`` -- the golden generators must not end up comparing the
    mirror with itself."""
    import os
    import types
    for name in [m for m in sys.modules if m == "dagr" or m.startswith("dagr.")]:
        del sys.modules[name]
    pkg = types.ModuleType("dagr")
    pkg.__path__ = [os.path.join(src, "dagr")]
    sys.modules["dagr"] = pkg
    sys.path.insert(0, src)
    return pkg
