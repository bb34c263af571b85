"""Mirror of ``src/dagr/model/layers/spline_conv.py`` (MySplineConv :9-78, SplineConvToDense :110-118)
over PyG ``SplineConv``: same parameters/buffers (``weight[25,Cin,Cout]``, ``lin.weight[Cout,Cin]``,
optional ``bias[Cout]``, buffers ``kernel_size`` / ``is_open_spline``) so reference checkpoints load.

The reference's ``init_lut`` expands the 25 kernel taps into a per-integer-offset table
(3.5-16 GB per model); here it only records the offset domain (rx, ry, M): the HIP kernels evaluate
the degree-1 open B-spline basis per edge from the integer offset, which is the same function the
table tabulates (csrc/spline_conv.hip)."""
import math

import torch


class _RootLinear(torch.nn.Module):
    """PyG ``Linear(in, out, bias=False, weight_initializer='uniform')``: key ``lin.weight``."""

    def __init__(self, ic, oc):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.empty(oc, ic))
        bound = 1.0 / math.sqrt(ic)
        torch.nn.init.uniform_(self.weight, -bound, bound)


class MySplineConv(torch.nn.Module):
    def __init__(self, in_channels, out_channels, args, bias=False, degree=1, **kwargs):
        super().__init__()
        assert degree == 1, "only degree-1 B-splines are used by the reference configs"
        self.in_channels, self.out_channels = in_channels, out_channels
        self.dim = args.edge_attr_dim
        self.degree = degree
        assert args.aggr == "sum", "reference configs use aggr: sum (config/*.yaml:17)"
        ks = int(args.kernel_size)
        self.register_buffer("kernel_size", torch.tensor([ks] * self.dim, dtype=torch.long))
        self.register_buffer("is_open_spline", torch.tensor([1] * self.dim, dtype=torch.uint8))
        K = ks ** self.dim
        self.weight = torch.nn.Parameter(torch.empty(K, in_channels, out_channels))
        bound = 1.0 / math.sqrt(in_channels * K)  # PyG SplineConv.reset_parameters: uniform(size=in*K)
        torch.nn.init.uniform_(self.weight, -bound, bound)
        self.lin = _RootLinear(in_channels, out_channels)
        if bias:
            self.bias = torch.nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter("bias", None)
        self.lut_domain = None  # (rx, ry, Mx, My, height, width) once init_lut was called

    def init_lut(self, height, width, rx=None, Mx=None, ry=None, My=None):
        """``spline_conv.py:16-37`` -- records the integer-offset domain only."""
        ry = ry or rx
        My = My or Mx
        # spline_conv.py:23-24 (fp32 matrix built from python doubles or 0-dim fp32 tensors)
        remap = torch.Tensor([[2 * Mx * width, 0, -Mx * width + rx],
                              [0, 2 * My * height, -My * height + ry]])
        # spline_conv.py:29-30: dxy / (2*M*extent) -- the divisor as the fp32 value torch divides by
        den_x = float(torch.as_tensor(2 * Mx * width, dtype=torch.float32))
        den_y = float(torch.as_tensor(2 * My * height, dtype=torch.float32))
        self.lut_domain = dict(rx=int(rx), ry=int(ry), remap=remap, den_x=den_x, den_y=den_y,
                               height=int(height), width=int(width))

    def forward(self, data):
        """``spline_conv.py:49-62``: ``data.x <- conv(data.x)`` over ``data.edge_index`` / ``edge_attr`` (+ root, + bias)."""
        from . import _ops
        data.x = _ops.conv_on_data(self, data)
        return data


class SplineConvToDense(MySplineConv):
    """Conv (with bias) + scatter into a dense [B,C,H,W] map (spline_conv.py:80-118)."""

    def forward(self, data, batch_size=None):
        from . import _ops
        data = MySplineConv.forward(self, data)
        batch = data.batch if getattr(data, "batch", None) is not None else \
            torch.zeros(len(data.x), dtype=torch.long, device=data.x.device)
        if batch_size is None:
            batch_size = getattr(self, "batch_size", None) or (int(batch.max().item()) + 1 if len(batch) else 1)
        self.batch_size = batch_size
        return _ops.to_dense(data.x, data.pos, data.pooling, batch, batch_size)
