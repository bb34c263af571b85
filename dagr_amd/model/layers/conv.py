"""Mirror of ``src/dagr/model/layers/conv.py``: ``ConvBlock`` (:10-28: ``conv``, ``norm``), ``ConvBlockWithSkip``
(:31-56: + ``lin``, ``norm_skip``), ``Layer`` (:59-72: ``conv_block1``, ``conv_block2``), same names and state_dict.
Whole windows go through ``dagr_amd/engine.py``; each module is ALSO a ``Data -> Data`` callable like its reference
twin: in eval mode ONE fused contraction -- conv + BN(eval) + ReLU (+ skip Linear + BN) -- by the kernels behind
``dagr_spline_conv_fused`` (``_ops.py``); in training mode the reference's op sequence (conv.py:23-28,47-56) with the
differentiable SplineConv of ``autograd.py`` and batch-statistics BatchNorm (``torch.nn.BatchNorm1d``, which is what the
reference's BatchNormData wraps)."""
import torch

from . import _ops
from .components import BatchNormData, Linear
from .spline_conv import MySplineConv
from ..utils import shallow_copy


def _bn(norm, x):
    """BatchNormData.forward (components.py:9-12) on node features; a batch of < 2 rows has no batch statistics (torch
    raises): such levels pass through the running statistics like an eval call."""
    m = norm.module
    if m.training and x.shape[0] < 2:
        return torch.nn.functional.batch_norm(x, m.running_mean, m.running_var, m.weight, m.bias, False, 0.0, m.eps)
    return m(x)


def _require_relu(args):
    if args.activation != "relu":
        raise NotImplementedError("the fused epilogues implement activation: relu (config/*.yaml:15)")


class ConvBlock(torch.nn.Module):
    def __init__(self, in_channels, out_channels, args, degree=1):
        super().__init__()
        _require_relu(args)
        self.conv = MySplineConv(in_channels, out_channels, args=args, bias=False, degree=degree)
        self.norm = BatchNormData(in_channels=out_channels)

    def forward(self, data):                                    # conv.py:23-28
        if self.training:
            data = self.conv(data)
            data.x = torch.relu(_bn(self.norm, data.x))
            return data
        data.x = _ops.conv_on_data(self.conv, data, norm=self.norm, relu=True)
        return data


class ConvBlockWithSkip(ConvBlock):
    """relu(norm(conv(h)) + norm_skip(lin(x_in)))"""

    def __init__(self, in_channel, out_channel, skip_in_channel, args):
        super().__init__(in_channel, out_channel, args)
        self.lin = Linear(skip_in_channel, out_channel, bias=False)
        self.norm_skip = BatchNormData(in_channels=out_channel)

    def forward(self, data, data_skip):                         # conv.py:47-56
        if self.training:
            data = self.conv(data)
            from .autograd import tall_linear
            skip = _bn(self.norm_skip, tall_linear(self.lin.mlp, data_skip.x))
            data.x = torch.relu(_bn(self.norm, data.x) + skip)
            return data
        data.x = _ops.conv_on_data(self.conv, data, norm=self.norm, skip=(self.lin, self.norm_skip), xskip=data_skip.x,
                                   relu=True)
        return data


class Layer(torch.nn.Module):
    def __init__(self, in_channels, out_channels, args):
        super().__init__()
        self.in_channel, self.out_channel = in_channels, out_channels
        self.conv_block1 = ConvBlock(in_channels, out_channels, args)
        self.conv_block2 = ConvBlockWithSkip(out_channels, out_channels, in_channels, args=args)

    def forward(self, data):                                    # conv.py:68-72
        data_skip = shallow_copy(data)
        data = self.conv_block1(data)
        return self.conv_block2(data, data_skip)
