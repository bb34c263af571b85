"""Parameter holders with the reference's names for ``src/dagr/model/layers/conv.py``: ``ConvBlock`` (:10-28:
``conv``, ``norm``), ``ConvBlockWithSkip`` (:31-56: + ``lin``, ``norm_skip``), ``Layer`` (:59-72: ``conv_block1``,
``conv_block2``).  They do not execute anything themselves: the engine packs conv + BN(eval) + ReLU
(+ skip Linear + BN) into one fused contraction per block (``dagr_amd/engine.py``)."""
import torch

from .components import BatchNormData, Linear
from .spline_conv import MySplineConv


def _require_relu(args):
    if args.activation != "relu":
        raise NotImplementedError("the fused epilogues implement activation: relu (config/*.yaml:15)")


class ConvBlock(torch.nn.Module):
    def __init__(self, in_channels, out_channels, args, degree=1):
        super().__init__()
        _require_relu(args)
        self.conv = MySplineConv(in_channels, out_channels, args=args, bias=False, degree=degree)
        self.norm = BatchNormData(in_channels=out_channels)


class ConvBlockWithSkip(ConvBlock):
    """relu(norm(conv(h)) + norm_skip(lin(x_in)))"""

    def __init__(self, in_channel, out_channel, skip_in_channel, args):
        super().__init__(in_channel, out_channel, args)
        self.lin = Linear(skip_in_channel, out_channel, bias=False)
        self.norm_skip = BatchNormData(in_channels=out_channel)


class Layer(torch.nn.Module):
    def __init__(self, in_channels, out_channels, args):
        super().__init__()
        self.in_channel, self.out_channel = in_channels, out_channels
        self.conv_block1 = ConvBlock(in_channels, out_channels, args)
        self.conv_block2 = ConvBlockWithSkip(out_channels, out_channels, in_channels, args=args)
