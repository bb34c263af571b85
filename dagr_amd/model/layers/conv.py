"""Mirrors of ``src/dagr/model/layers/conv.py``: ConvBlock :10-28, ConvBlockWithSkip :31-56,
Layer :59-72 (parameter layout identical; execution is fused in the engine:
conv + BN(eval) + ReLU (+ skip Linear + BN_skip) in one kernel sequence)."""
import torch

from .components import BatchNormData, Linear
from .spline_conv import MySplineConv


class ConvBlock(torch.nn.Module):
    def __init__(self, in_channels, out_channels, args, degree=1):
        super().__init__()
        assert args.activation == "relu", "reference configs use activation: relu (config/*.yaml:15)"
        self.conv = MySplineConv(in_channels, out_channels, args=args, bias=False, degree=degree)
        self.norm = BatchNormData(in_channels=out_channels)


class ConvBlockWithSkip(torch.nn.Module):
    def __init__(self, in_channel, out_channel, skip_in_channel, args):
        super().__init__()
        assert args.activation == "relu"
        self.conv = MySplineConv(in_channel, out_channel, args=args, bias=False)
        self.norm = BatchNormData(in_channels=out_channel)
        self.lin = Linear(skip_in_channel, out_channel, bias=False)
        self.norm_skip = BatchNormData(in_channels=out_channel)


class Layer(torch.nn.Module):
    def __init__(self, in_channels, out_channels, args):
        super().__init__()
        self.in_channel, self.out_channel = in_channels, out_channels
        self.conv_block1 = ConvBlock(in_channels, out_channels, args)
        self.conv_block2 = ConvBlockWithSkip(out_channels, out_channels, in_channels, args=args)
