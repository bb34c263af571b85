"""Parameter-compatible re-declaration of the parts of ``yolox.models`` (pinned by the reference at
618fd8c0, download_and_install_dependencies.sh:13-15) that DAGR subclasses: ``BaseConv`` and the dense
``YOLOXHead`` module lists.  ``GNNHead`` never runs the dense lists (dagr.py:137) but they are in every
checkpoint, and ``CNNHead`` (dagr.py:106-122) runs them on the image branch.  PyTorch-ROCm only."""
import torch
import torch.nn as nn


class BaseConv(nn.Module):
    """yolox/models/network_blocks.py BaseConv: Conv2d(bias=False) -> BatchNorm2d(eps=1e-3 is NOT set
    there; default eps) -> SiLU."""

    def __init__(self, in_channels, out_channels, ksize, stride, groups=1, bias=False, act="silu"):
        super().__init__()
        pad = (ksize - 1) // 2
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=ksize, stride=stride, padding=pad,
                              groups=groups, bias=bias)
        self.bn = nn.BatchNorm2d(out_channels)
        self.act = nn.SiLU(inplace=True) if act == "silu" else nn.ReLU(inplace=True)

    def forward(self, x):
        return self.act(self.bn(self.conv(x)))


class YOLOXHeadParams(nn.Module):
    """The ModuleLists ``YOLOXHead.__init__`` creates (yolox/models/yolo_head.py): stems, cls_convs,
    reg_convs, cls_preds, reg_preds, obj_preds."""

    def __init__(self, num_classes, width=1.0, strides=(8, 16, 32), in_channels=(256, 512, 1024), act="silu"):
        super().__init__()
        self.n_anchors = 1
        self.num_classes = num_classes
        self.cls_convs = nn.ModuleList()
        self.reg_convs = nn.ModuleList()
        self.cls_preds = nn.ModuleList()
        self.reg_preds = nn.ModuleList()
        self.obj_preds = nn.ModuleList()
        self.stems = nn.ModuleList()
        hidden = int(256 * width)
        for i in range(len(in_channels)):
            self.stems.append(BaseConv(int(in_channels[i] * width), hidden, ksize=1, stride=1, act=act))
            self.cls_convs.append(nn.Sequential(BaseConv(hidden, hidden, 3, 1, act=act),
                                                BaseConv(hidden, hidden, 3, 1, act=act)))
            self.reg_convs.append(nn.Sequential(BaseConv(hidden, hidden, 3, 1, act=act),
                                                BaseConv(hidden, hidden, 3, 1, act=act)))
            self.cls_preds.append(nn.Conv2d(hidden, self.n_anchors * self.num_classes, 1, 1, 0))
            self.reg_preds.append(nn.Conv2d(hidden, 4, 1, 1, 0))
            self.obj_preds.append(nn.Conv2d(hidden, self.n_anchors * 1, 1, 1, 0))
