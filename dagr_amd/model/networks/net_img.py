"""Image branch (PyTorch-ROCm, as north_star prescribes): mirror of
``src/dagr/model/networks/net_img.py`` (``HookModule`` :42-134) plus a parameter-compatible
re-declaration of torchvision's ResNet-18/34/50 (torchvision is not part of this stack; module names
``conv1, bn1, layer1..4, fc`` and block layouts follow torchvision 0.12 so ``backbone.net.module.*``
checkpoint keys load).  Pretrained weights cannot be fetched offline: random init."""
import torch
import torch.nn as nn


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + identity)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)  # stride on the 3x3 (torchvision v1.5)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + identity)


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                       nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def forward_features(self, x):
        """conv1 (raw, pre-BN: the reference hooks the ``conv1`` module itself, net.py:47), layer1..4."""
        c1 = self.conv1(x)
        x = self.maxpool(self.relu(self.bn1(c1)))
        l1 = self.layer1(x)
        l2 = self.layer2(l1)
        l3 = self.layer3(l2)
        l4 = self.layer4(l3)
        return dict(conv1=c1, layer1=l1, layer2=l2, layer3=l3, layer4=l4)


def make_img_net(name):
    cfg = dict(resnet18=(BasicBlock, [2, 2, 2, 2]), resnet34=(BasicBlock, [3, 4, 6, 3]),
               resnet50=(Bottleneck, [3, 4, 6, 3]))
    if name not in cfg:
        raise ValueError(f"img_net must be one of {sorted(cfg)} (net.py:12,42), got {name!r}")
    return ResNet(*cfg[name])


class HookModule(nn.Module):
    """``net_img.py:42-134``: collects the outputs of ``feature_layers`` / ``output_layers`` and maps
    them through 1x1 "dconv"s.  The reference gathers them with forward hooks while running the whole
    classifier (avgpool + fc, result discarded); here the trunk returns them directly -- same tensors."""

    def __init__(self, module, height, width, input_channels=3, feature_layers=(), output_layers=(),
                 feature_channels=None, output_channels=None):
        super().__init__()
        assert input_channels == 3
        self.module = module
        self.feature_layers = list(feature_layers)
        self.output_layers = list(output_layers)
        with torch.no_grad():
            was_training = self.module.training
            self.module.eval()
            d = self.module.forward_features(torch.zeros((1, input_channels, height, width)))
            self.module.train(was_training)
        self.feature_channels = [d[l].shape[1] for l in self.feature_layers]
        self.output_channels = [d[l].shape[1] for l in self.output_layers]
        self.feature_dconv = nn.ModuleList()
        if feature_channels is not None:
            assert len(feature_channels) == len(self.feature_channels)
            self.feature_dconv = nn.ModuleList([nn.Conv2d(cin, cout, 1, 1, 0)
                                                for cin, cout in zip(self.feature_channels, feature_channels)])
            self.feature_channels = list(feature_channels)
        self.output_dconv = nn.ModuleList()
        if output_channels is not None:
            assert len(output_channels) == len(self.output_channels)
            self.output_dconv = nn.ModuleList([nn.Conv2d(cin, cout, 1, 1, 0)
                                               for cin, cout in zip(self.output_channels, output_channels)])
            self.output_channels = list(output_channels)

    def remove_hooks(self):
        pass

    def register_hooks(self):
        pass

    def forward(self, x):
        d = self.module.forward_features(x)
        features = [d[l] for l in self.feature_layers]
        if len(self.feature_dconv) > 0:
            features = [dconv(f) for f, dconv in zip(features, self.feature_dconv)]
        outputs = [d[l] for l in self.output_layers]
        if len(self.output_dconv) > 0:
            outputs = [dconv(o) for o, dconv in zip(outputs, self.output_dconv)]
        return features, outputs
