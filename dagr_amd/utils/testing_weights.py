"""Seeded, non-degenerate random weights: BN running statistics and affine
parameters are randomised so that BN folding is exercised; everything else keeps its default init."""
import torch


def randomize_(model, seed=0):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, m in model.named_modules():
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
        for name, p in model.named_parameters():
            if name.endswith("bias") and p.dim() == 1 and "module" not in name and "bn" not in name:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    return model
