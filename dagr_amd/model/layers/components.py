"""Parameter-holding mirrors of ``src/dagr/model/layers/components.py`` (BatchNormData :9-12,
Linear :15-22, Cartesian :25-35).  State-dict layout identical to the reference
(``norm.module.{weight,bias,running_mean,running_var,num_batches_tracked}``, ``lin.mlp.weight``)."""
import torch


class BatchNormData(torch.nn.Module):
    """PyG ``BatchNorm`` wraps ``torch.nn.BatchNorm1d`` as ``.module`` (components.py:9-12)."""

    def __init__(self, in_channels):
        super().__init__()
        self.module = torch.nn.BatchNorm1d(in_channels)

    def affine(self):
        """Eval-mode BN as y = x*scale + shift (eps 1e-5)."""
        m = self.module
        scale = m.weight / torch.sqrt(m.running_var + m.eps)
        shift = m.bias - m.running_mean * scale
        return scale, shift


class Linear(torch.nn.Module):
    def __init__(self, ic, oc, bias=True):
        super().__init__()
        self.mlp = torch.nn.Linear(ic, oc, bias=bias)


class Cartesian(torch.nn.Module):
    """``T.Cartesian(norm=True, cat=False, max_value=M)`` holder: no parameters; the engine computes
    the (integer) edge offsets in the graph/pooling kernels."""

    def __init__(self, norm=True, cat=False, max_value=None):
        super().__init__()
        self.norm, self.cat, self.max = norm, cat, max_value

    def forward(self, data):                                    # components.py:30-35
        from . import _ops
        if getattr(data, "is_lazy", None) is not None and data.is_lazy("edge_index"):
            # the graph is held as CSR + integer pixel offsets (EV_TGN's training graph): the attribute tensor is a recipe,
            # like edge_index itself -- the training-mode convs read the offsets, not the attributes
            data.set_lazy("edge_attr", lambda d, m=self.max: _ops.cartesian(d.pos, d.edge_index, m))
        else:
            data.edge_attr = _ops.cartesian(data.pos, data.edge_index, self.max)
        data.edge_attr_max = self.max       # read by the training-mode conv to recover exact pixel offsets
        return data
