"""SplineConv backward (training path, first slice): gradients of ``MySplineConv.forward`` w.r.t. x, weight[25, cin, cout],
the root weight and the bias -- tap aggregation + its transpose in HIP, the two weight-side contractions as library
GEMMs -- against autograd through a float64 evaluation of the op built on the oracle's torch_spline_conv basis."""
import numpy as np
import pytest
import torch

from oracle import model as om
from oracle import ops as oo
from dagr_amd.data import Data

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["train_exact_offsets", "eval_lut_domain"])
@pytest.mark.parametrize("n,cin,cout,max_deg", [(300, 8, 6, 9), (1000, 66, 64, 12), (40, 3, 16, 30),
                                                 (66000, 16, 16, 4)])     # event-level shape: skinny-GEMM forward, gA rebuilt in the scatter
def test_spline_conv_gradients_match_float64_autograd(n, cin, cout, max_deg, mode):
    from dagr_amd.model.layers.spline_conv import MySplineConv
    rng = np.random.default_rng(n + cin)
    args = om.default_args()
    W_, H_ = 320, 215
    torch.manual_seed(n)
    conv = MySplineConv(cin, cout, args=args, bias=True)
    with torch.no_grad():
        conv.bias.uniform_(-0.5, 0.5)
    conv = conv.cuda()
    rx, ry, M = 12, 11, 0.0625
    conv.init_lut(height=H_, width=W_, Mx=M, rx=rx, ry=ry)
    # a graph whose Cartesian attributes sit on the integer offset grid the table covers
    deg = rng.integers(0, max_deg + 1, size=n)
    dst = np.repeat(np.arange(n), deg)
    src = rng.integers(0, n, size=len(dst))
    dx, dy = rng.integers(-rx, rx + 1, len(dst)), rng.integers(-ry, ry + 1, len(dst))
    attr = np.stack([dx / (2 * M * W_) + 0.5, dy / (2 * M * H_) + 0.5], 1).astype(np.float32)
    x = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).cuda().requires_grad_(True)
    data = Data(x=x, edge_index=torch.from_numpy(np.stack([src, dst])).cuda(), edge_attr=torch.from_numpy(attr).cuda())
    if mode == "eval_lut_domain":
        conv.eval()                  # codes = message_lut's table coordinates (spline_conv.py:41-42)
    else:
        data.edge_attr_max = M       # training mode: exact offsets recovered from the attributes (Cartesian sets this)
    out = conv(data).x
    g = torch.from_numpy(rng.standard_normal((n, cout)).astype(np.float32)).cuda()
    out.backward(g)
    # float64 reference: out[i] = sum_e sum_s basis_s x[src] W[idx_s] + x root^T + b, gradients by autograd
    xd = x.detach().double().cpu().requires_grad_(True)
    Wd = conv.weight.detach().double().cpu().requires_grad_(True)
    Rd = conv.lin.weight.detach().double().cpu().requires_grad_(True)
    bd = conv.bias.detach().double().cpu().requires_grad_(True)
    basis, index = oo.spline_basis(torch.from_numpy(attr).double())
    msg = torch.zeros((len(dst), cout), dtype=torch.float64)
    xs = xd[torch.from_numpy(src)]
    for s in range(4):
        msg = msg + basis[:, s:s + 1] * torch.einsum("ei,eio->eo", xs, Wd[index[:, s]])
    ref = torch.zeros((n, cout), dtype=torch.float64).index_add(0, torch.from_numpy(dst), msg) + xd @ Rd.t() + bd
    ref.backward(g.double().cpu())
    scale = lambda t: max(1.0, float(t.abs().max()))
    assert float((out.detach().cpu().double() - ref.detach()).abs().max()) <= 1e-4 * scale(ref)
    for name, got, want in (("x", x.grad, xd.grad), ("weight", conv.weight.grad, Wd.grad),
                            ("root", conv.lin.weight.grad, Rd.grad), ("bias", conv.bias.grad, bd.grad)):
        assert got is not None, name
        err = float((got.cpu().double() - want).abs().max()) / scale(want)
        assert err <= 1e-4, (name, err)
