"""Differentiable SplineConv (training path, first slice: SURVEY.md section 8f rank 4).

``MySplineConv`` in the reference trains through PyG's ``propagate`` + torch_spline_conv's autograd kernels
(``spline_conv.py:64-78``).  Here the op is ``out = A(x) . Wm + bias`` with ``A`` the tap aggregation of
``dagr_spline_tap_aggregate`` (linear in x) and ``Wm = [W[25, cin, cout] flattened | root^T]``:
  * forward: tap aggregation (HIP) + one GEMM (hipBLASLt through torch);
  * backward: ``grad_Wm = A^T . g`` and ``grad_A = g . Wm^T`` are plain library GEMMs, ``grad_x`` is the transposed
    aggregation (``dagr_spline_tap_scatter_grad``), ``grad_bias = sum g``.
BatchNorm in training mode, pooling backward and the YOLOX loss are not built yet."""
import torch

from ... import _lib


class SplineConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, root, bias, rowptr, col, code, rx, ry, den_x, den_y):
        L, P = _lib.lib(), _lib.ptr
        n, cin = x.shape
        cout = weight.shape[2]
        K = 26 * cin
        lda = (K + 3) // 4 * 4
        x = x.float().contiguous()
        A = torch.zeros((n, lda), dtype=torch.float32, device=x.device)
        counts = torch.tensor([n, col.shape[0]], dtype=torch.int32, device=x.device)
        if n:
            _lib.check(L.dagr_spline_tap_aggregate(P(counts), n, P(rowptr), P(col), P(code), P(x), cin, cin, None, 0, 0, rx,
                                                   ry, den_x, den_y, P(A), lda, _lib.cur_stream(x.device)), "tap_aggregate")
        Wm = torch.cat([weight.reshape(25 * cin, cout), root.t()], 0)
        out = A[:, :K] @ Wm
        if bias is not None:
            out = out + bias
        ctx.save_for_backward(A, Wm, rowptr, col, code, counts)
        ctx.dom = (rx, ry, den_x, den_y)
        ctx.shape = (n, cin, cout, K, lda, bias is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        L, P = _lib.lib(), _lib.ptr
        A, Wm, rowptr, col, code, counts = ctx.saved_tensors
        n, cin, cout, K, lda, has_bias = ctx.shape
        rx, ry, den_x, den_y = ctx.dom
        g = g.float().contiguous()
        gWm = A[:, :K].t() @ g
        gW = gWm[:25 * cin].reshape(25, cin, cout)
        groot = gWm[25 * cin:].t().contiguous()
        gA = torch.zeros((n, lda), dtype=torch.float32, device=g.device)
        gA[:, :K] = g @ Wm.t()
        gx = torch.zeros((n, cin), dtype=torch.float32, device=g.device)
        if n:
            _lib.check(L.dagr_spline_tap_scatter_grad(P(counts), n, P(rowptr), P(col), P(code), P(gA), lda, cin, rx, ry,
                                                      den_x, den_y, P(gx), cin, _lib.cur_stream(g.device)),
                       "tap_scatter_grad")
        gb = g.sum(0) if has_bias else None
        return gx, gW, groot, gb, None, None, None, None, None, None, None


def spline_conv_autograd(conv, x, rowptr, col, code):
    """``MySplineConv._forward`` with gradients w.r.t. x, weight[25], root weight and bias."""
    d = conv.lut_domain
    if d is None:
        raise RuntimeError("call init_lut() / DAGR.cache_luts() first")
    return SplineConvFn.apply(x, conv.weight, conv.lin.weight, conv.bias, rowptr, col, code, d["rx"], d["ry"], d["den_x"],
                              d["den_y"])
