"""Differentiable SplineConv (training path, first slice: SURVEY.md section 8f rank 4).

``MySplineConv`` in the reference trains through PyG's ``propagate`` + torch_spline_conv's autograd kernels
(``spline_conv.py:64-78``).  Here the op is ``out = A(x) . Wm + bias`` with ``A`` the tap aggregation of
``dagr_spline_tap_aggregate`` (linear in x) and ``Wm = [W[25, cin, cout] flattened | root^T]``:
  * forward: tap aggregation (HIP) + one GEMM (hipBLASLt through torch);
  * backward: ``grad_Wm = A^T . g`` and ``grad_A = g . Wm^T`` are plain library GEMMs, ``grad_x`` is the transposed
    aggregation (``dagr_spline_tap_scatter_grad``: deterministic fixed-point scatter), ``grad_bias = sum g``.
``PoolFeatFn`` / ``ToDenseFn`` attach the backward of the voxel pooling's feature aggregation (torch_scatter's
``scatter_max`` / ``scatter_mean`` autograd, pooling.py:74-77) and of ``to_dense`` (an ``index_put``, spline_conv.py:80-107)
to the forward entry points the eval path already uses; both backwards are gathers (csrc/train_ops.hip).  BatchNorm in
training mode is ``torch.nn.BatchNorm1d`` (the reference's BatchNormData wraps the same module), the loss is
``networks/yolox_loss.py``."""
import torch

from ... import _lib


def _at_g(A, g, K=None):
    """A[:, :K]^T . g for a tall A [n, lda >= K] and g [n, cout] (the weight gradient of a SplineConv).  The library picks a kernel without
    split-K for this shape -- 0.6 - 0.9 ms at n = 400 k, a seventh of HBM peak -- so the long dimension is cut into a batch
    of P blocks (one batched product, then a fixed-order sum of the P partial results): 0.05 - 0.15 ms
    (tools/skinny_gemm_bench.py).  Deterministic; only the summation order differs from the plain product."""
    n, lda = A.shape
    K = lda if K is None else K
    P = 64 if n >= 131072 else (16 if n >= 8192 else 1)
    if P > 1 and K < 128:
        # a narrow A (the network's first conv: K = 26) gives the batched kernel one 16-row tile per batch -- 64 workgroups,
        # 0.94 ms for 42 MB (rocprofv3, profiles/r4_train_step_kernel_stats.md); more, shorter batches fill the chip
        P = min(1024, P * (128 // max(16, (K + 15) // 16 * 16)) * 2)
    m = n // P * P
    if P == 1 or m == 0:
        return A[:, :K].t() @ g
    r = torch.bmm(A[:m].view(P, m // P, lda)[:, :, :K].transpose(1, 2), g[:m].view(P, m // P, g.shape[1])).sum(0)
    if m < n:
        r = r + A[m:, :K].t() @ g[m:]
    return r


def _g_wt(g, WmT):
    """g . Wm^T for a tall g [n, cout] (the gradient w.r.t. the aggregated rows): row blocks as a batch -- the batched kernel
    the library picks is up to 2.6 x faster on these shapes than the one it picks for the flat product."""
    n, cout = g.shape
    K = WmT.shape[1]
    P = 64
    m = n // P * P
    if n < 4096 or m == 0:
        return g @ WmT
    out = torch.empty((n, K), dtype=g.dtype, device=g.device)
    torch.matmul(g[:m].view(P, m // P, cout), WmT, out=out[:m].view(P, m // P, K))
    if m < n:
        torch.mm(g[m:], WmT, out=out[m:])
    return out


class TallLinearFn(torch.autograd.Function):
    """y = x . W^T for a tall x [n, cin] (the skip Linear of the event level: 400 k rows, 3 -> 16).  torch's backward forms the
    weight gradient g^T x as ONE product reduced over n with no split -- 0.94 ms at n = 400 k, the largest kernel of a
    training step (rocprofv3, profiles/r4_train_step_kernel_stats.md); ``_at_g`` cuts n into a batch of partial products."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return x @ weight.t()

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        gx = g @ weight if ctx.needs_input_grad[0] else None
        gw = _at_g(g, x.contiguous()) if ctx.needs_input_grad[1] else None       # [cout, cin] = g^T . x
        return gx, gw


def tall_linear(mlp, x):
    """``mlp(x)`` (a bias-free ``torch.nn.Linear``) with the weight gradient of ``TallLinearFn`` on tall inputs."""
    if mlp.bias is None and x.shape[0] >= 131072 and x.is_cuda and torch.is_grad_enabled():
        return TallLinearFn.apply(x, mlp.weight)
    return mlp(x)


class SplineConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, root, bias, rowptr, col, code, rx, ry, den_x, den_y):
        L, P = _lib.lib(), _lib.ptr
        n, cin = x.shape
        cout = weight.shape[2]
        K = 26 * cin
        # the event level's products go through the library's own skinny GEMM, which wants 16-byte aligned rows (the pad
        # columns are never read); elsewhere dense rows: library GEMMs read the matrix and the aggregation writes every entry
        own_gemm = n >= 65536 and cout in (8, 16)
        lda = (K + 3) // 4 * 4 if own_gemm else K
        x = x.float().contiguous()
        A = torch.empty((n, lda), dtype=torch.float32, device=x.device)
        if n:
            _lib.check(L.dagr_spline_tap_aggregate(None, n, P(rowptr), P(col), P(code), P(x), cin, cin, None, 0, 0, rx,
                                                   ry, den_x, den_y, P(A), lda, _lib.cur_stream(x.device)), "tap_aggregate")
        Wm = torch.cat([weight.reshape(25 * cin, cout), root.t()], 0)
        if own_gemm:
            # the event level: a [400 k, 416] . [416, 16] product for which the library picks a kernel at a seventh of HBM
            # speed (0.94 ms); the library's own skinny GEMM streams A once (bias in its epilogue)
            out = torch.empty((n, cout), dtype=torch.float32, device=x.device)
            _lib.check(L.dagr_gemm_bias_act(None, n, P(A), lda, P(Wm), cout, P(bias.float().contiguous()) if bias is not None
                                            else None, P(out), cout, K, cout, 0, _lib.cur_stream(x.device)), "gemm")
        elif bias is not None:
            out = torch.addmm(bias, A[:, :K], Wm)
        else:
            out = A[:, :K] @ Wm
        ctx.save_for_backward(A, Wm, rowptr, col, code)
        ctx.dom = (rx, ry, den_x, den_y)
        ctx.shape = (n, cin, cout, K, lda, bias is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        L, P = _lib.lib(), _lib.ptr
        A, Wm, rowptr, col, code = ctx.saved_tensors
        n, cin, cout, K, lda, has_bias = ctx.shape
        rx, ry, den_x, den_y = ctx.dom
        g = g.float().contiguous()
        gWm = _at_g(A, g, K)
        gW = gWm[:25 * cin].reshape(25, cin, cout)
        groot = gWm[25 * cin:].t().contiguous()
        # (the first conv of the network consumes the events' own features: no input gradient is asked for)
        gx = torch.zeros((n, cin), dtype=torch.float32, device=g.device) if ctx.needs_input_grad[0] else None
        if n and cin <= 16 and cout <= 16 and ctx.needs_input_grad[0]:
            # narrow convs (the event level): the node's row of gA = g . Wm^T is rebuilt inside the scatter kernel -- no
            # [n, 26 cin] matrix (0.67 GB for 16 -> 16 at 400 k rows), no reduction over it; the fixed-point scale is the
            # bound max|g| * max_k sum_co |Wm[k, co]| >= max|gA| (device scalars, no host sync)
            bound = (torch.linalg.vector_norm(g, ord=float("inf")) * Wm.abs().sum(1).amax()).reshape(1).contiguous()
            acc = torch.zeros((n, cin), dtype=torch.int64, device=g.device)
            _lib.check(L.dagr_spline_tap_scatter_grad_w(None, n, P(rowptr), P(col), P(code), P(g), cout, cout, P(Wm), cout,
                                                        cin, rx, ry, den_x, den_y, P(bound), P(acc), P(gx), cin,
                                                        _lib.cur_stream(g.device)), "tap_scatter_grad_w")
        elif n and ctx.needs_input_grad[0]:
            gA = _g_wt(g, Wm.t().contiguous())                # [n, K]: no zero fill, no copy
            # deterministic scatter: 64-bit fixed-point sums scaled by max |gA| (a device scalar, no host sync; one
            # reduction pass, no |gA| temporary)
            amax = torch.linalg.vector_norm(gA, ord=float("inf")).reshape(1).contiguous()
            acc = torch.zeros((n, cin), dtype=torch.int64, device=g.device)
            _lib.check(L.dagr_spline_tap_scatter_grad(None, n, P(rowptr), P(col), P(code), P(gA), gA.shape[1], cin, rx, ry,
                                                      den_x, den_y, P(amax), P(acc), P(gx), cin,
                                                      _lib.cur_stream(g.device)), "tap_scatter_grad")
        gb = g.sum(0) if has_bias else None
        return gx, gW, groot, gb, None, None, None, None, None, None, None


def spline_conv_autograd(conv, x, rowptr, col, code):
    """``MySplineConv._forward`` with gradients w.r.t. x, weight[25], root weight and bias."""
    d = conv.lut_domain
    if d is None:
        raise RuntimeError("call init_lut() / DAGR.cache_luts() first")
    return SplineConvFn.apply(x, conv.weight, conv.lin.weight, conv.bias, rowptr, col, code, d["rx"], d["ry"], d["den_x"],
                              d["den_y"])


class PoolFeatFn(torch.autograd.Function):
    """x -> pooled x of ``dagr_pool_csr`` (already computed: ``holder.pooled``) with the gradient the reference gets from
    torch_scatter: max routes each (cluster, channel) gradient to the first member holding the maximum, mean spreads
    it evenly.  ``cluster`` int32[n]: consecutive cluster id per node, -1 for nodes outside the grid."""

    @staticmethod
    def forward(ctx, x, cluster, aggr, holder):
        L, P = _lib.lib(), _lib.ptr
        pooled = holder.pooled
        n, C = x.shape
        nc = pooled.shape[0]
        stream = _lib.cur_stream(x.device)
        arg = count = None
        if aggr == 0:
            arg = torch.empty((max(nc, 1), C), dtype=torch.int32, device=x.device)
            xc = x.detach().float().contiguous()
            _lib.check(L.dagr_pool_argmax(P(cluster), n, P(xc), C, C, P(pooled), pooled.stride(0), nc, P(arg), stream),
                       "pool_argmax")
        else:
            # members per cluster; nodes outside the grid (-1) go to a spare slot (no mask indexing: no host sync)
            count = torch.zeros((max(nc, 1) + 1,), dtype=torch.int32, device=x.device)
            count.scatter_add_(0, torch.where(cluster >= 0, cluster, torch.full_like(cluster, max(nc, 1))).long(),
                               torch.ones_like(cluster))
            count = count[:max(nc, 1)].contiguous()
        ctx.save_for_backward(cluster, arg if arg is not None else count)
        ctx.meta = (n, C, aggr)
        return pooled

    @staticmethod
    def backward(ctx, g):
        L, P = _lib.lib(), _lib.ptr
        cluster, aux = ctx.saved_tensors
        n, C, aggr = ctx.meta
        g = g.float().contiguous()
        gx = torch.empty((n, C), dtype=torch.float32, device=g.device)
        if n:
            _lib.check(L.dagr_pool_grad(P(cluster), n, C, aggr, P(aux) if aggr == 0 else None,
                                        P(aux) if aggr == 1 else None, P(g), g.stride(0) if g.shape[0] else C, P(gx), C,
                                        _lib.cur_stream(g.device)), "pool_grad")
        return gx, None, None, None


class ToDenseFn(torch.autograd.Function):
    """``to_dense`` (spline_conv.py:80-107): rows scattered into a zeroed [B, C, Hc, Wc] map; backward gathers the map's
    gradient at the cell of every node -- overwritten duplicates included, as torch's index_put backward does in the
    reference (pinned by the reference's own training branch: tests/golden/ref_py_model.npz)."""

    @staticmethod
    def forward(ctx, x, pos, batch, vx, vy, batch_size, Hc, Wc):
        L, P = _lib.lib(), _lib.ptr
        dev = x.device
        n, C = x.shape
        dense = torch.zeros((batch_size, C, Hc, Wc), dtype=torch.float32, device=dev)
        winner = torch.full((batch_size * Hc * Wc,), -1, dtype=torch.int32, device=dev)
        pos = pos.float().contiguous()
        batch = batch.int().contiguous()
        if n:
            status = torch.zeros((1,), dtype=torch.int32, device=dev)
            _lib.check(L.dagr_to_dense(None, n, P(x.detach().float().contiguous()), C, C, P(pos), P(batch), vx, vy,
                                       batch_size, Hc, Wc, P(winner), P(dense), P(status), _lib.cur_stream(dev)),
                       "to_dense")
        ctx.save_for_backward(pos, batch)
        ctx.meta = (n, C, vx, vy, batch_size, Hc, Wc)
        return dense

    @staticmethod
    def backward(ctx, g):
        L, P = _lib.lib(), _lib.ptr
        pos, batch = ctx.saved_tensors
        n, C, vx, vy, B, Hc, Wc = ctx.meta
        g = g.float().contiguous()
        gx = torch.empty((n, C), dtype=torch.float32, device=g.device)
        if n:
            _lib.check(L.dagr_to_dense_grad(n, C, P(pos), P(batch), vx, vy, B, Hc, Wc, P(g), P(gx), C,
                                            _lib.cur_stream(g.device)), "to_dense_grad")
        return gx, None, None, None, None, None, None, None
