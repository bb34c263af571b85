"""Module-level operators over libdagr_hip for graphs in the reference's representation (PyG ``Data``: ``x``, ``pos``,
``batch``, ``edge_index int64[2,E]`` = (source, destination), ``edge_attr`` = T.Cartesian values).

``DAGR.forward`` runs whole windows through ``dagr_amd.engine`` (device-resident levels, no host synchronisation).  The
functions here back the ``forward`` of the individual layer modules, so that each of them is a ``Data -> Data`` callable
like its reference twin (``model/layers/*.py``) and a maintainer can swap one layer at a time: the graph is brought into
the kernels' CSR-by-destination form with a few torch ops, and every contraction / pooling / scatter is the C entry
point the engine uses (include/dagr_hip.h)."""
import ctypes

import torch

from ... import _lib


def csr_by_destination(edge_index, num_nodes):
    """(rowptr int32[n+1], col int32[E], perm int64[E]): edges sorted by (destination, source) -- the order
    ``ToSparseTensor`` establishes (spline_conv.py:12,52-54)."""
    src, dst = edge_index[0].long(), edge_index[1].long()
    perm = torch.argsort(dst * num_nodes + src, stable=True)
    counts = torch.bincount(dst, minlength=num_nodes)
    rowptr = torch.zeros(num_nodes + 1, dtype=torch.int32, device=edge_index.device)
    rowptr[1:] = torch.cumsum(counts, 0).int()
    return rowptr, src[perm].int().contiguous(), perm


def lut_codes(edge_attr, domain):
    """``message_lut``'s integer table coordinates (spline_conv.py:41-42) packed as ix | iy << 16."""
    rm = domain["remap"].to(edge_attr.device)
    ix = (edge_attr[:, 0] * rm[0, 0] + rm[0, 2] + 1e-3).long()
    iy = (edge_attr[:, 1] * rm[1, 1] + rm[1, 2] + 1e-3).long()
    return (ix | (iy << 16)).int().contiguous()


def graph_csr(data):
    """CSR of ``data``'s graph, cached on the object like the reference caches ``adj_t`` (spline_conv.py:51-54)."""
    cached = getattr(data, "_dagr_csr", None)
    n = data.x.shape[0]
    if cached is not None and cached[3] == ("csr", n):
        # the graph was produced as CSR (graph builder / pooling kernel); ``edge_index`` is derived from it on demand
        return cached[:3]
    if cached is None or cached[3] != (n, data.edge_index.data_ptr(), data.edge_index.shape[1]):
        rowptr, col, perm = csr_by_destination(data.edge_index, n)
        cached = (rowptr, col, perm, (n, data.edge_index.data_ptr(), data.edge_index.shape[1]))
        data._dagr_csr = cached
    return cached[:3]


def edge_index_from_csr(data):
    """Recipe of ``edge_index`` (int64[2, E], destinations ascending, a destination's sources in the builder's order) for
    a level-0 graph held as CSR by destination (EV_TGN's training graph); synchronises to learn E."""
    rowptr, col, _, _ = data._dagr_csr
    n = rowptr.shape[0] - 1
    E = int(rowptr[-1])
    dst = torch.repeat_interleave(torch.arange(n, device=col.device), (rowptr[1:] - rowptr[:-1]).long(), output_size=E)
    return torch.stack([col[:E].long(), dst])


def pooled_edge_index(data):
    """Recipe of a pooled level's ``edge_index`` in the reference's order (``edge_index.unique(dim=-1)``: by source, then
    destination, pooling.py:66-68) from the pooling kernel's CSR; the CSR's edge permutation rides along for consumers of
    ``edge_attr`` (which follows ``edge_index``'s order)."""
    rowptr, col, _, tag = data._dagr_csr
    nc, ne = rowptr.shape[0] - 1, col.shape[0]
    dst = torch.repeat_interleave(torch.arange(nc, device=col.device), (rowptr[1:] - rowptr[:-1]).long(), output_size=ne)
    src = col.long()
    order = torch.argsort(src * max(nc, 1) + dst, stable=True)
    inv = torch.empty_like(order)
    inv[order] = torch.arange(ne, device=col.device)
    data._dagr_csr = (rowptr, col, inv, tag)        # perm: position of a CSR edge in edge_index
    return torch.stack([src[order], dst[order]])


def _in_csr_order(edge_attr, perm):
    """Edge attributes in the CSR's edge order (perm None: the graph was built in that order)."""
    return edge_attr if perm is None else edge_attr[perm]


def spline_conv(conv, x, rowptr, col, code, norm=None, skip=None, xskip=None, relu=False):
    """out = act(BN(sum_j x_j . What(code_j) + x . root (+ bias)) (+ BN_skip(xskip . Wskip))) on a CSR graph."""
    from ...engine import _pack_generic
    if conv.lut_domain is None:
        raise RuntimeError("call init_lut() / DAGR.cache_luts() first: the kernels evaluate the spline basis on the integer "
                           "offset domain the reference tabulates (spline_conv.py:16-37)")
    L, P = _lib.lib(), _lib.ptr
    dev = x.device
    n = x.shape[0]
    pack = _pack_generic([conv], [norm], skip=skip, relu=relu, device=dev)
    out = torch.empty((n, pack.N), dtype=torch.float32, device=dev)
    if n == 0:
        return out
    x = x.float().contiguous()
    # the device-side node count of the kernels is optional: here it IS n (a device scalar would cost a host-to-device copy)
    dom = conv.lut_domain
    stream = _lib.cur_stream(dev)
    xs, lds = (None, 0) if xskip is None else (xskip.float().contiguous(), xskip.shape[1])
    passes = L.dagr_spline_conv_fused_passes(pack.cin, pack.cskip)
    if passes == 1 or (passes > 1 and n <= 1600):       # wide rows in passes: small graphs only (include/dagr_hip.h)
        _lib.check(L.dagr_spline_conv_fused(None, n, P(rowptr), P(col), P(code), P(x), x.shape[1], pack.cin, P(xs),
                                            lds, pack.cskip, dom["rx"], dom["ry"], dom["den_x"], dom["den_y"], P(pack.Wq),
                                            P(pack.bias), P(out), pack.N, pack.N, 1 if relu else 0, stream),
                   "spline_conv_fused")
        return out
    lda = (pack.K + 3) // 4 * 4
    A = torch.empty((n, lda), dtype=torch.float32, device=dev)
    _lib.check(L.dagr_spline_tap_aggregate(None, n, P(rowptr), P(col), P(code), P(x), x.shape[1], pack.cin, P(xs), lds,
                                           pack.cskip, dom["rx"], dom["ry"], dom["den_x"], dom["den_y"], P(A), lda, stream),
               "tap_aggregate")
    _lib.check(L.dagr_gemm_bias_act(None, n, P(A), lda, P(pack.Wm), pack.ldw, P(pack.bias), P(out), pack.N, pack.K,
                                    pack.N, 1 if relu else 0, stream), "gemm")
    return out


EXACT_R = 1 << 14       # offset bias of the exact (training) codes: any |offset| < 16384 pixels fits 16 bits per axis


def exact_codes(edge_attr, max_value, width, height):
    """Integer pixel offsets of the edges recovered from their Cartesian attributes, attr = (pos[src] - pos[dst]) / (2 max)
    + 0.5 with pos in sensor fractions, packed like ``lut_codes`` around the bias EXACT_R, plus the matching (den_x, den_y):
    the kernels then evaluate the spline basis at pseudo = offset / den + 0.5 = the attribute itself.  This is the
    training-mode message (PyG ``SplineConv.message`` on the float attributes, spline_conv.py:64-78): no table, hence none
    of ``message_lut``'s index truncation -- which matters for the one-scale head, whose table covers another level's
    domain (dagr.py:52-62)."""
    den_x = float(torch.as_tensor(2 * max_value * width, dtype=torch.float32))
    den_y = float(torch.as_tensor(2 * max_value * height, dtype=torch.float32))
    dx = torch.round((edge_attr[:, 0] - 0.5) * den_x).int() + EXACT_R
    dy = torch.round((edge_attr[:, 1] - 0.5) * den_y).int() + EXACT_R
    return (dx | (dy << 16)).contiguous(), den_x, den_y


def conv_on_data(conv, data, norm=None, skip=None, xskip=None, relu=False):
    rowptr, col, perm = graph_csr(data)
    if conv.training:
        # training mode (spline_conv.py:64-78 without init_lut's message_lut): exact basis at the edge's own offset
        if conv.lut_domain is None or getattr(data, "edge_attr_max", None) is None:
            raise RuntimeError("training-mode SplineConv needs the sensor size (init_lut / cache_luts) and the Cartesian "
                               "maximum of the graph's edge attributes (set by Cartesian / Pooling.forward)")
        if not (norm is None and skip is None and not relu):
            raise RuntimeError("fused conv + BN epilogues are eval-mode only")
        d = conv.lut_domain
        px = getattr(data, "_dagr_pixel_codes", None)
        if px is not None and px[1] == d["width"] and px[2] == d["height"]:
            # the kernels that built this graph also wrote every edge's integer pixel offset (what its Cartesian attribute
            # encodes, whatever the attribute's maximum): no attribute tensor, no rounding back
            code = px[0]
            den_x = float(torch.as_tensor(2 * data.edge_attr_max * d["width"], dtype=torch.float32))
            den_y = float(torch.as_tensor(2 * data.edge_attr_max * d["height"], dtype=torch.float32))
        else:
            ea = data.edge_attr                     # (may build edge_index, which settles the CSR's permutation)
            rowptr, col, perm = graph_csr(data)
            # every conv on this graph sees the same codes (a level runs 2 convs on it, a head 3-6): cached with the CSR
            key = (ea.data_ptr(), col.data_ptr(), float(data.edge_attr_max), d["width"], d["height"])
            cached = getattr(data, "_dagr_exact", None)
            if cached is None or cached[0] != key:
                cached = (key, exact_codes(_in_csr_order(ea, perm), data.edge_attr_max, d["width"], d["height"])
                          if col.shape[0] else (col, 1.0, 1.0))
                data._dagr_exact = cached
            code, den_x, den_y = cached[1]
        from .autograd import SplineConvFn
        return SplineConvFn.apply(data.x, conv.weight, conv.lin.weight, conv.bias, rowptr, col, code, EXACT_R, EXACT_R,
                                  den_x, den_y)
    if col.shape[0]:
        ea = data.edge_attr                         # (may build edge_index, which settles the CSR's permutation)
        rowptr, col, perm = graph_csr(data)
        code = lut_codes(_in_csr_order(ea, perm), conv.lut_domain)
    else:
        code = col
    if norm is None and skip is None and not relu and torch.is_grad_enabled() and \
            (data.x.requires_grad or conv.weight.requires_grad):
        from .autograd import spline_conv_autograd       # eval-mode (LUT-domain) conv with gradients
        return spline_conv_autograd(conv, data.x, rowptr, col, code)
    return spline_conv(conv, data.x, rowptr, col, code, norm=norm, skip=skip, xskip=xskip, relu=relu)


def cartesian(pos, edge_index, max_value):
    """``T.Cartesian(norm=True, cat=False, max_value)``: (pos[src] - pos[dst]) / (2 max) + 0.5."""
    if edge_index.shape[1] == 0:
        return torch.zeros((0, pos.shape[1]), dtype=pos.dtype, device=pos.device)
    return (pos[edge_index[0]] - pos[edge_index[1]]) / (2 * max_value) + 0.5


def voxel_pool(pool, data):
    """``Pooling.forward`` (pooling.py:51-97) through ``dagr_pool_csr``; reads back the two output counts."""
    L, P = _lib.lib(), _lib.ptr
    dev = data.x.device
    n, C = data.x.shape
    if n == 0:
        return data
    # the module's constants as python numbers, read back from its buffers once (each read-back is a host synchronisation;
    # a step runs four poolings); keyed by the buffers' versions so that an edited / reloaded module is read again
    ckey = (pool.voxel_size._version, pool.voxel_size.data_ptr(), pool.wh_inv._version, pool.wh_inv.data_ptr())
    consts = pool.__dict__.get("_dagr_consts")
    if consts is None or consts[0] != ckey:
        vs = pool.voxel_size.detach().float().cpu()
        g = ((torch.Tensor([0.9999999, 0.9999999]) - 0) / vs[:2]).to(torch.int64) + 1
        wh = pool.wh_inv.detach().float().cpu()
        consts = (ckey, int(g[0]), int(g[1]), float(vs[0]), float(vs[1]), float(wh[0, 0]), float(wh[0, 1]))
        pool.__dict__["_dagr_consts"] = consts
    _, gx, gy, vx, vy, inv_w, inv_h = consts
    g = (gx, gy)
    B = int(pool.batch_size)
    desc = _lib.PoolDesc(batch_size=B, channels=C, gx=gx, gy=gy, vx=vx, vy=vy, inv_w=inv_w, inv_h=inv_h,
                         two_max=float(torch.as_tensor(2 * pool.transform.max, dtype=torch.float32)), r00=1.0, r02=0.0,
                         r11=1.0, r12=0.0, rx=1 << 14, ry=1 << 14, aggr=0 if pool.aggr == "max" else 1, append_pos=0)
    key = (str(dev), C)
    ws = pool.__dict__.setdefault("_dagr_ws", {}).get(key)
    stream = _lib.cur_stream(dev)
    if ws is None:
        nbytes = L.dagr_pool_workspace_bytes(ctypes.byref(desc))
        if nbytes == 0:
            raise RuntimeError("libdagr_hip: " + L.dagr_last_error().decode())
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _lib.check(L.dagr_pool_workspace_init(ctypes.byref(desc), P(ws), nbytes, stream), "pool_ws_init")
        pool._dagr_ws[key] = ws
    T = gx * gy * (B + 1)
    pool.__dict__.setdefault("_dagr_status_off", {})
    rowptr, col, _ = graph_csr(data)
    keep_order = bool(getattr(pool, "keep_temporal_ordering", False))
    if not keep_order:
        # the kernel writes every coarse edge's integer pixel offset (what its Cartesian attribute encodes) as the exact
        # code of the training-mode convs: ix = trunc(attr * den + (R + 0.5 - den / 2 - 1e-3) + 1e-3) = round((attr - 0.5) * den) + R
        W, H = int(round(1.0 / inv_w)), int(round(1.0 / inv_h))
        den_x = float(torch.as_tensor(2 * pool.transform.max * W, dtype=torch.float32))
        den_y = float(torch.as_tensor(2 * pool.transform.max * H, dtype=torch.float32))
        desc.r00, desc.r02 = den_x, EXACT_R + 0.5 - 0.5 * den_x - 1e-3
        desc.r11, desc.r12 = den_y, EXACT_R + 0.5 - 0.5 * den_y - 1e-3
    i32 = dict(dtype=torch.int32, device=dev)
    # (every entry a consumer reads is written by the kernel: clusters [0, nc), CSR rows [0, nc], edges [0, ne), all n raw ids)
    x_out = torch.empty((T, C), dtype=torch.float32, device=dev)
    pos_out = torch.empty((T, 3), dtype=torch.float32, device=dev)
    batch_out, counts = torch.empty((T,), **i32), torch.zeros((3,), **i32)
    rowptr_out = torch.empty((T + 2,), **i32)
    e_cap = T * 64
    col_out, code_out = torch.empty((e_cap,), **i32), torch.empty((e_cap,), **i32)
    batch = (data.batch if data.batch is not None else torch.zeros(n, dtype=torch.int64, device=dev)).int().contiguous()
    scratch = torch.empty((n,), **i32)
    _lib.check(L.dagr_pool_csr(ctypes.byref(desc), P(ws), None, n, P(data.x.float().contiguous()), C,
                               P(data.pos.float().contiguous()), P(batch), P(rowptr), P(col), P(scratch), P(x_out), C, 0,
                               P(pos_out), P(batch_out), P(counts), P(rowptr_out), P(col_out), P(code_out),
                               ctypes.c_void_p(counts.data_ptr() + 4), e_cap, stream), "pool_csr")
    # the two output counts and the pooling's status word in ONE read-back (the only host synchronisation of this call:
    # the output shapes depend on it)
    soff = pool._dagr_status_off.get(key)
    if soff is None:
        soff = pool._dagr_status_off[key] = L.dagr_pool_status_ptr(ctypes.byref(desc), P(ws)) - ws.data_ptr()
    counts[2:3].copy_(ws[soff:soff + 4].view(torch.int32))
    nc, ne, flags = [int(v) for v in counts.tolist()]
    if flags & ~8:       # bit 3 (LUT range) is meaningless here: no consumer table was given
        raise RuntimeError(f"pooling flagged {flags:#x}")
    out = data.__class__()
    out.__dict__.update({k: v for k, v in data.__dict__.items() if not k.startswith("_dagr") and k != "_lazy"})
    out.x, out.pos, out.batch = x_out[:nc], pos_out[:nc], batch_out[:nc].long()
    if not keep_order:
        # the kernel's own CSR (rows = destinations, sources ascending = the order csr_by_destination would establish) IS
        # the level's graph for the convs and the next pooling; the reference-shaped edge_index (unique's order) and
        # edge_attr are built from it when something asks for them -- nothing in a training step does
        out._dagr_csr = (rowptr_out[:nc + 1], col_out[:ne], None, ("csr", nc))
        out._dagr_pixel_codes = (code_out[:ne], W, H)
        out.set_lazy("edge_index", pooled_edge_index)
        out.set_lazy("edge_attr", lambda d_, m=pool.transform.max: cartesian(d_.pos, d_.edge_index, m))
    else:
        dst = torch.repeat_interleave(torch.arange(nc, device=dev), (rowptr_out[1:nc + 1] - rowptr_out[:nc]).long(),
                                      output_size=ne)
        src = col_out[:ne].long()
        order = torch.argsort(src * max(nc, 1) + dst, stable=True)    # edge_index.unique(dim=-1): by source, then destination
        ei = torch.stack([src[order], dst[order]])
        if ne > 0:
            # pooling.py:69-72: coarse edges only towards clusters whose newest member is strictly newer than the source's.
            # scratch holds every node's raw voxel id; clusters are the occupied voxels in ascending id order.
            raw = scratch.long()
            ok = raw >= 0
            _, inv = torch.unique(raw[ok], return_inverse=True)
            t_max = torch.full((nc,), float("-inf"), dtype=torch.float32, device=dev)
            t_max.scatter_reduce_(0, inv, data.pos[ok][:, -1].float(), "amax")
            ei = ei[:, t_max[ei[1]] > t_max[ei[0]]]
        out.edge_index = ei
        out.edge_attr = cartesian(out.pos, ei, pool.transform.max)
    if torch.is_grad_enabled() and data.x.requires_grad:
        # training path: attach the backward of the feature aggregation (torch_scatter's autograd in the reference).
        # scratch holds each node's raw voxel id; the output clusters are the occupied voxels in ascending id order.
        from types import SimpleNamespace
        from .autograd import PoolFeatFn
        # consecutive id of a node's voxel = number of occupied voxels below it (no sort, no host synchronisation)
        raw = scratch.long()
        valid = raw >= 0
        idx = raw.clamp(min=0)
        occ = torch.zeros((T + 1,), **i32)
        occ.scatter_reduce_(0, idx, valid.to(torch.int32), "amax")
        newid = torch.cumsum(occ, 0, dtype=torch.int32) - occ
        cluster = torch.where(valid, newid[idx], torch.full_like(newid[:1], -1)).contiguous()
        out.x = PoolFeatFn.apply(data.x, cluster, 0 if pool.aggr == "max" else 1, SimpleNamespace(pooled=x_out[:nc]))
    out.edge_attr_max = pool.transform.max
    return out


_DENSE_CONSTS = {}


def _dense_consts(pooling):
    """(Wc, Hc, vx, vy) of a voxel-size tensor as python numbers, read back once per buffer (and version of it): the
    entry is tied to the buffer OBJECT the (possibly sliced) argument views -- an address alone could be reused by another
    model's buffer after this one is gone."""
    import weakref
    base = pooling._base if pooling._base is not None else pooling
    key = (id(base), pooling.data_ptr(), pooling._version, str(pooling.device))
    hit = _DENSE_CONSTS.get(key)
    if hit is not None and hit[0]() is base:
        return hit[1]
    if len(_DENSE_CONSTS) > 64:
        _DENSE_CONSTS.clear()
    p = pooling.detach().float().cpu()
    Wc, Hc = [int(v) for v in (1 / p[:2] + 1e-3).long()]
    c = (Wc, Hc, float(p[0]), float(p[1]))
    _DENSE_CONSTS[key] = (weakref.ref(base), c)
    return c


def to_dense(x, pos, pooling, batch, batch_size):
    """``to_dense`` (spline_conv.py:80-107): node rows scattered into a zeroed [B, C, H, W] map."""
    L, P = _lib.lib(), _lib.ptr
    dev = x.device
    Wc, Hc, vx, vy = _dense_consts(pooling)
    n, C = x.shape
    if torch.is_grad_enabled() and x.requires_grad:
        from .autograd import ToDenseFn
        return ToDenseFn.apply(x, pos, batch, vx, vy, int(batch_size), Hc, Wc)
    dense = torch.zeros((batch_size, C, Hc, Wc), dtype=torch.float32, device=dev)
    if n == 0:
        return dense
    winner = torch.zeros((batch_size * Hc * Wc,), dtype=torch.int32, device=dev)
    status = torch.zeros((1,), dtype=torch.int32, device=dev)
    _lib.check(L.dagr_to_dense(None, n, P(x.float().contiguous()), C, C, P(pos.float().contiguous()),
                               P(batch.int().contiguous()), vx, vy, batch_size, Hc, Wc,
                               P(winner), P(dense), P(status), _lib.cur_stream(dev)), "to_dense")
    return dense


def sample_features(data, image_feat, width, height):
    """``sample_features`` (net.py:193-221) of an NCHW feature map at the nodes of ``data``."""
    L, P = _lib.lib(), _lib.ptr
    dev = data.x.device
    n = data.pos.shape[0]
    Bf, C, h, w = image_feat.shape
    out = torch.empty((n, C), dtype=torch.float32, device=dev)
    if n == 0:
        return out
    nhwc = image_feat.float().permute(0, 2, 3, 1).contiguous()
    batch = data.batch if data.batch is not None else torch.zeros(n, dtype=torch.int64, device=dev)
    b64 = 1 if batch.dtype == torch.int64 else 0
    _lib.check(L.dagr_sample_features(None, n, P(data.pos.float().contiguous()), P(batch.contiguous()), b64, P(nhwc), Bf, h,
                                      w, C, int(width), int(height), P(out), C, 0, _lib.cur_stream(dev)), "sample_features")
    return out
