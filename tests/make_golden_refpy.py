#!/usr/bin/env python
"""Golden vectors from the reference's OWN Python functions, run here on CPU.

The reference package cannot be imported as is (torch_geometric, torch_scatter, torch_cluster,
torch_spline_conv, torchvision, yolox, detectron2 and its compiled ev_graph_cuda are absent), but the functions
below are plain torch code of the reference itself.  This script stubs the absent packages (empty placeholder
classes: enough for `import` and class definitions, never called), imports the reference modules from
/root/reference/src, calls those functions on seeded inputs and stores inputs + outputs in
tests/golden/ref_py_functions.npz.  tests/test_oracle_refpy.py then checks the oracle restatements (and the host
mirror where it has the same function) against the file -- without /root/reference.

Functions covered (reference file:line):
  net.py:204-221       _sample_features          (3-D grid_sample of the image features)
  spline_conv.py:80-107 to_dense
  pooling.py:12-16     consecutive_cluster
  pooling.py:47-49     Pooling.round_to_pixel
  net.py:19-28         compute_pooling_at_each_layer
  model/utils.py:112-116 voxel_size_to_params
  model/utils.py:119-131 init_grid_and_stride  +  dagr.py:306-312 GNNHead.decode_outputs
  model/utils.py:61-110 postprocess_network_output (with torchvision.ops.nms := oracle.postprocess.nms, the one
                        third-party call inside it)
  utils/buffers.py:33-44 format_data
  ev_tgn.py:11-16      denormalize_pos
and, with the third-party calls INSIDE them served by the oracle's restatements (so these two pin the reference's own
glue around those calls, not the third-party arithmetic):
  spline_conv.py:16-47 MySplineConv.init_lut + message_lut   (torch_spline_conv.spline_basis := oracle.ops.spline_basis)
  pooling.py:51-97     Pooling.forward, transform=None        (torch_cluster.grid_cluster, torch_scatter.scatter_max,
                                                               PyG pool_pos := oracle.ops.grid_cluster / scatter_max / scatter_mean)
  graph/ev_graph.py:18-166 + graph/utils.py:6-23  AsyncGraph / SlidingWindowGraph host state machine over three
                        consecutive windows (ev_graph_cuda := the C emulation oracle/graph_oracle.c, itself pinned to the
                        real kernels by tests/golden/graph_ref_small.npz)
Run: python tests/make_golden_refpy.py   (only in the build container; /root/reference is not on the GPU box)."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refpy_fakes  # noqa: E402  (only use_reference_package(); this generator keeps its own stubs)


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (), {"__init__": lambda self, *a, **k: None})
        setattr(self, name, cls)
        return cls


def _stub(name):
    parts = name.split(".")
    for i in range(1, len(parts) + 1):
        n = ".".join(parts[:i])
        if n not in sys.modules:
            m = _Stub(n)
            m.__path__ = []
            sys.modules[n] = m


def import_reference():
    refpy_fakes.use_reference_package("/root/reference/src")
    mods = None
    for _ in range(60):
        try:
            import dagr.model.networks.net as rnet
            import dagr.model.layers.spline_conv as rsc
            import dagr.model.layers.pooling as rpool
            import dagr.model.layers.ev_tgn as rtgn
            import dagr.model.utils as rutils
            import dagr.utils.buffers as rbuf
            import dagr.model.networks.dagr as rdagr
            mods = (rnet, rsc, rpool, rtgn, rutils, rbuf, rdagr)
            break
        except ModuleNotFoundError as e:
            _stub(e.name)
            refpy_fakes.use_reference_package("/root/reference/src")     # drops the half-imported dagr.* modules
    assert mods is not None
    return mods


def main():
    rnet, rsc, rpool, rtgn, rutils, rbuf, rdagr = import_reference()
    from oracle import postprocess as opost
    import torchvision                                   # the stub
    torchvision.ops = types.SimpleNamespace(nms=opost.nms)
    g = torch.Generator().manual_seed(20250925)
    out = {}

    # ---- _sample_features
    for k, (B, C, h, w, W, H, N) in enumerate([(1, 5, 7, 9, 40, 30, 150), (3, 4, 15, 20, 320, 215, 400),
                                               (2, 8, 24, 32, 640, 480, 300)]):
        feat = torch.randn((B, C, h, w), generator=g)
        x = torch.randint(0, W, (N,), generator=g).float()
        y = torch.randint(0, H, (N,), generator=g).float()
        x[:4] = torch.tensor([0.0, W - 1.0, 0.0, W - 1.0])
        y[:4] = torch.tensor([0.0, 0.0, H - 1.0, H - 1.0])
        b = torch.randint(0, B, (N,), generator=g)
        s = rnet._sample_features(x, y, b.float(), feat, W, H, B, "bilinear")
        out.update({f"samp{k}_feat": feat, f"samp{k}_x": x, f"samp{k}_y": y, f"samp{k}_b": b,
                    f"samp{k}_WH": torch.tensor([W, H]), f"samp{k}_out": s})

    # ---- compute_pooling_at_each_layer, voxel_size_to_params
    for k, spec in enumerate(["5x7", "4x5"]):
        out[f"pool_sizes{k}"] = rnet.compute_pooling_at_each_layer(spec, 4)
    ps = rnet.compute_pooling_at_each_layer("5x7", 4)
    vp = []
    for i in range(4):
        layer = types.SimpleNamespace(voxel_size=ps[i], transform=types.SimpleNamespace(max=float(2 * ps[i][0])))
        vp.append(list(rutils.voxel_size_to_params(layer, 215, 320)))
    out["voxel_params"] = torch.tensor(vp, dtype=torch.float64)

    # ---- consecutive_cluster
    src = torch.randint(0, 50, (300,), generator=g) * 7
    u, inv, perm, cnt = rpool.consecutive_cluster(src)
    out.update(cc_src=src, cc_unique=u, cc_inv=inv, cc_perm=perm, cc_counts=cnt)

    # ---- round_to_pixel (in place in the reference: pass a copy)
    W, H = 320, 215
    wh_inv = torch.tensor([1.0 / W, 1.0 / H])
    pos = torch.rand((500, 2), generator=g)
    pos[:50] = (torch.randint(0, W, (50, 1), generator=g).float() * wh_inv[0]).expand(50, 2).clone()   # on pixel edges
    pos[:50, 1] = torch.randint(0, H, (50,), generator=g).float() * wh_inv[1]
    out["rtp_in"] = pos.clone()
    out["rtp_whinv"] = wh_inv
    out["rtp_out"] = rpool.Pooling.round_to_pixel(None, pos.clone(), wh_inv)

    # ---- to_dense: one node per cell at most (what pooling guarantees), B = 2
    pooling = ps[2]
    Wc, Hc = [int(v) for v in (1 / pooling[:2] + 1e-3).long()]
    cells = torch.randperm(2 * Wc * Hc, generator=g)[:60]
    bt = cells // (Wc * Hc)
    cy = (cells % (Wc * Hc)) // Wc
    cx = cells % Wc
    pos = torch.stack([(cx.float() + 0.3) * pooling[0], (cy.float() + 0.6) * pooling[1], torch.rand(60, generator=g)], 1)
    xd = torch.randn((60, 6), generator=g)
    dense = rsc.to_dense(types.SimpleNamespace(), xd, pos, pooling, batch=bt, batch_size=2)
    out.update(dense_x=xd, dense_pos=pos, dense_pooling=pooling, dense_batch=bt, dense_out=dense.clone())

    # ---- init_grid_and_stride + decode_outputs
    hw, strides = [(20, 27), (10, 14)], [11, 22]
    grid, stride = rutils.init_grid_and_stride(hw, strides, torch.float32)
    raw = torch.randn((2, grid.shape[1], 7), generator=g)
    head = types.SimpleNamespace(grid_cache=None, stride_cache=None, hw=hw, strides=strides)
    dec = rdagr.GNNHead.decode_outputs(head, raw.clone(), torch.float32)
    out.update(dec_grid=grid, dec_stride=stride, dec_raw=raw, dec_out=dec, dec_hw=torch.tensor(hw),
               dec_strides=torch.tensor(strides))

    # ---- postprocess_network_output (2 images, 3 classes)
    pred = torch.rand((2, 120, 8), generator=g)
    pred[..., :2] *= 300
    pred[..., 2:4] = pred[..., 2:4] * 60 + 5
    res = rutils.postprocess_network_output(pred.clone(), 3, conf_thre=0.2, nms_thre=0.5, height=215, width=320)
    out["post_pred"] = pred
    for i, r in enumerate(res):
        out[f"post{i}_boxes"], out[f"post{i}_scores"], out[f"post{i}_labels"] = r["boxes"], r["scores"], r["labels"]

    # ---- format_data, denormalize_pos
    N = 257
    d = types.SimpleNamespace(width=torch.tensor([320]), height=torch.tensor([215]), time_window=torch.tensor([1000000]),
                              pos=torch.stack([torch.randint(0, 320, (N,), generator=g), torch.randint(0, 215, (N,), generator=g)],
                                              1).to(torch.int16),
                              t=torch.sort(torch.randint(950000, 1000001, (N,), generator=g)).values.to(torch.int32),
                              x=(2 * torch.randint(0, 2, (N, 1), generator=g) - 1).to(torch.int8))
    out.update(fmt_pos=d.pos.clone(), fmt_t=d.t.clone(), fmt_x=d.x.clone())
    d = rbuf.format_data(d)
    out.update(fmt_out_pos=d.pos.clone(), fmt_out_x=d.x.clone())
    out["denorm_out"] = rtgn.denormalize_pos(d)

    # ---- MySplineConv.init_lut + message_lut (basis from the oracle)
    from oracle import ops as oo
    rsc.spline_basis = lambda pseudo, kernel_size, is_open_spline, degree: oo.spline_basis(
        pseudo, int(kernel_size[0]), int(is_open_spline[0]), int(degree))
    Himg, Wimg, rx, ry, Mx, My = 215, 320, 3, 4, 3.0 / 320, 4.0 / 215
    conv = types.SimpleNamespace(weight=torch.randn((25, 3, 5), generator=g), kernel_size=torch.tensor([5, 5]),
                                 is_open_spline=torch.tensor([1, 1]), degree=1, message_lut=None)
    rsc.MySplineConv.init_lut(conv, Himg, Wimg, rx, Mx, ry, My)
    E = 400
    off = torch.stack([torch.randint(-rx, rx + 1, (E,), generator=g), torch.randint(-ry, ry + 1, (E,), generator=g)], 1).float()
    edge_attr = off / torch.tensor([2 * Mx * Wimg, 2 * My * Himg]) + 0.5
    x_j = torch.randn((E, 3), generator=g)
    msg = rsc.MySplineConv.message_lut(conv, x_j, edge_attr)
    out.update(lut_weight=conv.weight, lut_params=torch.tensor([Himg, Wimg, rx, ry, Mx, My], dtype=torch.float64),
               lut_table=conv.lut_weights, lut_remap=conv.attr_remapping_matrix, lut_edge_attr=edge_attr, lut_xj=x_j,
               lut_msg=msg)

    # ---- Pooling.forward (third-party primitives from the oracle)
    class _Batch:
        def __init__(self, **kw):
            self.__dict__.update(kw)
    rpool.Batch = _Batch
    rpool.grid_cluster = lambda pos, size, start, end: oo.grid_cluster(pos, size, start, end)
    rpool.torch_scatter = types.SimpleNamespace(
        scatter_max=lambda src, index, dim=0: (
            oo.scatter_max(src.reshape(src.shape[0], -1), index, int(index.max()) + 1).reshape((-1,) + tuple(src.shape[1:])),
            None))
    rpool.pool_pos = lambda cluster, pos: oo.scatter_mean(pos, cluster, int(cluster.max()) + 1)
    rpool._avg_pool_x = lambda cluster, x: oo.scatter_mean(x, cluster, int(cluster.max()) + 1)
    W, H, B, N, E = 320, 215, 2, 900, 5000
    for k, aggr in enumerate(["max", "mean"]):
        pl = rpool.Pooling(ps[0], width=W, height=H, batch_size=B, transform=None, aggr=aggr)
        pos = torch.stack([torch.randint(0, W, (N,), generator=g).float() / W, torch.randint(0, H, (N,), generator=g).float() / H,
                           torch.rand(N, generator=g)], 1)
        pos[-1, 2] = 1.0                                        # a t == 1.0 node (leaks into the next sample's ids)
        batch = torch.sort(torch.randint(0, B, (N,), generator=g)).values
        x = torch.randn((N, 4), generator=g)
        ei = torch.randint(0, N, (2, E), generator=g)
        d = types.SimpleNamespace(x=x.clone(), pos=pos.clone(), batch=batch.clone(), edge_index=ei.clone(),
                                  height=torch.tensor([H]), width=torch.tensor([W]))
        r = pl.forward(d)
        out.update({f"pool{k}_size": ps[0], f"pool{k}_x": x, f"pool{k}_pos": pos, f"pool{k}_batch": batch,
                    f"pool{k}_ei": ei, f"pool{k}_out_x": r.x, f"pool{k}_out_pos": r.pos, f"pool{k}_out_batch": r.batch,
                    f"pool{k}_out_ei": r.edge_index})

    # ---- Pooling.forward with keep_temporal_ordering=True (pooling.py:69-72): coarse edges only towards clusters whose newest
    # member is newer than the source cluster's (its own generator: the entries above stay as they were)
    g2 = torch.Generator().manual_seed(2024)
    pl = rpool.Pooling(ps[1], width=W, height=H, batch_size=B, transform=None, aggr="max", keep_temporal_ordering=True)
    pos = torch.stack([torch.randint(0, W, (N,), generator=g2).float() / W, torch.randint(0, H, (N,), generator=g2).float() / H,
                       torch.rand(N, generator=g2)], 1)
    pos[5:9, 2] = pos[5, 2]                                     # equal newest timestamps in places: `>` is strict
    batch = torch.sort(torch.randint(0, B, (N,), generator=g2)).values
    x = torch.randn((N, 4), generator=g2)
    ei = torch.randint(0, N, (2, E), generator=g2)
    d = types.SimpleNamespace(x=x.clone(), pos=pos.clone(), batch=batch.clone(), edge_index=ei.clone(),
                              height=torch.tensor([H]), width=torch.tensor([W]))
    r = pl.forward(d)
    out.update({"poolt_size": ps[1], "poolt_x": x, "poolt_pos": pos, "poolt_batch": batch, "poolt_ei": ei,
                "poolt_out_x": r.x, "poolt_out_pos": r.pos, "poolt_out_batch": r.batch, "poolt_out_ei": r.edge_index})

    # ---- AsyncGraph / SlidingWindowGraph host logic (kernels from the oracle's C emulation)
    from oracle import graph as og
    import dagr.graph.ev_graph as rgraph
    import dagr.graph.utils as rgutils
    L = og.lib()

    def _np32(t):
        a = t.numpy()
        assert a.dtype == np.int32 and a.flags.c_contiguous
        return a

    def insert_in_queue_cuda(sorted_indices, unique_coords, cumsum_counter, queue):
        B, Q, H, W = queue.shape
        uc = np.ascontiguousarray(unique_coords.numpy().astype(np.int32))
        L.oracle_insert_in_queue(og._p32(_np32(sorted_indices.contiguous())), og._p32(uc),
                                 og._p32(_np32(cumsum_counter.contiguous())), og._p32(_np32(queue)), B, Q, H, W, len(uc))
        return queue

    def insert_in_queue_single_cuda(indices, pos, queue):
        B, Q, H, W = queue.shape
        L.oracle_insert_in_queue_single(og._p32(_np32(indices.contiguous())), og._p32(_np32(pos.contiguous())),
                                        og._p32(_np32(queue)), B, Q, H, W)
        return queue

    def fill_edges_cuda(batch, pos, all_timestamps, queue, indices, K, radius, delta_t_us, edges, min_index):
        B, Q, H, W = queue.shape
        e = edges.numpy()
        L.oracle_fill_edges(og._p32(_np32(batch.contiguous())), og._p32(_np32(pos.contiguous())),
                            og._p32(_np32(all_timestamps)), og._p32(_np32(indices.contiguous())), og._p32(_np32(queue)),
                            e.ctypes.data_as(og._i64p), B, Q, H, W, len(batch), e.shape[1], float(radius),
                            float(delta_t_us), int(K), int(min_index))

    fake = types.SimpleNamespace(insert_in_queue_cuda=insert_in_queue_cuda,
                                 insert_in_queue_single_cuda=insert_in_queue_single_cuda, fill_edges_cuda=fill_edges_cuda)
    rgutils.ev_graph_cuda = fake
    Wg, Hg, Bg, Kg, Qg, rg, dtg = 48, 36, 2, 16, 8, 3, 30000
    swg = rgraph.SlidingWindowGraph(width=Wg, height=Hg, batch_size=Bg, max_num_neighbors=Kg, max_queue_size=Qg, radius=rg,
                                    delta_t_us=dtg)
    t0 = 0
    for w, n in enumerate([400, 1, 350, 0, 300]):          # incl. the single-event insert and an empty window
        bt = torch.sort(torch.randint(0, Bg, (n,), generator=g)).values.to(torch.int32)
        ts = torch.sort(torch.randint(t0, t0 + 50000, (n,), generator=g)).values
        t0 += 50000
        ps_ = torch.stack([torch.randint(0, Wg, (n,), generator=g), torch.randint(0, Hg, (n,), generator=g), ts], 1).to(torch.int32)
        if n == 1:
            bt = torch.zeros(1, dtype=torch.int32)
        ret = swg.forward(bt, ps_, return_node_counts=True, return_total_edges=True, delete_nodes=True, collect_edges=True)
        edges_w, deleted_w, total_w, counts_w = ret
        out.update({f"swg{w}_batch": bt, f"swg{w}_pos": ps_, f"swg{w}_edges": edges_w.clone(),
                    f"swg{w}_deleted": (deleted_w.clone() if deleted_w is not None else torch.zeros((2, 0), dtype=torch.long)),
                    f"swg{w}_total": total_w.clone(), f"swg{w}_counts": torch.tensor(counts_w)})
    out["swg_params"] = torch.tensor([Wg, Hg, Bg, Kg, Qg, rg, dtg])

    # ---- ModelEMA (networks/ema.py:6-51): three updates of a toy module
    import dagr.model.networks.ema as rema

    def toy(seed):
        torch.manual_seed(seed)
        m = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3))
        with torch.no_grad():
            m[1].running_mean.normal_()
            m[1].running_var.uniform_(0.5, 1.5)
        return m
    ema = rema.ModelEMA(toy(0))
    for s in (1, 2, 3):
        ema.update(toy(s))
    for k, v in ema.ema.state_dict().items():
        out["ema_" + k] = v.clone()
    out["ema_updates"] = torch.tensor(ema.updates)

    # ---- utils/args.py FLAGS() on the shipped model configs (what `args` the model constructors see)
    import json
    import dagr.utils.args as rargs
    flags = {}
    argv0 = list(sys.argv)
    for name in ("dagr-n", "dagr-s", "dagr-m", "dagr-l"):
        sys.argv = ["x", "--config", f"/root/reference/config/{name}-dsec.yaml", "--dataset_directory", "/d",
                    "--output_directory", "/o", "--batch_size", "8"]
        ns = rargs.FLAGS()
        flags[name] = {k: (str(v) if not isinstance(v, (int, float, bool, str)) else v) for k, v in vars(ns).items()}
    sys.argv = argv0
    out["flags_json"] = np.frombuffer(json.dumps(flags, sort_keys=True).encode(), dtype=np.uint8)

    path = os.path.join(os.environ.get("GOLDEN_OUT", os.path.join(ROOT, "tests", "golden")), "ref_py_functions.npz")
    np.savez_compressed(path, **{k: (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()})
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
