#!/usr/bin/env python
"""Golden outputs of the reference's OWN model code, executed on CPU.

tests/refpy_fakes.py puts functional stand-ins for the absent third-party packages (PyG, torch_scatter,
torch_cluster, torch_spline_conv, yolox, ev_graph_cuda) into sys.modules, built on the oracle's restatements of
those primitives.  With them the reference's Net / Layer / ConvBlock / MySplineConv (LUT path) / Pooling /
EV_TGN + SlidingWindowGraph / GNNHead / DAGR.cache_luts run unmodified from /root/reference/src.  This script loads
reference-layout weights (the host mirror's randomised state_dict, strict=True), pushes synthetic windows through
``YOLOX.forward(model, data)`` in eval mode and stores events + decoded outputs in
tests/golden/ref_py_model.npz.  tests/test_oracle_refpy.py::test_whole_model_* then requires
oracle.model.forward_events -- the function every GPU parity test compares the HIP path with -- to reproduce
them: the oracle's wiring of the whole path is thereby pinned to the reference's code, independently of the
reading of it that produced oracle/model.py.
Run: python tests/make_golden_refpy_model.py   (build container only)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = [  # name, W, H, B, events/sample, stream, seed, model overrides
    ("s_b2", 320, 215, 2, 2500, "uniform", 11, {}),
    ("s_b1_edges", 240, 180, 1, 3000, "edges", 12, {}),
    ("l_b2", 320, 215, 2, 1500, "edges", 13, dict(net_stem_width=1.0, yolo_stem_width=1.0)),
    ("s_img18_b2", 320, 215, 2, 2000, "edges", 14, dict(use_image=True, img_net="resnet18")),
    # config/dagr-l-ncaltech.yaml: one head scale -- head "1" then consumes out4 but keeps the pool3 LUT that
    # DAGR.cache_luts gives it (dagr.py:52-62): the message_lut index is a half-resolution look-up
    ("l_ncaltech_b2", 240, 180, 2, 2500, "edges", 15, dict(net_stem_width=1.0, yolo_stem_width=1.0, num_scales=1,
                                                          dataset="ncaltech101")),
]

TRAIN_CASES = [
    ("train_s_b2", 240, 180, 2, 2500, "edges", 21, {}),
    ("train_l_ncaltech_b3", 240, 180, 3, 1500, "edges", 22, dict(net_stem_width=1.0, yolo_stem_width=1.0, num_scales=1,
                                                                dataset="ncaltech101")),
    # train_dsec.py with --use_image: image features sampled into the graph detached, CNN-head logits added detached,
    # a second get_losses on the image branch's own outputs against the earlier frame's boxes (dagr.py:241-268)
    ("train_s_img18_b2", 240, 180, 2, 2000, "edges", 23, dict(use_image=True, img_net="resnet18")),
]
TRAIN_GRAD_KEYS = ["backbone.conv_block1.conv_block1.conv.weight", "backbone.conv_block1.conv_block2.lin.mlp.weight",
                   "backbone.layer3.conv_block1.norm.module.weight", "backbone.layer5.conv_block2.conv.lin.weight",
                   "head.stem1.conv.weight", "head.cls_pred1.bias", "head.reg_pred1.weight", "head.obj_pred2.bias",
                   "backbone.net.module.conv1.weight", "backbone.net.module.layer3.0.conv1.weight",
                   "backbone.net.feature_dconv.0.weight", "head.cnn_head.stems.1.conv.weight",
                   "head.cnn_head.cls_preds.0.bias", "head.cnn_head.reg_preds.1.weight"]


def main():
    import refpy_fakes
    refpy_fakes.install()
    refpy_fakes.use_reference_package("/root/reference/src")
    import dagr.model.networks.dagr as rdagr
    from oracle import model as om
    from dagr_amd.model.networks.dagr import DAGR as MirrorDAGR
    from dagr_amd.utils import synthetic as syn
    from dagr_amd.utils.testing_weights import randomize_
    out = {}
    for name, W, H, B, n, stream, seed, over in CASES:
        args = om.default_args(batch_size=B, **over)
        torch.manual_seed(seed)
        mirror = randomize_(MirrorDAGR(args, height=H, width=W), seed=seed).eval()
        sd = mirror.state_dict()
        ref = rdagr.DAGR(argparse.Namespace(**vars(args)), height=H, width=W)
        ref.load_state_dict(sd, strict=True)      # the mirror's state_dict layout IS the reference's
        ref.eval()
        ref.cache_luts(width=W, height=H, radius=args.radius)
        gen = syn.uniform_window if stream == "uniform" else syn.edges_window
        x, y, t, p, b = syn.batch_windows(gen, n, B, W, H, seed=seed * 3 + 1)
        data = refpy_fakes.Data(x=torch.from_numpy(p.astype(np.float32)).view(-1, 1),
                                pos=torch.from_numpy(syn.format_data_np(x, y, t, W, H)), batch=torch.from_numpy(b),
                                width=torch.tensor([W] * B), height=torch.tensor([H] * B),
                                time_window=torch.tensor([1000000] * B), num_graphs=B, reset=True)
        if getattr(args, "use_image", False):     # format_data'd frames (uint8 / 255, utils/buffers.py:37-38)
            img = torch.randint(0, 256, (B, 3, H, W), generator=torch.Generator().manual_seed(seed)).float() / 255.0
            data.image = img                      # re-drawn from the seed by the test, not stored
        ref.head.output_sizes = ref.backbone.get_output_sizes()
        with torch.no_grad():
            outputs = rdagr.YOLOX.forward(ref, data)
        print(name, "outputs", tuple(outputs.shape), "edges", int(data.edge_index.shape[1]) if hasattr(data, "edge_index") else "-")
        out.update({f"{name}_x": x, f"{name}_y": y, f"{name}_t": t, f"{name}_p": p, f"{name}_b": b,
                    f"{name}_out": outputs.numpy()})
    # ---- training mode (train_ncaltech101.py:49-58): the reference's own DAGR.forward training branch -> YOLOX.forward ->
    # GNNHead.forward losses; no cache_luts (an evaluation-time step, run_test.py:59): MySplineConv evaluates the spline
    # basis on the edge attributes; BatchNorm on batch statistics; the losses are the restated yolox ones
    # (oracle/yolox_loss.py).  Stored: events, boxes, the six outputs, gradients of a few parameters along the depth.
    for name, W, H, B, n, stream, seed, over in TRAIN_CASES:
        args = om.default_args(batch_size=B, **over)
        torch.manual_seed(seed)
        mirror = randomize_(MirrorDAGR(args, height=H, width=W), seed=seed)
        ref = rdagr.DAGR(argparse.Namespace(**vars(args)), height=H, width=W)
        ref.load_state_dict(mirror.state_dict(), strict=True)
        ref.train()
        gen = syn.uniform_window if stream == "uniform" else syn.edges_window
        x, y, t, p, b = syn.batch_windows(gen, n, B, W, H, seed=seed * 3 + 1)
        rng = np.random.default_rng(seed)
        counts = [1 + (s + seed) % 3 for s in range(B)]
        counts[-1] = 0 if B > 2 else counts[-1]            # a sample without boxes
        bbox = np.concatenate([np.stack([rng.uniform(5, W / 2, c), rng.uniform(5, H / 2, c), rng.uniform(20, W / 3, c),
                                         rng.uniform(20, H / 3, c), rng.integers(0, 2, c).astype(float), np.ones(c),
                                         np.zeros(c)], 1) for c in counts]).astype(np.float32)
        bbox_batch = np.concatenate([np.full(c, s, np.int64) for s, c in enumerate(counts)])
        data = refpy_fakes.Data(x=torch.from_numpy(p.astype(np.float32)).view(-1, 1),
                                pos=torch.from_numpy(syn.format_data_np(x, y, t, W, H)), batch=torch.from_numpy(b),
                                width=torch.tensor([W] * B), height=torch.tensor([H] * B),
                                time_window=torch.tensor([1000000] * B), num_graphs=B,
                                bbox=torch.from_numpy(bbox), bbox_batch=torch.from_numpy(bbox_batch))
        if getattr(args, "use_image", False):
            data.image = torch.randint(0, 256, (B, 3, H, W), generator=torch.Generator().manual_seed(seed)).float() / 255.0
            bbox0 = bbox.copy()
            bbox0[:, :2] -= 3.0                 # the earlier frame's boxes: same tracks, shifted
            data.bbox0, data.bbox0_batch = torch.from_numpy(bbox0), torch.from_numpy(bbox_batch)
            out[f"{name}_bbox0"] = bbox0
        losses = ref(data)
        losses["total_loss"].backward()
        grads = {k: v.grad for k, v in ref.named_parameters() if v.grad is not None}
        picked = [k for k in TRAIN_GRAD_KEYS if k in grads]
        print(name, {k: float(v) for k, v in losses.items()}, len(grads), "parameters with gradients")
        out.update({f"{name}_x": x, f"{name}_y": y, f"{name}_t": t, f"{name}_p": p, f"{name}_b": b, f"{name}_bbox": bbox,
                    f"{name}_bbox_batch": bbox_batch,
                    f"{name}_losses": np.array([float(losses[k]) for k in ("total_loss", "iou_loss", "conf_loss",
                                                                           "cls_loss", "l1_loss", "num_fg")]),
                    f"{name}_grad_keys": np.array(picked),
                    f"{name}_n_grads": np.array(len(grads))})
        for k in picked:
            g = grads[k].numpy()
            if g.size > 20000:        # big image-branch tensors: a strided sample and the norm keep the fixture small
                out[f"{name}_gradnorm:{k}"] = np.array(np.sqrt((g.astype(np.float64) ** 2).sum()))
                g = g.reshape(-1)[::g.size // 4096]
            out[f"{name}_grad:{k}"] = g

    # ---- EV_TGN over consecutive calls: reset=True, reset=False (nodes attach to the running graph), reset=True
    import dagr.model.layers.ev_tgn as rtgn
    W, H, B = 64, 48, 2
    tgn = rtgn.EV_TGN(argparse.Namespace(radius=0.05, max_neighbors=16))
    for k, (reset, n, seed) in enumerate([(True, 500, 31), (False, 300, 32), (True, 400, 33)]):
        x, y, t, p, b = syn.batch_windows(syn.uniform_window, n, B, W, H, seed=seed)
        ev = refpy_fakes.Data(x=torch.from_numpy(p.astype(np.float32)).view(-1, 1),
                              pos=torch.from_numpy(syn.format_data_np(x, y, t, W, H)), batch=torch.from_numpy(b),
                              width=torch.tensor([W] * B), height=torch.tensor([H] * B),
                              time_window=torch.tensor([1000000] * B), num_graphs=B)
        ev = tgn.forward(ev, reset=reset)
        out.update({f"tgn{k}_x": x, f"tgn{k}_y": y, f"tgn{k}_t": t, f"tgn{k}_b": b, f"tgn{k}_reset": np.array(reset),
                    f"tgn{k}_edges": ev.edge_index.numpy()})
        print("EV_TGN call", k, "reset", reset, "edges", tuple(ev.edge_index.shape))
    out["tgn_params"] = np.array([W, H, B])

    # ---- the reference's in-repo restatements of two third-party primitives (asynchronous/): T.Cartesian and the
    # 2-D voxel index of grid_cluster -- anchors for the oracle's restatements of those
    refpy_fakes._module("asy_tools")
    refpy_fakes._module("torch_geometric.nn.norm")
    import importlib
    rcart = importlib.import_module("dagr.asynchronous.cartesian")
    rmp = importlib.import_module("dagr.asynchronous.max_pool")
    gq = torch.Generator().manual_seed(77)
    pos = torch.rand((400, 3), generator=gq)
    ei = torch.randint(0, 400, (2, 1500), generator=gq)
    out.update(cart_pos=pos.numpy(), cart_ei=ei.numpy(),
               cart_out=getattr(rcart, "__edge_attr")(pos, ei, True, 0.0625).numpy())
    ps = om.compute_pooling_at_each_layer("5x7", 4)
    for i in range(4):
        module = argparse.Namespace(voxel_size=torch.cat([ps[i], torch.Tensor([1])]))
        p2 = torch.stack([torch.randint(0, 320, (600,), generator=gq).float() / 320,
                          torch.randint(0, 215, (600,), generator=gq).float() / 215], 1)
        out[f"vox{i}_pos"] = p2.numpy()
        out[f"vox{i}_idx"] = getattr(rmp, "__get_global_cluster_index")(module, p2).numpy()
    out["vox_sizes"] = ps.numpy()

    path = os.path.join(os.environ.get("GOLDEN_OUT", os.path.join(ROOT, "tests", "golden")), "ref_py_model.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
