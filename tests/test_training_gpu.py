"""GPU parity of the training path (BASELINE config 5): the differentiable operators over libdagr_hip
(SplineConv, voxel pooling, to_dense) and the whole ``model.train()`` forward + backward against the CPU oracle pushed
through torch autograd (oracle/ops.py with ``batch_statistics()``; the reference trains through torch_spline_conv /
torch_scatter autograd on the same op sequence, train_ncaltech101.py:41-74, dagr.py:78-88,238-282).

Bars: forward tensors 1e-4 (the eval bar); gradients 2e-3 of the tensor's largest magnitude (fp32 through ~14 convs with
batch-statistics BatchNorm on both sides, atomics in the HIP scatter)."""
import numpy as np
import pytest
import torch

from oracle import graph as og
from oracle import model as om
from oracle import ops as oo
from dagr_amd.data import Batch, Data
from dagr_amd.utils import synthetic as syn
from dagr_amd.utils.buffers import format_data
from dagr_amd.utils.testing_weights import randomize_

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    if not a.numel():
        return 0.0
    return ((a - b).abs().max() / max(1e-12, float(b.abs().max()))).item()


def _graph(W, H, B, n, seed):
    x, y, t, p, b = syn.batch_windows(syn.edges_window, n, B, W, H, seed=seed)
    pos = torch.from_numpy(syn.format_data_np(x, y, t, W, H))
    r, dt = og.graph_params(0.01, W, 1000000)
    d = og.denormalize_pos(pos.numpy(), W, H, 1000000)
    ei = torch.from_numpy(og.build_window_graph(d[:, 0], d[:, 1], d[:, 2], b.astype(np.int32), W, H, B, r, dt, K=16, Q=128))
    return pos, torch.from_numpy(b.astype(np.int64)), ei


@pytest.mark.parametrize("aggr", ["max", "mean"])
def test_pooling_backward_matches_scatter_autograd(aggr):
    from dagr_amd.model.layers.components import Cartesian
    from dagr_amd.model.layers.pooling import Pooling
    W, H, B, C = 240, 180, 2, 8
    pos, batch, ei = _graph(W, H, B, 1500, seed=11)
    size = torch.tensor([1 / 14.0, 1 / 10.0, 1.0])
    cart_max = float(2 * size[:2].max())
    pool = Pooling(size, width=W, height=H, batch_size=B, transform=Cartesian(True, False, cart_max), aggr=aggr).cuda()
    torch.manual_seed(3)
    x = torch.randn(len(pos), C)
    xo = x.clone().requires_grad_(True)
    pp = oo.PoolingParams(size, W, H, B, cart_max, aggr=aggr)
    x_ref, pos_ref, batch_ref, ei_ref, _ = oo.pooling(pp, xo, pos, batch, ei, exact_mean=True)
    g = torch.randn_like(x_ref)
    (x_ref * g).sum().backward()
    xh = x.clone().cuda().requires_grad_(True)
    data = Data(x=xh, pos=pos.cuda(), batch=batch.cuda(), edge_index=ei.cuda())
    out = pool(data)
    assert out.x.shape == x_ref.shape and torch.equal(out.batch.cpu(), batch_ref)
    assert _rel(out.x, x_ref) < 1e-5
    (out.x * g.cuda()).sum().backward()
    assert _rel(xh.grad, xo.grad) < 1e-5


def test_to_dense_backward_gathers_every_written_row():
    from dagr_amd.model.layers import _ops
    B, C = 2, 5
    pooling = torch.tensor([1 / 7.0, 1 / 5.0, 1.0])
    torch.manual_seed(5)
    cells = torch.randperm(35 * B)[:40]
    b, cy, cx = cells // 35, (cells % 35) // 7, cells % 7
    pos = torch.stack([(cx + 0.3) / 7, (cy + 0.6) / 5, torch.rand(40)], 1).float()
    pos = torch.cat([pos, pos[:3]])                  # three shared cells: the later row survives; index_put's backward
    #                                                  still hands the overwritten rows their cell's gradient
    b = torch.cat([b, b[:3]])
    x = torch.randn(43, C)
    xo = x.clone().requires_grad_(True)
    ref = oo.to_dense(xo, pos, pooling, b, B)
    g = torch.randn_like(ref)
    (ref * g).sum().backward()
    xh = x.clone().cuda().requires_grad_(True)
    out = _ops.to_dense(xh, pos.cuda(), pooling.cuda(), b.cuda(), B)
    assert torch.allclose(out.cpu(), ref.detach(), atol=0, rtol=0)
    (out * g.cuda()).sum().backward()
    assert torch.equal(xh.grad.cpu(), xo.grad)
    assert torch.equal(xh.grad[:3], xh.grad[40:43]) and float(xh.grad[:3].abs().sum()) > 0


def _training_case(W, H, B, n, seed, **over):
    from dagr_amd.model.networks.dagr import DAGR
    torch.manual_seed(seed)
    args = om.default_args(batch_size=B, **over)
    model = randomize_(DAGR(args, height=H, width=W), seed=seed)
    sd = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() and "running" not in k
              else v.detach().clone()) for k, v in model.state_dict().items()}
    model = model.cuda().train()
    model.cache_luts(width=W, height=H, radius=args.radius)
    samples, raw = [], []
    rng = np.random.default_rng(seed)
    for s in range(B):
        x, y, t, p = syn.edges_window(n, W, H, seed=seed * 10 + s)
        raw.append((x, y, t, p))
        nb = 1 + s % 2
        boxes = np.stack([rng.uniform(5, W / 2, nb), rng.uniform(5, H / 2, nb), rng.uniform(20, W / 3, nb),
                          rng.uniform(20, H / 3, nb), rng.integers(0, 2, nb), np.ones(nb), np.zeros(nb)], 1)
        samples.append(Data(x=torch.from_numpy(p.reshape(-1, 1)), pos=torch.from_numpy(np.stack([x, y], -1)),
                            t=torch.from_numpy(t), width=W, height=H, time_window=1000000,
                            bbox=torch.from_numpy(boxes.astype(np.float32)), sequence=f"s{s}"))
    batch = Batch.from_data_list(samples, follow_batch=["bbox"])
    ev = [np.concatenate([r[k] for r in raw]) for k in range(4)]
    b = np.concatenate([np.full(len(r[0]), i, np.int64) for i, r in enumerate(raw)])
    return args, model, sd, batch, ev, b


@pytest.mark.parametrize("case", [dict(W=240, H=180, B=2, n=2500, seed=1),
                                  dict(W=320, H=215, B=3, n=1500, seed=2, over=dict(num_scales=1))],
                         ids=["two_scales", "one_scale"])
def test_training_loss_and_gradients_match_the_oracle(case):
    from oracle import train as otr
    W, H, B = case["W"], case["H"], case["B"]
    args, model, sd, batch, ev, b = _training_case(W, H, B, case["n"], case["seed"], **case.get("over", {}))
    # the oracle's training forward is pinned to the reference's own training branch (tests/test_oracle_refpy.py)
    ref = otr.training_losses(sd, args, H, W, ev[0], ev[1], ev[2], ev[3], b, B, batch.bbox, batch.bbox_batch)
    ref[0].backward()
    out = model(format_data(batch.cuda()))
    assert set(out) == {"total_loss", "iou_loss", "l1_loss", "conf_loss", "cls_loss", "num_fg"}
    assert out["num_fg"] == ref[5], "SimOTA matched a different number of anchors"
    for k, r in zip(("total_loss", "iou_loss", "conf_loss", "cls_loss"), (ref[0], ref[1], ref[2], ref[3])):
        assert abs(float(out[k]) - float(r)) <= 5e-4 * max(1.0, abs(float(r))), (k, float(out[k]), float(r))
    out["total_loss"].backward()
    params = dict(model.named_parameters())
    checked, errs = 0, []
    for k, v in sd.items():
        if not v.requires_grad or v.grad is None:
            assert k not in params or params[k].grad is None or float(params[k].grad.abs().max()) == 0.0, k
            continue
        gh = params[k].grad
        assert gh is not None, f"no gradient reached {k}"
        errs.append((_rel(gh, v.grad), k))
        checked += 1
    assert checked >= 60
    # every tensor within 2e-3 of its own scale (typically 1e-5); the backward is deterministic (fixed-point scatter in
    # dagr_spline_tap_scatter_grad, gathers everywhere else), so this does not depend on the run
    errs.sort()
    assert errs[-1][0] < 2e-3, errs[-5:]
    # batch statistics moved the running buffers (momentum 0.1), as nn.BatchNorm1d does in the reference
    bn = model.backbone.conv_block1.conv_block1.norm.module
    assert int(bn.num_batches_tracked) == 1 and float(bn.running_mean.abs().sum()) > 0


def test_training_backward_is_deterministic():
    """Two forward + backward passes over the same batch give bit-identical losses and gradients: the SplineConv input
    gradient is a fixed-point scatter (integer atomics: order-free), pooling / to_dense backwards are gathers."""
    W, H, B = 240, 180, 2
    args, model, _, batch, _, _ = _training_case(W, H, B, 3000, seed=6)
    snaps = []
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        out = model(format_data(batch.clone().cuda()))
        out["total_loss"].backward()
        snaps.append((float(out["total_loss"]), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    assert snaps[0][0] == snaps[1][0]
    assert snaps[0][1].keys() == snaps[1][1].keys() and len(snaps[0][1]) >= 60
    for k in snaps[0][1]:
        assert torch.equal(snaps[0][1][k], snaps[1][1][k]), k


@pytest.mark.parametrize("mode", ["outputs_kept", "outputs_dropped", "use_image_outputs_dropped"])
def test_two_forwards_before_one_backward_do_not_share_the_loss_graph(mode):
    """Gradient accumulation: two forwards of the same call site, ONE backward over the sum.  The graphed YOLOX loss owns
    static buffers, so the second forward must not replay it while the first one's backward is pending (it
    silently produced wrong gradients); it takes the launch-by-launch form, and the result equals the all-eager run.
    ``outputs_dropped``: the caller keeps only the running sum, so the first forward's loss TENSORS are gone while their
    backward is pending; with ``--use_image`` the hybrid sum (dagr.py:262-268) drops them inside forward itself (ADVICE r5:
    the guard must follow the autograd node, not the Python tensor)."""
    import os
    W, H, B = 240, 180, 2
    over = dict(use_image=True, img_net="resnet18") if mode.startswith("use_image") else {}
    args, model, _, batch, _, _ = _training_case(W, H, B, 3000, seed=6, **over)
    if over:
        batch.image = torch.randint(0, 256, (B, 3, H, W), generator=torch.Generator().manual_seed(6), dtype=torch.uint8)
        batch.bbox0 = batch.bbox.clone()
        batch.bbox0[:, :2] -= 3.0
        batch.bbox0_batch = batch.bbox_batch.clone()

    def accumulate():
        model.zero_grad(set_to_none=True)
        if mode == "outputs_kept":
            o1 = model(format_data(batch.clone().cuda()))
            o2 = model(format_data(batch.clone().cuda()))
            (o1["total_loss"] + o2["total_loss"]).backward()
            l1, l2 = float(o1["total_loss"]), float(o2["total_loss"])
        else:
            vals = []
            acc = None
            for _ in range(2):
                loss = model(format_data(batch.clone().cuda()))["total_loss"]      # the output dict dies here
                vals.append(loss.detach().clone())
                acc = loss if acc is None else acc + loss
                del loss
            import gc
            gc.collect()
            acc.backward()
            l1, l2 = float(vals[0]), float(vals[1])
        return l1, l2, {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    old = os.environ.pop("DAGR_GRAPH_LOSS", None)
    try:
        a1, a2, ga = accumulate()                 # graph for the first forward, launch-by-launch for the second
        b1, b2, gb = accumulate()                 # again: the mark of the first round was cleared by its backward
        os.environ["DAGR_GRAPH_LOSS"] = "0"
        e1, e2, ge = accumulate()
    finally:
        os.environ.pop("DAGR_GRAPH_LOSS", None)
        if old is not None:
            os.environ["DAGR_GRAPH_LOSS"] = old
    tol = 1e-5 if mode.startswith("use_image") else 1e-6
    if not mode.startswith("use_image"):
        assert a1 == b1 and a2 == b2 and e1 == e2
    assert abs(a1 - e1) <= tol * max(1.0, abs(e1)) and abs(a2 - e2) <= tol * max(1.0, abs(e2))
    assert ga.keys() == ge.keys() and len(ga) >= 60
    image = mode.startswith("use_image")        # (the image branch's convolutions are library kernels with atomics)
    # biases in front of a batch-statistics BatchNorm (output_dconv -> stems) have an analytically zero gradient: both runs
    # hold rounding noise there, so the image case's bar has an absolute floor tied to the run's largest gradient
    top = max(float(v.abs().max()) for v in ge.values())
    for k in ga:
        assert image or torch.equal(ga[k], gb[k]), k
        if image:
            err = float((ga[k] - ge[k]).abs().max()) / (float(ge[k].abs().max()) + 1e-5 * top)
            assert err < 1e-3, (k, err)
        else:
            assert _rel(ga[k], ge[k]) < 1e-6, (k, _rel(ga[k], ge[k]))
    # and a forward whose loss was dropped without a backward does not block the graph for good
    model.zero_grad(set_to_none=True)
    model(format_data(batch.clone().cuda()))
    import gc
    gc.collect()
    head = model.head
    marks = list(head.__dict__.get("_loss_graph_pending", {}).values())
    assert marks and all(m[0]() is None or m[1] for m in marks)


def test_a_few_optimizer_steps_reduce_the_loss():
    W, H, B = 240, 180, 2
    args, model, _, batch, _, _ = _training_case(W, H, B, 2000, seed=4)
    opt = torch.optim.AdamW(model.parameters(), lr=2e-3, weight_decay=1e-5)      # train_ncaltech101.py:134
    losses = []
    for _ in range(6):
        opt.zero_grad(set_to_none=True)
        out = model(format_data(batch.clone().cuda()))
        out["total_loss"].backward()
        torch.nn.utils.clip_grad_value_(model.parameters(), 0.1)                 # train_ncaltech101.py:61, clip: 0.1
        opt.step()
        losses.append(float(out["total_loss"]))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_train_script_on_the_hip_layers_then_evaluate_the_checkpoint(tmp_path):
    """scripts/train_ncaltech101.py for a few iterations on SyntheticObjects (dagr-s widths, one scale), then the
    checkpoint's ``ema`` state into a fresh model and an eval-mode forward through the window engine (run_test.py:54-62)."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "scripts"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    import train_ncaltech101 as T
    from dagr_amd.data import DataLoader
    from dagr_amd.data.synthetic_data import SyntheticObjects
    from dagr_amd.model.networks.dagr import DAGR
    from dagr_amd.model.networks.ema import ModelEMA
    out_dir, log = T.main(["--config", "dagr-s", "--epochs", "1", "--samples", "12", "--val_samples", "4", "--batch_size", "4",
                           "--n_nodes", "3000", "--output_directory", str(tmp_path), "--l_r", "0.002"])
    assert len(log) == 3 and all(np.isfinite(r["loss"]) for r in log)
    state = torch.load(out_dir / "last_model.pth", weights_only=False)
    assert state["ema_updates"] == 3
    ds = SyntheticObjects(4, 3000, seed=100007)
    model = DAGR(state["args"], height=ds.height, width=ds.width).cuda()
    ema = ModelEMA(model)
    ema.ema.load_state_dict(state["ema"])                                      # strict, run_test.py:57-58
    ema.ema.cache_luts(radius=state["args"].radius, height=ds.height, width=ds.width)
    batch = next(iter(DataLoader(ds, batch_size=4, follow_batch=["bbox"])))
    with torch.no_grad():
        detections, targets = ema.ema(format_data(batch.cuda()))
    assert len(detections) == 4 and len(targets) == 4 and all(torch.isfinite(d["boxes"]).all() for d in detections)


def test_training_with_the_image_branch_matches_the_oracle():
    """``--use_image`` training (train_dsec.py; dagr.py:197-222,241-268): ResNet-18 features sampled into the graph detached,
    CNN-head logits added detached, the image branch trained by its own ``get_losses`` against the earlier frame's boxes.
    HIP layers + PyTorch-ROCm image branch vs the oracle (pinned to the reference's training branch, train_s_img18_b2)."""
    from oracle import train as otr
    from tests.test_oracle_refpy import _image_branch_functional
    W, H, B, seed = 240, 180, 2, 5
    args, model, sd, batch, ev, b = _training_case(W, H, B, 2000, seed, use_image=True, img_net="resnet18")
    img = torch.randint(0, 256, (B, 3, H, W), generator=torch.Generator().manual_seed(seed), dtype=torch.uint8)
    batch.image = img
    batch.bbox0 = batch.bbox.clone()
    batch.bbox0[:, :2] -= 3.0
    batch.bbox0_batch = batch.bbox_batch.clone()
    from dagr_amd.model.networks.dagr import DAGR
    cpu_model = DAGR(args, height=H, width=W).train()          # module structure for functional_call; parameters come from sd
    image_feat, cnn_out = _image_branch_functional(cpu_model, sd, img.float() / 255.0, om.NetConstants(args, H, W),
                                                   args.num_scales)
    ref = otr.training_losses(sd, args, H, W, ev[0], ev[1], ev[2], ev[3], b, B, batch.bbox, batch.bbox_batch,
                              image_feat=image_feat, cnn_out=cnn_out, bbox0=batch.bbox0, bbox0_batch=batch.bbox0_batch)
    ref[0].backward()
    out = model(format_data(batch.cuda()))
    for k, r in zip(("total_loss", "iou_loss", "conf_loss", "cls_loss"), ref[:4]):
        assert abs(float(out[k]) - float(r)) <= 1e-3 * max(1.0, abs(float(r))), (k, float(out[k]), float(r))
    out["total_loss"].backward()
    params = dict(model.named_parameters())
    top = max(float(v.grad.abs().max()) for v in sd.values() if v.requires_grad and v.grad is not None)
    gnn, image, dot, na, nb = [], [], 0.0, 0.0, 0.0
    for k, v in sd.items():
        if v.requires_grad and v.grad is not None and float(v.grad.abs().max()) > 0:
            assert params[k].grad is not None, f"no gradient reached {k}"
            g = params[k].grad.cpu()
            # biases in front of a batch-statistics BatchNorm (output_dconv -> stems) have an analytically zero gradient:
            # both sides hold rounding noise there, so the bar has an absolute floor tied to the run's largest gradient
            err = float((g - v.grad).abs().max()) / (float(v.grad.abs().max()) + 1e-5 * top)
            if k.startswith("backbone.net.") or "cnn_head" in k:
                image.append((err, k))
                dot += float((g.double() * v.grad.double()).sum())
                na += float(g.double().pow(2).sum())
                nb += float(v.grad.double().pow(2).sum())
            else:
                gnn.append((err, k))
    assert len(gnn) >= 80 and len(image) >= 60, (len(gnn), len(image))
    # the graph network's gradients.  Its inputs now include features sampled from the image branch, which differ between
    # the two runs at the 1e-5 level (MIOpen vs CPU convolutions through batch-statistics BatchNorm2d); ReLU / max-pool
    # switches near ties then move single gradient entries by ~1 %.  The kernels themselves are held to 2e-3 by the
    # events-only cases above; here: the bulk within 5e-3, nothing beyond 5e-2.
    assert max(gnn)[0] < 5e-2, max(gnn)
    assert sum(e < 5e-3 for e, _ in gnn) / len(gnn) > 0.8, sorted(gnn)[-8:]
    # the image branch is PyTorch on both sides (MIOpen kernels here, CPU kernels in the oracle run); its batch-statistics
    # BatchNorm2d on two random-weight frames amplifies kernel-level rounding on a few deep tensors, so it is held to
    # the direction of the whole gradient and to the bulk of its tensors
    cos = dot / (na ** 0.5 * nb ** 0.5)
    close = sum(e < 2e-2 for e, _ in image) / len(image)
    assert cos > 0.995 and close > 0.8, (cos, close, sorted(image)[-5:])


def test_train_script_under_distributed_data_parallel_on_one_gpu(tmp_path):
    """The data-parallel wrapper on real device tensors: a one-rank RCCL process group, ``DistributedDataParallel`` around
    the model (bucketed reducer with gradient-as-bucket-view, frozen dense YOLOX lists, the custom autograd Functions
    of the HIP layers inside), three training steps.  Same losses as the unwrapped run (one rank: the all-reduce is the
    identity).  The 2-rank equivalence itself is the gloo test in tests/test_training_cpu.py."""
    import os
    import socket
    import sys
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "scripts"))
    import train_ncaltech101 as T
    argv = ["--config", "dagr-s", "--epochs", "1", "--samples", "12", "--val_samples", "4", "--batch_size", "4",
            "--n_nodes", "2500", "--l_r", "0.002"]
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "DAGR_FORCE_DDP"):
        os.environ.pop(k, None)
    _, plain = T.main(argv + ["--output_directory", str(tmp_path / "plain")])
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      DAGR_FORCE_DDP="1")
    try:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        _, wrapped = T.main(argv + ["--output_directory", str(tmp_path / "ddp")])
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
        for k in ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE", "LOCAL_RANK", "DAGR_FORCE_DDP"):
            os.environ.pop(k, None)
    assert len(plain) == len(wrapped) == 3
    assert abs(plain[0]["loss"] - wrapped[0]["loss"]) <= 1e-5 * max(1.0, abs(plain[0]["loss"]))
    # later steps: float atomics in the scatter gradients differ run to run and a discrete switch (SimOTA cost, arg-max
    # near tie) may then go the other way, so only the same overall course is required
    for a, b in zip(plain, wrapped):
        assert np.isfinite(b["loss"]) and abs(a["loss"] - b["loss"]) <= 0.25 * max(1.0, abs(a["loss"])), (a["loss"], b["loss"])
