"""CPU suite of the training path's host side: the YOLOX loss restatement on hand-computable cases, target formatting,
the learning-rate schedule, the checkpointer, the training augmentations, the N-Caltech101 reader on stand-in files, and
``scripts/train_ncaltech101.py`` end to end with world_size 2 on gloo (stand-in model: the HIP layers need a GPU) --
data-parallel slices of every global batch + DistributedDataParallel must reproduce the single-process run."""
import math
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dagr_amd.model.networks import yolox_loss as yl


def test_iou_loss_and_pairwise_iou_hand_cases():
    a = torch.tensor([[10.0, 10.0, 4.0, 4.0]])
    b = torch.tensor([[10.0, 10.0, 4.0, 4.0], [12.0, 10.0, 4.0, 4.0], [30.0, 30.0, 2.0, 2.0]])
    iou = yl.pairwise_iou_cxcywh(a, b)
    assert torch.allclose(iou, torch.tensor([[1.0, 8.0 / 24.0, 0.0]]))
    loss = yl.iou_loss(b, a.expand(3, -1))
    assert torch.allclose(loss, torch.tensor([0.0, 1 - (1 / 3) ** 2, 1.0]), atol=1e-6)


def test_output_and_grid_decodes_like_the_eval_branch():
    torch.manual_seed(0)
    raw = torch.randn(2, 7, 3, 4)
    out, grid = yl.output_and_grid(raw, 16)
    assert out.shape == (2, 12, 7) and grid.shape == (1, 12, 2)
    # cell (row 1, col 2) is anchor 1*4+2
    assert grid[0, 6].tolist() == [2.0, 1.0]
    assert torch.allclose(out[1, 6, :2], (raw[1, :2, 1, 2] + torch.tensor([2.0, 1.0])) * 16)
    assert torch.allclose(out[1, 6, 2:4], torch.exp(raw[1, 2:4, 1, 2]) * 16)
    assert torch.equal(out[1, 6, 4:], raw[1, 4:, 1, 2])


def _anchors(h, w, stride):
    yv, xv = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    g = torch.stack((xv, yv), 2).view(-1, 2).float()
    return (g + 0.5) * stride, torch.full((h * w,), float(stride))


def test_simota_single_gt_picks_the_cells_under_the_box():
    centers, strides = _anchors(5, 7, 32)
    gt = torch.tensor([[3.5 * 32, 2.5 * 32, 60.0, 60.0]])            # centred on cell (row 2, col 3)
    boxes = torch.cat([centers, torch.full((35, 2), 60.0)], 1)        # every anchor predicts a 60x60 box at its centre
    cls = torch.zeros(35, 2)
    obj = torch.zeros(35, 1)
    fg, mgt, miou = yl.simota_assign(gt, torch.tensor([1.0]), boxes, cls, obj, centers, strides, 2)
    # IoU is 1 at the centre cell, (60-32)*60 / (2*3600 - (60-32)*60) = 0.304 for the 4 direct neighbours:
    # dynamic k = int(1 + 4 * 0.304 + 4 * 0.122 + ...) = 2 -> the centre cell and the cheapest neighbour
    assert int(fg.sum()) == int(miou.numel()) == 2
    assert bool(fg[2 * 7 + 3])
    assert torch.all(mgt == 0)
    assert float(miou.max()) == pytest.approx(1.0)


def test_simota_contested_anchor_goes_to_the_cheaper_ground_truth():
    centers, strides = _anchors(5, 7, 32)
    gt = torch.tensor([[3.5 * 32, 2.5 * 32, 40.0, 40.0], [3.5 * 32 + 4, 2.5 * 32, 40.0, 40.0]])
    boxes = torch.cat([centers, torch.full((35, 2), 40.0)], 1)
    fg, mgt, miou = yl.simota_assign(gt, torch.tensor([0.0, 1.0]), boxes, torch.zeros(35, 2), torch.zeros(35, 1), centers,
                                     strides, 2)
    c = 2 * 7 + 3
    assert bool(fg[c])
    pos = int(fg[:c].sum())
    assert int(mgt[pos]) == 0        # the first box is centred exactly on that anchor: higher IoU, lower cost
    assert len(set(fg.nonzero().flatten().tolist())) == int(fg.sum())


def test_detection_losses_are_zero_for_a_perfect_prediction_and_differentiable():
    h, w, stride, C = 5, 7, 32, 3
    centers, _ = _anchors(h, w, stride)
    labels = torch.zeros(2, 100, 5)
    labels[0, 0] = torch.tensor([2.0, 3.5 * 32, 2.5 * 32, 40.0, 40.0])
    raw = torch.zeros(2, 5 + C, h, w)
    raw[:, 2:4] = math.log(40.0 / 32)
    raw[:, 4] = -20.0                                   # objectness off everywhere ...
    raw[:, 5:] = -20.0
    raw[0, 4, 2, 3] = 20.0                              # ... except the matching cell, which also predicts class 2
    raw[0, 7, 2, 3] = 20.0
    raw.requires_grad_(True)
    out, grid = yl.output_and_grid(raw, stride)
    # cell (2,3): xy decode = (grid + 0) * stride = cell corner, so move the target onto the decoded position
    labels[0, 0, 1:3] = torch.tensor([3.0 * 32, 2.0 * 32])
    total, l_iou, l_obj, l_cls, l_l1, ratio = yl.detection_losses(labels, out, [grid], [stride], C)
    assert float(l_iou) == pytest.approx(0.0, abs=1e-6)
    assert float(l_obj) < 1e-6 and float(l_cls) < 1e-6 and l_l1 == 0.0
    assert ratio == 1.0
    total.backward()
    assert torch.isfinite(raw.grad).all()
    # a sample without boxes contributes only objectness
    labels2 = torch.zeros(2, 100, 5)
    t2 = yl.detection_losses(labels2, out.detach(), [grid], [stride], C)
    assert float(t2[1]) == 0.0 and float(t2[3]) == 0.0 and t2[5] == 1.0   # max(num_fg, 1) / max(num_gts, 1), as YOLOX reports it


def test_lr_schedule_shape():
    from dagr_amd.utils.learning_rate_scheduler import LRSchedule
    s = LRSchedule(warmup_epochs=0.3, num_iters_per_epoch=100, tot_num_epochs=10, steps_at_iteration=[700])
    assert s(0) == 0.0 and s(15) == pytest.approx(0.25) and s(30) == pytest.approx(1.0)
    mid = 30 + (1000 - 30) / 2
    assert s(mid) == pytest.approx(0.05 + 0.5 * 0.95)
    assert s(699) > 2 * s(700) * 0.99 and s(1000) == pytest.approx(0.05 * 0.5)
    assert all(s(i) >= s(i + 1) for i in range(30, 699))


def test_checkpointer_round_trip(tmp_path):
    from dagr_amd.model.networks.ema import ModelEMA
    from dagr_amd.utils.logging import Checkpointer
    torch.manual_seed(0)
    model = torch.nn.Linear(3, 2)
    ema = ModelEMA(model)
    opt = torch.optim.AdamW(model.parameters(), lr=0.1)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda i: 1.0 / (1 + i))
    ck = Checkpointer(output_directory=tmp_path, model=model, optimizer=opt, scheduler=sched, ema=ema, args={"a": 1})
    model(torch.ones(1, 3)).sum().backward()
    opt.step(); sched.step(); ema.update(model)
    ck.checkpoint(4, name="last_model")
    ck.process({"mAP": 0.25}, 4)
    ck.process({"mAP": 0.125}, 5)
    assert sorted(p.name for p in tmp_path.glob("*.pth")) == ["best_model_mAP_0.25.pth", "last_model.pth"]
    want = {k: v.clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        model.weight.zero_()
    ema.updates = 0
    ck2 = Checkpointer(output_directory=tmp_path, model=model, optimizer=opt, scheduler=sched, ema=ema)
    assert ck2.restore_if_existing(tmp_path) == 4
    assert all(torch.equal(model.state_dict()[k], v) for k, v in want.items()) and ema.updates == 1
    assert ck2.search_for_checkpoint(tmp_path, best=True).name == "best_model_mAP_0.25.pth"
    assert set(torch.load(tmp_path / "last_model.pth", weights_only=False)) == {"ema", "ema_updates", "model", "optimizer",
                                                                                 "scheduler", "epoch", "args"}


def _sample(n=4000, W=240, H=180, seed=0):
    from dagr_amd.data.utils import to_data
    rng = np.random.default_rng(seed)
    d = to_data(x=rng.integers(0, W, n), y=rng.integers(0, H, n), t=np.sort(rng.integers(0, 50000, n)),
                p=rng.choice(np.array([-1, 1], dtype=np.int8), n), width=W, height=H, time_window=1000000,
                bbox=np.array([[50., 40, 60, 50, 3, 1]], dtype=np.float32))
    d.image = torch.randint(0, 255, (1, 3, H, W), dtype=torch.uint8)
    return d


def test_training_augmentations_keep_the_sample_consistent():
    from dagr_amd.data import augment as A
    torch.manual_seed(1)
    d = _sample()
    flipped = A.RandomHFlip(p=1.0)(d.clone())
    assert torch.equal(flipped.pos[:, 0], 239 - d.pos[:, 0]) and torch.equal(flipped.pos[:, 1], d.pos[:, 1])
    assert flipped.bbox[0].tolist() == [239 - 110.0, 40, 60, 50, 3, 1]
    assert torch.equal(flipped.image, torch.flip(d.image, dims=[-1]))
    z = A.RandomZoom([2.0, 2.0])(d.clone())                                       # exact factor 2 about (120, 90)
    assert torch.equal(z.pos[:, 0].long(), (d.pos[:, 0].long() - 120) * 2 + 120)
    assert z.bbox[0, :4].tolist() == [(50 - 120) * 2 + 120, (40 - 90) * 2 + 90, 120, 100]
    assert torch.equal(z.image[0, :, 90, 120], d.image[0, :, 90, 120])            # the centre pixel stays
    tr = A.RandomTranslate([0.1, 0.1, 0])
    tr.init(180, 240)
    t = tr(d.clone())
    move = (t.pos[0] - d.pos[0]).tolist()
    assert abs(move[0]) <= 24 and abs(move[1]) <= 18 and torch.equal(t.pos - d.pos, (t.pos - d.pos)[0].expand_as(d.pos))
    assert (t.bbox[0, :2] - d.bbox[0, :2]).tolist() == move
    y, x = 60, 100
    if 0 <= y + move[1] < 180 and 0 <= x + move[0] < 240:
        assert torch.equal(t.image[0, :, y + move[1], x + move[0]], d.image[0, :, y, x])
    aug = A.Augmentations(types.SimpleNamespace(aug_p_flip=0.5, aug_zoom=1.5, aug_trans=0.1))
    A.init_transforms(aug.transform_training.transforms, 180, 240)
    for _ in range(8):
        o = aug.transform_training(d.clone())
        n = o.pos.shape[0]
        assert o.pos.dtype == torch.int16 and o.x.shape == (n, 1) and o.t.shape == (n,) and 0 < n <= 4000
        assert int(o.pos[:, 0].min()) >= 0 and int(o.pos[:, 0].max()) < 240 and int(o.pos[:, 1].max()) < 180
        assert bool((o.t[1:] >= o.t[:-1]).all())                                  # time order survives every transform
        bx = o.bbox[0]
        assert 0 <= float(bx[0]) <= 239 and 0 <= float(bx[1]) <= 179 and float(bx[0] + bx[2]) <= 239.001
        assert o.image.shape == (1, 3, 180, 240)


def test_subsample_is_an_integrate_and_fire_per_pixel():
    from dagr_amd.data.augment import subsample_events
    # 5 positive events on the same spot at zoom 0.5: threshold 4 -> the 5th crossing... the accumulator of pixel (3, 2)
    # receives 1 per event, fires on the event that lifts it above 4 (strictly), i.e. the fifth
    pos = np.tile(np.array([[3.0, 2.0]]), (6, 1))
    out, keep = subsample_events(pos, np.ones(6), 0.5)
    assert keep.tolist() == [False, False, False, False, True, False]
    assert out[4].tolist() == [3.0, 2.0]


def test_ncaltech101_reader_on_stand_in_files(tmp_path):
    from dagr_amd.data.ncaltech101_data import NCaltech101
    rng = np.random.default_rng(0)
    for split in ("training",):
        for cls in ("airplanes", "zebra"):
            (tmp_path / split / cls).mkdir(parents=True)
            (tmp_path / "annotations" / cls).mkdir(parents=True, exist_ok=True)
            for k in (1, 2):
                n = 300 * k
                np.savez(tmp_path / split / cls / f"image_{k:04d}.npz", x=rng.integers(0, 240, n), y=rng.integers(0, 180, n),
                         t=np.sort(rng.integers(0, 300000, n)), p=rng.integers(0, 2, n))
                words = np.array([0, 0, 10, 20, 110, 20, 110, 90, 10, 90, 0, 0], dtype=np.int16)   # corners, clockwise
                words.tofile(tmp_path / "annotations" / cls / f"annotation_{k:04d}.bin")
    ds = NCaltech101(tmp_path, "training", transform=None, num_events=250, reader=lambda p: np.load(p), suffix=".npz")
    assert ds.classes == ["airplanes", "zebra"] and len(ds) == 4 and (ds.height, ds.width) == (180, 240)
    d = ds[3]                                                   # zebra / image_0002: 600 events, the last 250 kept
    assert d.pos.shape == (250, 2) and d.pos.dtype == torch.int16 and d.t.dtype == torch.int32
    assert int(d.t[-1]) == 1000000 - 1                          # newest event at T - 1 (preprocess)
    assert d.bbox.tolist() == [[10.0, 20.0, 100.0, 70.0, 1.0, 1.0]]
    with pytest.raises(RuntimeError, match="h5py"):
        NCaltech101(tmp_path, "training", suffix=".npz")[0]


# ---------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _StandInDetector(torch.nn.Module):
    """Same call contract as ``DAGR`` in training mode (Data batch -> loss dict): a two-layer regressor from per-sample
    event statistics to the box; the loss is a mean over the samples, so equal slices average to the global batch."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(5)
        self.body = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4))
        self.head = torch.nn.Module()
        self.head.stems = torch.nn.Linear(2, 2)          # never used, like the dense YOLOX lists inside GNNHead

    def forward(self, data):
        B = data.num_graphs
        feats, tgt = [], []
        for i in range(B):
            m = data.batch == i
            p = data.pos[m][:, :2].double()
            feats.append(torch.cat([p.mean(0), p.std(0), p.min(0).values]).float())
            tgt.append(data.bbox[data.bbox_batch == i][0, :4] / 240.0)
        pred = self.body(torch.stack(feats))
        loss = ((pred - torch.stack(tgt)) ** 2).sum(1).mean()
        return {"total_loss": loss, "iou_loss": loss.detach(), "num_fg": 1.0}


def _train_worker(rank, world, port, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "scripts"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import train_ncaltech101 as T
    _, log = T.main(["--epochs", "2", "--samples", "16", "--val_samples", "8", "--batch_size", "4", "--n_nodes", "400",
                     "--output_directory", out_dir, "--l_r", "0.01"], model_factory=lambda args, ds: _StandInDetector())
    torch.save(log, os.path.join(out_dir, f"log{rank}.pt"))


def test_train_script_data_parallel_equals_single_process(tmp_path):
    outs = {}
    for world in (1, 2):
        out = str(tmp_path / f"w{world}")
        os.makedirs(out)
        if world == 1:
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
                os.environ.pop(k, None)
            mp.spawn(_train_worker, args=(1, 0, out), nprocs=1, join=True)
        else:
            mp.spawn(_train_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        outs[world] = torch.load(os.path.join(out, "ncaltech101", "detection", "train", "last_model.pth"), weights_only=False)
    a, b = outs[1], outs[2]
    assert a["epoch"] == b["epoch"] == 1 and a["ema_updates"] == b["ema_updates"] == 8      # 2 epochs x 4 global batches
    for k in a["model"]:
        assert torch.allclose(a["model"][k], b["model"][k], atol=1e-6), k                   # averaged slices == global batch
        assert torch.allclose(a["ema"][k], b["ema"][k], atol=1e-6), k
    l1 = torch.load(tmp_path / "w1" / "log0.pt", weights_only=False)
    l2 = [torch.load(tmp_path / "w2" / f"log{r}.pt", weights_only=False) for r in (0, 1)]
    assert len(l1) == len(l2[0]) == 8
    # the global loss of step k is the mean of the two slice losses
    for k in range(8):
        assert l1[k]["loss"] == pytest.approx(0.5 * (l2[0][k]["loss"] + l2[1][k]["loss"]), rel=1e-5)
    assert l1[-1]["loss"] < l1[0]["loss"]
    best = [p for p in os.listdir(os.path.join(str(tmp_path / "w2"), "ncaltech101", "detection", "train")) if "best" in p]
    assert best, "validation pass did not record a best checkpoint"


@pytest.mark.parametrize("seed", range(12))
def test_product_loss_equals_the_literal_oracle_restatement(seed):
    """dagr_amd/model/networks/yolox_loss.py (vectorised) against oracle/yolox_loss.py (the published YOLOX code path,
    loops and all) on random head maps: same assignment, same six outputs, same gradient."""
    from oracle.yolox_loss import LossHead
    g = torch.Generator().manual_seed(100 + seed)
    B, C = 3, (2 if seed % 2 else 5)
    shapes, strides = ([(10, 14), (5, 7)], [22, 43]) if seed % 3 else ([(5, 7)], [43])
    maps = [torch.randn(B, 5 + C, h, w, generator=g) * 1.5 for h, w in shapes]
    labels = torch.zeros(B, 100, 5)
    for b in range(B):
        for k in range(int(torch.randint(0, 4, (1,), generator=g))):
            cx, cy = torch.rand(2, generator=g) * torch.tensor([300.0, 200.0])
            wh = 15 + torch.rand(2, generator=g) * 120
            labels[b, k] = torch.tensor([float(torch.randint(0, C, (1,), generator=g)), cx, cy, wh[0], wh[1]])
    ref_maps = [m.clone().requires_grad_(True) for m in maps]
    ref = LossHead(C, len(shapes)).losses_from_maps([m * 1 for m in ref_maps], strides, labels)   # (cat output: not a leaf)
    my_maps = [m.clone().requires_grad_(True) for m in maps]
    outs, grids = zip(*(yl.output_and_grid(m, s) for m, s in zip(my_maps, strides)))
    mine = yl.detection_losses(labels, torch.cat(outs, 1), list(grids), strides, C)
    for a, b in zip(mine, ref):
        assert float(a) == pytest.approx(float(b), rel=1e-5, abs=1e-6)
    ref[0].backward()
    mine[0].backward()
    for a, b in zip(my_maps, ref_maps):
        assert torch.allclose(a.grad, b.grad, rtol=1e-4, atol=1e-6)


def test_train_dsec_script_preset_on_cpu(tmp_path):
    """scripts/train_dsec.py (the DSEC preset of the training script): two head scales, flip / zoom / translate
    augmentations on events + frames + both box sets, ``--use_image`` samples (synthetic frames, bbox0) through the loader,
    the loop, the validation pass and the checkpointer -- with a stand-in model (the HIP layers need a GPU)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "scripts"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    import train_dsec
    seen = {}

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.l = torch.nn.Linear(2, 1)

        def forward(self, data):
            assert tuple(data.image.shape[1:]) == (3, 215, 320) and data.image.dtype == torch.float32
            assert float(data.image.max()) <= 1.0
            assert data.bbox.shape[0] == data.bbox0.shape[0] and hasattr(data, "bbox0_batch")
            assert int(data.pos[:, 0].max() * 320 + 0.5) < 320 and data.pos.shape[1] == 3
            seen["n"] = seen.get("n", 0) + 1
            return {"total_loss": (self.l(data.pos[:, :2]).mean() - 0.3) ** 2, "num_fg": 1.0}

    def factory(args, ds):
        assert args.dataset == "dsec" and args.num_scales == 2 and args.use_image and args.aug_zoom == 1.5
        assert (ds.height, ds.width) == (215, 320)
        return M()
    out, log = train_dsec.main(["--epochs", "1", "--samples", "8", "--val_samples", "4", "--batch_size", "4", "--n_nodes", "500",
                                "--use_image", "--config", "dagr-s", "--output_directory", str(tmp_path)], model_factory=factory)
    assert len(log) == 2 and seen["n"] == 3                      # two training batches + one validation batch
    assert sorted(p.name.split("_")[0] for p in out.glob("*.pth")) == ["best", "last"]
    assert out.parts[-3:] == ("dsec", "detection", "train")


@pytest.mark.parametrize("seed", range(40))
def test_batched_simota_equals_the_per_image_form(seed):
    """``simota_assign_batch`` (one masked pass over the padded label rows, no host synchronisation) gives every image the
    assignment ``simota_assign`` gives it alone: same foreground anchors, same matched ground truths, same IoUs."""
    g = torch.Generator().manual_seed(7000 + seed)
    B = int(torch.randint(1, 5, (1,), generator=g))
    C = [2, 5, 100][seed % 3]
    shapes, strides_ = ([(10, 14), (5, 7)], [22, 43]) if seed % 2 else ([(5, 7)], [43])
    maps = [torch.randn(B, 5 + C, h, w, generator=g) * 1.5 for h, w in shapes]
    labels = torch.zeros(B, 100, 5)
    for b in range(B):
        n = int(torch.randint(0, 9, (1,), generator=g)) if seed % 7 else 0          # images without ground truth too
        for k in range(n):
            cx, cy = torch.rand(2, generator=g) * torch.tensor([300.0, 200.0])
            wh = 2 + torch.rand(2, generator=g) * (150 if seed % 5 else 10)        # tiny boxes: no candidate anchors
            labels[b, k] = torch.tensor([float(torch.randint(0, C, (1,), generator=g)), cx, cy, wh[0], wh[1]])
    outs, grids = zip(*(yl.output_and_grid(m, s) for m, s in zip(maps, strides_)))
    out = torch.cat(outs, 1)
    grid = torch.cat(grids, 1)[0]
    stride = torch.cat([torch.full((gg.shape[1],), float(s)) for gg, s in zip(grids, strides_)])
    centers = (grid + 0.5) * stride[:, None]
    box, obj, cls = out[..., :4], out[..., 4:5], out[..., 5:]
    fg, mg, mi = yl.simota_assign_batch(labels, box, cls, obj, centers, stride, C)
    for b in range(B):
        G = int((labels[b].sum(1) > 0).sum())
        if G == 0:
            assert not fg[b].any()
            continue
        f1, g1, i1 = yl.simota_assign(labels[b, :G, 1:5], labels[b, :G, 0], box[b], cls[b], obj[b], centers, stride, C)
        assert torch.equal(f1, fg[b]) and torch.equal(g1, mg[b][fg[b]]) and torch.equal(i1, mi[b][fg[b]])


def test_ema_update_is_the_per_entry_recurrence():
    """ModelEMA.update (two multi-tensor launches over cached tensor lists) == avg = d * avg + (1 - d) * value per
    floating-point state_dict entry (ema.py:33-45), also after the model was moved (its storage re-allocated)."""
    from dagr_amd.model.networks.ema import ModelEMA
    torch.manual_seed(3)
    m = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.BatchNorm1d(7), torch.nn.Linear(7, 3))
    e = ModelEMA(m)
    ref = {k: v.clone() for k, v in e.ema.state_dict().items()}
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    for it in range(6):
        m(torch.randn(4, 5)).sum().backward()
        opt.step()
        opt.zero_grad()
        if it == 3:
            m = m.double().float()          # new parameter storage: the cached lists must notice
            opt = torch.optim.SGD(m.parameters(), lr=0.1)
        e.update(m)
        d = e.decay(e.updates)
        for k, v in m.state_dict().items():
            if ref[k].dtype.is_floating_point:
                ref[k].mul_(d).add_(v.detach(), alpha=1 - d)
    for k, v in e.ema.state_dict().items():
        if v.dtype.is_floating_point:
            assert torch.equal(ref[k], v), k
