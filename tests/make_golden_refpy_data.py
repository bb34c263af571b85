#!/usr/bin/env python
"""Golden outputs of the reference's OWN data-layer / training-helper code, executed on CPU (build container only:
it imports /root/reference/src and /root/reference/scripts).  Absent third-party modules are stubbed; where the
reference calls into them, a functional stand-in takes their place (tests/dsec_fixture.py for ``DSECDet``, nearest
resize for cv2, identity decorators for numba).  Pins, in tests/golden/ref_py_data.npz:

  * ``utils/learning_rate_scheduler.LRSchedule`` on a grid of iterations
  * ``data/dsec_utils``: construct_pairs, rescale / crop / map / size filter, compute_class_mapping, compute_iou,
    filter_tracks (with and without only_perfect_tracks)
  * ``data/dsec_data``: interpolate_tracks, and ``DSEC.__getitem__`` of every sample of the stand-in recordings with the
    test transform -- whole windows and ``set_num_us(20000)`` (interframe mode), eval and no_eval
  * ``data/augment``: the training chain ``Augmentations(args).transform_training`` under fixed torch seeds (events +
    boxes; frames need cv2), and the integrate-and-fire ``_subsample``
  * ``data/ncaltech101_data.NCaltech101``: class list, box decoding, event tail + time shift, on stand-in files
  * ``scripts/downsample_events.py``: ``downsample_events`` over two chunks with the carried change map, and the script's
    main loop over a 230 k-event recording (digests)
  * ``utils/buffers.py`` record helpers + ``DictBuffer``, ``scripts/run_test_interframe.py`` ``save_detections``
  * ``utils/logging.py`` ``Checkpointer``; ``utils/coco_eval.py`` ``_convert_to_coco_format`` (what reaches pycocotools)
  * ``data/dsec_utils._load_events`` over an h5py-shaped in-memory file

tests/test_data_refpy.py holds this repository's data layer (and oracle/downsample.py) to them."""
import argparse
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dsec_fixture  # noqa: E402
import refpy_fakes  # noqa: E402


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (), {"__init__": lambda self, *a, **k: None})
        setattr(self, name, cls)
        return cls


def _mod(name, **attrs):
    parts = name.split(".")
    for i in range(1, len(parts) + 1):
        n = ".".join(parts[:i])
        if n not in sys.modules:
            m = _Stub(n)
            m.__path__ = []
            sys.modules[n] = m
            if i > 1:
                setattr(sys.modules[".".join(parts[:i - 1])], parts[i - 1], m)
    for k, v in attrs.items():
        setattr(sys.modules[name], k, v)
    return sys.modules[name]


class _Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
        return data


class _BaseDirectory:
    def __init__(self, root):
        self.root = root


def install_stubs():
    ident = lambda *a, **k: (a[0] if len(a) == 1 and callable(a[0]) and not k else (lambda f: f))
    _mod("numba", njit=ident, jit=ident)
    _mod("cv2", resize=dsec_fixture.nearest_resize_hwc, INTER_NEAREST=0, INTER_CUBIC=2)
    _mod("h5py")
    _mod("hdf5plugin")
    _mod("tqdm", tqdm=lambda *a, **k: None)
    _mod("torch_geometric")
    _mod("torch_geometric.data", Data=refpy_fakes.Data, Dataset=type("Dataset", (), {"__init__": lambda self, *a, **k: None}))
    _mod("torch_geometric.transforms", BaseTransform=type("BaseTransform", (), {}), Compose=_Compose)
    _mod("dsec_det")
    _mod("dsec_det.dataset", DSECDet=dsec_fixture.FakeDSECDet)
    _mod("dsec_det.io", yaml_file_to_dict=lambda p: {"train": [], "val": [], "test": []})
    _mod("dsec_det.directory", BaseDirectory=_BaseDirectory)
    _mod("dagr_visualization_placeholder")


def import_ref(name):
    for _ in range(60):
        try:
            return importlib.import_module(name)
        except ModuleNotFoundError as e:
            _mod(e.name)
            for n in [m for m in sys.modules if m.startswith("dagr.") and getattr(sys.modules[m], "__file__", None) is None
                      and not isinstance(sys.modules[m], _Stub)]:
                del sys.modules[n]
    raise RuntimeError(name)


def stream_recording(n=230123, seed=21):
    """The synthetic 64 x 48 recording of the streaming-downsampler golden (shared with tests/test_data_refpy.py)."""
    g = np.random.default_rng(seed)
    hot = g.random(n) < 0.5                       # half of the events on a few busy pixels, so cells do fire
    x = np.where(hot, g.integers(20, 28, n), g.integers(0, 64, n)).astype(np.uint16)
    y = np.where(hot, g.integers(10, 16, n), g.integers(0, 48, n)).astype(np.uint16)
    p = (g.random(n) < np.where(hot, 0.8, 0.5)).astype(np.uint8)
    t = (5_000_000 + np.sort(g.integers(0, 900_000, n))).astype(np.int64)
    return dict(x=x, y=y, t=t, p=p)


def sample_fields(d):
    out = {"pos": d.pos.numpy(), "x": d.x.numpy(), "t": d.t.numpy(), "bbox": d.bbox.numpy()}
    if hasattr(d, "bbox0"):
        out["bbox0"] = d.bbox0.numpy()
    if hasattr(d, "image"):                   # frames are big: keep a strided sub-grid and the checksum
        out["image_grid"] = d.image.numpy()[..., ::16, ::16]
        out["image_sum"] = np.asarray(d.image.numpy().astype(np.int64).sum())
        out["image_shape"] = np.asarray(d.image.shape)
    for k in ("t0", "t1"):
        if hasattr(d, k):
            out[k] = np.asarray(getattr(d, k))
    return out


def main():
    install_stubs()
    refpy_fakes.use_reference_package("/root/reference/src")
    out = {}

    # ---- LRSchedule
    rlr = import_ref("dagr.utils.learning_rate_scheduler")
    its = np.array([0, 1, 7, 29, 30, 31, 100, 500, 777, 999, 1000, 49999, 50000, 60000])
    for k, kw in enumerate([dict(warmup_epochs=.3, num_iters_per_epoch=100, tot_num_epochs=801),
                            dict(warmup_epochs=1, num_iters_per_epoch=37, tot_num_epochs=40, min_lr_ratio=0.1,
                                 warmup_lr_start=0.2, steps_at_iteration=[500, 900], reduction_at_step=0.3)]):
        s = rlr.LRSchedule(**kw)
        out[f"lr{k}_iters"] = its
        out[f"lr{k}_vals"] = np.array([s(int(i)) for i in its])

    # ---- dsec_utils
    ru = import_ref("dagr.data.dsec_utils")
    rng = np.random.default_rng(5)
    idx = np.array([3, 4, 5, 9, 11, 12, 20, 21, 22, 23])
    out["pairs_in"], out["pairs2"], out["pairs3"] = idx, ru.construct_pairs(idx, 2), ru.construct_pairs(idx, 3)
    src = dsec_fixture.FakeDSECDet()
    tr = src.directories["zurich_city_12_a"].tracks.tracks
    out["tracks_rescaled"] = ru.rescale_tracks(tr, 2)
    out["tracks_cropped"] = ru.crop_tracks(ru.rescale_tracks(tr, 2), 320, 215)
    rdata = import_ref("dagr.data.dsec_data")
    mapping = ru.compute_class_mapping(("car", "pedestrian"), src.classes, rdata.DSEC.MAPPING)
    out["class_mapping"] = mapping
    ids, ok = ru.map_classes(tr["class_id"], mapping)
    out["mapped_ids"], out["mapped_ok"] = ids, ok
    c = out["tracks_cropped"]
    out["small_mask"] = ru.filter_small_bboxes(c["w"], c["h"], 15, 25)
    m = len(c) // 2
    out["iou"] = ru.compute_iou(c[:m], c[m:2 * m])
    for tag, kw in (("plain", {}), ("sized", dict(min_bbox_height=12, min_bbox_diag=20)),
                    ("perfect", dict(only_perfect_tracks=True))):
        pairs, masks = ru.filter_tracks(src, 320, 215, mapping, scale=2, **kw)
        for name in pairs:
            out[f"ft_{tag}_{name}_pairs"], out[f"ft_{tag}_{name}_mask"] = pairs[name], masks[name]

    # ---- dsec_utils._load_events over an h5py-shaped in-memory file (events/{x,y,t,p}, t_offset, ms_to_idx)
    g = np.random.default_rng(11)
    n = 20000
    t = np.sort(g.integers(0, 400000, n)).astype(np.int64)
    h5 = {"events/x": g.integers(0, 320, n).astype(np.uint16), "events/y": g.integers(0, 240, n).astype(np.uint16),
          "events/t": t, "events/p": g.integers(0, 2, n).astype(np.uint8), "t_offset": np.array(7_000_000, dtype=np.int64),
          "ms_to_idx": np.searchsorted(t, np.arange(0, 401) * 1000, side="left").astype(np.uint64)}

    class FakeFile:
        def __init__(self, *a, **k):
            pass

        def __enter__(self):
            return h5

        def __exit__(self, *exc):
            return False
    ru.h5py.File = FakeFile
    out.update({"h5_" + k.replace("/", "_"): v for k, v in h5.items()})
    for k, kw in enumerate([dict(num_events=3000, height=215, time_window=1000000), dict(num_us=-50000, time_window=1000000),
                            dict(num_us=30000, height=100, time_window=1000000), dict(num_events=-1500, time_window=1000000)]):
        (xy, tt, pp), tq = ru._load_events("unused", 7_000_000 + 200_000, **kw)
        out[f"h5w{k}_xy"], out[f"h5w{k}_t"], out[f"h5w{k}_p"], out[f"h5w{k}_tq"] = xy, tt, pp, np.asarray(tq)

    # ---- dsec_data: interpolate_tracks + DSEC.__getitem__
    f0 = src.get_tracks(2, None, "thun_01_a")
    f1 = src.get_tracks(3, None, "thun_01_a")
    common = np.intersect1d(f0["track_id"], f1["track_id"])
    f0, f1 = f0[np.isin(f0["track_id"], common)][::-1], f1[np.isin(f1["track_id"], common)]
    out["interp_f0"], out["interp_f1"] = f0, f1
    out["interp_out"] = rdata.interpolate_tracks(f0, f1, f0["t"][0] + 20000)
    raug = import_ref("dagr.data.augment")
    for tag, kw, num_us in (("full", {}, -1), ("us20k", dict(only_perfect_tracks=True), 20000), ("us20k_noeval", dict(no_eval=True), 20000),
                            ("sized", dict(min_bbox_height=12, min_bbox_diag=20), -1)):
        ds = rdata.DSEC(root="/dsec", split="test", transform=raug.Augmentations.transform_testing, demo=True, **kw)
        ds.set_num_us(num_us)
        out[f"dsec_{tag}_len"] = np.array(len(ds))
        for i in range(len(ds)):
            for k, v in sample_fields(ds[i]).items():
                out[f"dsec_{tag}_{i}_{k}"] = v
        print("DSEC", tag, "samples", len(ds), (ds.height, ds.width))

    # ---- training augmentations under fixed seeds (events + boxes)
    from dagr_amd.data.utils import to_data
    args = argparse.Namespace(aug_p_flip=0.5, aug_zoom=1.5, aug_trans=0.1)
    aug = raug.Augmentations(args)
    raug.init_transforms(aug.transform_training.transforms, 180, 240)
    rng = np.random.default_rng(9)
    N = 3000
    base = dict(x=rng.integers(0, 240, N), y=rng.integers(0, 180, N), t=np.sort(rng.integers(0, 50000, N)),
                p=rng.choice(np.array([-1, 1], dtype=np.int8), N),
                bbox=np.array([[50., 40, 60, 50, 3, 1], [120., 30, 90, 120, 1, 1]], dtype=np.float32))
    out.update({f"aug_base_{k}": v for k, v in base.items()})
    for seed in range(8):
        d = to_data(**{k: v.copy() for k, v in base.items()}, width=240, height=180, time_window=1000000)
        d = refpy_fakes.Data(**d.__dict__)
        torch.manual_seed(seed)
        o = aug.transform_training(d)
        for k in ("pos", "x", "t", "bbox"):
            out[f"aug{seed}_{k}"] = getattr(o, k).numpy()
    pos = np.stack([rng.uniform(0, 60, 4000), rng.uniform(0, 40, 4000)], 1).astype(np.float32)
    pol = np.ones((4000, 1), dtype=np.float32)
    pol[rng.random(4000) < 0.2] = -1
    mask = np.zeros(4000, dtype=bool)
    count = np.zeros((42, 62), dtype=np.float32)
    out["sub_pos_in"], out["sub_pol"] = pos.copy(), pol
    raug._subsample(pos, pol, mask, count, threshold=1 / 0.6 ** 2)
    out["sub_pos_out"], out["sub_mask"] = pos, mask

    # ---- NCaltech101 on stand-in files
    import tempfile
    rnc = import_ref("dagr.data.ncaltech101_data")
    rnc._load_events = lambda f_path, num_events: {k: np.load(open(f_path, "rb"))[k][-num_events:] for k in "xytp"}
    with tempfile.TemporaryDirectory() as tmp:
        from pathlib import Path
        tmp = Path(tmp)
        g = np.random.default_rng(0)
        for cls in ("airplanes", "zebra"):
            (tmp / "training" / cls).mkdir(parents=True)
            (tmp / "annotations" / cls).mkdir(parents=True)
            for k in (1, 2):
                n = 300 * k
                with open(tmp / "training" / cls / f"image_{k:04d}.h5", "wb") as fh:
                    np.savez(fh, x=g.integers(0, 240, n), y=g.integers(0, 180, n), t=np.sort(g.integers(0, 300000, n)),
                             p=g.integers(0, 2, n))
                np.array([0, 0, 10 + k, 20, 110, 20, 110, 90 + k, 10 + k, 90 + k, 0, 0], dtype=np.int16).tofile(
                    tmp / "annotations" / cls / f"annotation_{k:04d}.bin")
        ds = rnc.NCaltech101(tmp, "training", transform=None, num_events=250)
        out["nc_classes"] = np.array(ds.classes)
        for i in range(len(ds)):
            ev = {k: np.load(open(ds.files[i], "rb"))[k] for k in "xytp"}
            out.update({f"nc{i}_raw_{k}": v for k, v in ev.items()})
            d = ds[i]
            out[f"nc{i}_pos"], out[f"nc{i}_t"], out[f"nc{i}_x"], out[f"nc{i}_bbox"] = (d.pos.numpy(), d.t.numpy(), d.x.numpy(),
                                                                                         d.bbox.numpy())

    # ---- scripts/downsample_events.py
    spec = importlib.util.spec_from_file_location("ref_downsample", "/root/reference/scripts/downsample_events.py")
    rds = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rds)
    g = np.random.default_rng(3)
    cm = None
    for c in range(2):
        n = 6000
        ev = dict(x=g.integers(0, 64, n).astype(np.uint16), y=g.integers(0, 48, n).astype(np.uint16),
                  t=np.sort(g.integers(0, 100000, n)).astype(np.int64) + 100000 * c,
                  p=(2 * g.integers(0, 2, n) - 1).astype(np.int8))
        out.update({f"ds{c}_in_{k}": v for k, v in ev.items()})
        res, cm = rds.downsample_events(ev, 48, 64, 24, 32, change_map=cm)
        out.update({f"ds{c}_out_{k}": v for k, v in res.items()})
        out[f"ds{c}_change_map"] = cm.copy()

    # ---- record writers: utils/buffers.py bbox_t_to_ndarray / compile / DictBuffer, run_test_interframe.py to_npy / save_detections
    rbuf = import_ref("dagr.utils.buffers")
    g = np.random.default_rng(17)
    dets, seqs, stamps = [], [], []
    for i in range(7):
        n = int(g.integers(0, 6))
        x1y1 = g.uniform(0, 200, (n, 2)).astype(np.float32)
        boxes = np.concatenate([x1y1, x1y1 + g.uniform(5, 90, (n, 2)).astype(np.float32)], 1)
        dets.append(dict(boxes=torch.from_numpy(boxes), labels=torch.from_numpy(g.integers(0, 2, n)),
                         scores=torch.from_numpy(g.uniform(0, 1, n).astype(np.float32))))
        seqs.append(["zurich_city_12_a", "thun_01_a"][i % 2])
        stamps.append(int(50_000_000 + 50_000 * (6 - i)))           # decreasing: the writers must sort by time
    for i, d in enumerate(dets):
        for k, v in d.items():
            out[f"rec_in{i}_{k}"] = v.numpy()
    out["rec_seqs"], out["rec_stamps"] = np.array(seqs), np.array(stamps)
    out["rec_single"] = rbuf.bbox_t_to_ndarray(dets[1], stamps[1])
    out["rec_single_gt"] = rbuf.bbox_t_to_ndarray({k: v for k, v in dets[1].items() if k != "scores"}, stamps[1])
    comp = rbuf.compile(dets, seqs, stamps)
    for k, v in comp.items():
        out[f"rec_compiled_{k}"] = v
    db = rbuf.DictBuffer()
    for i in range(4):
        db.update({"a": float(i), "b": float(i * i)})
    out["dictbuffer"] = np.array([db.compute()["a"], db.compute()["b"]])
    _mod("wandb")
    spec = importlib.util.spec_from_file_location("ref_interframe", "/root/reference/scripts/run_test_interframe.py")
    rif = importlib.util.module_from_spec(spec)
    rif.__dict__["__name__"] = "ref_interframe"
    for _ in range(40):
        try:
            spec.loader.exec_module(rif)
            break
        except ModuleNotFoundError as e:
            _mod(e.name)
    rif.tqdm = types.SimpleNamespace(tqdm=lambda it, **k: it)
    rif.np = np                                   # the script imports numpy inside its __main__ block
    import tempfile as _tf
    from pathlib import Path as _P
    with _tf.TemporaryDirectory() as tmp:
        flat = [dict(boxes=d["boxes"].numpy(), labels=d["labels"].numpy(), scores=d["scores"].numpy(), sequence=s_, t=t_)
                for d, s_, t_ in zip(dets, seqs, stamps)]
        rif.save_detections(_P(tmp), flat)
        for f in sorted(_P(tmp).glob("*.npy")):
            out[f"rec_saved_{f.stem}"] = np.load(f)

    # ---- utils/logging.py Checkpointer: which file a directory of checkpoints resolves to, and the saved dict's keys
    rlog = import_ref("dagr.utils.logging")
    with _tf.TemporaryDirectory() as tmp:
        tmp = _P(tmp)
        lin = torch.nn.Linear(2, 2)
        opt = torch.optim.SGD(lin.parameters(), lr=0.1)
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda i: 1.0)
        ema = types.SimpleNamespace(ema=torch.nn.Linear(2, 2), updates=7)
        ck = rlog.Checkpointer(output_directory=tmp, model=lin, optimizer=opt, scheduler=sched, ema=ema, args={"x": 1})
        picks = {"empty": str(ck.search_for_checkpoint(tmp))}
        for name in ("best_model_mAP_0.125", "best_model_mAP_0.5", "best_model_mAP_0.25"):
            ck.checkpoint(3, name=name)
        picks["best_only_last"] = ck.search_for_checkpoint(tmp, best=False).name
        ck.checkpoint(9, name="last_model")
        picks["last"] = ck.search_for_checkpoint(tmp, best=False).name
        picks["best"] = ck.search_for_checkpoint(tmp, best=True).name
        out["ckpt_picks"] = np.array([picks[k] for k in ("empty", "best_only_last", "last", "best")])
        out["ckpt_keys"] = np.array(sorted(torch.load(tmp / "last_model.pth", weights_only=False)))
        out["ckpt_restored_epoch"] = np.array(ck.restore_checkpoint(tmp, best=False))

    # ---- utils/coco_eval.py: the reference's own half of the metric -- which images / boxes / ids reach COCO
    _mod("pycocotools"); _mod("pycocotools.coco"); _mod("detectron2"); _mod("detectron2.evaluation")
    _mod("detectron2.evaluation.fast_eval_api")
    rco = import_ref("dagr.utils.coco_eval")
    gts = []
    for i, d in enumerate(dets):                       # ground truth: a perturbed subset of the detections, two images empty
        keep = slice(0, 0) if i in (2, 5) else slice(0, max(1, len(d["boxes"]) - 1))
        gts.append(dict(boxes=d["boxes"][keep] + 1.5, labels=d["labels"][keep]))
    (dataset, results), n_img = rco._convert_to_coco_format(gts, dets, classes=("car", "pedestrian"), height=215, width=320)
    out["coco_n_images"] = np.array(n_img)
    out["coco_ann"] = np.array([[a["image_id"], a["category_id"], *a["bbox"], a["area"]] for a in dataset["annotations"]],
                               dtype=np.float64).reshape(-1, 7)
    out["coco_res"] = np.array([[r["image_id"], r["category_id"], *r["bbox"], r["score"]] for r in results],
                               dtype=np.float64).reshape(-1, 7)
    for i, gdict in enumerate(gts):
        out[f"coco_gt{i}_boxes"], out[f"coco_gt{i}_labels"] = gdict["boxes"].numpy(), gdict["labels"].numpy()

    # the script's own main loop (:139-167) over a 230 123-event recording in chunks of 100 000: every full chunk with
    # p -> {-1, +1}, the trailing partial chunk as it is read (p in {0, 1}); then the writer's casts and ms_to_idx.
    # Input re-drawn from the seed by the test; outputs stored as digests + a few probes.
    import hashlib
    ev_all = stream_recording()
    n, chunk = len(ev_all["t"]), 100000
    cm, kept = None, []
    for i in range(n // chunk):
        ev = {k: v[i * chunk:(i + 1) * chunk].copy() for k, v in ev_all.items()}
        ev["p"] = 2 * ev["p"].astype("int8") - 1
        res, cm = rds.downsample_events(ev, 48, 64, 24, 32, change_map=cm)
        kept.append(res)
    ev = {k: v[(n // chunk) * chunk:].copy() for k, v in ev_all.items()}
    res, cm = rds.downsample_events(ev, 48, 64, 24, 32, change_map=cm)
    kept.append(res)
    t_offset = kept[0]["t"][0]
    cat = dict(x=np.concatenate([r["x"] for r in kept]).astype("u2"), y=np.concatenate([r["y"] for r in kept]).astype("u2"),
               p=np.clip(np.concatenate([r["p"].astype(np.int64) for r in kept]), 0, 255).astype("u1"),   # HDF5 clamps -1 -> 0
               t=(np.concatenate([r["t"] for r in kept]) - t_offset).astype("u4"))
    ms = rds.create_ms_to_idx(cat["t"])
    out["stream_count"] = np.array(len(cat["t"]))
    out["stream_t_offset"] = np.array(t_offset)
    for k, v in list(cat.items()) + [("ms_to_idx", ms)]:
        out[f"stream_sha_{k}"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(v).tobytes()).digest(), dtype=np.uint8)
        out[f"stream_head_{k}"] = v[:64]
    print("stream:", n, "->", len(cat["t"]), "events; tail chunk kept", len(kept[-1]["t"]))

    path = os.path.join(os.environ.get("GOLDEN_OUT", os.path.join(ROOT, "tests", "golden")), "ref_py_data.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
