"""CPU suite: this repository's data layer and training helpers against golden outputs of the reference's OWN code
(tests/make_golden_refpy_data.py -> tests/golden/ref_py_data.npz; the reference's modules imported from /root/reference
in the build container with stubs for absent third-party packages).  Nothing here reads /root/reference."""
import os
import types

import numpy as np
import pytest
import torch

from tests import dsec_fixture

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_py_data.npz"), allow_pickle=False)


def _same_struct(a, b):
    assert a.dtype == b.dtype and a.shape == b.shape
    for name in a.dtype.names:
        assert np.array_equal(a[name], b[name]), name


def test_lr_schedule_values():
    from dagr_amd.utils.learning_rate_scheduler import LRSchedule
    for k, kw in enumerate([dict(warmup_epochs=.3, num_iters_per_epoch=100, tot_num_epochs=801),
                            dict(warmup_epochs=1, num_iters_per_epoch=37, tot_num_epochs=40, min_lr_ratio=0.1,
                                 warmup_lr_start=0.2, steps_at_iteration=[500, 900], reduction_at_step=0.3)]):
        s = LRSchedule(**kw)
        got = np.array([s(int(i)) for i in G[f"lr{k}_iters"]])
        assert np.allclose(got, G[f"lr{k}_vals"], rtol=1e-13, atol=0)


def test_dsec_track_utilities():
    from dagr_amd.data import dsec_utils as U
    from dagr_amd.data.dsec_data import MAPPING
    assert np.array_equal(U.construct_pairs(G["pairs_in"], 2), G["pairs2"])
    assert np.array_equal(U.construct_pairs(G["pairs_in"], 3), G["pairs3"])
    src = dsec_fixture.FakeDSECDet()
    tr = src.directories["zurich_city_12_a"].tracks.tracks
    _same_struct(U.rescale_tracks(tr, 2), G["tracks_rescaled"])
    cropped = U.crop_tracks(U.rescale_tracks(tr, 2), 320, 215)
    _same_struct(cropped, G["tracks_cropped"])
    mapping = U.compute_class_mapping(("car", "pedestrian"), src.classes, MAPPING)
    assert np.array_equal(mapping, G["class_mapping"])
    ids, ok = U.map_classes(tr["class_id"], mapping)
    assert np.array_equal(ids, G["mapped_ids"]) and np.array_equal(ok, G["mapped_ok"])
    assert np.array_equal(U.filter_small_bboxes(cropped["w"], cropped["h"], 15, 25), G["small_mask"])
    m = len(cropped) // 2
    assert np.array_equal(U.box_iou(cropped[:m], cropped[m:2 * m]), G["iou"])
    for tag, kw in (("plain", {}), ("sized", dict(min_bbox_height=12, min_bbox_diag=20)),
                    ("perfect", dict(only_perfect_tracks=True))):
        pairs, masks = U.filter_tracks(src, 320, 215, mapping, scale=2, **kw)
        for name in pairs:
            assert np.array_equal(pairs[name], G[f"ft_{tag}_{name}_pairs"]), (tag, name)
            assert np.array_equal(masks[name], G[f"ft_{tag}_{name}_mask"]), (tag, name)
    assert len(G["ft_perfect_thun_01_a_pairs"]) < len(G["ft_plain_thun_01_a_pairs"])     # the filter does something


def test_interpolate_tracks():
    from dagr_amd.data.dsec_data import interpolate_tracks
    out = interpolate_tracks(G["interp_f0"], G["interp_f1"], G["interp_f0"]["t"][0] + 20000)
    _same_struct(out, G["interp_out"])


def _nearest(image, width, height):
    return torch.from_numpy(np.ascontiguousarray(dsec_fixture.nearest_resize_hwc(image, (width, height)))).permute(2, 0, 1)[None]


@pytest.mark.parametrize("tag,kw,num_us", [("full", {}, -1), ("us20k", dict(only_perfect_tracks=True), 20000),
                                           ("us20k_noeval", dict(no_eval=True), 20000),
                                           ("sized", dict(min_bbox_height=12, min_bbox_diag=20), -1)])
def test_dsec_samples_match_the_reference_class(tag, kw, num_us):
    """Every sample of the stand-in recordings through ``DSEC.__getitem__`` (frame-pair selection, class remap, box
    rescale / clip, event crop + time shift + polarity, interframe cut with box interpolation, the test transform, the
    small-box filter) == the reference's DSEC class on the same recordings."""
    from dagr_amd.data.augment import Augmentations
    from dagr_amd.data.dsec_data import DSEC
    ds = DSEC(source=dsec_fixture.FakeDSECDet(), transform=Augmentations.transform_testing, resize=_nearest, **kw)
    ds.set_num_us(num_us)
    assert (ds.height, ds.width) == (215, 320) and len(ds) == int(G[f"dsec_{tag}_len"])
    for i in range(len(ds)):
        d = ds[i]
        pre = f"dsec_{tag}_{i}_"
        assert d.pos.dtype == torch.int16 and d.t.dtype == torch.int32 and d.x.dtype == torch.int8
        for k in ("pos", "x", "t", "bbox", "bbox0"):
            got = getattr(d, k).numpy()
            assert got.shape == G[pre + k].shape and np.array_equal(got, G[pre + k]), (i, k)
        assert int(d.t0) == int(G[pre + "t0"]) and int(d.t1) == int(G[pre + "t1"])
        assert tuple(d.image.shape) == tuple(G[pre + "image_shape"])
        assert int(d.image.numpy().astype(np.int64).sum()) == int(G[pre + "image_sum"])
        assert np.array_equal(d.image.numpy()[..., ::16, ::16], G[pre + "image_grid"])
        if len(d.t):
            assert int(d.t[-1]) == 1000000


def test_training_augmentation_chain_under_fixed_seeds():
    """``Augmentations(args).transform_training`` (flip, random crop, zoom, translate, crop) draws from torch's global RNG:
    under the same seed the chain here reproduces the reference's chain exactly -- events, polarities, times and boxes."""
    from dagr_amd.data import Data
    from dagr_amd.data.augment import Augmentations, init_transforms
    from dagr_amd.data.utils import to_data
    aug = Augmentations(types.SimpleNamespace(aug_p_flip=0.5, aug_zoom=1.5, aug_trans=0.1))
    init_transforms(aug.transform_training.transforms, 180, 240)
    base = {k: G[f"aug_base_{k}"] for k in ("x", "y", "t", "p", "bbox")}
    seen = set()
    for seed in range(8):
        d = to_data(**{k: v.copy() for k, v in base.items()}, width=240, height=180, time_window=1000000)
        torch.manual_seed(seed)
        o = aug.transform_training(d)
        for k in ("pos", "x", "t"):
            assert np.array_equal(getattr(o, k).numpy(), G[f"aug{seed}_{k}"]), (seed, k)
        assert np.allclose(o.bbox.numpy(), G[f"aug{seed}_bbox"], rtol=0, atol=1e-4), seed
        seen.add(len(o.pos))
    assert len(seen) > 3            # the seeds exercise different branches (crop / no crop, flips, zooms)


def test_subsample_matches_the_reference_loop():
    from dagr_amd.data.augment import subsample_events
    out, keep = subsample_events(G["sub_pos_in"], G["sub_pol"].reshape(-1), 0.6)
    assert np.array_equal(keep, G["sub_mask"]) and keep.sum() > 50
    assert np.array_equal(out[keep], G["sub_pos_out"][keep])


def test_ncaltech101_samples(tmp_path):
    from dagr_amd.data.ncaltech101_data import NCaltech101
    i = 0
    for cls in ("airplanes", "zebra"):
        (tmp_path / "training" / cls).mkdir(parents=True)
        (tmp_path / "annotations" / cls).mkdir(parents=True)
        for k in (1, 2):
            np.savez(tmp_path / "training" / cls / f"image_{k:04d}.npz", **{c: G[f"nc{i}_raw_{c}"] for c in "xytp"})
            np.array([0, 0, 10 + k, 20, 110, 20, 110, 90 + k, 10 + k, 90 + k, 0, 0], dtype=np.int16).tofile(
                tmp_path / "annotations" / cls / f"annotation_{k:04d}.bin")
            i += 1
    ds = NCaltech101(tmp_path, "training", transform=None, num_events=250, reader=lambda p: np.load(p), suffix=".npz")
    assert list(ds.classes) == list(G["nc_classes"]) and len(ds) == 4
    for i in range(4):
        d = ds[i]
        assert np.array_equal(d.pos.numpy(), G[f"nc{i}_pos"]) and np.array_equal(d.t.numpy(), G[f"nc{i}_t"])
        assert np.array_equal(d.x.numpy(), G[f"nc{i}_x"]) and np.array_equal(d.bbox.numpy(), G[f"nc{i}_bbox"])


def test_downsample_oracle_matches_the_reference_script():
    """oracle/downsample.py (what the device downsampler is checked against on the GPU) == ``downsample_events`` of the
    reference's scripts/downsample_events.py over two chunks with the carried change map."""
    from oracle import downsample as od
    cm = None
    for c in range(2):
        ev = {k: G[f"ds{c}_in_{k}"] for k in ("x", "y", "t", "p")}
        res, cm = od.downsample_events(ev, 48, 64, 24, 32, change_map=cm)
        for k in ("x", "y", "t", "p"):
            assert res[k].dtype == G[f"ds{c}_out_{k}"].dtype and np.array_equal(res[k], G[f"ds{c}_out_{k}"]), (c, k)
        assert np.array_equal(cm, G[f"ds{c}_change_map"])
    assert 0 < len(G["ds1_out_t"]) < len(G["ds1_in_t"])


def test_event_window_reader_on_an_h5_shaped_mapping():
    """``load_event_window`` (dsec_utils.py:82-126: ms_to_idx look-up, forward / backward windows by count or duration,
    row crop, time shift to ``time_window``) on a dict with the datasets of an ``events_2x.h5``."""
    from dagr_amd.data.dsec_utils import load_event_window
    h5 = {k: G["h5_" + k.replace("/", "_")] for k in ("events/x", "events/y", "events/t", "events/p", "t_offset", "ms_to_idx")}
    for k, kw in enumerate([dict(num_events=3000, height=215, time_window=1000000), dict(num_us=-50000, time_window=1000000),
                            dict(num_us=30000, height=100, time_window=1000000), dict(num_events=-1500, time_window=1000000)]):
        (xy, t, p), tq = load_event_window(h5, 7_000_000 + 200_000, **kw)
        for got, name in ((xy, "xy"), (t, "t"), (p, "p")):
            want = G[f"h5w{k}_{name}"]
            assert got.dtype == want.dtype and np.array_equal(got, want), (k, name)
        assert int(tq) == int(G[f"h5w{k}_tq"])
        assert len(t) > 100


def _stream_case(tmp_path):
    from tests.make_golden_refpy_data import stream_recording
    ev = stream_recording()
    src = tmp_path / "events.npz"
    np.savez(src, **ev)
    return src


def _check_stream_output(path):
    import hashlib
    z = np.load(path)
    assert len(z["t"]) == int(G["stream_count"]) and int(z["t_offset"]) == int(G["stream_t_offset"])
    for k in ("x", "y", "p", "t", "ms_to_idx"):
        assert np.array_equal(z[k][:64], G[f"stream_head_{k}"]), k
        digest = np.frombuffer(hashlib.sha256(np.ascontiguousarray(z[k]).tobytes()).digest(), dtype=np.uint8)
        assert np.array_equal(digest, G[f"stream_sha_{k}"]), k


def test_downsample_script_stream_matches_the_reference_main_loop(tmp_path):
    """scripts/downsample_events.py's chunk loop, casts, t_offset and ms_to_idx over a 230 123-event recording == the
    reference script's main loop (digests of every output column), with the CPU oracle standing in for the device kernel
    (the GPU twin of this test runs the kernel: tests/test_downsample_gpu.py)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import downsample_events as D
    from oracle import downsample as od
    state = {"map": None}

    def cpu(ev, ih, iw, oh, ow):
        out, state["map"] = od.downsample_events(ev, ih, iw, oh, ow, change_map=state["map"])
        return out
    dst = tmp_path / "events_2x.npz"
    counts = D.main(["--input_path", str(_stream_case(tmp_path)), "--output_path", str(dst), "--input_height", "48",
                     "--input_width", "64", "--output_height", "24", "--output_width", "32"], downsampler=cpu)
    assert counts["t"] == int(G["stream_count"])
    _check_stream_output(dst)


def _rec_inputs():
    dets = []
    for i in range(7):
        dets.append({k: torch.from_numpy(G[f"rec_in{i}_{k}"]) for k in ("boxes", "labels", "scores")})
    return dets, [str(s) for s in G["rec_seqs"]], [int(t) for t in G["rec_stamps"]]


def test_record_writers_match_the_reference_functions(tmp_path):
    """``bbox_t_to_ndarray`` / ``compile`` / ``DictBuffer`` (utils/buffers.py:46-80,124-146) and the per-sequence,
    time-sorted ``detections_<sequence>.npy`` files of run_test_interframe.py:21-45 -- the repository's record helpers
    and the scripts' ``gather_and_save`` against the reference's own functions."""
    import sys
    from dagr_amd.utils import buffers as B
    dets, seqs, stamps = _rec_inputs()
    got = B.bbox_t_to_ndarray(dets[1], stamps[1])
    assert got.dtype == G["rec_single"].dtype and np.array_equal(got, G["rec_single"])
    gt = B.bbox_t_to_ndarray({k: v for k, v in dets[1].items() if k != "scores"}, stamps[1])
    assert gt.dtype == G["rec_single_gt"].dtype and np.array_equal(gt, G["rec_single_gt"])
    comp = B.compile(dets, seqs, stamps)
    assert sorted(comp) == sorted(set(seqs))
    for k, v in comp.items():
        assert np.array_equal(v, G[f"rec_compiled_{k}"]), k
    db = B.DictBuffer()
    for i in range(4):
        db.update({"a": float(i), "b": float(i * i)})
    assert np.allclose([db.compute()["a"], db.compute()["b"]], G["dictbuffer"], rtol=1e-15)
    # the scripts' writer: rows (index of the sequence name, t, x1, y1, x2, y2, score, label) -> detections_{sequence}.npy,
    # one file per sequence STRING as the reference's save_detections writes them -- names that share their digits
    # (interlaken_00_a / interlaken_00_b on the DSEC test split) stay apart
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import _common as C
    names = sorted(set(seqs) | {"thun_01_b", "interlaken_01_a"})
    flat = [dict(boxes=d["boxes"].numpy(), labels=d["labels"].numpy(), scores=d["scores"].numpy(), sequence=s, t=t)
            for d, s, t in zip(dets, seqs, stamps)]
    twin = dict(flat[-1], sequence="thun_01_b")                       # same digits as thun_01_a, another recording
    files = C.gather_and_save(C.detection_rows(flat + [twin], torch.device("cpu"), names), tmp_path, 0, names)
    assert sorted(files) == sorted(f"detections_{s}.npy" for s in set(seqs) | {"thun_01_b"})
    assert files["detections_thun_01_b.npy"] == len(twin["boxes"])
    for fname, ref in (("detections_zurich_city_12_a.npy", "rec_saved_detections_zurich_city_12_a"),
                       ("detections_thun_01_a.npy", "rec_saved_detections_thun_01_a")):
        mine, want = np.load(tmp_path / fname), G[ref]
        assert mine.dtype == want.dtype and len(mine) == len(want)
        assert np.array_equal(mine["t"], want["t"]) and bool((np.diff(mine["t"].astype(np.int64)) >= 0).all())
        # within one timestamp the reference's argsort is not stable: compare as sets of records per timestamp
        for t in np.unique(want["t"]):
            a, b = mine[mine["t"] == t], want[want["t"] == t]
            key = lambda r: np.lexsort((r["class_confidence"], r["x"]))
            for f in ("x", "y", "w", "h", "class_id", "class_confidence"):
                assert np.allclose(a[key(a)][f], b[key(b)][f], rtol=1e-6, atol=1e-4), (fname, f)


def test_metric_inputs_match_what_the_reference_hands_to_coco():
    """The reference's own half of ``evaluate_detection`` (utils/coco_eval.py:15-60,96-144,175-233): which images are
    evaluated (those with ground truth), their boxes as (x, y, w, h) through float32, category = class + 1, the scores --
    ``coco_eval.evaluated_images`` against the annotation / result lists the reference builds for pycocotools."""
    from dagr_amd.utils.coco_eval import evaluated_images
    dets, _, _ = _rec_inputs()
    gts = [dict(boxes=torch.from_numpy(G[f"coco_gt{i}_boxes"]), labels=torch.from_numpy(G[f"coco_gt{i}_labels"]))
           for i in range(7)]
    images = evaluated_images(gts, dets)
    assert len(images) == int(G["coco_n_images"]) == 4                      # three of the seven images carry no box
    ann, res = G["coco_ann"], G["coco_res"]
    for k, (g_box, g_cls, d_box, d_cls, d_score) in enumerate(images):
        a = ann[ann[:, 0] == k + 1]
        r = res[res[:, 0] == k + 1]
        assert np.array_equal(a[:, 2:6], g_box) and np.array_equal(a[:, 1], g_cls + 1)
        assert np.allclose(a[:, 6], g_box[:, 2] * g_box[:, 3], rtol=1e-6)   # area: float32 product there, float64 here
        assert np.array_equal(r[:, 2:6], d_box) and np.array_equal(r[:, 1], d_cls + 1)
        assert np.array_equal(r[:, 6], d_score)


def test_checkpointer_resolves_directories_like_the_reference(tmp_path):
    """``Checkpointer.search_for_checkpoint`` / ``checkpoint`` / ``restore_checkpoint`` (utils/logging.py:25-88): last model
    preferred unless the best is asked for, best = highest mAP in the file name, the saved dictionary's keys."""
    import types as _types
    from dagr_amd.utils.logging import Checkpointer
    lin = torch.nn.Linear(2, 2)
    opt = torch.optim.SGD(lin.parameters(), lr=0.1)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda i: 1.0)
    ema = _types.SimpleNamespace(ema=torch.nn.Linear(2, 2), updates=7)
    ck = Checkpointer(output_directory=tmp_path, model=lin, optimizer=opt, scheduler=sched, ema=ema, args={"x": 1})
    picks = [str(ck.search_for_checkpoint(tmp_path))]
    for name in ("best_model_mAP_0.125", "best_model_mAP_0.5", "best_model_mAP_0.25"):
        ck.checkpoint(3, name=name)
    picks.append(ck.search_for_checkpoint(tmp_path, best=False).name)
    ck.checkpoint(9, name="last_model")
    picks.append(ck.search_for_checkpoint(tmp_path, best=False).name)
    picks.append(ck.search_for_checkpoint(tmp_path, best=True).name)
    assert picks == [str(v) for v in G["ckpt_picks"]]
    assert sorted(torch.load(tmp_path / "last_model.pth", weights_only=False)) == [str(v) for v in G["ckpt_keys"]]
    assert ck.restore_checkpoint(tmp_path, best=False) == int(G["ckpt_restored_epoch"])


def test_random_zoom_subsample_branch_matches_the_reference():
    """RandomZoom(subsample=True) with zoom < 1 (augment.py:146-198; dead with the shipped configs): the zoomed
    coordinates are int16 before the integrate-and-fire pass, as in the reference (tests/make_golden_refpy_zoom.py)."""
    from dagr_amd.data.augment import RandomZoom
    from dagr_amd.data.utils import to_data
    Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_py_zoom_subsample.npz"))
    base = {k: Z[f"base_{k}"] for k in ("x", "y", "t", "p", "bbox")}
    sizes = set()
    for seed in range(4):
        zoom = RandomZoom(zoom=[0.5, 0.9], subsample=True)
        zoom.init(180, 240)
        d = to_data(**{k: v.copy() for k, v in base.items()}, width=240, height=180, time_window=1000000)
        torch.manual_seed(seed)
        o = zoom(d)
        for k in ("pos", "x", "t"):
            assert np.array_equal(getattr(o, k).numpy(), Z[f"zoom{seed}_{k}"]), (seed, k)
        assert np.allclose(o.bbox.numpy(), Z[f"zoom{seed}_bbox"], rtol=0, atol=1e-4), seed
        sizes.add(len(o.pos))
    assert len(sizes) == 4 and min(sizes) > 100


def test_random_crop_with_a_frame_matches_the_reference():
    """RandomCrop on a sample that carries a frame: events, boxes AND the frame as the reference's ``_crop_image``
    (augment.py:51-58) leaves it -- it indexes the first two dimensions of the [1, 3, H, W] tensor, so every window that
    does not start in row 0 blanks the whole frame (mirrored, see dagr_amd/data/augment.py:_keep_window)."""
    from dagr_amd.data import augment as A
    from dagr_amd.data.utils import to_data
    Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_py_zoom_subsample.npz"))
    base = {k: Z[f"base_{k}"] for k in ("x", "y", "t", "p", "bbox")}
    for seed in range(6):
        crop = A.RandomCrop([0.75, 0.75], p=1.0)
        crop.init(180, 240)
        d = to_data(**{k: v.copy() for k, v in base.items()}, width=240, height=180, time_window=1000000)
        d.image = torch.from_numpy(Z["crop_frame"].copy())
        torch.manual_seed(seed)
        o = crop(d)
        for k in ("pos", "x", "t"):
            assert np.array_equal(getattr(o, k).numpy(), Z[f"crop{seed}_{k}"]), (seed, k)
        assert np.allclose(o.bbox.numpy(), Z[f"crop{seed}_bbox"], rtol=0, atol=1e-4), seed
        assert np.array_equal(o.image.numpy().astype(np.int64).sum(axis=(0, 2, 3)), Z[f"crop{seed}_channel_sums"]), seed
        assert np.array_equal(o.image.numpy()[..., ::12, ::12], Z[f"crop{seed}_grid"]), seed
    # the spatial form stays available
    A.REFERENCE_FRAME_CROP = False
    try:
        d = to_data(**{k: v.copy() for k, v in base.items()}, width=240, height=180, time_window=1000000)
        d.image = torch.from_numpy(Z["crop_frame"].copy())
        torch.manual_seed(0)
        o = A.RandomCrop([0.75, 0.75], p=1.0)(d)
        assert int(o.image.sum()) > 0
    finally:
        A.REFERENCE_FRAME_CROP = True
