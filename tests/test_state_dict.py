"""CPU suite: the module tree exposes the reference's state_dict layout, so
``ModelEMA(model).ema.load_state_dict(checkpoint['ema'])`` (scripts/run_test.py:57-58) would work."""
import torch

from oracle.model import default_args
from dagr_amd.model.networks.dagr import DAGR


def _shapes(**over):
    m = DAGR(default_args(**over), height=215, width=320)
    return {k: tuple(v.shape) for k, v in m.state_dict().items()}


def test_events_only_dagr_s_layout():
    sd = _shapes(batch_size=2)
    # backbone Layer 1 (net.py:75): Layer(2+1 -> 16)
    assert sd["backbone.conv_block1.conv_block1.conv.weight"] == (25, 3, 16)
    assert sd["backbone.conv_block1.conv_block1.conv.lin.weight"] == (16, 3)
    assert sd["backbone.conv_block1.conv_block1.conv.kernel_size"] == (2,)
    assert sd["backbone.conv_block1.conv_block1.conv.is_open_spline"] == (2,)
    assert sd["backbone.conv_block1.conv_block2.lin.mlp.weight"] == (16, 3)
    for k in ("weight", "bias", "running_mean", "running_var"):
        assert sd[f"backbone.conv_block1.conv_block2.norm_skip.module.{k}"] == (16,)
    assert sd["backbone.conv_block1.conv_block2.norm.module.num_batches_tracked"] == ()
    # layers 2..5 (net.py:81-99), dagr-s channels [1,16,64,64,64,64]
    assert sd["backbone.layer2.conv_block1.conv.weight"] == (25, 18, 64)
    assert sd["backbone.layer2.conv_block2.lin.mlp.weight"] == (64, 18)
    for l in (3, 4, 5):
        assert sd[f"backbone.layer{l}.conv_block1.conv.weight"] == (25, 66, 64)
        assert sd[f"backbone.layer{l}.conv_block2.conv.weight"] == (25, 64, 64)
    # GNN head (dagr.py:150-163): predictors carry a bias, conv blocks do not
    for s in ("1", "2"):
        assert sd[f"head.stem{s}.conv.weight"] == (25, 64, 64)
        assert sd[f"head.cls_pred{s}.weight"] == (25, 64, 2) and sd[f"head.cls_pred{s}.bias"] == (2,)
        assert sd[f"head.reg_pred{s}.weight"] == (25, 64, 4) and sd[f"head.obj_pred{s}.bias"] == (1,)
        assert f"head.stem{s}.conv.bias" not in sd
    # unused dense YOLOXHead parameters are present (dagr.py:137; yolo_stem_width 0.5 -> hidden 128)
    assert sd["head.stems.0.conv.weight"] == (128, 32, 1, 1)
    assert sd["head.cls_preds.1.weight"] == (2, 128, 1, 1)
    # pooling buffers are non-persistent (pooling.py:24-35)
    assert not any("pool" in k for k in sd)


def test_num_scales_1_drops_second_head_and_100_classes():
    sd = _shapes(batch_size=1, num_scales=1, dataset="ncaltech101", net_stem_width=1.0, yolo_stem_width=1.0)
    assert "head.stem2.conv.weight" not in sd
    assert sd["head.cls_pred1.weight"] == (25, 128, 100)
    assert sd["backbone.layer4.conv_block1.conv.weight"] == (25, 130, 128)


def test_state_dict_roundtrip_strict():
    a = DAGR(default_args(batch_size=1), height=215, width=320)
    b = DAGR(default_args(batch_size=1), height=215, width=320)
    b.load_state_dict(a.state_dict(), strict=True)
    for (k1, v1), (k2, v2) in zip(a.state_dict().items(), b.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


def test_asynchronous_entry_points_resolve():
    """``dagr.asynchronous.make_model_asynchronous / make_model_synchronous`` (asynchronous/__init__.py:30-110) exist under
    the reference's import path and return the module they were given; they switch ``reset=False`` calls between the
    incremental update and the re-evaluation of the running window; the reference's FLOP log is refused."""
    import pytest
    import torch
    from dagr.asynchronous import make_model_asynchronous, make_model_synchronous
    m = torch.nn.Linear(1, 1)
    assert make_model_asynchronous(m) is m and m.asynchronous is True
    assert make_model_synchronous(m) is m and m.asynchronous is False
    with pytest.raises(NotImplementedError):
        make_model_asynchronous(m, log_flops=True)
