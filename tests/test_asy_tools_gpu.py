"""The masked row operators of the asynchronous update (1:1 replacements of the reference's `asy_tools`,
src/dagr/asynchronous/asy_tools/main.cu) and DAGR.forward(reset=False):
  * libdagr_hip's kernels vs the reference's OWN kernels (oracle/_ref/libasy_tools_ref.so, compiled from
    /root/reference by oracle/Makefile) -- bit for bit -- and vs the numpy restatement (oracle/asy.py);
  * events fed in micro-batches with reset=False give exactly the outputs of one reset=True call on all of them (the
    equality the reference's asynchronous model guarantees, evaluate_flops.py:139-147)."""
import numpy as np
import pytest
import torch

from oracle import asy as oasy
from oracle import model as om
from oracle import ref_harness
from dagr_amd.data import Batch, Data
from dagr_amd.utils import synthetic as syn
from dagr_amd.utils.buffers import format_data
from dagr_amd.utils.testing_weights import randomize_

pytestmark = pytest.mark.gpu


def _case(n_rows, K, Cin, Cout, seed):
    rng = np.random.default_rng(seed)
    idx = np.sort(rng.choice(n_rows, size=K, replace=False)).astype(np.int64)
    return dict(idx=idx, x_in=rng.standard_normal((n_rows, Cin)).astype(np.float32),
                x_out=rng.standard_normal((n_rows, Cout)).astype(np.float32),
                w=(rng.standard_normal((Cout, Cin)) * 0.2).astype(np.float32),
                b=rng.standard_normal(Cout).astype(np.float32))


@pytest.mark.parametrize("n_rows,K,Cin,Cout", [(500, 37, 18, 16), (2000, 600, 64, 64), (90, 90, 130, 7), (10, 1, 3, 100)])
@pytest.mark.parametrize("add", [False, True])
def test_masked_lin_matches_reference_kernels_and_oracle(n_rows, K, Cin, Cout, add):
    from dagr_amd.asynchronous import asy_tools
    c = _case(n_rows, K, Cin, Cout, seed=n_rows + Cin)
    T = lambda a: torch.from_numpy(a).cuda()
    P = lambda t: t.data_ptr()
    for with_bias in (True, False):
        out = T(c["x_out"])
        if with_bias:
            asy_tools.masked_lin(T(c["idx"]), T(c["x_in"]), out, T(c["w"]), T(c["b"]), add)
        else:
            asy_tools.masked_lin_no_bias(T(c["idx"]), T(c["x_in"]), out, T(c["w"]), add)
        want = oasy.masked_lin(c["idx"], c["x_in"], c["x_out"], c["w"], c["b"] if with_bias else None, add)
        got = out.cpu().numpy()
        untouched = np.setdiff1d(np.arange(n_rows), c["idx"])
        assert np.array_equal(got[untouched], c["x_out"][untouched])                 # only the masked rows move
        assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
        if ref_harness.asy_available():
            R = ref_harness.asy_lib()
            ro, ri, rx, rw, rb = T(c["x_out"]), T(c["idx"]), T(c["x_in"]), T(c["w"]), T(c["b"])
            if with_bias:
                assert R.ref_masked_lin(P(ri), K, P(rx), P(ro), n_rows, P(rw), P(rb), Cin, Cout, int(add)) == 0
            else:
                assert R.ref_masked_lin_no_bias(P(ri), K, P(rx), P(ro), n_rows, P(rw), Cin, Cout, int(add)) == 0
            assert torch.equal(ro, out), "differs from the reference's own kernel"


def test_masked_isdiff_and_bn_match_reference_kernels_and_oracle():
    from dagr_amd.asynchronous import asy_tools
    rng = np.random.default_rng(5)
    n_rows, K, C = 3000, 700, 66
    idx = np.sort(rng.choice(n_rows, size=K, replace=False)).astype(np.int64)
    x_old = rng.standard_normal((n_rows, C)).astype(np.float32)
    x_new = x_old.copy()
    changed = rng.choice(idx, size=K // 3, replace=False)
    x_new[changed, rng.integers(0, C, len(changed))] += rng.choice([1e-2, 1e-4, -1e-3, 5e-7], len(changed)).astype(np.float32)
    T = lambda a: torch.from_numpy(a).cuda()
    P = lambda t: t.data_ptr()
    for atol, rtol in ((1e-3, 1e-3), (1e-8, 0.0), (0.0, 1e-5)):
        ti = T(idx)
        kept = asy_tools.masked_isdiff(ti, T(x_old), T(x_new), atol, rtol)
        marked, want = oasy.masked_isdiff(idx, x_old, x_new, atol, rtol)
        assert np.array_equal(ti.cpu().numpy(), marked) and np.array_equal(kept.cpu().numpy(), want)
        if ref_harness.asy_available():
            ri = T(idx)
            assert ref_harness.asy_lib().ref_masked_isdiff(P(ri), K, P(T(x_old)), P(T(x_new)), n_rows, C, atol, rtol) == 0
            assert torch.equal(ri, ti)
    mean, var = rng.standard_normal(C).astype(np.float32), rng.uniform(0.5, 1.5, C).astype(np.float32)
    w, b = rng.uniform(0.5, 1.5, C).astype(np.float32), rng.standard_normal(C).astype(np.float32)
    out = T(x_new)
    asy_tools.masked_inplace_BN(T(idx), T(x_old), out, T(mean), T(var), T(w), T(b), 1e-5)
    want = oasy.masked_inplace_BN(idx, x_old, x_new, mean, var, w, b, 1e-5)
    assert np.abs(out.cpu().numpy() - want).max() <= 1e-5
    if ref_harness.asy_available():
        ro = T(x_new)
        assert ref_harness.asy_lib().ref_masked_inplace_BN(P(T(idx)), K, P(T(x_old)), P(ro), n_rows, C, P(T(mean)), P(T(var)),
                                                           P(T(w)), P(T(b)), 1e-5) == 0
        assert torch.equal(ro, out), "differs from the reference's own kernel"
    with pytest.raises(RuntimeError):
        asy_tools.masked_lin(torch.from_numpy(idx), T(x_old), T(x_new), T(w[:, None]), T(b), False)   # host tensor


@pytest.mark.parametrize("over", [{}, dict(use_image=True, img_net="resnet18")])
def test_forward_reset_false_equals_one_call_on_all_events(over):
    from dagr_amd.model.networks.dagr import DAGR
    W, H, B = 320, 215, 2
    torch.manual_seed(0)
    args = om.default_args(batch_size=B, **over)
    model = randomize_(DAGR(args, height=H, width=W)).eval().cuda()
    model.cache_luts(width=W, height=H, radius=args.radius)
    raw = [syn.edges_window(4000, W, H, seed=70 + s) for s in range(B)]
    images = [torch.randint(0, 256, (1, 3, H, W), generator=torch.Generator().manual_seed(s), dtype=torch.uint8)
              for s in range(B)]

    def batch_of(parts):
        samples = []
        for s, (lo, hi) in enumerate(parts):
            x, y, t, p = (a[lo:hi] for a in raw[s])
            d = Data(x=torch.from_numpy(p.reshape(-1, 1)), pos=torch.from_numpy(np.stack([x, y], -1)),
                     t=torch.from_numpy(t), width=W, height=H, time_window=1000000)
            if over:
                d.image = images[s]
            samples.append(d)
        return format_data(Batch.from_data_list(samples).cuda())

    with torch.no_grad():
        full, = model(batch_of([(0, 4000)] * B), reset=True, return_targets=False)
        full = [{k: v.clone() for k, v in d.items()} for d in full]
        cuts = [0, 2500, 3300, 3999, 4000]            # one large part, then micro-batches down to a single event
        for k in range(len(cuts) - 1):
            det, = model(batch_of([(cuts[k], cuts[k + 1])] * B), reset=(k == 0), return_targets=False)
        again, = model(batch_of([(0, 4000)] * B), reset=True, return_targets=False)      # a reset starts over
    for a, b, c in zip(full, det, again):
        for key in ("boxes", "scores", "labels"):
            if over:    # the image branch is PyTorch-ROCm convolutions (split-K kernels accumulate with atomics): equal to
                # fp32 rounding from call to call, not bit for bit
                assert a[key].shape == b[key].shape == c[key].shape, key
                assert torch.allclose(a[key].float(), b[key].float(), rtol=1e-4, atol=1e-3), key
                assert torch.allclose(a[key].float(), c[key].float(), rtol=1e-4, atol=1e-3), key
            else:
                assert torch.equal(a[key], b[key]), key
                assert torch.equal(a[key], c[key]), key
    assert sum(len(d["boxes"]) for d in full) > 0
