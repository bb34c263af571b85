"""CPU suite: the pieces of the training-path refactor that are plain torch -- lazily built graph attributes on ``Data``,
the edge-index recipes over a CSR, the loss written as masked sums, the target rows without mask indexing -- against the
straightforward forms they replace."""
import copy

import numpy as np
import torch

from dagr_amd.data import Batch, Data


def test_lazy_attribute_is_built_once_and_behaves_like_a_plain_one():
    calls = []
    d = Data(x=torch.zeros(3, 2))
    d.set_lazy("edge_index", lambda s: calls.append(1) or torch.arange(4).view(2, 2))
    assert d.is_lazy("edge_index") and "edge_index" in d and "edge_index" in d.keys() and not calls
    c = copy.copy(d)                                   # what Data.to / clone / shallow_copy do first
    assert torch.equal(d.edge_index, torch.arange(4).view(2, 2)) and len(calls) == 1
    assert torch.equal(d.edge_index, torch.arange(4).view(2, 2)) and len(calls) == 1 and not d.is_lazy("edge_index")
    assert c.is_lazy("edge_index")                     # the copy still holds the recipe, not the other object's value
    c.edge_index = torch.zeros(2, 0)                   # an assignment replaces the recipe
    assert c.edge_index.shape == (2, 0) and len(calls) == 1
    assert not hasattr(d, "no_such_attribute")
    e = Data(x=torch.zeros(1))
    e.set_lazy("a", lambda s: 1)
    f = copy.copy(e)
    f.set_lazy("b", lambda s: 2)                       # recipes are never shared with the object one was copied from
    assert not e.is_lazy("b") and f.a == 1 and f.b == 2


def test_clone_and_to_materialise_recipes_and_shallow_copy_carries_them():
    from dagr_amd.model.utils import shallow_copy
    d = Data(x=torch.ones(2, 1), pos=torch.zeros(2, 3))
    d.set_lazy("edge_index", lambda s: torch.tensor([[0, 1], [1, 1]]))
    s = shallow_copy(d)
    assert s.is_lazy("edge_index") and torch.equal(s.edge_index, torch.tensor([[0, 1], [1, 1]])) and d.is_lazy("edge_index")
    c = d.clone()
    assert torch.equal(c.edge_index, torch.tensor([[0, 1], [1, 1]]))


def test_collation_stashes_the_sensor_geometry_and_format_data_uses_it():
    from dagr_amd.utils.buffers import format_data
    parts = [Data(x=torch.ones(4, 1, dtype=torch.int8), pos=torch.randint(0, 30, (4, 2), dtype=torch.int16),
                  t=torch.arange(4, dtype=torch.int32), width=64, height=48, time_window=1000) for _ in range(2)]
    b = Batch.from_data_list(parts)
    assert b._geometry == (64, 48, 1000) and b.clone()._geometry == (64, 48, 1000)
    b.width = None                                      # the stash, not the collated tensors, is what format_data reads
    out = format_data(b)
    assert out.pos.shape == (8, 3) and float(out.pos[:, 0].max()) < 1.0


def _random_csr(rng, n, max_deg):
    deg = rng.integers(0, max_deg + 1, size=n)
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    col = np.concatenate([np.sort(rng.choice(n, size=k, replace=False)) for k in deg] + [np.zeros(0, int)]).astype(np.int32)
    return torch.from_numpy(rowptr), torch.from_numpy(col)


def test_edge_index_recipes_over_a_csr():
    from dagr_amd.model.layers import _ops
    rng = np.random.default_rng(3)
    rowptr, col = _random_csr(rng, 40, 6)
    E = int(rowptr[-1])
    dst = torch.repeat_interleave(torch.arange(40), (rowptr[1:] - rowptr[:-1]).long())
    # level 0: capacity-sized col, the builder's order
    d = Data(x=torch.zeros(40, 1))
    d._dagr_csr = (rowptr, torch.cat([col, torch.full((17,), -7, dtype=torch.int32)]), None, ("csr", 40))
    ei = _ops.edge_index_from_csr(d)
    assert torch.equal(ei, torch.stack([col.long(), dst]))
    assert _ops.graph_csr(d)[0] is rowptr                # the tagged CSR is authoritative: no sort, no edge_index access
    # pooled level: unique's order (by source, then destination) + the permutation that brings edge_attr back to CSR order
    p = Data(x=torch.zeros(40, 1))
    p._dagr_csr = (rowptr, col, None, ("csr", 40))
    ei = _ops.pooled_edge_index(p)
    want = torch.unique(torch.stack([col.long(), dst]), dim=-1)
    assert torch.equal(ei, want)
    perm = p._dagr_csr[2]
    assert torch.equal(ei[:, perm], torch.stack([col.long(), dst]))
    # and csr_by_destination on that edge_index reproduces the kernel's CSR
    r2, c2, _ = _ops.csr_by_destination(ei, 40)
    assert torch.equal(r2, rowptr) and torch.equal(c2, col)
    assert E == ei.shape[1]


def test_training_rows_without_mask_indexing_equal_the_indexed_form():
    from dagr_amd.model.utils import convert_to_training_format
    rng = np.random.default_rng(5)
    for trial in range(20):
        B = int(rng.integers(1, 6))
        counts = rng.integers(0, 5, size=B)
        if counts.sum() == 0:
            counts[0] = 1
        batch = torch.from_numpy(np.repeat(np.arange(B), counts))
        bbox = torch.from_numpy(rng.uniform(1, 50, size=(len(batch), 6)).astype(np.float32))
        got = convert_to_training_format(bbox, batch, B)
        want = torch.zeros((B, 100, 5))
        for i in range(B):
            rows = bbox[batch == i][:, :5].clone()
            rows[:, :2] += rows[:, 2:4] * .5
            want[i, :len(rows)] = torch.roll(rows, shifts=1, dims=1)
        assert torch.equal(got, want), trial


def test_masked_loss_sums_equal_the_indexed_form_in_value_and_gradient():
    """detection_losses sums its matched-anchor terms as masked sums over all anchors (no boolean indexing: no host
    synchronisation, capturable): same values and gradients as selecting the matched anchors first."""
    import torch.nn.functional as F
    from dagr_amd.model.networks import yolox_loss as yl
    torch.manual_seed(0)
    B, C = 3, 2
    grids, outs = [], []
    for (h, w), st in (((6, 8), 16), ((3, 4), 32)):
        o, g = yl.output_and_grid(torch.randn(B, 5 + C, h, w), st)
        grids.append(g); outs.append(o)
    base = torch.cat(outs, 1)
    labels = torch.zeros(B, 100, 5)
    labels[0, 0] = torch.tensor([1.0, 40.0, 30.0, 50.0, 40.0]); labels[0, 1] = torch.tensor([0.0, 90.0, 60.0, 30.0, 30.0])
    labels[2, 0] = torch.tensor([0.0, 64.0, 48.0, 80.0, 60.0])           # image 1 has no box at all
    a = base.clone().requires_grad_(True)
    got = yl.detection_losses(labels, a, grids, [16, 32], C)
    got[0].backward()
    # the indexed form, on the same assignment
    b = base.clone().requires_grad_(True)
    grid = torch.cat(grids, 1)[0]
    stride = torch.cat([torch.full((g.shape[1],), float(s)) for g, s in zip(grids, [16, 32])])
    centers = (grid + 0.5) * stride[:, None]
    box, obj, cls = b[..., :4], b[..., 4:5], b[..., 5:]
    fg, m_gt, m_iou = yl.simota_assign_batch(labels, box.detach(), cls.detach(), obj.detach(), centers, stride, C)
    rows = torch.gather(labels, 1, m_gt.unsqueeze(2).expand(-1, -1, 5))
    sel = fg.view(-1)
    assert 0 < int(sel.sum()) < sel.numel()
    num_fg = fg.float().sum().clamp(min=1.0)
    cls_t = F.one_hot(rows[..., 0].long().clamp(0, C - 1), C).float() * m_iou.unsqueeze(2)
    l_iou = yl.iou_loss(box.reshape(-1, 4)[sel], rows[..., 1:5].reshape(-1, 4)[sel]).sum() / num_fg
    l_obj = F.binary_cross_entropy_with_logits(obj.reshape(-1, 1), fg.float().view(-1, 1), reduction="none").sum() / num_fg
    l_cls = F.binary_cross_entropy_with_logits(cls.reshape(-1, C)[sel], cls_t.reshape(-1, C)[sel], reduction="none").sum() / num_fg
    total = yl.REG_WEIGHT * l_iou + l_obj + l_cls
    total.backward()
    for g_, w_ in ((got[0], total), (got[1], yl.REG_WEIGHT * l_iou), (got[2], l_obj), (got[3], l_cls)):
        assert abs(float(g_.detach()) - float(w_.detach())) <= 1e-5 * max(1.0, abs(float(w_.detach())))
    assert float((a.grad - b.grad).abs().max()) <= 1e-6 * max(1.0, float(b.grad.abs().max()))
    assert bool(torch.isfinite(a.grad).all())


def test_tall_linear_and_batched_weight_gradient_equal_the_plain_forms():
    """TallLinearFn (the event level's skip Linear) and _at_g (A^T g as a batch of partial products over the long dimension,
    any batch count, padded rows) against torch's own Linear / matmul."""
    from dagr_amd.model.layers.autograd import TallLinearFn, _at_g
    torch.manual_seed(1)
    x = torch.randn(9000, 3, dtype=torch.float64, requires_grad=True)
    w = torch.randn(16, 3, dtype=torch.float64, requires_grad=True)
    y = TallLinearFn.apply(x, w)
    g = torch.randn_like(y)
    y.backward(g)
    x2, w2 = x.detach().clone().requires_grad_(True), w.detach().clone().requires_grad_(True)
    y2 = torch.nn.functional.linear(x2, w2)
    y2.backward(g)
    assert torch.equal(y.detach(), y2.detach()) and torch.allclose(x.grad, x2.grad) and torch.allclose(w.grad, w2.grad, rtol=1e-12)
    for n, K, lda in ((140000, 26, 28), (140001, 26, 26), (9000, 416, 416), (100, 5, 8)):
        A = torch.randn(n, lda, dtype=torch.float64)
        gg = torch.randn(n, 16, dtype=torch.float64)
        assert torch.allclose(_at_g(A, gg, K), A[:, :K].t() @ gg, rtol=1e-10, atol=1e-9), (n, K, lda)


def test_assigning_edge_index_drops_the_kernel_side_forms_of_the_old_graph():
    """``Data.set_lazy``'s contract (assigning the attribute replaces the recipe) includes what was derived from the old
    graph: the cached CSR, the integer pixel offsets and the exact-offset cache.  Moving the SAME graph
    (``to`` / ``clone``) keeps them, and materialising the lazy ``edge_index`` is not an assignment."""
    from dagr_amd.model.layers import _ops
    rowptr, col = torch.tensor([0, 1, 3], dtype=torch.int32), torch.tensor([0, 0, 1], dtype=torch.int32)
    d = Data(x=torch.zeros(2, 1), pos=torch.zeros(2, 3))
    d._dagr_csr = (rowptr, col, None, ("csr", 2))
    d._dagr_pixel_codes = (torch.zeros(3, dtype=torch.int32), 8, 8)
    d.set_lazy("edge_index", _ops.edge_index_from_csr)
    assert d.edge_index.tolist() == [[0, 0, 1], [0, 1, 1]] and "_dagr_csr" in d.__dict__     # materialised: caches stay
    moved = d.to("cpu")
    assert "_dagr_csr" in moved.__dict__ and "_dagr_pixel_codes" in moved.__dict__
    d.edge_index = torch.tensor([[1], [0]])
    assert "_dagr_csr" not in d.__dict__ and "_dagr_pixel_codes" not in d.__dict__
    r, c, _ = _ops.graph_csr(d)                                     # rebuilt from the new edge_index
    assert r.tolist() == [0, 1, 1] and c.tolist() == [1]
