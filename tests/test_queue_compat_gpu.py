"""GPU parity of the 1:1 `ev_graph_cuda` replacements (dagr_fill_edges / dagr_insert_in_queue*):
  * the mirrored SlidingWindowGraph against the CPU oracle over multi-call sequences, including
    reset=False incremental calls with delete_nodes (min_index > 0) and a single-event call;
  * directly against the reference's OWN kernels (oracle/_ref, compiled from /root/reference) when that
    library travelled to the box: identical FIFO volume and edge buffer, bit for bit."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import graph as og
from oracle import ref_harness
from dagr_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu


def _seq(seed, n_calls, n_per_call, W, H, B):
    rng = np.random.default_rng(seed)
    t0 = 0
    out = []
    for c in range(n_calls):
        n = n_per_call if c != 2 else 1      # third call: the single-event kernel (QUIRK-5)
        x = rng.integers(0, W, n); y = rng.integers(0, H, n)
        t = np.sort(rng.integers(t0, t0 + 20000, n)); t0 += 20000
        b = np.sort(rng.integers(0, B, n)).astype(np.int32)
        out.append((b, np.stack([x, y, t], -1).astype(np.int32)))
    return out


@pytest.mark.parametrize("delete_nodes", [False, True])
def test_sliding_window_graph_incremental_matches_oracle(delete_nodes):
    from dagr_amd.graph.ev_graph import SlidingWindowGraph
    W, H, B = 48, 40, 2
    dev = torch.device("cuda:0")
    g = SlidingWindowGraph(width=W, height=H, batch_size=B, max_num_neighbors=16, max_queue_size=8, radius=3,
                           delta_t_us=15000)
    o = og.SlidingWindowGraph(width=W, height=H, batch_size=B, max_num_neighbors=16, max_queue_size=8, radius=3,
                              delta_t_us=15000)
    for k, (b, pos) in enumerate(_seq(5, 5, 700, W, H, B)):
        rg = g.forward(torch.from_numpy(b).to(dev), torch.from_numpy(pos).to(dev), delete_nodes=delete_nodes)
        ro = o.forward(b, pos, delete_nodes=delete_nodes)
        eg = (rg[0] if delete_nodes else rg).cpu().numpy()
        eo = ro[0] if delete_nodes else ro
        assert eg.shape == eo.shape and (eg == eo).all(), f"call {k}"
        if delete_nodes and rg[1] is not None:
            assert (rg[1].cpu().numpy() == ro[1]).all()
        assert (g.event_queue.cpu().numpy() == o.event_queue).all(), f"FIFO volume after call {k}"
        assert g.min_index == o.min_index and g.max_index == o.max_index
    g.reset(); o.reset()
    b, pos = _seq(9, 1, 300, W, H, B)[0]
    e1 = g.forward(torch.from_numpy(b).to(dev), torch.from_numpy(pos).to(dev), delete_nodes=False).cpu().numpy()
    assert (e1 == o.forward(b, pos, delete_nodes=False)).all()


@pytest.mark.skipif(not ref_harness.available(), reason="oracle/_ref (reference kernels) not on this box")
def test_kernels_match_reference_kernels_bit_exact():
    from dagr_amd import _lib
    L, P = _lib.lib(), _lib.ptr
    R = ref_harness.lib()
    dev = torch.device("cuda:0")
    W, H, B, Q, K, r, dt = 64, 48, 2, 16, 16, 4, 10000
    x, y, t, p, b = syn.batch_windows(syn.edges_window, 4000, B, W, H, seed=3)
    N = len(x)
    batch = torch.from_numpy(b.astype(np.int32)).to(dev)
    pos = torch.from_numpy(np.stack([x, y, t], -1).astype(np.int32)).to(dev)
    indices = torch.arange(N, dtype=torch.int32, device=dev)
    lin = pos[:, 0] + W * pos[:, 1] + W * H * batch
    sorted_lin, sort_index = torch.sort(lin, stable=True)
    sorted_indices = indices[sort_index].int().contiguous()
    uniq, counts = torch.unique_consecutive(sorted_lin, return_counts=True)
    cumsum = torch.cumsum(counts, 0).int().contiguous(); uniq = uniq.int().contiguous()
    q_ref = torch.full((B, Q, H, W), -1, dtype=torch.int32, device=dev); q_hip = q_ref.clone()
    st = _lib.cur_stream(dev)
    assert R.ref_insert_in_queue(P(sorted_indices), N, P(uniq), P(cumsum), len(uniq), P(q_ref), B, Q, H, W) == 0
    _lib.check(L.dagr_insert_in_queue(P(sorted_indices), P(uniq), P(cumsum), len(uniq), P(q_hip), B, Q, H, W, st))
    torch.cuda.synchronize()
    assert torch.equal(q_ref, q_hip)
    e_ref = torch.full((2, K * N), -1, dtype=torch.int64, device=dev); e_hip = e_ref.clone()
    ts = pos[:, 2].contiguous()
    assert R.ref_fill_edges(P(batch), P(pos), P(ts), N, P(q_ref), P(indices), K, float(r), float(dt), P(e_ref), K * N, 0,
                            N, B, Q, H, W) == 0
    _lib.check(L.dagr_fill_edges(P(batch), P(pos), P(ts), P(q_hip), P(indices), K, float(r), float(dt), P(e_hip), K * N,
                                 0, N, B, Q, H, W, st))
    torch.cuda.synchronize()
    assert torch.equal(e_ref, e_hip)
