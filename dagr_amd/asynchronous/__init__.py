"""Asynchronous (per-event) operation.

``asy_tools``: the reference's native module of masked row operators (``src/dagr/asynchronous/asy_tools/main.cu``),
replaced 1:1 by libdagr_hip -- the building blocks of the reference's incremental layer wrappers
(``asynchronous/linear.py``, ``batch_norm.py``, ``conv.py``, ``max_pool.py``).

The update itself.  The reference converts a model layer by layer (``make_model_asynchronous``): every conv / pooling /
norm module keeps its input and output graph and, when ``forward(events_new, reset=False)`` brings new events,
recomputes what they touch (``conv.py:94-227``, ``max_pool.py:123-243``).  Here ``DAGR.forward(x, reset=False)`` is
native (``WindowEngine.forward_append``, csrc/async_update.hip):
  * edges point from older to newer events, so the rows of the events already in the window never change -- the update
    APPENDS level-0 rows: the new events are linked into per-pixel chains beside the window's pixel index, their in-edges
    are searched with the reference's FIFO semantics, and ``conv_block1`` runs on the new rows only
    (the reference's ``graph_new_nodes`` branch, conv.py:107-125,196-207);
  * pool1 keeps its per-voxel accumulators resident (maximum / exact position sums / counts / source-cell bitmaps: the
    reference's cluster caches, max_pool.py:41-62,126-158) and the new rows are added to them -- a maximum over more
    members and an exact sum do not need the old members again;
  * from level 1 on (at most 2240 nodes per sample, the receptive field of a handful of events covers it after two
    layers) the network is evaluated as for a window.
The result equals the synchronous forward on all events so far -- here bit for bit, the reference checks 1e-3
(``evaluate_flops.py:139-147``)."""
from . import asy_tools  # noqa: F401


def make_model_asynchronous(module, log_flops=False):
    """``asynchronous/__init__.py:41-110`` (used as ``model = make_model_asynchronous(model)``, then
    ``model.forward(events_initial, reset=True)`` and ``model.forward(events_new, reset=False)``,
    evaluate_flops.py:113-118): ``reset=False`` calls update the resident window incrementally.  The reference's
    per-layer FLOP log counts the operations of ITS update scheme (per-layer residual scatters); this stack runs another
    one, so asking for that log is an error rather than a made-up number."""
    if log_flops:
        raise NotImplementedError("log_flops counts the reference's per-layer incremental updates (asynchronous/flops)")
    if not hasattr(module, "forward"):
        raise TypeError("module must be a torch.nn.Module")
    module.asynchronous = True
    return module


def make_model_synchronous(module):
    """``asynchronous/__init__.py:30-39``: ``reset=False`` calls evaluate the whole running window again -- the
    synchronous forward on all events so far, the side of the consistency check (evaluate_flops.py:139-147) the
    incremental update is compared with."""
    module.asynchronous = False
    return module
