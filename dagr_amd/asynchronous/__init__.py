"""Asynchronous (per-event) operation.

``asy_tools``: the reference's native module of masked row operators (``src/dagr/asynchronous/asy_tools/main.cu``),
replaced 1:1 by libdagr_hip -- the building blocks of the reference's incremental layer wrappers
(``asynchronous/linear.py``, ``batch_norm.py``, ``conv.py``, ``max_pool.py``).

``StreamingWindow``: what ``DAGR.forward(x, reset=False)`` runs on.  The reference's asynchronous model keeps per-layer
caches and propagates only what a new event changes, and guarantees that this equals the synchronous forward on all
events so far (``evaluate_flops.py:139-147``).  On this stack a whole window costs well under a millisecond through the
engine, so the running window is kept on the device and re-evaluated as a whole when events arrive: the same outputs
(bit-identical to one ``reset=True`` call on the concatenated events) without the per-layer caches."""
from . import asy_tools  # noqa: F401
from .streaming import StreamingWindow  # noqa: F401


def make_model_asynchronous(module, log_flops=False):
    """Entry point of the reference's conversion (``asynchronous/__init__.py:41-110``; used as
    ``model = make_model_asynchronous(model, log_flops=True)`` followed by ``model.forward(events_initial, reset=True)``
    and ``model.forward(events_new, reset=False)``, evaluate_flops.py:113-118).  ``DAGR.forward(reset=False)`` is native
    here, so there is nothing to convert: the model is returned as it is.  The reference's per-layer FLOP log counts the
    operations of ITS incremental update scheme, which this stack does not run (the window is re-evaluated): asking for
    it is an error rather than a made-up number."""
    if log_flops:
        raise NotImplementedError("log_flops counts the reference's per-layer incremental updates (asynchronous/flops); "
                                  "this stack re-evaluates the running window -- see DESIGN.md section 7")
    if not hasattr(module, "forward"):
        raise TypeError("module must be a torch.nn.Module")
    return module


def make_model_synchronous(module):
    """``asynchronous/__init__.py:30-39``: back to the synchronous forward -- the same object here."""
    return module
