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
