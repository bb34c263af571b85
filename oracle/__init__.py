"""oracle/ -- CPU restatement of the reference's event-graph hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package, and only as the
checker.  Nothing under ``dagr_amd/`` imports it; the product path fails loudly
when the HIP library is missing instead of falling back to this code.

Parity pin status
-----------------
* graph build (integer): pinned against the reference's own ``ev_graph.cu``
  compiled from ``/root/reference`` (``oracle/Makefile`` -> ``oracle/_ref``) and
  run on an MI355X; outputs committed under ``tests/golden/graph_ref_*.npz``.
* SplineConv / pooling / head (fp32): the arithmetic lives in un-vendored,
  un-pinned third-party packages (torch_spline_conv, torch_scatter,
  torch_cluster, torch_sparse, torch_geometric -- ``install_env.sh:3-11``) that
  are absent here, and the reference has no tests or golden vectors for this
  path: **parity unpinned** for those ops.  They restate the published
  algorithms and are anchored on the reference's in-repo
  call sites and restatements cited in each docstring.
"""
