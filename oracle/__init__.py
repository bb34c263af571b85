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
* the reference's own plain-torch functions on the path -- ``_sample_features``, ``to_dense``,
  ``consecutive_cluster``, ``round_to_pixel``, ``compute_pooling_at_each_layer``,
  ``voxel_size_to_params``, ``init_grid_and_stride`` + ``decode_outputs``,
  ``postprocess_network_output``, ``format_data``, ``denormalize_pos``, and (with the third-party
  calls inside them served by the restatements below) ``MySplineConv.init_lut`` / ``message_lut``,
  ``Pooling.forward`` and the ``AsyncGraph`` / ``SlidingWindowGraph`` host state machine: pinned.
  ``tests/make_golden_refpy.py`` imports ``/root/reference/src`` with the absent packages stubbed,
  runs that code on CPU and commits ``tests/golden/ref_py_functions.npz``;
  ``tests/test_oracle_refpy.py`` checks the restatements (and the host mirror's twins) against it.
* the whole path end to end -- the reference's own ``Net`` / ``Layer`` / ``ConvBlock`` / ``MySplineConv`` (LUT path)
  / ``Pooling`` / ``EV_TGN`` + ``SlidingWindowGraph`` / ``GNNHead`` / ``CNNHead`` / ``HookModule`` /
  ``DAGR.cache_luts`` code, executed unmodified on CPU over functional stand-ins of the absent packages that are
  built on this oracle's primitives (``tests/refpy_fakes.py``, ``tests/make_golden_refpy_model.py``): pinned.
  ``oracle.model.forward_events`` reproduces its decoded outputs to 1e-6 for dagr-s, dagr-l, a 240x180 sensor and
  the ResNet-18 image-fusion configuration (``tests/golden/ref_py_model.npz``,
  ``tests/test_oracle_refpy.py::test_whole_model_wiring_matches_the_reference_code``), i.e. the wiring of the
  oracle is the reference's, independently of the reading that produced ``oracle/model.py``.
* third-party SplineConv / scatter / cluster arithmetic (fp32): it lives in un-vendored,
  un-pinned third-party packages (torch_spline_conv, torch_scatter,
  torch_cluster, torch_sparse, torch_geometric -- ``install_env.sh:3-11``) that
  are absent here, and the reference has no tests or golden vectors for this
  path: **parity unpinned** for those primitives (``spline_basis``, ``spline_weighting``,
  ``grid_cluster``, ``scatter_max/mean``, ``T.Cartesian``, ``ToSparseTensor`` ordering).  They restate the published
  algorithms and are anchored on the reference's in-repo
  call sites and restatements cited in each docstring.
"""
