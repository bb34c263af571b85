"""``dagr`` -- the reference's import paths on top of ``dagr_amd``.

The reference's scripts import ``dagr.utils.args``, ``dagr.model.networks.dagr``, ``dagr.model.networks.ema``,
``dagr.utils.testing``, ``dagr.utils.buffers``, ``dagr.data.*`` (``scripts/run_test.py:8-17``,
``scripts/run_test_interframe.py:10-20``).  This package maps every ``dagr.<x>`` onto the SAME module object as
``dagr_amd.<x>`` (no second copy of any class), so a script written against the reference resolves to the MI355X
engine by having this repository on ``sys.path`` instead of the reference's ``src/``.
"""
import importlib
import importlib.abc
import importlib.machinery
import sys

import dagr_amd

__version__ = dagr_amd.__version__
__path__ = []          # a package: submodules are resolved by the finder below, never from disk


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, real):
        self.real = real
        self.attrs = {k: getattr(real, k, None) for k in ("__spec__", "__loader__", "__package__", "__name__")}

    def create_module(self, spec):
        return self.real          # the very module object of dagr_amd.<x>

    def exec_module(self, module):
        for k, v in self.attrs.items():    # the import machinery re-stamped the shared module: put its identity back
            if v is not None:
                setattr(module, k, v)


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith("dagr."):
            return None
        try:
            real = importlib.import_module("dagr_amd" + fullname[4:])
        except ModuleNotFoundError as e:
            if e.name and e.name.startswith("dagr_amd"):
                return None       # no such sub-module: a normal ImportError for dagr.<x>
            raise
        return importlib.machinery.ModuleSpec(fullname, _AliasLoader(real), is_package=hasattr(real, "__path__"))


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
