"""Drop-in import name of the reference's Python module.

Generated flowgraphs and user scripts do ``import clenabled`` and construct blocks positionally, e.g.
``clenabled.clFFT(fft_size, clenabled.CLFFT_FORWARD, window, 1, 1, 2, 0, 0, 0, 1, True)`` (the ``make:`` templates in the
reference's grc/clenabled_*.block.yml, bindings in python/bindings/*_python.cc).  Put this directory on PYTHONPATH
(``.../gr-clenabled_amd/python``) and the same lines construct the MI355X blocks of ``gr-clenabled_amd/blocks.py``.
"""
import importlib.util
import os
import sys

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # .../gr-clenabled_amd


def _load():
    name = "gr_clenabled_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


_pkg = _load()
for _k in dir(_pkg.blocks):
    if not _k.startswith("_"):
        globals()[_k] = getattr(_pkg.blocks, _k)
del _k
