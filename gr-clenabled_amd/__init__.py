"""gr-clenabled_amd: MI355X (gfx950) implementation of gr-clenabled's streaming-DSP hot path.

The directory name carries a hyphen (it mirrors the reference's module name), so
load it with ``importlib`` under the module name ``gr_clenabled_amd`` -- see
``__graft_entry__.load_package()`` -- or put the repo root on sys.path and call
``importlib.import_module("gr-clenabled_amd")``.
"""
from ._lib import LIB_PATH, Mi355Error, lib, set_log_callback  # noqa: F401
from .blocks import *  # noqa: F401,F403
from . import blocks as clenabled  # noqa: F401  (flowgraph-style alias: clenabled.clFFT(...))
from . import shard  # noqa: F401
