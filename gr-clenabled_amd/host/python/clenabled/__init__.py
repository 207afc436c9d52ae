"""`import clenabled` of an installed build: the pybind11 module of the C++ block classes (host/python/bindings).

With GNU Radio the classes are flowgraph blocks (what grc/clenabled_*.block.yml constructs).  The repository's own test suite uses
the ctypes mirror `gr-clenabled_amd/python/clenabled` instead (same names, same positional arguments, no compiled module needed)."""
from .clenabled_python import *  # noqa: F401,F403
