"""graphlily_amd -- MI355X-native SpMV / SpMSpV-with-semiring hot path behind GraphLily's
graphlily::module operator API.

  csrc/      hand-written HIP kernels (gfx950) + the C ABI  -> lib/libgraphlily_hip.so
  capi.py    ctypes binding of include/graphlily_hip.h
  io.py      graphlily::io host-side containers / formatters
  module.py  graphlily::module operator classes (Python mirror; C++ twin in include/graphlily/)
  app.py     BFS / PageRank / SSSP drivers written against the module API
  dist.py    row-range partitioning + all-gather exchange for one-process-per-GPU runs
"""
from . import capi, io, module  # noqa: F401

__all__ = ["capi", "io", "module"]
