"""MI355X-native geometric-multigrid hot path (drop-in for the GeometricMultigrid cycle of
Stefo01/multigrid_prj).  The product is libmg_hip.so (include/mg_hip.h); this package is
the thin Python host mirror over its C-ABI used by tests and bench.py."""
from . import build, capi  # noqa: F401
from .capi import MgCycleStats, MgDesc, MgError, Solver, make_desc  # noqa: F401
