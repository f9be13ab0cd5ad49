"""denseflow_amd — MI355X-native dense optical flow behind denseflow's calc_optflows_imp boundary.

This package is a thin ctypes binding of the C ABI in include/dfx.h (implemented by the
hand-written HIP kernels under denseflow_amd/csrc/).  It exists so that tests and bench.py can
drive the exact entry points a denseflow maintainer would bind; the compute path is entirely
inside libdfx.so.  There is no CPU fallback: a missing library or GPU raises.
"""
from .engine import (  # noqa: F401
    DfxError,
    DfxParams,
    DfxStats,
    FlowEngine,
    algo_from_name,
    build_library,
    device_count,
    library_path,
    load_library,
)

__all__ = [
    "DfxError",
    "DfxParams",
    "DfxStats",
    "FlowEngine",
    "algo_from_name",
    "build_library",
    "device_count",
    "library_path",
    "load_library",
]
