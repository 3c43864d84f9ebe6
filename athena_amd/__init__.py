"""athena_amd -- MI355X-native message-passing engine behind athena's layer API.

Only the hot path lives here: the forward/backward of athena's
kipf / duvenaud / graph_nop message-passing layers as hand-written gfx950 HIP kernels behind a
C ABI (include/athena_mp.h), plus the host-side mirrors of the reference interface.
"""
from . import _capi  # noqa: F401
from .graph import DeviceGraph, graph_type  # noqa: F401

__all__ = ["_capi", "DeviceGraph", "graph_type"]
