"""vof2d -- MI355X-native drop-in for the hot path of taichi-2d-vof's 2dvof.py.

Host-side mirror of the reference interface (kernel verbs + fields) over the
C ABI of libvof2d_hip.so (include/vof2d.h).
"""
from ._abi import VOF_F32, VOF_F64, VOF_FLAG_NO_GRAPH, halo_rows  # noqa: F401
from .engine import Engine, VofError, make_desc  # noqa: F401
from .solver import VOF2D, Field  # noqa: F401
