"""Loader of the product library libvof2d_hip.so (HIP/gfx950 kernels + C ABI).

There is deliberately no fallback: if the library is missing or cannot be
loaded, importing the solver fails loudly.  Build it with
``make -C taichi-2d-vof_amd/csrc`` or ``python -c "import __graft_entry__ as g; g.build()"``.
"""
import ctypes
import os

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.normpath(os.path.join(_HERE, "..", "csrc", "build", "libvof2d_hip.so"))

_api = None


def hip_api():
    """Bound `_abi.Api` of libvof2d_hip.so (loaded once)."""
    global _api
    if _api is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libvof2d_hip.so not found at %s -- the HIP extension is not built; "
                "run `make -C taichi-2d-vof_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        api = _abi.bind(lib, "vof_")
        backend = api.backend().decode()
        if backend != "hip-gfx950":
            raise ImportError("unexpected backend %r in %s" % (backend, LIB_PATH))
        _api = api
    return _api
