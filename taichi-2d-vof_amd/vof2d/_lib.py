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

FAST_LIB_PATH = LIB_PATH.replace("libvof2d_hip.so", "libvof2d_hip_fast.so")

_api = None


def kernel_source_hash():
    """sha256 over what decides a kernel's traffic per launch: the HIP kernels (csrc/vof2d_device.h,
    vof2d_kernels.h, kernels/*.h) and the launch geometry (runtime/launches.h: chunk-length heuristics, grids
    and arguments).  Profiles that quote per-kernel hardware counters record it
    (profiles/jacobi_pmc.json), and bench.py only repeats such a number while it matches the built sources."""
    import hashlib
    src = os.path.normpath(os.path.join(_HERE, "..", "csrc"))
    h = hashlib.sha256()
    files = ["vof2d_device.h", "vof2d_kernels.h", os.path.join("runtime", "launches.h")]
    files += [os.path.join("kernels", n) for n in os.listdir(os.path.join(src, "kernels")) if n.endswith(".h")]
    for rel in sorted(files):
        h.update(rel.encode() + b"\0")
        with open(os.path.join(src, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def hip_api(fast=False):
    """Bound `_abi.Api` of libvof2d_hip.so (loaded once).

    fast=True: the FMA-contracted build (libvof2d_hip_fast.so) instead -- measurement only (bench.py's
    fast leg, tests/test_fast_build.py), its results are not the reference's.  One process loads one
    of the two (both export the same symbols)."""
    global _api
    path = FAST_LIB_PATH if fast else LIB_PATH
    if _api is not None and getattr(_api, "path", path) != path:
        raise ImportError("this process already loaded %s" % _api.path)
    if _api is None:
        if not os.path.exists(path):
            raise ImportError(
                "%s not found -- the HIP extension is not built; "
                "run `make -C taichi-2d-vof_amd/csrc` (there is no CPU fallback)" % path)
        lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        api = _abi.bind(lib, "vof_")
        backend = api.backend().decode()
        if backend != "hip-gfx950":
            raise ImportError("unexpected backend %r in %s" % (backend, path))
        api.path = path
        _api = api
    return _api
