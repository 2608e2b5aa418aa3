"""CPU-side checks of the drop-in boundary: the product library loads, exports every
symbol include/vof2d.h declares, and validates arguments without touching a GPU."""
import ctypes as C
import os
import re

import pytest

from vof2d import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "vof2d.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vof_[A-Za-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_list_agree():
    assert header_symbols() == sorted("vof_" + n for n in _abi.SIGNATURES)


def test_library_exports_every_declared_symbol(hip_api):
    for sym in header_symbols():
        assert hasattr(hip_api.lib, sym), sym
    assert hip_api.backend() == b"hip-gfx950"


def test_desc_default_matches_reference_constants(hip_api):
    d = _abi.Desc()
    assert hip_api.desc_default(C.byref(d), 200, 200, _abi.VOF_F32) == 0
    # 2dvof.py:19-33
    assert (d.nx, d.ny, d.row_lo, d.row_hi, d.own_lo, d.own_hi, d.jacobi_iters) == (200, 200, 0, 201, 1, 200, 10)
    assert (d.Lx, d.Ly, d.rho_l, d.rho_g, d.nu_l, d.nu_g) == (0.1, 0.1, 1000.0, 50.0, 1.0e-6, 1.5e-5)
    assert (d.sigma, d.gx, d.gy, d.dt) == (0.007, 0.0, -5.0, 4e-6)
    assert d.coord_cast_f32 == 1 and d.abi_version == _abi.VOF_ABI_VERSION


def test_bad_arguments_are_rejected_without_a_gpu(hip_api):
    d = _abi.Desc()
    assert hip_api.desc_default(C.byref(d), 2, 200, _abi.VOF_F64) == _abi.VOF_EINVAL
    assert hip_api.desc_default(C.byref(d), 64, 64, 7) == _abi.VOF_EINVAL
    assert hip_api.desc_default(C.byref(d), 64, 64, _abi.VOF_F64) == 0
    h = _abi.H()
    d.abi_version = 99
    assert hip_api.create(C.byref(d), None, C.byref(h)) == _abi.VOF_EINVAL
    d.abi_version = _abi.VOF_ABI_VERSION
    d.row_hi = 70
    assert hip_api.create(C.byref(d), None, C.byref(h)) == _abi.VOF_EINVAL
    assert hip_api.step(None, 1) == _abi.VOF_EINVAL
    assert hip_api.last_error(None) == b"null handle"


def test_oracle_exports_the_same_abi(oracle_api):
    assert oracle_api.backend() == b"cpu-oracle"
