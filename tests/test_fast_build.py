"""The FMA-contracted build (libvof2d_hip_fast.so) is built, loads, runs -- and is
NOT the product: its F differs from the parity build's (= the oracle's = the reference run's)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fast_library_is_built_and_exports_the_abi():
    import ctypes
    from vof2d import _abi, _lib
    assert os.path.exists(_lib.FAST_LIB_PATH), "make -C taichi-2d-vof_amd/csrc builds both libraries"
    lib = ctypes.CDLL(_lib.FAST_LIB_PATH, mode=ctypes.RTLD_LOCAL)
    for name in _abi.SIGNATURES:
        assert hasattr(lib, "vof_" + name), name


def test_product_loader_refuses_to_mix_the_two_builds():
    code = ("import sys; sys.path.insert(0, %r); from vof2d._lib import hip_api; hip_api();\n"
            "try:\n    hip_api(fast=True)\nexcept ImportError as e:\n    print('refused')\n" % os.path.join(ROOT, "taichi-2d-vof_amd"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "refused" in r.stdout, r.stderr


@pytest.mark.gpu
def test_fast_leg_reports_speed_and_difference():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--fast-leg", "--nx", "512", "--steps", "4",
                        "--warmup", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    assert d["value"] > 0 and "ffp-contract=fast" in d["build"]
    # contraction changes results: tiny after 100 steps, beyond rounding after 1000 (chaotic growth)
    assert 0.0 <= d["F_Linf_vs_parity_build_128x128_step_100"] < 1e-6
    assert d["F_Linf_vs_parity_build_128x128_step_1000"] >= d["F_Linf_vs_parity_build_128x128_step_100"]
