"""Generates the committed golden fixtures (tests/golden/*.npz).

SELF-GENERATED, NOT TAICHI OUTPUT (the vectors made by executing the reference's
own text are ref_*.npz, see make_ref_golden.py): taichi==1.4.1 cannot be installed in this
image, so these fixtures come from the two independent restatements of
/root/reference/2dvof.py in oracle/ (NumPy-vectorised and scalar C); the script
refuses to write a fixture unless both agree value for value.

    python tests/golden/make_golden.py
"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "taichi-2d-vof_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]

import vof_oracle_np as onp  # noqa: E402
from util import engine  # noqa: E402
from vof2d import _abi  # noqa: E402

NPDT = {"f64": np.float64, "f32": np.float32}
# name: (nx, ny, ic, dtype, coord_cast, steps at which F,u,v,p are stored)
CASES = {
    "dam128_f64": (128, 128, 1, "f64", "f32", (100, 1000)),      # BASELINE configs[0], the parity config
    "dam32_f64": (32, 32, 1, "f64", "f32", (1, 2, 10, 100)),
    "bubble33x17_f64": (33, 17, 2, "f64", "f32", (1, 2, 10, 100)),
    "drop24x40_f64_nocast": (24, 40, 3, "f64", "none", (1, 2, 10, 100)),
    "dam200_f32": (200, 200, 1, "f32", "f32", (1, 10, 100)),     # the reference's shipped size/dtype
    "bubble48_f32": (48, 48, 2, "f32", "f32", (1, 10, 100)),
}


def main():
    api = _abi.bind(ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libvof_oracle.so")), "ovof_",
                    optional=_abi.GPU_ONLY)
    for name, (nx, ny, ic, dtype, cast, steps) in CASES.items():
        s = onp.new_state(nx, ny, ic, dtype=NPDT[dtype], coord_cast=cast)
        e = engine(api, nx, ny, dtype, cast, ic=ic)
        out = {"meta": np.array([nx, ny, ic, 0 if dtype == "f64" else 1, 1 if cast == "f32" else 0]),
               "steps": np.array(steps), "F_0": s.F.copy()}
        done = 0
        for st in steps:
            onp.step(s, st - done)
            e.step(st - done)
            done = st
            for f in ("F", "u", "v", "p"):
                a, b = getattr(s, f), e.get(f)
                if not np.array_equal(a, b):
                    raise SystemExit("%s step %d field %s: NumPy and C restatements disagree" % (name, st, f))
                out["%s_%d" % (f, st)] = a.copy()
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, "->", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
