"""Committed golden vectors (tests/golden/, self-generated -- see make_golden.py):
the C oracle reproduces them on CPU, the HIP library reproduces them on the GPU."""
import glob
import os

import numpy as np
import pytest

from util import engine, same, diff_report

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(HERE, "golden", "*.npz")))


def replay(api, name, max_step=None):
    z = np.load(os.path.join(HERE, "golden", name + ".npz"))
    nx, ny, ic, dt, cast = (int(v) for v in z["meta"])
    e = engine(api, nx, ny, "f64" if dt == 0 else "f32", "f32" if cast else "none", ic=ic)
    assert same(e.get("F"), z["F_0"]), diff_report(e.get("F"), z["F_0"], "F_0")
    done = 0
    for st in (int(s) for s in z["steps"]):
        if max_step and st > max_step:
            break
        e.step(st - done)
        done = st
        for f in ("F", "u", "v", "p"):
            a, b = e.get(f), z["%s_%d" % (f, st)]
            assert same(a, b), "%s step %d %s" % (name, st, diff_report(a, b, f))
    return e, z


def test_fixtures_present():
    assert "dam128_f64" in CASES and len(CASES) >= 6


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(oracle_api, name):
    replay(oracle_api, name, max_step=100)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_reproduces_golden(hip_api, name):
    e, z = replay(hip_api, name)
    if name == "dam128_f64":
        # BASELINE north_star bar: F L-inf <= 1e-5 at step 1000 (we get exact equality)
        assert np.max(np.abs(e.get("F") - z["F_1000"])) <= 1e-5
