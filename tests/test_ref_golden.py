"""Vectors produced by executing the reference's own source text (tests/golden/ref_*.npz,
made by tests/golden/make_ref_golden.py in the build container: /root/reference/2dvof.py run
unmodified under a pure-Python stand-in for the taichi module, 200 x 200 as shipped -- in doubles, and in
the shipped precision f32 with Taichi's static typing emulated (--f32) -- plus runs of the same text with
only its grid-size literals replaced: rectangular cells, BASELINE configs[0] (128 x 128), and the display
path (--vis: the event loop fed SPACE releases)).

These pin the oracle -- and through it, or directly, the HIP kernels -- to the reference's text
rather than to a reading of it: every field the reference holds (19 arrays) after steps 0..10, every
10th step and the last one, as sha256 digests of the values, plus F, u, v, p in full at a few steps.
"""
import glob
import hashlib
import os

import numpy as np
import pytest

from util import engine, diff_report

HERE = os.path.dirname(os.path.abspath(__file__))
REF_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(HERE, "golden", "ref_*.npz")))
ALL_FIELDS = ("F", "Ftd", "ax", "ay", "cx", "cy", "rp", "rm", "u", "v", "u_star", "v_star", "p", "pt",
              "rho", "nu", "mx", "my", "kappa")
STATE = ("F", "u", "v", "p")


def digest(a):
    """sha256 of the values (+0.0 folds the sign of an exact zero, which parity does not compare)."""
    return hashlib.sha256(np.ascontiguousarray(a + 0.0).tobytes()).hexdigest()


class Ref:
    def __init__(self, name):
        self.z = np.load(os.path.join(HERE, "golden", name + ".npz"))
        self.nx, self.ny, self.ic = (int(v) for v in self.z["meta"][:3])
        self.dtype = "f32" if int(self.z["meta"][3]) == 1 else "f64"     # f32: the stand-in's --f32 typing emulation
        names = [str(n) for n in self.z["digest_names"]]
        self.sha = dict(zip(names, (str(s) for s in self.z["digest_sha256"])))
        self.sum = dict(zip(names, self.z["digest_sum"]))
        self.absmax = dict(zip(names, self.z["digest_absmax"]))
        self.steps = sorted({int(n.rsplit("_", 1)[1]) for n in names})
        self.nsteps = int(self.z["nsteps"]) if "nsteps" in self.z.files else self.steps[-1]

    def check(self, arr, field, st, who):
        key = "%s_%d" % (field, st)
        if digest(arr) == self.sha[key]:
            return
        if key in self.z.files:
            detail = diff_report(arr, self.z[key], key)
        else:
            detail = "sum %r vs %r, max|.| %r vs %r" % (float(arr.sum()), float(self.sum[key]),
                                                        float(np.abs(arr).max()), float(self.absmax[key]))
        raise AssertionError("%s differs from the reference's own output at %s: %s" % (who, key, detail))


def replay(api, name, fields, who, max_step=None):
    ref = Ref(name)
    e = engine(api, ref.nx, ref.ny, ref.dtype, "f32", ic=ref.ic)
    done = 0
    for st in ref.steps:
        if max_step is not None and st > max_step:
            break
        e.step(st - done)
        done = st
        for f in fields:
            ref.check(e.get(f), f, st, who)
    return e, ref


SHIPPED = ("ref_ic1_200_f64", "ref_ic2_200_f64", "ref_ic3_200_f64")
# the reference exactly as shipped -- default_fp = ti.f32 (2dvof.py:9) -- under the stand-in's emulation of
# Taichi's static typing (make_ref_golden.py --f32): the reference's own dtype, pinned by its own text
SHIPPED_F32 = ("ref_ic1_200_f32", "ref_ic2_200_f32", "ref_ic3_200_f32")
# the reference text with only its two grid-size literals (:19-20) replaced: rectangular cells
# (dx != dy -- the general Jacobi stencil and the `qp = (fmax - Ftd) * dx` of the y sweep, :417), odd sizes
RESIZED = ("ref_ic1_96x40_f64", "ref_ic2_48x80_f64", "ref_ic3_33x17_f64")


# configs[0] of BASELINE.json (128 x 128 dam-break, 1000 steps): the reference text itself, grid literals replaced
BASELINE0 = "ref_ic1_128x128_f64"
# runs whose event loop was fed a SPACE release after every display (make_ref_golden.py --vis): the reference
# walks through its five display branches (:531-559); rgb_buf / V / the gui.arrows arguments are recorded
VIS_CASES = tuple(n for n in REF_CASES if n.endswith("_vis"))


def test_reference_vectors_present():
    assert set(SHIPPED) | set(RESIZED) | {BASELINE0} <= set(REF_CASES) and len(VIS_CASES) >= 3
    ref = Ref(BASELINE0)
    assert (ref.nx, ref.ny, ref.ic, ref.nsteps) == (128, 128, 1, 1000) and "F_1000" in ref.z.files
    shas = {str(Ref(n).z["ref_sha256"]) for n in REF_CASES if "ref_sha256" in Ref(n).z.files}
    assert len(shas) == 1 and len(shas.pop()) == 64       # the newer files say which 2dvof.py they came from
    f32_cases = [n for n in REF_CASES if Ref(n).dtype == "f32"]
    assert set(SHIPPED_F32) | {"ref_ic2_48x80_f32", "ref_ic3_40x56_f32_vis"} <= set(f32_cases)   # as shipped; rectangular cells; the display path
    for name in SHIPPED + SHIPPED_F32:
        ref = Ref(name)
        assert ref.dtype == ("f32" if name in SHIPPED_F32 else "f64") and ref.z["F_1000"].dtype == (np.float32 if name in SHIPPED_F32 else np.float64)
        assert (ref.nx, ref.ny) == (200, 200) and ref.nsteps == 1000      # the shipped size; odd and even istep
        # constants the reference derived at Python scope (2dvof.py:47-50)
        c = dict(zip((str(k) for k in ref.z["const_names"]), ref.z["const"]))
        # (sigma is a 0-D field, :28-29: read back at Python scope it is the stored f32 value in the f32 runs)
        assert c["dx"] == 0.00050000002374872565 and c["dt"] == 4e-6
        assert c["sigma"] == (float(np.float32(0.007)) if ref.dtype == "f32" else 0.007)
    for name in RESIZED:
        ref = Ref(name)
        c = dict(zip((str(k) for k in ref.z["const_names"]), ref.z["const"]))
        assert ref.nx != ref.ny and c["dx"] != c["dy"] and ref.nsteps >= 300


@pytest.mark.parametrize("name", REF_CASES)
def test_oracle_reproduces_reference_run(oracle_api, name):
    """All 19 arrays of 2dvof.py:53-89 at every recorded step, C restatement vs the reference's text."""
    replay(oracle_api, name, ALL_FIELDS, "oracle (C)")


@pytest.mark.parametrize("name", REF_CASES)
def test_numpy_oracle_reproduces_reference_run(name):
    import vof_oracle_np as onp
    ref = Ref(name)
    s = onp.new_state(ref.nx, ref.ny, ref.ic, dtype=np.float32 if ref.dtype == "f32" else np.float64, coord_cast="f32")
    done = 0
    for st in [t for t in ref.steps if t <= 20]:
        onp.step(s, st - done)
        done = st
        for f in ALL_FIELDS:
            ref.check(getattr(s, f), f, st, "oracle (NumPy)")


def check_display_path(api, name, who):
    """rgb_buf after get_vof_field / get_u_field / get_v_field / get_vnorm_field, V after interp_velocity and
    the (orig, direction) arrays plot_arrow_field handed to gui.arrows, at the 100-step marks of a --vis run."""
    from vof2d import vis
    ref = Ref(name)
    e = engine(api, ref.nx, ref.ny, ref.dtype, "f32", ic=ref.ic)
    seen = set()
    for st in (int(s) for s in ref.z["vis_steps"]):
        e.step(st - e.istep)
        opt = int(ref.z["vis_option_%d" % st])
        seen.add(opt)
        got = {}
        if opt < 4:
            got["rgb"] = e.vis_field(vis.OPTIONS[opt][1])
        else:
            got["V"] = e.interp_velocity()
            got["orig"], got["direction"] = vis.arrow_field(got["V"], 4)      # arrow_spacing=4, 2dvof.py:557
        for k, arr in got.items():
            full, sha = "vis_%s_%d" % (k, st), "vis_%s_sha256_%d" % (k, st)
            if full in ref.z.files:
                assert arr.shape == ref.z[full].shape and np.array_equal(arr, ref.z[full]), \
                    "%s: %s" % (who, diff_report(arr.reshape(arr.shape[0], -1), ref.z[full].reshape(arr.shape[0], -1), full))
            else:
                assert digest(arr) == str(ref.z[sha]), "%s differs from the reference's own %s" % (who, sha)
    assert seen == {0, 1, 2, 3, 4}
    assert int(ref.z["gui_shown"]) == ref.nsteps // 100


@pytest.mark.parametrize("name", VIS_CASES)
def test_oracle_display_path_matches_reference_run(oracle_api, name):
    check_display_path(oracle_api, name, "oracle (C)")


@pytest.mark.parametrize("name", VIS_CASES)
def test_numpy_oracle_display_path(name):
    import vof_oracle_np as onp
    from vof2d import vis
    ref = Ref(name)
    if ref.nx * ref.ny > 5000:
        pytest.skip("the NumPy restatement is checked on the small display runs")
    s = onp.new_state(ref.nx, ref.ny, ref.ic, dtype=np.float32 if ref.dtype == "f32" else np.float64, coord_cast="f32")
    for st in (int(x) for x in ref.z["vis_steps"]):
        onp.step(s, st - s.istep)
        opt = int(ref.z["vis_option_%d" % st])
        if opt < 4:
            assert np.array_equal(onp.vis_field(s, vis.OPTIONS[opt][1]), ref.z["vis_rgb_%d" % st]), (st, opt)
        else:
            assert np.array_equal(onp.interp_velocity(s), ref.z["vis_V_%d" % st])


@pytest.mark.parametrize("name", REF_CASES)
def test_reference_png_and_gui_path(name):
    """-s of the reference itself (:563-571): one PNG per 100 steps, numbered from 000000."""
    ref = Ref(name)
    if name not in SHIPPED + SHIPPED_F32:
        pytest.skip("-s was passed to the shipped-size runs only")
    pngs = [str(p) for p in ref.z["pngs"]]
    assert pngs == ["%06d-f.png" % k for k in range(ref.nsteps // 100)]
    assert int(ref.z["gui_shown"]) == ref.nsteps // 100


@pytest.mark.gpu
@pytest.mark.parametrize("name", REF_CASES)
def test_hip_reproduces_reference_run(hip_api, name):
    """The fused HIP step (what bench.py times) vs the reference's text: F, u, v, p and the
    predictor's u*, v* at every recorded step up to the last (step 1000 for the committed files)."""
    e, ref = replay(hip_api, name, STATE + ("u_star", "v_star"), "HIP fused step")
    # north-star bar: F L-inf <= 1e-5 at step 1000 (we get exact equality)
    last = "F_%d" % ref.nsteps
    if last in ref.z.files:
        assert np.max(np.abs(e.get("F") - ref.z[last])) <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name", REF_CASES)
def test_hip_verbs_reproduce_reference_run(hip_api, name):
    """The verb-by-verb HIP path (one C-ABI call per reference kernel, 2dvof.py:513-528)."""
    from vof2d.solver import VOF2D
    ref = Ref(name)
    s = VOF2D(ref.nx, ref.ny, dtype=ref.dtype, coord_cast="f32", api=hip_api)
    s.set_init_F(ref.ic)
    done = 0
    for st in [t for t in ref.steps if t <= 30]:
        s.step_verbs(st - done)
        done = st
        for f in STATE + ("u_star", "v_star", "rho", "nu", "mx", "my", "kappa"):
            ref.check(s.eng.get(f), f, st, "HIP verbs")
    s.close()
