"""GPU checks of the host-side pieces around the C ABI: the reference-shaped Python interface,
the drop-in command line, and the torch aliasing of library-owned device memory that the RCCL
halo exchange relies on."""
import os
import subprocess
import sys

import numpy as np
import pytest

from util import engine, same, diff_report

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_shaped_interface(hip_api, oracle_api):
    """VOF2D: fields with to_numpy/from_numpy, sigma[None], kernels by their reference names, and
    the literal main loop (step_verbs) equal to the fused step and to the oracle."""
    from vof2d import VOF2D
    a = VOF2D(64, 48, dtype="f64")
    b = VOF2D(64, 48, dtype="f64")
    ref = engine(oracle_api, 64, 48, "f64", "f32", ic=3)
    for s in (a, b):
        s.set_init_F(3)
    assert a.F.shape == (66, 50) and a.sigma[None] == 0.007
    a.step(9)
    b.step_verbs(9)
    ref.step(9)
    for f in ("F", "u", "v", "p"):
        x, y, z = getattr(a, f).to_numpy(), getattr(b, f).to_numpy(), ref.get(f)
        assert same(x, y), diff_report(x, y, f + " fused vs verbs")
        assert same(x, z), diff_report(x, z, f + " vs oracle")
    Fn = a.F.to_numpy()
    Fn[5:9, 7] = 0.25
    a.F.from_numpy(Fn)
    assert a.F[6, 7] == 0.25 and a.istep == 9
    a.sigma[None] = 0.01
    assert a.sigma[None] == 0.01


def test_command_line_is_a_drop_in(tmp_path):
    """python 2dvof.py -ic 2 -s: banner, status line every 100 steps, output/NNNNNN-f.png (2dvof.py:95-99,533,563-571)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "2dvof.py"), "-ic", "2", "-s", "--steps", "200",
                        "--nx", "64", "--ny", "64", "--vis", "3"], cwd=tmp_path, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    assert ">>> Grid resolution: 64 x 64, dt = 4.00e-06" in out
    assert ">>> Density ratio:  20.00, gravity : -5.00, sigma :  0.01" in out
    assert ">>> Number of steps:100  , Time:4.00e-04 sec. Displaying velocity norm." in out
    assert ">>> Number of steps:200  , Time:8.00e-04 sec." in out
    assert (tmp_path / "output" / "000000-f.png").stat().st_size > 1000
    assert (tmp_path / "output" / "000001-f.png").exists() and (tmp_path / "data").is_dir()
    assert (tmp_path / "output" / "000000-vis.png").stat().st_size > 200


def test_strip_solver_aliases_device_memory():
    """StripSolver (world = 1) on the GPU: torch tensors alias the library's field memory
    (what RCCL send/recv operate on), the solver runs on a torch stream, results equal Engine's."""
    torch = pytest.importorskip("torch")
    from vof2d.strips import StripSolver
    from vof2d._lib import hip_api
    s = StripSolver(96, 64, "f64", ic=1, rank=0, world=1, device=0)
    e = engine(hip_api(), 96, 64, "f64", "f32", ic=1)
    s.step(7); e.step(7)
    s.sync()
    for f in ("F", "u", "v", "p"):
        assert same(s.gather(f), e.get(f)), f
    t = s._rows_view("p", 10, 12)
    assert t.is_cuda and t.shape[0] == 3 and t.is_contiguous()
    base, pitch, col0, _ = s.eng.field_view("p")
    assert t.data_ptr() == base + 10 * pitch * 8
    with torch.cuda.stream(s.stream):
        t[:, col0 + 1: col0 + 65] = 3.5          # write through torch ...
    torch.cuda.synchronize()
    assert np.all(s.eng.get("p", (10, 12))[:, 1:65] == 3.5)   # ... read back through the C ABI


def test_display_fields(hip_api, oracle_api):
    """get_vof_field / get_u_field / get_v_field / get_vnorm_field / interp_velocity (2dvof.py:458-492)."""
    for dtype in ("f64", "f32"):
        a, b = engine(hip_api, 40, 56, dtype, "f32", ic=3), engine(oracle_api, 40, 56, dtype, "f32", ic=3)
        a.step(40); b.step(40)
        for which in ("vof", "u", "v", "vnorm"):
            x, y = a.vis_field(which), b.vis_field(which)
            assert x.shape == (80, 112) and same(x, y), diff_report(x, y, which)
        assert same(a.interp_velocity(), b.interp_velocity())
    from vof2d.engine import VofError
    with pytest.raises(VofError):
        a.vis_field("pressure")
    strip = engine(hip_api, 64, 32, "f64", "f32", ic=1, rows=(0, 40), own=(1, 24))
    with pytest.raises(VofError):
        strip.vis_field("vof")


def _loopback_worker(*args):
    """The RCCL loopback checks run in a fresh, torch-free process (tests/_loopback_worker.py): an
    earlier test of THIS process imports torch, whose bundled RCCL 2.26.6 the library would then bind,
    and that copy cannot capture the exchange -- the tests would silently exercise the eager path."""
    env = {k: v for k, v in os.environ.items() if k not in ("VOF2D_RCCL", "VOF2D_XCHG_GRAPH")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_loopback_worker.py")] + [str(a) for a in args],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "OK" in r.stdout.split(), (r.stdout[-1500:], r.stderr[-3000:])   # (RCCL prints its banner at exit)


def test_native_rccl_exchange_loopback():
    """vof_comm_init / vof_comm_exchange / vof_step_exchange on one GPU with both neighbours looped
    back to the calling rank, all four overlap modes, eager first step and captured graphs."""
    _loopback_worker("modes")


@pytest.mark.parametrize("own", [(41, 120), (1, 80), (81, 160), (70, 95)])
def test_exchange_mode4_equals_phases_plus_copies(own):
    """vof_step_exchange overlap 4 -- the default of bench.py --gpus N -- replayed from the captured
    graph, on an interior strip, next to either wall and on a strip whose bands meet."""
    _loopback_worker("mode4", own[0], own[1])


@pytest.mark.parametrize("own", [(61, 140), (1, 100), (101, 200), (90, 117)])
def test_exchange_mode5_equals_pieces_plus_copies(own):
    """vof_step_exchange overlap 5 -- the strips run the pair kernels of the single GPU, F, u*, v*, rhs, p exchanged once
    per step -- replayed from captured graphs on an interior strip, next to either wall and on a strip whose bands meet."""
    _loopback_worker("mode5", own[0], own[1])


@pytest.mark.parametrize("own,dtype,iters", [((61, 140), "f32", 10), ((1, 100), "f32", 10), ((71, 150), "f64", 20), ((101, 200), "f32", 20)])
def test_exchange_mode5_in_fp32_and_with_more_sweeps(own, dtype, iters):
    """Overlap mode 5 in fp32 (what bench.py --gpus N --dtype f32 runs since round 6) and with 20 sweeps per step: every
    middle step runs jacobi_iters / 10 launches of k_jacobi_pair (round 5 ran one whatever the count: ADVICE r05)."""
    _loopback_worker("mode5", own[0], own[1], dtype, iters)


def test_exchange_modes_4_and_5_on_random_strips():
    """The two deterministic exchange modes on strips, grids, precisions, sweep counts, chunk knobs and call lengths drawn from a
    seed (tests/_loopback_worker.py fuzz; a campaign of hundreds of cases: profiles/r06_fuzz.md)."""
    _loopback_worker("fuzz", int(os.environ.get("VOF_FUZZ_SEED", "20261003")), 10)


def test_command_line_residual_terminated_solve(tmp_path):
    """2dvof.py --jacobi-tol / --jacobi-crit (extension): the main loop :513-528 with vof_solve_p in place of
    the ten fixed sweeps runs headless and reports like the reference."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "2dvof.py"), "-ic", "1", "--dtype", "f64", "--nx", "96", "--ny", "64",
                        "--jacobi-tol", "1e-3", "--jacobi-crit", "rel", "--jacobi-max", "4000", "--steps", "100"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert ">>> Grid resolution: 96 x 64, dt = 4.00e-06" in r.stdout
    assert ">>> Number of steps:100  , Time:4.00e-04 sec. Displaying VOF field." in r.stdout


def _cli(tmp, *argv, timeout=600):
    os.makedirs(tmp, exist_ok=True)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "2dvof.py")] + [str(a) for a in argv], cwd=tmp,
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return r.stdout


def test_command_line_gpus_jacobi_iters_checkpoint_resume(tmp_path, oracle_api):
    """2dvof.py --gpus 1 --jacobi-iters N --save-every / --resume on the HIP library: the checkpoints equal
    the oracle's state, a resumed run reaches the same final state as the uninterrupted one, and the
    PNG of the run is byte-identical either way (2dvof.py:500-501, :521, :563-571)."""
    base = ["-ic", "1", "-s", "--gpus", "1", "--nx", "96", "--ny", "64", "--dtype", "f64", "--jacobi-iters", "12",
            "--save-every", "70", "--steps", "210"]
    out = _cli(tmp_path / "a", *base)
    assert ">>> Grid resolution: 96 x 64, dt = 4.00e-06" in out and "row strips" not in out
    _cli(tmp_path / "b", *base, "--resume", tmp_path / "a" / "data" / "00000070.npz")
    ref = engine(oracle_api, 96, 64, "f64", "f32", ic=1, jacobi_iters=12)
    for st in (70, 140, 210):
        ref.step(st - ref.istep)
        za = np.load(tmp_path / "a" / "data" / ("%08d.npz" % st))
        for f in ("F", "u", "v", "p"):
            assert same(za[f], ref.get(f)), diff_report(za[f], ref.get(f), "%s at step %d" % (f, st))
            if st > 70:
                zb = np.load(tmp_path / "b" / "data" / ("%08d.npz" % st))
                assert same(za[f], zb[f]), diff_report(za[f], zb[f], "%s resumed, step %d" % (f, st))
    for k in (0, 1):
        pa = (tmp_path / "a" / "output" / ("%06d-f.png" % k)).read_bytes()
        if k == 1:
            assert pa == (tmp_path / "b" / "output" / ("%06d-f.png" % k)).read_bytes()


def test_display_path_matches_reference_run(hip_api):
    """The display kernels (2dvof.py:458-492) and the arrow list of flow_visualization.py:35-55 against what
    the reference's own text produced when its event loop was fed SPACE releases (ref_*_vis.npz)."""
    import test_ref_golden as trg
    for name in trg.VIS_CASES:
        trg.check_display_path(hip_api, name, "HIP")
