"""The command-line program (vof2d/cli.py, what 2dvof.py runs) on CPU: the CPU oracle is handed in
as the engine (same C ABI; the product itself only ever loads the HIP library) and gloo carries the
ranks.  Reference behaviour checked: banner :95-99, status line :533, output/NNNNNN-f.png :563-571,
output/ and data/ :500-501."""
import os
import socket

import numpy as np
import pytest

import _cli_worker


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(tmp, world, argv):
    import torch.multiprocessing as mp
    os.makedirs(tmp, exist_ok=True)
    mp.spawn(_cli_worker.run, args=(world, _free_port(), list(argv), str(tmp)), nprocs=world, join=True)
    return open(os.path.join(tmp, "stdout.0")).read()


ARGS = ["-ic", "1", "-s", "--nx", "72", "--ny", "40", "--dtype", "f64", "--steps", "200", "--save-every", "100"]


def test_two_ranks_write_the_same_files_as_one(tmp_path):
    """`--gpus 2` (two row strips, F gathered on rank 0 every 100 steps) produces byte-identical
    output/NNNNNN-f.png files and identical checkpoints to the single-domain run."""
    out1 = _run(tmp_path / "w1", 1, ARGS)
    out2 = _run(tmp_path / "w2", 2, ARGS + ["--gpus", "2"])
    assert ">>> Grid resolution: 72 x 40, dt = 4.00e-06" in out1 and ", 2 row strips" in out2
    assert ">>> Density ratio:  20.00, gravity : -5.00, sigma :  0.01" in out1
    for out in (out1, out2):
        assert ">>> Number of steps:100  , Time:4.00e-04 sec. Displaying VOF field." in out
        assert ">>> Number of steps:200  , Time:8.00e-04 sec. Displaying VOF field." in out
    for k in (0, 1):
        a = (tmp_path / "w1" / "output" / ("%06d-f.png" % k)).read_bytes()
        b = (tmp_path / "w2" / "output" / ("%06d-f.png" % k)).read_bytes()
        assert len(a) > 1000 and a == b
    assert not (tmp_path / "w2" / "output" / "000000-vis.png").exists()     # the display kernels are single-domain
    assert (tmp_path / "w1" / "output" / "000000-vis.png").exists() and (tmp_path / "w2" / "data").is_dir()
    for st in (100, 200):
        z1 = np.load(tmp_path / "w1" / "data" / ("%08d.npz" % st))
        z2 = np.load(tmp_path / "w2" / "data" / ("%08d.npz" % st))
        assert int(z1["istep"]) == int(z2["istep"]) == st
        for f in ("F", "u", "v", "p"):
            assert z1[f].shape == (74, 42) and np.array_equal(z1[f], z2[f]), (st, f)


@pytest.mark.parametrize("world", [1, 2])
def test_resume_continues_the_run_exactly(tmp_path, world):
    """--resume from the step-100 checkpoint reaches the same step-200 state as the uninterrupted
    run (odd and even steps on both sides of the cut; ghost cells travel with the fields)."""
    extra = ["--gpus", str(world)] if world > 1 else []
    base = ["-ic", "3", "--nx", "64", "--ny", "48", "--dtype", "f64", "--save-every", "50", "--jacobi-iters", "12"] + extra
    _run(tmp_path / "a", world, base + ["--steps", "150"])
    out = _run(tmp_path / "b", world, base + ["--steps", "150", "--resume", str(tmp_path / "a" / "data" / "00000050.npz")])
    assert ">>> Resumed from" in out and "at step 50." in out
    za, zb = np.load(tmp_path / "a" / "data" / "00000150.npz"), np.load(tmp_path / "b" / "data" / "00000150.npz")
    for f in ("F", "u", "v", "p"):
        assert np.array_equal(za[f], zb[f]), f
    assert not (tmp_path / "b" / "data" / "00000050.npz").exists()


def test_two_ranks_with_a_residual_terminated_solve_equal_one(tmp_path):
    """`--gpus 2 --jacobi-tol`: the pressure solve of every step runs until the GLOBAL residual is below the tolerance
    (2dvof.py:521-522 has a fixed count; extension) -- norms all-reduced over the strips, p exchanged between the sweep
    batches -- and ends after the same number of sweeps as on one domain: same fields, same PNG bytes."""
    args = ["-ic", "2", "-s", "--nx", "64", "--ny", "48", "--dtype", "f64", "--steps", "100", "--save-every", "50",
            "--jacobi-tol", "2e-3", "--jacobi-max", "60", "--jacobi-crit", "abs"]
    out1 = _run(tmp_path / "w1", 1, args)
    out2 = _run(tmp_path / "w2", 2, args + ["--gpus", "2"])
    assert ">>> Number of steps:100" in out1 and ">>> Number of steps:100" in out2
    a = (tmp_path / "w1" / "output" / "000000-f.png").read_bytes()
    assert len(a) > 1000 and a == (tmp_path / "w2" / "output" / "000000-f.png").read_bytes()
    for st in (50, 100):
        z1 = np.load(tmp_path / "w1" / "data" / ("%08d.npz" % st))
        z2 = np.load(tmp_path / "w2" / "data" / ("%08d.npz" % st))
        for f in ("F", "u", "v", "p"):
            assert np.array_equal(z1[f], z2[f]), (st, f)
        assert float(z1["num_jacobi_tol"]) == 2e-3 and int(z2["num_jacobi_max"]) == 60
    # and the tolerance really decides the sweep count: a run with the reference's fixed ten sweeps differs
    _run(tmp_path / "w3", 1, [x for x in args if x not in ("--jacobi-tol", "2e-3", "--jacobi-max", "60", "--jacobi-crit", "abs")])
    z3 = np.load(tmp_path / "w3" / "data" / "00000100.npz")
    assert not np.array_equal(np.load(tmp_path / "w1" / "data" / "00000100.npz")["p"], z3["p"])


def test_resume_refuses_other_numerics(tmp_path):
    """A checkpoint records dt, the sweep count, the coordinate cast and the residual criterion; resuming it with other
    values is refused instead of silently continuing a different run.  The Courant count travels with it."""
    base = ["-ic", "1", "--nx", "40", "--ny", "40", "--dtype", "f64", "--save-every", "20", "--steps", "40"]
    _run(tmp_path / "a", 1, base + ["--jacobi-iters", "8"])
    ck = str(tmp_path / "a" / "data" / "00000020.npz")
    z = np.load(ck)
    assert int(z["num_jacobi_iters"]) == 8 and float(z["num_dt"]) == 4e-6 and str(z["num_coord_cast"]) == "f32"
    from vof2d import cli
    p = cli.build_parser()
    for extra, what in ((["--jacobi-iters", "10"], "jacobi-iters"), (["--jacobi-iters", "8", "--dt", "2e-6"], "dt"),
                        (["--jacobi-iters", "8", "--coord-cast", "none"], "coord-cast")):
        with pytest.raises(SystemExit) as e:
            cli.load_state(ck, 40, 40, "f64", cli.numerics_of(p.parse_args(base + extra), 2e-6 if "--dt" in extra else 4e-6))
        assert what in str(e.value)
    fields, istep, warn = cli.load_state(ck, 40, 40, "f64", cli.numerics_of(p.parse_args(base + ["--jacobi-iters", "8"]), 4e-6))
    assert istep == 20 and warn == int(z["courant_violations"]) and fields["F"].shape == (42, 42)


def test_a_launcher_without_gpus_flag_is_refused(monkeypatch):
    """Under torchrun with WORLD_SIZE > 1 but --gpus 1 every rank would run the whole problem on device 0 and race on
    output/ and data/: refused before any engine is made."""
    from vof2d import cli
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "1")
    with pytest.raises(SystemExit) as e:
        cli.run(cli.build_parser().parse_args(["--steps", "1"]), api=object())
    assert "--gpus 4" in str(e.value)


def test_flags_of_the_reference_and_refusals():
    from vof2d import cli
    p = cli.build_parser()
    a = p.parse_args([])
    assert (a.ic, a.s, a.nx, a.ny, a.dtype, a.jacobi_iters, a.gpus) == (1, False, 200, 200, "f32", 10, 1)   # 2dvof.py:9,13-14,19-20,521
    with pytest.raises(SystemExit):
        p.parse_args(["-ic", "4"])
    with pytest.raises(SystemExit):       # the single-GPU extensions are refused on strips, before any engine is made
        cli.run(p.parse_args(["--gpus", "2", "--verbs"]), api=object(), rank=0, world=2)
    with pytest.raises(SystemExit):
        cli.run(p.parse_args(["--gpus", "2", "--vis", "3"]), api=object(), rank=0, world=2)
    with pytest.raises(SystemExit):
        cli.run(p.parse_args(["--gpus", "2"]), api=object(), rank=0, world=3)


def test_main_becomes_the_launcher_only_when_nobody_else_did(monkeypatch):
    """`2dvof.py --gpus N` with no WORLD_SIZE / RANK around it hands the same argv to the launcher
    (vof2d/launch.py: N fresh workers, this process never touches the GPU); under a launcher -- and
    with one GPU -- it runs the rank itself."""
    from vof2d import cli, launch
    calls = []
    monkeypatch.setattr(launch, "spawn_ranks", lambda script, argv, n, **kw: calls.append(("spawn", script, list(argv), n)) or 0)
    monkeypatch.setattr(cli, "run", lambda args, **kw: calls.append(("run", args.gpus)) or 0)
    for k in ("WORLD_SIZE", "RANK"):
        monkeypatch.delenv(k, raising=False)
    assert cli.main("/x/2dvof.py", ["-ic", "2", "--gpus", "4", "--steps", "10"]) == 0
    assert calls == [("spawn", "/x/2dvof.py", ["-ic", "2", "--gpus", "4", "--steps", "10"], 4)]
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "2")
    cli.main("/x/2dvof.py", ["--gpus", "4"])
    cli.main("/x/2dvof.py", [])
    assert calls[1:] == [("run", 4), ("run", 1)]
    with pytest.raises(SystemExit):
        cli.main("/x/2dvof.py", ["--gpus", "0"])
