"""N > 1 path on CPU: two gloo processes, each a StripSolver (row strip + deep-halo exchange via
torch.distributed batch_isend_irecv).  The compute engine is the CPU oracle behind the same C ABI;
what is under test is the host logic the GPU path shares: partitioning, halo geometry, the
zero-copy row views, the exchange schedule and the residual all-reduce."""
import os
import socket

import numpy as np
import pytest

from util import engine, same, diff_report
from vof2d.strips import partition, stored_rows, balanced_partition
from vof2d import halo_rows


def test_partition_covers_the_grid():
    for nx, world in [(8192, 8), (4096, 3), (100, 4), (64, 2), (17, 1)]:
        parts = partition(nx, world)
        assert parts[0][0] == 1 and parts[-1][1] == nx
        assert all(parts[k][1] + 1 == parts[k + 1][0] for k in range(world - 1))
        sizes = [hi - lo + 1 for lo, hi in parts]
        assert max(sizes) - min(sizes) <= 1
    assert stored_rows(64, (1, 32), 16) == (0, 48)
    assert stored_rows(64, (33, 64), 16) == (17, 65)
    assert halo_rows(10) == 18


def test_balanced_partition_equalises_the_measured_cost():
    nx, world = 8192, 8
    eq = partition(nx, world)
    # ranks 0-2 hold liquid / the interface and are 10 % / 5 % slower than the gas strips
    costs = [391, 391, 373, 357, 357, 357, 357, 357]
    parts = balanced_partition(nx, eq, costs, min_rows=16)
    assert parts[0][0] == 1 and parts[-1][1] == nx
    assert all(parts[k][1] + 1 == parts[k + 1][0] for k in range(world - 1))
    dens = np.concatenate([np.full(hi - lo + 1, c / (hi - lo + 1)) for (lo, hi), c in zip(eq, costs)])
    new_costs = [dens[lo - 1:hi].sum() for lo, hi in parts]
    assert max(new_costs) < 1.002 * sum(costs) / world < 0.96 * max(costs)
    assert parts[0][1] - parts[0][0] + 1 < 1024 < parts[-1][1] - parts[-1][0] + 1
    # equal costs leave the partition alone; a floor on the strip height is kept
    assert balanced_partition(nx, eq, [1.0] * world) == eq
    thin = balanced_partition(64, partition(64, 4), [100, 1, 1, 1], min_rows=16)
    assert [hi - lo + 1 for lo, hi in thin] == [16, 16, 16, 16]
    assert balanced_partition(100, [(1, 100)], [3.0]) == [(1, 100)]


def test_strips_with_an_uneven_partition_equal_single_domain(oracle_api, tmp_path):
    """Strips of different heights (what bench.py's cost balancing produces) give the same fields."""
    import torch.multiprocessing as mp
    import _strip_worker
    nx, ny, steps, world = 100, 24, 10, 3
    parts = [(1, 22), (23, 71), (72, 100)]
    mp.spawn(_strip_worker.run, args=(world, _free_port(), nx, ny, 1, "f64", steps, str(tmp_path), True, parts),
             nprocs=world, join=True)
    z = np.load(tmp_path / "strips.npz")
    ref = engine(oracle_api, nx, ny, "f64", "f32", ic=1)
    ref.step(steps)
    for f in ("F", "u", "v", "p"):
        assert same(z[f], ref.get(f)), diff_report(z[f], ref.get(f), f)


def test_strips_with_host_staged_halos_equal_single_domain(oracle_api, tmp_path):
    """StripSolver(stage_host=True): the halo rows go through host buffers around the P2P ops (what the one-GPU rehearsal of the
    N > 1 path, bench.py --gpus 2 --same-device, runs over gloo with the HIP engine) -- three ranks, uneven strips, phased
    exchanges, then the residual solve's p exchanges."""
    import torch.multiprocessing as mp
    import _strip_worker
    nx, ny, steps, world = 90, 28, 9, 3
    parts = [(1, 30), (31, 52), (53, 90)]
    mp.spawn(_strip_worker.run, args=(world, _free_port(), nx, ny, 3, "f64", steps, str(tmp_path), True, parts, True),
             nprocs=world, join=True)
    z = np.load(tmp_path / "strips.npz")
    ref = engine(oracle_api, nx, ny, "f64", "f32", ic=3)
    ref.step(steps)
    for f in ("F", "u", "v", "p"):
        assert same(z[f], ref.get(f)), diff_report(z[f], ref.get(f), f)
    it, res = ref.solve_p_residual(1e-9, 40, 10)
    assert int(z["it"]) == it and float(z["res"]) == res and same(z["p_after"][1:-1], ref.get("p")[1:-1])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("nx,ny,ic,dtype,world,overlap", [
    (72, 40, 1, "f64", 2, True), (72, 40, 1, "f64", 2, False), (66, 24, 2, "f32", 2, True),
    (96, 20, 3, "f64", 3, True), (128, 33, 1, "f64", 4, True),
    (72, 40, 1, "f64", 2, 5), (96, 20, 3, "f64", 3, 5), (66, 24, 2, "f32", 2, 5)])
def test_strips_equal_single_domain(oracle_api, tmp_path, nx, ny, ic, dtype, world, overlap):
    """overlap=True: the phased step with each field's halo sent as soon as it is final, the
    transfers running concurrently (gloo threads here, RCCL's stream on the GPU) with the rest of
    the step; overlap=False: one exchange after the whole step; overlap=5: the exchange state of the pair kernels -- the
    step boundary behind the next step's predictor, F, p and u*, v* across the strip edges once per step, u and v only at
    the end of a call.  All must equal the single domain."""
    import torch.multiprocessing as mp
    import _strip_worker
    steps = 12
    mp.spawn(_strip_worker.run, args=(world, _free_port(), nx, ny, ic, dtype, steps, str(tmp_path), overlap),
             nprocs=world, join=True)
    z = np.load(tmp_path / "strips.npz")
    ref = engine(oracle_api, nx, ny, dtype, "f32", ic=ic)
    ref.step(steps)
    for f in ("F", "u", "v", "p"):
        assert same(z[f], ref.get(f)), diff_report(z[f], ref.get(f), f)
    # residual-terminated solve (extension): same sweeps, same global max-norm as the single domain
    it, res = ref.solve_p_residual(1e-9, 40, 10)
    assert int(z["it"]) == it and float(z["res"]) == res
    assert same(z["p_after"][1:-1], ref.get("p")[1:-1])
    it2, res2 = ref.solve_p(1e-3, 100, 25, "rel")      # vof_solve_p, relative criterion
    assert int(z["it2"]) == it2 and float(z["res2"]) == res2 and it2 % 25 == 0
    assert same(z["p_after2"][1:-1], ref.get("p")[1:-1])


def _envcomm_worker(rank, world, rdzv, out):
    from vof2d.comms import EnvComm
    c = EnvComm(rank, world, rank, rdzv_dir=rdzv, timeout=60)
    uid = c.broadcast_bytes(b"\x01" * 128 if rank == 0 else None)
    second = c.broadcast_bytes(b"two" if rank == 0 else None)
    parts = c.gather_object({"rank": rank, "rows": np.full((2, 3), rank)})
    with open(os.path.join(out, "r%d" % rank), "w") as f:
        f.write("%d %s %s" % (len(uid), second.decode(), "none" if parts is None else
                              ",".join(str(int(p["rows"].sum())) for p in parts)))


def test_envcomm_file_rendezvous(tmp_path):
    """The torch-free carrier of bench.py's N > 1 path: bytes from rank 0 reach every rank through
    the rendezvous directory, objects gather on rank 0 (3 processes, no torch.distributed)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    rdzv, out = str(tmp_path / "rdzv"), str(tmp_path)
    procs = [ctx.Process(target=_envcomm_worker, args=(r, 3, rdzv, out)) for r in (2, 1, 0)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert open(os.path.join(out, "r0")).read() == "128 two 0,6,12"
    assert open(os.path.join(out, "r1")).read() == "128 two none"
    assert open(os.path.join(out, "r2")).read() == "128 two none"
