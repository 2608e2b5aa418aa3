"""Worker of the world_size-2 gloo test: each process runs StripSolver over the CPU oracle
(test double for the HIP engine; same ABI) and rank 0 compares with the single-domain run."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "taichi-2d-vof_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def oracle_api():
    from vof2d import _abi
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libvof_oracle.so"))
    return _abi.bind(lib, "ovof_", optional=_abi.GPU_ONLY)


def run(rank, world, port, nx, ny, ic, dtype, steps, outdir, overlap=True, parts=None, stage_host=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      OMP_NUM_THREADS="1")
    import torch.distributed as dist
    from vof2d.strips import StripSolver
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s = StripSolver(nx, ny, dtype, ic=ic, rank=rank, world=world, api=oracle_api(), dist=dist, parts=parts, stage_host=stage_host)
        assert s.stage_host == bool(stage_host)
        s.step(steps, overlap=overlap)
        fields = {f: s.gather(f) for f in ("F", "u", "v", "p")}
        it, res = s.solve_p_residual(1e-9, 40, 10)
        p_after = s.gather("p")
        # relative criterion, a check interval longer than the halo of p covers (25 sweeps = 10 + 10 + 5 with
        # a halo refresh in between), both norms all-reduced
        it2, res2 = s.solve_p(1e-3, 100, 25, "rel")
        p_after2 = s.gather("p")
        if rank == 0:
            np.savez(os.path.join(outdir, "strips.npz"), it=it, res=res, p_after=p_after, it2=it2, res2=res2,
                     p_after2=p_after2, **fields)
        dist.barrier()
    finally:
        dist.destroy_process_group()
