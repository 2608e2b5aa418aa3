"""Worker of tests/test_cli_cpu.py: one rank of `2dvof.py --gpus N` with the CPU oracle standing in
for the HIP library (same C ABI; test double) and torch.distributed/gloo as the carrier.  What is
under test is the host program vof2d/cli.py: strips, gathering F for the PNG, checkpoints, resume."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "taichi-2d-vof_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def oracle_api():
    from vof2d import _abi
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libvof_oracle.so"))
    return _abi.bind(lib, "ovof_", optional=_abi.GPU_ONLY)


def run(rank, world, port, argv, cwd):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      OMP_NUM_THREADS="1", MPLBACKEND="Agg")
    os.chdir(cwd)
    from vof2d import cli
    args = cli.build_parser().parse_args(argv)
    lines = []
    if world == 1:
        cli.run(args, api=oracle_api(), world=1, rank=0, out=lambda *a: lines.append(" ".join(str(x) for x in a)))
    else:
        import torch.distributed as dist
        from vof2d.comms import TorchComm
        dist.init_process_group("gloo", rank=rank, world_size=world)
        try:
            cli.run(args, api=oracle_api(), comm=TorchComm(dist, rank, world), rank=rank, world=world,
                    out=lambda *a: lines.append(" ".join(str(x) for x in a)))
            dist.barrier()
        finally:
            dist.destroy_process_group()
    with open(os.path.join(cwd, "stdout.%d" % rank), "w") as f:
        f.write("\n".join(lines))
