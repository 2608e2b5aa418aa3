"""Process-group plumbing of the strip solver: who am I, how do bytes get from rank 0 to
everybody, how is a scalar maximised over the ranks.

Two carriers with the same four methods (`broadcast_bytes`, `allreduce_max`, `barrier`,
`gather_object`):

* `TorchComm`  wraps torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" on CPU).  The CPU
  tests drive the strips through it, and it is the transport of the torch halo exchange.
* `EnvComm`    needs no torch in the process: rank / world / local rank come from the launcher's
  environment (RANK, WORLD_SIZE, LOCAL_RANK -- what `python -m torch.distributed.run` exports),
  the few bytes that must travel before an RCCL communicator exists (its unique id) go through a
  rendezvous directory on the node, and reductions run on the library's own RCCL communicator
  (vof_comm_allreduce_max).  One node only -- which is what row strips over xGMI are for.
"""
import os
import pickle
import time


class TorchComm:
    def __init__(self, dist, rank, world, device=None):
        self.dist, self.rank, self.world, self.device = dist, rank, world, device

    def broadcast_bytes(self, data, engine=None):
        box = [data if self.rank == 0 else None]
        self.dist.broadcast_object_list(box, src=0)
        return box[0]

    def allreduce_max(self, value, engine=None):
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def barrier(self, engine=None):
        self.dist.barrier()

    def gather_object(self, obj):
        parts = [None] * self.world if self.rank == 0 else None
        self.dist.gather_object(obj, parts, dst=0)
        return parts


class EnvComm:
    """torch-free carrier for one node (see module docstring)."""

    def __init__(self, rank=None, world=None, local_rank=None, rdzv_dir=None, timeout=300.0):
        env = os.environ
        self.rank = int(env.get("RANK", 0)) if rank is None else rank
        self.world = int(env.get("WORLD_SIZE", 1)) if world is None else world
        self.local_rank = int(env.get("LOCAL_RANK", self.rank)) if local_rank is None else local_rank
        # one directory per launch: the launcher's pid (the workers' common parent) and its port
        # (VOF2D_RDZV_TAG: set by a supervising parent process, whose own parent the workers share)
        tag = "%s_%s_%s" % (env.get("MASTER_PORT", "0"), env.get("TORCHELASTIC_RUN_ID", "none"),
                            env.get("VOF2D_RDZV_TAG") or os.getppid())
        self.dir = rdzv_dir or os.path.join(env.get("VOF2D_RDZV_DIR", "/tmp"), "vof2d_rdzv_" + tag)
        os.makedirs(self.dir, exist_ok=True)
        self.timeout = timeout
        self._seq = 0

    def _path(self, name, rank=None):
        return os.path.join(self.dir, name if rank is None else "%s.%d" % (name, rank))

    def _put(self, path, data):
        tmp = "%s.tmp%d" % (path, os.getpid())
        with open(tmp, "wb") as f:
            f.write(data)
        os.replace(tmp, path)   # atomic: readers see the whole file or none

    def _get(self, path):
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > self.timeout:
                raise TimeoutError("rendezvous: %s did not appear within %.0f s" % (path, self.timeout))
            time.sleep(0.002)
        with open(path, "rb") as f:
            return f.read()

    def broadcast_bytes(self, data, engine=None):
        self._seq += 1
        path = self._path("bcast%d" % self._seq)
        if self.rank == 0:
            self._put(path, data)
            return data
        return self._get(path)

    def allreduce_max(self, value, engine=None):
        if self.world == 1:
            return float(value)
        if engine is None:
            raise RuntimeError("EnvComm reduces through the strip's RCCL communicator: pass the engine")
        return engine.comm_allreduce_max(value)

    def barrier(self, engine=None):
        self.allreduce_max(0.0, engine)

    def gather_object(self, obj):
        self._seq += 1
        name = "gather%d" % self._seq
        self._put(self._path(name, self.rank), pickle.dumps(obj))
        if self.rank != 0:
            return None
        return [pickle.loads(self._get(self._path(name, r))) for r in range(self.world)]

    def cleanup(self):
        """Remove this launch's rendezvous files (rank 0, after a barrier)."""
        if self.rank == 0:
            for n in os.listdir(self.dir):
                try:
                    os.remove(os.path.join(self.dir, n))
                except OSError:
                    pass
            try:
                os.rmdir(self.dir)
            except OSError:
                pass
