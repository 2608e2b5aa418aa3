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
import base64
import io
import json
import os
import stat
import time

import numpy as np


def _npy_bytes(a):
    buf = io.BytesIO()
    np.save(buf, np.asarray(a), allow_pickle=False)
    return buf.getvalue()


def _dumps(obj):
    """Rendezvous payloads are data, never code: JSON for plain values, the .npy format (no pickled
    objects) for arrays -- bare, or base64-wrapped inside a JSON structure.  Nothing read from the
    rendezvous directory is ever unpickled."""
    if isinstance(obj, np.ndarray):
        return b"NPY0" + _npy_bytes(obj)

    def enc(o):
        if isinstance(o, np.ndarray):
            return {"__npy__": base64.b64encode(_npy_bytes(o)).decode("ascii")}
        if isinstance(o, np.generic):
            return o.item()
        raise TypeError("rendezvous payloads are JSON values and arrays, not %r" % type(o).__name__)
    return b"JSN0" + json.dumps(obj, default=enc).encode()


def _loads(data):
    tag, body = data[:4], data[4:]
    if tag == b"NPY0":
        return np.load(io.BytesIO(body), allow_pickle=False)
    if tag == b"JSN0":
        def dec(d):
            if set(d) == {"__npy__"}:
                return np.load(io.BytesIO(base64.b64decode(d["__npy__"])), allow_pickle=False)
            return d
        return json.loads(body.decode(), object_hook=dec)
    raise ValueError("rendezvous: unknown payload tag %r" % (tag,))


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

    def broadcast_object(self, obj, engine=None):
        box = [obj if self.rank == 0 else None]
        self.dist.broadcast_object_list(box, src=0)
        return box[0]


class EnvComm:
    """torch-free carrier for one node (see module docstring).

    The rendezvous directory is private to the launch: `bench.py --gpus N` started without a
    launcher creates it with tempfile.mkdtemp (mode 0700, unpredictable name) and hands it to its
    workers in VOF2D_RDZV_DIR.  Under an external launcher the name is derived from the launcher's
    port / run id / restart count / pid, and the directory is refused unless it is a real directory
    owned by this user with no access for anybody else."""

    def __init__(self, rank=None, world=None, local_rank=None, rdzv_dir=None, timeout=300.0):
        env = os.environ
        self.rank = int(env.get("RANK", 0)) if rank is None else rank
        self.world = int(env.get("WORLD_SIZE", 1)) if world is None else world
        self.local_rank = int(env.get("LOCAL_RANK", self.rank)) if local_rank is None else local_rank
        # one directory per launch AND per restart of it: a restarted torchrun keeps its pid, port and
        # run id, but counts its restarts (stale files of the crashed attempt stay behind, unread)
        tag = "%s_%s_%s_%s" % (env.get("MASTER_PORT", "0"), env.get("TORCHELASTIC_RUN_ID", "none"),
                               env.get("TORCHELASTIC_RESTART_COUNT", "0"), env.get("VOF2D_RDZV_TAG") or os.getppid())
        if rdzv_dir is not None:
            self.dir = rdzv_dir                  # the caller's directory: only IT has to be private
        else:
            # VOF2D_RDZV_DIR names a PRIVATE parent directory (created 0700 if missing; refused if it
            # is shared, e.g. /tmp itself -- see INTEGRATION.md); default /tmp/vof2d-<uid>
            base = env.get("VOF2D_RDZV_DIR") or os.path.join("/tmp", "vof2d-%d" % os.getuid())
            self._private_dir(base)
            self.dir = os.path.join(base, "rdzv_" + tag)
        self._private_dir(self.dir)
        self.timeout = timeout
        self._seq = 0

    @staticmethod
    def _private_dir(path):
        try:
            os.mkdir(path, 0o700)
        except FileExistsError:
            pass
        st = os.lstat(path)                     # lstat: a symlink planted by somebody else is refused
        if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
            raise PermissionError("rendezvous directory %s is not a private directory of uid %d "
                                  "(mode %o, owner %d)" % (path, os.getuid(), st.st_mode & 0o7777, st.st_uid))

    def _path(self, name, rank=None):
        return os.path.join(self.dir, name if rank is None else "%s.%d" % (name, rank))

    def _put(self, path, data):
        tmp = "%s.tmp%d" % (path, os.getpid())
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        with os.fdopen(fd, "wb") as f:
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
        """Raw bytes from rank 0 to everybody (e.g. the 128-byte RCCL unique id)."""
        self._seq += 1
        path = self._path("bcast%d" % self._seq)
        if self.rank == 0:
            self._put(path, bytes(data))
            return data
        return self._get(path)

    def broadcast_object(self, obj, engine=None):
        """A JSON-able value (or an array) from rank 0 to everybody."""
        return _loads(self.broadcast_bytes(_dumps(obj) if self.rank == 0 else None))

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
        self._put(self._path(name, self.rank), _dumps(obj))
        if self.rank != 0:
            return None
        return [_loads(self._get(self._path(name, r))) for r in range(self.world)]

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
