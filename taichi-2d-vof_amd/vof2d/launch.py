"""One process per GPU on one node, started without an external launcher.

`bench.py --gpus N` and `2dvof.py --gpus N` call `spawn_ranks` when no WORLD_SIZE is in the
environment: the calling process becomes the launcher.  It must not have touched the GPU (no HIP
call, no torch import) and it never does: it starts N fresh worker processes -- RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in their environment exactly as `python -m torch.distributed.run` would set
them, plus a private rendezvous directory for `comms.EnvComm` -- relays rank 0's standard output
and returns the first non-zero worker exit code.  A worker that dies takes the others down (they
would otherwise wait for it until their own watchdogs fire); the workers die with the launcher.
"""
import ctypes
import os
import selectors
import shutil
import signal
import socket
import subprocess
import sys
import tempfile
import time


def die_with_parent():
    """preexec_fn: PR_SET_PDEATHSIG = SIGKILL, so a child never outlives the process that started
    it (a launcher that is torn down would otherwise leave GPU-holding orphans)."""
    ctypes.CDLL(None).prctl(1, int(signal.SIGKILL), 0, 0, 0)


def under_launcher():
    """True if a launcher (torch.distributed.run, or spawn_ranks) already gave this process a rank."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def spawn_ranks(script, argv, nranks, extra_env=None, relay=sys.stdout):
    """Run `python script argv...` once per rank; returns the exit code (0 if every rank returned 0)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    rdzv = tempfile.mkdtemp(prefix="vof2d_rdzv_")        # mode 0700, unpredictable name
    procs = []
    for r in range(nranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks), LOCAL_WORLD_SIZE=str(nranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VOF2D_RDZV_DIR=rdzv,
                   VOF2D_RDZV_TAG="self_%d" % os.getpid(),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      start_new_session=True, preexec_fn=die_with_parent))

    def kill_all(*_):
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)      # exactly the process groups started above
                except ProcessLookupError:
                    pass

    old = {sig: signal.signal(sig, lambda s_, f: (kill_all(), sys.exit(128 + s_))) for sig in (signal.SIGTERM, signal.SIGINT)}
    rc = 0
    try:
        pending = set(range(nranks))
        sel = selectors.DefaultSelector()
        sel.register(procs[0].stdout, selectors.EVENT_READ)
        eof = False
        while pending:
            if not eof:
                for key, _ in sel.select(timeout=0.2):
                    chunk = os.read(key.fileobj.fileno(), 65536)
                    if chunk:
                        relay.write(chunk.decode(errors="replace"))
                        relay.flush()
                    else:
                        eof = True
                        sel.unregister(key.fileobj)
            else:
                time.sleep(0.2)
            for r in list(pending):
                code = procs[r].poll()
                if code is not None:
                    pending.discard(r)
                    if code != 0 and rc == 0:
                        rc = code
                        print("[launch] worker %d exited with code %d: stopping the others" % (r, code), file=sys.stderr)
                        kill_all()
        if not eof:
            rest = procs[0].stdout.read() or b""
            relay.write(rest.decode(errors="replace"))
            relay.flush()
    finally:
        kill_all()
        shutil.rmtree(rdzv, ignore_errors=True)
        for sig, h in old.items():
            signal.signal(sig, h)
    return rc
