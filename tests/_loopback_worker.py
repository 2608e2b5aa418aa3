"""Worker of tests/test_host_gpu.py::test_native_rccl_exchange_* -- run as a FRESH process that
never imports torch, so that the library binds the system RCCL (ROCm 7.2: 2.27.7, whose send/recv
groups can be captured into the step graph) and not the 2.26.6 PyTorch bundles, and the captured
exchange -- what `bench.py --gpus N` runs by default -- is what is asserted, strictly.

    python tests/_loopback_worker.py modes | mode4 LO HI | mode5 LO HI [f64|f32 [JACOBI_ITERS]] | fuzz SEED NCASES
"""
import contextlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "taichi-2d-vof_amd"), os.path.join(ROOT, "tests")]


@contextlib.contextmanager
def raises(exc):
    try:
        yield
    except exc:
        return
    raise AssertionError("%s not raised" % exc.__name__)


def native_rccl_exchange_loopback(hip_api):
    """vof_comm_init / vof_comm_exchange / vof_step_exchange on one GPU with both neighbours looped
    back to the calling rank: each halo must receive the W owned rows next to it (RCCL pairs the k-th
    send to a peer with the k-th receive from it) -- checks row ranges, byte counts, the F buffer
    swap and the stream ordering of the in-library exchange."""
    from vof2d import _abi
    from vof2d.engine import Engine, make_desc, comm_unique_id, VofError
    nx, ny, W = 160, 96, _abi.halo_rows(10)
    own = (41, 120)
    rows = (own[0] - W, own[1] + W)
    e = Engine(hip_api, make_desc(hip_api, nx, ny, "f64", "f32", rows=rows, own=own, device=0))
    e.set_init_F(1)
    with raises(VofError):
        e.step_exchange(1)                      # no communicator yet
    uid = comm_unique_id(hip_api)
    assert len(uid) == _abi.VOF_COMM_ID_BYTES
    with raises(VofError):
        e.comm_init(uid, 0, 1)                  # an interior strip cannot be rank 0 of 1
    e.comm_init(uid, 0, 1, loopback=True)
    rng = np.random.default_rng(5)
    for f in ("F", "u", "v", "p"):
        e.set(f, rng.random((rows[1] - rows[0] + 1, ny + 2)), rows)
    before = {f: e.get(f, rows) for f in ("F", "u", "v", "p")}
    e.comm_exchange(_abi.VOF_XCHG_F | _abi.VOF_XCHG_P)
    e.sync()
    lo, hi = own[0] - rows[0], own[1] - rows[0]    # array indices of own_lo / own_hi
    for f in ("F", "u", "v", "p"):
        got, was = e.get(f, rows), before[f]
        if f in ("F", "p"):
            assert np.array_equal(got[lo - W:lo], was[lo:lo + W]), f
            assert np.array_equal(got[hi + 1:hi + 1 + W], was[hi - W + 1:hi + 1]), f
            assert np.array_equal(got[lo:hi + 1], was[lo:hi + 1]), f
        else:
            assert np.array_equal(got, was), f
    # the stepping loop: same result as the phases with a loopback copy after each, done by hand
    ref = Engine(hip_api, make_desc(hip_api, nx, ny, "f64", "f32", rows=rows, own=own, device=0))
    for f in ("F", "u", "v", "p"):
        ref.set(f, e.get(f, rows), rows)
    ref.istep = e.istep

    def loop(fields):
        for f in fields:
            a = ref.get(f, rows)
            a[lo - W:lo] = a[lo:lo + W]
            a[hi + 1:hi + 1 + W] = a[hi - W + 1:hi + 1]
            ref.set(f, a, rows)

    # non-overlapped: deterministic, equal to the hand-made copies on every stored row.  The first
    # step of a communicator is launched eagerly, later ones replay one captured graph per parity.
    e.step_exchange(5, 0)
    for _ in range(5):
        for ph in (0, 1, 2):
            ref.step_phase(ph)
        loop(("F", "u", "v", "p"))
    for f in ("F", "u", "v", "p"):
        assert np.array_equal(e.get(f, rows), ref.get(f, rows), equal_nan=True), f
    # overlapped: with a looped-back neighbour the halos change *value* under the running kernels
    # (between real neighbours they are rewritten with identical values), so rows near the edges
    # depend on timing here.  Deterministic and checked: every halo ends up holding the final
    # owned rows next to it, and rows deeper than one step's dependency cone equal the reference.
    for mode in (1, 3, 3, 1, 1, 3, 3, 1, 4, 4, 4, 1, 4, 4, 3, 4, 3, 4):   # 4: fused transport, one F / twin swap per step
        for f in ("F", "u", "v", "p"):
            ref.set(f, e.get(f, rows), rows)
        ref.istep = e.istep
        e.step_exchange(1, mode)
        if mode == 4:   # all four fields together once the edge bands of the fused transport exist
            ref.step_phase(0); ref.step_phase(1); ref.step_phase(2); loop(("p", "u", "v", "F"))
        else:
            ref.step_phase(0); loop(("p",)); ref.step_phase(1); loop(("u", "v")); ref.step_phase(2); loop(("F",))
        for f in ("F", "u", "v", "p"):
            got = e.get(f, rows)
            assert np.array_equal(got[lo - W:lo], got[lo:lo + W], equal_nan=True), (f, mode)
            assert np.array_equal(got[hi + 1:hi + 1 + W], got[hi - W + 1:hi + 1], equal_nan=True), (f, mode)
            assert np.array_equal(got[lo + W:hi + 1 - W], ref.get(f, rows)[lo + W:hi + 1 - W], equal_nan=True), (f, mode)
    # one captured graph per (parity, mode): this process has no torch in it, so the library binds the
    # system RCCL (ROCm 7.2: 2.27.7), whose send/recv groups can be captured -- required, not optional
    version, graphs = e.comm_info()
    assert version >= 22707, "RCCL %d cannot capture the exchange" % version
    assert graphs == 1 and e.get_counter("exchange_graph_steps") >= 18      # (a communicator's first step is eager)
    e.comm_destroy()
    e.close(); ref.close()


def _resync(e, ref, rows):
    """The first step of a communicator runs as overlap mode 1, whose rows near the edges depend on timing when the neighbour
    is the strip itself (native_rccl_exchange_loopback): on grids where its kernels last long enough to meet their own
    messages the comparison starts from the state that step left."""
    for f in ("F", "u", "v", "p"):
        ref.set(f, e.get(f, rows), rows)
    ref.set_BC()
    ref.istep = e.istep


def exchange_mode4_equals_phases_plus_copies(hip_api, own, nx=160, ny=96, dtype="f64", iters=10, ic=3, calls=(1, 1, 3, 2), resync_first=False):
    """vof_step_exchange overlap 4 (fused transport on the edge bands, one send/recv group, fused
    transport on the other rows) on an interior strip, on the strips next to the left / right wall (one
    band only) and on a strip so thin that its bands meet, neighbours looped back.  Nothing the second transport launch reads is being
    received meanwhile, so -- unlike modes 1-3 on a loopback -- the result is deterministic and must
    equal the phased step followed by hand-made halo copies on every stored row, ghost cells included."""
    from vof2d import _abi
    from vof2d.engine import Engine, make_desc, comm_unique_id
    W = _abi.halo_rows(iters)
    rows = (max(0, own[0] - W), min(nx + 1, own[1] + W))
    wall_lo, wall_hi = own[0] == 1, own[1] == nx
    e = Engine(hip_api, make_desc(hip_api, nx, ny, dtype, "f32", rows=rows, own=own, device=0, jacobi_iters=iters))
    ref = Engine(hip_api, make_desc(hip_api, nx, ny, dtype, "f32", rows=rows, own=own, device=0, jacobi_iters=iters))
    for x in (e, ref):
        x.set_init_F(ic)
    e.comm_init(comm_unique_id(hip_api), 0, 1, loopback=True)
    lo, hi = own[0] - rows[0], own[1] - rows[0]

    def loop(fields):
        for f in fields:
            a = ref.get(f, rows)
            if not wall_lo:
                a[lo - W:lo] = a[lo:lo + W]
            if not wall_hi:
                a[hi + 1:hi + 1 + W] = a[hi - W + 1:hi + 1]
            ref.set(f, a, rows)

    for n in calls:   # the first step of a communicator is eager (and runs as mode 1), later ones are captured
        if e.istep == 0:
            e.step_exchange(1, 0); ref_modes = 1
        else:
            e.step_exchange(n, 4); ref_modes = n
        for _ in range(ref_modes):
            for ph in (0, 1, 2):
                ref.step_phase(ph)
            loop(("p", "u", "v", "F"))
        # the captured path is REQUIRED here (fresh torch-free process, system RCCL): mode 4 is
        # deterministic on a loopback and must equal the phased step + copies on every stored row
        assert e.comm_info()[1] == 1, "exchange graph capture unavailable (RCCL %d)" % e.comm_info()[0]
        if resync_first and e.istep == 1:
            _resync(e, ref, rows)
            continue
        if resync_first and not all(bool(np.isfinite(ref.get(f, rows)).all()) for f in ("F", "u", "v", "p")):
            # (a run that blew up -- odd sweep counts do within tens of steps: the exact shortcuts of the kernels are exact for
            #  finite values, 0 * inf is not the 0 a bypass writes; nothing after this point says anything)
            e.comm_destroy(); e.close(); ref.close()
            return "blew up"
        for f in ("F", "u", "v", "p"):
            got, want = e.get(f, rows), ref.get(f, rows)
            assert np.array_equal(got, want, equal_nan=True), (own, f, int(e.istep), np.argwhere(got != want)[:4])
    assert e.get_counter("exchange_graph_steps") >= sum(calls[1:]) - 1      # every step but the communicator's first (and one after dirty ghosts)
    e.comm_destroy(); e.close(); ref.close()


def exchange_mode5_equals_pieces_plus_copies(hip_api, own, dtype="f64", iters=10, nx=200, ny=96, ic=3, calls=(1, 4, 7, 2, 1, 12), knobs=None, resync_first=False):
    """vof_step_exchange overlap 5 (the strips run k_jacobi_pair and k_tm; F, u*, v*, rhs, p exchanged once per step, the
    edge bands of k_tm on the communication stream in front of the send / recv group, the other rows beside them) with the
    neighbours looped back: the middle steps replayed from captured graphs, two per launch, must equal the same kernels
    piece by piece (vof_step_tm_piece) with hand-made halo copies, on every stored row, ghost cells included."""
    from vof2d import _abi
    from vof2d.engine import Engine, make_desc, comm_unique_id
    # (iters = 20, 30: every middle step runs iters / 10 launches of k_jacobi_pair -- ADVICE r05; fp32: the same kernels, mode 5 in
    # both precisions since round 6)
    W = _abi.halo_rows(iters)
    rows = (max(0, own[0] - W), min(nx + 1, own[1] + W))
    wall_lo, wall_hi = own[0] == 1, own[1] == nx
    e = Engine(hip_api, make_desc(hip_api, nx, ny, dtype, "f32", rows=rows, own=own, device=0, jacobi_iters=iters))
    ref = Engine(hip_api, make_desc(hip_api, nx, ny, dtype, "f32", rows=rows, own=own, device=0, jacobi_iters=iters))
    for x in (e, ref):
        x.set_init_F(ic)
        for k, v in (knobs or {}).items():
            x.set_param(k, v)
    e.comm_init(comm_unique_id(hip_api), 0, 1, loopback=True)
    lo, hi = own[0] - rows[0], own[1] - rows[0]

    def loop(fields, D=W):
        for f in fields:
            a = ref.get(f, rows)
            if not wall_lo:
                a[lo - D:lo] = a[lo:lo + D]
            if not wall_hi:
                a[hi + 1:hi + 1 + D] = a[hi - D + 1:hi + 1]
            ref.set(f, a, rows)
        ref.set_BC()      # (set marked the ghost cells of F, u, v unknown; they are the copied rows' own: settle them)

    middle = 0
    for n in calls:   # the first step of a communicator is eager and runs as mode 1
        first = e.istep == 0
        e.step_exchange(n, 5)
        if first:
            for ph in (0, 1, 2):
                ref.step_phase(ph)
            loop(("p", "u", "v", "F"))
            n -= 1
        if n > 0:
            middle += (n - 1) // 2 * 2
            ref.step_tm_piece(0); loop(("u_star", "v_star", "rhs"))
            for _ in range(n - 1):
                # (p and rhs travel W deep -- the ten sweeps --, F, u*, v* as deep as k_tm reads them: a looped-back neighbour is a
                # translate of the strip by the depth of the message)
                ref.step_tm_piece(1); loop(("rhs", "p")); loop(("F", "u_star", "v_star"), 8)
            ref.step_tm_piece(2); loop(("p", "u", "v", "F"))
        assert e.istep == ref.istep
        assert e.comm_info()[1] == 1, "exchange graph capture unavailable (RCCL %d)" % e.comm_info()[0]
        if e.istep == 1:
            # The communicator's first step ran as overlap mode 1: p travels under the first sweep, u and v under the second, and
            # a looped-back neighbour's rows are a translate of the strip's own -- the halo rows change VALUE under the kernels that
            # read them (between real neighbours they are rewritten with identical values), so the rows within one step's reach of
            # an interior edge depend on timing (native_rccl_exchange_loopback; seen once in some twenty runs of this test).  What
            # does not: every halo ends up holding the final owned rows next to it, the rows deeper in equal the reference.  The
            # comparison proper starts from the state this step left.
            for f in ("F", "u", "v", "p"):
                got = e.get(f, rows)
                if not wall_lo:
                    assert np.array_equal(got[lo - W:lo], got[lo:lo + W], equal_nan=True), (own, f, "lower halo after the first step")
                if not wall_hi:
                    assert np.array_equal(got[hi + 1:hi + 1 + W], got[hi - W + 1:hi + 1], equal_nan=True), (own, f, "upper halo after the first step")
                a, b = (lo if wall_lo else lo + W), (hi + 1 if wall_hi else hi + 1 - W)
                if b > a:
                    assert np.array_equal(got[a:b], ref.get(f, rows)[a:b], equal_nan=True), (own, f, "rows away from the edges after the first step")
            _resync(e, ref, rows)
            continue
        if resync_first and not all(bool(np.isfinite(ref.get(f, rows)).all()) for f in ("F", "u", "v", "p")):
            # (a run that blew up -- odd sweep counts do within tens of steps: the exact shortcuts of the kernels are exact for
            #  finite values, 0 * inf is not the 0 a bypass writes; nothing after this point says anything)
            e.comm_destroy(); e.close(); ref.close()
            return "blew up"
        for f in ("F", "u", "v", "p"):
            got, want = e.get(f, rows), ref.get(f, rows)
            assert np.array_equal(got, want, equal_nan=True), (own, f, int(e.istep), np.argwhere(got != want)[:4], "rows", np.unique(np.argwhere(got != want)[:, 0]) + rows[0])
    assert e.get_counter("exchange_graph_steps") >= middle      # the middle steps, two per launch
    e.comm_destroy(); e.close(); ref.close()


def fuzz(hip_api, seed, ncases):
    """Random strips (interior, at either wall, so thin that the edge bands meet), grids, precisions, sweep counts, chunk knobs and
    call lengths through the two deterministic exchange modes: captured send / recv graphs against pieces + copies, every stored row."""
    from vof2d import _abi
    ran, blew = {4: 0, 5: 0}, 0
    for k in range(ncases):
        rng = np.random.default_rng(seed + k)
        mode = int(rng.choice([4, 5, 5]))
        iters = int(rng.choice([10, 10, 20, 30] if mode == 4 else [10, 10, 10, 20, 5, 15, 30]))
        W = _abi.halo_rows(iters)
        dtype = "f64" if rng.random() < 0.6 else "f32"
        nx = int(rng.integers(4 * W + 2, 4 * W + 400))
        ny = int(rng.choice([rng.integers(16, 140), rng.integers(100, 700)]))
        if rng.random() < 0.4:
            ny = nx
        kind = rng.choice(["interior", "lo", "hi", "thin"])
        if kind == "lo":
            own = (1, int(rng.integers(W, nx - W)))
        elif kind == "hi":
            own = (int(rng.integers(W + 1, nx - W + 2)), nx)
        else:
            n = int(rng.integers(W, 2 * W)) if kind == "thin" else int(rng.integers(W, nx - 2 * W))
            lo = int(rng.integers(W + 1, nx - W - n + 2))
            own = (lo, lo + n - 1)
        calls = [1] + [int(c) for c in rng.integers(1, 14, size=int(rng.integers(2, 6)))]
        knobs = {}
        if mode == 5:
            for name, vals in (("tm_rows", (0, 3, 6, 16, 37)), ("jacobi_pair_rows", (0, 5, 12, 27, 80)), ("jacobi_pair", (0, 1, 2)),
                               ("buffer_stores", (0, 3, 7)), ("jacobi_tb_rows", (0, 7, 16))):
                if rng.random() < 0.3:
                    knobs[name] = int(rng.choice(vals))
        what = "seed %d: mode %d %dx%d %s iters %d own %r calls %r knobs %r" % (seed + k, mode, nx, ny, dtype, iters, own, calls, knobs)
        try:
            if mode == 4:
                how = exchange_mode4_equals_phases_plus_copies(hip_api, own, nx, ny, dtype, iters, int(rng.integers(1, 4)), calls, True)
            else:
                how = exchange_mode5_equals_pieces_plus_copies(hip_api, own, dtype, iters, nx, ny, int(rng.integers(1, 4)), calls, knobs, True)
        except AssertionError as err:
            print("DIVERGES " + what + "\n    -> %r" % (err.args,), flush=True)
            raise
        ran[mode] += 1
        blew += how == "blew up"
    print("loopback fuzz: %d cases from seed %d (mode 4: %d, mode 5: %d; %d blew up on the way and were compared up to there), all equal" % (
        ncases, seed, ran[4], ran[5], blew))


if __name__ == "__main__":
    from vof2d._lib import hip_api as load
    api = load()
    if sys.argv[1] == "modes":
        native_rccl_exchange_loopback(api)
    elif sys.argv[1] == "fuzz":
        fuzz(api, int(sys.argv[2]), int(sys.argv[3]))
    elif sys.argv[1] == "mode5":
        exchange_mode5_equals_pieces_plus_copies(api, (int(sys.argv[2]), int(sys.argv[3])), *(sys.argv[4:5] or ["f64"]), *(int(x) for x in sys.argv[5:6]))
    else:
        exchange_mode4_equals_phases_plus_copies(api, (int(sys.argv[2]), int(sys.argv[3])))
    assert "torch" not in sys.modules, "the worker must stay torch-free"
    print("OK")
