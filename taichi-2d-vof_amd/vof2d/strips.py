"""Row-strip domain decomposition: one process per GPU, halo exchange over RCCL.

The grid is cut along the slow index i (x) into contiguous strips.
Each rank stores its owned rows plus W = VOF_HALO_ROWS(jacobi_iters) halo rows
per interior side and runs the *whole* fused step on the extended strip; the
invalid fringe that grows inward from the strip edge during a step (one row
per stencil radius) never reaches the owned rows.  One exchange per step then
refreshes the halos of F, u, v, p from the neighbours' owned rows:

    per step and interior edge:  4 fields x W rows x pitch x sizeof(T)  each way
    (8192^2 fp64, W = 16: 4 x 16 x 65.8 KB = 4.2 MB -> ~30 us on one xGMI link)

instead of one 1-row exchange per dependency stage (>= 16 latency-bound
messages per step).  A halo row is `pitch` contiguous elements, so the rows of
one field form one contiguous message: send/recv go straight from/to field
memory (vof_field_view), no packing kernels.  Only point-to-point traffic is
on the data path -- the reference has no reduction -- so results are identical
for any strip count.  The residual-terminated pressure solve (extension) adds
one all-reduce(MAX) of a scalar per check.

torch / torch.distributed are plumbing here (stream handles, P2P ops);
backend "nccl" is RCCL on ROCm, "gloo" drives the CPU tests.
"""
import ctypes

import numpy as np

from . import _abi
from .engine import Engine, make_desc

EXCHANGED = ("F", "u", "v", "p")
EXCHANGED_MODE5 = ("u_star", "v_star")      # next to F and p: the exchange state with the step boundary behind the predictor
EXCHANGED_PIECES = ("rhs",)                 # ... and what the library's pair kernels ship with them (vof_step_tm_piece)


class _StagedWork:
    """The works of one host-staged exchange (see StripSolver.stage_host): wait for the gloo transfers, then copy the
    received rows from their host buffers into field memory."""

    def __init__(self, works, landings):
        self.works, self.landings = works, landings

    def wait(self):
        for w in self.works:
            w.wait()
        for dst, src in self.landings:
            dst.copy_(src)


def partition(nx, world):
    """Owned interior rows (lo, hi) of every rank: contiguous, balanced to within one row."""
    bounds = [(k * nx) // world for k in range(world + 1)]
    return [(bounds[k] + 1, bounds[k + 1]) for k in range(world)]


def balanced_partition(nx, parts, costs, min_rows=1):
    """Re-cut the rows so that every rank gets the same share of the measured cost.

    parts: the partition the measurement was taken with; costs[r]: what rank r's strip cost there
    (e.g. seconds per step of its kernels).  The cost of a row is taken as uniform within each old
    strip -- strips differ because their contents do: rows full of gas take the sweeps' zero
    shortcuts, rows at the interface pay for normals and surface tension -- and the new boundaries
    are where the cumulative cost reaches k / world of the total.  Every strip keeps >= min_rows rows
    (the halo depth).  Results do not depend on the partition (no collective on the data path)."""
    world = len(parts)
    if world == 1:
        return list(parts)
    dens = np.concatenate([np.full(hi - lo + 1, float(c) / (hi - lo + 1)) for (lo, hi), c in zip(parts, costs)])
    assert len(dens) == nx and np.all(dens > 0)
    cum = np.concatenate([[0.0], np.cumsum(dens)])
    cuts = [0]
    for k in range(1, world):
        b = int(np.searchsorted(cum, cum[-1] * k / world))      # first row count whose cumulative cost reaches the share
        b = max(b, cuts[-1] + min_rows)
        b = min(b, nx - (world - k) * min_rows)
        cuts.append(b)
    cuts.append(nx)
    return [(cuts[k] + 1, cuts[k + 1]) for k in range(world)]


def stored_rows(nx, own, halo):
    lo, hi = own
    return max(0, lo - halo), min(nx + 1, hi + halo)


class _DevArray:
    """Minimal __cuda_array_interface__ carrier so torch can alias library-owned device memory."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class StripSolver:
    """One rank's strip + the per-step halo exchange.

    exchange: who moves the halos.  "native" = the library's own RCCL communicator
    (vof_step_exchange: the whole step loop is one C call, one hipGraph launch per step);
    "torch" = torch.distributed P2P on tensors aliasing the field memory (what the CPU / gloo tests
    drive); "auto" = native on the GPU, torch otherwise.
    comm: a `comms.TorchComm` / `comms.EnvComm`; default: TorchComm over `dist` (torch.distributed).
    With an EnvComm and the native exchange the process never imports torch."""

    def __init__(self, nx, ny, dtype="f64", ic=1, coord_cast="f32", jacobi_iters=10, rank=0, world=1,
                 device=None, api=None, dist=None, exchange="auto", comm=None, parts=None, stage_host=None, **consts):
        if exchange not in ("auto", "native", "torch"):
            raise ValueError("exchange must be 'auto', 'native' or 'torch'")
        if api is None:
            from ._lib import hip_api
            api = hip_api()
        self.api = api
        self.on_gpu = api.backend() == b"hip-gfx950"
        self.rank, self.world = rank, world
        self.nx, self.ny = nx, ny
        self.halo = _abi.halo_rows(jacobi_iters)
        # parts: the owned rows of every rank (default: equal strips; see balanced_partition)
        parts = [tuple(int(x) for x in pr) for pr in parts] if parts is not None else partition(nx, world)
        if len(parts) != world or parts[0][0] != 1 or parts[-1][1] != nx or \
                any(parts[k][1] + 1 != parts[k + 1][0] for k in range(world - 1)):
            raise ValueError("parts must be %d contiguous row ranges covering 1..%d" % (world, nx))
        if min(hi - lo + 1 for lo, hi in parts) < self.halo:
            raise ValueError("strips of %d rows are thinner than the %d-row halo" % (min(hi - lo + 1 for lo, hi in parts), self.halo))
        self.parts = parts
        self.own = parts[rank]
        self.rows = stored_rows(nx, self.own, self.halo)
        from .comms import EnvComm, TorchComm
        torch_free = isinstance(comm, EnvComm)
        if torch_free and world > 1 and not (self.on_gpu and exchange in ("auto", "native")):
            raise ValueError("an EnvComm carries only the native RCCL exchange (GPU library)")
        self.torch = self.dist = self.stream = self.device = None
        stream_ptr = None
        if not torch_free:
            import torch
            self.torch = torch
            self.dist = dist if dist is not None else (torch.distributed if world > 1 else None)
            if self.on_gpu:
                dev = torch.device("cuda", device if device is not None else 0)
                torch.cuda.set_device(dev)
                # a dedicated non-default stream: hipGraph capture is illegal on the legacy stream, and
                # torch.distributed orders its RCCL work against the *current* stream
                self.stream = torch.cuda.Stream(device=dev)
                stream_ptr = self.stream.cuda_stream
                self.device = dev
        # stage_host: the torch carrier's P2P ops run on host buffers, the rows copied out of / into field memory around
        # them -- what a backend that cannot move device memory needs (gloo: the N > 1 rehearsal of bench.py --same-device,
        # two processes on ONE GPU, where RCCL refuses to form a communicator).  Default: on the GPU under gloo.
        if stage_host is None:
            stage_host = bool(self.on_gpu and self.dist is not None and world > 1 and self.dist.get_backend() == "gloo")
        self.stage_host = bool(stage_host) and world > 1 and not torch_free      # (asked for explicitly it also runs over the CPU engine: the gloo tests)
        self._fresh = True          # no step since set_init_F: the first one runs with the reference's intermediate set_BC calls
        self.comm = comm if comm is not None else (TorchComm(self.dist, rank, world, None if self.stage_host else self.device) if world > 1 else None)
        desc = make_desc(api, nx, ny, dtype, coord_cast, rows=self.rows, own=self.own,
                         jacobi_iters=jacobi_iters, device=(device if device is not None else 0) if self.on_gpu else -1,
                         **consts)
        self.eng = Engine(api, desc, stream=stream_ptr)   # no stream given: the library creates its own
        self.eng.set_init_F(ic)
        self._views = {}
        self._ops = {}
        self.exchange_kind = "torch"
        if world > 1 and self.on_gpu and exchange in ("auto", "native"):
            self._native_comm_init(strict=(exchange == "native" or torch_free))
        if self.exchange_kind == "torch" and not torch_free:
            self._build_views()

    def _native_comm_init(self, strict):
        """One RCCL communicator per strip handle; its unique id travels over `self.comm`.  With a
        TorchComm every rank ends up in the same mode: failures are agreed on before committing."""
        import sys
        from .comms import TorchComm
        from .engine import comm_unique_id, VofError
        uid, err, ok = None, None, 0
        if self.rank == 0:
            try:
                uid = comm_unique_id(self.api)
            except VofError as e:
                err = e
        uid = self.comm.broadcast_bytes(uid if uid is not None else b"")   # empty: rank 0 has no RCCL
        if uid:
            try:
                self.eng.comm_init(uid, self.rank, self.world)
                ok = 1
            except VofError as e:
                err = e
        if isinstance(self.comm, TorchComm):
            ok_all = -self.comm.allreduce_max(-ok)   # min over ranks
        else:
            ok_all = ok                              # ncclCommInitRank is collective: all or none
        if ok_all >= 1:
            self.exchange_kind = "native"
            return
        if ok:
            self.eng.comm_destroy()
        if strict:
            raise RuntimeError("native RCCL exchange unavailable on some rank: %s" % (err,))
        if self.rank == 0:
            print("[vof2d] native RCCL exchange unavailable (%s); using torch.distributed P2P" % (err,), file=sys.stderr)

    # -- zero-copy tensor views of the exchanged fields -----------------------------
    def _build_views(self):
        torch = self.torch
        tdt = torch.float64 if self.eng.np_dtype == np.float64 else torch.float32
        typestr = "<f8" if self.eng.np_dtype == np.float64 else "<f4"
        for f in EXCHANGED + EXCHANGED_MODE5 + (EXCHANGED_PIECES if self.on_gpu else ()):
            base, pitch, col0, nrows = self.eng.field_view(f)
            if self.on_gpu:
                t = torch.as_tensor(_DevArray(base, (nrows, pitch), typestr), device=self.device)
            else:
                n = nrows * pitch
                buf = (ctypes.c_double if tdt == torch.float64 else ctypes.c_float) * n
                t = torch.from_numpy(np.ctypeslib.as_array(buf.from_address(base))).view(nrows, pitch)
            assert t.dtype == tdt and t.data_ptr() == base
            self._views[f] = (t, base)

    def _rows_view(self, f, g0, g1):
        t, base = self._views[f]
        cur = self.eng.field_view(f)[0]
        if cur != base:  # F / p buffers were swapped by single-verb calls: re-alias
            self._build_views()
            t, _ = self._views[f]
        return t[g0 - self.rows[0]: g1 - self.rows[0] + 1]

    # -- exchange ----------------------------------------------------------------------
    def _p2p_ops(self, f):
        """The (cached) send/recv descriptors of one field: W owned rows out, W halo rows in, per
        neighbour.  Row views alias library memory whose address is stable across whole steps; the
        cache is dropped if a single-sweep verb swapped the F buffers (`_rows_view` re-aliases)."""
        base = self.eng.field_view(f)[0]
        hit = self._ops.get(f)
        if hit is not None and hit[0] == base:
            return hit[1]
        dist, W = self.dist, self.halo
        lo, hi = self.own
        ops = []
        if self.rank > 0:  # lower neighbour owns rows < lo
            ops.append(dist.P2POp(dist.isend, self._rows_view(f, lo, lo + W - 1), self.rank - 1))
            ops.append(dist.P2POp(dist.irecv, self._rows_view(f, lo - W, lo - 1), self.rank - 1))
        if self.rank < self.world - 1:
            ops.append(dist.P2POp(dist.isend, self._rows_view(f, hi - W + 1, hi), self.rank + 1))
            ops.append(dist.P2POp(dist.irecv, self._rows_view(f, hi + 1, hi + W), self.rank + 1))
        self._ops[f] = (base, ops)
        return ops

    def _exchange_async(self, fields):
        """Post the halo send/recvs of `fields` with both neighbours as one batched P2P group;
        returns the outstanding works.  RCCL orders the transfers after everything already
        enqueued on the solver's stream and runs them on its own stream."""
        if self.stage_host:
            return self._exchange_staged(fields)
        ops = []
        for f in fields:
            ops += self._p2p_ops(f)
        return self.dist.batch_isend_irecv(ops) if ops else []

    def _exchange_staged(self, fields):
        """The same messages through host memory: the W owned rows next to each interior edge are copied to the host
        (which waits for the kernels that produced them), sent / received by the host backend, and the received rows
        copied into the halo rows when the work is waited for."""
        dist, W = self.dist, self.halo
        lo, hi = self.own
        ops, landings = [], []
        for f in fields:
            if self.rank > 0:
                ops.append(dist.P2POp(dist.isend, self._rows_view(f, lo, lo + W - 1).cpu(), self.rank - 1))
                dst = self._rows_view(f, lo - W, lo - 1)
                buf = self.torch.empty(dst.shape, dtype=dst.dtype, device="cpu")
                ops.append(dist.P2POp(dist.irecv, buf, self.rank - 1))
                landings.append((dst, buf))
            if self.rank < self.world - 1:
                ops.append(dist.P2POp(dist.isend, self._rows_view(f, hi - W + 1, hi).cpu(), self.rank + 1))
                dst = self._rows_view(f, hi + 1, hi + W)
                buf = self.torch.empty(dst.shape, dtype=dst.dtype, device="cpu")
                ops.append(dist.P2POp(dist.irecv, buf, self.rank + 1))
                landings.append((dst, buf))
        return [_StagedWork(dist.batch_isend_irecv(ops), landings)] if ops else []

    def exchange(self, fields=EXCHANGED):
        """Refresh the halo rows of `fields` from both neighbours and wait for them."""
        if self.world == 1:
            return
        if self.exchange_kind == "native":
            bit = {"F": _abi.VOF_XCHG_F, "u": _abi.VOF_XCHG_U, "v": _abi.VOF_XCHG_V, "p": _abi.VOF_XCHG_P,
                   "u_star": _abi.VOF_XCHG_US, "v_star": _abi.VOF_XCHG_VS, "rhs": _abi.VOF_XCHG_RHS}
            self.eng.comm_exchange(sum(bit[f] for f in fields))
            return
        with self._ctx():          # (ordered behind the kernels on the solver's stream)
            for w in self._exchange_async(fields):
                w.wait()

    def _ctx(self):
        import contextlib
        return self.torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def step(self, nsteps=1, overlap=True):
        """nsteps time steps.  With overlap (default) each field's halo is shipped as soon as the
        field is final for the step -- p after the pressure solve, u and v after the velocity
        correction, F after the transport -- so only F's exchange (a quarter of the bytes) is
        exposed; the rest runs on RCCL's stream under the remaining kernels.  A halo row that is
        still being read by a later kernel of the same step while it is received is either being
        overwritten with the identical value (rows inside this rank's still-valid region) or lies
        in the invalid fringe, so owned rows are unaffected; no kernel of a later phase writes a
        field whose exchange is in flight (DESIGN.md "strips")."""
        if self.world > 1 and self.exchange_kind == "native":
            # the whole loop in the library; True = the default schedule (mode 4: fused transport, edge bands
            # first, one send/recv group per step), an int = that mode of vof_step_exchange
            self.eng.step_exchange(nsteps, 4 if overlap is True else int(overlap))
            self._fresh = False
            return
        if self.world > 1 and overlap == 5 and overlap is not True:
            if self.on_gpu:
                return self._step_pieces(nsteps)
            return self._step_boundary_behind_the_predictor(nsteps)
        with self._ctx():
            for _ in range(nsteps):
                if self.world == 1:
                    self.eng.step(1)
                elif not overlap:
                    self.eng.step(1)
                    self.exchange()
                else:
                    self._phased_step()
            self._fresh = False

    def _phased_step(self):
        self.eng.step_phase(0)
        works = self._exchange_async(("p",))
        self.eng.step_phase(1)
        works += self._exchange_async(("u", "v"))
        self.eng.step_phase(2)
        works += self._exchange_async(("F",))
        for w in works:
            w.wait()

    def probe_cost(self, n=10, skip=2, overlap=True):
        """ms per step of THIS rank's kernels (device-timed on the handle's stream: no waiting for neighbours in the figure), on
        valid data: one kernel-only step, then a full halo exchange before the next (the kernels are data dependent -- zero
        shortcuts, division tiers --, stale halos would feed them garbage rows).  overlap == 5 on the GPU: the middle step of
        mode 5 (k_jacobi_pair on all stored rows + k_tm on the owned rows, vof_step_tm_piece(1)) -- what a run in that mode
        spends its time on; else the strip's plain step.  Either carrier.  Leaves the solver n + 2 steps into the run: the
        caller starts again from a new one (bench.py: strips.balanced_partition)."""
        e = self.eng
        samples = []
        if overlap == 5 and overlap is not True and self.on_gpu and self.world > 1:
            with self._ctx():
                if self.exchange_kind == "native":
                    e.step_exchange(1, 1)            # the first step after set_init_F, with its exchanges
                else:
                    self._phased_step()
                self._fresh = False
                e.step_tm_piece(0)
            self.exchange(EXCHANGED_MODE5 + EXCHANGED_PIECES)
            for _ in range(n):
                e.timer_start()
                e.step_tm_piece(1)
                samples.append(e.timer_stop())
                self.exchange(("F", "p") + EXCHANGED_MODE5 + EXCHANGED_PIECES)
            e.step_tm_piece(2)
            self.exchange(EXCHANGED)
        else:
            for _ in range(n):
                e.timer_start()
                e.step(1)
                samples.append(e.timer_stop())
                self.exchange()
            self._fresh = False
        return sum(samples[skip:]) / len(samples[skip:])

    def _step_pieces(self, nsteps):
        """Overlap mode 5 on the GPU with the halos carried by torch.distributed instead of the library's own RCCL
        communicator: the kernels of vof_step_exchange(.., 5) piece by piece (vof_step_tm_piece: k_momentum on the owned
        rows; k_jacobi_pair + k_tm per middle step; the last step as mode 4's) with one exchange behind each -- what the
        second carrier of bench.py --gpus N runs, and its one-GPU rehearsal (--same-device, gloo, host-staged)."""
        e = self.eng

        def trade(fields):
            for w in self._exchange_async(fields):
                w.wait()
        with self._ctx():
            if nsteps > 0 and self._fresh:       # (the first step after set_init_F: the phases carry the reference's set_BC calls)
                self._phased_step()
                self._fresh = False
                nsteps -= 1
            if nsteps <= 0:
                return
            e.step_tm_piece(0)
            trade(EXCHANGED_MODE5 + EXCHANGED_PIECES)
            for _ in range(nsteps - 1):
                e.step_tm_piece(1)
                trade(("F", "p") + EXCHANGED_MODE5 + EXCHANGED_PIECES)
            e.step_tm_piece(2)
            trade(EXCHANGED)

    def _step_boundary_behind_the_predictor(self, nsteps):
        """The torch carrier's form of overlap mode 5 (verb by verb: what the gloo tests drive; on the GPU the library runs
        it with its pair kernels, vof_step_exchange).  The step boundary moves behind the momentum predictor of the NEXT
        step (2dvof.py:513-517), so that what crosses a strip edge once per step is F, p and u*, v* -- the inputs of the
        ten sweeps and of the transport -- instead of F, u, v, p; u and v are exchanged only at the end of the call.
        (The library's kernels recompute rho, nu from F per cell and ship rhs with u*, v*; here rho and nu are fields, so F
        travels in front of cal_nu_rho.)"""
        e = self.eng
        first = e.istep + 1

        def predictor():
            e.cal_nu_rho(); e.get_normal_young(); e.advect_upwind(); e.set_BC()      # :513-518 of the coming step
            for w in self._exchange_async(EXCHANGED_MODE5):
                w.wait()

        def pressure_and_transport(istep):
            e.solve_p_jacobi(self.halo - 8)                                         # :521-522 (jacobi_iters sweeps)
            e.update_uv(); e.set_BC(); e.solve_VOF_rudman(istep); e.post_process_f(); e.set_BC()   # :524-528

        with self._ctx():
            predictor()
            for k in range(nsteps):
                pressure_and_transport(first + k)
                if k < nsteps - 1:
                    for w in self._exchange_async(("F", "p")):
                        w.wait()
                    predictor()
            e.istep = first + nsteps - 1
            for w in self._exchange_async(EXCHANGED):
                w.wait()

    def solve_p(self, tol, max_iters, check_every=10, criterion="abs"):
        """Extension: Jacobi until the GLOBAL residual <= tol -- "abs": max|p_new - p|, "rel": that
        over max(max|p_new|, tiny) -- both norms all-reduced (MAX) over the ranks.  With strips the
        sweeps between two halo refreshes of p are limited by the halo depth (jacobi_iters rows of
        it serve the sweeps), so a check covers at most that many sweeps.  Same sweep counts and
        the same rule as vof_solve_p on a single domain: n = min(check_every, max_iters - done)."""
        crit = {"abs": _abi.VOF_RESID_ABS, "rel": _abi.VOF_RESID_REL}[criterion]
        depth = self.halo - 8           # sweeps the deep halo of p covers between two exchanges
        if self.world > 1 and depth < 1:
            raise ValueError("solve_p on strips needs jacobi_iters >= 1 (the halo of p covers jacobi_iters sweeps "
                             "between two exchanges; this solver was made with jacobi_iters = %d)" % depth)
        if check_every < 1 or max_iters < 0:
            raise ValueError("check_every must be >= 1 and max_iters >= 0")
        done, res = 0, float("inf")
        first = True
        with self._ctx():
            while done < max_iters:
                n = min(check_every, max_iters - done)
                left = n
                while left > 0:             # (world == 1: one batch)
                    k = left if self.world == 1 else min(left, depth)
                    upd, pmax = self.eng.jacobi_sweeps_norms(k, build_rhs=first)
                    first = False
                    left -= k
                    if self.world > 1:
                        self.exchange(("p",))
                done += n
                if self.world > 1:
                    upd = self.comm.allreduce_max(upd, self.eng)
                    pmax = self.comm.allreduce_max(pmax, self.eng)
                res = self.api.residual_value(upd, pmax, crit)
                if res <= tol or not res < float("inf"):
                    break
        return done, res

    def solve_p_residual(self, tol, max_iters, check_every=10):
        return self.solve_p(tol, max_iters, check_every, "abs")

    def sync(self):
        self.eng.sync()

    def owned(self, name):
        """Owned rows of a field (plus the wall ghost row on the first/last rank), dense numpy."""
        g0 = 0 if self.own[0] == 1 else self.own[0]
        g1 = self.nx + 1 if self.own[1] == self.nx else self.own[1]
        return self.eng.get(name, (g0, g1))

    def gather(self, name):
        """Full (nx+2, ny+2) field on rank 0 (None elsewhere)."""
        mine = self.owned(name)
        if self.world == 1:
            return mine
        parts = self.comm.gather_object(mine)
        return np.concatenate(parts, axis=0) if self.rank == 0 else None

    def barrier(self):
        if self.world > 1:
            self.eng.sync()
            self.comm.barrier(self.eng)

    def close(self):
        self.eng.close()
