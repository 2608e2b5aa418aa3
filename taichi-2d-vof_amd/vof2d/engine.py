"""Thin object wrapper over one C-ABI handle (include/vof2d.h).

An Engine is one strip of the grid (the full domain when row_lo = 0 and
row_hi = nx+1).  It owns nothing but the handle; every method is one call
through the ABI.  `api` is an `_abi.Api`; the product passes the HIP library's
(`_lib.hip_api()`), tests may pass the oracle's.
"""
import ctypes as C

import numpy as np

from . import _abi

NP_DTYPE = {_abi.VOF_F64: np.float64, _abi.VOF_F32: np.float32}
DTYPE_CODE = {"f64": _abi.VOF_F64, "float64": _abi.VOF_F64, "fp64": _abi.VOF_F64,
              "f32": _abi.VOF_F32, "float32": _abi.VOF_F32, "fp32": _abi.VOF_F32}


class VofError(RuntimeError):
    pass


def dtype_code(dtype):
    if isinstance(dtype, str):
        return DTYPE_CODE[dtype.lower()]
    if dtype in (_abi.VOF_F64, _abi.VOF_F32) and not isinstance(dtype, type):
        return int(dtype)
    return {np.dtype(np.float64): _abi.VOF_F64, np.dtype(np.float32): _abi.VOF_F32}[np.dtype(dtype)]


def make_desc(api, nx, ny, dtype="f64", coord_cast="f32", rows=None, own=None, jacobi_iters=10,
              device=-1, flags=0, **consts):
    """vof_desc_default + overrides.  rows=(row_lo,row_hi), own=(own_lo,own_hi)."""
    d = _abi.Desc()
    rc = api.desc_default(C.byref(d), int(nx), int(ny), dtype_code(dtype))
    if rc != 0:
        raise VofError("vof_desc_default(%d, %d) failed: %s" % (nx, ny, _abi.ERRNAMES.get(rc, rc)))
    if coord_cast not in ("f32", "none"):
        raise ValueError("coord_cast must be 'f32' or 'none'")
    d.coord_cast_f32 = 1 if coord_cast == "f32" else 0
    if rows is not None:
        d.row_lo, d.row_hi = int(rows[0]), int(rows[1])
        d.own_lo = max(1, d.row_lo)
        d.own_hi = min(int(nx), d.row_hi)
    if own is not None:
        d.own_lo, d.own_hi = int(own[0]), int(own[1])
    d.jacobi_iters = int(jacobi_iters)
    d.device = int(device)
    d.flags = int(flags)
    for k, v in consts.items():
        if k not in ("Lx", "Ly", "rho_l", "rho_g", "nu_l", "nu_g", "sigma", "gx", "gy", "dt"):
            raise TypeError("unknown constant %r" % k)
        setattr(d, k, float(v))
    return d


class Engine:
    def __init__(self, api, desc, stream=None):
        self.api = api
        self.desc = desc
        self._h = _abi.H()
        rc = api.create(C.byref(desc), C.c_void_p(stream) if stream else None, C.byref(self._h))
        if rc != 0:
            self._h = None
            raise VofError("vof_create failed: %s" % _abi.ERRNAMES.get(rc, rc))
        self.nx, self.ny = desc.nx, desc.ny
        self.row_lo, self.row_hi = desc.row_lo, desc.row_hi
        self.own_lo, self.own_hi = desc.own_lo, desc.own_hi
        self.nrows = self.row_hi - self.row_lo + 1
        self.np_dtype = NP_DTYPE[desc.dtype]

    # -- plumbing ---------------------------------------------------------
    @property
    def handle(self):
        return self._h

    def _ck(self, rc, what):
        if rc != 0:
            msg = self.api.last_error(self._h) if self._h else b""
            raise VofError("%s%s failed: %s %s" % (self.api.prefix, what, _abi.ERRNAMES.get(rc, rc),
                                                   (msg or b"").decode(errors="replace")))

    def close(self):
        if self._h:
            self.api.destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- reference verbs (2dvof.py kernels) --------------------------------
    def set_init_F(self, ic):
        self._ck(self.api.set_init_F(self._h, int(ic)), "set_init_F")

    def set_BC(self):
        self._ck(self.api.set_BC(self._h), "set_BC")

    def cal_nu_rho(self):
        self._ck(self.api.cal_nu_rho(self._h), "cal_nu_rho")

    def get_normal_young(self):
        self._ck(self.api.get_normal_young(self._h), "get_normal_young")

    def advect_upwind(self):
        self._ck(self.api.advect_upwind(self._h), "advect_upwind")

    def solve_p_jacobi(self, n=1):
        self._ck(self.api.solve_p_jacobi(self._h, int(n)), "solve_p_jacobi")

    def update_uv(self):
        self._ck(self.api.update_uv(self._h), "update_uv")

    def fct_x_sweep(self):
        self._ck(self.api.fct_x_sweep(self._h), "fct_x_sweep")

    def fct_y_sweep(self):
        self._ck(self.api.fct_y_sweep(self._h), "fct_y_sweep")

    def solve_VOF_rudman(self, istep):
        self._ck(self.api.solve_VOF_rudman(self._h, int(istep)), "solve_VOF_rudman")

    def post_process_f(self):
        self._ck(self.api.post_process_f(self._h), "post_process_f")

    def step(self, nsteps=1):
        self._ck(self.api.step(self._h, int(nsteps)), "step")

    def step_phase(self, phase):
        self._ck(self.api.step_phase(self._h, int(phase)), "step_phase")

    @property
    def istep(self):
        v = C.c_int64()
        self._ck(self.api.get_istep(self._h, C.byref(v)), "get_istep")
        return v.value

    @istep.setter
    def istep(self, value):
        self._ck(self.api.set_istep(self._h, int(value)), "set_istep")

    # -- extensions ---------------------------------------------------------
    def solve_p_residual(self, tol, max_iters, check_every=10):
        it, res = C.c_int32(), C.c_double()
        self._ck(self.api.solve_p_residual(self._h, float(tol), int(max_iters), int(check_every),
                                           C.byref(it), C.byref(res)), "solve_p_residual")
        return it.value, res.value

    def solve_p(self, tol, max_iters, check_every=10, criterion="abs"):
        """vof_solve_p: sweeps until the residual (\"abs\": max|p_new - p|; \"rel\": that over
        max(max|p_new|, tiny)) is <= tol; returns (sweeps done, residual).  Same defaults as
        StripSolver.solve_p and vof_solve_p_residual (absolute criterion, a check every 10 sweeps)."""
        crit = {"abs": _abi.VOF_RESID_ABS, "rel": _abi.VOF_RESID_REL}[criterion]
        it, res = C.c_int32(), C.c_double()
        self._ck(self.api.solve_p(self._h, float(tol), int(max_iters), int(check_every), crit,
                                  C.byref(it), C.byref(res)), "solve_p")
        return it.value, res.value

    def jacobi_sweeps_norms(self, n, build_rhs=True):
        """(max|p_new - p|, max|p_new|) of the last of n sweeps over the owned rows."""
        upd, pm = C.c_double(), C.c_double()
        self._ck(self.api.jacobi_sweeps_norms(self._h, int(n), 1 if build_rhs else 0, C.byref(upd), C.byref(pm)),
                 "jacobi_sweeps_norms")
        return upd.value, pm.value

    def jacobi_sweeps_residual(self, n, build_rhs=True):
        res = C.c_double()
        self._ck(self.api.jacobi_sweeps_residual(self._h, int(n), 1 if build_rhs else 0, C.byref(res)),
                 "jacobi_sweeps_residual")
        return res.value

    # -- fields -------------------------------------------------------------
    def get(self, name, rows=None):
        g0, g1 = (self.row_lo, self.row_hi) if rows is None else rows
        out = np.empty((g1 - g0 + 1, self.ny + 2), dtype=self.np_dtype)
        self._ck(self.api.get_rows(self._h, name.encode(), int(g0), int(g1),
                                   out.ctypes.data_as(C.c_void_p), out.nbytes), "get_rows(%s)" % name)
        return out

    def set(self, name, arr, rows=None):
        g0, g1 = (self.row_lo, self.row_hi) if rows is None else rows
        a = np.ascontiguousarray(arr, dtype=self.np_dtype)
        if a.shape != (g1 - g0 + 1, self.ny + 2):
            raise ValueError("field %s rows %d..%d: expected shape %s, got %s" %
                             (name, g0, g1, (g1 - g0 + 1, self.ny + 2), a.shape))
        self._ck(self.api.set_rows(self._h, name.encode(), int(g0), int(g1),
                                   a.ctypes.data_as(C.c_void_p), a.nbytes), "set_rows(%s)" % name)

    def vis_field(self, which):
        """rgb_buf of get_vof_field / get_u_field / get_v_field / get_vnorm_field (2dvof.py:458-486)."""
        out = np.empty((2 * self.nx, 2 * self.ny), dtype=self.np_dtype)
        self._ck(self.api.get_vis_field(self._h, which.encode(), out.ctypes.data_as(C.c_void_p), out.nbytes),
                 "get_vis_field(%s)" % which)
        return out

    def interp_velocity(self):
        """V of interp_velocity (2dvof.py:488-492): (nx+2, ny+2, 2)."""
        out = np.empty((self.nx + 2, self.ny + 2, 2), dtype=self.np_dtype)
        self._ck(self.api.interp_velocity(self._h, out.ctypes.data_as(C.c_void_p), out.nbytes), "interp_velocity")
        return out

    def field_view(self, name):
        """(device base pointer, pitch, col0, nrows) -- element (i, j) at
        base + ((i-row_lo)*pitch + col0 + j) * itemsize."""
        base, pitch, col0, nrows = C.c_void_p(), C.c_int64(), C.c_int64(), C.c_int64()
        self._ck(self.api.field_view(self._h, name.encode(), C.byref(base), C.byref(pitch),
                                     C.byref(col0), C.byref(nrows)), "field_view(%s)" % name)
        return base.value, pitch.value, col0.value, nrows.value

    def copy_rows_from(self, src, name, g0, g1):
        self._ck(self.api.copy_rows(self._h, src._h, name.encode(), int(g0), int(g1)), "copy_rows")

    # -- scalars --------------------------------------------------------------
    def set_param(self, name, value):
        self._ck(self.api.set_param(self._h, name.encode(), float(value)), "set_param(%s)" % name)

    def get_param(self, name):
        v = C.c_double()
        self._ck(self.api.get_param(self._h, name.encode(), C.byref(v)), "get_param(%s)" % name)
        return v.value

    def get_counter(self, name):
        v = C.c_int64()
        self._ck(self.api.get_counter(self._h, name.encode(), C.byref(v)), "get_counter(%s)" % name)
        return v.value

    # -- sync / timing ----------------------------------------------------------
    # -- strips over RCCL, inside the library (include/vof2d.h "strips over RCCL") ------
    def comm_init(self, uid, rank, world, loopback=False):
        buf = C.create_string_buffer(bytes(uid), _abi.VOF_COMM_ID_BYTES)
        self._ck(self.api.comm_init(self._h, buf, int(rank), int(world),
                                    _abi.VOF_COMM_LOOPBACK if loopback else 0), "comm_init")

    def comm_exchange(self, mask):
        self._ck(self.api.comm_exchange(self._h, int(mask)), "comm_exchange")

    def step_exchange(self, nsteps=1, overlap=1):
        """overlap: 0 = one exchange after the step, 1 = per field as soon as final, 3 = p, u, v in one
        group after the first sweep, 4 = fused transport on the edge bands first, one group for all
        four fields under the transport of the other rows, 5 = the pair kernels of the single GPU (k_jacobi_pair, k_tm)
        with u*, v*, rhs, F, p exchanged once per step (include/vof2d.h)."""
        self._ck(self.api.step_exchange(self._h, int(nsteps), int(overlap)), "step_exchange")

    def step_tm_piece(self, piece):
        """The kernels of one piece of an overlap-mode-5 call without the exchange: 0 = the first step's k_momentum,
        1 = one middle step (k_jacobi_pair, k_tm), 2 = the last step (k_jacobi_tb x 2, k_transport)."""
        self._ck(self.api.step_tm_piece(self._h, int(piece)), "step_tm_piece")

    def comm_allreduce_max(self, value):
        v = C.c_double(float(value))
        self._ck(self.api.comm_allreduce_max(self._h, C.byref(v)), "comm_allreduce_max")
        return v.value

    def comm_info(self):
        """(RCCL version code, 1 if step_exchange replays captured graphs)."""
        ver, gr = C.c_int32(), C.c_int32()
        self._ck(self.api.comm_info(self._h, C.byref(ver), C.byref(gr)), "comm_info")
        return ver.value, gr.value

    def comm_destroy(self):
        self._ck(self.api.comm_destroy(self._h), "comm_destroy")

    def sync(self):
        self._ck(self.api.sync(self._h), "sync")

    def timer_start(self):
        self._ck(self.api.timer_start(self._h), "timer_start")

    def timer_stop(self):
        ms = C.c_float()
        self._ck(self.api.timer_stop(self._h, C.byref(ms)), "timer_stop")
        return ms.value

    def profile_steps(self, nsteps, reset=True):
        """In-situ per-kernel profile of nsteps fused steps: {kernel: (avg_us, launches)}."""
        if reset:
            self._ck(self.api.reset_profile(self._h), "reset_profile")
        self._ck(self.api.profile_steps(self._h, int(nsteps)), "profile_steps")
        out = {}
        for k in ("k_momentum", "k_set_bc", "k_jacobi", "k_jacobi_tb", "k_correct", "k_fct_x", "k_fct_y",
                  "k_transport", "k_normals", "k_kappa", "k_predictor", "k_rhs", "k_jacobi_pair", "k_tm", "k_tm_uv"):
            us, n = C.c_double(), C.c_int64()
            if self.api.get_profile(self._h, k.encode(), C.byref(us), C.byref(n)) != 0:
                continue   # a kernel this build of the library does not have
            if n.value:
                out[k] = (us.value, n.value)
        return out

    def time_jacobi(self, n):
        ms = C.c_float()
        self._ck(self.api.time_jacobi(self._h, int(n), C.byref(ms)), "time_jacobi")
        return ms.value


def selftest_division(api, dtype, n, seed):
    """vof_selftest_division: (a, b, q) arrays with q = the kernels' exact division of a by b."""
    dt = np.float64 if dtype_code(dtype) == _abi.VOF_F64 else np.float32
    a, b, q = (np.empty(n, dt) for _ in range(3))
    rc = api.selftest_division(dtype_code(dtype), n, seed, a.ctypes.data, b.ctypes.data, q.ctypes.data)
    if rc != 0:
        raise VofError("vof_selftest_division failed: %s" % _abi.ERRNAMES.get(rc, rc))
    return a, b, q


def comm_unique_id(api):
    """vof_comm_get_unique_id: the VOF_COMM_ID_BYTES one rank creates and every rank passes to comm_init."""
    buf = C.create_string_buffer(_abi.VOF_COMM_ID_BYTES)
    rc = api.comm_get_unique_id(buf)
    if rc != 0:
        raise VofError("vof_comm_get_unique_id failed: %s (is librccl.so.1 loadable?)" % _abi.ERRNAMES.get(rc, rc))
    return buf.raw

