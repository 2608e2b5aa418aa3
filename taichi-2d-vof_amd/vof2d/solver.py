"""VOF2D -- the reference's program interface, object-shaped.

2dvof.py exposes module-global ``ti.field`` objects (``F.to_numpy()``,
``x.from_numpy()``; :44,:46,:535,:565), a 0-D field ``sigma[None]`` (:28-29)
and zero-argument kernels (``set_BC()``, ``advect_upwind()`` ...; :498,:513-528).
VOF2D keeps those names and meanings; each call is one C-ABI call into the HIP
library.  ``step(n)`` runs the solver part of the main loop (:506-528) with
the fused kernel schedule.
"""
import numpy as np

from . import _abi
from ._lib import hip_api
from .engine import Engine, make_desc

FIELD_NAMES = ("F", "u", "v", "p", "u_star", "v_star", "mx", "my", "kappa", "rho", "nu", "rhs")


class Field:
    """Stand-in for a ``ti.field`` of shape (nx+2, ny+2): to_numpy / from_numpy / [i, j]."""

    def __init__(self, eng, name):
        self._eng, self.name = eng, name
        self.shape = (eng.nrows, eng.ny + 2)
        self.dtype = eng.np_dtype

    def to_numpy(self):
        return self._eng.get(self.name)

    def from_numpy(self, arr):
        self._eng.set(self.name, arr)

    def __getitem__(self, idx):
        i, j = idx
        return self._eng.get(self.name, rows=(int(i), int(i)))[0, int(j)]


class _Sigma:
    """sigma[None] of 2dvof.py:28-29."""

    def __init__(self, eng):
        self._eng = eng

    def __getitem__(self, key):
        return self._eng.get_param("sigma")

    def __setitem__(self, key, value):
        self._eng.set_param("sigma", float(value))


class VOF2D:
    def __init__(self, nx=200, ny=200, dtype="f32", coord_cast="f32", jacobi_iters=10, device=-1,
                 use_graph=True, stream=None, api=None, **consts):
        """Defaults are the reference's shipped problem (2dvof.py:9,19-20): 200x200, f32."""
        self.api = api if api is not None else hip_api()
        flags = 0 if use_graph else _abi.VOF_FLAG_NO_GRAPH
        self.desc = make_desc(self.api, nx, ny, dtype, coord_cast, jacobi_iters=jacobi_iters, device=device,
                              flags=flags, **consts)
        self.eng = Engine(self.api, self.desc, stream=stream)
        self.nx, self.ny = nx, ny
        self.imin, self.jmin, self.imax, self.jmax = 1, 1, nx, ny  # :37-40
        for name in FIELD_NAMES:
            setattr(self, name, Field(self.eng, name))
        self.sigma = _Sigma(self.eng)
        self.dt = self.eng.get_param("dt")
        self.dx, self.dy = self.eng.get_param("dx"), self.eng.get_param("dy")
        self.dxi, self.dyi = self.eng.get_param("dxi"), self.eng.get_param("dyi")

    # reference verbs ------------------------------------------------------
    def set_init_F(self, ic):
        self.eng.set_init_F(ic)

    def set_BC(self):
        self.eng.set_BC()

    def cal_nu_rho(self):
        self.eng.cal_nu_rho()

    def get_normal_young(self):
        self.eng.get_normal_young()

    def advect_upwind(self):
        self.eng.advect_upwind()

    def solve_p_jacobi(self, n=1):
        self.eng.solve_p_jacobi(n)

    def update_uv(self):
        self.eng.update_uv()

    def fct_x_sweep(self):
        self.eng.fct_x_sweep()

    def fct_y_sweep(self):
        self.eng.fct_y_sweep()

    def solve_VOF_rudman(self, istep=None):
        self.eng.solve_VOF_rudman(self.istep if istep is None else istep)

    def post_process_f(self):
        self.eng.post_process_f()

    # display fields (2dvof.py:458-492) -------------------------------------------
    def get_vof_field(self):
        return self.eng.vis_field("vof")

    def get_u_field(self):
        return self.eng.vis_field("u")

    def get_v_field(self):
        return self.eng.vis_field("v")

    def get_vnorm_field(self):
        return self.eng.vis_field("vnorm")

    def interp_velocity(self):
        return self.eng.interp_velocity()

    # main loop --------------------------------------------------------------
    def step(self, nsteps=1):
        self.eng.step(nsteps)

    @property
    def istep(self):
        return self.eng.istep

    @istep.setter
    def istep(self, v):
        self.eng.istep = v

    def step_verbs(self, nsteps=1):
        """The main loop of 2dvof.py:506-528 written verb by verb (literal schedule)."""
        for _ in range(nsteps):
            self.istep = self.istep + 1
            self.cal_nu_rho()
            self.get_normal_young()
            self.advect_upwind()
            self.set_BC()
            self.solve_p_jacobi(self.desc.jacobi_iters)
            self.update_uv()
            self.set_BC()
            self.solve_VOF_rudman(self.istep)
            self.post_process_f()
            self.set_BC()

    def sync(self):
        self.eng.sync()

    @property
    def courant_violations(self):
        return self.eng.get_counter("courant_violations")

    def state(self, names=("F", "u", "v", "p")):
        return {n: self.eng.get(n) for n in names}

    def close(self):
        self.eng.close()
