"""Golden vectors produced by EXECUTING the reference's own source text.

BUILD CONTAINER ONLY (needs /root/reference; listed in .gpurunignore, only the
``ref_*.npz`` files it writes travel to the GPU box).

    python tests/golden/make_ref_golden.py --ic 1 --steps 1000 --full 2 100 1000 -s      (as shipped: 200 x 200)
    python tests/golden/make_ref_golden.py --ic 2 --steps 300 --grid 48 80               (rectangular cells)

What it does: runs ``/root/reference/2dvof.py`` unmodified (``runpy.run_path``,
``sys.argv = ['2dvof.py', '-ic', N]``) with a pure-Python stand-in for the one
module the image lacks, ``taichi`` (taichi==1.4.1, requirements.txt:3, no wheel
and no network here).  The kernels' arithmetic is the reference's text,
interpreted statement by statement by CPython -- not a restatement:

  * ``@ti.kernel`` / ``@ti.func``      -> the function itself (see ``_scoped``),
  * ``ti.field``                       -> a zero-initialised NumPy array behind
                                          ``[]``, ``to_numpy``, ``from_numpy``,
  * ``ti.ndrange / ti.grouped``        -> ``itertools.product`` (sequential),
  * ``ti.max / min / sqrt / abs``      -> Python's, on IEEE doubles,
  * ``ti.GUI``                         -> headless; ``running`` turns False after
                                          ``--steps`` iterations of the reference's
                                          own ``while gui.running`` loop (:505).

Every field value is an IEEE double (``np.float64`` scalars: the reference with
only ``default_fp`` changed to f64, which is what BASELINE's fp64 configs are;
the float32 cast of the coordinates, :43-46, is the reference's and is kept).
In double precision the result does not depend on Taichi's static typing (every
int -> float conversion is exact, every Python-scope constant expression is a
double either way); in fp32 it would (``ap`` at :262, the cell coordinates at
:105-106 mix ints, constants and field values), so no fp32 vectors are made.

What this pins: the oracle's and the HIP kernels' reading of 2dvof.py:102-455,
:505-528 -- operation order, switches, loop ranges, zero ghosts, BC order, the
step-parity rule.  What it does NOT pin: Taichi's code generation
(``fast_math=True`` may contract / reassociate) and its parallel
execution -- the stand-in runs each loop sequentially.  That is value-identical
here because no loop of the hot path reads what another iteration of the same
loop writes, except the face fluxes ``ax`` / ``ay`` that two iterations store
with bit-identical expressions.

Three deviations from plain CPython: (1) reads one row past a field return 0
(Field.__getitem__; interp_velocity :490-492 does that, undefined behaviour in
Taichi; --vis only); (2) `x ** 2` on run-time values is x * x as in Taichi, not
libm's pow(), which is off by an ulp now and then (_Scoper.visit_BinOp;
find_area :120-124 and get_vnorm_field :486); and (3), needed to run the text
at all:
``cal_nu_rho`` (:200-201) assigns a local ``F`` from the global field ``F``
(``F = var(0.0, 1.0, F[I])``).  Taichi resolves the right-hand side before the
local exists; CPython raises UnboundLocalError.  ``_scoped`` renames such
locals (``F`` -> ``F__local`` after the assignment, inside the enclosing block)
and changes nothing else.
"""
import argparse
import ast
import hashlib
import inspect
import itertools
import math
import os
import runpy
import sys
import tempfile
import textwrap
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
STATE = ("F", "u", "v", "p")
ALL_FIELDS = ("F", "Ftd", "ax", "ay", "cx", "cy", "rp", "rm", "u", "v", "u_star", "v_star", "p", "pt",
              "rho", "nu", "mx", "my", "kappa")


# ----------------------------------------------------------------------------------------------
# the stand-in module
F32 = False        # --f32: the reference AS SHIPPED (default_fp = ti.f32, 2dvof.py:9), see f32_notes below
_KERNEL_DEPTH = [0]   # > 0 while a @ti.kernel / @ti.func body runs

f32_notes = """--f32 emulates Taichi's static typing with NumPy scalars:
  * fields hold float32; inside a kernel an entry reads as np.float32, at Python scope as a Python
    float (what ti.field.__getitem__ returns), so Python-scope constants (dx, dxi, dx*dy ...) are
    doubles folded by Python exactly as in the reference;
  * np.float32 (op) Python float -> float32 (NumPy's weak-scalar rule): a folded double constant is
    rounded to f32 once, where it meets a run-time value -- Taichi's rule for captured constants;
  * loop indices are TiInt: index arithmetic stays integer, and an index that meets a float is
    converted to f32 first (`(i - imin) * dx`, :105-106: i32 -> f32, then an f32 multiply);
  * a kernel-local assigned from a Python float, and the value of `a if c else b` (a run-time
    select), are f32 (`ap`, :258-262, is an f32 sum of f32 coefficients);
  * ti.sqrt of a run-time value is the correctly rounded f32 square root; of a Python constant,
    math.sqrt (Taichi evaluates it in Python)."""


def _ti_loc(v):
    """A kernel-local variable / a run-time select holding a Python float is default_fp-typed."""
    return np.float32(v) if (F32 and type(v) is float) else v


class TiInt(int):
    """A loop index (ti.i32): integer arithmetic stays TiInt; meeting a float it becomes f32 first."""

    def _f(self):
        return np.float32(int(self))

    def __add__(self, o):
        return TiInt(int(self) + o) if isinstance(o, int) else self._f() + o

    def __radd__(self, o):
        return TiInt(o + int(self)) if isinstance(o, int) else o + self._f()

    def __sub__(self, o):
        return TiInt(int(self) - o) if isinstance(o, int) else self._f() - o

    def __rsub__(self, o):
        return TiInt(o - int(self)) if isinstance(o, int) else o - self._f()

    def __mul__(self, o):
        return TiInt(int(self) * o) if isinstance(o, int) else self._f() * o

    def __rmul__(self, o):
        return TiInt(o * int(self)) if isinstance(o, int) else o * self._f()

    def __floordiv__(self, o):
        return TiInt(int(self) // o)

    def __truediv__(self, o):
        return self._f() / o


def _idx(k):
    return TiInt(k) if F32 else k


class Field:
    """ti.field(float, shape): dense, zero-initialised, row-major."""

    def __init__(self, dtype=float, shape=()):
        shape = (shape,) if isinstance(shape, int) else tuple(shape)
        self.a = np.zeros(shape, dtype=(np.float32 if F32 else np.float64) if dtype is float else dtype)

    oob_reads = 0

    def __getitem__(self, idx):
        if idx is None:                       # sigma[None]
            return self.a[()] if (_KERNEL_DEPTH[0] or not F32) else float(self.a[()])
        if F32 and _KERNEL_DEPTH[0] == 0:     # Python scope (dx = x[3] - x[2], :47): a Python float
            return float(self.a[idx])
        if isinstance(idx, tuple) and len(idx) == 2 and idx[0] == self.a.shape[0]:
            # interp_velocity (:490-492) loops i up to imax+1 and reads u[i+1, j] = u[imax+2, j], one row
            # past the field: undefined in Taichi's release mode (no bounds check).  Reads as 0 here
            # (what follows a dense field in Taichi's zero-filled root buffer is another field or
            # padding); only the display vector V[imax+1, :, 0] depends on it.  Counted.
            Field.oob_reads += 1
            return self.a.dtype.type(0)
        return self.a[idx]

    def __setitem__(self, idx, val):
        if idx is None:
            self.a[()] = val
        else:
            self.a[idx] = val

    def __iter__(self):                       # ``for i, j in F`` (:453)
        return iter(tuple(_idx(k) for k in t) for t in itertools.product(*(range(n) for n in self.a.shape)))

    @property
    def shape(self):
        return self.a.shape

    def to_numpy(self):
        return self.a.copy()

    def from_numpy(self, arr):
        assert arr.shape == self.a.shape
        self.a[...] = arr                     # float32 coordinates widen exactly (:44, :46)


class IVec(tuple):
    """Index vector of ti.grouped: supports ``I // r`` (:461)."""

    def __floordiv__(self, r):
        return IVec(int(k) // int(r) for k in self)


class Vector(list):
    @staticmethod
    def field(n, dtype=float, shape=()):
        shape = (shape,) if isinstance(shape, int) else tuple(shape)
        return Field(dtype, shape + (n,))


def ndrange(*dims):
    rs = [range(*d) if isinstance(d, tuple) else range(d) for d in dims]
    if len(rs) == 1:
        return (_idx(k) for k in rs[0])
    return (tuple(_idx(k) for k in t) for t in itertools.product(*rs))


def grouped(f):
    return (IVec(t) for t in itertools.product(*(range(n) for n in f.shape)))


class _Scoper(ast.NodeTransformer):
    """See the module docstring: locals that shadow a module global are renamed from the
    assignment on, block-scoped like Taichi's kernel variables."""

    def __init__(self, glob):
        self.glob = glob
        self.map = {}

    def visit_Name(self, node):
        if isinstance(node.ctx, ast.Load) and node.id in self.map:
            return ast.copy_location(ast.Name(self.map[node.id], ast.Load()), node)
        return node

    def _block(self, stmts):
        saved = dict(self.map)
        out = [self.visit(s) for s in stmts]
        self.map = saved
        return out

    def visit_FunctionDef(self, node):
        node.body = self._block(node.body)
        return node

    def visit_For(self, node):
        node.iter = self.visit(node.iter)
        node.body = self._block(node.body)
        return node

    def visit_If(self, node):
        node.test = self.visit(node.test)
        node.body = self._block(node.body)
        node.orelse = self._block(node.orelse)
        return node

    def visit_BinOp(self, node):
        node.left, node.right = self.visit(node.left), self.visit(node.right)
        # `x ** n`, n an integer literal, on a RUN-TIME value (a field entry, a kernel variable): Taichi
        # lowers it to repeated multiplication (demote_operations: square-and-multiply), while CPython
        # hands np.float64 ** 2 to libm's pow(), which is not correctly rounded.  A bare module-level
        # constant (`dxi ** 2`, :216-261) is folded by Python in Taichi too and stays Python's `**`.
        if isinstance(node.op, ast.Pow) and isinstance(node.right, ast.Constant) and isinstance(node.right.value, int) \
                and not (isinstance(node.left, ast.Name) and isinstance(self.glob.get(node.left.id), (int, float))):
            call = ast.Call(ast.Name("__ti_pow__", ast.Load()), [node.left, node.right], [])
            return ast.copy_location(call, node)
        return node

    def visit_IfExp(self, node):
        self.generic_visit(node)
        return ast.copy_location(ast.Call(ast.Name("__ti_loc__", ast.Load()), [node], []), node)

    def visit_Assign(self, node):
        node.value = self.visit(node.value)   # right-hand side first, with the names as they were
        if all(isinstance(t, ast.Name) for t in node.targets):   # a kernel local: default_fp-typed (--f32)
            node.value = ast.copy_location(ast.Call(ast.Name("__ti_loc__", ast.Load()), [node.value], []), node.value)
        for t in node.targets:
            if isinstance(t, ast.Name) and t.id in self.glob and isinstance(self.glob[t.id], Field):
                self.map[t.id] = t.id + "__local"
                t.id = t.id + "__local"
            else:
                self.visit(t)
        return node


def _ti_pow(a, n):
    """Taichi's integer power of a run-time value: square-and-multiply, starting from 1."""
    result, b = type(a)(1), abs(n)
    while b:
        if b & 1:
            result = result * a
        a = a * a
        b >>= 1
    return 1 / result if n < 0 else result


def _scoped(fn):
    src = textwrap.dedent(inspect.getsource(fn))
    tree = ast.parse(src)
    fdef = tree.body[0]
    fdef.decorator_list = []
    sc = _Scoper(fn.__globals__)
    sc.visit(fdef)
    ast.fix_missing_locations(tree)
    ast.increment_lineno(tree, fn.__code__.co_firstlineno - 1)
    ns = {}
    fn.__globals__["__ti_pow__"] = _ti_pow
    fn.__globals__["__ti_loc__"] = _ti_loc
    exec(compile(tree, fn.__code__.co_filename, "exec"), fn.__globals__, ns)
    body = ns[fn.__name__]

    def scoped(*a, **k):
        _KERNEL_DEPTH[0] += 1
        try:
            return body(*a, **k)
        finally:
            _KERNEL_DEPTH[0] -= 1
    scoped.__name__ = fn.__name__
    return scoped


class GUI:
    RELEASE, SPACE = "release", " "
    hook = None                               # called at the top of every iteration of :505
    limit = 0
    press_space_after_display = False

    def __init__(self, *a, **k):
        self._reads = 0
        self.shown = 0

    @property
    def running(self):                        # read once per iteration by ``while gui.running``
        self._reads += 1
        return self._reads <= GUI.limit

    @running.setter
    def running(self, v):
        pass

    def get_events(self, *a):
        if GUI.hook:
            GUI.hook()
        # --vis: one SPACE release right after every display (:507-509), so the reference's own
        # event loop walks vis_option through 0..4 and its display branches (:531-559) all run
        if GUI.press_space_after_display and self._reads > 1 and (self._reads - 1) % 100 == 0:
            return [types.SimpleNamespace(key=GUI.SPACE)]
        return []

    def arrows(self, orig=None, direction=None, **k):     # flow_visualization.py:55
        self.arrow_args = (np.array(orig), np.array(direction))

    def set_image(self, img):
        self.image = np.asarray(img)

    def show(self, *a):
        self.shown += 1

    def lines(self, *a, **k):
        pass

    def triangles(self, *a, **k):
        pass


def _tmax(*a):
    m = a[0]
    for b in a[1:]:
        m = m if m > b else b                  # ti.max(a, b) = a > b ? a : b, folded left to right
    return m


def _tmin(*a):
    m = a[0]
    for b in a[1:]:
        m = m if m < b else b
    return m


def make_taichi():
    ti = types.ModuleType("taichi")
    ti.cpu, ti.gpu, ti.f32, ti.f64, ti.i32 = "cpu", "gpu", np.float32, np.float64, int
    ti.init = lambda **kw: None               # default_fp is overridden: doubles (see docstring)
    ti.field = lambda dtype=float, shape=(): Field(dtype, shape)
    ti.Vector = Vector
    ti.kernel = _scoped
    ti.func = _scoped
    ti.ndrange, ti.grouped = ndrange, grouped
    ti.max, ti.min, ti.abs = _tmax, _tmin, abs
    def _sqrt(x):
        if isinstance(x, np.float32):
            return np.sqrt(x) if x >= 0 else np.float32("nan")
        return math.sqrt(x) if x == x and x >= 0 else float("nan")
    ti.sqrt = _sqrt
    ti.GUI = GUI
    return ti


# ----------------------------------------------------------------------------------------------
def digest(a):
    """sha256 of the values (+0.0 folds the sign of an exact zero, which parity does not compare)."""
    return hashlib.sha256(np.ascontiguousarray(a + 0.0).tobytes()).hexdigest()


def digest_steps(steps):
    """Steps whose 19 fields are recorded as digests: 0..10, then every 10th, and the last."""
    return sorted(set(range(0, min(steps, 10) + 1)) | set(range(0, steps + 1, 10)) | {steps})


def run_reference(ic, steps, save_fig, full, grid=None, vis=False):
    sys.modules["taichi"] = make_taichi()
    sys.path.insert(0, REF)
    os.environ.setdefault("MPLBACKEND", "Agg")
    want_digest, want_full = set(digest_steps(steps)), set(full)
    kept, rows, t0, mod, shown_vis = {}, [], time.time(), {}, {}

    def record(g, st):
        if vis and st > 0 and st % 100 == 0:
            # what the display branch of :531-559 left behind at this step: vis_option has not been
            # advanced yet (the injected SPACE is processed after this hook returns)
            opt = g["vis_option"] % 5
            d = {"option": opt, "shown": g["gui"].shown}
            if opt < 4:
                d["rgb"] = g["rgb_buf"].to_numpy()
            else:
                d["V"] = g["V"].to_numpy()
                d["orig"], d["direction"] = g["gui"].arrow_args
            shown_vis[st] = d
        if st in want_full:
            kept[st] = {n: g[n].to_numpy() for n in STATE}
        if st in want_digest:
            for f in ALL_FIELDS:
                arr = g[f].to_numpy()
                rows.append(("%s_%d" % (f, st), digest(arr), float(arr.sum()), float(np.abs(arr).max())))
        if st % 10 == 0:
            print("  [ref ic%d] step %d done, %.0f s" % (ic, st, time.time() - t0), flush=True)

    def hook():
        # called by gui.get_events at the top of iteration istep (already incremented, :506-507):
        # the fields hold the state after step istep-1.  The module's globals are the caller's.
        if "g" not in mod:
            mod["g"] = sys._getframe(2).f_globals
        record(mod["g"], mod["g"]["istep"] - 1)

    GUI.hook, GUI.limit, GUI.press_space_after_display = hook, steps, vis
    argv, cwd = sys.argv, os.getcwd()
    work = tempfile.mkdtemp(prefix="refrun_")
    os.chdir(work)
    sys.argv = ["2dvof.py", "-ic", str(ic)] + (["-s"] if save_fig else [])
    try:
        path = os.path.join(REF, "2dvof.py")
        if grid is None:
            g = runpy.run_path(path, run_name="__main__")
        else:
            # the reference's grid size is an edit-the-source constant (:19-20): the same text with ONLY
            # the two literals of `nx = 200` / `ny = 200` replaced (line numbers unchanged, so the
            # kernels' source is still read from the reference file itself)
            tree = ast.parse(open(path).read(), filename=path)
            hit = 0
            for node in tree.body:
                if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name) \
                        and node.targets[0].id in ("nx", "ny") and isinstance(node.value, ast.Constant):
                    node.value = ast.copy_location(ast.Constant(grid[0] if node.targets[0].id == "nx" else grid[1]), node.value)
                    hit += 1
            assert hit == 2
            g = {"__name__": "__main__", "__file__": path}
            exec(compile(tree, path, "exec"), g)
    finally:
        sys.argv = argv
        os.chdir(cwd)
    assert g["istep"] == steps
    record(g, steps)
    const = {k: float(g[k]) for k in ("dx", "dy", "dxi", "dyi", "dt", "Lx", "Ly", "rho_l", "rho_g", "nu_l", "nu_g")}
    const["sigma"] = float(g["sigma"][None])
    pngs = sorted(os.listdir(os.path.join(work, "output")))
    return g["nx"], g["ny"], kept, rows, const, pngs, g["gui"].shown, shown_vis


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ic", type=int, required=True)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--full", type=int, nargs="*", default=None,
                    help="steps whose F,u,v,p are stored in full (default 2 and the last)")
    ap.add_argument("-s", action="store_true", help="pass -s to the reference (PNG path, needs >= 100 steps)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--grid", type=int, nargs=2, default=None, metavar=("NX", "NY"),
                    help="run the reference text with its two grid-size literals (:19-20) replaced (default: as shipped, 200 200)")
    ap.add_argument("--f32", action="store_true",
                    help="the reference as shipped (default_fp = ti.f32): emulate Taichi's static typing (see f32_notes)")
    ap.add_argument("--vis", action="store_true",
                    help="inject a SPACE key release after every display so the reference cycles through its five "
                         "display branches (:531-559); records rgb_buf / V / the gui.arrows arguments at the 100-step marks")
    a = ap.parse_args()
    global F32
    F32 = a.f32
    full = sorted(set(a.full if a.full is not None else (2, a.steps)) | {0})
    nx, ny, kept, rows, const, pngs, shown, shown_vis = run_reference(a.ic, a.steps, a.s, full, a.grid, a.vis)
    out = {"meta": np.array([nx, ny, a.ic, 1 if a.f32 else 0, 1]),      # same meta as make_golden.py: dtype code (0 f64, 1 f32), coord cast kept
           "steps": np.array([s for s in full if s > 0]), "nsteps": np.array(a.steps), "F_0": kept[0]["F"],
           "const_names": np.array(sorted(const)), "const": np.array([const[k] for k in sorted(const)]),
           "pngs": np.array(pngs), "gui_shown": np.array(shown)}
    for st in full:
        if st > 0:
            for f in STATE:
                out["%s_%d" % (f, st)] = kept[st][f]
    # sha256 of the reference file the vectors came from (tests check it is recorded)
    out["oob_reads"] = np.array(Field.oob_reads)
    out["ref_sha256"] = np.array(hashlib.sha256(open(os.path.join(REF, "2dvof.py"), "rb").read()).hexdigest())
    if shown_vis:
        out["vis_steps"] = np.array(sorted(shown_vis))
        for st, d in shown_vis.items():
            out["vis_option_%d" % st] = np.array(d["option"])
            for k in ("rgb", "V", "orig", "direction"):
                if k in d and d[k].size > 40000:      # as-shipped size: digest only (1.3 MB per image)
                    out["vis_%s_sha256_%d" % (k, st)] = np.array(digest(d[k]))
                elif k in d:
                    out["vis_%s_%d" % (k, st)] = d[k]
    # all 19 fields at digest_steps() as sha256 + two moments (diagnostics when a digest differs)
    out["digest_names"] = np.array([r[0] for r in rows])
    out["digest_sha256"] = np.array([r[1] for r in rows])
    out["digest_sum"] = np.array([r[2] for r in rows])
    out["digest_absmax"] = np.array([r[3] for r in rows])
    prec = "f32" if a.f32 else "f64"
    path = a.out or os.path.join(HERE, ("ref_ic%d_200_%s.npz" % (a.ic, prec)) if a.grid is None else
                                 ("ref_ic%d_%dx%d_%s%s.npz" % (a.ic, nx, ny, prec, "_vis" if a.vis else "")))
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
