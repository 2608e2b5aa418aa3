"""HIP == oracle at the sizes and over the step ranges the bench numbers are quoted on.

The reference arithmetic these runs must reproduce: /root/reference/2dvof.py:258-266 (Jacobi), :321-448 (FCT
sweeps), :521-522 (ten sweeps per step), in the main-loop order :505-528.  The oracle (oracle/vof_oracle.c) is
pinned to that text by tests/test_ref_golden.py; here it runs on the GPU box's host cores next to the HIP library,
both through the same C-ABI prototypes, and every comparison is value for value (IEEE ==) on all four state fields.

  (i)   BASELINE configs[2], 4096^2 fp64 dam-break: steps 60, 90 (the tiny-value front appears, the equal-cost work
        plan of the fused Jacobi kernels switches on), 300, 600 (inside the front) and 1000 (past it) -- the whole range of
        bench.py's `sustained` record -- with THREE HIP handles beside the one oracle run, each pinned to one batch form
        (the default chosen by the library's rule, k_tm + k_jacobi_pair forced in 16-step batch graphs, chains forced),
        and what each handle really ran asserted from its counters;
  (ii)  BASELINE configs[3] on one GPU, 8192^2 fp64 (dt 1e-6): steps 30 and 70 -- 70 planned tile
        columns, the two-mask-word case of the plan; the same three forms;
  (iii) BASELINE configs[4], 2048^2 rising bubble fp32: steps 50, 150, 300 against the fp32 oracle, bit for bit
        (next to the mixed-precision tolerance test of test_parity_gpu.py);
  (iv)  the REAL fp32 front (|p| in 1e-45 .. 1e-25 behind the pressure wave: the fp32 scaled division tier, the
        integer tiny-test, the plan with k_jacobi_tb<float>): a 1536^2 fp32 dam-break, plan on == plan off == oracle.
"""
import numpy as np
import pytest

from util import STATE, engine, same, diff_report

pytestmark = pytest.mark.gpu


def _tiny_cells(p, dtype="f64"):
    lim = 1e-280 if dtype == "f64" else 1e-25      # (the decaying front: DivLimits<T>::lo of the kernels is 1e-289 / 1e-25)
    q = p[1:-1, 1:-1]
    return int(((np.abs(q) < lim) & (q != 0)).sum())


def _compare(a, b, ctx, dtype="f64"):
    """All four state fields, one at a time (537 MB each at 8192^2); returns the tiny-cell count of p."""
    tiny = 0
    for f in STATE:
        x, y = a.get(f), b.get(f)
        assert same(x, y), "%s: %s" % (ctx, diff_report(x, y, f))
        if f == "p":
            tiny = _tiny_cells(x, dtype)
        del x, y
    assert a.get_counter("courant_violations") == b.get_counter("courant_violations"), ctx
    return tiny


def _advance(a, st, chunk=10):
    """Step the HIP engine to step st in batches, counting the steps whose k_jacobi_tb launches were planned."""
    active = 0
    while a.istep < st:
        a.step(min(chunk, st - a.istep))
        active += a.get_counter("tb_plan_active")
    return active


def _three_forms(hip_api, n, **kw):
    """The three batch forms of a large fp64 full domain, each PINNED by its knobs (what ran is asserted by the caller from
    the handles' counters, not left to a choice made inside the library):
      default -- what a user gets: the rule of vof_step (k_tm + k_jacobi_pair on a dam-break), stepped in chunks of 10;
      pairs   -- fuse_tm = 1, jacobi_pair = 1, stepped in chunks of >= 32 so that the 16-step batch graphs run;
      chains  -- fuse_tm = 0: the four-kernel sequence as chains of launches on row blocks."""
    d = engine(hip_api, n, n, "f64", "f32", ic=1, **kw)
    p = engine(hip_api, n, n, "f64", "f32", ic=1, **kw)
    p.set_param("fuse_tm", 1)
    p.set_param("jacobi_pair", 1)
    c = engine(hip_api, n, n, "f64", "f32", ic=1, **kw)
    c.set_param("fuse_tm", 0)
    return d, p, c


def test_baseline_4096_matches_oracle_through_step_1000(hip_api, oracle_api):
    """(i) the whole `sustained` record of bench.py, 35-38 planned tile columns x 1000 steps, in all three batch forms
    beside ONE oracle run: the default, the pair kernels forced (16-step k_tm batch graphs, k_jacobi_pair's FAST and
    general sub-iterations, its work plan through the front), the chains forced."""
    n = 4096
    d, p, c = _three_forms(hip_api, n)
    b = engine(oracle_api, n, n, "f64", "f32", ic=1)
    active, tiny = {}, {}
    for st in (60, 90, 300, 600, 1000):
        active[st] = _advance(d, st)
        _advance(p, st, chunk=48)
        _advance(c, st, chunk=48)
        b.step(st - b.istep)
        tiny[st] = _compare(d, b, "4096^2 fp64 step %d, default form (tm_choice %d)" % (st, d.get_counter("tm_choice")))
        _compare(p, b, "4096^2 fp64 step %d, k_tm + k_jacobi_pair forced" % st)
        _compare(c, b, "4096^2 fp64 step %d, chains forced" % st)
    assert tiny[90] > 10000 and tiny[300] > 10000 and tiny[600] > 1000, tiny   # the front was there ...
    assert active[90] >= 1 and active[300] >= 10 and active[600] >= 10, active  # ... the plan ran through it ...
    assert tiny[1000] == 0 and d.get_counter("tb_plan_active") == 0, (tiny, active)   # ... and both are gone at the end
    # which kernels met the oracle: counted, per handle
    forms = {k: (e.get_counter("tm_choice"), e.get_counter("tm_steps"), e.get_counter("pair_launches"), e.get_counter("halves_steps"))
             for k, e in (("default", d), ("pairs", p), ("chains", c))}
    assert forms["pairs"][1] >= 950 and forms["pairs"][2] >= 950 and forms["pairs"][3] == 0, forms
    assert forms["chains"][1] == 0 and forms["chains"][2] == 0 and forms["chains"][3] >= 900, forms
    assert forms["default"][0] == 1 and forms["default"][1] >= 900 and forms["default"][2] >= 900, forms   # the rule: a dam-break runs the pair kernels
    assert d.get_counter("courant_violations") == 0
    F = d.get("F")
    assert F.min() >= 0.0 and F.max() <= 1.0


def test_baseline_8192_matches_oracle(hip_api, oracle_api):
    """(ii) configs[3]'s grid on one GPU against the oracle: 70-76 tile columns, two mask words per band, pair chunks
    of two residency rounds -- in the same three pinned forms."""
    n = 8192
    d, p, c = _three_forms(hip_api, n, dt=1e-6)
    b = engine(oracle_api, n, n, "f64", "f32", ic=1, dt=1e-6)
    active, tiny = {}, {}
    for st in (30, 70):
        active[st] = _advance(d, st, chunk=5)
        _advance(p, st, chunk=40)
        _advance(c, st, chunk=40)
        b.step(st - b.istep)
        tiny[st] = _compare(d, b, "8192^2 fp64 step %d, default form (tm_choice %d)" % (st, d.get_counter("tm_choice")))
        _compare(p, b, "8192^2 fp64 step %d, k_tm + k_jacobi_pair forced" % st)
        _compare(c, b, "8192^2 fp64 step %d, chains forced" % st)
    assert tiny[70] > 10000, tiny
    assert active[70] >= 1 and d.get_counter("tb_plan_active") == 1, active
    forms = {k: (e.get_counter("tm_choice"), e.get_counter("tm_steps"), e.get_counter("pair_launches"), e.get_counter("halves_steps"))
             for k, e in (("default", d), ("pairs", p), ("chains", c))}
    assert forms["pairs"][1] >= 60 and forms["pairs"][2] >= 60 and forms["pairs"][3] == 0, forms
    assert forms["chains"][1] == 0 and forms["chains"][3] >= 48, forms
    assert forms["default"][0] == 1 and forms["default"][1] >= 40, forms


def test_baseline_config4_2048_bubble_fp32_matches_fp32_oracle(hip_api, oracle_api):
    """(iii) configs[4] bit for bit: 17 tile columns of k_jacobi_tb<float>, the CSF path, fp32 work plan."""
    n = 2048
    a = engine(hip_api, n, n, "f32", "f32", ic=2)
    b = engine(oracle_api, n, n, "f32", "f32", ic=2)
    for st in (50, 150, 300):
        _advance(a, st)
        b.step(st - b.istep)
        _compare(a, b, "2048^2 bubble fp32 step %d" % st, "f32")
    F = a.get("F")
    assert F.dtype == np.float32 and F.min() >= 0.0 and F.max() <= 1.0


def test_real_fp32_front_matches_oracle(hip_api, oracle_api):
    """(iv) fp32 dam-break on 14 tile columns: behind the pressure wave p walks down through 1e-25 .. 1.4e-45 to exact
    zero from the first steps on (an fp32 number runs out of exponent after ~ 100 cells), so the scaled tier of the
    fp32 division and the work plan are active for the whole run: plan on == plan off == oracle."""
    n = 1536
    on = engine(hip_api, n, n, "f32", "f32", ic=1)
    off = engine(hip_api, n, n, "f32", "f32", ic=1)
    off.set_param("jacobi_tb_adapt", 0)
    b = engine(oracle_api, n, n, "f32", "f32", ic=1)
    active, tiny = 0, []
    for st in (10, 30, 80, 160, 300):
        active += _advance(on, st, chunk=5)
        off.step(st - off.istep)
        b.step(st - b.istep)
        tiny.append(_compare(on, b, "1536^2 fp32 plan on, step %d" % st, "f32"))
        _compare(off, b, "1536^2 fp32 plan off, step %d" % st, "f32")
    assert min(tiny[1:]) > 1000, tiny
    assert active >= 20 and off.get_counter("tb_plan_active") == 0, active
