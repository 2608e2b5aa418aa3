"""GPU parity in the regime the sustained headline number runs in.

Started from p = 0 (2dvof.py:67, fields are zero-initialised), the pressure iteration
(2dvof.py:258-266, ten sweeps per step, :521-522) sends a decaying front across the grid: behind the
ordinary values comes a band where p walks down through 1e-280 ... 4.9e-324 to exact zero.  There the
kernels' exact division takes its scaled tier, k_jacobi_tb reports the (row band, tile column) cells
where it did, and the NEXT step cuts every tile column into chunks of equal cost (tb_make_plan, the
planner block inside k_momentum, mask sets alternating with the step parity).  Which rows a wave
takes must never change a value.  These tests run that machinery on grids where the REAL front
exists and several tile columns are planned -- not the synthetic ring of
test_equal_cost_work_plan_of_the_fused_jacobi:

  (i)   against the CPU oracle through the whole life of the front on a 14-tile-column grid,
        square and rectangular cells (the SQ and the general form of k_jacobi_tb);
  (ii)  plan on == plan off at BASELINE sizes (4096^2: 35 tile columns, 250 steps; 8192^2: 70 tile
        columns, the two-mask-word case);
  (iii) strips (2 and 8) == single domain while the front crosses them, every strip planning its
        own chunks.

The oracle comparisons AT the BASELINE sizes (4096^2 through step 1000, 8192^2, 2048^2 fp32, the fp32 front)
are in tests/test_parity_at_size_gpu.py.
"""
import numpy as np
import pytest

from util import STATE, assert_fields_same, engine, same, diff_report
from vof2d import halo_rows

pytestmark = pytest.mark.gpu


def _tiny_cells(p, dtype="f64"):
    lim = 1e-280 if dtype == "f64" else 1e-25      # (the decaying front: DivLimits<T>::lo of the kernels is 1e-289 / 1e-25)
    q = p[1:-1, 1:-1]
    return int(((np.abs(q) < lim) & (q != 0)).sum())


@pytest.mark.parametrize("nx,ny,checkpoints", [
    (1536, 1536, (40, 55, 70, 85, 100, 150)),      # front alive in steps ~ 40 .. 100 (tiny cells: 36 000 at step 70)
    (1024, 2304, (40, 60, 80, 110)),               # rectangular cells: k_jacobi_tb<..., SQ = false>
])
def test_real_front_matches_oracle(hip_api, oracle_api, nx, ny, checkpoints):
    """(i) dam-break, fp64: all four state fields == oracle at checkpoints before, inside and after the
    front; the equal-cost plan was in use on the way; switching it off gives the same numbers."""
    a = engine(hip_api, nx, ny, "f64", "f32", ic=1)
    a0 = engine(hip_api, nx, ny, "f64", "f32", ic=1)
    a0.set_param("jacobi_tb_adapt", 0)
    b = engine(oracle_api, nx, ny, "f64", "f32", ic=1)
    active, tiny = [], []
    for st in checkpoints:
        while a.istep < st:
            a.step(1)
            active.append(a.get_counter("tb_plan_active"))
        a0.step(st - a0.istep)
        b.step(st - b.istep)
        assert_fields_same(a, b, STATE, ctx="%dx%d plan on vs oracle, step %d" % (nx, ny, st))
        assert_fields_same(a0, b, STATE, ctx="%dx%d plan off vs oracle, step %d" % (nx, ny, st))
        assert a.get_counter("courant_violations") == b.get_counter("courant_violations")
        tiny.append(_tiny_cells(b.get("p")))
    assert a0.get_counter("tb_plan_active") == 0
    assert max(tiny) > 1000, tiny                   # the front really was there ...
    assert sum(1 for x in active if x) >= 20, (sum(active), len(active))   # ... and the plan really ran


@pytest.mark.parametrize("n,steps,checks,dt", [
    (4096, 250, (70, 130, 250), None),              # BASELINE configs[2]: 35 tile columns
    (8192, 120, (60, 120), 1e-6),                   # configs[3] on one GPU: 70 tile columns (dt: DESIGN section 4)
])
def test_plan_on_equals_plan_off_at_baseline_sizes(hip_api, n, steps, checks, dt):
    """(ii) the same run with and without the work plan, value for value, at the sizes the bench runs."""
    kw = {} if dt is None else {"dt": dt}
    on = engine(hip_api, n, n, "f64", "f32", ic=1, **kw)
    off = engine(hip_api, n, n, "f64", "f32", ic=1, **kw)
    off.set_param("jacobi_tb_adapt", 0)
    off.set_param("overlap_halves", 0)               # (`on` runs the default at these sizes -- by the rule of vof_step, k_tm +
    off.set_param("fuse_tm", 0)                      #  k_jacobi_pair on a dam-break --, `off` the plain four-kernel sequence)
    active = 0
    for st in checks:
        while on.istep < st:
            k = min(10, st - on.istep)
            on.step(k)
            active += on.get_counter("tb_plan_active")
        off.step(st - off.istep)
        for f in STATE:                              # one field at a time: 537 MB each at 8192^2
            x, y = on.get(f), off.get(f)
            assert same(x, y), "step %d: %s" % (st, diff_report(x, y, f))
            if f == "p":
                assert _tiny_cells(x) > 10000, "no tiny-value front at step %d" % st
            del x, y
    assert active >= len(checks) and off.get_counter("tb_plan_active") == 0
    assert on.get_counter("tm_choice") == 1 and on.get_param("gas_share") > 0.8, (on.get_counter("tm_choice"), on.get_param("gas_share"))
    assert on.get_counter("tm_steps") >= steps - 20 and on.get_counter("pair_launches") >= steps - 20 and on.get_counter("halves_steps") == 0
    assert off.get_counter("halves_steps") + off.get_counter("tm_steps") == 0
    F = on.get("F")
    assert F.min() >= 0.0 and F.max() <= 1.0 and on.get_counter("courant_violations") == 0


@pytest.mark.parametrize("nstrips", [2, 8])
def test_strips_inside_the_front(hip_api, nstrips):
    """(iii) 2048^2 fp64 dam-break: strips with VOF_HALO_ROWS-deep halos refreshed once per step
    (device copies stand in for RCCL send/recv) equal the single domain on their owned rows while the
    tiny-value front crosses them; each strip plans its own chunks from its own reports."""
    from vof2d.strips import partition, stored_rows
    n, W = 2048, halo_rows(10)
    full = engine(hip_api, n, n, "f64", "f32", ic=1)
    owns = partition(n, nstrips)
    strips = [engine(hip_api, n, n, "f64", "f32", ic=1, rows=stored_rows(n, o, W), own=o) for o in owns]
    planned = [0] * nstrips
    for step in range(1, 121):
        full.step(1)
        for s in strips:
            s.step(1)
        for k in range(nstrips - 1):
            lo_s, hi_s = strips[k], strips[k + 1]
            edge = owns[k][1]
            for f in STATE:
                lo_s.copy_rows_from(hi_s, f, edge + 1, edge + W)
                hi_s.copy_rows_from(lo_s, f, edge + 1 - W, edge)
        for k, s in enumerate(strips):
            planned[k] += s.get_counter("tb_plan_active")
        if step in (50, 80, 120):
            assert _tiny_cells(full.get("p")) > 10000
            for k, s in enumerate(strips):
                g0 = 0 if k == 0 else owns[k][0]
                g1 = n + 1 if k == nstrips - 1 else owns[k][1]
                assert_fields_same(s, full, STATE, rows=(g0, g1), ctx="step %d strip %d of %d" % (step, k, nstrips))
    assert full.get_counter("tb_plan_active") == 1
    assert sum(1 for x in planned if x > 10) >= nstrips // 2, planned     # most strips met the front and planned
    assert sum(s.get_counter("courant_violations") for s in strips) == full.get_counter("courant_violations")


def test_batch_forms_are_timed_again_and_switching_changes_no_value(hip_api):
    """fuse_tm = -2 (exploration; the default, -1, chooses by a rule on the state) on a large fp64 grid: the handle alternates its two batch forms over four 8-step
    batches, keeps the faster, and does so again every `tune_period` batches.  With a period of 2 batches (of 16 steps) a run of 250
    steps goes through the timing three to four times: both forms run for dozens of steps each, in turns -- and the state
    equals the plain four-kernel sequence's value for value (4096^2 dam-break, the headline configuration)."""
    n = 4096
    a = engine(hip_api, n, n, "f64", "f32", ic=1)
    a.set_param("fuse_tm", -2)
    a.set_param("tune_period", 2)
    b = engine(hip_api, n, n, "f64", "f32", ic=1)
    b.set_param("overlap_halves", 0)
    b.set_param("fuse_tm", 0)
    for st in (90, 250):
        a.step(st - a.istep)
        b.step(st - b.istep)
        for f in STATE:
            x, y = a.get(f), b.get(f)
            assert same(x, y), "step %d: %s" % (st, diff_report(x, y, f))
            del x, y
    tm, ch = a.get_counter("tm_steps"), a.get_counter("halves_steps")
    assert tm >= 3 * 16 and ch >= 3 * 16 and tm + ch >= 230, (tm, ch)
    assert a.get_counter("tm_choice") in (0, 1) and b.get_counter("tm_steps") + b.get_counter("halves_steps") == 0


def test_batch_form_follows_a_rule_on_the_state(hip_api):
    """The default (fuse_tm = -1): which batch form a large full domain runs is a function of the state -- the share
    of exact-zero cells of F when the handle first batches steps (counted when set_init_F replaces F) -- not of a stopwatch: two
    fresh handles agree, a dam-break (5/6 gas) runs k_tm + k_jacobi_pair, a rising bubble (2 % gas) the chains -- up to 16 M
    cells; beyond, the pair kernels whatever the grid holds (round 6: 4096^2 included) --, in fp32 the pair kernels also where the
    grid is too small for the chains (2048^2 bubble fp32 = BASELINE configs[4]), and where the rule does not apply (small grids)
    the counter says so.  Replacing F makes the handle look again."""
    n = 4096
    a = engine(hip_api, n, n, "f64", "f32", ic=1)
    b = engine(hip_api, n, n, "f64", "f32", ic=1)
    for e in (a, b):
        assert e.get_counter("tm_choice") == -1 and e.get_param("gas_share") == -1.0
        e.step(40)
    assert a.get_counter("tm_choice") == b.get_counter("tm_choice") == 1
    assert a.get_param("gas_share") == b.get_param("gas_share") and 0.8 < a.get_param("gas_share") < 0.85
    assert a.get_counter("tm_steps") == b.get_counter("tm_steps") >= 32 and a.get_counter("halves_steps") == 0
    b.close()
    t0 = a.get_counter("tm_steps")
    a.set_init_F(2)                                   # the same handle, now a bubble: the rule looks at the new F -- 16.8 M cells: still the pairs
    a.step(40)
    assert a.get_counter("tm_choice") == 1 and a.get_param("gas_share") < 0.05 and a.get_counter("tm_steps") >= t0 + 32
    a.close()
    a = engine(hip_api, 3072, 3072, "f64", "f32", ic=1)   # 9.4 M cells: the gas share decides
    a.step(40)
    assert a.get_counter("tm_choice") == 1 and a.get_counter("tm_steps") >= 32
    a.set_init_F(2)
    a.step(40)
    assert a.get_counter("tm_choice") == 0 and a.get_param("gas_share") < 0.05 and a.get_counter("halves_steps") >= 16
    a.close()
    for dtype, want in (("f32", 1), ("f64", 0)):      # 4.2 M cells, bubble: too small for the chains -- fp32 runs the pairs, fp64 the plain sequence
        g = engine(hip_api, 2048, 2048, dtype, "f32", ic=2)
        g.step(40)
        assert g.get_counter("tm_choice") == want and (g.get_counter("tm_steps") >= 32) == bool(want) and g.get_counter("halves_steps") == 0, dtype
        g.close()
    g = engine(hip_api, 5120, 5120, "f64", "f32", ic=2)   # from 16 M cells on the pair kernels win whatever the grid holds
    g.step(36)
    assert g.get_counter("tm_choice") == 1 and g.get_param("gas_share") < 0.05 and g.get_counter("tm_steps") >= 32
    g.close()
    # ... in fp32 too, value for value what the chains give
    g = engine(hip_api, 5120, 5120, "f32", "f32", ic=1)
    k = engine(hip_api, 5120, 5120, "f32", "f32", ic=1)
    k.set_param("fuse_tm", 0)
    for e in (g, k):
        e.step(44)
    assert g.get_counter("tm_choice") == 1 and g.get_counter("tm_steps") >= 40 and g.get_counter("pair_launches") >= 40
    assert k.get_counter("tm_steps") == 0 and k.get_counter("halves_steps") >= 32
    for f in STATE:
        x, y = g.get(f), k.get(f)
        assert same(x, y), "5120^2 fp32, pair kernels by the rule against the chains: %s" % diff_report(x, y, f)
        del x, y
    g.close(); k.close()
    c = engine(hip_api, n, n, "f32", "f32", ic=1)          # (the same rule in fp32)
    c.step(40)
    assert c.get_counter("tm_choice") == 1 and c.get_counter("tm_steps") >= 32
    c.close()
    c = engine(hip_api, 1536, 1536, "f32", "f32", ic=1)
    d = engine(hip_api, 1536, 1536, "f64", "f32", ic=1)   # (below 4 M cells neither form pays)
    for e in (c, d):
        e.step(20)
        assert e.get_counter("tm_choice") == -1 and e.get_counter("tm_steps") == 0


@pytest.mark.parametrize("nstrips", [2, 8])
def test_strips_run_the_pair_kernels_inside_the_front(hip_api, nstrips):
    """Overlap mode 5 of vof_step_exchange, piece by piece (vof_step_tm_piece) with device copies standing in for the
    send / recv groups: the strips run k_jacobi_pair and k_tm (this step's transport + the next step's momentum), the
    step boundary sits behind the momentum predictor and ONE exchange per step carries F, u*, v*, rhs, p,
    VOF_HALO_ROWS deep.  2048^2 fp64 dam-break, calls of 7, 12 and 1 steps, checked against the single domain on the owned
    rows while the tiny-value front crosses the strips (each strip plans its pairs' chunks from its own reports)."""
    from vof2d.strips import partition, stored_rows
    n, W = 2048, halo_rows(10)
    full = engine(hip_api, n, n, "f64", "f32", ic=1)
    owns = partition(n, nstrips)
    strips = [engine(hip_api, n, n, "f64", "f32", ic=1, rows=stored_rows(n, o, W), own=o) for o in owns]

    def trade(fields):
        for k in range(nstrips - 1):
            lo_s, hi_s = strips[k], strips[k + 1]
            edge = owns[k][1]
            for f in fields:
                lo_s.copy_rows_from(hi_s, f, edge + 1, edge + W)
                hi_s.copy_rows_from(lo_s, f, edge + 1 - W, edge)

    full.step(1)
    for s in strips:
        s.step(1)                                    # (the first step after set_init_F: the reference's set_BC calls)
    trade(STATE)
    planned = [0] * nstrips
    done = 1
    for call in (7, 12, 30, 1, 29, 40):
        full.step(call)
        for s in strips:
            s.step_tm_piece(0)
        trade(("u_star", "v_star", "rhs"))
        for _ in range(call - 1):
            for s in strips:
                s.step_tm_piece(1)
            trade(("F", "u_star", "v_star", "rhs", "p"))
            for k, s in enumerate(strips):
                planned[k] += s.get_counter("tb_plan_active")
        for s in strips:
            s.step_tm_piece(2)
        trade(STATE)
        done += call
        assert all(s.istep == done for s in strips) and full.istep == done
        if done >= 50:
            assert _tiny_cells(full.get("p")) > 10000
        for k, s in enumerate(strips):
            g0 = 0 if k == 0 else owns[k][0]
            g1 = n + 1 if k == nstrips - 1 else owns[k][1]
            assert_fields_same(s, full, STATE, rows=(g0, g1), ctx="step %d strip %d of %d (pair kernels)" % (done, k, nstrips))
    assert done == 120 and sum(1 for x in planned if x > 10) >= nstrips // 2, planned     # most strips met the front and planned their pairs
    assert sum(s.get_counter("courant_violations") for s in strips) == full.get_counter("courant_violations")
