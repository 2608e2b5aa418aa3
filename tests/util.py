"""Shared helpers of the parity tests."""
import numpy as np

from vof2d.engine import Engine, make_desc

STATE = ("F", "u", "v", "p")


def engine(api, nx, ny, dtype="f64", coord_cast="f32", ic=None, **kw):
    e = Engine(api, make_desc(api, nx, ny, dtype, coord_cast, **kw))
    if ic is not None:
        e.set_init_F(ic)
    return e


def same(a, b):
    """Value-for-value equality (IEEE ==; the sign of an exact zero is not compared)."""
    return a.shape == b.shape and a.dtype == b.dtype and bool(np.array_equal(a, b))


def diff_report(a, b, name=""):
    bad = np.argwhere(a != b)
    if len(bad) == 0:
        return "%s equal" % name
    i, j = bad[0]
    return "%s: %d cells differ, max|d|=%.3e, first at [%d,%d]: %r vs %r; rows %d..%d cols %d..%d" % (
        name, len(bad), float(np.nanmax(np.abs(a.astype(np.float64) - b.astype(np.float64)))), i, j, a[i, j],
        b[i, j], bad[:, 0].min(), bad[:, 0].max(), bad[:, 1].min(), bad[:, 1].max())


def assert_fields_same(ea, eb, names=STATE, rows=None, ctx=""):
    msgs = []
    for n in names:
        a, b = ea.get(n, rows), eb.get(n, rows)
        if not same(a, b):
            msgs.append(diff_report(a, b, n))
    assert not msgs, ctx + " | " + " ; ".join(msgs)
