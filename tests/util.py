"""Shared helpers of the parity tests."""
import numpy as np

RTOL = 1e-4   # north_star: 1e-4 relative float tolerance
ATOL = 1e-5


def assert_close(got, want, what="", rtol=RTOL, atol=ATOL, scale=None):
    """Relative tolerance of the north star, with an absolute floor proportional to the data scale."""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if scale is None:
        finite = np.isfinite(want)
        scale = max(1.0, float(np.max(np.abs(want[finite]))) if finite.any() else 1.0)
    both_nan = np.isnan(got) & np.isnan(want)
    err = np.abs(got - want)
    tol = atol * scale + rtol * np.abs(want)
    bad = ~(err <= tol) & ~both_nan
    if bad.any():
        idx = np.argwhere(bad)[:8]
        msg = ["%s: %d / %d elements out of tolerance" % (what, int(bad.sum()), bad.size)]
        for i in idx:
            t = tuple(i)
            msg.append("  at %s got %.9g want %.9g" % (t, got[t], want[t]))
        raise AssertionError("\n".join(msg))
