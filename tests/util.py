"""Shared helpers of the parity tests."""
import numpy as np

RTOL = 1e-4   # north_star: 1e-4 relative float tolerance
ATOL = 1e-5   # absolute floor, as a fraction of the SAME component's scale (see assert_close)


def assert_bits_equal(got, want, what=""):
    """Bit-for-bit equality of float32 arrays (any NaN matches any NaN): what DESIGN.md claims for life values, table indices, tap values."""
    got = np.ascontiguousarray(got, np.float32)
    want = np.ascontiguousarray(want, np.float32)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    if not same.all():
        idx = np.argwhere(~same)[:8]
        msg = ["%s: %d / %d elements are not bit-identical" % (what, int((~same).sum()), same.size)]
        for i in idx:
            t = tuple(i)
            msg.append("  at %s got %.9g (0x%08x) want %.9g (0x%08x)" % (t, got[t], got.view(np.uint32)[t], want[t], want.view(np.uint32)[t]))
        raise AssertionError("\n".join(msg))


def assert_close(got, want, what="", rtol=RTOL, atol=ATOL, scale=None, life_exact=False):
    """|got - want| <= rtol * |want| + atol * scale_c for every element, where scale_c is the largest finite |want| of the element's
    OWN component (index on the last axis when that axis has <= 4 entries: x, y, z, life / category / alpha are priced separately --
    a position plane's life is not allowed the slack of its x coordinates).  The floor exists because sums cancel: a velocity
    component v + a that ends near zero carries the absolute error of its terms.
    life_exact: the array is a PositionAndLife plane and its .w must equal the oracle's bit for bit (liveness never drifts)."""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if life_exact:
        assert_bits_equal(np.asarray(got, np.float32)[..., 3], np.asarray(want, np.float32)[..., 3], what + " life")
    if scale is None:
        finite = np.where(np.isfinite(want), np.abs(want), 0.0)
        if want.ndim >= 2 and want.shape[-1] <= 4:
            scale = finite.reshape(-1, want.shape[-1]).max(axis=0) if finite.size else np.zeros(want.shape[-1])
        else:
            scale = float(finite.max()) if finite.size else 0.0
    both_nan = np.isnan(got) & np.isnan(want)
    err = np.abs(got - want)
    tol = atol * np.asarray(scale, np.float64) + rtol * np.abs(want)
    bad = ~(err <= tol) & ~both_nan
    if bad.any():
        idx = np.argwhere(bad)[:8]
        msg = ["%s: %d / %d elements out of tolerance (component scales %s)" % (what, int(bad.sum()), bad.size, np.asarray(scale))]
        for i in idx:
            t = tuple(i)
            msg.append("  at %s got %.9g want %.9g" % (t, got[t], want[t]))
        raise AssertionError("\n".join(msg))
