"""Shared helpers of the parity tests."""
import numpy as np

import re

RTOL = 1e-4   # north_star: 1e-4 relative float tolerance
# The absolute floor, as a fraction of the SAME component's scale (see assert_close), by what is compared (r05, from the census below:
# profiles/r05_tolerance_census.txt, 232 M elements of one GPU session):
#   velocities (labels "plane 1" / "velocity"): 1e-5.  v + a with a ~ -v ends near zero and carries the absolute error of its terms; the
#       session's 1 180 such elements needed at most 3.3e-6 of the component's scale, the 24 000-seed sweep of r04 two elements more.
#   everything else (positions, render colour / data, lightmaps, probes, G-buffers): 1e-7 -- five position elements of the session
#       needed a floor at all, the largest 8.7e-9; no lightmap element ever did: for them the criterion is the north star's pure 1e-4.
ATOL = 1e-5
ATOL_TIGHT = 1e-7

# Census of what the criterion was asked to forgive (VERDICT r04 #5b): per kind of comparison (the `what` label with its numbers
# blanked) and per component, how many elements were compared, how many of them are outside a PURE 1e-4 relative bound -- i.e. needed
# the absolute floor -- and the smallest floor (as a fraction of the component's scale) that would still have passed.  Written at the
# end of a session by tests/conftest.py (profiles/r05_tolerance_census.txt is one such run on the GPU box).
CENSUS = {}


def _census_add(what, err, want, scale, rtol, both_nan):
    # (numbers blanked -- seeds, chunk indices, sizes -- except the plane index, which is what the census is by)
    key = re.sub(r"\d+(\.\d+)?", "#", re.sub(r"plane (\d)", lambda m: "plane " + "abcdefghij"[int(m.group(1))], what)).strip() or "(unlabelled)"
    key = re.sub(r"plane ([a-j])\b", lambda m: "plane %d" % "abcdefghij".index(m.group(1)), key)
    scale = np.broadcast_to(np.asarray(scale, np.float64), want.shape[-1:] if (want.ndim >= 2 and want.shape[-1] <= 4) else ())
    comps = want.shape[-1] if (want.ndim >= 2 and want.shape[-1] <= 4) else 1
    e2 = err.reshape(-1, comps)
    w2 = np.abs(want).reshape(-1, comps)
    ok2 = ~both_nan.reshape(-1, comps) & np.isfinite(e2)
    row = CENSUS.setdefault(key, {"calls": 0, "components": comps, "elements": np.zeros(comps, np.int64), "needed_floor": np.zeros(comps, np.int64),
                                  "least_floor": np.zeros(comps, np.float64), "worst_pure_relative": np.zeros(comps, np.float64)})
    if row["components"] != comps:
        return
    row["calls"] += 1
    over = np.where(ok2, e2 - rtol * w2, 0.0)                       # what the relative term leaves uncovered
    sc = np.maximum(np.asarray(scale, np.float64).reshape(-1) if comps > 1 else np.full(1, float(np.max(scale)) if np.size(scale) else 0.0), 1e-300)
    row["elements"] += ok2.sum(axis=0)
    row["needed_floor"] += (over > 0).sum(axis=0)
    row["least_floor"] = np.maximum(row["least_floor"], (np.maximum(over, 0.0) / sc).max(axis=0) if e2.size else 0.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = np.where(ok2 & (w2 > 0), e2 / w2, 0.0)
    row["worst_pure_relative"] = np.maximum(row["worst_pure_relative"], rel.max(axis=0) if e2.size else 0.0)


def census_report():
    lines = ["# tolerance census: criterion |got - want| <= %g |want| + atol x (largest |want| of the same component); atol = %g for velocities, %g for everything else, unless a test passes its own" % (RTOL, ATOL, ATOL_TIGHT),
             "# per kind of comparison and component: elements compared | outside a PURE %g relative bound (needed the floor) | smallest floor that passes (fraction of the component's scale) | worst pure relative error" % RTOL]
    tot_e = tot_f = 0
    for key in sorted(CENSUS):
        r = CENSUS[key]
        lines.append("%s   (%d call(s))" % (key, r["calls"]))
        for c in range(r["components"]):
            lines.append("    component %d: %12d | %10d (%.2e) | %.3e | %.3e" % (c, r["elements"][c], r["needed_floor"][c], r["needed_floor"][c] / max(r["elements"][c], 1),
                                                                              r["least_floor"][c], r["worst_pure_relative"][c]))
            tot_e += int(r["elements"][c]); tot_f += int(r["needed_floor"][c])
    lines.append("# total: %d elements, %d needed the floor (%.2e)" % (tot_e, tot_f, tot_f / max(tot_e, 1)))
    return "\n".join(lines)


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


def default_atol(what):
    return ATOL if re.search(r"plane 1\b|velocity", what) else ATOL_TIGHT


def assert_close(got, want, what="", rtol=RTOL, atol=None, scale=None, life_exact=False):
    """|got - want| <= rtol * |want| + atol * scale_c for every element, where scale_c is the largest finite |want| of the element's
    OWN component (index on the last axis when that axis has <= 4 entries: x, y, z, life / category / alpha are priced separately --
    a position plane's life is not allowed the slack of its x coordinates).  The floor exists because sums cancel: a velocity
    component v + a that ends near zero carries the absolute error of its terms.
    life_exact: the array is a PositionAndLife plane and its .w must equal the oracle's bit for bit (liveness never drifts)."""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if atol is None:
        atol = default_atol(what)
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
    try:
        _census_add(what, err, want, scale, rtol, both_nan)
    except Exception:        # (the census must never fail a test)
        pass
    tol = atol * np.asarray(scale, np.float64) + rtol * np.abs(want)
    bad = ~(err <= tol) & ~both_nan
    if bad.any():
        idx = np.argwhere(bad)[:8]
        msg = ["%s: %d / %d elements out of tolerance (component scales %s)" % (what, int(bad.sum()), bad.size, np.asarray(scale))]
        for i in idx:
            t = tuple(i)
            msg.append("  at %s got %.9g want %.9g" % (t, got[t], want[t]))
        raise AssertionError("\n".join(msg))
