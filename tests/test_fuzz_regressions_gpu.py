"""The particle steps of the 24 000-seed sweeps that fell outside the suite's own float criterion -- r04's two
(profiles/r04_fuzz_24000_seeds_final_kernels.txt: seeds 1222260 and 1223153, V.z of one newborn particle each, 7e-4 and 1.7e-4 relative) and
r05's one (profiles/r05_fuzz_24000_seeds_head.txt: seed 1405703, V.y of a newborn particle, 9e-4 relative; d^2 - radius = 0.19 of d^2 = 288),
r06's two (profiles/r06_fuzz_33381_seeds_6500000.txt: seed 6533961, V.y and V.z of a newborn particle, 1.1e-2 and 2e-4 relative;
profiles/r06_fuzz_32000_seeds_6700000_8_processes.txt: seed 6726686, V.x, 2.5e-4 relative, d^2 - radius = 1.6 of d^2 = 146; the same family) --
as named tests that assert what is actually true of them:

  * the step's integers are exact: live counts, the liveness of every slot, every life value bit for bit;
  * every float of the step except that one velocity component meets the criterion (1e-4 relative + 1e-5 of the component's scale);
  * the newborn particle's POSITION and velocity after spawn + Update alone are within 2 ulp of the oracle's (3 for the last seed's V.z:
    OCML's sin / cos / acos against glibc's in the spawn formula, SpawnerCommon.fxh:47-57);
  * its velocity lies inside the envelope the ORACLE ITSELF produces when its own post-spawn position is moved by +-2 (3) ulp per coordinate:
    the step has an attractor of the physical type whose radius all but cancels the particle's squared distance
    (Gravity.fx:44-47: strength / max(d^2 - radius, 0.001)), so the reference's own formula amplifies the last bit of a coordinate;
  * from the oracle's OWN post-spawn state (same bits in) the shipped kernel meets the criterion on every element, and the
    -DILM_GRAVITY_EXACT build (Gravity's IEEE sqrt / division form) reproduces the oracle's Gravity pass bit for bit, on that slot and
    on every live slot of the chunk.
"""
import copy
import itertools
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from illuminant_amd import abi
from tests import _variant_worker as vw
from tests import fuzz_scenes
from tests.util import ATOL, ATOL_TIGHT, RTOL, assert_bits_equal, assert_close

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXACT_LIB = os.path.join(ROOT, "illuminant_amd", "lib", "libilluminant_hip_gravity_exact.so")

# (seed, slot of chunk 1, velocity component(s) -- the first is the one the cancellation amplifies most --, the physical attractor in front of
# the cancellation: position, radius, strength, and how many ulp the spawn formula + Update alone leave between device and oracle: 2, or
# 3 for the last seed's V.z)
CASES = [(1222260, 4913, (2,), (154.30, 147.45, 0.85), 280.08, 150.59, 2),
         (1223153, 2180, (2,), (120.94, 116.24, 12.75), 188.54, -12.33, 2),
         (1405703, 12779, (1,), (67.33, 170.43, 16.95), 287.52, 43.68, 2),
         (6533961, 2231, (1, 2, 0), (153.23, 120.27, 4.55), 231.17, -38.28, 2),
         (6726686, 1194, (0,), (106.13, 145.52, 1.90), 144.15, -23.01, 3)]


def outside_criterion(got, want, plane):
    """element mask of the suite's criterion (tests/util.py assert_close), per-component scale; the floor is the velocity plane's or the tight one"""
    g, w = np.asarray(got, np.float64), np.asarray(want, np.float64)
    scale = np.where(np.isfinite(w), np.abs(w), 0.0).max(axis=0)
    atol = ATOL if plane == 1 else ATOL_TIGHT
    return ~(np.abs(g - w) <= atol * scale[None, :] + RTOL * np.abs(w)) & ~(np.isnan(g) & np.isnan(w))


@pytest.mark.parametrize("seed,slot,comps,apos,aradius,astrength,ulps", CASES)
def test_fuzz_miss_is_an_ulp_of_the_spawn_formula_in_front_of_the_references_own_cancellation(ctx, oracle, seed, slot, comps, apos, aradius, astrength, ulps):
    comp = comps[0]
    cs, rnd, chunks, d = fuzz_scenes.particle_step_of_seed(seed)
    # the replay IS the step the sweep reported: its spawn range holds the slot, its op list starts with a Gravity that has that attractor
    sp = d.Spawns[0].Params
    assert d.SpawnCount == 1 and d.OpCount >= 1 and d.Ops[0].Type == abi.OP_GRAVITY
    assert sp.ChunkSizeAndIndices[1] <= slot <= sp.ChunkSizeAndIndices[2]
    g = d.Ops[0].u.Gravity
    att = [k for k in range(int(g.AttractorCount)) if g.AttractorRadiusesAndStrengths[k][2] < 0.5
           and np.allclose(list(g.AttractorPositions[k])[:3], apos, atol=0.01) and abs(g.AttractorRadiusesAndStrengths[k][0] - aradius) < 0.01
           and abs(g.AttractorRadiusesAndStrengths[k][1] - astrength) < 0.01]
    assert len(att) == 1

    initial = [[a.copy() for a in c] for c in chunks]
    got, got_counts = vw.device_step(ctx, cs, rnd, initial, d)
    want = [[a.copy() for a in c] for c in chunks]
    want_counts = oracle.step(want, cs, rnd, d, want_counts=True)

    # integers: exact
    assert np.array_equal(got_counts, want_counts)
    for c in range(2):
        assert np.array_equal(got[c][0][:, 3] > 0, want[c][0][:, 3] > 0)
        assert_bits_equal(got[c][0][:, 3], want[c][0][:, 3], "seed %d chunk %d life" % (seed, c))
    # floats: the criterion everywhere but that velocity component of the one slot (if OCML / glibc ever agree on it, nothing is outside at all)
    outside = set()
    for c in range(2):
        for k in range(5):
            for (i, j) in np.argwhere(outside_criterion(got[c][k], want[c][k], k)):
                outside.add((c, k, int(i), int(j)))
    assert outside <= {(1, 1, slot, j) for j in comps}, outside

    # the newborn particle as the spawn formula and Update alone leave it (the same step without its Gravity op): position and velocity
    # within 2 ulp of the oracle's -- OCML's sin / cos / acos against glibc's
    d_plain = copy.copy(d); d_plain.OpCount = 0
    got_plain, _ = vw.device_step(ctx, cs, rnd, [[a.copy() for a in c] for c in chunks], d_plain)
    want_plain = [[a.copy() for a in c] for c in chunks]
    oracle.step(want_plain, cs, rnd, d_plain)
    for plane in (0, 1):
        gp, wp = got_plain[1][plane][slot, :3], want_plain[1][plane][slot, :3]
        assert (np.abs(gp.astype(np.float64) - wp.astype(np.float64)) <= float(ulps) * np.spacing(np.abs(wp)).astype(np.float64)).all(), (plane, gp, wp)
    # (in the full step its position then carries the velocity's difference x dt on top of that)
    dt = float(d.System.GlobalSettings.x) / 1000.0
    gp, wp = got[1][0][slot, :3].astype(np.float64), want[1][0][slot, :3].astype(np.float64)
    dv = np.abs(got[1][1][slot, :3].astype(np.float64) - want[1][1][slot, :3].astype(np.float64))
    assert (np.abs(gp - wp) <= (ulps + 1.0) * np.spacing(np.abs(want[1][0][slot, :3])).astype(np.float64) + dv * dt * 1.01).all(), (gp, wp, dv * dt)
    # ... in front of the cancellation the note describes: d^2 - radius is a small fraction of d^2
    cs_, rnd_, spawned, d0 = vw.post_spawn_state(seed)
    p0 = spawned[1][0][slot, :3].astype(np.float64)
    d2 = float(((np.asarray(apos) - p0) ** 2).sum())
    assert 0.0 < d2 - aradius < 0.12 * d2, (d2, aradius)
    # ... and its velocity inside the envelope of the oracle's own answers for +-2 ulp of its post-spawn position
    lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
    for signs in itertools.product(tuple(range(-ulps, ulps + 1)), repeat=3):
        trial = [[a.copy() for a in c] for c in spawned]
        for axis, s in enumerate(signs):
            trial[1][0][slot, axis] += np.float32(s) * np.spacing(np.abs(trial[1][0][slot, axis]))
        one = [trial[1]]                                # chunk 1 alone (chunks never interact)
        d1 = copy.copy(d0); d1.FirstChunk, d1.ChunkCount = 0, -1
        oracle.step(one, cs, rnd, d1)
        v = one[0][1][slot, :3].astype(np.float64)
        lo, hi = np.minimum(lo, v), np.maximum(hi, v)
    gv = got[1][1][slot, :3].astype(np.float64)
    slack = RTOL * np.abs(want[1][1][slot, :3].astype(np.float64)) + 1e-6
    assert ((gv >= lo - slack) & (gv <= hi + slack)).all(), (gv, lo, hi)
    assert hi[comp] - lo[comp] > 1e-4 * abs(float(want[1][1][slot, comp])), "the envelope shows the amplification: +-2 ulp of position moves the component by more than 1e-4 relative"

    # same bits in -> the shipped kernel meets the criterion on every element
    got0, _ = vw.device_step(ctx, cs, rnd, spawned, d0)
    want0 = [[a.copy() for a in c] for c in spawned]
    oracle.step(want0, cs, rnd, d0)
    for c in range(2):
        m = want0[c][0][:, 3] > 0
        assert np.array_equal(got0[c][0][:, 3] > 0, m)
        for k in (0, 1, 3, 4):
            assert_close(got0[c][k][m], want0[c][k][m], "seed %d from the oracle's post-spawn state: chunk %d plane %d" % (seed, c, k), life_exact=(k == 0))


@pytest.mark.parametrize("seed,slot", [(c[0], c[1]) for c in CASES])
def test_exact_gravity_build_reproduces_the_oracle_from_the_oracles_inputs(seed, slot):
    """The -DILM_GRAVITY_EXACT variant (built beside the shipped library by csrc/Makefile) in a process of its own: the step's Gravity op
    alone (ILM_UPDATE_NONE) on the oracle's post-spawn state."""
    assert os.path.exists(EXACT_LIB), "build the variant: make -C illuminant_amd/csrc (the Makefile's default goal builds both)"
    env = dict(os.environ, ILM_HIP_LIB=EXACT_LIB)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_variant_worker.py"), str(seed), str(slot)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["lib"] == "libilluminant_hip_gravity_exact.so"
    assert out["slot_velocity_bits_equal"], out
    assert out["chunk_velocity_elements_differing"] == 0, out
