#!/usr/bin/env python3
"""Generates tests/golden/*.json -- the substitute known-answer vectors that pin the CPU oracle.

WHY "SUBSTITUTE".  sq/Illuminant ships no tests, golden vectors or fixtures for the two hot paths, its GPU
code is HLSL that needs fxc + Direct3D, and its host code needs .NET 4.8 + Fracture + XNA/FNA: none of that
can run in the build container, so the oracle (oracle/ilm_oracle.c, a restatement of the HLSL) cannot be checked
against outputs of the reference itself ("parity unpinned").  What the reference DOES contain is a
second, independent statement of some of the same arithmetic in C# -- CPU mirrors the engine uses on the host --
plus pure integer/layout logic and closed-form identities.  This script restates THOSE (not the HLSL) in plain
Python, one function per cited C# block, evaluates them on fixed inputs and writes inputs + expected outputs
as data.  tests/test_oracle_kat.py then requires the oracle (and, on the GPU box, the HIP path) to reproduce
them.  Two restatements written from two different sources agreeing is the strongest pin available here.

Everything is float32-faithful where the C# is float (numpy.float32 arithmetic, one rounding per operation).
All file:line citations are relative to the reference checkout.  The script needs nothing but numpy, does not
read /root/reference, and is deterministic: re-running it must reproduce the committed JSON byte for byte.
"""
import json
import math
import os

import numpy as np

F = np.float32
HERE = os.path.dirname(os.path.abspath(__file__))


def f32(x):
    return float(F(x))


# ---------------------------------------------------------------------------------------------------------
# 1. ClampedBezier1/4.Evaluate + tForScaledBezier -- Illuminant/Bezier.cs:461-490 (Bezier1), :759-832 (Bezier4)
#    (the C# CPU mirror of Bezier.fxh:21-177, which is what the oracle restates)
# ---------------------------------------------------------------------------------------------------------
def cs_lerp(a, b, t):
    # Arithmetic.Lerp (Fracture Squared.Util): a + (b - a) * t in float
    return F(a) + (F(b) - F(a)) * F(t)


def cs_saturate(t):
    return F(min(max(F(t), F(0)), F(1)))


def cs_wrap_exclusive(t, lo, hi):
    # Arithmetic.WrapExclusive for t >= lo: t - floor((t-lo)/(hi-lo))*(hi-lo).  Vectors below keep t >= 0, where it
    # coincides with HLSL's truncating `%` (Bezier.fxh:33-46); negative t is where the two differ and is left out.
    d = F(hi) - F(lo)
    return F(np.fmod(F(t) - F(lo), d)) + F(lo)


def t_for_scaled_bezier(range_and_count, value):
    min_value, inv_divisor, count, mode_f = [F(v) for v in range_and_count]
    mode = int(mode_f)
    repeating, bouncing = mode > 255, mode > 511
    t = (F(value) - min_value) * F(abs(inv_divisor))
    if bouncing:
        t = t * F(2)
        t = (F(2) - cs_wrap_exclusive(t, 0, 2)) if inv_divisor < 0 else cs_wrap_exclusive(t, 0, 2)
        if t > 1:
            t = F(1) - (t - F(1))
    elif repeating:
        t = (F(1) - cs_wrap_exclusive(t, 0, 1)) if inv_divisor < 0 else cs_wrap_exclusive(t, 0, 1)
    else:
        t = (F(1) - cs_saturate(t)) if inv_divisor < 0 else cs_saturate(t)
    m = mode % 256
    if m == 1:      # BezierTimeMode.Sine: (float)Math.Sin(t * Math.PI * 0.5) -- double sin, rounded once
        t = F(math.sin(float(t) * math.pi * 0.5))
    elif m == 2:    # BezierTimeMode.Exp
        t = t * t
    return int(count), F(t)


def bezier_evaluate(range_and_count, a, b, c, d, value):
    """a..d: scalars (ClampedBezier1) or 4-tuples (ClampedBezier4); returns a list of float."""
    a, b, c, d = [np.atleast_1d(np.asarray(v, dtype=F)) for v in (a, b, c, d)]
    count, t = t_for_scaled_bezier(range_and_count, value)
    if count <= 1.5:
        r = a
    else:
        ab = cs_lerp(a, b, t)
        if count <= 2.5:
            r = ab
        elif count <= 3.5:   # "HACK: Shelf mode"
            r = a if t <= 0 else (c if t >= 1 else b)
        else:
            bc = cs_lerp(b, c, t)
            abbc = cs_lerp(ab, bc, t)
            cd = cs_lerp(c, d, t)
            bccd = cs_lerp(bc, cd, t)
            r = cs_lerp(abbc, bccd, t)
    return [float(x) for x in np.asarray(r, dtype=F)]


def clamped_range(min_value, max_value, count, mode):
    # ClampedBezier1 ctor, Bezier.cs:444-459: (min(minValue, maxValue), 1/range, Count, Mode); range 0 or Count<=1 -> 1
    rng = F(max_value) - F(min_value)
    if rng == 0 or count <= 1:
        rng = F(1)
    return [f32(min(min_value, max_value)), f32(F(1) / rng), float(count), float(mode)]


def gen_bezier():
    cases = []
    values = [0.0, 0.125, 0.5, 0.75, 1.0, 1.5, 2.25, 3.0, 7.625]
    a1, b1, c1, d1 = 0.25, 1.5, -0.75, 2.0
    a4, b4, c4, d4 = (1.0, 0.5, 0.25, 0.0), (0.2, 0.9, 0.4, 1.0), (0.7, 0.1, 0.95, 0.5), (0.0, 0.3, 0.6, 1.0)
    for count in (1, 2, 3, 4):
        for mode in (0, 1, 2, 256, 257, 512, 514):
            for (lo, hi) in ((0.0, 1.0), (0.5, 3.0), (3.0, 0.5)):
                rc = clamped_range(lo, hi, count, mode)
                for v in values:
                    if v < min(lo, hi):
                        continue       # keep t >= 0 (see cs_wrap_exclusive)
                    cases.append({"kind": "bezier1", "range_and_count": rc, "abcd": [a1, b1, c1, d1], "value": v,
                                  "expected": bezier_evaluate(rc, a1, b1, c1, d1, v)})
                    cases.append({"kind": "bezier4", "range_and_count": rc, "a": list(a4), "b": list(b4), "c": list(c4),
                                  "d": list(d4), "value": v, "expected": bezier_evaluate(rc, a4, b4, c4, d4, v)})
    # the ColorFromLife curve SetSystemUniforms builds from OpacityFromLife = o (ParticleSystem.cs:554-563):
    # A = (1,1,1,0), B = 1, RangeAndCount = (0, 1/o, 2, 0)  =>  alpha ramps 0 -> 1 over life in [0, o]
    for o in (0.5, 2.5):
        rc = [0.0, f32(F(1) / F(o)), 2.0, 0.0]
        for v in (0.0, 0.1, 0.25, 1.0, 2.5, 40.0):
            cases.append({"kind": "bezier4", "range_and_count": rc, "a": [1, 1, 1, 0], "b": [1, 1, 1, 1], "c": [1, 1, 1, 1],
                          "d": [1, 1, 1, 1], "value": v, "expected": bezier_evaluate(rc, (1, 1, 1, 0), (1, 1, 1, 1), (1, 1, 1, 1), (1, 1, 1, 1), v)})
    return {"source": "Illuminant/Bezier.cs:444-490,759-832 (C# CPU mirror of Bezier.fxh:21-177)", "tolerance": "rtol 1e-6 (Sine mode: C# uses double sin)", "cases": cases}


# ---------------------------------------------------------------------------------------------------------
# 2. DistanceField constructor -- Illuminant/SDF/DistanceField.cs:43-109 (pure integer / double layout math)
# ---------------------------------------------------------------------------------------------------------
def cs_round(x):
    # Math.Round(double): round half to even -- Python's round() on a float does the same
    return float(round(x))


def distance_field_layout(vw, vh, requested_slices, requested_resolution):
    max_surface, packed = 8192, 3           # DistanceField.cs:19, LightingRenderer.PackedSliceCount
    rr = min(max(requested_resolution, 0.05), 1.0)
    cw, ch = int(cs_round(vw * rr)), int(cs_round(vh * rr))
    frac = (vw / cw + vh / ch) / 2
    res = round(1.0 / frac, 3)              # Math.Round(x, 3); the cases below are exact at 3 digits
    res = min(max(res, 0.05), 1.0)
    sw, sh = int(cs_round(vw * res)), int(cs_round(vh * res))
    mx, my = max_surface // sw, max_surface // sh
    max_slices = mx * my * packed
    sc = max(3, requested_slices)
    sc = ((sc + 2) // 3) * 3
    sc = min(sc, max_slices)
    phys = int(math.ceil(sc / packed))
    cols = min(mx, phys)
    rows = min(my, max(int(math.ceil(phys / mx)), 1))
    while rows < cols and rows < my:
        nr = rows + 1
        nc = int(math.ceil(phys / nr))
        nr = min(nr, mx)
        nc = min(nc, my)
        if nr * nc < phys:
            break
        rows, cols = nr, nc
    return {"resolution": res, "slice_width": sw, "slice_height": sh, "slice_count": sc, "physical_slice_count": phys,
            "column_count": cols, "row_count": rows, "atlas_width": sw * cols, "atlas_height": sh * rows}


def gen_layout():
    cases = []
    for (vw, vh, n, r) in ((512, 512, 32, 1.0),          # SURVEY 8c KAT: 33 slices, 11 physical, 3x4, 1536x2048
                           (1920, 1080, 9, 0.25),         # TestGame Scenes/SimpleParticles.cs:216-219: 480x270, 2x2
                           (256, 256, 9, 1.0),            # cfg1: 3 physical, 2x2, 512x512
                           (2048, 2048, 32, 0.25),        # cfg3 field: 512x512 slices, 1536x2048 atlas
                           (4096, 4096, 32, 0.125),       # cfg5 field
                           (128, 128, 6, 0.5), (1024, 768, 1, 1.0), (640, 360, 64, 0.5), (8192, 8192, 16, 1.0),
                           (300, 200, 24, 0.01), (1000, 1000, 7, 3.0)):
        cases.append({"virtual_width": vw, "virtual_height": vh, "requested_slice_count": n, "requested_resolution": r,
                      "expected": distance_field_layout(vw, vh, n, r)})
    return {"source": "Illuminant/SDF/DistanceField.cs:43-109", "tolerance": "exact (resolution: 1e-12)", "cases": cases}


# ---------------------------------------------------------------------------------------------------------
# 3. SpawnerBase.BeginTick / EndTick + RunSpawner / PickTargetForSpawn slot allocation --
#    Illuminant/Particles/ParticleSpawner.cs:152-194, ParticleSpawning.cs:115-231, ParticleSystem.cs:725-741
#    RNG draws are explicit inputs (the reference's Xoshiro is unseeded).
# ---------------------------------------------------------------------------------------------------------
def begin_tick(state, min_rate, max_rate, count_scale, draw, dt, maximum_total):
    min_rate, max_rate = f32(min_rate), f32(max_rate)
    if min_rate > max_rate:
        min_rate = max_rate
    current = ((draw * (max_rate - min_rate)) + min_rate) * count_scale * dt
    current += state["rate_error"]
    state["rate_error"] = 0.0
    if current < 1:
        state["rate_error"] = max(current, 0.0)
        count = 0
    else:
        count = int(current)
        state["rate_error"] = current - count
    if maximum_total is not None:
        remaining = maximum_total * count_scale - state["total_spawned"]
        if count > remaining:
            count = remaining
            state["rate_error"] = 0.0
    return count


def end_tick(state, requested, actual):
    state["rate_error"] += requested - actual
    state["total_spawned"] += actual


def run_system_spawns(chunk_capacity, ticks, min_rate, max_rate, count_scale, dt, maximum_total, partial_allowed=True):
    """One spawner, a fresh ParticleSystem; per tick returns the (chunk id, first, last) ranges RunSpawner issues
    (second pass included, ParticleSystem.cs:732-740)."""
    state = {"rate_error": 0.0, "total_spawned": 0}
    chunks = {}          # id -> NextSpawnOffset
    retired = set()
    target = -1
    next_id = 1
    trace = []
    for draws in ticks:
        issued = []
        passes = 0
        while passes < 2:
            draw = draws[passes]
            requested = begin_tick(state, min_rate, max_rate, count_scale, draw, dt, maximum_total)
            if requested <= 0:
                break
            count = min(requested, chunk_capacity)
            # PickTargetForSpawn, ParticleSpawning.cs:199-231
            if target != -1:
                free = chunk_capacity - chunks[target]
                if free < (16 if partial_allowed else count):
                    retired.add(target)
                    target = -1
            if target == -1:
                target = next_id
                next_id += 1
                chunks[target] = 0
            free = chunk_capacity - chunks[target]
            if count > free:
                if partial_allowed:
                    count = free
                else:
                    break
            first = chunks[target]
            last = first + count - 1
            chunks[target] += count
            end_tick(state, requested, count)
            issued.append([target, first, last])
            passes += 1
            if not (requested > count):      # isPartialSpawn
                break
        trace.append({"draws": list(draws), "issued": issued, "rate_error_after": state["rate_error"], "total_spawned_after": state["total_spawned"]})
    return trace


def gen_spawner():
    cases = []
    # (a) BeginTick traces: cfg2's spawner, MinRate = MaxRate = 65536/s at dt = 1/60 => 1092 / 1093 through the RateError carry
    st = {"rate_error": 0.0, "total_spawned": 0}
    seq = []
    for i in range(12):
        n = begin_tick(st, 65536.0, 65536.0, 1, 0.5, 1.0 / 60.0, None)
        end_tick(st, n, n)
        seq.append({"draw": 0.5, "count": n, "rate_error_after": st["rate_error"], "total_spawned_after": st["total_spawned"]})
    cases.append({"kind": "begin_tick", "min_rate": 65536.0, "max_rate": 65536.0, "count_scale": 1, "dt": 1.0 / 60.0, "maximum_total": None, "ticks": seq})
    # (b) random rate in [min, max], low rates accumulate through RateError, MaximumTotal clamp
    st = {"rate_error": 0.0, "total_spawned": 0}
    seq = []
    draws = [0.0, 0.999, 0.25, 0.5, 0.75, 0.125, 0.875, 0.0625, 0.3, 0.6, 0.9, 0.45, 0.05, 0.95]
    for dr in draws:
        n = begin_tick(st, 20.0, 240.0, 2, dr, 1.0 / 60.0, 25)
        end_tick(st, n, n)
        seq.append({"draw": dr, "count": n, "rate_error_after": st["rate_error"], "total_spawned_after": st["total_spawned"]})
    cases.append({"kind": "begin_tick", "min_rate": 20.0, "max_rate": 240.0, "count_scale": 2, "dt": 1.0 / 60.0, "maximum_total": 25, "ticks": seq})
    # (c) min > max is clamped to max (ParticleSpawner.cs:163-164)
    st = {"rate_error": 0.0, "total_spawned": 0}
    seq = []
    for dr in (0.1, 0.9, 0.5):
        n = begin_tick(st, 900.0, 300.0, 1, dr, 0.05, None)
        end_tick(st, n, n)
        seq.append({"draw": dr, "count": n, "rate_error_after": st["rate_error"], "total_spawned_after": st["total_spawned"]})
    cases.append({"kind": "begin_tick", "min_rate": 900.0, "max_rate": 300.0, "count_scale": 1, "dt": 0.05, "maximum_total": None, "ticks": seq})
    # (d) slot allocation across chunk roll-over: 64^2 chunks, 1500/tick => partial spawn + second pass + Free<16 rule.
    #     dt = 1/64 s is exact in binary, so a host that derives dt from clock differences (ParticleSystem.cs:635-669)
    #     reproduces it bit for bit.
    ticks = [(0.5, 0.5)] * 9
    cases.append({"kind": "allocation", "chunk_capacity": 4096, "min_rate": 96000.0, "max_rate": 96000.0, "count_scale": 1, "dt": 1.0 / 64.0,
                  "maximum_total": None, "trace": run_system_spawns(4096, ticks, 96000.0, 96000.0, 1, 1.0 / 64.0, None)})
    ticks = [(0.2, 0.7), (0.9, 0.1), (0.4, 0.4), (0.99, 0.01), (0.6, 0.3), (0.5, 0.5), (0.05, 0.95), (0.8, 0.2)]
    cases.append({"kind": "allocation", "chunk_capacity": 1024, "min_rate": 6000.0, "max_rate": 40000.0, "count_scale": 1, "dt": 1.0 / 64.0,
                  "maximum_total": None, "trace": run_system_spawns(1024, ticks, 6000.0, 40000.0, 1, 1.0 / 64.0, None)})
    return {"source": "Illuminant/Particles/ParticleSpawner.cs:152-194, ParticleSpawning.cs:115-231, ParticleSystem.cs:725-741",
            "tolerance": "counts / slot indices exact; rate_error 1e-9", "cases": cases}


# ---------------------------------------------------------------------------------------------------------
# 4. Liveness decode -- CountLiveParticles.fx:38 (+1/65535 per live particle into a unorm16 channel, additive blend,
#    saturating) and ProcessLivenessInfoData, ParticleEngine.cs:244-247 (count = raw & 0xFFFF)
# ---------------------------------------------------------------------------------------------------------
def gen_liveness():
    cases = []
    for live in (0, 1, 255, 4095, 4096, 65534, 65535, 65536, 70000, 1048576):
        unorm = min(live, 65535)                 # additive blending saturates the 16-bit channel at 1.0
        raw = (0xABCD << 16) | unorm             # the other Rg32 channel carries a flag: masked off by the decode
        cases.append({"live_slots": live, "raw": raw, "expected_count": raw & 0xFFFF})
    return {"source": "Illuminant/Shaders/CountLiveParticles.fx:5-40, Illuminant/Particles/ParticleEngine.cs:244-247", "tolerance": "exact", "cases": cases}


# ---------------------------------------------------------------------------------------------------------
# 5. Distance encoding -- DistanceFieldCommon.fxh:8,264-270: encode(d) = 192/255 - d/maxDist, decode(e) = (192/255 - e)*maxDist
# ---------------------------------------------------------------------------------------------------------
def gen_encoding():
    cases = []
    for max_d in (128.0, 256.0):
        z = F(192.0) / F(255.0)
        for d in (0.0, 1.0, -1.0, 17.5, -31.25, 96.0, -32.0):
            e = z - F(d) / F(max_d)
            cases.append({"distance": d, "max_distance": max_d, "encoded": float(e), "decoded": float((z - e) * F(max_d))})
    return {"source": "Illuminant/Shaders/DistanceFieldCommon.fxh:8,264-270", "tolerance": "1e-6 absolute on encoded, 1e-4 on decoded", "cases": cases}


# ---------------------------------------------------------------------------------------------------------
# 6. G-buffer texel <-> world position / normal -- encode side GBufferShaderCommon.fxh:10-35 +
#    encodeNormalSpherical EnvironmentCommon.fxh:34-38; decode side LightCommon.fxh:58-144, EnvironmentCommon.fxh:40-51.
#    The expected values are the INPUTS of the encode (round trip), i.e. independent of the decode restatement --
#    except for a fullbright texel, whose z is not recoverable: the decode zeroes it BEFORE the *1024 - 1024 rescale
#    (LightCommon.fxh:91-99), so such a pixel reports z = -1024 (it is skipped by every light anyway).
# ---------------------------------------------------------------------------------------------------------
def encode_normal_spherical(n):
    # float2(atan2(n.y, n.x) / PI, n.z) * 0.5 + 0.5
    return [(math.atan2(n[1], n[0]) / math.pi) * 0.5 + 0.5, n[2] * 0.5 + 0.5]


def gen_gbuffer():
    cases = []
    for normal, rel_y, z, shadows, fullbright in (((0.0, 0.0, 1.0), 0.0, 0.0, True, False),       # ground plane: texel (0.5, 1.0, 0, 1.0)
                                                 ((0.0, 1.0, 0.0), -12.0, 12.0, True, False),      # front face of a height volume
                                                 ((0.6, 0.0, 0.8), 3.0, 40.0, True, False),
                                                 ((-0.48, 0.64, 0.6), 0.0, 0.0, False, False),     # shadows disabled at z = 0
                                                 ((0.0, 0.0, 1.0), 0.0, 25.0, False, False),
                                                 ((0.0, 0.0, 1.0), 0.0, 0.0, True, True)):         # fullbright
        enc = encode_normal_spherical(normal)
        if fullbright:
            w = 99999.0
        else:
            w = ((z + 1024.0) / 1024.0) * (1 if shadows else -1) + (0 if shadows else -1)
        cases.append({"texel": [enc[0], enc[1], rel_y, w], "pixel": [37.0, 21.0],
                      "expected": {"normal": list(normal), "world_z": -1024.0 if fullbright else z, "world_xy": [37.0, 21.0 + rel_y],
                                   "enable_shadows": bool(shadows and not fullbright), "fullbright": bool(fullbright)}})
    return {"source": "Illuminant/Shaders/GBufferShaderCommon.fxh:10-35, EnvironmentCommon.fxh:34-51, LightCommon.fxh:58-144",
            "tolerance": "1e-5 absolute on the normal, 2e-4 on world_z (the /1024 round trip)", "cases": cases}


# ---------------------------------------------------------------------------------------------------------
# 7. Closed-form single-light / single-particle answers, derived by hand from the cited lines
# ---------------------------------------------------------------------------------------------------------
def gen_closed_form():
    cases = []
    # (a) light centred on the pixel, no distance field: distance 0 < radius => saturate(radius - distance) = 1 =>
    #     opacity 1 (LightCommon.fxh:208-213); lightmap = ambient + rgb * a (SphereLight.fx:42-45), alpha += 1
    cases.append({"kind": "light_at_pixel", "light": {"position": [8.0, 6.0, 0.0], "radius": 4.0, "ramp": 10.0, "color": [0.25, 0.5, 0.75, 0.5]},
                  "ambient": [0.05, 0.06, 0.07, 1.0], "pixel": [8, 6], "expected": [0.05 + 0.125, 0.06 + 0.25, 0.07 + 0.375, 2.0]})
    # (b) linear ramp on the ground plane, no field, light at height 0: at distance radius + k*ramp the opacity is (1 - k)
    #     (normal factor: light in the plane => dot = 0 => pow(saturate(0.15/0.15), .85) = 1; LightCommon.fxh:154-214)
    for k in (0.25, 0.5, 0.75):
        d = 4.0 + k * 16.0
        # (the pixel shader works at the integer VPOS, so pixel (x, y) shades world point (x, y): light on a pixel corner)
        cases.append({"kind": "linear_ramp", "light": {"position": [4.0, 4.0, 0.0], "radius": 4.0, "ramp": 16.0, "color": [1.0, 1.0, 1.0, 1.0]},
                      "ambient": [0.0, 0.0, 0.0, 1.0], "pixel": [4 + int(d), 4], "expected_rgb": 1.0 - k})
    # (c) Gravity, one Linear attractor (Gravity.fx:36-60): a = normalize(c - p) * (1 - d/r) * dt * strength, capped by
    #     MaximumAcceleration * dt; particle (100,100,0) v = 0, attractor (160,180,0) r = 200 strength 60, dt = 1/60:
    #     d = 100, a = (.6,.8,0) * .5 * (1/60) * 60 = (.3,.4,0); cap 8/60 = .1333 => a = (.08, .10667, 0)
    cases.append({"kind": "gravity_linear", "position": [100.0, 100.0, 0.0, 1.0], "velocity": [0.0, 0.0, 0.0, 0.0],
                  "attractor": {"position": [160.0, 180.0, 0.0], "radius": 200.0, "strength": 60.0, "type": 1},
                  "dt": 1.0 / 60.0, "maximum_acceleration": 8.0, "expected_velocity": [0.6 * 8.0 / 60.0, 0.8 * 8.0 / 60.0, 0.0, 0.0]})
    cases.append({"kind": "gravity_linear", "position": [100.0, 100.0, 0.0, 1.0], "velocity": [1.0, -2.0, 0.5, 0.0],
                  "attractor": {"position": [160.0, 180.0, 0.0], "radius": 200.0, "strength": 60.0, "type": 1},
                  "dt": 1.0 / 60.0, "maximum_acceleration": 1024.0, "expected_velocity": [1.3, -1.6, 0.5, 0.0]})
    # (d) UpdatePositions (UpdateParticleSystem.fx:9-38, UpdateCommon.fxh:20-35): v = (30, 40, 0) |v| = 50, friction .1,
    #     dt = .1: l = 50 - 50*.1*.1 = 49.5 => v' = (29.7, 39.6, 0); p' = p + v'*dt; life' = life - decay*dt
    cases.append({"kind": "update_positions", "position": [10.0, 20.0, 5.0, 2.0], "velocity": [30.0, 40.0, 0.0, 3.0],
                  "dt": 0.1, "friction": 0.1, "max_velocity": 2048.0, "life_decay": 1.5,
                  "expected_position": [12.97, 23.96, 5.0, 1.85], "expected_velocity": [29.7, 39.6, 0.0, 3.0]})
    # life runs out => slot zeroed (UpdateParticleSystem.fx:28-35)
    cases.append({"kind": "update_positions", "position": [10.0, 20.0, 5.0, 0.1], "velocity": [30.0, 40.0, 0.0, 3.0],
                  "dt": 0.1, "friction": 0.1, "max_velocity": 2048.0, "life_decay": 1.5,
                  "expected_position": [0.0, 0.0, 0.0, 0.0], "expected_velocity": [0.0, 0.0, 0.0, 0.0]})
    # (e) FMA (FMA.fx:22-51, Transforms.cs:38-45): no area => weight = Strength; t = Strength * dtMs / (1000/cps);
    #     Strength 1, dt 1/60 s, 10 cycles/s => t = 16.6667/100 = 1/6; v' = lerp(v, v*mul + add, t)
    cases.append({"kind": "fma", "position": [6.0, 12.0, 0.0, 1.0], "velocity": [60.0, -30.0, 0.0, 0.0], "dt": 1.0 / 60.0, "strength": 1.0,
                  "cycles_per_second": 10.0, "position_add": [6.0, 0.0, 0.0], "position_multiply": [1.0, 1.0, 1.0],
                  "velocity_add": [0.0, 0.0, 0.0], "velocity_multiply": [0.4, 0.4, 1.0],
                  "expected_position": [7.0, 12.0, 0.0, 1.0], "expected_velocity": [54.0, -27.0, 0.0, 0.0]})
    return {"source": "hand-derived from the cited HLSL lines (see comments in make_golden.py)", "tolerance": "1e-5 relative", "cases": cases}


# ---------------------------------------------------------------------------------------------------------
# 8. Distance-field generation: closed-form stored codes of single obstructions / height volumes,
#    derived by hand from DistanceFunctionCommon.fxh:48-124, DistanceField.fx:47-73 (finalEval, PolygonXyBias 1.5),
#    encodeDistance (DistanceFieldCommon.fxh:264-266), SliceIndexToZ (LightingRenderer.DistanceField.cs:32-35) and the
#    unorm16 render-target write.  Evaluated in float64; the oracle / HIP path must hit the code within +-1.
# ---------------------------------------------------------------------------------------------------------
def gen_distance_field_generation():
    vw, vh, depth, slices, max_enc = 128, 128, 64.0, 9, 128.0     # resolution 1: texel (x, y) of a slice is world (x, y)

    def code(distance):
        enc = 192.0 / 255.0 - distance / max_enc
        return int(math.floor(min(max(enc, 0.0), 1.0) * 65535.0 + 0.5))

    def slice_z(k):
        return k / float(slices) * depth

    cases = []
    # (a) axis-aligned box, centre (64,64,0) half-size (16,16,32): outside along x the distance is x - 64 - 16
    #     (z = 0 and y = 64 are inside); inside, max(dx, dy, dz) = -16 at the centre column of slice 0
    box = {"type": 1, "center": [64.0, 64.0, 0.0], "size": [16.0, 16.0, 32.0], "rotation": 0.0}
    cases.append({"kind": "obstruction", "obstruction": box, "texel": [100, 64], "slice": 0, "expected_code": code(20.0), "distance": 20.0})
    cases.append({"kind": "obstruction", "obstruction": box, "texel": [64, 64], "slice": 0, "expected_code": code(-16.0), "distance": -16.0})
    #     slice 6 is z = 6/9*64 = 42.667: above the box top (32) by 10.667, laterally inside
    cases.append({"kind": "obstruction", "obstruction": box, "texel": [64, 64], "slice": 6, "expected_code": code(slice_z(6) - 32.0), "distance": slice_z(6) - 32.0})
    #     corner: outside in x by 12 and in y by 5 at z = 0 -> length((12,5,0)) = 13
    cases.append({"kind": "obstruction", "obstruction": box, "texel": [92, 85], "slice": 0, "expected_code": code(13.0), "distance": 13.0})
    # (b) sphere as a Spheroid of equal radii 10 at (40,50,0): |p - c| - r; texel (46,58) slice 0 -> 10 - 10 = 0 -> DISTANCE_ZERO
    sph = {"type": 3, "center": [40.0, 50.0, 0.0], "size": [10.0, 10.0, 10.0], "rotation": 0.0}
    cases.append({"kind": "obstruction", "obstruction": sph, "texel": [46, 58], "slice": 0, "expected_code": code(0.0), "distance": 0.0})
    cases.append({"kind": "obstruction", "obstruction": sph, "texel": [70, 90], "slice": 0, "expected_code": code(40.0), "distance": 40.0})
    #     slice 3 (z = 21.333) straight above the centre: 21.333 - 10
    cases.append({"kind": "obstruction", "obstruction": sph, "texel": [40, 50], "slice": 3, "expected_code": code(slice_z(3) - 10.0), "distance": slice_z(3) - 10.0})
    # (c) ellipsoid (improvedV2), inside branch (k0 < 1): (k0 - 1) * min(r); centre (64,64,0) radii (40,20,30), texel (84,64): k0 = .5
    ell = {"type": 0, "center": [64.0, 64.0, 0.0], "size": [40.0, 20.0, 30.0], "rotation": 0.0}
    cases.append({"kind": "obstruction", "obstruction": ell, "texel": [84, 64], "slice": 0, "expected_code": code(-10.0), "distance": -10.0})
    #     outside on the x axis: k0 = x/rx, k1 = x/rx^2 -> k0 (k0 - 1) / k1 = (k0 - 1) rx = x - rx; texel (124,64): 60 - 40 = 20
    cases.append({"kind": "obstruction", "obstruction": ell, "texel": [124, 64], "slice": 0, "expected_code": code(20.0), "distance": 20.0})
    # (d) cylinder: radius = length(size.xy) = 5 for size (3,4,h) (DistanceFunctionCommon.fxh:122); texel 20 px off-axis, z inside
    cyl = {"type": 2, "center": [30.0, 30.0, 0.0], "size": [3.0, 4.0, 48.0], "rotation": 0.0}
    cases.append({"kind": "obstruction", "obstruction": cyl, "texel": [50, 30], "slice": 0, "expected_code": code(15.0), "distance": 15.0})
    # (e) a box rotated by 90 degrees about z swaps its x / y half-sizes: half-size (30,6,32) rotated -> (6,30) footprint
    rbox = {"type": 1, "center": [64.0, 64.0, 0.0], "size": [30.0, 6.0, 32.0], "rotation": math.pi / 2}
    cases.append({"kind": "obstruction", "obstruction": rbox, "texel": [80, 64], "slice": 0, "expected_code": code(10.0), "distance": 10.0})
    cases.append({"kind": "obstruction", "obstruction": rbox, "texel": [64, 104], "slice": 0, "expected_code": code(10.0), "distance": 10.0})
    # (f) far texel: encoded distance < 0 saturates to code 0 at the render target
    cases.append({"kind": "obstruction", "obstruction": cyl, "texel": [127, 127], "slice": 0, "expected_code": 0, "distance": math.hypot(97, 97) - 5.0})
    # (g) height volume: square (32,32)-(96,96) over z in [0, 20].  finalEval: inside on xy and z -> (d_xy + 1.5) + d_z
    sq = {"polygon": [[32.0, 32.0], [96.0, 32.0], [96.0, 96.0], [32.0, 96.0]], "z_base": 0.0, "height": 20.0}
    zz = slice_z(1)     # 7.111: inside [0, 20]; distanceZ = max(z - 20, 0 - z) = -7.111
    cases.append({"kind": "height_volume", "volume": sq, "texel": [64, 64], "slice": 1, "expected_code": code((-32.0 + 1.5) + max(zz - 20.0, -zz)),
                  "distance": (-32.0 + 1.5) + max(zz - 20.0, -zz)})
    #     outside on xy by 24 (+1.5 bias), z inside -> 25.5 + max(dz, 0) = 25.5
    cases.append({"kind": "height_volume", "volume": sq, "texel": [120, 64], "slice": 1, "expected_code": code(25.5), "distance": 25.5})
    #     inside on xy, above the top: slice 6 z = 42.667 -> just the z distance 22.667
    cases.append({"kind": "height_volume", "volume": sq, "texel": [64, 64], "slice": 6, "expected_code": code(slice_z(6) - 20.0), "distance": slice_z(6) - 20.0})
    #     outside on both: 25.5 + 22.667
    cases.append({"kind": "height_volume", "volume": sq, "texel": [120, 64], "slice": 6, "expected_code": code(25.5 + slice_z(6) - 20.0), "distance": 25.5 + slice_z(6) - 20.0})
    return {"source": "hand-derived from DistanceFunctionCommon.fxh:48-124, DistanceField.fx:47-73, DistanceFieldCommon.fxh:264-266, "
                      "LightingRenderer.DistanceField.cs:32-35 (see comments in make_golden.py)",
            "field": {"virtual_width": vw, "virtual_height": vh, "virtual_depth": depth, "requested_slices": slices, "resolution": 1.0,
                      "maximum_encoded_distance": max_enc},
            "tolerance": "+-1 unorm16 code (1.5e-5 of the encoded range, 0.002 world units)", "cases": cases}


# ---------------------------------------------------------------------------------------------------------
# 9. Remaining particle techniques: closed forms derived by hand from MatrixMultiply.fx:22-52 + mul3
#    (ParticleCommon.fxh:183-196), Noise.fx:74-116 (SpatialNoise), SpawnParticles.fx:32-118 (position texture, feedback)
# ---------------------------------------------------------------------------------------------------------
def gen_transforms_ext():
    cases = []
    dt = 1.0 / 60.0
    w = (dt * 1000.0) / 100.0          # Strength 1, no area => weight 1; timeScale = dtMs / (1000 / 10 cycles per second)
    # (a) translation by (5,-3,2) (row-vector convention: M41..M43) + velocity scaled by .5: lerp(p, p + T, w), lerp(v, .5 v, w);
    #     life / category (w components) come back unchanged from mul3
    T = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 5, -3, 2, 1]
    S = [0.5, 0, 0, 0, 0, 0.5, 0, 0, 0, 0, 0.5, 0, 0, 0, 0, 1]
    pos, vel = [10.0, 20.0, 30.0, 2.0], [6.0, -12.0, 3.0, 7.0]
    cases.append({"kind": "matrix_multiply", "position": pos, "velocity": vel, "position_matrix": T, "velocity_matrix": S,
                  "cycles_per_second": 10.0, "strength": 1.0, "dt": dt,
                  "expected_position": [10.0 + 5 * w, 20.0 - 3 * w, 30.0 + 2 * w, 2.0],
                  "expected_velocity": [6.0 * (1 - 0.5 * w), -12.0 * (1 - 0.5 * w), 3.0 * (1 - 0.5 * w), 7.0]})
    # (b) M44 = 2: positions are divided by temp.w (w argument 1), velocities are not (w argument 0)
    H2 = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 2]
    cases.append({"kind": "matrix_multiply", "position": pos, "velocity": vel, "position_matrix": H2, "velocity_matrix": H2,
                  "cycles_per_second": 10.0, "strength": 1.0, "dt": dt,
                  "expected_position": [10.0 * (1 - 0.5 * w), 20.0 * (1 - 0.5 * w), 30.0 * (1 - 0.5 * w), 2.0],
                  "expected_velocity": vel})
    # (c) CyclesPerSecond == null => TimeDivisor -1 => timeScale 1: the weight is the Strength itself
    cases.append({"kind": "matrix_multiply", "position": pos, "velocity": vel, "position_matrix": T, "velocity_matrix": S,
                  "cycles_per_second": None, "strength": 0.25, "dt": dt,
                  "expected_position": [10.0 + 5 * 0.25, 20.0 - 3 * 0.25, 30.0 + 2 * 0.25, 2.0],
                  "expected_velocity": [6.0 * 0.875, -12.0 * 0.875, 3.0 * 0.875, 7.0]})
    # (d) dead slots and filtered categories pass through untouched
    cases.append({"kind": "matrix_multiply", "position": [1.0, 2.0, 3.0, 0.0], "velocity": vel, "position_matrix": T, "velocity_matrix": S,
                  "cycles_per_second": 10.0, "strength": 1.0, "dt": dt, "expected_position": [1.0, 2.0, 3.0, 0.0], "expected_velocity": vel})
    # (e) SpatialNoise on a constant randomness table c: every bilinear tap returns q = round(c * 65535) / 65535, so
    #     positionDelta = (q + offset) * scale everywhere; t = 1/6; ReplaceOldVelocity => v' = lerp(v, vDelta, weight 1) + normalize(v) * delta.w
    c = 0.75
    q = round(c * 65535.0) / 65535.0
    pd = [(q - 0.5) * 2.0, (q - 0.5) * 4.0, (q - 0.5) * 6.0]
    vd = [(q - 0.5) * 10.0, (q - 0.5) * 20.0, (q - 0.5) * 30.0]
    cases.append({"kind": "spatial_noise", "table_value": c, "position": [100.0, 50.0, 5.0, 3.0], "velocity": [3.0, 4.0, 0.0, 1.0],
                  "position_scale": [2.0, 4.0, 6.0, 0.0], "velocity_scale": [10.0, 20.0, 30.0], "speed_scale": 5.0, "space_scale": [8.0, 8.0],
                  "cycles_per_second": 10.0, "dt": dt, "replace_old_velocity": True,
                  "expected_position": [100.0 + pd[0] * w, 50.0 + pd[1] * w, 5.0 + pd[2] * w, 3.0],
                  "expected_velocity": [vd[0] + 0.6 * (q - 0.5) * 5.0, vd[1] + 0.8 * (q - 0.5) * 5.0, vd[2], 1.0]})
    cases.append({"kind": "spatial_noise", "table_value": c, "position": [100.0, 50.0, 5.0, 3.0], "velocity": [3.0, 4.0, 0.0, 1.0],
                  "position_scale": [2.0, 4.0, 6.0, 0.0], "velocity_scale": [10.0, 20.0, 30.0], "speed_scale": 0.0, "space_scale": [8.0, 8.0],
                  "cycles_per_second": 10.0, "dt": dt, "replace_old_velocity": False,
                  "expected_position": [100.0 + pd[0] * w, 50.0 + pd[1] * w, 5.0 + pd[2] * w, 3.0],
                  "expected_velocity": [3.0 + vd[0] * w, 4.0 + vd[1] * w, 0.0 + vd[2] * w, 1.0]})
    # (f) position texture: 6 positions, no polygon rate: slot first + k spawns at position (k + TotalSpawned % 6) % 6 exactly
    #     (position scale / offset 0 => the constant itself; life constant 2.5)
    positions = [[10.0 * i, 100.0 - i, 1.0 + i] for i in range(6)]
    cases.append({"kind": "position_buffer", "positions": positions, "first": 40, "last": 52, "total_spawned": 15, "life": 2.5,
                  "expected": [{"slot": 40 + k, "position": positions[(k + 15 % 6) % 6] + [2.5]} for k in range(13)]})
    # (g) feedback: new particle k (slot first + k) reads source slot floor(k / InstanceMultiplier) + FeedbackSourceIndex, lands on the
    #     source position (AlignPositionConstant, constant 0) and inherits SourceVelocityFactor * source velocity; sources outside
    #     SourceLifeRange are skipped (the slot keeps its old contents)
    cases.append({"kind": "feedback", "instance_multiplier": 3, "source_index": 20, "first": 100, "last": 111, "source_velocity_factor": 0.5,
                  "source_life_range": [0.5, 9999.0], "dead_source_slots": [22],
                  "expected_source_of_slot": {str(100 + k): 20 + k // 3 for k in range(12)}})
    # (h) PatternSpawner (PatternSpawner.fx:21-97, SpecialSpawners.cs:208-256).  With Linear formulas and zero random scales the new
    #     particle k (slot first + k) of a whole spawn has indexXy = (k % perRow, k / perRow), sits at constant + indexXy * Divisor -
    #     size / 2 and takes the texture colour at uv = indexXy * Divisor / size - 0.5 / size.  In texel space (u * w - 0.5) that is
    #     indexXy * Divisor - 1 at level 0 -- the reference's half-texel offset lands one texel up and to the left of the pixel it
    #     stands on; CLAMP repeats the first row / column.
    def pattern_case(name, tex_w, tex_h, levels, level_used, divisor, current_row, count, color_constant, multiply, first=32):
        lw, lh = max(1, tex_w >> level_used), max(1, tex_h >> level_used)
        texel = lambda x, y: [10.0 * x + 5.0, 100.0 * y + 7.0, 0.5, 1.0]
        chain = []
        for l in range(levels):
            w_, h_ = max(1, tex_w >> l), max(1, tex_h >> l)
            if l == level_used:
                chain.append([[texel(x, y) for x in range(w_)] for y in range(h_)])
            else:
                chain.append([[[999.0, 999.0, 999.0, 999.0] for x in range(w_)] for y in range(h_)])     # must not be read
        npot = lambda v: 0 if v <= 0 else 1 << (v - 1).bit_length()
        per_row = npot(tex_w // divisor)
        constant, life = [100.0, 200.0, 5.0], 2.0
        expected, rejected = [], []
        for k in range(count):
            ix, iy = k % per_row, k // per_row + current_row
            u = ix * divisor / tex_w - 0.5 / tex_w
            v = iy * divisor / tex_h - 0.5 / tex_h + (current_row * divisor) // tex_h
            if u > 1 or v > 1:
                rejected.append(first + k)
                continue
            # bilinear on the used level: the texel function is linear in x and y, so the filtered value is the function at the
            # clamped sample position
            sx = min(max(u * lw - 0.5, 0.0), lw - 1.0)
            sy = min(max(v * lh - 0.5, 0.0), lh - 1.0)
            colour = [10.0 * sx + 5.0, 100.0 * sy + 7.0, 0.5, 1.0]
            attr = [c * k_ for c, k_ in zip(colour, color_constant)] if multiply else [c + k_ for c, k_ in zip(colour, color_constant)]
            expected.append({"slot": first + k, "position": [constant[0] + ix * divisor - tex_w * 0.5, constant[1] + iy * divisor - tex_h * 0.5,
                                                              constant[2], life], "attributes": attr})
        return {"kind": "pattern", "name": name, "texture_levels": chain, "divisor": divisor, "current_row": current_row, "first": first,
                "last": first + count - 1, "position_constant": constant, "life": life, "color_constant": color_constant, "multiply": multiply,
                "expected": expected, "rejected_slots": rejected}
    cases.append(pattern_case("whole 8x4", 8, 4, 1, 0, 1, 0, 32, [0.5, 2.0, 1.0, 1.0], True))
    cases.append(pattern_case("row 2 of 8x4", 8, 4, 1, 0, 1, 2, 8, [0.5, 2.0, 1.0, 1.0], True, first=0))
    # Divisor 2: LOD = log2(2) - 0.5 = 0.5 => nearest level 1 (4x4), additive colour constant
    cases.append(pattern_case("divisor 2 reads mip 1", 8, 8, 2, 1, 2, 0, 16, [0.25, 0.0, 0.0, 0.0], False))
    # 6 texels wide => 8 particles per row (next power of two); the 8th has u > 1 and is rejected, the 7th is one of the
    # "garbage particles on the right" the shader comment mentions and survives
    cases.append(pattern_case("npot 6x4", 6, 4, 1, 0, 1, 0, 32, [1.0, 1.0, 1.0, 1.0], True))
    # (i) AlignVelocityAndPosition lives INSIDE evaluateRandomForIndex (SpawnerCommon.fxh:106-117: "The x and y element of random samples
    #     determines the normal": random2.xy = random1.xy), which PS_SpawnFeedback (SpawnParticles.fx:83) and PS_SpawnPattern
    #     (PatternSpawner.fx:63) call like Spawn_Stage1 does.  With Spherical position and velocity formulas (constants 0, offsets 0) the new
    #     position is n(random1.xy) * random1.z * scale and the velocity n(random2.xy) * random2.z * scale: aligned, the velocity of every
    #     new particle points along its offset from the formula's centre -- whatever the randomness table holds.  (Unaligned they are
    #     unrelated directions.)  The centre is the position constant, plus the pattern spawner's per-pixel offset.
    cases.append({"kind": "aligned_spawn", "spawner": "feedback", "first": 64, "last": 64 + 95, "position_scale": 40.0, "velocity_scale": 9.0})
    cases.append({"kind": "aligned_spawn", "spawner": "pattern", "first": 16, "last": 16 + 31, "position_scale": 40.0, "velocity_scale": 9.0,
                  "texture_size": [8, 4], "position_constant": [100.0, 200.0, 5.0]})
    return {"source": "hand-derived from MatrixMultiply.fx:22-52, ParticleCommon.fxh:183-196, Noise.fx:74-116, RandomCommon.fxh:36-39, "
                      "SpawnParticles.fx:32-118, ParticleSpawner.cs:301-367, PatternSpawner.fx:21-97, SpecialSpawners.cs:208-256 "
                      "(see comments in make_golden.py)",
            "tolerance": "1e-5 relative", "cases": cases}


# ---------------------------------------------------------------------------------------------------------
# 10. Particle lights and light probes: closed forms from ParticleLight.fx:16-118, SphereLightProbe.fx:19-44,
#     LightCommon.fxh:154-214 (no distance field => cone trace 1, no AO)
# ---------------------------------------------------------------------------------------------------------
def gen_lights_ext():
    cases = []
    # (a) a particle on the shaded pixel: distance 0 < radius => opacity 1.  Colour = (renderColor.rgb / renderColor.a, renderColor.a) *
    #     LightColor, blended as rgb * a on top of what the lightmap holds; alpha counts the light (ParticleLight.fx:40-45,72,113-116)
    rc = [0.2, 0.1, 0.05, 0.5]
    lc = [1.0, 0.5, 2.0, 0.8]
    un = [rc[0] / rc[3], rc[1] / rc[3], rc[2] / rc[3], rc[3]]
    light = [un[i] * lc[i] for i in range(4)]
    base = [0.05, 0.06, 0.07, 1.0]
    cases.append({"kind": "particle_light", "particle": {"position": [8.0, 6.0, 0.0, 1.5], "render_color": rc}, "light_color": lc,
                  "radius": 4.0, "ramp": 10.0, "lightmap_before": base, "pixel": [8, 6],
                  "expected": [base[0] + light[0] * light[3], base[1] + light[1] * light[3], base[2] + light[2] * light[3], 2.0]})
    # (b) the same particle dead (life 0), transparent (render alpha 0) or beyond the chunk's quad count: the lightmap keeps its contents
    for why, part, quads in (("dead", {"position": [8.0, 6.0, 0.0, 0.0], "render_color": rc}, 64),
                             ("transparent", {"position": [8.0, 6.0, 0.0, 1.5], "render_color": [0.2, 0.1, 0.05, 0.0]}, 64),
                             ("beyond_quad_count", {"position": [8.0, 6.0, 0.0, 1.5], "render_color": rc}, 3)):
        cases.append({"kind": "particle_light", "why": why, "particle": part, "slot": 5, "quad_count": quads, "light_color": lc,
                      "radius": 4.0, "ramp": 10.0, "lightmap_before": base, "pixel": [8, 6], "expected": base})
    # (c) linear ramp: ground plane (normal +z), particle in the plane => dot(light direction, normal) = 0 => normal factor 1;
    #     at distance radius + k ramp the opacity is 1 - k (LightCommon.fxh:154-214)
    for k in (0.25, 0.5):
        d = 4.0 + k * 16.0
        cases.append({"kind": "particle_light", "particle": {"position": [4.0, 4.0, 0.0, 1.0], "render_color": [1.0, 1.0, 1.0, 1.0]},
                      "light_color": [1.0, 1.0, 1.0, 1.0], "radius": 4.0, "ramp": 16.0, "lightmap_before": [0.0, 0.0, 0.0, 1.0],
                      "pixel": [4 + int(d), 4], "expected": [1.0 - k, 1.0 - k, 1.0 - k, 2.0]})
    # (d) light probe without a normal (normal factor 1), 3 lights at distances inside the radius / mid-ramp / beyond the ramp:
    #     value = sum colour.rgb * colour.a * opacity; alpha = number of lights that were not discarded (opacity 0 => discard)
    probe = [100.0, 50.0, 10.0]
    lights = [{"position": [100.0, 50.0, 12.0], "radius": 5.0, "ramp": 20.0, "color": [1.0, 0.0, 0.0, 0.5]},      # distance 2 < radius: 1
              {"position": [115.0, 50.0, 10.0], "radius": 5.0, "ramp": 20.0, "color": [0.0, 1.0, 0.0, 1.0]},      # distance 15: 1 - 10/20 = .5
              {"position": [100.0, 90.0, 10.0], "radius": 5.0, "ramp": 20.0, "color": [0.0, 0.0, 1.0, 1.0]}]      # distance 40 > 25: discarded
    cases.append({"kind": "light_probe", "probe": {"position": probe, "normal": None, "enable_shadows": True}, "lights": lights,
                  "expected": [0.5, 0.5, 0.0, 2.0]})
    # (e) a probe facing away from the light (normal . direction to light = -1): normal factor = pow(saturate((-1 + .15) / .15), .85) = 0 => discarded
    cases.append({"kind": "light_probe", "probe": {"position": probe, "normal": [-1.0, 0.0, 0.0], "enable_shadows": True}, "lights": [lights[1]],
                  "expected": [0.0, 0.0, 0.0, 0.0]})
    cases.append({"kind": "light_probe", "probe": {"position": probe, "normal": [1.0, 0.0, 0.0], "enable_shadows": True}, "lights": [lights[1]],
                  "expected": [0.0, 0.5, 0.0, 1.0]})
    # (f) technique SphereLightProbeWithDistanceRamp (SphereLightProbe.fx:46-72, SphereLightPixelEpilogueWithRamp, SphereLightCore.fxh:99-119):
    #     the opacity becomes SampleFromRamp2(preTraceOpacity, (angle + rampOffsetForGPU) * rampRateForGPU).rgb * coneOpacity.
    #     A 2 x 1 ramp red -> blue read at u = preTraceOpacity = .5 (mid-ramp distance) is the midpoint of its two texels.
    ramp_u = [[[1.0, 0.0, 0.0, 1.0], [0.0, 0.0, 1.0, 1.0]]]
    cases.append({"kind": "light_probe", "probe": {"position": probe, "normal": None, "enable_shadows": True},
                  "lights": [{"position": [115.0, 50.0, 10.0], "radius": 5.0, "ramp": 20.0, "color": [1.0, 1.0, 1.0, 1.0]}],
                  "ramp_texture": ramp_u, "expected": [0.5, 0.0, 0.5, 1.0]})
    #     A 1 x 4 ramp addressed by the angle around the light (V WRAP): the probe sits on the light's -x side, angle = atan2(0, -15) = pi;
    #     RampOffsetForGPU = -pi + RampOffset, RampRateForGPU = RampRate / 2 pi => v = RampOffset / 2 pi; RampOffset = 2 pi * 0.375 => v * 4 - .5 = 1:
    #     exactly row 1.  u = 1 (inside the radius) is clamped.
    ramp_v = [[[0.1, 0.2, 0.3, 1.0]], [[0.9, 0.8, 0.7, 1.0]], [[0.4, 0.5, 0.6, 1.0]], [[0.0, 1.0, 0.0, 1.0]]]
    cases.append({"kind": "light_probe", "probe": {"position": probe, "normal": None, "enable_shadows": True},
                  "lights": [{"position": [115.0, 50.0, 10.0], "radius": 20.0, "ramp": 20.0, "color": [1.0, 1.0, 1.0, 0.5], "ramp_offset": 2 * math.pi * 0.375}],
                  "ramp_texture": ramp_v, "expected": [0.45, 0.4, 0.35, 1.0]})
    #     a 1 x 1 ramp is no ramp at all (LightingRenderer.cs:822-827): the plain technique's answer
    cases.append({"kind": "light_probe", "probe": {"position": probe, "normal": None, "enable_shadows": True},
                  "lights": [{"position": [115.0, 50.0, 10.0], "radius": 5.0, "ramp": 20.0, "color": [0.0, 1.0, 0.0, 1.0]}],
                  "ramp_texture": [[[0.3, 0.3, 0.3, 1.0]]], "expected": [0.0, 0.5, 0.0, 1.0]})
    return {"source": "hand-derived from ParticleLight.fx:16-118, SphereLightProbe.fx:19-72, SphereLightCore.fxh:99-119, RampCommon.fxh, LightCommon.fxh:154-254 (see comments in make_golden.py)",
            "tolerance": "1e-5 relative", "cases": cases}


# ---------------------------------------------------------------------------------------------------------
# 11. Output side: FillReadbackResult (ParticleReadback.cs:73-167, C# CPU code) and the lightmap resolve
#     (Resolve.fx:62-139, HDR.fxh, clamps of IlluminantMaterials.cs:81-137), evaluated by hand / in float64
# ---------------------------------------------------------------------------------------------------------
def gen_output_ext():
    cases = []
    two_pi = float(F(2 * math.pi))
    # (a) one live particle, no animation frames (region = unit): bytes are truncated ((byte)(rc * 255)), scale = pSize * size,
    #     rotation = renderData.y % 2 pi when RotationFromVelocity, SortOrder = y + ZToY when SortedReadback
    cases.append({"kind": "readback", "params": {"size": [3.0, 2.0], "region": [0.0, 0.0, 1.0, 1.0], "animation_rate": [0.0, 0.0], "z_to_y": 0.5,
                                                  "column_from_velocity": False, "row_from_velocity": False, "rotation_from_velocity": True, "sorted": True},
                  "particles": [{"slot": 3, "position": [10.0, 20.0, 5.0, 1.3], "render_color": [0.5, 0.25, 1.0, 0.999], "render_data": [1.5, 7.0, 9.0, 0.0]}],
                  "expected": [{"position": [10.0, 20.0], "scale": [4.5, 3.0], "region": [0.0, 0.0, 1.0, 1.0], "rotation": float(F(math.fmod(7.0, two_pi))),
                                "sort_order": 20.5, "color": [127, 63, 255, 254]}]})
    # (b) a 4 x 2 frame sheet (region .25 x .5): frame x = floor(|rate.x| * life) % 4, frame y = clamp(floor(renderData.w), 0, 1);
    #     no RotationFromVelocity => rotation 0; unsorted => SortOrder 0; dead slots are skipped and the order is slot order
    cases.append({"kind": "readback", "params": {"size": [1.0, 1.0], "region": [0.0, 0.0, 0.25, 0.5], "animation_rate": [2.0, 0.0], "z_to_y": 0.0,
                                                  "column_from_velocity": False, "row_from_velocity": False, "rotation_from_velocity": False, "sorted": False},
                  "particles": [{"slot": 9, "position": [1.0, 2.0, 0.0, 1.3], "render_color": [1.0, 1.0, 1.0, 1.0], "render_data": [2.0, 1.0, 0.0, 1.0]},
                                {"slot": 4, "position": [5.0, 6.0, 0.0, 0.0], "render_color": [1.0, 1.0, 1.0, 1.0], "render_data": [1.0, 0.0, 0.0, 0.0]},
                                {"slot": 2, "position": [7.0, 8.0, 0.0, 3.6], "render_color": [0.0, 0.0, 0.0, 0.0], "render_data": [1.0, 0.0, 0.0, 5.0]}],
                  "expected": [{"position": [7.0, 8.0], "scale": [1.0, 1.0], "region": [0.75, 0.5, 1.0, 1.0], "rotation": 0.0, "sort_order": 0.0, "color": [0, 0, 0, 0]},
                               {"position": [1.0, 2.0], "scale": [2.0, 2.0], "region": [0.5, 0.5, 0.75, 1.0], "rotation": 0.0, "sort_order": 0.0, "color": [255, 255, 255, 255]}]})
    # (c) ColumnFromVelocity: + Math.Round(rot / (2 pi / 4)); rot = 4.0 -> 4 / 1.5708 = 2.546 -> 3; frame x = (0 + 3) % 4 = 3
    cases.append({"kind": "readback", "params": {"size": [1.0, 1.0], "region": [0.0, 0.0, 0.25, 1.0], "animation_rate": [0.0, 0.0], "z_to_y": 0.0,
                                                  "column_from_velocity": True, "row_from_velocity": False, "rotation_from_velocity": False, "sorted": False},
                  "particles": [{"slot": 0, "position": [0.0, 0.0, 0.0, 1.0], "render_color": [1.0, 1.0, 1.0, 1.0], "render_data": [1.0, 4.0, 0.0, 0.0]}],
                  "expected": [{"position": [0.0, 0.0], "scale": [1.0, 1.0], "region": [0.75, 0.0, 1.0, 1.0], "rotation": 0.0, "sort_order": 0.0, "color": [255, 255, 255, 255]}]})

    texel = [0.5, 0.25, 1.0, 7.0]
    # (d) HDRMode.None: rgb = pow(max(0, rgb * inverseScale + offset) * exposure, gamma), alpha 1
    r = [max(0.0, v * 2.0 - 0.1) * 1.5 for v in texel[:3]]
    cases.append({"kind": "resolve", "texel": texel, "hdr": {"mode": 0, "inverse_scale": 2.0, "offset": -0.1, "exposure": 1.5, "gamma": 2.0},
                  "expected": [r[0] ** 2, r[1] ** 2, r[2] ** 2, 1.0]})
    #     InverseScaleFactor 0 means 1 (LightingRenderer.cs:1468-1472); gamma is clamped to [.1, 4], exposure to [1/256, 99999]
    r = [max(0.0, v) * (1.0 / 256.0) for v in texel[:3]]
    cases.append({"kind": "resolve", "texel": texel, "hdr": {"mode": 0, "inverse_scale": 0.0, "offset": 0.0, "exposure": 0.0, "gamma": 9.0},
                  "expected": [r[0] ** 4, r[1] ** 4, r[2] ** 4, 1.0]})
    # (e) GammaCompress (HDR.fxh:11-18): L = dot(rgb, (.299,.587,.114)); s = L * middleGray / average; c = s (1 + s / max^2) / (1 + s); rgb *= c / L
    rgb = [max(v * 1.0 + 0.05, 0.0) for v in texel[:3]]
    L = rgb[0] * 0.299 + rgb[1] * 0.587 + rgb[2] * 0.114
    sL = L * 0.6 / 0.4
    cL = sL * (1 + sL / (2.0 * 2.0)) / (1 + sL)
    cases.append({"kind": "resolve", "texel": texel, "hdr": {"mode": 1, "inverse_scale": 1.0, "offset": 0.05, "middle_gray": 0.6, "average_luminance": 0.4,
                                                             "maximum_luminance": 2.0},
                  "expected": [rgb[0] * cL / L, rgb[1] * cL / L, rgb[2] * cL / L, 1.0]})
    # (f) ToneMap (Resolve.fx:113-139): Uncharted2(max(0, rgb + offset) * exposure) / Uncharted2(whitePoint), then gamma
    def u2(v):
        kA, kB, kC, kD, kE, kF = 0.15, 0.50, 0.10, 0.20, 0.02, 0.30
        return ((v * (kA * v + kC * kB) + kD * kE) / (v * (kA * v + kB) + kD * kF)) - kE / kF
    pre = [max(0.0, v * 1.0 + 0.0) * 2.0 for v in texel[:3]]
    cases.append({"kind": "resolve", "texel": texel, "hdr": {"mode": 2, "inverse_scale": 1.0, "offset": 0.0, "exposure": 2.0, "gamma": 0.8, "white_point": 4.0},
                  "expected": [(u2(pre[0]) / u2(4.0)) ** 0.8, (u2(pre[1]) / u2(4.0)) ** 0.8, (u2(pre[2]) / u2(4.0)) ** 0.8, 1.0]})
    # (g) with albedo (ResolveWithAlbedoCommon, Resolve.fx:43-60): light = texel * (inverseScale * 2); rgb = lerp(albedo, albedo * light.rgb,
    #     saturate(light.a)); alpha = albedo.a; then HDRMode.None.  light.a = 0.2 * 1.5 * 2 = 0.6
    light_texel, albedo = [0.5, 0.25, 1.0, 0.2], [0.8, 0.4, 0.2, 0.5]
    k = 1.5 * 2.0
    t = min(max(light_texel[3] * k, 0.0), 1.0)
    rgb = [a + (a * (l * k) - a) * t for a, l in zip(albedo[:3], light_texel[:3])]
    r = [max(0.0, v + 0.05) * 1.2 for v in rgb]
    cases.append({"kind": "resolve", "texel": light_texel, "albedo": albedo,
                  "hdr": {"mode": 0, "inverse_scale": 1.5, "offset": 0.05, "exposure": 1.2, "gamma": 1.5},
                  "expected": [r[0] ** 1.5, r[1] ** 1.5, r[2] ** 1.5, 0.5]})
    #     light alpha above 1 saturates: rgb = albedo * light.rgb exactly; tone-mapped afterwards, alpha still the albedo's
    light_texel, albedo = [0.5, 0.25, 1.0, 7.0], [0.8, 0.4, 0.2, 0.25]
    rgb = [a * (l * 2.0) for a, l in zip(albedo[:3], light_texel[:3])]
    pre = [max(0.0, v) * 2.0 for v in rgb]
    cases.append({"kind": "resolve", "texel": light_texel, "albedo": albedo,
                  "hdr": {"mode": 2, "inverse_scale": 1.0, "offset": 0.0, "exposure": 2.0, "gamma": 0.8, "white_point": 4.0},
                  "expected": [(u2(pre[0]) / u2(4.0)) ** 0.8, (u2(pre[1]) / u2(4.0)) ** 0.8, (u2(pre[2]) / u2(4.0)) ** 0.8, 0.25]})
    #     ... and gamma-compressed (GammaCompress keeps color.a, HDR.fxh:17): light alpha 0 leaves the albedo unlit
    light_texel, albedo = [3.0, 3.0, 3.0, 0.0], [0.5, 0.25, 1.0, 0.75]
    rgb = [max(v + 0.05, 0.0) for v in albedo[:3]]
    L = rgb[0] * 0.299 + rgb[1] * 0.587 + rgb[2] * 0.114
    sL = L * 0.6 / 0.4
    cL = sL * (1 + sL / (2.0 * 2.0)) / (1 + sL)
    cases.append({"kind": "resolve", "texel": light_texel, "albedo": albedo,
                  "hdr": {"mode": 1, "inverse_scale": 1.0, "offset": 0.05, "middle_gray": 0.6, "average_luminance": 0.4, "maximum_luminance": 2.0},
                  "expected": [rgb[0] * cL / L, rgb[1] * cL / L, rgb[2] * cL / L, 0.75]})
    return {"source": "ParticleReadback.cs:73-167, Resolve.fx:25-233, HDR.fxh:1-44, IlluminantMaterials.cs:81-137, LightingRenderer.cs:1463-1580 "
                      "(hand-evaluated; see comments in make_golden.py)", "tolerance": "1e-5 relative; colour bytes and record order exact", "cases": cases}


# ---------------------------------------------------------------------------------------------------------
# 12. Particle rasterisation: closed forms from RasterizeParticleSystem.fx:61-163,228-241 (technique
#     RasterizeParticlesNoTexture), Uniforms.cs:238-290.  Pixel (x, y) belongs to a quad when its centre (x + .5, y + .5) maps to
#     unit coordinates in [-1, 1) x [-1, 1); blending is BlendState.AlphaBlend on premultiplied colour unless "additive".
# ---------------------------------------------------------------------------------------------------------
def gen_rasterize():
    cases = []
    clear = [0.1, 0.1, 0.1, 1.0]
    over = lambda src, dst: [src[k] + dst[k] * (1.0 - src[3]) for k in range(4)]
    add = lambda src, dst: [src[k] + dst[k] for k in range(4)]
    sq = lambda x0, x1, y0, y1: [[x, y] for y in range(y0, y1 + 1) for x in range(x0, x1 + 1)]
    # (a) axis-aligned square of half size 3 around (10, 8): centres with |d| < 3 => x 7..12, y 5..10
    src = [0.2, 0.4, 0.6, 0.8]
    cases.append({"name": "axis-aligned square", "width": 24, "height": 20, "clear": clear, "live_quads": 1, "shaded_pixels": 36,
                  "particles": [{"slot": 5, "position": [10.0, 8.0, 0.0, 1.0], "size": 3.0, "rotation": 0.0, "color": src}],
                  "pixels": [{"x": 7, "y": 5, "rgba": over(src, clear)}, {"x": 12, "y": 10, "rgba": over(src, clear)},
                             {"x": 6, "y": 5, "rgba": clear}, {"x": 13, "y": 10, "rgba": clear}],
                  "covered": sq(7, 12, 5, 10)})
    # (b) the half-open rule: centre on a pixel-centre lattice point (10.5, 8.5) => d = x - 10 in [-3, 3): the left / top edge pixel
    #     (d = -3) is in, the right / bottom one (d = +3) is out
    cases.append({"name": "top-left rule", "width": 24, "height": 20, "clear": clear, "live_quads": 1, "shaded_pixels": 36,
                  "particles": [{"slot": 0, "position": [10.5, 8.5, 0.0, 1.0], "size": 3.0, "rotation": 0.0, "color": src}],
                  "pixels": [{"x": 7, "y": 5, "rgba": over(src, clear)}, {"x": 13, "y": 8, "rgba": clear}, {"x": 10, "y": 11, "rgba": clear}],
                  "covered": sq(7, 12, 5, 10)})
    # (c) rotation by pi / 2 swaps the extents: size (2, 1) * 2 = (4, 2) => 2 wide, 4 high half extents around (10.25, 8.25)
    cases.append({"name": "quarter turn", "width": 24, "height": 20, "clear": clear, "live_quads": 1, "shaded_pixels": 32,
                  "params": {"size": [2.0, 1.0]},
                  "particles": [{"slot": 9, "position": [10.25, 8.25, 0.0, 1.0], "size": 2.0, "rotation": math.pi / 2, "color": src}],
                  "pixels": [{"x": 8, "y": 4, "rgba": over(src, clear)}, {"x": 11, "y": 11, "rgba": over(src, clear)}, {"x": 7, "y": 8, "rgba": clear},
                             {"x": 12, "y": 8, "rgba": clear}],
                  "covered": sq(8, 11, 4, 11)})
    # (d) draw order = slot order: half-transparent red (slot 3) under half-transparent green (slot 7)
    red, green = [0.5, 0.0, 0.0, 0.5], [0.0, 0.5, 0.0, 0.5]
    black = [0.0, 0.0, 0.0, 0.0]
    two = [{"slot": 7, "position": [12.0, 10.0, 0.0, 1.0], "size": 3.0, "rotation": 0.0, "color": green},
           {"slot": 3, "position": [10.0, 10.0, 0.0, 1.0], "size": 3.0, "rotation": 0.0, "color": red}]
    cases.append({"name": "slot order under alpha blending", "width": 24, "height": 20, "clear": black, "live_quads": 2, "shaded_pixels": 72,
                  "particles": two,
                  "pixels": [{"x": 10, "y": 10, "rgba": over(green, over(red, black))}, {"x": 7, "y": 10, "rgba": red}, {"x": 14, "y": 10, "rgba": green}]})
    cases.append({"name": "additive", "width": 24, "height": 20, "clear": black, "live_quads": 2, "shaded_pixels": 72,
                  "params": {"additive": True}, "particles": two,
                  "pixels": [{"x": 10, "y": 10, "rgba": add(green, add(red, black))}, {"x": 7, "y": 10, "rgba": red}]})
    # (e) computeCircularAlpha with rounding power .5: alpha 1 up to distance .5, 1 - sqrt((d - .5) / .5) beyond, 0 from d = 1:
    #     the quad's corners are discarded (alpha <= 0), so fewer pixels are shaded than covered
    white = [1.0, 1.0, 1.0, 1.0]
    a75 = 1.0 - math.sqrt(0.5)
    cases.append({"name": "rounded", "width": 34, "height": 34, "clear": black, "live_quads": 1,
                  "params": {"rounded": True, "rounding": 0.5},
                  "particles": [{"slot": 1, "position": [16.5, 16.5, 0.0, 1.0], "size": 8.0, "rotation": 0.0, "color": white}],
                  "pixels": [{"x": 16, "y": 16, "rgba": white}, {"x": 20, "y": 16, "rgba": white}, {"x": 22, "y": 16, "rgba": [a75] * 4},
                             {"x": 23, "y": 23, "rgba": black}, {"x": 9, "y": 9, "rgba": black}]})
    # (f) origin / scale / viewport / global colour: display = p * Scale + origin = (13, 11); pixel = (display - (1, 1)) * 2 = (24, 20);
    #     half extent 1.5 * 2 * 2 = 6 px; GlobalColor (.5, 1, 1, .5) is premultiplied by the uniform's constructor
    g = [0.5 * 0.5, 1.0 * 0.5, 1.0 * 0.5, 0.5]
    cases.append({"name": "transforms and global colour", "width": 40, "height": 32, "clear": black, "live_quads": 1, "shaded_pixels": 144,
                  "params": {"global_color": [0.5, 1.0, 1.0, 0.5], "origin": [3.0, 1.0], "scale": [2.0, 2.0], "viewport_scale": [2.0, 2.0],
                             "viewport_position": [1.0, 1.0]},
                  "particles": [{"slot": 2, "position": [5.0, 5.0, 0.0, 1.0], "size": 1.5, "rotation": 0.0, "color": white}],
                  "pixels": [{"x": 18, "y": 14, "rgba": g}, {"x": 29, "y": 25, "rgba": g}, {"x": 17, "y": 14, "rgba": black}, {"x": 30, "y": 25, "rgba": black}],
                  "covered": sq(18, 29, 14, 25)})
    # (g) z: display.y = y - z * ZToY = 12 - 4 * .5 = 10; size * max(0, 1 + z * SizeFromZ) = 2 * (1 + 4 * .25) = 4
    cases.append({"name": "z to y and size from z", "width": 24, "height": 20, "clear": black, "live_quads": 1, "shaded_pixels": 64,
                  "params": {"z_to_y": 0.5, "size_from_z": 0.25},
                  "particles": [{"slot": 4, "position": [10.0, 12.0, 4.0, 1.0], "size": 2.0, "rotation": 0.0, "color": white}],
                  "pixels": [{"x": 6, "y": 6, "rgba": white}, {"x": 13, "y": 13, "rgba": white}, {"x": 10, "y": 14, "rgba": black}],
                  "covered": sq(6, 13, 6, 13)})
    # (h) dead particles and empty quads draw nothing; a transparent one is discarded per pixel (alpha <= 0)
    cases.append({"name": "dead, empty, transparent", "width": 24, "height": 20, "clear": clear, "live_quads": 1, "shaded_pixels": 0,
                  "particles": [{"slot": 0, "position": [10.0, 8.0, 0.0, 0.0], "size": 3.0, "rotation": 0.0, "color": white},
                                {"slot": 1, "position": [10.0, 8.0, 0.0, 1.0], "size": 0.0, "rotation": 0.0, "color": white},
                                {"slot": 2, "position": [10.0, 8.0, 0.0, 1.0], "size": 3.0, "rotation": 0.0, "color": [0.3, 0.3, 0.3, 0.0]}],
                  "pixels": [{"x": 10, "y": 8, "rgba": clear}], "covered": []})
    # (i) technique TexturePoint on a 4 x 2 bitmap: RelativeSize => SizeFactor = SizePx / 2 = (2, 1); RenderData.x = 4 => half extents
    #     (8, 4) around (16, 12): a 16 x 8-pixel quad, each texel a 4 x 4-pixel block.  Fragment = colour x texel x GlobalColor.
    texel = lambda i, j: [0.1 + 0.2 * i, 0.2 + 0.5 * j, 1.0 - 0.2 * i, 1.0]
    bitmap = [[texel(i, j) for i in range(4)] for j in range(2)]
    col = [0.5, 1.0, 0.5, 1.0]
    mul = lambda a, b: [a[k] * b[k] for k in range(4)]
    cases.append({"name": "point-sampled bitmap", "width": 32, "height": 24, "clear": black, "live_quads": 1, "shaded_pixels": 128,
                  "bitmap": bitmap, "params": {"bilinear": False},
                  "particles": [{"slot": 0, "position": [16.0, 12.0, 0.0, 1.0], "size": 4.0, "rotation": 0.0, "color": col}],
                  "pixels": [{"x": 8, "y": 8, "rgba": mul(col, texel(0, 0))}, {"x": 11, "y": 11, "rgba": mul(col, texel(0, 0))},
                             {"x": 12, "y": 8, "rgba": mul(col, texel(1, 0))}, {"x": 23, "y": 15, "rgba": mul(col, texel(3, 1))},
                             {"x": 20, "y": 12, "rgba": mul(col, texel(3, 1))}, {"x": 7, "y": 8, "rgba": black}],
                  "covered": sq(8, 23, 8, 15)})
    # (j) frame sheet: an 8 x 2 bitmap of four 2 x 2 frames in a row (frame rectangle 2 x 2 px at offset 0): AnimationRate .5 => the
    #     uniform is 2 => frame column floor(2 * life) % 4 = floor(2.6) = 2 for life 1.3; a negative rate plays the sheet backwards:
    #     (4 - 2) - 1 = 1.  The quad (half extents SizePx / 2 * 3 = (3, 3)) shows texels (2 f, 0) .. (2 f + 1, 1) of frame f.
    sheet = [[[0.1 * i, 0.3 + 0.4 * j, 0.05 * i * (j + 1), 1.0] for i in range(8)] for j in range(2)]
    for (name, rate, frame) in (("frame from life", 0.5, 2), ("frame from life, reversed", -0.5, 1)):
        cases.append({"name": name, "width": 24, "height": 20, "clear": black, "live_quads": 1, "shaded_pixels": 36,
                      "bitmap": sheet, "params": {"bilinear": False, "size_px": [2.0, 2.0], "animation_rate": [rate, 0.0]},
                      "particles": [{"slot": 3, "position": [10.0, 8.0, 0.0, 1.3], "size": 3.0, "rotation": 0.0, "color": white}],
                      "pixels": [{"x": 7, "y": 5, "rgba": sheet[0][2 * frame]}, {"x": 12, "y": 5, "rgba": sheet[0][2 * frame + 1]},
                                 {"x": 7, "y": 10, "rgba": sheet[1][2 * frame]}, {"x": 12, "y": 10, "rgba": sheet[1][2 * frame + 1]}]})
    # (k) row of the sheet from RenderData.w (the particle's type): a 2 x 4 bitmap of two 2 x 2 frames stacked vertically
    tall = [[[0.2 + 0.1 * i, 0.1 * j, 0.5, 1.0] for i in range(2)] for j in range(4)]
    cases.append({"name": "row from the particle type", "width": 24, "height": 20, "clear": black, "live_quads": 1, "shaded_pixels": 36,
                  "bitmap": tall, "params": {"bilinear": False, "size_px": [2.0, 2.0]},
                  "particles": [{"slot": 1, "position": [10.0, 8.0, 0.0, 1.0], "size": 3.0, "rotation": 0.0, "color": white, "row": 1.0}],
                  "pixels": [{"x": 7, "y": 5, "rgba": tall[2][0]}, {"x": 12, "y": 10, "rgba": tall[3][1]}]})
    # (l) technique TextureLinear: a 2 x 1 bitmap black -> white; the pixel whose centre is the quad's centre has u = 0 => texCoord .5 =>
    #     halfway between the two texel centres
    ramp = [[[0.0, 0.0, 0.0, 1.0], [1.0, 1.0, 1.0, 1.0]]]
    cases.append({"name": "bilinear bitmap", "width": 24, "height": 20, "clear": black, "live_quads": 1,
                  "bitmap": ramp, "params": {"bilinear": True, "relative_size": False},
                  "particles": [{"slot": 0, "position": [10.5, 8.5, 0.0, 1.0], "size": 4.0, "rotation": 0.0, "color": white}],
                  "pixels": [{"x": 10, "y": 8, "rgba": [0.5, 0.5, 0.5, 1.0]}, {"x": 7, "y": 8, "rgba": [0.0, 0.0, 0.0, 1.0]},
                             {"x": 13, "y": 8, "rgba": [1.0, 1.0, 1.0, 1.0]}, {"x": 11, "y": 8, "rgba": [0.75, 0.75, 0.75, 1.0]}]})
    return {"source": "hand-derived from RasterizeParticleSystem.fx:61-163,191-241, Uniforms.cs:230-290 (see comments in make_golden.py)",
            "tolerance": "1e-5 relative", "cases": cases}


def main():
    out = {"rasterize.json": gen_rasterize(), "output_ext.json": gen_output_ext(), "lights_ext.json": gen_lights_ext(), "transforms_ext.json": gen_transforms_ext(), "distance_field_generation.json": gen_distance_field_generation(), "bezier.json": gen_bezier(), "distance_field_layout.json": gen_layout(), "spawner.json": gen_spawner(),
           "liveness.json": gen_liveness(), "distance_encoding.json": gen_encoding(), "gbuffer.json": gen_gbuffer(),
           "closed_form.json": gen_closed_form()}
    for name, doc in out.items():
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(doc, f, indent=1, sort_keys=True)
            f.write("\n")
        print("%-28s %d cases" % (name, len(doc["cases"])))


if __name__ == "__main__":
    main()
