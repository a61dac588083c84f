"""Pins the CPU oracle (and the host mirror's pure logic) on the committed known-answer vectors of tests/golden/.

The vectors come from tests/golden/make_golden.py: Python restatements of the reference's C# CPU mirrors
(Bezier.cs, DistanceField.cs, ParticleSpawner.cs, ParticleSpawning.cs, ParticleEngine.cs) and closed-form
answers derived by hand from the cited HLSL lines -- a second source, independent of the HLSL the oracle
restates.  The reference has no tests of its own for these paths ("parity unpinned"); this is the
substitute pin.  Runs without a GPU.
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from illuminant_amd import abi, scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def test_fixtures_are_reproducible(tmp_path):
    """make_golden.py regenerates the committed JSON byte for byte (the vectors are data + their generator)."""
    # (reference_constants.json has its own generator, tools/pin_reference_constants.py, checked by tests/test_reference_pin.py)
    # (full_frame_bands.json is the oracle over two whole frames: five minutes of CPU, its own generator make_full_frame_bands.py; the GPU
    # suite checks every band of it against the kernel, tests/test_full_frame_bands_gpu.py)
    own_generator = ("reference_constants.json", "full_frame_bands.json")
    before = {n: open(os.path.join(GOLDEN, n), "rb").read() for n in os.listdir(GOLDEN) if n.endswith(".json") and n not in own_generator}
    assert len(before) >= 7
    # run a copy of the generator in a scratch directory so the committed files are never rewritten by the test
    script = open(os.path.join(GOLDEN, "make_golden.py")).read()
    scratch = tmp_path / "make_golden.py"
    scratch.write_text(script)
    subprocess.run([sys.executable, str(scratch)], check=True, capture_output=True)
    for n, data in before.items():
        assert (tmp_path / n).read_bytes() == data, n


# ---- Bezier.cs C# mirror vs the oracle's Bezier.fxh restatement ------------------------------------------------------

def test_bezier_matches_csharp_mirror(oracle):
    doc = load("bezier.json")
    n1 = n4 = 0
    for c in doc["cases"]:
        rc = abi.f4(*c["range_and_count"])
        if c["kind"] == "bezier1":
            b = abi.ClampedBezier1()
            b.RangeAndCount = rc
            b.ABCD = abi.f4(*c["abcd"])
            got = [oracle.bezier1(b, c["value"])]
            n1 += 1
        else:
            b = abi.ClampedBezier4()
            b.RangeAndCount = rc
            b.A, b.B, b.C, b.D = abi.f4(*c["a"]), abi.f4(*c["b"]), abi.f4(*c["c"]), abi.f4(*c["d"])
            got = list(oracle.bezier4(b, c["value"]))
            n4 += 1
        np.testing.assert_allclose(got, c["expected"], rtol=1e-6, atol=1e-6, err_msg=json.dumps(c))
    assert n1 > 500 and n4 > 500


# ---- DistanceField ctor layout (oracle, Python scene builder, C++ host mirror) ---------------------------------------

LAYOUT_KEYS = ("slice_width", "slice_height", "slice_count", "physical_slice_count", "column_count", "row_count", "atlas_width", "atlas_height")


def test_distance_field_layout_kats(oracle):
    doc = load("distance_field_layout.json")
    # the three layouts quoted in SURVEY.md 8c, written out here so a reader sees the numbers
    hand = {(512, 512, 32, 1.0): (512, 512, 33, 11, 3, 4, 1536, 2048),
            (1920, 1080, 9, 0.25): (480, 270, 9, 3, 2, 2, 960, 540),
            (256, 256, 9, 1.0): (256, 256, 9, 3, 2, 2, 512, 512)}
    seen = 0
    for c in doc["cases"]:
        key = (c["virtual_width"], c["virtual_height"], c["requested_slice_count"], c["requested_resolution"])
        exp = c["expected"]
        if key in hand:
            assert tuple(exp[k] for k in LAYOUT_KEYS) == hand[key]
            seen += 1
        lay = oracle.distance_field_layout(key[0], key[1], 64.0, key[2], key[3])
        assert tuple(getattr(lay, k) for k in LAYOUT_KEYS) == tuple(exp[k] for k in LAYOUT_KEYS), key
        assert abs(lay.resolution - exp["resolution"]) < 1e-12
        # the scene builder used by tests and bench computes the same layout
        sl = scenes.DistanceFieldLayout(key[0], key[1], 64.0, key[2], key[3])
        assert (sl.slice_width, sl.slice_height, sl.slice_count, sl.physical_slice_count, sl.column_count, sl.row_count,
                sl.atlas_width, sl.atlas_height) == tuple(exp[k] for k in LAYOUT_KEYS), key
    assert seen == 3


def test_host_mirror_distance_field_layout():
    from illuminant_amd import _host as H
    for c in load("distance_field_layout.json")["cases"]:
        L = H.DistanceField.ComputeLayout(c["virtual_width"], c["virtual_height"], c["requested_slice_count"], c["requested_resolution"])
        exp = c["expected"]
        got = (L.SliceWidth, L.SliceHeight, L.SliceCount, L.PhysicalSliceCount, L.ColumnCount, L.RowCount, L.TextureWidth, L.TextureHeight)
        assert got == tuple(exp[k] for k in LAYOUT_KEYS), c
        assert abs(L.Resolution - exp["resolution"]) < 1e-12


def test_distance_field_uniforms_packing(oracle):
    """Uniforms.DistanceField ctor (Uniforms.cs:90-110) + DistanceFieldPacked1 (LightingRenderer.cs:1933-1939) on the cfg3 field."""
    lay = oracle.distance_field_layout(2048, 2048, 128.0, 32, 0.25)
    u = oracle.distance_field_uniforms(lay, max_cone_radius=24.0, power=0.7, step_limit=64, min_step_size=1.0, long_step_factor=0.5)
    assert (u.Extent.x, u.Extent.y, u.Extent.z, u.Extent.w) == (2048.0, 2048.0, 128.0, 128.0)
    assert (u.TextureSliceCount.x, u.TextureSliceCount.y, u.TextureSliceCount.w) == (3.0, 4.0, 33.0)
    assert u.TextureSliceCount.z == pytest.approx(128.0, rel=1e-6)          # all 33 slices valid => validZ = depth
    assert u.TextureSliceAndTexelSize.x == pytest.approx(1 / 3, rel=1e-6) and u.TextureSliceAndTexelSize.y == 0.25
    assert u.TextureSliceAndTexelSize.z == pytest.approx(1 / (2048 * 3), rel=1e-6)
    assert u.TextureSliceAndTexelSize.w == pytest.approx(1 / (2048 * 4), rel=1e-6)
    assert u.ConeAndMisc.x == 24.0 and u.ConeAndMisc.z == pytest.approx(0.7) and u.ConeAndMisc.w == 4.0   # InvScaleFactorX = 2048/512
    assert u.StepAndMisc2.x == 64.0 and u.StepAndMisc2.z == 0.5 and u.StepAndMisc2.w == 4.0
    assert u.Packed1.x == pytest.approx(1 / 9, rel=1e-6)                    # 1 / (3 * cols)
    assert u.Packed1.y == pytest.approx(33 / 128, rel=1e-6)                 # sliceCount / extentZ
    assert u.Packed1.z == pytest.approx(128.0, rel=1e-6) and u.Packed1.w == 1.0
    # and the Python scene builder packs the very same bytes
    su = scenes.DistanceFieldLayout(2048, 2048, 128.0, 32, 0.25).uniforms(power=0.7, min_step_size=1.0, long_step_factor=0.5)
    assert bytes(su) == bytes(u)


# ---- spawner tick arithmetic + slot allocation ----------------------------------------------------------------------

def test_spawner_begin_tick_traces(oracle):
    from illuminant_amd import _host as H
    for c in load("spawner.json")["cases"]:
        if c["kind"] != "begin_tick":
            continue
        st = oracle.SpawnerState()
        mt = -1 if c["maximum_total"] is None else c["maximum_total"]
        # the C++ host mirror replays the same scripted draws (count scale 2 = a Spawner with one additional position)
        sp = H.Spawner(1)
        sp.MinRate, sp.MaxRate = c["min_rate"], c["max_rate"]
        if c["maximum_total"] is not None:
            sp.MaximumTotal = c["maximum_total"]
        if c["count_scale"] == 2:
            sp.AdditionalPositions = [[1.0, 2.0, 3.0]]
        sp.ScriptedDraws = [t["draw"] for t in c["ticks"]]
        for t in c["ticks"]:
            n = oracle.spawner_begin_tick(st, c["min_rate"], c["max_rate"], c["count_scale"], t["draw"], c["dt"], mt)
            oracle.spawner_end_tick(st, n, n)
            assert n == t["count"]
            assert st.rate_error == pytest.approx(t["rate_error_after"], abs=1e-9)
            assert st.total_spawned == t["total_spawned_after"]
            hn = sp.BeginTick(0.0, c["dt"])
            sp.EndTick(hn, hn)
            assert hn == t["count"]
            assert sp.RateError == pytest.approx(t["rate_error_after"], abs=1e-9)
            assert sp.TotalSpawned == t["total_spawned_after"]
    # cfg2's spawner: 65536/s at 1/60 s alternates 1092,1092,1092,1093 through the RateError carry
    first = load("spawner.json")["cases"][0]
    assert [t["count"] for t in first["ticks"]][:8] == [1092, 1092, 1092, 1093, 1092, 1092, 1092, 1093]


def replay_allocation(oracle, case):
    """RunSpawner + PickTargetForSpawn (ParticleSpawning.cs:115-231) driven by the oracle's tick arithmetic."""
    cap = case["chunk_capacity"]
    st = oracle.SpawnerState()
    offsets, target, next_id = {}, -1, 1
    out = []
    for tick in case["trace"]:
        issued = []
        for p in range(2):
            req = oracle.spawner_begin_tick(st, case["min_rate"], case["max_rate"], case["count_scale"], tick["draws"][p], case["dt"], -1)
            if req <= 0:
                break
            count = min(req, cap)
            if target != -1 and cap - offsets[target] < 16:
                target = -1
            if target == -1:
                target, next_id = next_id, next_id + 1
                offsets[target] = 0
            count = min(count, cap - offsets[target])
            first = offsets[target]
            offsets[target] += count
            oracle.spawner_end_tick(st, req, count)
            issued.append([target, first, first + count - 1])
            if not req > count:
                break
        out.append((issued, st.rate_error, st.total_spawned))
    return out


def test_spawner_slot_allocation_traces(oracle):
    n = 0
    for c in load("spawner.json")["cases"]:
        if c["kind"] != "allocation":
            continue
        got = replay_allocation(oracle, c)
        for (issued, err, total), want in zip(got, c["trace"]):
            assert issued == want["issued"]                       # slot indices: bit-exact
            assert total == want["total_spawned_after"]
            assert err == pytest.approx(want["rate_error_after"], abs=1e-9)
        # a partial spawn must have happened (chunk roll-over + second pass), otherwise the case pins nothing
        assert any(len(w["issued"]) == 2 for w in c["trace"])
        n += 1
    assert n == 2


# ---- liveness count decode --------------------------------------------------------------------------------------------

def test_liveness_decode(oracle):
    for c in load("liveness.json")["cases"]:
        live = c["live_slots"]
        pos = np.zeros((max(live, 1) + 3, 4), np.float32)
        pos[:live, 3] = 1.0
        assert oracle.count_live(pos, saturate16=True) == c["expected_count"]
        assert oracle.count_live(pos, saturate16=False) == live


# ---- distance encoding + sampling -------------------------------------------------------------------------------------

def test_distance_encoding(oracle):
    for c in load("distance_encoding.json")["cases"]:
        e = oracle.encode_distance(c["distance"], c["max_distance"])
        assert e == pytest.approx(c["encoded"], abs=1e-6)
        assert oracle.decode_distance(e, c["max_distance"]) == pytest.approx(c["decoded"], abs=1e-4)
        assert oracle.decode_distance(e, c["max_distance"]) == pytest.approx(c["distance"], abs=1e-4)
    assert oracle.encode_distance(0.0, 128.0) == pytest.approx(192.0 / 255.0, abs=1e-7)   # DISTANCE_ZERO, DistanceFieldCommon.fxh:8


def test_sample_of_a_constant_field_decodes_to_the_constant(oracle):
    """Every texel of the atlas = encode(d): any bilinear / slice blend returns d (plus the distance to the volume outside it)."""
    lay = scenes.DistanceFieldLayout(256, 256, 64.0, 9, 1.0)
    dfu = lay.uniforms()
    for d in (0.0, 10.0, -20.0):
        q = int(round((192.0 / 255.0 - d / 128.0) * 65535.0))
        atlas = np.full((lay.atlas_height, lay.atlas_width, 4), q, np.uint16)
        tex = oracle.make_texture(atlas, abi.SDF_UNORM16)
        want = (192.0 / 255.0 - q / 65535.0) * 128.0
        for p in ((10.0, 20.0, 5.0), (128.3, 77.7, 31.9), (255.9, 0.1, 63.0)):
            assert oracle.sample_distance_field(p, dfu, tex) == pytest.approx(want, abs=2e-3)
        # 3 units left of the volume, 4 units below it => + 5 (DistanceFieldCommon.fxh:321-327)
        assert oracle.sample_distance_field((-3.0, 50.0, -4.0), dfu, tex) == pytest.approx(want + 5.0, abs=2e-3)


# ---- G-buffer decode ----------------------------------------------------------------------------------------------------

def test_gbuffer_round_trip(oracle):
    env = scenes.environment()
    for c in load("gbuffer.json")["cases"]:
        g = np.zeros((48, 64, 4), np.float32)
        px, py = int(c["pixel"][0]), int(c["pixel"][1])
        g[py, px] = c["texel"]
        env2 = scenes.environment(gbuffer_size=(64, 48))
        wp, n, shadows, fullbright, _cam = oracle.sample_gbuffer(float(px), float(py), env2, oracle.make_texture(g, abi.GBUFFER_FLOAT4))
        exp = c["expected"]
        assert fullbright == exp["fullbright"] and shadows == exp["enable_shadows"]
        np.testing.assert_allclose(n, exp["normal"], atol=1e-5)
        np.testing.assert_allclose(wp[:2], exp["world_xy"], atol=1e-4)
        assert wp[2] == pytest.approx(exp["world_z"], abs=2e-4)
    # ground plane without a G-buffer (LightCommon.fxh:130-141): normal +z, z = GroundZ
    wp, n, shadows, fullbright, _ = oracle.sample_gbuffer(5.0, 7.0, env, None)
    assert tuple(n) == (0.0, 0.0, 1.0) and wp[2] == 0.0 and shadows and not fullbright
    # the encoded ground-plane texel quoted in SURVEY 8c
    assert load("gbuffer.json")["cases"][0]["texel"] == [0.5, 1.0, 0.0, 1.0]


# ---- closed-form answers ------------------------------------------------------------------------------------------------

def one_slot_step(oracle, pos, vel, dt, op=None, update=False, friction=0.0, max_velocity=9999.0, life_decay=1.0):
    cs = 4
    p = np.zeros((cs * cs, 4), np.float32); v = np.zeros_like(p); a = np.ones_like(p)
    p[5], v[5] = pos, vel
    su = scenes.system_uniforms(cs, dt_seconds=dt, friction=friction, max_velocity=max_velocity, life_decay=life_decay)
    return cs, p, v, a, su


def test_closed_form_particles(oracle):
    rnd = scenes.randomness_table(7)
    for c in load("closed_form.json")["cases"]:
        k = c["kind"]
        if k == "gravity_linear":
            cs, p, v, a, su = one_slot_step(oracle, c["position"], c["velocity"], c["dt"])
            at = c["attractor"]
            g = scenes.gravity_params([(tuple(at["position"]), at["radius"], at["strength"], at["type"])], maximum_acceleration=c["maximum_acceleration"])
            oracle.gravity(p, v, cs, su, g)
            np.testing.assert_allclose(v[5], c["expected_velocity"], rtol=1e-5, atol=1e-6)
            np.testing.assert_array_equal(p[5], np.float32(c["position"]))
            assert not v[np.arange(16) != 5].any()                      # dead slots: passthrough
        elif k == "update_positions":
            cs, p, v, a, su = one_slot_step(oracle, c["position"], c["velocity"], c["dt"], friction=c["friction"],
                                            max_velocity=c["max_velocity"], life_decay=c["life_decay"])
            rc = np.zeros_like(p); rd = np.zeros_like(p)
            oracle.update(p, v, a, rc, rd, cs, su, abi.UpdateParams.default())
            np.testing.assert_allclose(p[5], c["expected_position"], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(v[5], c["expected_velocity"], rtol=1e-5, atol=1e-6)
            if c["expected_position"][3] > 0:
                assert rd[5, 2] == pytest.approx(49.5, rel=1e-5)         # RenderData.z = |v| after friction
                assert rd[5, 3] == 3.0                                    # RenderData.w = category
            else:
                assert not rc[5].any() and not rd[5].any()
        elif k == "fma":
            cs, p, v, a, su = one_slot_step(oracle, c["position"], c["velocity"], c["dt"])
            f = scenes.fma_params(scenes.area_none(strength=c["strength"]), cycles_per_second=c["cycles_per_second"],
                                  position_add=tuple(c["position_add"]), position_multiply=tuple(c["position_multiply"]),
                                  velocity_add=tuple(c["velocity_add"]), velocity_multiply=tuple(c["velocity_multiply"]))
            oracle.fma(p, v, cs, su, f)
            np.testing.assert_allclose(p[5], c["expected_position"], rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(v[5], c["expected_velocity"], rtol=1e-5, atol=1e-5)


def test_closed_form_lights(oracle):
    env = scenes.environment()
    dfu = scenes.DistanceFieldLayout(64, 64, 64.0, 3, 1.0).uniforms()
    n = 0
    for c in load("closed_form.json")["cases"]:
        if c["kind"] not in ("light_at_pixel", "linear_ramp"):
            continue
        L = c["light"]
        lights = (abi.LightVertex * 1)(scenes.sphere_light(tuple(L["position"]), L["radius"], L["ramp"], color=tuple(L["color"]), casts_shadows=False))
        img, _ = oracle.render_sphere_lights(lights, env, dfu, None, None, tuple(c["ambient"]), 32, 16)
        px, py = c["pixel"]
        if c["kind"] == "light_at_pixel":
            np.testing.assert_allclose(img[py, px], c["expected"], rtol=1e-6)
        else:
            # the light sits on pixel (4, 4)'s VPOS: the distance to pixel (4 + d, 4) is exactly d
            assert img[py, px, 0] == pytest.approx(c["expected_rgb"], rel=1e-5)
            assert img[py, px, 3] == 2.0
        n += 1
    assert n == 4


def test_oracle_is_test_infrastructure_only():
    """Nothing in the product package may import or link the oracle (it would void every parity claim)."""
    root = os.path.dirname(GOLDEN.rstrip("/"))
    root = os.path.dirname(root)
    bad = []
    for base, _dirs, files in os.walk(os.path.join(root, "illuminant_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h", "Makefile")):
                text = open(os.path.join(base, f), errors="replace").read()
                if "oracle" in text and ("import oracle" in text or "from oracle" in text or "ilm_oracle" in text or "libilm_oracle" in text):
                    bad.append(os.path.join(base, f))
    assert not bad, bad
