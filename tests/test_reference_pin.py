"""Pins the oracle, the kernels and the ABI mirrors to the reference's own text.  No GPU.

The reference has no tests or golden vectors for the two hot paths, and nothing of it can be built in this image: what CAN be
machine-checked is that every number and every struct layout the restatements take from it is the one its sources hold.
tools/pin_reference_constants.py extracts them in the build container (where /root/reference exists) into
tests/golden/reference_constants.json -- names, values, file:line; no source text.  This file checks, against that fixture,

  * the oracle's constants         (oracle/ilm_oracle_constants.h through orc_reference_constant),
  * the kernels' constants         (csrc/reference_constants.hpp through ilm_debug_reference_constant of the shipped library),
  * the header's #defines and the host mirror's defaults,
  * the field order and sizes of the POD mirrors (abi.py, checked against the C header in test_abi_layout.py),

and, when the reference is present, that the fixture is what the script extracts today.
"""
import ctypes as C
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from illuminant_amd import abi, native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_constants.json")))
VALUES = {e["key"]: e for e in FIXTURE["defines"] + FIXTURE["literals"]}


def f32(x):
    return float(np.float32(x))


def test_fixture_covers_both_paths():
    assert len(FIXTURE["defines"]) >= 30 and len(FIXTURE["literals"]) >= 20 and len(FIXTURE["structs"]) == 8
    for must in ("ConeTrace.fxh:FULLY_SHADOWED_THRESHOLD", "ConeTrace.fxh:MIN_CONE_RADIUS", "SphereLightCore.fxh:SELF_OCCLUSION_HACK",
                 "UpdateParticleSystemWithDistanceField.fx:BOUNCE_DELAY", "Gravity.fx:MAX_ATTRACTORS", "DistanceFieldCommon.fxh:DISTANCE_ZERO",
                 "SpawnerCommon.fxh:randomOffset3.y modulus", "CountLiveParticles.fx:count increment denominator"):
        assert must in VALUES
    for e in VALUES.values():
        assert e["file"].startswith("Illuminant/") and e["line"] > 0


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference checkout only exists in the build container")
def test_fixture_is_what_the_reference_says_today():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pin_reference_constants.py"), "--check"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr


def check_table(table, who):
    assert len(table) >= 35
    for key, value in table.items():
        assert key in VALUES, "%s uses a constant the fixture does not hold: %s" % (who, key)
        want = VALUES[key]["value"]
        # the code holds float32 constants (HLSL literals are floats); the fixture holds the literal's value in double
        assert f32(value) == f32(want), "%s: %s is %r, the reference (%s:%d) says %r" % (who, key, value, VALUES[key]["file"], VALUES[key]["line"], want)


def test_oracle_constants_are_the_references(oracle):
    check_table(oracle.reference_constants(), "oracle/ilm_oracle_constants.h")


def test_kernel_constants_are_the_references():
    check_table(native.reference_constants(), "csrc/reference_constants.hpp")
    lib = native.lib()
    v = C.c_double()
    assert lib.ilm_debug_reference_constant(b"ConeTrace.fxh:NOT_A_CONSTANT", C.byref(v)) == abi.ERR_OUT_OF_RANGE


def test_both_lists_cover_every_shader_constant_of_the_paths(oracle):
    """Every #define the fixture extracts for the restated shaders is used (by name) by the oracle AND by the kernels -- except
    the formula-type ids, which travel as data (IlmSpawnParams.FormulaTypes)."""
    needed = {e["key"] for e in FIXTURE["defines"] if "FormulaType_" not in e["key"]}
    needed |= {e["key"] for e in FIXTURE["literals"] if e["file"].endswith((".fx", ".fxh"))}
    assert needed <= set(oracle.reference_constants()), sorted(needed - set(oracle.reference_constants()))
    assert needed <= set(native.reference_constants()), sorted(needed - set(native.reference_constants()))


def test_changing_a_number_in_the_oracle_alone_is_caught(tmp_path, oracle):
    """The point of the pin: FULLY_SHADOWED_THRESHOLD edited in a COPY of the oracle's constants no longer matches the fixture."""
    src = os.path.join(ROOT, "oracle")
    for name in os.listdir(src):
        if name.endswith((".c", ".h")):
            text = open(os.path.join(src, name)).read()
            if name == "ilm_oracle_constants.h":
                assert "#define CT_FULLY_SHADOWED_THRESHOLD 0.075f" in text
                text = text.replace("#define CT_FULLY_SHADOWED_THRESHOLD 0.075f", "#define CT_FULLY_SHADOWED_THRESHOLD 0.08f")
            text = text.replace('#include "../include/illuminant_hip.h"', '#include "%s"' % os.path.join(ROOT, "include", "illuminant_hip.h"))
            (tmp_path / name).write_text(text)
    so = tmp_path / "liborc_edited.so"
    subprocess.run(["gcc", "-O0", "-std=gnu11", "-ffp-contract=off", "-fopenmp", "-fPIC", "-w", "-shared", "-o", str(so), str(tmp_path / "ilm_oracle.c"), "-lm"],
                   check=True)
    lib = C.CDLL(str(so))
    lib.orc_reference_constant.argtypes = [C.c_char_p, C.POINTER(C.c_double)]
    v = C.c_double()
    assert lib.orc_reference_constant(b"ConeTrace.fxh:FULLY_SHADOWED_THRESHOLD", C.byref(v)) == 1
    with pytest.raises(AssertionError):
        check_table({"ConeTrace.fxh:FULLY_SHADOWED_THRESHOLD": v.value, **{k: x for k, x in oracle.reference_constants().items()
                                                                        if k != "ConeTrace.fxh:FULLY_SHADOWED_THRESHOLD"}}, "edited oracle")


def test_header_defines_and_host_defaults():
    text = open(os.path.join(ROOT, "include", "illuminant_hip.h")).read()
    defines = dict(re.findall(r"#define (ILM_\w+)\s+\(?(-?[\d.]+)[uf]?\)?", text))
    assert float(defines["ILM_MAX_ATTRACTORS"]) == VALUES["Gravity.fx:MAX_ATTRACTORS"]["value"]
    assert float(defines["ILM_MAX_INLINE_POSITION_CONSTANTS"]) == VALUES["SpawnerCommon.fxh:MAX_INLINE_POSITION_CONSTANTS"]["value"]
    assert float(defines["ILM_RANDOMNESS_WIDTH"]) == VALUES["ParticleEngine.cs:RandomnessTextureWidth"]["value"]
    assert float(defines["ILM_RANDOMNESS_HEIGHT"]) == VALUES["ParticleEngine.cs:RandomnessTextureHeight"]["value"]
    assert float(defines["ILM_DISTANCE_LIMIT"]) == VALUES["LightingRenderer.cs:DistanceLimit"]["value"]
    # formula type ids as the scene helpers pack them
    from illuminant_amd import scenes
    assert (scenes.FORMULA_LINEAR, scenes.FORMULA_SPHERICAL, scenes.FORMULA_TOWARDS, scenes.FORMULA_RECTANGULAR) == tuple(
        int(VALUES["SpawnerCommon.fxh:FormulaType_%s" % n]["value"]) for n in ("Linear", "Spherical", "Towards", "Rectangular"))
    assert scenes.CYLINDER_NORMAL_FACTOR == VALUES["LightingRenderer.GBuffer.cs:cylinder normal factor"]["value"]
    # the C++ host mirror's defaults (RendererQualitySettings, ParticleSystem.MaxChunkCount, liveness bookkeeping)
    from illuminant_amd import _host as H
    q = H.RendererQualitySettings()
    assert q.MinStepSize == VALUES["LightingRenderer.Configuration.cs:MinStepSize default"]["value"]
    assert q.LongStepFactor == VALUES["LightingRenderer.Configuration.cs:LongStepFactor default"]["value"]
    assert q.MaxStepCount == VALUES["LightingRenderer.Configuration.cs:MaxStepCount default"]["value"]
    assert q.MaxConeRadius == VALUES["LightingRenderer.Configuration.cs:MaxConeRadius default"]["value"]
    assert q.OcclusionToOpacityPower == VALUES["LightingRenderer.Configuration.cs:OcclusionToOpacityPower default"]["value"]


def test_pod_mirrors_have_the_references_field_order():
    """[StructLayout(Sequential)] structs of the reference vs the ctypes mirrors (whose offsets test_abi_layout.py checks against
    the C header): same fields, same order, packed (Pack = 4: a VectorN is N floats at the running offset)."""
    by_name = {"IlmEnvironment": abi.Environment, "IlmDistanceFieldUniforms": abi.DistanceFieldUniforms,
               "IlmParticleSystemUniforms": abi.ParticleSystemUniforms, "IlmLightVertex": abi.LightVertex,
               "IlmClampedBezier1": abi.ClampedBezier1, "IlmClampedBezier4": abi.ClampedBezier4,
               "IlmHeightVolumeVertex": abi.HeightVolumeVertex, "IlmBillboardVertex": abi.BillboardVertex}
    sizes = {"Vector4": 16, "Vector3": 12, "Vector2": 8, "float": 4}
    for s in FIXTURE["structs"]:
        mirror = by_name[s["mirror"]]
        want = [f["name"] for f in s["fields"]]
        got = [name for name, _ in mirror._fields_][:len(want)]
        assert got == want, (s["struct"], got, want)
        offset = 0
        for f in s["fields"]:
            field = getattr(mirror, f["name"])
            assert (field.offset, field.size) == (offset, sizes[f["type"]]), (s["struct"], f["name"])
            offset += sizes[f["type"]]
        if s["mirror"] in ("IlmHeightVolumeVertex", "IlmBillboardVertex"):
            assert C.sizeof(mirror) == offset
        else:
            assert all(f["type"] == "Vector4" for f in s["fields"]), s["struct"]
    # the mirrors that stop where the reference struct stops
    assert C.sizeof(abi.ParticleSystemUniforms) == 64 and C.sizeof(abi.LightVertex) == 128
    assert C.sizeof(abi.ClampedBezier1) == 32 and C.sizeof(abi.ClampedBezier4) == 80
    assert C.sizeof(abi.HeightVolumeVertex) == 36 and C.sizeof(abi.BillboardVertex) == 48
