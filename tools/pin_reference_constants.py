#!/usr/bin/env python3
"""Pin the oracle, the kernels and the ABI mirrors to the reference's own TEXT (build container only: reads /root/reference).

The reference ships no tests or golden vectors and cannot be built here, so its numbers used to be typed twice by hand -- once into
oracle/, once into the kernels.  This script extracts every `#define` constant, C# `const`, shader-literal and default the two hot
paths use, and the field order of the structs the ABI mirrors, from the reference sources, and writes them -- names, values and
file:line, no source text -- to tests/golden/reference_constants.json.  tests/test_reference_pin.py (CPU) then checks

    oracle/ilm_oracle_constants.h   (orc_reference_constant)          against the JSON,
    csrc/reference_constants.hpp    (ilm_debug_reference_constant)    against the JSON,
    illuminant_amd/abi.py and include/illuminant_hip.h                against the struct field orders,

and, when /root/reference is present, that the JSON is what this script extracts today.

    python tools/pin_reference_constants.py            writes the fixture
    python tools/pin_reference_constants.py --check    exit 1 if the committed fixture differs from a fresh extraction
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden", "reference_constants.json")

# (file, macro names) -- every numeric #define the restated code paths read
DEFINES = [
    ("Illuminant/Shaders/ConeTrace.fxh", ["MIN_CONE_RADIUS", "MAX_STEP_RAMP_WINDOW", "TRACE_INITIAL_OFFSET_PX", "FULLY_SHADOWED_THRESHOLD",
                                          "UNSHADOWED_THRESHOLD", "HACK_DISTANCE_OFFSET"]),
    ("Illuminant/Shaders/SphereLightCore.fxh", ["SELF_OCCLUSION_HACK", "SHADOW_OPACITY_THRESHOLD"]),
    ("Illuminant/Shaders/LightCommon.fxh", ["DOT_OFFSET", "DOT_RAMP_RANGE", "DOT_EXPONENT", "GBUFFER_Z_SCALE", "GBUFFER_Z_OFFSET"]),
    ("Illuminant/Shaders/GBufferShaderCommon.fxh", ["GBUFFER_Z_SCALE", "GBUFFER_Z_OFFSET"]),
    ("Illuminant/Shaders/UpdateParticleSystemWithDistanceField.fx", ["NO_NORMAL_THRESHOLD", "MAX_STEP_COUNT", "BOUNCE_DELAY", "INITIAL_ESCAPE_SPEED",
                                                                    "ESCAPE_SPEED_ACCELERATION"]),
    ("Illuminant/Shaders/Gravity.fx", ["MAX_ATTRACTORS"]),
    ("Illuminant/Shaders/DistanceFieldCommon.fxh", ["DISTANCE_ZERO", "PI"]),
    ("Illuminant/Shaders/ParticleCommon.fxh", ["PI", "VelocityConstantScale"]),
    ("Illuminant/Shaders/SpawnerCommon.fxh", ["MAX_INLINE_POSITION_CONSTANTS", "FormulaType_Linear", "FormulaType_Spherical", "FormulaType_Towards",
                                              "FormulaType_Rectangular"]),
]
# (file, name, regex with one group = the literal) -- numbers that are literals in shader / C# code rather than macros
LITERALS = [
    ("Illuminant/Shaders/SpawnerCommon.fxh", "randomOffset1.x modulus", r"randomOffset1 = float2\(index % (\d+),"),
    ("Illuminant/Shaders/SpawnerCommon.fxh", "randomOffset1.y modulus", r"randomOffset1 = float2\(index % \d+, 0 \+ \(index % (\d+)\)\)"),
    ("Illuminant/Shaders/SpawnerCommon.fxh", "randomOffset2.x modulus", r"randomOffset2 = float2\(index % (\d+),"),
    ("Illuminant/Shaders/SpawnerCommon.fxh", "randomOffset2.y modulus", r"randomOffset2 = float2\(index % \d+, 1 \+ \(index % (\d+)\)\)"),
    ("Illuminant/Shaders/SpawnerCommon.fxh", "randomOffset3.x modulus", r"randomOffset3 = float2\(index % (\d+),"),
    ("Illuminant/Shaders/SpawnerCommon.fxh", "randomOffset3.y modulus", r"randomOffset3 = float2\(index % \d+, 2 \+ \(index % (\d+)\)\)"),
    ("Illuminant/Shaders/UpdateCommon.fxh", "computeRenderData index row pitch", r"float index = \S+ \+ \(\S+ \* (\d+)\)"),
    ("Illuminant/Particles/ParticleEngine.cs", "RandomnessTextureWidth", r"RandomnessTextureWidth = (\d+)"),
    ("Illuminant/Particles/ParticleEngine.cs", "RandomnessTextureHeight", r"RandomnessTextureHeight = (\d+)"),
    ("Illuminant/Particles/ParticleSystem.cs", "MaxChunkCount", r"const int MaxChunkCount = (\d+)"),
    ("Illuminant/Particles/ParticleLiveness.cs", "LivenessCheckInterval", r"const int LivenessCheckInterval = (\d+)"),
    ("Illuminant/Particles/ParticleLiveness.cs", "DeadFrameThreshold factor", r"DeadFrameThreshold = LivenessCheckInterval \* (\d+)"),
    ("Illuminant/Uniforms.cs", "VelocityConstantScale", r"const int VelocityConstantScale = (\d+)"),
    ("Illuminant/Lighting/LightingRenderer.cs", "DistanceLimit", r"DistanceLimit = (\d+)"),
    ("Illuminant/Lighting/LightingRenderer.cs", "PackedSliceCount", r"const int PackedSliceCount = (\d+)"),
    ("Illuminant/Lighting/LightingRenderer.Configuration.cs", "MinStepSize default", r"public float MinStepSize\s+= ([\d.]+)f?;"),
    ("Illuminant/Lighting/LightingRenderer.Configuration.cs", "LongStepFactor default", r"public float LongStepFactor\s+= ([\d.]+)f?;"),
    ("Illuminant/Lighting/LightingRenderer.Configuration.cs", "MaxStepCount default", r"public int\s+MaxStepCount\s+= (\d+);"),
    ("Illuminant/Lighting/LightingRenderer.Configuration.cs", "MaxConeRadius default", r"public float MaxConeRadius\s+= ([\d.]+)f?;"),
    ("Illuminant/Lighting/LightingRenderer.Configuration.cs", "ConeGrowthFactor default", r"public float ConeGrowthFactor\s+= ([\d.]+)f?;"),
    ("Illuminant/Lighting/LightingRenderer.Configuration.cs", "OcclusionToOpacityPower default", r"public float OcclusionToOpacityPower = ([\d.]+)f?;"),
    ("Illuminant/Shaders/CountLiveParticles.fx", "count increment denominator", r"color = float4\(1\.0 / (\d+),"),
    ("Illuminant/Shaders/RasterizeParticleSystem.fx", "dither discard threshold numerator", r"premultipliedToDithered[\s\S]*?discardThreshold = \(([\d.]+) / 255\.0\)"),
    ("Illuminant/Shaders/GBufferBitmap.fx", "mask discard threshold numerator", r"MaskBillboardPixelShader[\s\S]*?discardThreshold = \(([\d.]+) / 255\.0\)"),
    ("Illuminant/Shaders/GBufferBitmap.fx", "gdata discard threshold numerator", r"GDataBillboardPixelShader[\s\S]*?discardThreshold = \(([\d.]+) / 255\.0\)"),
    ("Illuminant/Shaders/GBufferShaderCommon.fxh", "dead texel value", r"if \(dead\)[\s\S]*?0, 0,\s*-(\d+),"),
    ("Illuminant/Lighting/LightingRenderer.GBuffer.cs", "ground plane half extent", r"var tl = new Vector3\(-(\d+),"),
    ("Illuminant/Lighting/LightingRenderer.GBuffer.cs", "ground plane lift", r"var huge = new Vector3\(0, 0, (\d+)\)"),
    ("Illuminant/Lighting/LightingRenderer.GBuffer.cs", "cylinder normal factor", r"normal1\.X = 0f - \(([\d.]+)f \* billboard\.CylinderFactor\)"),
]
# (file, struct, ABI mirror) -- field ORDER of the [StructLayout(Sequential)] structs the ABI mirrors
STRUCTS = [
    ("Illuminant/Uniforms.cs", "Environment", "IlmEnvironment"),
    ("Illuminant/Uniforms.cs", "DistanceField", "IlmDistanceFieldUniforms"),
    ("Illuminant/Uniforms.cs", "ParticleSystem", "IlmParticleSystemUniforms"),
    ("Illuminant/Vertices.cs", "LightVertex", "IlmLightVertex"),
    ("Illuminant/Bezier.cs", "ClampedBezier1", "IlmClampedBezier1"),
    ("Illuminant/Bezier.cs", "ClampedBezier4", "IlmClampedBezier4"),
    ("Illuminant/Vertices.cs", "HeightVolumeVertex", "IlmHeightVolumeVertex"),
    ("Illuminant/Vertices.cs", "BillboardVertex", "IlmBillboardVertex"),
]


def evaluate(text):
    """Numeric value of a macro body such as `0.33`, `(0.75 / 255.0)`, `1024`; None when it is not a number."""
    t = text.strip()
    if re.fullmatch(r"[\d\s.()/*+\-eE]+", t) and re.search(r"\d", t):
        return float(eval(t, {"__builtins__": {}}, {}))      # digits, brackets and arithmetic only (checked above)
    return None


def extract():
    out = {"source": "extracted from /root/reference by tools/pin_reference_constants.py (names, values and file:line -- no source text)",
           "defines": [], "literals": [], "structs": []}
    for rel, names in DEFINES:
        lines = open(os.path.join(REF, rel), encoding="utf-8", errors="replace").read().splitlines()
        for name in names:
            hits = [(i + 1, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"\s*#define\s+%s\s+(.+?)\s*(?://.*)?$" % re.escape(name), l)] if m]
            assert len(hits) == 1, (rel, name, hits)
            line, body = hits[0]
            value = evaluate(body)
            assert value is not None, (rel, name, body)
            out["defines"].append({"key": "%s:%s" % (os.path.basename(rel), name), "file": rel, "line": line, "value": value})
    for rel, name, pattern in LITERALS:
        text = open(os.path.join(REF, rel), encoding="utf-8", errors="replace").read()
        ms = list(re.finditer(pattern, text, flags=re.M))
        assert len(ms) >= 1, (rel, name)
        m = ms[0]
        line = text.count("\n", 0, m.start(1)) + 1
        out["literals"].append({"key": "%s:%s" % (os.path.basename(rel), name), "file": rel, "line": line, "value": float(m.group(1))})
    for rel, struct, mirror in STRUCTS:
        text = open(os.path.join(REF, rel), encoding="utf-8", errors="replace").read()
        m = re.search(r"public (?:unsafe )?struct %s\b[^{]*\{" % struct, text)
        assert m, (rel, struct)
        # the struct's body: up to the matching brace
        depth, i = 1, m.end()
        while depth and i < len(text):
            depth += {"{": 1, "}": -1}.get(text[i], 0)
            i += 1
        body = re.sub(r"//[^\n]*", "", text[m.end():i])          # comments carry no fields
        # instance fields at nesting depth 0 of the struct: `public|internal|private Vector4 A, B;` (no static / const, no properties)
        fields, level = [], 0
        for stmt in re.split(r"(?<=[;{}])", body):
            head = stmt.strip()
            if level == 0:
                fm = re.match(r"(?:\[[^\]]*\]\s*)*(?:public|internal|private)?\s*(?!static|const)(Vector[234]|float|int|short|Quaternion)\s+([\w\s,]+);$", head)
                if fm and "(" not in head:
                    fields += [(fm.group(1), n.strip().lstrip("_")) for n in fm.group(2).split(",")]
            level += stmt.count("{") - stmt.count("}")
        assert fields, (rel, struct)
        out["structs"].append({"struct": struct, "mirror": mirror, "file": rel, "line": text.count("\n", 0, m.start()) + 1,
                               "fields": [{"type": t, "name": n} for t, n in fields]})
    return out


def main():
    if not os.path.isdir(REF):
        print("%s is not here: the fixture can only be regenerated in the build container" % REF)
        return 0 if "--check" in sys.argv else 2
    fresh = json.dumps(extract(), indent=1, sort_keys=True) + "\n"
    if "--check" in sys.argv:
        if not os.path.exists(OUT) or open(OUT).read() != fresh:
            print("tests/golden/reference_constants.json differs from a fresh extraction: run python tools/pin_reference_constants.py")
            return 1
        return 0
    with open(OUT, "w") as f:
        f.write(fresh)
    d = json.loads(fresh)
    print("wrote %s: %d defines, %d literals, %d structs" % (OUT, len(d["defines"]), len(d["literals"]), len(d["structs"])))
    return 0


if __name__ == "__main__":
    sys.exit(main())
