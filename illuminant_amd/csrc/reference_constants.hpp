// reference_constants.hpp -- every number the kernels take from the reference's text, by name.
//
// Used where the reference uses its constant; the table at the end maps each to the reference's own name ("<file>:<macro or
// description>") and is served by ilm_debug_reference_constant (api.hip).  tests/test_reference_pin.py compares every entry with
// tests/golden/reference_constants.json, which tools/pin_reference_constants.py extracts from the reference sources in the build
// container: a number changed here alone turns a CPU test red.  (The CPU checker under oracle/ keeps its own, separately typed list, pinned the same way.)
#pragma once

#include "../../include/illuminant_hip.h"

namespace ilm {
namespace ref {

constexpr float kPi = 3.14159265358979323846f;      // ParticleCommon.fxh:23, DistanceFieldCommon.fxh:1
constexpr float kVelocityConstantScale = 1000.0f;   // ParticleCommon.fxh:24, Uniforms.cs:199
constexpr float kDistanceZero = 192.0f / 255.0f;    // DistanceFieldCommon.fxh:8

// ConeTrace.fxh:5-23
constexpr float kMinConeRadius = 0.33f;
constexpr float kMaxStepRampWindow = 2.0f;
constexpr float kTraceInitialOffsetPx = 0.5f;
constexpr float kFullyShadowedThreshold = 0.075f;
constexpr float kUnshadowedThreshold = 0.95f;
constexpr float kHackDistanceOffset = 1.5f;
// SphereLightCore.fxh:10-11
constexpr float kSelfOcclusionHack = 1.6f;
constexpr float kShadowOpacityThreshold = 0.75f / 255.0f;
// LightCommon.fxh:5-10,27-28 (GBufferShaderCommon.fxh:3-4 repeats the two G-buffer ones)
constexpr float kDotOffset = 0.15f;
constexpr float kDotRampRange = 0.15f;
constexpr float kDotExponent = 0.85f;
constexpr float kGBufferZScale = 1024.0f;
constexpr float kGBufferZOffset = 1024.0f;
// UpdateParticleSystemWithDistanceField.fx:14-25
constexpr float kNoNormalThreshold = 0.33f;
constexpr int kMaxStepCount = 3;
constexpr float kBounceDelay = 3.0f;
constexpr float kInitialEscapeSpeed = 0.33f;
constexpr float kEscapeSpeedAcceleration = 1.1f;
// evaluateRandomForIndex, SpawnerCommon.fxh:107-109: index % these
constexpr unsigned kRandom1XModulus = 8039u, kRandom1YModulus = 57u;
constexpr unsigned kRandom2XModulus = 6180u, kRandom2YModulus = 4031u;
constexpr unsigned kRandom3XModulus = 2025u, kRandom3YModulus = 65531u;
// computeRenderData, UpdateCommon.fxh:107: index = x + y * 256 whatever the chunk size
constexpr float kRenderDataIndexRowPitch = 256.0f;
// CountLiveParticles.fx:38 + ParticleEngine.cs:244-247: each live particle adds 1 / 65535 to a 16-bit target
constexpr unsigned kLiveCountSaturation = 65535u;

// G-buffer passes: GBufferBitmap.fx:40,72 (discard thresholds, numerators over 255); GBufferShaderCommon.fxh:14-18 (a dead texel);
// LightingRenderer.GBuffer.cs:275-281 (the ground plane's quad and its lift when RenderGroundPlane is off)
constexpr float kMaskDiscardNumerator = 1.0f;
constexpr float kGDataDiscardNumerator = 127.0f;
constexpr float kDeadTexel = 99999.0f;
constexpr float kGroundHalfExtent = 999999.0f;
constexpr float kGroundLift = 99999.0f;

// premultipliedToDithered, RasterizeParticleSystem.fx:161: discardThreshold = 6.0 / 255.0
constexpr float kDitherDiscardNumerator = 6.0f;

struct Entry { const char* key; double value; };
constexpr Entry kTable[] = {
    { "ParticleCommon.fxh:PI", kPi }, { "DistanceFieldCommon.fxh:PI", kPi },
    { "ParticleCommon.fxh:VelocityConstantScale", kVelocityConstantScale }, { "Uniforms.cs:VelocityConstantScale", kVelocityConstantScale },
    { "DistanceFieldCommon.fxh:DISTANCE_ZERO", kDistanceZero },
    { "ConeTrace.fxh:MIN_CONE_RADIUS", kMinConeRadius }, { "ConeTrace.fxh:MAX_STEP_RAMP_WINDOW", kMaxStepRampWindow },
    { "ConeTrace.fxh:TRACE_INITIAL_OFFSET_PX", kTraceInitialOffsetPx }, { "ConeTrace.fxh:FULLY_SHADOWED_THRESHOLD", kFullyShadowedThreshold },
    { "ConeTrace.fxh:UNSHADOWED_THRESHOLD", kUnshadowedThreshold }, { "ConeTrace.fxh:HACK_DISTANCE_OFFSET", kHackDistanceOffset },
    { "SphereLightCore.fxh:SELF_OCCLUSION_HACK", kSelfOcclusionHack }, { "SphereLightCore.fxh:SHADOW_OPACITY_THRESHOLD", kShadowOpacityThreshold },
    { "LightCommon.fxh:DOT_OFFSET", kDotOffset }, { "LightCommon.fxh:DOT_RAMP_RANGE", kDotRampRange }, { "LightCommon.fxh:DOT_EXPONENT", kDotExponent },
    { "LightCommon.fxh:GBUFFER_Z_SCALE", kGBufferZScale }, { "LightCommon.fxh:GBUFFER_Z_OFFSET", kGBufferZOffset },
    { "GBufferShaderCommon.fxh:GBUFFER_Z_SCALE", kGBufferZScale }, { "GBufferShaderCommon.fxh:GBUFFER_Z_OFFSET", kGBufferZOffset },
    { "UpdateParticleSystemWithDistanceField.fx:NO_NORMAL_THRESHOLD", kNoNormalThreshold },
    { "UpdateParticleSystemWithDistanceField.fx:MAX_STEP_COUNT", kMaxStepCount },
    { "UpdateParticleSystemWithDistanceField.fx:BOUNCE_DELAY", kBounceDelay },
    { "UpdateParticleSystemWithDistanceField.fx:INITIAL_ESCAPE_SPEED", kInitialEscapeSpeed },
    { "UpdateParticleSystemWithDistanceField.fx:ESCAPE_SPEED_ACCELERATION", kEscapeSpeedAcceleration },
    { "SpawnerCommon.fxh:randomOffset1.x modulus", kRandom1XModulus }, { "SpawnerCommon.fxh:randomOffset1.y modulus", kRandom1YModulus },
    { "SpawnerCommon.fxh:randomOffset2.x modulus", kRandom2XModulus }, { "SpawnerCommon.fxh:randomOffset2.y modulus", kRandom2YModulus },
    { "SpawnerCommon.fxh:randomOffset3.x modulus", kRandom3XModulus }, { "SpawnerCommon.fxh:randomOffset3.y modulus", kRandom3YModulus },
    { "UpdateCommon.fxh:computeRenderData index row pitch", kRenderDataIndexRowPitch },
    { "CountLiveParticles.fx:count increment denominator", kLiveCountSaturation },
    { "GBufferBitmap.fx:mask discard threshold numerator", kMaskDiscardNumerator },
    { "GBufferBitmap.fx:gdata discard threshold numerator", kGDataDiscardNumerator },
    { "GBufferShaderCommon.fxh:dead texel value", kDeadTexel },
    { "LightingRenderer.GBuffer.cs:ground plane half extent", kGroundHalfExtent },
    { "LightingRenderer.GBuffer.cs:ground plane lift", kGroundLift },
    { "RasterizeParticleSystem.fx:dither discard threshold numerator", kDitherDiscardNumerator },
    { "Gravity.fx:MAX_ATTRACTORS", ILM_MAX_ATTRACTORS },
    { "SpawnerCommon.fxh:MAX_INLINE_POSITION_CONSTANTS", ILM_MAX_INLINE_POSITION_CONSTANTS },
    { "ParticleEngine.cs:RandomnessTextureWidth", ILM_RANDOMNESS_WIDTH }, { "ParticleEngine.cs:RandomnessTextureHeight", ILM_RANDOMNESS_HEIGHT },
    { "LightingRenderer.cs:DistanceLimit", ILM_DISTANCE_LIMIT },
};
constexpr int kTableSize = (int)(sizeof(kTable) / sizeof(kTable[0]));

}  // namespace ref
}  // namespace ilm
