/*
 * ilm_oracle_constants.h -- every number the oracle takes from the reference's text, by name.
 *
 * TEST INFRASTRUCTURE ONLY (see ilm_oracle.h).  Each macro is used where the reference uses its constant; the table at the end maps
 * it to the reference's own name ("<file>:<macro or description>"), and tests/test_reference_pin.py compares every entry with
 * tests/golden/reference_constants.json -- which tools/pin_reference_constants.py extracts from the reference sources.  A number that
 * is changed here alone turns that test red.  (The kernels keep their own list, csrc/reference_constants.hpp, checked the same way.)
 */
#ifndef ILM_ORACLE_CONSTANTS_H
#define ILM_ORACLE_CONSTANTS_H

#define H_PI 3.14159265358979323846f            /* ParticleCommon.fxh:23, DistanceFieldCommon.fxh:1, EnvironmentCommon.fxh:4 */
#define VELOCITY_CONSTANT_SCALE 1000.0f          /* ParticleCommon.fxh:24, Uniforms.cs:199 */
#define DISTANCE_ZERO (192.0f / 255.0f)          /* DistanceFieldCommon.fxh:8 */

/* ConeTrace.fxh:5-23 */
#define CT_MIN_CONE_RADIUS 0.33f
#define CT_MAX_STEP_RAMP_WINDOW 2.0f
#define CT_TRACE_INITIAL_OFFSET_PX 0.5f
#define CT_FULLY_SHADOWED_THRESHOLD 0.075f
#define CT_UNSHADOWED_THRESHOLD 0.95f
#define CT_HACK_DISTANCE_OFFSET 1.5f
/* SphereLightCore.fxh:10-11 */
#define SL_SELF_OCCLUSION_HACK 1.6f
#define SL_SHADOW_OPACITY_THRESHOLD (0.75f / 255.0f)
/* LightCommon.fxh:5-10,27-28 (GBufferShaderCommon.fxh:3-4 repeats the two G-buffer ones) */
#define LC_DOT_OFFSET 0.15f
#define LC_DOT_RAMP_RANGE 0.15f
#define LC_DOT_EXPONENT 0.85f
#define GBUFFER_Z_SCALE 1024.0f
#define GBUFFER_Z_OFFSET 1024.0f
/* UpdateParticleSystemWithDistanceField.fx:14-25 */
#define DF_NO_NORMAL_THRESHOLD 0.33f
#define DF_MAX_STEP_COUNT 3
#define DF_BOUNCE_DELAY 3.0f
#define DF_INITIAL_ESCAPE_SPEED 0.33f
#define DF_ESCAPE_SPEED_ACCELERATION 1.1f
/* evaluateRandomForIndex, SpawnerCommon.fxh:107-109: index % these */
#define SP_RANDOM1_X_MODULUS 8039.0f
#define SP_RANDOM1_Y_MODULUS 57.0f
#define SP_RANDOM2_X_MODULUS 6180.0f
#define SP_RANDOM2_Y_MODULUS 4031.0f
#define SP_RANDOM3_X_MODULUS 2025.0f
#define SP_RANDOM3_Y_MODULUS 65531.0f
/* computeRenderData, UpdateCommon.fxh:107: index = x + y * 256 whatever the chunk size */
#define RD_INDEX_ROW_PITCH 256.0f
/* CountLiveParticles.fx:38 + ParticleEngine.cs:244-247: each live particle adds 1 / 65535 to a 16-bit target */
#define LIVE_COUNT_SATURATION 65535u

/* G-buffer passes: GBufferBitmap.fx:40,72 (discard thresholds, numerators over 255); GBufferShaderCommon.fxh:14-18 (a dead texel);
 * LightingRenderer.GBuffer.cs:275-281 (the ground plane's quad and its lift when RenderGroundPlane is off) */
#define GB_MASK_DISCARD_NUMERATOR 1.0f
#define GB_GDATA_DISCARD_NUMERATOR 127.0f
#define GB_DEAD_TEXEL 99999.0f
#define GB_GROUND_HALF_EXTENT 999999.0f
#define GB_GROUND_LIFT 99999.0f

/* premultipliedToDithered, RasterizeParticleSystem.fx:161: discardThreshold = 6.0 / 255.0 */
#define RASTER_DITHER_DISCARD_NUMERATOR 6.0f

struct OrcReferenceConstant { const char* key; double value; };
static const struct OrcReferenceConstant orc_reference_constants[] = {
    { "ParticleCommon.fxh:PI", H_PI }, { "DistanceFieldCommon.fxh:PI", H_PI },
    { "ParticleCommon.fxh:VelocityConstantScale", VELOCITY_CONSTANT_SCALE }, { "Uniforms.cs:VelocityConstantScale", VELOCITY_CONSTANT_SCALE },
    { "DistanceFieldCommon.fxh:DISTANCE_ZERO", DISTANCE_ZERO },
    { "ConeTrace.fxh:MIN_CONE_RADIUS", CT_MIN_CONE_RADIUS }, { "ConeTrace.fxh:MAX_STEP_RAMP_WINDOW", CT_MAX_STEP_RAMP_WINDOW },
    { "ConeTrace.fxh:TRACE_INITIAL_OFFSET_PX", CT_TRACE_INITIAL_OFFSET_PX }, { "ConeTrace.fxh:FULLY_SHADOWED_THRESHOLD", CT_FULLY_SHADOWED_THRESHOLD },
    { "ConeTrace.fxh:UNSHADOWED_THRESHOLD", CT_UNSHADOWED_THRESHOLD }, { "ConeTrace.fxh:HACK_DISTANCE_OFFSET", CT_HACK_DISTANCE_OFFSET },
    { "SphereLightCore.fxh:SELF_OCCLUSION_HACK", SL_SELF_OCCLUSION_HACK }, { "SphereLightCore.fxh:SHADOW_OPACITY_THRESHOLD", SL_SHADOW_OPACITY_THRESHOLD },
    { "LightCommon.fxh:DOT_OFFSET", LC_DOT_OFFSET }, { "LightCommon.fxh:DOT_RAMP_RANGE", LC_DOT_RAMP_RANGE }, { "LightCommon.fxh:DOT_EXPONENT", LC_DOT_EXPONENT },
    { "LightCommon.fxh:GBUFFER_Z_SCALE", GBUFFER_Z_SCALE }, { "LightCommon.fxh:GBUFFER_Z_OFFSET", GBUFFER_Z_OFFSET },
    { "GBufferShaderCommon.fxh:GBUFFER_Z_SCALE", GBUFFER_Z_SCALE }, { "GBufferShaderCommon.fxh:GBUFFER_Z_OFFSET", GBUFFER_Z_OFFSET },
    { "UpdateParticleSystemWithDistanceField.fx:NO_NORMAL_THRESHOLD", DF_NO_NORMAL_THRESHOLD },
    { "UpdateParticleSystemWithDistanceField.fx:MAX_STEP_COUNT", DF_MAX_STEP_COUNT },
    { "UpdateParticleSystemWithDistanceField.fx:BOUNCE_DELAY", DF_BOUNCE_DELAY },
    { "UpdateParticleSystemWithDistanceField.fx:INITIAL_ESCAPE_SPEED", DF_INITIAL_ESCAPE_SPEED },
    { "UpdateParticleSystemWithDistanceField.fx:ESCAPE_SPEED_ACCELERATION", DF_ESCAPE_SPEED_ACCELERATION },
    { "SpawnerCommon.fxh:randomOffset1.x modulus", SP_RANDOM1_X_MODULUS }, { "SpawnerCommon.fxh:randomOffset1.y modulus", SP_RANDOM1_Y_MODULUS },
    { "SpawnerCommon.fxh:randomOffset2.x modulus", SP_RANDOM2_X_MODULUS }, { "SpawnerCommon.fxh:randomOffset2.y modulus", SP_RANDOM2_Y_MODULUS },
    { "SpawnerCommon.fxh:randomOffset3.x modulus", SP_RANDOM3_X_MODULUS }, { "SpawnerCommon.fxh:randomOffset3.y modulus", SP_RANDOM3_Y_MODULUS },
    { "UpdateCommon.fxh:computeRenderData index row pitch", RD_INDEX_ROW_PITCH },
    { "CountLiveParticles.fx:count increment denominator", LIVE_COUNT_SATURATION },
    { "GBufferBitmap.fx:mask discard threshold numerator", GB_MASK_DISCARD_NUMERATOR },
    { "GBufferBitmap.fx:gdata discard threshold numerator", GB_GDATA_DISCARD_NUMERATOR },
    { "GBufferShaderCommon.fxh:dead texel value", GB_DEAD_TEXEL },
    { "LightingRenderer.GBuffer.cs:ground plane half extent", GB_GROUND_HALF_EXTENT },
    { "LightingRenderer.GBuffer.cs:ground plane lift", GB_GROUND_LIFT },
    { "RasterizeParticleSystem.fx:dither discard threshold numerator", RASTER_DITHER_DISCARD_NUMERATOR },
    { "Gravity.fx:MAX_ATTRACTORS", ILM_MAX_ATTRACTORS },
    { "SpawnerCommon.fxh:MAX_INLINE_POSITION_CONSTANTS", ILM_MAX_INLINE_POSITION_CONSTANTS },
    { "ParticleEngine.cs:RandomnessTextureWidth", ILM_RANDOMNESS_WIDTH }, { "ParticleEngine.cs:RandomnessTextureHeight", ILM_RANDOMNESS_HEIGHT },
};

#endif
