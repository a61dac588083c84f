// bezier.hpp -- tForScaledBezier / evaluateBezier1 / evaluateBezier4 of Illuminant/Shaders/Bezier.fxh:21-177, shared by the particle
// update pass (particles.hip) and the particle rasteriser's vertex stage (raster.hip).
#pragma once

#include "hlsl_math.hpp"

namespace ilm {

// tForScaledBezier, Bezier.fxh:21-67
ILM_DEV float t_for_scaled_bezier(const IlmFloat4& rc, float value, float& t) {
    const float inv_divisor = rc.y;
    const unsigned mode = (unsigned)fabsf(rc.w);
    t = (value - rc.x) * fabsf(inv_divisor);
    if (mode > 511u) {
        t *= 2.0f;
        t = (inv_divisor < 0.0f) ? (2.0f - fmodf(t, 2.0f)) : fmodf(t, 2.0f);
        if (t > 1.0f)
            t = 1.0f - (t - 1.0f);
    } else if (mode > 255u) {
        t = (inv_divisor < 0.0f) ? (1.0f - fmodf(t, 1.0f)) : fmodf(t, 1.0f);
    } else {
        t = (inv_divisor < 0.0f) ? (1.0f - sat(t)) : sat(t);
    }
    const unsigned m = mode % 256u;
    if (m == 1u)
        t = sinf(t * kPi * 0.5f);
    else if (m == 2u)
        t = t * t;
    return rc.z;
}
// evaluateBezier1, Bezier.fxh:69-105
ILM_DEV float bezier1(const IlmClampedBezier1& bz, float value) {
    const float a = bz.ABCD.x, b = bz.ABCD.y, c = bz.ABCD.z, d = bz.ABCD.w;
    if (bz.RangeAndCount.z <= 1.5f) return a;   // constant curve (ClampedBezier1.One): t is not needed
    float t;
    const float count = t_for_scaled_bezier(bz.RangeAndCount, value, t);
    const float ab = lerp(a, b, t);
    if (count <= 2.5f) return ab;
    if (count <= 3.5f) return (t <= 0.0f) ? a : ((t >= 1.0f) ? c : b);
    const float bc = lerp(b, c, t), cd = lerp(c, d, t);
    return lerp(lerp(ab, bc, t), lerp(bc, cd, t), t);
}
// evaluateBezier4, Bezier.fxh:141-177
ILM_DEV float4 bezier4(const IlmClampedBezier4& bz, float value) {
    const float4 a = ld4(bz.A);
    if (bz.RangeAndCount.z <= 1.5f) return a;   // constant curve (ClampedBezier4.One): t is not needed
    float t;
    const float count = t_for_scaled_bezier(bz.RangeAndCount, value, t);
    const float4 b = ld4(bz.B);
    const float4 ab = lerp4(a, b, t);
    if (count <= 2.5f) return ab;
    const float4 c = ld4(bz.C);
    if (count <= 3.5f) return (t <= 0.0f) ? a : ((t >= 1.0f) ? c : b);
    const float4 d = ld4(bz.D);
    const float4 bc = lerp4(b, c, t), cd = lerp4(c, d, t);
    return lerp4(lerp4(ab, bc, t), lerp4(bc, cd, t), t);
}

// ---- the same curves with their uniform decisions taken on the host ------------------------------------------------------------
// Every branch of tForScaledBezier / evaluateBezier depends only on the curve's RangeAndCount, which is uniform: a float compare on
// a uniform value costs a vector compare + a vcc branch per wave (gfx950 has no scalar float unit), so the host evaluates them once
// per launch with the same comparisons (api.hip, bezier_code) and the wave tests bits with scalar integer instructions.
//   bits 0-1: count class  0: count <= 1.5 (constant)  1: <= 2.5 (linear)  2: <= 3.5 (three-point step)  3: cubic (NaN lands here too)
//   bits 2-3: range mode   0: clamp  1: repeat (mode > 255)  2: mirror (mode > 511)         bit 4: inverse divisor < 0
//   bits 5-6: shaping      0: none  1: sine (mode % 256 == 1)  2: square (== 2)
ILM_DEV float t_for_coded_bezier(const IlmFloat4& rc, float value, uint32_t code) {
    const float inv_divisor = rc.y;
    float t = (value - rc.x) * fabsf(inv_divisor);
    const uint32_t range = (code >> 2) & 3u;
    const bool neg = (code & 16u) != 0u;
    if (range == 2u) {
        t *= 2.0f;
        t = neg ? (2.0f - fmodf(t, 2.0f)) : fmodf(t, 2.0f);
        if (t > 1.0f)
            t = 1.0f - (t - 1.0f);
    } else if (range == 1u) {
        t = neg ? (1.0f - fmodf(t, 1.0f)) : fmodf(t, 1.0f);
    } else {
        t = neg ? (1.0f - sat(t)) : sat(t);
    }
    const uint32_t m = (code >> 5) & 3u;
    if (m == 1u)
        t = sinf(t * kPi * 0.5f);
    else if (m == 2u)
        t = t * t;
    return t;
}
ILM_DEV float bezier1_coded(const IlmClampedBezier1& bz, float value, uint32_t code) {
    const uint32_t cls = code & 3u;
    const float a = bz.ABCD.x;
    if (cls == 0u) return a;
    const float t = t_for_coded_bezier(bz.RangeAndCount, value, code);
    const float b = bz.ABCD.y;
    const float ab = lerp(a, b, t);
    if (cls == 1u) return ab;
    const float c = bz.ABCD.z;
    if (cls == 2u) return (t <= 0.0f) ? a : ((t >= 1.0f) ? c : b);
    const float d = bz.ABCD.w;
    const float bc = lerp(b, c, t), cd = lerp(c, d, t);
    return lerp(lerp(ab, bc, t), lerp(bc, cd, t), t);
}
ILM_DEV float4 bezier4_coded(const IlmClampedBezier4& bz, float value, uint32_t code) {
    const uint32_t cls = code & 3u;
    const float4 a = ld4(bz.A);
    if (cls == 0u) return a;
    const float t = t_for_coded_bezier(bz.RangeAndCount, value, code);
    const float4 b = ld4(bz.B);
    const float4 ab = lerp4(a, b, t);
    if (cls == 1u) return ab;
    const float4 c = ld4(bz.C);
    if (cls == 2u) return (t <= 0.0f) ? a : ((t >= 1.0f) ? c : b);
    const float4 d = ld4(bz.D);
    const float4 bc = lerp4(b, c, t), cd = lerp4(c, d, t);
    return lerp4(lerp4(ab, bc, t), lerp4(bc, cd, t), t);
}

}  // namespace ilm
