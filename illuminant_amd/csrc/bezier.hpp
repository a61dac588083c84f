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

}  // namespace ilm
