// particles.hip -- ParticleEngine per-chunk state update for gfx950.
//
// One launch = one ParticleSystem.Update (Illuminant/Particles/ParticleSystem.cs:725-745):
// the spawner pass, every active transform and the update pass are applied to a
// slot in registers, in the reference's pass order, so a chunk streams through
// HBM once (48 B read + 64 B written per live slot) instead of once per pass.
// State is SoA; a wave owns 64 consecutive slots (one lane = one slot), so every
// access is one dword per lane on an SGPR plane base: 256 contiguous bytes per instruction.
//
// HBM-bound integer/float streaming work: no MFMA, no LDS (no cross-slot reuse).
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>

#include <atomic>
#include "internal.hpp"
#include "distance_functions.hpp"
#include "bezier.hpp"

namespace ilm {

// ---------------------------------------------------------------------------------------------
// ParticleCommon.fxh:54-72 accessors
// ---------------------------------------------------------------------------------------------
ILM_DEV float dt_seconds(const IlmParticleSystemUniforms& s) { return s.GlobalSettings.x / kVelocityConstantScale; }

// checkCategoryFilter, ParticleCommon.fxh:187-189
ILM_DEV bool category_ok(float type, const float mm[2]) { return (type >= mm[0]) && (type <= mm[1]); }

// randomCustom, RandomCommon.fxh:27-30 (POINT, WRAP)
ILM_DEV float4 random_custom(const float4* __restrict__ rnd, int rw, int rh, float texel_x, float texel_y, float x, float y,
                             float off_x, float off_y, float rate_x, float rate_y) {
#pragma clang fp contract(off)   // table indices are bit-exact
    const float u = ((x * rate_x) + off_x) * texel_x;
    const float v = ((y * rate_y) + off_y) * texel_y;
    const int tx = wrap_index_fast(floorf(u * (float)rw), rw);
    const int ty = wrap_index_fast(floorf(v * (float)rh), rh);
    return rnd[ty * rw + tx];
}

// ---------------------------------------------------------------------------------------------
// DistanceFunctionCommon.fxh -- area weights for FMA / Noise
// ---------------------------------------------------------------------------------------------
// evaluateByTypeId, DistanceFunctionCommon.fxh:170-187
ILM_DEV float evaluate_area(int type_id, f3 wp, const IlmAreaParams& a) {
    const int t = abs(type_id);
    if (t < 1 || t > 5)
        return 0.0f;
    const f3 center = mk3(a.AreaCenter[0], a.AreaCenter[1], a.AreaCenter[2]);
    const f3 size = mk3(a.AreaSize[0], a.AreaSize[1], a.AreaSize[2]);
    const f3 p = rotate_local(wp - center, a.AreaRotation);
    return evaluate_shape(t, p, size);
}
// computeWeight, FMA.fx:15-20 / Noise.fx:21-26
ILM_DEV float compute_weight(const IlmAreaParams& a, f3 wp) {
#pragma clang fp contract(off)
    const float distance = evaluate_area(a.AreaType, wp, a);
    return (1.0f - sat(distance / a.AreaFalloff)) * a.Strength;
}
// weight and time factor of an area transform.  Without an area evaluateByTypeId returns 0, so
// weight = (1 - saturate(0 / falloff)) * Strength = Strength and t = Strength * dtMs / TimeDivisor are uniform:
// both come from the host (StepDerived), computed with the same operations.
ILM_DEV void area_weight_and_t(const IlmAreaParams& a, const StepDerived::Op& dv, f3 wp, float dt_ms, float time_divisor, float& weight, float& t) {
#pragma clang fp contract(off)
    if (dv.area_none) {
        weight = a.Strength;
        t = dv.t;
    } else {
        weight = compute_weight(a, wp);
        t = weight * dt_ms / time_divisor;
    }
}

// ---------------------------------------------------------------------------------------------
// transforms
// ---------------------------------------------------------------------------------------------
// PS_Gravity, Gravity.fx:12-61.  Every +, -, * rounds as the oracle's (this file is compiled with -ffp-contract=off like the rest of
// the library), so d^2 and the type-0 branch's d^2 - radius are bit-identical to it.  normalize(toCenter), distance / radius and the
// acceleration cap use v_rsq_f32 / v_rcp_f32 (1 ulp -- the accuracy Direct3D itself grants the reference's rcp / rsq / div): relative
// error of each attractor's term <= 4e-7, velocities only.  -DILM_GRAVITY_EXACT builds the IEEE sqrt / division form, bit-identical to
// the oracle, at +14 % step time on cfg2 and cfg4 (measured r02: 25.9 -> 29.3 us, 175 -> 201 us).
ILM_DEV void apply_gravity(float4& pos, float4& vel, const IlmParticleSystemUniforms& sys, const IlmGravityParams& p, const StepDerived::Op& dv) {
    if ((pos.w <= 0.0f) || !category_ok(vel.w, p.CategoryFilter))
        return;
    const float dt_ms = sys.GlobalSettings.x;
    f3 acceleration = mk3(0.0f, 0.0f, 0.0f);
    for (int i = 0; i < p.AttractorCount; i++) {
        const f3 apos = mk3(p.AttractorPositions[i][0], p.AttractorPositions[i][1], p.AttractorPositions[i][2]);
        const float radius = p.AttractorRadiusesAndStrengths[i][0];
        const float strength = p.AttractorRadiusesAndStrengths[i][1];
        const float type = p.AttractorRadiusesAndStrengths[i][2];
        const f3 to_center = apos - xyz(pos);
        float attraction;
        const float d2 = dot3(to_center, to_center);
#ifndef ILM_GRAVITY_EXACT
        const float inv_len = fast_rsq(d2);
        if (type >= 0.5f) {
            const float distance = d2 * inv_len;
            attraction = 1.0f - sat(distance * fast_rcp(radius));
            if (type >= 1.5f)
                attraction *= attraction;
            attraction = attraction * dt_ms * (1.0f / kVelocityConstantScale);
        } else {
            // the one place of this pass where an operand can cancel (d^2 - radius just above its 0.001 floor): the reference's own
            // operations, IEEE division included (the attractor type is uniform: physical-type attractors pay, the others do not)
            const float distance_squared = fmaxf(d2 - radius, 0.001f);
            attraction = 1.0f / distance_squared;
        }
        acceleration = acceleration + (((to_center * inv_len) * attraction) * strength);
#else
        const float distance = sqrtf(d2);
        if (type >= 0.5f) {
            attraction = 1.0f - sat(distance / radius);
            if (type >= 1.5f)
                attraction *= attraction;
            attraction = attraction * dt_ms / kVelocityConstantScale;
        } else {
            const float distance_squared = fmaxf(d2 - radius, 0.001f);
            attraction = 1.0f / distance_squared;
        }
        const f3 n = mk3(to_center.x / distance, to_center.y / distance, to_center.z / distance);
        acceleration = acceleration + ((n * attraction) * strength);
#endif
    }
    const float maximum_acceleration = dv.max_accel;
#ifndef ILM_GRAVITY_EXACT
    const float a2 = dot3(acceleration, acceleration);
    if (a2 > maximum_acceleration * maximum_acceleration)
        acceleration = acceleration * (fast_rsq(a2) * maximum_acceleration);
#else
    const float current_length = len3(acceleration);
    if (current_length > maximum_acceleration)
        acceleration = mk3(acceleration.x / current_length, acceleration.y / current_length, acceleration.z / current_length) * maximum_acceleration;
#endif
    const float mv = sys.GlobalSettings.z;
    vel.x = fminf(mv, vel.x + acceleration.x);
    vel.y = fminf(mv, vel.y + acceleration.y);
    vel.z = fminf(mv, vel.z + acceleration.z);
}

ILM_DEV float fma_term_exact(float v, float m, float a) {
#pragma clang fp contract(off)
    return (v * m) + a;
}
// PS_FMA, FMA.fx:22-51
ILM_DEV void apply_fma(float4& pos, float4& vel, const IlmParticleSystemUniforms& sys, const IlmFMAParams& p, const StepDerived::Op& dv) {
    if ((pos.w <= 0.0f) || !category_ok(vel.w, p.Area.CategoryFilter))
        return;
    float weight, t;
    area_weight_and_t(p.Area, dv, xyz(pos), sys.GlobalSettings.x, p.TimeDivisor, weight, t);
    float4 np = lerp4(pos, add4(mul4(pos, ld4(p.PositionMultiply)), ld4(p.PositionAdd)), t);
    np.w = lerp_exact(pos.w, fma_term_exact(pos.w, p.PositionMultiply.w, p.PositionAdd.w), t);   // life
    const float4 nv = lerp4(vel, add4(mul4(vel, ld4(p.VelocityMultiply)), ld4(p.VelocityAdd)), t);
    pos = np;
    vel = nv;
}

ILM_DEV float4 noise_shape(float4 r, const IlmFloat4& offset, const IlmFloat4& minimum, const IlmFloat4& scale) {
    const float4 d = add4(r, ld4(offset));
    return mk4(sign_times(d.x, fmaxf(fabsf(d.x), minimum.x)) * scale.x, sign_times(d.y, fmaxf(fabsf(d.y), minimum.y)) * scale.y,
               sign_times(d.z, fmaxf(fabsf(d.z), minimum.z)) * scale.z, sign_times(d.w, fmaxf(fabsf(d.w), minimum.w)) * scale.w);
}

// The two noise vectors of a wave whose 64 slots read the same texel in each of the four table lookups
// (StepDerived::NoiseFast): uniform values, held in SGPRs.
struct NoiseDeltas {
    bool valid;
    float4 position, velocity;
};

// positionDelta / velocityDelta of Noise.fx:49-60 from the four table samples
ILM_DEV void noise_deltas(float4 p1, float4 p2, float4 v1, float4 v2, const IlmNoiseParams& p, float4& position_delta, float4& velocity_delta) {
    position_delta = noise_shape(lerp4(p1, p2, p.FrequencyLerp), p.PositionOffset, p.PositionMinimum, p.PositionScale);
    velocity_delta = noise_shape(lerp4(v1, v2, p.FrequencyLerp), p.VelocityOffset, p.VelocityMinimum, p.VelocityScale);
}

// PS_Noise, Noise.fx:28-72 (no life check: dead slots go through the math, :40)
ILM_DEV void apply_noise(float4& pos, float4& vel, float x, float y, const float4* __restrict__ rnd, int rw, int rh,
                         const IlmParticleSystemUniforms& sys, const IlmNoiseParams& p, float inv_rw, float inv_rh, const StepDerived::Op& dv,
                         const NoiseDeltas& uniform) {
    if (!category_ok(vel.w, p.Area.CategoryFilter))
        return;
    float weight, t;
    area_weight_and_t(p.Area, dv, xyz(pos), sys.GlobalSettings.x, p.TimeDivisor, weight, t);

    float4 position_delta, velocity_delta;
    if (uniform.valid) {
        position_delta = uniform.position;
        velocity_delta = uniform.velocity;
    } else {
        const float rate_x = inv_rw, rate_y = inv_rh;  // rate = RandomnessTexel (Noise.fx:49-52)
        const float4 p1 = random_custom(rnd, rw, rh, rate_x, rate_y, x, y, p.RandomnessOffset[0], p.RandomnessOffset[1], rate_x, rate_y);
        const float4 p2 = random_custom(rnd, rw, rh, rate_x, rate_y, x, y, p.NextRandomnessOffset[0], p.NextRandomnessOffset[1], rate_x, rate_y);
        const float4 v1 = random_custom(rnd, rw, rh, rate_x, rate_y, x + 2.0f, y + 1.0f, p.RandomnessOffset[0], p.RandomnessOffset[1], rate_x, rate_y);
        const float4 v2 = random_custom(rnd, rw, rh, rate_x, rate_y, x + 2.0f, y + 1.0f, p.NextRandomnessOffset[0], p.NextRandomnessOffset[1], rate_x, rate_y);
        noise_deltas(p1, p2, v1, v2, p, position_delta, velocity_delta);
    }

    float4 np = lerp4(pos, add4(pos, position_delta), t);
    np.w = lerp_exact(pos.w, pos.w + position_delta.w, t);   // life
    f3 nv;
    if (p.ReplaceOldVelocity != 0.0f)
        nv = mk3(lerp(vel.x, velocity_delta.x, weight), lerp(vel.y, velocity_delta.y, weight), lerp(vel.z, velocity_delta.z, weight));
    else
        nv = mk3(lerp(vel.x, vel.x + velocity_delta.x, t), lerp(vel.y, vel.y + velocity_delta.y, t), lerp(vel.z, vel.z + velocity_delta.z, t));
    nv = nv + (norm3_fast(xyz(vel)) * velocity_delta.w);
    pos = np;
    vel = mk4(nv.x, nv.y, nv.z, vel.w);
}

// The wave-uniform form of the four lookups (StepDerived::NoiseFast): (x0, row) = slot coordinates of the wave's first lane, the wave
// covers x0 .. x0 + 63 of that row (chunk size a multiple of 64).  Scalar integer code only; the deltas arrive in SGPRs.
ILM_DEV NoiseDeltas noise_prepare(const StepDerived::NoiseFast& nf, const IlmStepDesc& desc, int x0, int row) {
    NoiseDeltas out;
    const uint32_t code = nf.wcode[x0 >> 6];
    out.valid = (code & 64u) != 0u;
    int yc0 = 0, yc1 = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        yc0 += (row >= nf.yb[k]) ? 1 : 0;
        yc1 += (row + 1 >= nf.yb[k]) ? 1 : 0;
    }
    const int xc0 = (int)(code & 7u), xc1 = (int)((code >> 3) & 7u);
    if (nf.classes == kNoiseBigClasses) {
        const IlmFloat4* table = reinterpret_cast<const IlmFloat4*>(&desc.Spawns[0]);
        out.position = ld4(table[yc0 * kNoiseBigClasses + xc0]);
        out.velocity = ld4(table[kNoiseBigClasses * kNoiseBigClasses + yc1 * kNoiseBigClasses + xc1]);
    } else {
        out.position = ld4(nf.position[yc0][xc0]);
        out.velocity = ld4(nf.velocity[yc1][xc1]);
    }
    return out;
}

// mul3, ParticleCommon.fxh:183-196
ILM_DEV float4 mul_point(f3 v, const IlmMatrix& M);
ILM_DEV float4 mul3(float4 old_value, const IlmMatrix& mat, float w) {
    const float4 temp = mul_point(xyz(old_value), mat);
    if (w != 0.0f)
        return mk4(temp.x / temp.w, temp.y / temp.w, temp.z / temp.w, old_value.w);
    return mk4(temp.x, temp.y, temp.z, old_value.w);
}

// PS_MatrixMultiply, MatrixMultiply.fx:22-52
ILM_DEV void apply_matrix_multiply(float4& pos, float4& vel, const IlmParticleSystemUniforms& sys, const IlmMatrixMultiplyParams& p) {
#pragma clang fp contract(off)
    if ((pos.w <= 0.0f) || !category_ok(vel.w, p.Area.CategoryFilter))
        return;
    const float time_scale = (p.TimeDivisor >= 0.0f) ? sys.GlobalSettings.x / p.TimeDivisor : 1.0f;
    const float distance = evaluate_area(p.Area.AreaType, xyz(pos), p.Area);
    const float w = ((1.0f - clampf(distance / p.Area.AreaFalloff, 0.0f, 1.0f)) * p.Area.Strength) * time_scale;
    const float4 np = lerp4(pos, mul3(pos, p.PositionMatrix, 1.0f), w);
    const float4 nv = lerp4(vel, mul3(vel, p.VelocityMatrix, 0.0f), w);
    pos = np;
    vel = nv;
}

// smoothRandomCustom, RandomCommon.fxh:36-39: bilinear, WRAP on both axes, on the Rgba64 copy of the randomness table
ILM_DEV float4 smooth_random_custom(const uint2* __restrict__ lp, int rw, int rh, float texel_x, float texel_y, float x, float y,
                                    float off_x, float off_y, float rate_x, float rate_y) {
#pragma clang fp contract(off)
    const float u = ((x * rate_x) + off_x) * texel_x;
    const float v = ((y * rate_y) + off_y) * texel_y;
    const float sx = u * (float)rw - 0.5f, sy = v * (float)rh - 0.5f;
    const float x0f = floorf(sx), y0f = floorf(sy);
    const float fx = sx - x0f, fy = sy - y0f;
    const int x0 = wrap_index(x0f, rw), y0 = wrap_index(y0f, rh);
    const int x1 = (x0 + 1 == rw) ? 0 : x0 + 1, y1 = (y0 + 1 == rh) ? 0 : y0 + 1;
    const uint2 t00 = lp[y0 * rw + x0], t10 = lp[y0 * rw + x1], t01 = lp[y1 * rw + x0], t11 = lp[y1 * rw + x1];
    float r[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const uint32_t w00 = (c < 2) ? t00.x : t00.y, w10 = (c < 2) ? t10.x : t10.y, w01 = (c < 2) ? t01.x : t01.y, w11 = (c < 2) ? t11.x : t11.y;
        const int sh = (c & 1) * 16;
        const float a = unorm16_to_float((float)((w00 >> sh) & 0xFFFFu)), b = unorm16_to_float((float)((w10 >> sh) & 0xFFFFu));
        const float cc = unorm16_to_float((float)((w01 >> sh) & 0xFFFFu)), d = unorm16_to_float((float)((w11 >> sh) & 0xFFFFu));
        r[c] = lerp(lerp(a, b, fx), lerp(cc, d, fx), fy);
    }
    return mk4(r[0], r[1], r[2], r[3]);
}

// PS_SpatialNoise, Noise.fx:74-116 (no life check, no Minimum shaping)
ILM_DEV void apply_spatial_noise(float4& pos, float4& vel, const uint2* __restrict__ lp, int rw, int rh,
                                 const IlmParticleSystemUniforms& sys, const IlmSpatialNoiseParams& sp, const StepDerived& sd) {
#pragma clang fp contract(off)
    const IlmNoiseParams& p = sp.Noise;
    if (!category_ok(vel.w, p.Area.CategoryFilter))
        return;
    const float weight = compute_weight(p.Area, xyz(pos));
    const float t = weight * sys.GlobalSettings.x / p.TimeDivisor;
    const float rx = pos.x, ry = pos.y, sx = sp.SpaceScale[0], sy = sp.SpaceScale[1];
    const float4 p1 = smooth_random_custom(lp, rw, rh, sd.inv_rw, sd.inv_rh, rx, ry, p.RandomnessOffset[0], p.RandomnessOffset[1], sx, sy);
    const float4 p2 = smooth_random_custom(lp, rw, rh, sd.inv_rw, sd.inv_rh, rx, ry, p.NextRandomnessOffset[0], p.NextRandomnessOffset[1], sx, sy);
    const float4 v1 = smooth_random_custom(lp, rw, rh, sd.inv_rw, sd.inv_rh, rx + 2.0f, ry + 1.0f, p.RandomnessOffset[0], p.RandomnessOffset[1], sx, sy);
    const float4 v2 = smooth_random_custom(lp, rw, rh, sd.inv_rw, sd.inv_rh, rx + 2.0f, ry + 1.0f, p.NextRandomnessOffset[0], p.NextRandomnessOffset[1], sx, sy);
    const float4 position_delta = mul4(add4(lerp4(p1, p2, p.FrequencyLerp), ld4(p.PositionOffset)), ld4(p.PositionScale));
    const float4 velocity_delta = mul4(add4(lerp4(v1, v2, p.FrequencyLerp), ld4(p.VelocityOffset)), ld4(p.VelocityScale));
    const float4 np = lerp4(pos, add4(pos, position_delta), t);
    f3 nv;
    if (p.ReplaceOldVelocity != 0.0f)
        nv = mk3(lerp(vel.x, velocity_delta.x, weight), lerp(vel.y, velocity_delta.y, weight), lerp(vel.z, velocity_delta.z, weight));
    else
        nv = mk3(lerp(vel.x, vel.x + velocity_delta.x, t), lerp(vel.y, vel.y + velocity_delta.y, t), lerp(vel.z, vel.z + velocity_delta.z, t));
    nv = nv + (norm3(xyz(vel)) * velocity_delta.w);
    pos = np;
    vel = mk4(nv.x, nv.y, nv.z, vel.w);
}

// ---------------------------------------------------------------------------------------------
// spawner -- SpawnerCommon.fxh + PS_Spawn (SpawnParticles.fx:10-30)
// ---------------------------------------------------------------------------------------------
ILM_DEV f3 random_normal3(float rx, float ry) {
    const float phi = rx * kPi * 2.0f;
    const float costheta = (ry - 0.5f) * 2.0f;
    const float theta = acosf(costheta);
    float st, ct, sp, cp;
    st = sinf(theta); ct = cosf(theta); sp = sinf(phi); cp = cosf(phi);
    return mk3(st * cp, st * sp, ct);
}

// evaluateFormula, SpawnerCommon.fxh:59-104
ILM_DEV float4 evaluate_formula(float4 origin, float4 constant, float4 scale, float4 offset, float4 randomness,
                                float type, const float axis_mask[3]) {
    const float4 type0 = add4(constant, mul4(add4(randomness, offset), scale));
    const unsigned itype = (unsigned)fabsf(floorf(type));
    if (itype == 1u || itype == 3u) {
        f3 rn = random_normal3(randomness.x, randomness.y);
        rn = norm3(mk3(rn.x * axis_mask[0], rn.y * axis_mask[1], rn.z * axis_mask[2]));
        f3 circular = mk3(rn.x * randomness.z * scale.x, rn.y * randomness.z * scale.y, rn.z * randomness.z * scale.z);
        f3 result;
        if (itype == 3u) {
            const float sqrt2 = 1.41421356237f;
            const f3 edge = mk3(fabsf(offset.x), fabsf(offset.y), fabsf(offset.z));
            result = mk3(clampf(offset.x * rn.x * sqrt2, -edge.x, edge.x), clampf(offset.y * rn.y * sqrt2, -edge.y, edge.y),
                         clampf(offset.z * rn.z * sqrt2, -edge.z, edge.z));
            result = result + (xyz(constant) + circular);
        } else {
            circular = circular + (rn * xyz(offset));
            result = xyz(constant) + circular;
        }
        return mk4(result.x, result.y, result.z, type0.w);
    }
    if (itype == 2u) {
        const f3 distance = xyz(constant) - xyz(origin);
        const float ldistance = len3(distance);
        if (ldistance < 0.1f)
            return mk4(0.0f, 0.0f, 0.0f, constant.w);
        const f3 direction = mk3(distance.x / ldistance, distance.y / ldistance, distance.z / ldistance);
        const f3 random_speed = mk3(randomness.x * scale.x * direction.x, randomness.x * scale.y * direction.y, randomness.x * scale.z * direction.z);
        const f3 s = random_speed + (xyz(offset) * direction);
        return mk4(s.x, s.y, s.z, type0.w);
    }
    return type0;
}

ILM_DEV float4 mul_point(f3 v, const IlmMatrix& M) {
    const float* m = M.m;
    return mk4(v.x * m[0] + v.y * m[4] + v.z * m[8] + m[12], v.x * m[1] + v.y * m[5] + v.z * m[9] + m[13],
               v.x * m[2] + v.y * m[6] + v.z * m[10] + m[14], v.x * m[3] + v.y * m[7] + v.z * m[11] + m[15]);
}

// index % m for an integer-valued index in [0, 2^24) and a literal integer m: HLSL's float % is exact here, and so
// is the integer remainder (which the compiler turns into a multiply-high); OCML's fmodf is a ~100-instruction loop.
template <unsigned M>
ILM_DEV float mod_const(float index) { return (float)((unsigned)index % M); }

// evaluateRandomForIndex, SpawnerCommon.fxh:106-117
ILM_DEV void evaluate_random_for_index(const float4* __restrict__ rnd, int rw, int rh, float tx_, float ty_, float index, const IlmSpawnParams& p,
                                       float4& random1, float4& random2, float4& random3) {
    const float ox = p.RandomnessOffset[0], oy = p.RandomnessOffset[1];
    random1 = random_custom(rnd, rw, rh, tx_, ty_, mod_const<ref::kRandom1XModulus>(index), 0.0f + mod_const<ref::kRandom1YModulus>(index), ox, oy, 1.0f, 1.0f);
    random2 = random_custom(rnd, rw, rh, tx_, ty_, mod_const<ref::kRandom2XModulus>(index), 1.0f + mod_const<ref::kRandom2YModulus>(index), ox, oy, 1.0f, 1.0f);
    random3 = random_custom(rnd, rw, rh, tx_, ty_, mod_const<ref::kRandom3XModulus>(index), 2.0f + mod_const<ref::kRandom3YModulus>(index), ox, oy, 1.0f, 1.0f);
    // "The x and y element of random samples determines the normal" (:114-116): inside evaluateRandomForIndex, so the feedback and
    // pattern spawners (SpawnParticles.fx:83, PatternSpawner.fx:63) align too
    if (p.AlignVelocityAndPosition != 0.0f) {
        random2.x = random1.x;
        random2.y = random1.y;
    }
}

// Spawn_Stage1, SpawnerCommon.fxh:119-160: the slot test, the three random vectors and the position-constant indices
ILM_DEV bool spawn_stage1(float x, float y, const float4* __restrict__ rnd, int rw, int rh, float tx_, float ty_, const IlmSpawnParams& p,
                          float4& random1, float4& random2, float4& random3, int& index1, int& index2, float& position_index_t) {
    const float index = x + (y * p.ChunkSizeAndIndices[0]);
    if ((index < p.ChunkSizeAndIndices[1]) || (index > p.ChunkSizeAndIndices[2]))
        return false;
    evaluate_random_for_index(rnd, rw, rh, tx_, ty_, index, p, random1, random2, random3);
    const float relative_index = index - p.ChunkSizeAndIndices[1];
    if (p.PolygonRate > 0.05f) {
        const float position_index_f = (relative_index / p.PolygonRate) + p.ChunkSizeAndIndices[3];
        const float divisor = p.PositionConstantCount;
        float position_index_i;
        position_index_t = modff(position_index_f, &position_index_i);
        index1 = (int)fmodf(position_index_i, divisor);
        if (p.PolygonLoop != 0.0f)
            index2 = (int)fmodf(position_index_i + 1.0f, divisor);
        else
            index2 = (int)fminf((float)(index1 + 1), divisor - 1.0f);
    } else {
        // integer-valued operands (slot index + TotalSpawned % count, the position count): exact either way
        index1 = index2 = wrap_index_fast(relative_index + p.ChunkSizeAndIndices[3], (int)p.PositionConstantCount);
        position_index_t = 0.0f;
    }
    return true;
}

// Spawn_Stage2, SpawnerCommon.fxh:162-190.  Returns false on the alpha discard.
ILM_DEV bool spawn_stage2(float4 position1, float4 position2, float position_index_t, float4 random1, float4 random2, float4 random3,
                          const IlmSpawnParams& p, float4& pos, float4& vel, float4& attr) {
    const float4 position_constant = lerp4(position1, position2, position_index_t);
    const float4 towards_next = sub4(position2, position1);

    const float4 zero = mk4(0.0f, 0.0f, 0.0f, 0.0f);
    const float4 temp_position = evaluate_formula(zero, position_constant, ld4(p.Configuration[0]), ld4(p.Configuration[1]),
                                                  random1, p.FormulaTypes[0], p.AxisMask);
    float4 new_position = mul_point(xyz(temp_position), p.PositionMatrix);
    new_position.w = temp_position.w;

    float4 temp_velocity = evaluate_formula(temp_position, ld4(p.Configuration[2]), ld4(p.Configuration[3]), ld4(p.Configuration[4]),
                                            random2, p.FormulaTypes[1], p.AxisMask);
    const float4 new_attributes = evaluate_formula(zero, ld4(p.Configuration[5]), ld4(p.Configuration[6]), ld4(p.Configuration[7]),
                                                   random3, p.FormulaTypes[2], p.AxisMask);

    const float towards_distance = sqrtf(towards_next.x * towards_next.x + towards_next.y * towards_next.y +
                                         towards_next.z * towards_next.z + towards_next.w * towards_next.w);
    if (towards_distance > 0.0001f) {
        const float c = p.Configuration[8].x, s = p.Configuration[8].y, o = p.Configuration[8].z, r = random3.w;
        const float towards_speed = evaluate_formula(zero, mk4(c, c, c, c), mk4(s, s, s, s), mk4(o, o, o, o), mk4(r, r, r, r),
                                                     p.FormulaTypes[3], p.AxisMask).x;
        temp_velocity = add4(temp_velocity, mk4((towards_next.x / towards_distance) * towards_speed, (towards_next.y / towards_distance) * towards_speed,
                                                (towards_next.z / towards_distance) * towards_speed, (towards_next.w / towards_distance) * towards_speed));
    }
    float4 new_velocity = mul_point(xyz(temp_velocity), p.VelocityMatrix);
    new_velocity.w = temp_velocity.w;

    if (new_attributes.w < p.AttributeDiscardThreshold)
        return false;
    pos = new_position;
    vel = new_velocity;
    attr = new_attributes;
    return true;
}

// PS_Spawn, SpawnParticles.fx:10-30.  Returns true when the slot was (re)written by the spawner.
ILM_DEV bool spawn_slot(float4& pos, float4& vel, float4& attr, float x, float y,
                        const float4* __restrict__ rnd, int rw, int rh, float tx_, float ty_, const IlmSpawnParams& p) {
    float4 random1, random2, random3;
    int index1, index2;
    float position_index_t;
    if (!spawn_stage1(x, y, rnd, rw, rh, tx_, ty_, p, random1, random2, random3, index1, index2, position_index_t))
        return false;
    index1 = min(max(index1, 0), ILM_MAX_INLINE_POSITION_CONSTANTS - 1);
    index2 = min(max(index2, 0), ILM_MAX_INLINE_POSITION_CONSTANTS - 1);
    return spawn_stage2(ld4(p.InlinePositionConstants[index1]), ld4(p.InlinePositionConstants[index2]), position_index_t,
                        random1, random2, random3, p, pos, vel, attr);
}

// tex2Dlod(PositionConstantSampler, index * PositionConstantTexel.x): POINT / CLAMP on the Spawner's PositionBuffer, whose width is
// the count rounded up to a multiple of 128 and whose padding stays zero (ParticleSpawner.cs:301-314, SpawnParticles.fx:47-48)
ILM_DEV float4 position_constant_fetch(const float4* __restrict__ positions, int count, int index) {
#pragma clang fp contract(off)
    const int width = (count + 127) / 128 * 128;
    const float texel = 1.0f / (float)width;
    const float u = (float)index * texel;
    const int tx = min(max((int)floorf(u * (float)width), 0), width - 1);
    return (tx < count) ? positions[tx] : mk4(0.0f, 0.0f, 0.0f, 0.0f);
}

// PS_SpawnFromPositionTexture, SpawnParticles.fx:32-52
ILM_DEV bool spawn_slot_position_buffer(float4& pos, float4& vel, float4& attr, float x, float y, const float4* __restrict__ rnd, int rw, int rh,
                                        float tx_, float ty_, const IlmSpawnParams& p, const float4* __restrict__ positions, int count) {
    float4 random1, random2, random3;
    int index1, index2;
    float position_index_t;
    if (!spawn_stage1(x, y, rnd, rw, rh, tx_, ty_, p, random1, random2, random3, index1, index2, position_index_t))
        return false;
    return spawn_stage2(position_constant_fetch(positions, count, index1), position_constant_fetch(positions, count, index2), position_index_t,
                        random1, random2, random3, p, pos, vel, attr);
}

// PS_SpawnFeedback, SpawnParticles.fx:54-118.  `src` = plane 0 of the source chunk (same stride S and chunk size as the target:
// SourceChunkSizeAndTexel = (size, 1/size, 1/size), ParticleTransform.cs:129-141).
ILM_DEV bool spawn_slot_feedback(float4& pos, float4& vel, float4& attr, float x, float y, const float4* __restrict__ rnd, int rw, int rh,
                                 float tx_, float ty_, const IlmSpawnParams& p, const IlmFeedbackParams& fb,
                                 const float* __restrict__ src, int64_t S, int chunk_size) {
#pragma clang fp contract(off)
    const float index = x + (y * p.ChunkSizeAndIndices[0]);
    if ((index < p.ChunkSizeAndIndices[1]) || (index > p.ChunkSizeAndIndices[2]))
        return false;
    const float size = (float)chunk_size, texel = 1.0f / (float)chunk_size;
    const float source_index = ((index - p.ChunkSizeAndIndices[1]) / fb.InstanceMultiplier) + fb.FeedbackSourceIndex;
    float source_y;
    const float source_x = modff(source_index / size, &source_y) * size;
    // readStateUv: POINT / CLAMP at uv = sourceXy * texel
    const int tx = min(max((int)floorf((source_x * texel) * size), 0), chunk_size - 1);
    const int ty = min(max((int)floorf((source_y * texel) * size), 0), chunk_size - 1);
    const int si = ty * chunk_size + tx;
    const float4 source_position = mk4(src[si], src[S + si], src[2 * S + si], src[3 * S + si]);
    if ((source_position.w <= fb.SourceLifeRange[0]) || (source_position.w >= fb.SourceLifeRange[1]))
        return false;
    const float4 source_velocity = mk4(src[4 * S + si], src[5 * S + si], src[6 * S + si], src[7 * S + si]);
    const float4 source_attributes = mk4(src[8 * S + si], src[9 * S + si], src[10 * S + si], src[11 * S + si]);

    float4 random1, random2, random3;
    evaluate_random_for_index(rnd, rw, rh, tx_, ty_, index, p, random1, random2, random3);

    const float4 zero = mk4(0.0f, 0.0f, 0.0f, 0.0f);
    float4 position_constant = ld4(p.InlinePositionConstants[0]);
    if (fb.AlignPositionConstant != 0.0f) {
        position_constant.x += source_position.x; position_constant.y += source_position.y; position_constant.z += source_position.z;
    }
    const float4 temp_position = evaluate_formula(zero, position_constant, ld4(p.Configuration[0]), ld4(p.Configuration[1]),
                                                  random1, p.FormulaTypes[0], p.AxisMask);
    float4 attribute_constant = ld4(p.Configuration[5]);
    if (fb.MultiplyAttributeConstant != 0.0f)
        attribute_constant = mul4(attribute_constant, source_attributes);
    float4 new_position = mul_point(xyz(temp_position), p.PositionMatrix);
    new_position.w = temp_position.w;
    if (fb.MultiplyLife != 0.0f)
        new_position.w *= source_position.w;
    float4 temp_velocity = evaluate_formula(temp_position, ld4(p.Configuration[2]), ld4(p.Configuration[3]), ld4(p.Configuration[4]),
                                            random2, p.FormulaTypes[1], p.AxisMask);
    temp_velocity = add4(temp_velocity, mk4(source_velocity.x * fb.SourceVelocityFactor, source_velocity.y * fb.SourceVelocityFactor,
                                            source_velocity.z * fb.SourceVelocityFactor, source_velocity.w * fb.SourceVelocityFactor));
    float4 new_velocity = mul_point(xyz(temp_velocity), p.VelocityMatrix);
    new_velocity.w = temp_velocity.w;
    const float4 new_attributes = evaluate_formula(temp_position, attribute_constant, ld4(p.Configuration[6]), ld4(p.Configuration[7]),
                                                   random3, p.FormulaTypes[2], p.AxisMask);
    if (new_attributes.w < p.AttributeDiscardThreshold)
        return false;
    pos = new_position;
    vel = new_velocity;
    attr = new_attributes;
    return true;
}

// tex2Dlod(PatternSampler, float4(uv, 0, lod)): CLAMP, LINEAR min / mag, POINT mip (PatternSpawner.fx:11-19).  `tex` holds the mip
// levels back to back (include/illuminant_hip.h, ilm_system_set_spawn_pattern); texel centres sit at integer + 0.5.
ILM_DEV float4 pattern_fetch(const float4* __restrict__ tex, int w, int h, int levels, float u, float v, float lod) {
#pragma clang fp contract(off)
    const int level = min(max((int)floorf(lod + 0.5f), 0), levels - 1);
    int lw = w, lh = h;
    for (int l = 0; l < level; l++) {
        tex += lw * lh;
        lw = max(1, lw >> 1); lh = max(1, lh >> 1);
    }
    const float sx = u * (float)lw - 0.5f, sy = v * (float)lh - 0.5f;
    const float x0f = floorf(sx), y0f = floorf(sy);
    const float fx = sx - x0f, fy = sy - y0f;
    // clamp in float first: u, v are unbounded below (the caller only rejects coordinates above 1)
    const int x0 = (int)fminf(fmaxf(x0f, 0.0f), (float)(lw - 1)), x1 = (int)fminf(fmaxf(x0f + 1.0f, 0.0f), (float)(lw - 1));
    const int y0 = (int)fminf(fmaxf(y0f, 0.0f), (float)(lh - 1)), y1 = (int)fminf(fmaxf(y0f + 1.0f, 0.0f), (float)(lh - 1));
    const float4 t00 = tex[y0 * lw + x0], t10 = tex[y0 * lw + x1], t01 = tex[y1 * lw + x0], t11 = tex[y1 * lw + x1];
    return lerp4(lerp4(t00, t10, fx), lerp4(t01, t11, fx), fy);
}

// PS_SpawnPattern, PatternSpawner.fx:21-97
ILM_DEV bool spawn_slot_pattern(float4& pos, float4& vel, float4& attr, float x, float y, const float4* __restrict__ rnd, int rw, int rh,
                                float tx_, float ty_, const IlmSpawnParams& p, const IlmPatternParams& pt,
                                const float4* __restrict__ tex, int tw, int th, int levels) {
#pragma clang fp contract(off)
    const float index = x + (y * p.ChunkSizeAndIndices[0]);
    if ((index < p.ChunkSizeAndIndices[1]) || (index > p.ChunkSizeAndIndices[2]))
        return false;
    const float relative_index = floorf(index - p.ChunkSizeAndIndices[1]);
    const float particles_per_row = pt.StepWidthAndSizeScale[1];
    // integer-valued operands below 2^24: the quotient's floor and the remainder are exact (HLSL's float %)
    const float row = floorf(relative_index / particles_per_row);
    float ix = floorf(relative_index - (particles_per_row * row)), iy = row;
    iy += pt.YOffsetsAndCoordScale[0];
    const float u = (ix * pt.StepWidthAndSizeScale[2]) + pt.TexelOffsetAndMipBias[0];
    const float v = ((iy * pt.StepWidthAndSizeScale[3]) + pt.TexelOffsetAndMipBias[1]) + pt.YOffsetsAndCoordScale[1];
    const float position_x = ix * pt.YOffsetsAndCoordScale[2] + pt.CenteringOffset[0];
    const float position_y = iy * pt.YOffsetsAndCoordScale[3] + pt.CenteringOffset[1];
    if ((u > 1.0f) || (v > 1.0f))
        return false;
    const float4 pattern_color = pattern_fetch(tex, tw, th, levels, u, v, pt.TexelOffsetAndMipBias[3]);

    float4 random1, random2, random3;
    evaluate_random_for_index(rnd, rw, rh, tx_, ty_, index, p, random1, random2, random3);

    const float4 zero = mk4(0.0f, 0.0f, 0.0f, 0.0f);
    float4 temp_position = evaluate_formula(zero, ld4(p.InlinePositionConstants[0]), ld4(p.Configuration[0]), ld4(p.Configuration[1]),
                                            random1, p.FormulaTypes[0], p.AxisMask);
    temp_position.x += position_x;
    temp_position.y += position_y;
    float4 attribute_constant = pattern_color;
    if (pt.MultiplyAttributeConstant != 0.0f)
        attribute_constant = mul4(attribute_constant, ld4(p.Configuration[5]));
    else
        attribute_constant = add4(attribute_constant, ld4(p.Configuration[5]));
    float4 new_position = mul_point(xyz(temp_position), p.PositionMatrix);
    new_position.w = temp_position.w;
    const float4 temp_velocity = evaluate_formula(temp_position, ld4(p.Configuration[2]), ld4(p.Configuration[3]), ld4(p.Configuration[4]),
                                                  random2, p.FormulaTypes[1], p.AxisMask);
    float4 new_velocity = mul_point(xyz(temp_velocity), p.VelocityMatrix);
    new_velocity.w = temp_velocity.w;
    const float4 new_attributes = evaluate_formula(temp_position, attribute_constant, ld4(p.Configuration[6]), ld4(p.Configuration[7]),
                                                   random3, p.FormulaTypes[2], p.AxisMask);
    if (new_attributes.w < p.AttributeDiscardThreshold)
        return false;
    pos = new_position;
    vel = new_velocity;
    attr = new_attributes;
    return true;
}

// ---------------------------------------------------------------------------------------------
// update -- Bezier.fxh, UpdateCommon.fxh, UpdateParticleSystem*.fx
// ---------------------------------------------------------------------------------------------
// applyFrictionAndMaximum, UpdateCommon.fxh:20-35
ILM_DEV f3 friction_and_maximum(f3 velocity, const IlmParticleSystemUniforms& sys, float dts) {
    // v_sqrt_f32 / v_rcp_f32 (1 ulp; only velocities depend on them) behave like the reference's length() / normalize() at the
    // edges: a NaN or infinite speed fails `l <= 0.001` and poisons the same components as dividing by it would
    float l = len3_fast(velocity);
    const float inv_l = fast_rcp(l);
    if (l <= 0.001f)
        return mk3(0.0f, 0.0f, 0.0f);
    const float mv = sys.GlobalSettings.z;
    if (l > mv)
        l = mv;
    const float friction = l * sys.GlobalSettings.y;
    l -= (friction * dts);
    l = clampf(l, 0.0f, mv);
    return velocity * (inv_l * l);
}

// computeRenderData, UpdateCommon.fxh:96-117.  `codes` / `bits`: the uniform decisions of the four curves, the life ramp and the
// velocity rotation, taken on the host (StepDerived::bezier_codes / update_bits).
ILM_DEV void render_data(float vx, float vy, float4 position, float4 velocity, float4 attributes,
                         const IlmParticleSystemUniforms& sys, const IlmUpdateParams& p, uint32_t codes, uint32_t bits,
                         const float4* __restrict__ ramp, int ramp_w, int ramp_h,
                         float4& render_color, float4& rdata) {
    if (position.w <= 0.0f) {
        render_color = rdata = mk4(0.0f, 0.0f, 0.0f, 0.0f);
        return;
    }
    const float index = vx + (vy * ref::kRenderDataIndexRowPitch);  // reference quirk: 256 regardless of ChunkSize (:107)
    const float velocity_length = fmaxf(len3_fast(xyz(velocity)), 0.0001f);

    float4 color = mul4(bezier4_coded(p.ColorFromLife, position.w, codes & 0xFFu), bezier4_coded(p.ColorFromVelocity, velocity_length, (codes >> 8) & 0xFFu));
    if (bits & 1u) {    // p.LifeRampSettings.x != 0
        float u = (position.w - p.LifeRampSettings.y) / p.LifeRampSettings.z;
        if (bits & 4u)  // p.LifeRampSettings.x < 0
            u = 1.0f - sat(u);
        const float v = index / p.LifeRampSettings.w;
        float4 texel = mk4(1.0f, 1.0f, 1.0f, 1.0f);
        if (ramp != nullptr && ramp_w > 0 && ramp_h > 0) {
            const int tx = min(max((int)floorf(u * (float)ramp_w), 0), ramp_w - 1);   // U CLAMP
            const int ty = wrap_index(floorf(v * (float)ramp_h), ramp_h);             // V WRAP
            texel = ramp[ty * ramp_w + tx];
        }
        color = lerp4(color, mul4(texel, color), sat(fabsf(p.LifeRampSettings.x)));
    }

    float4 rc = mul4(attributes, color);
    rc.w = sat(rc.w);
    rc.x *= rc.w; rc.y *= rc.w; rc.z *= rc.w;
    render_color = rc;

    // getRotationForVelocity, UpdateCommon.fxh:81-94
    float rotation = 0.0f;
    // the angle is multiplied by getVelocityRotation(): skipping atan2 when that is 0 is exact for every finite or infinite velocity
    // (the angle is finite); a NaN velocity (normalize(0) upstream) makes the angle NaN and NaN * 0 stays NaN
    if (bits & 2u) {    // sys.AnimationRateAndRotationAndZToY.z == 0
        rotation = ((velocity.x != velocity.x) || (velocity.y != velocity.y)) ? __builtin_nanf("") : 0.0f;
    } else if (!((fabsf(velocity.x) < 0.01f) && (fabsf(velocity.y) < 0.01f))) {
        rotation = atan2f(velocity.y, velocity.x);
        if (rotation < 0.0f)
            rotation += 2.0f * kPi;
    }
    rdata.x = bezier1_coded(p.SizeFromLife, position.w, (codes >> 16) & 0xFFu) * bezier1_coded(p.SizeFromVelocity, velocity_length, codes >> 24);
    rdata.y = (rotation * sys.AnimationRateAndRotationAndZToY.z) +
              ((position.w * p.RotationFromLifeAndIndex[0]) + (index * p.RotationFromLifeAndIndex[1]));
    rdata.z = velocity_length;
    rdata.w = velocity.w;
}

// PS_Update, UpdateParticleSystem.fx:9-38 (live slot)
ILM_DEV void update_positions(float4& pos, float4& vel, const IlmParticleSystemUniforms& sys, float dts) {
#pragma clang fp contract(off)   // life arithmetic is bit-exact
    const f3 velocity = friction_and_maximum(xyz(vel), sys, dts);
    const float new_life = pos.w - (sys.GlobalSettings.w * dts);
    if (new_life <= 0.0f) {
        pos = vel = mk4(0.0f, 0.0f, 0.0f, 0.0f);
    } else {
        pos = mk4(pos.x + velocity.x * dts, pos.y + velocity.y * dts, pos.z + velocity.z * dts, new_life);
        vel = mk4(velocity.x, velocity.y, velocity.z, vel.w);
    }
}

// Diagnostic (ilm_debug_step_sdf_samples): how many sampleDistanceFieldEx calls the collision update makes -- the unit of its roofline
// (bench.py collision_step_1m).  The flag is read by a scalar load; the counter costs nothing while it is off.
__device__ unsigned long long g_step_sdf_samples;
__device__ int g_step_count_sdf_samples;

// estimateNormal4, VisualizeCommon.fxh:44-63
template <int FMT>
ILM_DEV f3 estimate_normal4(f3 position, const IlmDistanceFieldUniforms& df, const SdfView& sdf) {
#pragma clang fp contract(off)
    const f3 texel = mk3(df.ConeAndMisc.w, df.StepAndMisc2.w, df.Extent.z / fmaxf(df.TextureSliceCount.w, 1.0f));
    f3 result = mk3(0.0f, 0.0f, 0.0f);
    const float W[4][3] = { { 1, -1, -1 }, { -1, -1, 1 }, { -1, 1, -1 }, { 1, 1, 1 } };
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const f3 w = mk3(W[i][0], W[i][1], W[i][2]);
        const float s = sample_distance_field<(FMT & 1), true, (FMT & 2) != 0, (FMT & 4) != 0>(position + (w * texel), df, sdf);
        result = result + (w * s);
    }
    return norm3(result);
}

// PS_Update, UpdateParticleSystemWithDistanceField.fx:29-147 (live slot)
template <int FMT>
ILM_DEV void update_with_distance_field(float4& pos, float4& vel, float x, float y, const IlmParticleSystemUniforms& sys, float dts,
                                        const IlmDistanceFieldUniforms& df, const SdfView& sdf, int& samples) {
#pragma clang fp contract(off)   // discontinuous collision state machine + life arithmetic: keep IEEE-exact
    const float4 zero = mk4(0.0f, 0.0f, 0.0f, 0.0f);
    float new_life = pos.w - (sys.GlobalSettings.w * dts);
    if (new_life <= 0.0f) {
        pos = vel = zero;
        return;
    }
    const float collision_distance = sys.CollisionSettings.z;
    const float max_velocity = sys.GlobalSettings.z;
    const f3 old_xyz = xyz(pos);
    const f3 unit_vector = norm3(xyz(vel));
    const f3 velocity = friction_and_maximum(xyz(vel), sys, dts);
    const f3 scaled_velocity = velocity * dts;

    bool collided = false, escaping = false;
    f3 collision_position = mk3(0.0f, 0.0f, 0.0f), new_position = old_xyz;
    float4 new_velocity = zero;

    const float initial_distance = sample_distance_field<(FMT & 1), true, (FMT & 2) != 0>(old_xyz, df, sdf);
    samples++;
    const bool was_colliding = initial_distance < collision_distance;
    float travel_distance = fmaxf(0.0f, fminf(initial_distance, len3(scaled_velocity)));
    int step_count = ref::kMaxStepCount;
    if (was_colliding)
        step_count = 1;
    else if (travel_distance <= 0.001f)
        step_count = 0;

    for (int i = 0; i < step_count; i++) {
        const f3 test_position = old_xyz + (unit_vector * travel_distance);
        const float step_distance = sample_distance_field<(FMT & 1), true, (FMT & 2) != 0>(test_position, df, sdf);
        samples++;
        if (step_distance < collision_distance) {
            collided = true;
            collision_position = test_position;
        }
        escaping = step_distance > initial_distance;
        if (collided && !escaping) {
            collision_position = test_position;
            const float offset = clampf(step_distance + collision_distance, 0.05f, 16.0f);
            travel_distance = fmaxf(0.0f, travel_distance - offset);
        } else
            step_count = 0;
        if (travel_distance <= 0.001f)
            step_count = 0;
    }

    if (collided) {
        const bool bounce = vel.w <= 0.0f;
        const bool redirect = was_colliding && !escaping;
        f3 normal = mk3(0.0f, 0.0f, 0.0f);
        if (bounce || redirect) {
            normal = estimate_normal4<FMT>(collision_position, df, sdf);
            samples += 4;
        }
        const float escape_speed = fminf(max_velocity, sys.CollisionSettings.x);
        if (redirect) {
            normal = normal * mk3(1.0f, 1.0f, 0.0f);  // ESCAPE_MASK
            if (len3(normal) < ref::kNoNormalThreshold) {
                const float a = (x / 67.0f) + (y / 13.0f);
                normal = mk3(sinf(a), cosf(a), 0.0f);
            }
            const f3 nv = (norm3(normal) * escape_speed) * ref::kInitialEscapeSpeed;
            new_velocity = mk4(nv.x, nv.y, nv.z, ref::kBounceDelay);
            new_position = old_xyz + (nv * dts);
        } else if (bounce) {
            const float d2 = 2.0f * dot3(normal, unit_vector);
            f3 bounce_vector = ((normal - unit_vector) * d2) * -1.0f;
            if (len3(bounce_vector) < ref::kNoNormalThreshold)
                bounce_vector = unit_vector * -1.0f;
            else
                bounce_vector = norm3(bounce_vector);
            new_position = collision_position;
            const f3 nv = bounce_vector * fminf(max_velocity, len3(velocity) * sys.CollisionSettings.y);
            new_velocity = mk4(nv.x, nv.y, nv.z, ref::kBounceDelay);
            new_life -= sys.CollisionSettings.w;
        } else {
            const float new_speed = fmaxf(len3(xyz(vel)) * ref::kEscapeSpeedAcceleration, escape_speed);
            const f3 nv = unit_vector * new_speed;
            new_velocity = mk4(nv.x, nv.y, nv.z, 0.0f);
            new_position = old_xyz + (unit_vector * travel_distance);
        }
    } else {
        new_velocity = mk4(velocity.x, velocity.y, velocity.z, fmaxf(vel.w - 1.0f, 0.0f));
        new_position = old_xyz + (unit_vector * travel_distance);
    }
    if (new_life <= 0.0f) {
        new_position = mk3(0.0f, 0.0f, 0.0f);
        new_velocity = zero;
    }
    pos = mk4(new_position.x, new_position.y, new_position.z, new_life);
    vel = new_velocity;
}

// ---------------------------------------------------------------------------------------------
// the fused step kernel
// ---------------------------------------------------------------------------------------------
// Persistent, software-pipelined: every wave walks units of 64 consecutive slots (u, u + W, u + 2W, ...);
// the 12 input loads of the next unit are issued before the current unit is computed, so HBM loads stay
// in flight during the arithmetic (the chunk base pointers are cast to the global address space so the
// compiler can use counted vmcnt waits instead of the flat-address vmcnt(0)).  One lane = one slot:
// a wave reads/writes 256 contiguous bytes of each component plane per instruction.
typedef float __attribute__((address_space(1))) gfloat;

struct SlotIn {
    float px, py, pz, life, vx, vy, vz, ct, ar, ag, ab, aa;
};

// The 20 planes of one unit through ONE buffer resource (the chunk's allocation) and one lane offset: plane c of the unit is
// `buffer_load/store_dword v, v_lane_offset, s[rsrc], s_plane_offset offen` with s_plane_offset = (first slot + c * stride) * 4 in an
// SGPR -- twenty 32-bit scalar adds per wave.  (The flat form, `global_load_dword v, v_lane_offset, s[base:base+1]`, needs a 64-bit
// base per plane: 2 scalar instructions per plane, computed once for the loads and again for the stores -- 80 of the ~370 scalar
// instructions a wave issued, and the scalar pipe, one instruction per cycle per CU against the four SIMDs' vector issue, is what
// the step's issue phase is bound by: tools/ubench/salu.)
struct UnitPlanes {
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t so[kComponents];
};

ILM_DEV UnitPlanes unit_planes(const float* chunk_base, int64_t stride, int first_slot) {
    UnitPlanes u;
    // raw buffer (stride 0), num_records = the chunk's bytes, DATA_FORMAT = 32 bits (0x00020000, the word gfx9 wants for untyped access)
    u.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)chunk_base, 0, (int)(stride * (kComponents * 4)), 0x00020000);
    const uint32_t s4 = (uint32_t)stride * 4u;
    u.so[0] = (uint32_t)first_slot * 4u;
#pragma unroll
    for (int c = 1; c < kComponents; c++) u.so[c] = u.so[c - 1] + s4;
    return u;
}

// STREAM: the launch's working set is larger than the Infinity Cache (api.hip decides), every plane is touched once per step: loads and
// stores carry the non-temporal hint so they do not evict each other on the way through (tools/ubench/stream: 8.4 M slots 187 -> 170 us;
// on a cache-resident working set the same hint costs 25 %, so small systems keep the default policy).
// Cache policy bits of the plane accesses (buffer intrinsic aux word on gfx94x / gfx950: 1 = sc0, 2 = nt, 16 = sc1).
// Stores of the cache-resident variant carry sc1: they write through the XCD's L2 instead of leaving dirty lines there.  Every L2 is
// private to its XCD, so a kernel's release writes back whatever is still dirty before the next launch of the stream may start --
// up to 8 x 4 MB after a cfg2 step, ~6 us during which nothing runs (per-wave timestamps, tools/step_trace_probe.py: the waves of a
// 16-chunk launch span 17.5 us, back-to-back launches took 24).  The written planes are read next by another launch, on whichever XCD,
// after an invalidate: keeping them in this L2 buys nothing.  tools/step_ab.py: cfg2 without a spawner 20.3 -> 17.5 us per step, with
// 23.4 -> 22.0 (21.0 -> 19.4 / 24.5 -> 23.5 on one stream); sc0, nt, nt + sc1 and non-temporal loads all lose on a resident working set.
#ifndef ILM_LD_AUX
#define ILM_LD_AUX 0
#endif
#ifndef ILM_ST_AUX
#define ILM_ST_AUX 16
#endif
#ifndef ILM_LD_AUX_STREAM
#define ILM_LD_AUX_STREAM 2
#endif
#ifndef ILM_ST_AUX_STREAM
#define ILM_ST_AUX_STREAM 2
#endif
template <bool STREAM>
ILM_DEV float ld_plane(const UnitPlanes& u, int c, unsigned lane4) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(u.rsrc, (int)lane4, (int)u.so[c], STREAM ? ILM_LD_AUX_STREAM : ILM_LD_AUX));
}
template <bool STREAM>
ILM_DEV void st_plane(const UnitPlanes& u, int c, unsigned lane4, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), u.rsrc, (int)lane4, (int)u.so[c], STREAM ? ILM_ST_AUX_STREAM : ILM_ST_AUX);
}

template <bool ATTR, bool STREAM>
ILM_DEV SlotIn load_slot(const UnitPlanes& u, unsigned lane4) {
    SlotIn s;
    s.life = ld_plane<STREAM>(u, 3, lane4);
    s.px = ld_plane<STREAM>(u, 0, lane4); s.py = ld_plane<STREAM>(u, 1, lane4); s.pz = ld_plane<STREAM>(u, 2, lane4);
    s.vx = ld_plane<STREAM>(u, 4, lane4); s.vy = ld_plane<STREAM>(u, 5, lane4); s.vz = ld_plane<STREAM>(u, 6, lane4); s.ct = ld_plane<STREAM>(u, 7, lane4);
    if (ATTR) {
        s.ar = ld_plane<STREAM>(u, 8, lane4); s.ag = ld_plane<STREAM>(u, 9, lane4); s.ab = ld_plane<STREAM>(u, 10, lane4); s.aa = ld_plane<STREAM>(u, 11, lane4);
    } else {
        s.ar = s.ag = s.ab = s.aa = 0.0f;
    }
    return s;
}

// The launch descriptor lives in the kernarg segment (constant address space) and the per-unit body reads
// every parameter through a pointer to it (s_load from a uniform address).  The persistent loop launders
// that pointer through an empty asm each iteration: otherwise LICM hoists all ~150 scalar parameters out
// of the loop, overflows the 102 SGPRs and spills them into VGPR lanes and scratch.
typedef const StepLaunch __attribute__((address_space(4))) CStepLaunch;

// DF: the update pass is UpdateWithDistanceField (pulls in the SDF sampler); SPAWN: spawn records present.
// Both are compile-time so the common no-field / no-spawn step does not pay their registers.
template <int FMT, bool DF, bool SPAWN, bool EXT, bool STREAM>
ILM_DEV bool process_unit(CStepLaunch* ap, const UnitPlanes& up, int chunk, int i, unsigned lane, int seg, SlotIn cur, const NoiseDeltas& noise) {
    const StepLaunch& a = *(const StepLaunch*)ap;
    const IlmStepDesc& d = a.desc;
    const int64_t S = a.stride;
    const unsigned lane4 = lane * 4u;
    const int mode = d.UpdateMode;
    const bool need_attr = (mode == ILM_UPDATE_POSITIONS) || (mode == ILM_UPDATE_WITH_DISTANCE_FIELD);
    // Noise has no life check (Noise.fx:40): dead slots go through it.  That only matters when its result survives --
    // no update pass follows (single-pass ilm_noise), or the op can bring a dead slot back to life (StepDerived).
    const uint32_t noise_ops = EXT ? ((1u << ILM_OP_NOISE) | (1u << ILM_OP_SPATIAL_NOISE)) : (1u << ILM_OP_NOISE);
    const bool has_noise = ((a.op_mask & noise_ops) != 0u) && ((mode == ILM_UPDATE_NONE) || (a.derived.noise_may_revive != 0));
    const float4 zero = mk4(0.0f, 0.0f, 0.0f, 0.0f);

    bool spawn_here = false;
    if constexpr (SPAWN) {
        for (int s = 0; s < d.SpawnCount; s++) {
            const IlmSpawnRecord& r = d.Spawns[s];
            if (r.ChunkIndex == chunk && (float)i >= r.Params.ChunkSizeAndIndices[1] && (float)i <= r.Params.ChunkSizeAndIndices[2])
                spawn_here = true;
        }
    }
    bool live_after = false;
    int sdf_samples = 0;
    // Stride padding (chunk sizes whose square is not a multiple of 1024: 10, 16, 48 ...): not a slot of the chunk.  Noise has no life
    // check, so without this a padding lane could be given a life, be updated and be counted (CountLiveParticles.fx counts ChunkSize^2 pixels).
    if (i >= a.slots)
        return false;
    if (mode == ILM_UPDATE_ERASE) {
        // PS_Erase, UpdateParticleSystem.fx:40-49
#pragma unroll
        for (int c = 0; c < 8; c++) st_plane<STREAM>(up, c, lane4, 0.0f);
#pragma unroll
        for (int c = 12; c < 20; c++) st_plane<STREAM>(up, c, lane4, 0.0f);
    } else if ((cur.life <= 0.0f) && !spawn_here && !has_noise) {     // `<= 0` as the shaders test it: a NaN life is not dead
        // dead and nothing writes it: the update pass leaves the cleared target
        // (UpdateHandler._BeforeDraw clears, ParticleTransform.cs:164-165; readStateOrDiscard discards)
        if (mode != ILM_UPDATE_NONE) {
#pragma unroll
            for (int c = 0; c < 8; c++) st_plane<STREAM>(up, c, lane4, 0.0f);
#pragma unroll
            for (int c = 12; c < 20; c++) st_plane<STREAM>(up, c, lane4, 0.0f);
        } else {
            live_after = cur.life > 0.0f;   // untouched slots keep their liveness when no update pass ran
        }
    } else {
        float4 pos = mk4(cur.px, cur.py, cur.pz, cur.life);
        float4 vel = mk4(cur.vx, cur.vy, cur.vz, cur.ct);
        float4 attr = mk4(cur.ar, cur.ag, cur.ab, cur.aa);
        // slot (x, y): the unit's first slot is divided on the scalar unit, lanes only fold the row wrap
        const int cs = a.chunk_size;
        const int row0 = (a.derived.cs_shift >= 0) ? ((seg * 64) >> a.derived.cs_shift) : __builtin_amdgcn_readfirstlane((seg * 64) / cs);
        int sy = row0, sx = (seg * 64 - row0 * cs) + (int)lane;
        while (sx >= cs) { sx -= cs; sy++; }
        const float fx = (float)sx, fy = (float)sy;
        bool spawned = false;

        if constexpr (SPAWN) {
            if (spawn_here) {
                for (int s = 0; s < d.SpawnCount; s++) {
                    if (d.Spawns[s].ChunkIndex == chunk) {
                        bool wrote;
                        if (EXT && d.Spawns[s].Kind == ILM_SPAWN_POSITION_BUFFER)
                            wrote = spawn_slot_position_buffer(pos, vel, attr, fx, fy, a.rnd, a.rw, a.rh, a.derived.inv_rw, a.derived.inv_rh,
                                                               d.Spawns[s].Params, a.spawn_positions[s], a.spawn_position_count[s]);
                        else if (EXT && d.Spawns[s].Kind == ILM_SPAWN_FEEDBACK)
                            wrote = spawn_slot_feedback(pos, vel, attr, fx, fy, a.rnd, a.rw, a.rh, a.derived.inv_rw, a.derived.inv_rh,
                                                        d.Spawns[s].Params, d.Spawns[s].Feedback, a.source_base[s], S, cs);
                        else if (EXT && d.Spawns[s].Kind == ILM_SPAWN_PATTERN)
                            wrote = spawn_slot_pattern(pos, vel, attr, fx, fy, a.rnd, a.rw, a.rh, a.derived.inv_rw, a.derived.inv_rh,
                                                       d.Spawns[s].Params, d.Spawns[s].Pattern, a.spawn_pattern[s], a.pattern_w[s],
                                                       a.pattern_h[s], a.pattern_levels[s]);
                        else
                            wrote = spawn_slot(pos, vel, attr, fx, fy, a.rnd, a.rw, a.rh, a.derived.inv_rw, a.derived.inv_rh, d.Spawns[s].Params);
                        if (wrote)
                            spawned = true;
                    }
                }
            }
        }

        for (int o = 0; o < d.OpCount; o++) {
            const IlmTransformOp& op = d.Ops[o];
            if (op.Type == ILM_OP_GRAVITY)
                apply_gravity(pos, vel, d.System, op.u.Gravity, a.derived.op[o]);
            else if (op.Type == ILM_OP_NOISE)
                apply_noise(pos, vel, fx, fy, a.rnd, a.rw, a.rh, d.System, op.u.Noise, a.derived.inv_rw, a.derived.inv_rh, a.derived.op[o],
                            (o == a.derived.noise.op) ? noise : NoiseDeltas{ false, zero, zero });
            else if (op.Type == ILM_OP_FMA)
                apply_fma(pos, vel, d.System, op.u.FMA, a.derived.op[o]);
            else if (EXT && op.Type == ILM_OP_MATRIX_MULTIPLY)
                apply_matrix_multiply(pos, vel, d.System, op.u.MatrixMultiply);
            else if (EXT && op.Type == ILM_OP_SPATIAL_NOISE)
                apply_spatial_noise(pos, vel, a.rnd_lp, a.rw, a.rh, d.System, op.u.SpatialNoise, a.derived);
        }

        float4 rc = zero, rd = zero;
        if (need_attr) {
            if (pos.w <= 0.0f) {
                pos = vel = zero;  // readStateOrDiscard: discard => cleared target
            } else {
                if constexpr (DF)
                    update_with_distance_field<FMT>(pos, vel, fx, fy, d.System, a.derived.dt_s, d.DistanceField, a.sdf, sdf_samples);
                else
                    update_positions(pos, vel, d.System, a.derived.dt_s);
                render_data(fx, fy, pos, vel, attr, d.System, d.Update, a.derived.bezier_codes, a.derived.update_bits, a.ramp, a.ramp_w, a.ramp_h, rc, rd);
            }
        }
        st_plane<STREAM>(up, 0, lane4, pos.x); st_plane<STREAM>(up, 1, lane4, pos.y); st_plane<STREAM>(up, 2, lane4, pos.z); st_plane<STREAM>(up, 3, lane4, pos.w);
        st_plane<STREAM>(up, 4, lane4, vel.x); st_plane<STREAM>(up, 5, lane4, vel.y); st_plane<STREAM>(up, 6, lane4, vel.z); st_plane<STREAM>(up, 7, lane4, vel.w);
        if (spawned) {
            st_plane<STREAM>(up, 8, lane4, attr.x); st_plane<STREAM>(up, 9, lane4, attr.y); st_plane<STREAM>(up, 10, lane4, attr.z); st_plane<STREAM>(up, 11, lane4, attr.w);
        }
        if (need_attr) {
            st_plane<STREAM>(up, 12, lane4, rc.x); st_plane<STREAM>(up, 13, lane4, rc.y); st_plane<STREAM>(up, 14, lane4, rc.z); st_plane<STREAM>(up, 15, lane4, rc.w);
            st_plane<STREAM>(up, 16, lane4, rd.x); st_plane<STREAM>(up, 17, lane4, rd.y); st_plane<STREAM>(up, 18, lane4, rd.z); st_plane<STREAM>(up, 19, lane4, rd.w);
        }
        live_after = pos.w > 0.0f;
    }
    if constexpr (DF) {
        // diagnostic count (one atomic per lane: only ever on while bench.py takes the collision row's sample count)
        if ((__builtin_amdgcn_readfirstlane(g_step_count_sdf_samples) != 0) && (sdf_samples != 0))
            atomicAdd(&g_step_sdf_samples, (unsigned long long)sdf_samples);
    }

    return live_after;
}

// CountLiveParticles.fx for a block of the step kernels: LDS sum over the block's waves, ONE 64-bit atomic per block on a bucket
// counter of its chunk (live particles in the low word, a ticket in the high word); the block that draws a bucket's last ticket
// carries the bucket's sum to the chunk's counter the same way, and the one that completes the chunk stores the total, tagged with
// the step's sequence number, straight into the host's table.  (Per-wave atomics on one address serialise at ~11 ns each: 1024 of
// them per chunk made the step 7x slower; one per block on ONE address per chunk still queued 4096 deep on 1024^2 chunks.)  The
// units of a block always belong to one chunk.
ILM_DEV void publish_block_count(uint32_t* wave_live, uint32_t n_live, unsigned lane, int wave, bool block_in_range, int chunk, int block_in_chunk,
                                 int blocks_per_chunk, int buckets, unsigned long long* live_counts, unsigned long long* zero_counts, int zero_n,
                                 unsigned long long* host_counts, uint32_t seq) {
    if (blockIdx.x == 0)
        for (int i = (int)threadIdx.x; i < zero_n; i += kStepThreads) zero_counts[i * kCountStride] = 0ull;
    if (lane == 0) wave_live[wave] = n_live;
    __syncthreads();
    if (threadIdx.x == 0 && block_in_range) {
        uint32_t block_live = 0;
#pragma unroll
        for (int w = 0; w < kStepThreads / 64; w++) block_live += wave_live[w];
        unsigned long long* lines = live_counts + (size_t)chunk * (kCountLines * kCountStride);
        const int bucket = block_in_chunk & (buckets - 1);
        const unsigned long long old = atomicAdd(&lines[(1 + bucket) * kCountStride], (1ull << 32) | (unsigned long long)block_live);
        if ((int)(old >> 32) + 1 == blocks_per_chunk / buckets) {
            const unsigned long long sum = (old & 0xFFFFFFFFull) + (unsigned long long)block_live;
            const unsigned long long old2 = atomicAdd(&lines[0], (1ull << 32) | sum);
            if ((int)(old2 >> 32) + 1 == buckets) {
                const unsigned long long total = (old2 & 0xFFFFFFFFull) + sum;
                __hip_atomic_store(&host_counts[chunk], ((unsigned long long)seq << 32) | (total & 0xFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// The launch descriptor is written by the host into the kernarg ring just before the launch, so the first wave of every scalar cache
// misses on each of its 64-byte lines all the way to memory -- and the step reads them one dependent phase after another (decode,
// planes, noise tables, each transform, the update pass): a chain of ~20 serial misses at the head of every launch.  One load per
// line, all in flight at once, turns the chain into a single miss; for every later wave they are ~60 cache hits the scalar pipe has
// room for (tools/step_ab.py with 200 extra scalar instructions per wave: no change in step time).  Measured (r02): a one-chunk launch
// 7.7 -> 6.0 us back to back (10.8 -> 8.3 us when it spawns), cfg2 without a spawner 24.0 -> 21.4 us per step.
constexpr int kTouchBlocks = 2048 / (kStepThreads / 64);      // the launch's first 2 048 waves
// The chunk table is read through the constant address space in both kernels: see step_lean_kernel.
typedef float* const __attribute__((address_space(4))) CBase;
#define ILM_T1(o) "s_load_dword %0, %1, " #o "\n"
#define ILM_T4(o) ILM_T1(o) ILM_T1(o + 0x40) ILM_T1(o + 0x80) ILM_T1(o + 0xc0)
#define ILM_T16(o) ILM_T4(o) ILM_T4(o + 0x100) ILM_T4(o + 0x200) ILM_T4(o + 0x300)
ILM_DEV void touch_kernarg_lines_lean() {       // LeanStep: 51 lines
    const auto kp = __builtin_amdgcn_kernarg_segment_ptr();
    uint32_t sink;
    asm volatile(ILM_T16(0x0) ILM_T16(0x400) ILM_T16(0x800) ILM_T1(0xc00) ILM_T1(0xc40) ILM_T1(0xc80) "s_waitcnt lgkmcnt(0)"
                 : "=&s"(sink) : "s"(kp));
}
ILM_DEV void touch_kernarg_lines_step() {       // StepLaunch: 63 lines
    const auto kp = __builtin_amdgcn_kernarg_segment_ptr();
    uint32_t sink;
    asm volatile(ILM_T16(0x0) ILM_T16(0x400) ILM_T16(0x800) ILM_T4(0xc00) ILM_T4(0xd00) ILM_T4(0xe00) ILM_T1(0xf00) ILM_T1(0xf40) ILM_T1(0xf80)
                 "s_waitcnt lgkmcnt(0)" : "=&s"(sink) : "s"(kp));
}
static_assert(sizeof(StepLaunch) >= 0xf84 && sizeof(StepLaunch) <= 0xfc0, "touch_kernarg_lines_step reads one dword of each 64-byte line of StepLaunch");

// One wave = one unit of 64 consecutive slots; the hardware dispatcher balances the waves.  (A persistent,
// software-pipelined variant of this kernel measured 12-25 % slower: the body is a long dependent chain --
// state loads, scalar parameter fetches, randomness gathers -- whose latency is hidden by wave occupancy,
// not by prefetching; see DESIGN.md.)  MINW = minimum waves per SIMD requested from the register allocator.
template <int FMT, bool DF, bool SPAWN, int MINW, bool EXT = false, bool STREAM = false>
__global__ __launch_bounds__(kStepThreads, MINW) void step_kernel(const StepLaunch a) {
    __shared__ uint32_t wave_live[kStepThreads / 64];
    CStepLaunch* ap = (CStepLaunch*)__builtin_amdgcn_kernarg_segment_ptr();
    if (blockIdx.x < kTouchBlocks) touch_kernarg_lines_step();
    // the wave index is uniform by construction; saying so keeps the unit / chunk / base-pointer arithmetic on the
    // scalar unit and lets every plane access use the SGPR-base + 32-bit lane-offset addressing form
    const unsigned lane = threadIdx.x & 63u;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    // Blocks are rotated so that block 0 holds the first unit of the first spawn range: a spawning wave is a long
    // dependent chain (randomness gathers, sin/cos/acos), so it has to start first to finish under the cover of
    // the streaming waves instead of forming the tail of the launch.  unit_rotate is a multiple of the units per block.
    constexpr int K = kUnitsPerWave;
    int v = (int)blockIdx.x * (kStepThreads / 64) * K + a.unit_rotate;      // first unit of the block
    const int total = a.unit_end - a.unit_begin;
    if (v >= a.total_padded) v -= a.total_padded;
    const int u = a.unit_begin + v + wave * K;                               // first of this wave's K consecutive units
    // units_per_chunk is a multiple of 16 = (4 waves) x (K <= 4): the K units of a wave (and the whole block) lie in
    // one chunk, and `total` is a multiple of 16, so the K units are active or inactive together
    const bool active = (v + wave * K) < total;
    uint32_t n_live = 0;
    if (active) {
        // (a scalar integer division costs ~35 instructions per wave)
        const int chunk_rel = (a.upc_shift >= 0) ? (u >> a.upc_shift) : (u / a.units_per_chunk);
        const int seg = u - chunk_rel * a.units_per_chunk;
        const int chunk = a.first_chunk + chunk_rel;
        // never-written tail of a spawn-target chunk: all planes are zero and stay zero (scalar compares, uniform branch)
        bool untouched = false;
        for (int k = 0; k < a.partial_count; k++)
            untouched = untouched || ((a.partial_chunk[k] == chunk) && (seg >= a.partial_units[k]));
        if (!untouched) {
        const float* chunk_base = ((CBase*)a.chunk_bases)[chunk];
        const unsigned lane4 = lane * 4u;
        UnitPlanes up[K];
#pragma unroll
        for (int j = 0; j < K; j++) up[j] = unit_planes(chunk_base, a.stride, (seg + j) * 64);
        // K > 1 issues the state loads of all K units before any arithmetic (K x 12 loads in flight per wave).  Measured
        // on cfg2 ("experiments"): K = 1 26.1 us, K = 2 30.5 us, K = 4 36.5 us per step -- the extra
        // registers cost more occupancy than the memory-level parallelism returns, so K = 1 ships.
        SlotIn q[K];
#pragma unroll
        for (int j = 0; j < K; j++) q[j] = load_slot<true, STREAM>(up[j], lane4);
#pragma nounroll
        for (int j = 0; j < K; j++) {
            const SlotIn cur = q[0];
#pragma unroll
            for (int r = 0; r + 1 < K; r++) q[r] = q[r + 1];
            NoiseDeltas noise;
            noise.valid = false;
            if (a.derived.noise.op >= 0) {
                // first slot of the unit -> (x0, y): scalar; the unit lies in one row (chunk size a multiple of 64)
                const int first = (seg + j) * 64;
                const int row = (a.derived.cs_shift >= 0) ? (first >> a.derived.cs_shift) : (first / a.chunk_size);
                noise = noise_prepare(((const StepLaunch*)ap)->derived.noise, ((const StepLaunch*)ap)->desc, first - row * a.chunk_size, row);
            }
            const bool live_after = process_unit<FMT, DF, SPAWN, EXT, STREAM>(ap, up[j], chunk, (seg + j) * 64 + (int)lane, lane, seg + j, cur, noise);
            n_live += (uint32_t)__popcll(__ballot(live_after));
        }
        }
    }
    if (a.desc.Flags & ILM_STEP_COUNT_LIVE) {
        const int first_unit = a.unit_begin + v;
        const int chunk_rel = (a.upc_shift >= 0) ? (first_unit >> a.upc_shift) : (first_unit / a.units_per_chunk);
        constexpr int kUnitsPerBlock = (kStepThreads / 64) * K;
        publish_block_count(wave_live, n_live, lane, wave, v < total, a.first_chunk + chunk_rel, (first_unit - chunk_rel * a.units_per_chunk) / kUnitsPerBlock,
                            a.units_per_chunk / kUnitsPerBlock, a.count_buckets, a.live_counts, a.zero_counts, a.zero_n, a.host_counts, a.count_seq);
    }
}

// ---------------------------------------------------------------------------------------------
// the lean step kernel
// ---------------------------------------------------------------------------------------------
// step_kernel above interprets an arbitrary IlmStepDesc: per wave ~370 scalar-ALU instructions, ~70 scalar loads and ~60 branches next
// to ~365 vector instructions.  The scalar pipe issues one instruction per cycle per CU against the four SIMDs' vector issue
// (tools/ubench/salu: ~550 G/s against ~950 G/s), and every dependent scalar load the interpreter waits for is latency in the life of
// a wave whose whole launch is only two to three wave generations long -- so on a cache-resident system (cfg2) the step's time
// follows the scalar work, not the bytes.  step_lean_kernel runs the common shape of a step -- power-of-two chunk size >= 64,
// UpdatePositions, Gravity / Noise (no area, wave-uniform deltas available) / FMA (no area) in any order, inline spawners, no life
// ramp, no Noise that can revive a slot -- from a pre-digested descriptor (LeanStep, built by launch_step from the same StepLaunch):
// every uniform decision arrives as an integer, the attractors as one 32-byte record each, the spawn ranges as unit ranges.  The
// per-slot arithmetic is the interpreter's own (same functions or the same operations in the same order): ILM_STEP_LEAN=0 runs the
// interpreter instead and tests/test_properties_gpu.py requires the two to agree bit for bit.
struct LeanAttractor { float x, y, z, radius, strength, _p0, _p1, _p2; };
constexpr int kLeanMaxAttractors = 8;
struct LeanGravity {
    int32_t count; uint32_t types;   // 2 bits per attractor: 0 physical (ars.z < 0.5), 1 linear, 2 squared (Gravity.fx:36-52)
    float max_accel, _pad0;
    float cat_lo, cat_hi, _pad1, _pad2;
    LeanAttractor a[kLeanMaxAttractors];
};
union LeanOp {
    LeanGravity gravity;
    IlmNoiseParams noise;
    IlmFMAParams fma;
};
struct LeanStep {
    // decode
    float* const* chunk_bases;
    unsigned long long* live_counts; unsigned long long* zero_counts; unsigned long long* host_counts;
    int64_t stride;
    int32_t upc_shift, unit_rotate, total_padded, total_units;
    int32_t first_chunk, cs_shift, zero_n; uint32_t flags;
    uint32_t count_seq; int32_t count_buckets; int32_t _pad3[2];
    int32_t partial_chunk[kMaxPartialChunks], partial_units[kMaxPartialChunks];     // unused entries: chunk -1
    int32_t partial_count, op_count, spawn_count, _pad0;
    int32_t op_type[ILM_MAX_OPS];
    int32_t spawn_chunk[ILM_MAX_SPAWNS], spawn_unit_lo[ILM_MAX_SPAWNS], spawn_unit_hi[ILM_MAX_SPAWNS];   // segments of the target chunk a record touches
    int32_t _pad1[2];
    // update pass
    float dt_s; uint32_t bezier_codes, update_bits; int32_t noise_op;
    IlmParticleSystemUniforms sys;
    IlmUpdateParams update;
    // transforms
    StepDerived::Op dop[ILM_MAX_OPS];
    LeanOp op[ILM_MAX_OPS];
    StepDerived::NoiseFast noise;
    // spawners (inline kind only) -- or, in a launch without them, the 5 x 5 noise tables (kNoiseBigClasses)
    const float4* rnd; int32_t rw, rh; float inv_rw, inv_rh; int32_t _pad2[2];
    union {
        IlmSpawnRecord spawns[ILM_MAX_SPAWNS];
        IlmFloat4 noise_big[2 * kNoiseBigClasses * kNoiseBigClasses];
    };
};
static_assert(sizeof(LeanStep) <= 4096, "LeanStep travels in the kernarg segment (4 KB)");

// PS_Gravity with the attractor records and type codes of LeanGravity: the operations of apply_gravity, in its order
ILM_DEV void apply_gravity_lean(float4& pos, float4& vel, const LeanGravity& g, float dt_ms, float mv) {
    if ((pos.w <= 0.0f) || !((vel.w >= g.cat_lo) && (vel.w <= g.cat_hi)))
        return;
    f3 acceleration = mk3(0.0f, 0.0f, 0.0f);
    const uint32_t types = g.types;
    int i = 0;
#ifndef ILM_GRAVITY_EXACT
    // Four attractors at a time while none of the four is of the physical type (the one with the IEEE division): their records come
    // in together, every decision is a select on a uniform mask, so the four dependent chains (subtract, dot, rsq, rcp, ...) sit in
    // one basic block and the scheduler interleaves them.  The same operations per attractor, the terms added in index order: the
    // same bits as the loop below.  (A wave alone on its SIMD spent 1.24 us of its 3.4 in the transforms, one attractor after the
    // other behind a scalar load each -- and that latency is what the head and the tail of every launch consist of.)
    for (; i + 4 <= g.count && ((((types >> (2 * i)) | (types >> (2 * i + 1))) & 0x55u) == 0x55u); i += 4) {
        f3 term[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const LeanAttractor A = g.a[i + k];
            const bool squared = ((types >> (2 * (i + k))) & 2u) != 0u;
            const f3 to_center = mk3(A.x, A.y, A.z) - xyz(pos);
            const float d2 = dot3(to_center, to_center);
            const float inv_len = fast_rsq(d2);
            const float distance = d2 * inv_len;
            float attraction = 1.0f - sat(distance * fast_rcp(A.radius));
            attraction = squared ? attraction * attraction : attraction;
            attraction = attraction * dt_ms * (1.0f / kVelocityConstantScale);
            term[k] = ((to_center * inv_len) * attraction) * A.strength;
        }
        acceleration = (((acceleration + term[0]) + term[1]) + term[2]) + term[3];
    }
#endif
    for (; i < g.count; i++) {
        const LeanAttractor A = g.a[i];
        const uint32_t type = (types >> (2 * i)) & 3u;
        const f3 to_center = mk3(A.x, A.y, A.z) - xyz(pos);
        float attraction;
        const float d2 = dot3(to_center, to_center);
#ifndef ILM_GRAVITY_EXACT
        const float inv_len = fast_rsq(d2);
        if (type != 0u) {
            const float distance = d2 * inv_len;
            attraction = 1.0f - sat(distance * fast_rcp(A.radius));
            if (type == 2u)
                attraction *= attraction;
            attraction = attraction * dt_ms * (1.0f / kVelocityConstantScale);
        } else {
            const float distance_squared = fmaxf(d2 - A.radius, 0.001f);
            attraction = 1.0f / distance_squared;
        }
        acceleration = acceleration + (((to_center * inv_len) * attraction) * A.strength);
#else
        const float distance = sqrtf(d2);
        if (type != 0u) {
            attraction = 1.0f - sat(distance / A.radius);
            if (type == 2u)
                attraction *= attraction;
            attraction = attraction * dt_ms / kVelocityConstantScale;
        } else {
            const float distance_squared = fmaxf(d2 - A.radius, 0.001f);
            attraction = 1.0f / distance_squared;
        }
        const f3 n = mk3(to_center.x / distance, to_center.y / distance, to_center.z / distance);
        acceleration = acceleration + ((n * attraction) * A.strength);
#endif
    }
    const float maximum_acceleration = g.max_accel;
#ifndef ILM_GRAVITY_EXACT
    const float a2 = dot3(acceleration, acceleration);
    if (a2 > maximum_acceleration * maximum_acceleration)
        acceleration = acceleration * (fast_rsq(a2) * maximum_acceleration);
#else
    const float current_length = len3(acceleration);
    if (current_length > maximum_acceleration)
        acceleration = mk3(acceleration.x / current_length, acceleration.y / current_length, acceleration.z / current_length) * maximum_acceleration;
#endif
    vel.x = fminf(mv, vel.x + acceleration.x);
    vel.y = fminf(mv, vel.y + acceleration.y);
    vel.z = fminf(mv, vel.z + acceleration.z);
}

// noise_prepare on LeanStep: class counts as sign bits (2 scalar instructions per boundary), one table for both class counts
ILM_DEV NoiseDeltas noise_prepare_lean(const LeanStep& a, int x0, int row) {
    const StepDerived::NoiseFast& nf = a.noise;
    NoiseDeltas out;
    const uint32_t code = nf.wcode[x0 >> 6];
    out.valid = (code & 64u) != 0u;
    uint32_t yc0 = 0, yc1 = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int b = nf.yb[k];                              // a row >= 0 or INT32_MAX
        yc0 += (uint32_t)(b - 1 - row) >> 31;                // row >= b
        yc1 += (uint32_t)(b - 2 - row) >> 31;                // row + 1 >= b
    }
    const uint32_t xc0 = code & 7u, xc1 = (code >> 3) & 7u;
    const uint32_t n = (uint32_t)nf.classes;
    const IlmFloat4* table = (n == (uint32_t)kNoiseBigClasses) ? a.noise_big : &nf.position[0][0];   // position[n][n] then velocity[n][n]
    out.position = ld4(table[yc0 * n + xc0]);
    out.velocity = ld4(table[n * n + yc1 * n + xc1]);
    return out;
}
static_assert(offsetof(StepDerived::NoiseFast, velocity) == offsetof(StepDerived::NoiseFast, position) + 9 * sizeof(IlmFloat4), "velocity[3][3] follows position[3][3]");

typedef const LeanStep __attribute__((address_space(4))) CLeanStep;
static_assert(sizeof(LeanStep) >= 0xc84 && sizeof(LeanStep) <= 0xcc0, "touch_kernarg_lines_lean reads one dword of each 64-byte line of LeanStep");


#ifdef ILM_STEP_TRACE      // EXPERIMENT (tools/step_trace_probe.py): per-wave start / loaded / end times of the last launch, 100 MHz clock
__device__ unsigned long long g_step_trace[5 * 131072];
extern "C" int ilm_experiment_step_trace(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_step_trace), sizeof(unsigned long long) * (size_t)n);
}
#endif
template <bool SPAWN, bool STREAM>
__global__ __launch_bounds__(kStepThreads) void step_lean_kernel(const LeanStep a_) {
    __shared__ uint32_t wave_live[kStepThreads / 64];
#ifdef ILM_STEP_TRACE
    const unsigned long long trace_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long trace_t1 = trace_t0, trace_t2 = trace_t0, trace_t3 = trace_t0;
#endif
    const LeanStep& a = *(const LeanStep*)(CLeanStep*)__builtin_amdgcn_kernarg_segment_ptr();
    // (only the launch's first generation of blocks can be the first to read a line; for the others the loads would just load the
    // scalar cache: one lookup per line per wave)
    if (blockIdx.x < kTouchBlocks) touch_kernarg_lines_lean();
    const unsigned lane = threadIdx.x & 63u;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    int v = (int)blockIdx.x * (kStepThreads / 64) + a.unit_rotate;      // first unit of the block (rotation: see step_kernel)
    if (v >= a.total_padded) v -= a.total_padded;
    const int u = v + wave;
    uint32_t n_live = 0;
    if (u < a.total_units) {
        const int chunk_rel = u >> a.upc_shift;
        const int seg = u - (chunk_rel << a.upc_shift);
        const int chunk = a.first_chunk + chunk_rel;
        bool untouched = false;
        if (a.partial_count != 0) {
#pragma unroll
            for (int k = 0; k < kMaxPartialChunks; k++)
                untouched = untouched || ((a.partial_chunk[k] == chunk) && (seg >= a.partial_units[k]));
        }
        if (!untouched) {
            const unsigned lane4 = lane * 4u;
            // the chunk table through the constant address space: a scalar load whatever the optimiser thinks may alias (behind the
            // volatile asm of touch_kernarg_lines it would otherwise fetch the base with a VECTOR load and wrap every plane access in
            // a waterfall loop over a "divergent" buffer resource: +160 vector instructions per wave); the table is written by a copy
            // that precedes the launch on its stream and never during one
            const UnitPlanes up = unit_planes(((CBase*)a.chunk_bases)[chunk], a.stride, seg * 64);
            const SlotIn cur = load_slot<true, STREAM>(up, lane4);
            // slot (x, y): the unit lies in one row (chunk size a multiple of 64)
            const int first = seg * 64;
            const int row = first >> a.cs_shift;
            const int x0 = first - (row << a.cs_shift);
            const float fx = (float)(x0 + (int)lane), fy = (float)row;
            NoiseDeltas noise;
            noise.valid = false;
            if (a.noise_op >= 0)
                noise = noise_prepare_lean(a, x0, row);

            float4 pos = mk4(cur.px, cur.py, cur.pz, cur.life);
            float4 vel = mk4(cur.vx, cur.vy, cur.vz, cur.ct);
            float4 attr = mk4(cur.ar, cur.ag, cur.ab, cur.aa);
#ifdef ILM_STEP_TRACE
            asm volatile("s_waitcnt vmcnt(0)");
            trace_t1 = __builtin_amdgcn_s_memrealtime();
#endif
            bool spawn_here = false, spawned = false;
            if constexpr (SPAWN) {
                for (int s = 0; s < a.spawn_count; s++) {
                    if (a.spawn_chunk[s] == chunk && seg >= a.spawn_unit_lo[s] && seg <= a.spawn_unit_hi[s]) {
                        const IlmSpawnRecord& r = a.spawns[s];
                        const float fi = (float)(first + (int)lane);
                        if (fi >= r.Params.ChunkSizeAndIndices[1] && fi <= r.Params.ChunkSizeAndIndices[2]) {
                            spawn_here = true;
                            if (spawn_slot(pos, vel, attr, fx, fy, a.rnd, a.rw, a.rh, a.inv_rw, a.inv_rh, r.Params))
                                spawned = true;
                        }
                    }
                }
            }
            // `<= 0` as the shaders test it: a NaN life is not dead.  A dead slot nothing writes keeps the cleared target's zeros.
            const bool process = !(cur.life <= 0.0f) || spawn_here;
            const float4 zero = mk4(0.0f, 0.0f, 0.0f, 0.0f);
            float4 rc = zero, rd = zero;
            if (__ballot(process) != 0ull) {
                if (process) {
                    for (int o = 0; o < a.op_count; o++) {
                        const int type = a.op_type[o];
                        if (type == ILM_OP_GRAVITY)
                            apply_gravity_lean(pos, vel, a.op[o].gravity, a.sys.GlobalSettings.x, a.sys.GlobalSettings.z);
                        else if (type == ILM_OP_NOISE)
                            apply_noise(pos, vel, fx, fy, a.rnd, a.rw, a.rh, a.sys, a.op[o].noise, a.inv_rw, a.inv_rh, a.dop[o],
                                        (o == a.noise_op) ? noise : NoiseDeltas{ false, zero, zero });
                        else
                            apply_fma(pos, vel, a.sys, a.op[o].fma, a.dop[o]);
                    }
#ifdef ILM_STEP_TRACE
                    asm volatile("" : "+v"(pos.x), "+v"(vel.x));
                    trace_t2 = __builtin_amdgcn_s_memrealtime();
#endif
                    if (pos.w <= 0.0f) {
                        pos = vel = zero;  // readStateOrDiscard: discard => cleared target
                    } else {
                        update_positions(pos, vel, a.sys, a.dt_s);
                        render_data(fx, fy, pos, vel, attr, a.sys, a.update, a.bezier_codes, a.update_bits, nullptr, 0, 0, rc, rd);
                    }
                } else {
                    pos = vel = zero;
                }
            } else {
                pos = vel = zero;
            }
#ifdef ILM_STEP_TRACE
            asm volatile("" : "+v"(pos.x), "+v"(rd.x), "+v"(rc.x));
            trace_t3 = __builtin_amdgcn_s_memrealtime();
#endif
            st_plane<STREAM>(up, 0, lane4, pos.x); st_plane<STREAM>(up, 1, lane4, pos.y); st_plane<STREAM>(up, 2, lane4, pos.z); st_plane<STREAM>(up, 3, lane4, pos.w);
            st_plane<STREAM>(up, 4, lane4, vel.x); st_plane<STREAM>(up, 5, lane4, vel.y); st_plane<STREAM>(up, 6, lane4, vel.z); st_plane<STREAM>(up, 7, lane4, vel.w);
            if constexpr (SPAWN) {
                if (spawned) {
                    st_plane<STREAM>(up, 8, lane4, attr.x); st_plane<STREAM>(up, 9, lane4, attr.y); st_plane<STREAM>(up, 10, lane4, attr.z); st_plane<STREAM>(up, 11, lane4, attr.w);
                }
            }
            st_plane<STREAM>(up, 12, lane4, rc.x); st_plane<STREAM>(up, 13, lane4, rc.y); st_plane<STREAM>(up, 14, lane4, rc.z); st_plane<STREAM>(up, 15, lane4, rc.w);
            st_plane<STREAM>(up, 16, lane4, rd.x); st_plane<STREAM>(up, 17, lane4, rd.y); st_plane<STREAM>(up, 18, lane4, rd.z); st_plane<STREAM>(up, 19, lane4, rd.w);
            n_live = (uint32_t)__popcll(__ballot(pos.w > 0.0f));
        }
    }
#ifdef ILM_STEP_TRACE
    if (lane == 0) {
        const unsigned w = (blockIdx.x * (kStepThreads / 64) + (unsigned)wave) & 131071u;
        g_step_trace[5 * w] = trace_t0; g_step_trace[5 * w + 1] = trace_t1; g_step_trace[5 * w + 2] = __builtin_amdgcn_s_memrealtime();
        g_step_trace[5 * w + 3] = trace_t2; g_step_trace[5 * w + 4] = trace_t3;
    }
#endif
    if (a.flags & ILM_STEP_COUNT_LIVE)
        publish_block_count(wave_live, n_live, lane, wave, v < a.total_units, a.first_chunk + (v >> a.upc_shift),
                            (v & ((1 << a.upc_shift) - 1)) / (kStepThreads / 64), (1 << a.upc_shift) / (kStepThreads / 64), a.count_buckets,
                            a.live_counts, a.zero_counts, a.zero_n, a.host_counts, a.count_seq);
}

// ---------------------------------------------------------------------------------------------
// the lean collision step (r06) -- UpdateParticleSystemWithDistanceField.fx:29-147 on the lean descriptor
// ---------------------------------------------------------------------------------------------
// The collision update is two things in one shader.  EVERY live particle samples the field where it is and where it is going (the
// first iteration of the sweep): two dependent lookups, the same for all lanes.  A particle whose first sweep lookup lands inside an
// obstacle (14 % of them on the demo's field) goes on: up to two more sweep lookups, estimateNormal4 (four more), the bounce / redirect
// / escape arithmetic -- about as many vector instructions again as the whole rest of the step, run by a wave for its few such lanes
// (the interpreter's step_kernel<.., DF>: 1 071 instructions per wave at 39.4 of 64 lanes, profiles/r06_collision_step.txt).
// Here a wave walks K consecutive units in three phases.
//   A  per unit: load, spawn, transforms, the common path at full width.  The lanes that collided are PARKED -- their state after the
//      transforms and the two distances already sampled, eleven words, in a ring of 128 entries the wave owns in LDS (no barrier: the
//      ring is private to the wave).  The unit's position and velocity stay in registers; nothing is stored yet.
//   L  whenever 64 are parked, and once more after the last unit: the parked particles, one per lane, through the rest of the reference
//      update (df_long: the operations of update_with_distance_field behind its first lookup); the results go back into
//      the same ring entries.
//   B  per unit: the parked lanes take their results from the ring, then computeRenderData and all sixteen stores at FULL width.
// So the long path runs at 64 lanes -- or, for the remainder, at whatever K units leave -- instead of at ~9 lanes once per unit, and
// every plane of a unit is still written by ONE full-width store (a first form of this kernel stored the finished lanes at once and
// the parked ones later, from the long pass: two partial writes per line, and cfg4's share took 507-597 us against the interpreter's
// 394 -- masked and scattered 4-byte stores cost a read-modify-write each once the line has left the L2).
// A ring that would overflow (more than 128 of a wave's K x 64 particles colliding) is not waited for: that unit's parked lanes take the
// long path in place, as the interpreter does.
struct LeanStepDf {
    LeanStep base;
    IlmDistanceFieldUniforms df;
    SdfView sdf;
};
static_assert(sizeof(LeanStepDf) <= 4096, "LeanStepDf travels in the kernarg segment (4 KB)");
static_assert(sizeof(LeanStepDf) > 0xd40 && sizeof(LeanStepDf) <= 0xe00, "touch_kernarg_lines_lean_df reads one dword of each 64-byte line of LeanStepDf");
ILM_DEV void touch_kernarg_lines_lean_df() {       // LeanStepDf: 54 .. 56 lines
    const auto kp = __builtin_amdgcn_kernarg_segment_ptr();
    uint32_t sink;
    asm volatile(ILM_T16(0x0) ILM_T16(0x400) ILM_T16(0x800) ILM_T4(0xc00) ILM_T1(0xd00) ILM_T1(0xd40) "s_waitcnt lgkmcnt(0)"
                 : "=&s"(sink) : "s"(kp));
}
typedef const LeanStepDf __attribute__((address_space(4))) CLeanStepDf;

constexpr int kDfRing = 128;                     // parked particles per wave
struct DfParked { float px[kDfRing], py[kDfRing], pz[kDfRing], life[kDfRing], vx[kDfRing], vy[kDfRing], vz[kDfRing], ct[kDfRing], d0[kDfRing], d1[kDfRing]; uint32_t slot[kDfRing]; };   // slot: bit 31 = d1 is the first iteration's lookup

// The common path of PS_Update (distance field) for a live slot (life > 0 on entry): everything a particle needs that meets no obstacle.
// The reference samples the field at the particle (initial_distance) and then at old + unit * travel with travel = max(0, min(
// initial_distance, |velocity| dt)) -- two DEPENDENT lookups.  Away from obstacles travel IS |velocity| dt, known before any lookup: both
// positions are sampled at once, and when min() did pick |velocity| dt (bit for bit) the second sample is the sweep's first
// step_distance -- same position, same bits.  Returns 0 when the particle is finished (position and velocity final; zeros for one that
// died); 1 when it must go on with the sweep's first iteration still to do (travel is not |velocity| dt: it sits at or inside an
// obstacle); 2 when the first iteration is done and collided (step_distance valid).  For 1 and 2 position and velocity are left as they
// came: the caller parks them for df_long.  Same operations in the same order as update_with_distance_field takes on these paths
// (tests/test_step_kernels_gpu.py holds the kernels bit-equal, sample counts included).
template <int FMT>
ILM_DEV int df_common_path(float4& pos, float4& vel, const IlmParticleSystemUniforms& sys, float dts, const IlmDistanceFieldUniforms& df, const SdfView& sdf, int& samples,
                           float& initial_distance, float& step_distance) {
#pragma clang fp contract(off)
    const float4 zero = mk4(0.0f, 0.0f, 0.0f, 0.0f);
    const float new_life = pos.w - (sys.GlobalSettings.w * dts);
    initial_distance = step_distance = 0.0f;
    if (new_life <= 0.0f) {
        pos = vel = zero;
        return 0;
    }
    const float collision_distance = sys.CollisionSettings.z;
    const f3 old_xyz = xyz(pos);
    const f3 unit_vector = norm3(xyz(vel));
    const f3 velocity = friction_and_maximum(xyz(vel), sys, dts);
    const f3 scaled_velocity = velocity * dts;
    const float reach = len3(scaled_velocity);
    const f3 ahead = old_xyz + (unit_vector * reach);
    initial_distance = sample_distance_field<(FMT & 1), true, (FMT & 2) != 0, (FMT & 4) != 0>(old_xyz, df, sdf);
    step_distance = sample_distance_field<(FMT & 1), true, (FMT & 2) != 0, (FMT & 4) != 0>(ahead, df, sdf);        // (independent of the first: both in flight together)
    samples++;
    const bool was_colliding = initial_distance < collision_distance;
    const float travel_distance = fmaxf(0.0f, fminf(initial_distance, reach));
    if (__builtin_bit_cast(uint32_t, travel_distance) != __builtin_bit_cast(uint32_t, reach))
        return 1;                                                 // the sweep starts somewhere else than `ahead`
    if (was_colliding || !(travel_distance <= 0.001f)) {          // step_count 1 or MAX_STEP_COUNT: the sweep's first iteration runs, at `ahead`
        samples++;
        if (step_distance < collision_distance)
            return 2;
    }
    pos = mk4(ahead.x, ahead.y, ahead.z, new_life);
    vel = mk4(velocity.x, velocity.y, velocity.z, fmaxf(vel.w - 1.0f, 0.0f));
    return 0;
}

// The rest of PS_Update (distance field) for a parked particle: update_with_distance_field with its first lookup (initial_distance) given
// and, when `first_done`, the first iteration's lookup (step_distance0) too.  Operation for operation the function above -- the uniform
// values and the vectors that depend on the inputs alone (unit vector, velocity after friction, travel distance) are formed again by
// the same operations.
template <int FMT>
ILM_DEV void df_long(float4& pos, float4& vel, float x, float y, float initial_distance, float step_distance0, bool first_done,
                     const IlmParticleSystemUniforms& sys, float dts, const IlmDistanceFieldUniforms& df, const SdfView& sdf, int& samples) {
#pragma clang fp contract(off)
    const float4 zero = mk4(0.0f, 0.0f, 0.0f, 0.0f);
    float new_life = pos.w - (sys.GlobalSettings.w * dts);
    const float collision_distance = sys.CollisionSettings.z;
    const float max_velocity = sys.GlobalSettings.z;
    const f3 old_xyz = xyz(pos);
    const f3 unit_vector = norm3(xyz(vel));
    const f3 velocity = friction_and_maximum(xyz(vel), sys, dts);
    const f3 scaled_velocity = velocity * dts;

    bool collided = false, escaping = false;
    f3 collision_position = mk3(0.0f, 0.0f, 0.0f), new_position = old_xyz;
    float4 new_velocity = zero;

    const bool was_colliding = initial_distance < collision_distance;
    float travel_distance = fmaxf(0.0f, fminf(initial_distance, len3(scaled_velocity)));
    int step_count = ref::kMaxStepCount;
    if (was_colliding)
        step_count = 1;
    else if (travel_distance <= 0.001f)
        step_count = 0;

    for (int i = 0; i < step_count; i++) {
        const f3 test_position = old_xyz + (unit_vector * travel_distance);
        float step_distance = step_distance0;
        if (!(first_done && i == 0)) {
            step_distance = sample_distance_field<(FMT & 1), true, (FMT & 2) != 0, (FMT & 4) != 0>(test_position, df, sdf);
            samples++;
        }
        if (step_distance < collision_distance) {
            collided = true;
            collision_position = test_position;
        }
        escaping = step_distance > initial_distance;
        if (collided && !escaping) {
            collision_position = test_position;
            const float offset = clampf(step_distance + collision_distance, 0.05f, 16.0f);
            travel_distance = fmaxf(0.0f, travel_distance - offset);
        } else
            step_count = 0;
        if (travel_distance <= 0.001f)
            step_count = 0;
    }

    if (collided) {
        const bool bounce = vel.w <= 0.0f;
        const bool redirect = was_colliding && !escaping;
        f3 normal = mk3(0.0f, 0.0f, 0.0f);
        if (bounce || redirect) {
            normal = estimate_normal4<FMT>(collision_position, df, sdf);
            samples += 4;
        }
        const float escape_speed = fminf(max_velocity, sys.CollisionSettings.x);
        if (redirect) {
            normal = normal * mk3(1.0f, 1.0f, 0.0f);  // ESCAPE_MASK
            if (len3(normal) < ref::kNoNormalThreshold) {
                const float a = (x / 67.0f) + (y / 13.0f);
                normal = mk3(sinf(a), cosf(a), 0.0f);
            }
            const f3 nv = (norm3(normal) * escape_speed) * ref::kInitialEscapeSpeed;
            new_velocity = mk4(nv.x, nv.y, nv.z, ref::kBounceDelay);
            new_position = old_xyz + (nv * dts);
        } else if (bounce) {
            const float d2 = 2.0f * dot3(normal, unit_vector);
            f3 bounce_vector = ((normal - unit_vector) * d2) * -1.0f;
            if (len3(bounce_vector) < ref::kNoNormalThreshold)
                bounce_vector = unit_vector * -1.0f;
            else
                bounce_vector = norm3(bounce_vector);
            new_position = collision_position;
            const f3 nv = bounce_vector * fminf(max_velocity, len3(velocity) * sys.CollisionSettings.y);
            new_velocity = mk4(nv.x, nv.y, nv.z, ref::kBounceDelay);
            new_life -= sys.CollisionSettings.w;
        } else {
            const float new_speed = fmaxf(len3(xyz(vel)) * ref::kEscapeSpeedAcceleration, escape_speed);
            const f3 nv = unit_vector * new_speed;
            new_velocity = mk4(nv.x, nv.y, nv.z, 0.0f);
            new_position = old_xyz + (unit_vector * travel_distance);
        }
    } else {
        new_velocity = mk4(velocity.x, velocity.y, velocity.z, fmaxf(vel.w - 1.0f, 0.0f));
        new_position = old_xyz + (unit_vector * travel_distance);
    }
    if (new_life <= 0.0f) {
        new_position = mk3(0.0f, 0.0f, 0.0f);
        new_velocity = zero;
    }
    pos = mk4(new_position.x, new_position.y, new_position.z, new_life);
    vel = new_velocity;
}

template <int FMT, bool SPAWN, bool STREAM, int K>
__global__ __launch_bounds__(kStepThreads) void step_lean_df_kernel(const LeanStepDf a_) {
    static_assert(K == 1 || K == 2 || K == 4, "units per wave");
    __shared__ uint32_t wave_live[kStepThreads / 64];
    __shared__ DfParked parked_all[kStepThreads / 64];
    const LeanStepDf& A = *(const LeanStepDf*)(CLeanStepDf*)__builtin_amdgcn_kernarg_segment_ptr();
    const LeanStep& a = A.base;
    if (blockIdx.x < kTouchBlocks / K) touch_kernarg_lines_lean_df();
    const unsigned lane = threadIdx.x & 63u;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    DfParked& parked = parked_all[wave];
    constexpr int kUnitsPerBlock = (kStepThreads / 64) * K;
    int v = (int)blockIdx.x * kUnitsPerBlock + a.unit_rotate;           // first unit of the block (rotation: see step_kernel; a multiple of kUnitsPerBlock)
    if (v >= a.total_padded) v -= a.total_padded;
    const int u0 = v + wave * K;                                        // this wave's K consecutive units: one chunk (units per chunk is a multiple of kUnitsPerBlock)
    uint32_t n_live = 0;
    int sdf_samples = 0;
    const float4 zero = mk4(0.0f, 0.0f, 0.0f, 0.0f);
    if (u0 < a.total_units) {
        const int chunk_rel = u0 >> a.upc_shift;
        const int seg0 = u0 - (chunk_rel << a.upc_shift);
        const int chunk = a.first_chunk + chunk_rel;
        int seg_end = 1 << a.upc_shift;                                 // units of this chunk that hold particles
        if (a.partial_count != 0) {
#pragma unroll
            for (int k = 0; k < kMaxPartialChunks; k++)
                if (a.partial_chunk[k] == chunk) seg_end = min(seg_end, a.partial_units[k]);
        }
        const int units = max(0, min(K, seg_end - seg0));               // (past seg_end: the never-written tail of a spawn-target chunk, zeros that stay zeros)
        float* const chunk_base = ((CBase*)a.chunk_bases)[chunk];
        const unsigned lane4 = lane * 4u;
        const unsigned long long lanes_below = (1ull << lane) - 1ull;
        // what a unit keeps in registers between phase A and phase B, and where its parked lanes sit in the ring
        float4 hp0 = zero, hp1 = zero, hp2 = zero, hp3 = zero, hv0 = zero, hv1 = zero, hv2 = zero, hv3 = zero;
        unsigned long long pm0 = 0ull, pm1 = 0ull, pm2 = 0ull, pm3 = 0ull;
        uint32_t pb0 = 0, pb1 = 0, pb2 = 0, pb3 = 0;
        uint32_t pending = 0, done = 0;                                 // ring entries [0, pending) hold parked particles, [0, done) their results

        // the parked particles [first, first + count), count <= 64, one per lane: the rest of the reference's update, results in place
        auto long_pass = [&](uint32_t first, uint32_t count) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < count) {
                const uint32_t e = first + lane;
                float4 pos = mk4(parked.px[e], parked.py[e], parked.pz[e], parked.life[e]);
                float4 vel = mk4(parked.vx[e], parked.vy[e], parked.vz[e], parked.ct[e]);
                const uint32_t tagged = parked.slot[e];
                const int slot = (int)(tagged & 0x7FFFFFFFu);
                const int sy = slot >> a.cs_shift;
                const float fx = (float)(slot - (sy << a.cs_shift)), fy = (float)sy;
                df_long<FMT>(pos, vel, fx, fy, parked.d0[e], parked.d1[e], (tagged >> 31) != 0u, a.sys, a.dt_s, A.df, A.sdf, sdf_samples);
                parked.px[e] = pos.x; parked.py[e] = pos.y; parked.pz[e] = pos.z; parked.life[e] = pos.w;
                parked.vx[e] = vel.x; parked.vy[e] = vel.y; parked.vz[e] = vel.z; parked.ct[e] = vel.w;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };

        // ---- phase A ----
#pragma nounroll
        for (int j = 0; j < units; j++) {
            const int seg = seg0 + j;
            const UnitPlanes up = unit_planes(chunk_base, a.stride, seg * 64);
            const SlotIn cur = load_slot<true, STREAM>(up, lane4);
            const int first = seg * 64;
            const int row = first >> a.cs_shift;
            const int x0 = first - (row << a.cs_shift);
            const float fx = (float)(x0 + (int)lane), fy = (float)row;
            NoiseDeltas noise;
            noise.valid = false;
            if (a.noise_op >= 0)
                noise = noise_prepare_lean(a, x0, row);
            float4 pos = mk4(cur.px, cur.py, cur.pz, cur.life);
            float4 vel = mk4(cur.vx, cur.vy, cur.vz, cur.ct);
            float4 attr = mk4(cur.ar, cur.ag, cur.ab, cur.aa);
            bool spawn_here = false, spawned = false;
            if constexpr (SPAWN) {
                for (int s = 0; s < a.spawn_count; s++) {
                    if (a.spawn_chunk[s] == chunk && seg >= a.spawn_unit_lo[s] && seg <= a.spawn_unit_hi[s]) {
                        const IlmSpawnRecord& r = a.spawns[s];
                        const float fi = (float)(first + (int)lane);
                        if (fi >= r.Params.ChunkSizeAndIndices[1] && fi <= r.Params.ChunkSizeAndIndices[2]) {
                            spawn_here = true;
                            if (spawn_slot(pos, vel, attr, fx, fy, a.rnd, a.rw, a.rh, a.inv_rw, a.inv_rh, r.Params))
                                spawned = true;
                        }
                    }
                }
            }
            const bool process = !(cur.life <= 0.0f) || spawn_here;
            int parking = 0;
            float d0 = 0.0f, d1 = 0.0f;
            if (__ballot(process) != 0ull) {
                if (process) {
                    for (int o = 0; o < a.op_count; o++) {
                        const int type = a.op_type[o];
                        if (type == ILM_OP_GRAVITY)
                            apply_gravity_lean(pos, vel, a.op[o].gravity, a.sys.GlobalSettings.x, a.sys.GlobalSettings.z);
                        else if (type == ILM_OP_NOISE)
                            apply_noise(pos, vel, fx, fy, a.rnd, a.rw, a.rh, a.sys, a.op[o].noise, a.inv_rw, a.inv_rh, a.dop[o],
                                        (o == a.noise_op) ? noise : NoiseDeltas{ false, zero, zero });
                        else
                            apply_fma(pos, vel, a.sys, a.op[o].fma, a.dop[o]);
                    }
                    if (pos.w <= 0.0f)
                        pos = vel = zero;  // readStateOrDiscard: discard => cleared target
                    else
                        parking = df_common_path<FMT>(pos, vel, a.sys, a.dt_s, A.df, A.sdf, sdf_samples, d0, d1);
                } else {
                    pos = vel = zero;
                }
            } else {
                pos = vel = zero;
            }
            if constexpr (SPAWN) {
                if (spawned) {      // (phase B reads the attributes back from the planes)
                    st_plane<STREAM>(up, 8, lane4, attr.x); st_plane<STREAM>(up, 9, lane4, attr.y); st_plane<STREAM>(up, 10, lane4, attr.z); st_plane<STREAM>(up, 11, lane4, attr.w);
                }
            }
            const bool park = parking != 0;
            unsigned long long park_mask = __ballot(park);
            const uint32_t n_park = (uint32_t)__popcll(park_mask);
            if (n_park != 0u && pending + n_park > (uint32_t)kDfRing) {
                // no room in the ring: these lanes take the long path here, as the interpreter's kernel does
                if (park) df_long<FMT>(pos, vel, fx, fy, d0, d1, parking == 2, a.sys, a.dt_s, A.df, A.sdf, sdf_samples);
                park_mask = 0ull;
            } else if (n_park != 0u) {
                if (park) {
                    const uint32_t e = pending + (uint32_t)__popcll(park_mask & lanes_below);
                    parked.px[e] = pos.x; parked.py[e] = pos.y; parked.pz[e] = pos.z; parked.life[e] = pos.w;
                    parked.vx[e] = vel.x; parked.vy[e] = vel.y; parked.vz[e] = vel.z; parked.ct[e] = vel.w;
                    parked.d0[e] = d0; parked.d1[e] = d1; parked.slot[e] = (uint32_t)(first + (int)lane) | ((parking == 2) ? 0x80000000u : 0u);
                }
            }
            if (K == 1 || j == 0) { hp0 = pos; hv0 = vel; pm0 = park_mask; pb0 = pending; }
            else if (K == 2 || j == 1) { hp1 = pos; hv1 = vel; pm1 = park_mask; pb1 = pending; }
            else if (j == 2) { hp2 = pos; hv2 = vel; pm2 = park_mask; pb2 = pending; }
            else { hp3 = pos; hv3 = vel; pm3 = park_mask; pb3 = pending; }
            pending += (uint32_t)__popcll(park_mask);
            if (pending - done >= 64u) { long_pass(done, 64u); done += 64u; }
        }
        // ---- phase L: what is still parked ----
        while (done < pending) {
            const uint32_t count = min(64u, pending - done);
            long_pass(done, count);
            done += count;
        }
        // ---- phase B ----
        if constexpr (SPAWN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the attributes stored for slots spawned in this launch are read back below
#pragma nounroll
        for (int j = 0; j < units; j++) {
            const int seg = seg0 + j;
            const UnitPlanes up = unit_planes(chunk_base, a.stride, seg * 64);
            const float4 attr = mk4(ld_plane<STREAM>(up, 8, lane4), ld_plane<STREAM>(up, 9, lane4), ld_plane<STREAM>(up, 10, lane4), ld_plane<STREAM>(up, 11, lane4));
            float4 pos, vel; unsigned long long pm; uint32_t pb;
            if (K == 1 || j == 0) { pos = hp0; vel = hv0; pm = pm0; pb = pb0; }
            else if (K == 2 || j == 1) { pos = hp1; vel = hv1; pm = pm1; pb = pb1; }
            else if (j == 2) { pos = hp2; vel = hv2; pm = pm2; pb = pb2; }
            else { pos = hp3; vel = hv3; pm = pm3; pb = pb3; }
            if ((pm >> lane) & 1ull) {
                const uint32_t e = pb + (uint32_t)__popcll(pm & lanes_below);
                pos = mk4(parked.px[e], parked.py[e], parked.pz[e], parked.life[e]);
                vel = mk4(parked.vx[e], parked.vy[e], parked.vz[e], parked.ct[e]);
            }
            const int first = seg * 64;
            const int row = first >> a.cs_shift;
            const float fx = (float)(first - (row << a.cs_shift) + (int)lane), fy = (float)row;
            float4 rc = zero, rd = zero;
            if (__ballot(pos.w > 0.0f) != 0ull)
                render_data(fx, fy, pos, vel, attr, a.sys, a.update, a.bezier_codes, a.update_bits, nullptr, 0, 0, rc, rd);
            st_plane<STREAM>(up, 0, lane4, pos.x); st_plane<STREAM>(up, 1, lane4, pos.y); st_plane<STREAM>(up, 2, lane4, pos.z); st_plane<STREAM>(up, 3, lane4, pos.w);
            st_plane<STREAM>(up, 4, lane4, vel.x); st_plane<STREAM>(up, 5, lane4, vel.y); st_plane<STREAM>(up, 6, lane4, vel.z); st_plane<STREAM>(up, 7, lane4, vel.w);
            st_plane<STREAM>(up, 12, lane4, rc.x); st_plane<STREAM>(up, 13, lane4, rc.y); st_plane<STREAM>(up, 14, lane4, rc.z); st_plane<STREAM>(up, 15, lane4, rc.w);
            st_plane<STREAM>(up, 16, lane4, rd.x); st_plane<STREAM>(up, 17, lane4, rd.y); st_plane<STREAM>(up, 18, lane4, rd.z); st_plane<STREAM>(up, 19, lane4, rd.w);
            n_live += (uint32_t)__popcll(__ballot(pos.w > 0.0f));
        }
    }
    if ((__builtin_amdgcn_readfirstlane(g_step_count_sdf_samples) != 0) && (sdf_samples != 0))
        atomicAdd(&g_step_sdf_samples, (unsigned long long)sdf_samples);
    if (a.flags & ILM_STEP_COUNT_LIVE)
        publish_block_count(wave_live, n_live, lane, wave, v < a.total_units, a.first_chunk + (v >> a.upc_shift),
                            (v & ((1 << a.upc_shift) - 1)) / kUnitsPerBlock, (1 << a.upc_shift) / kUnitsPerBlock, a.count_buckets,
                            a.live_counts, a.zero_counts, a.zero_n, a.host_counts, a.count_seq);
}

// LeanStep from a StepLaunch whose launch_step fields are filled; false when the step is not of the lean shape
static bool build_lean_step(const StepLaunch& a, LeanStep& f) {
    const IlmStepDesc& d = a.desc;
    if (kUnitsPerWave != 1) return false;
    if (d.UpdateMode != ILM_UPDATE_POSITIONS && d.UpdateMode != ILM_UPDATE_WITH_DISTANCE_FIELD) return false;   // (the collision update: launch_lean_df_step adds the field)
    if (a.derived.cs_shift < 6 || a.upc_shift < 0) return false;                   // power-of-two chunk size >= 64: no stride padding, a unit lies in one row
    if (a.slots != a.span) return false;
    if (a.derived.noise_may_revive != 0) return false;
    if (a.derived.update_bits & 1u) return false;                                   // life ramp
    if (a.op_mask & ~((1u << ILM_OP_GRAVITY) | (1u << ILM_OP_NOISE) | (1u << ILM_OP_FMA))) return false;
    memset(&f, 0, sizeof(f));
    for (int o = 0; o < d.OpCount; o++) {
        const IlmTransformOp& op = d.Ops[o];
        f.op_type[o] = op.Type;
        f.dop[o] = a.derived.op[o];
        if (op.Type == ILM_OP_GRAVITY) {
            const IlmGravityParams& g = op.u.Gravity;
            if (g.AttractorCount > kLeanMaxAttractors) return false;
            LeanGravity& l = f.op[o].gravity;
            l.count = g.AttractorCount < 0 ? 0 : g.AttractorCount;
            l.max_accel = a.derived.op[o].max_accel;
            l.cat_lo = g.CategoryFilter[0]; l.cat_hi = g.CategoryFilter[1];
            l.types = 0;
            for (int i = 0; i < l.count; i++) {
                l.a[i].x = g.AttractorPositions[i][0]; l.a[i].y = g.AttractorPositions[i][1]; l.a[i].z = g.AttractorPositions[i][2];
                l.a[i].radius = g.AttractorRadiusesAndStrengths[i][0];
                l.a[i].strength = g.AttractorRadiusesAndStrengths[i][1];
                const float type = g.AttractorRadiusesAndStrengths[i][2];
                l.types |= ((type >= 1.5f) ? 2u : ((type >= 0.5f) ? 1u : 0u)) << (2 * i);
            }
        } else if (op.Type == ILM_OP_NOISE) {
            // the per-slot form of evaluateByTypeId stays with the interpreter; a second Noise op has no uniform deltas
            if (!a.derived.op[o].area_none || a.derived.noise.op != o) return false;
            f.op[o].noise = op.u.Noise;
        } else {
            if (!a.derived.op[o].area_none) return false;
            f.op[o].fma = op.u.FMA;
        }
    }
    f.op_count = d.OpCount;
    f.noise_op = a.derived.noise.op;
    f.noise = a.derived.noise;
    f.spawn_count = 0;
    for (int s = 0; s < ILM_MAX_SPAWNS; s++) { f.spawn_chunk[s] = -1; f.spawn_unit_lo[s] = 1; f.spawn_unit_hi[s] = 0; }
    for (int s = 0; s < d.SpawnCount; s++) {
        const IlmSpawnRecord& r = d.Spawns[s];
        if (r.Kind != ILM_SPAWN_INLINE) return false;
        f.spawns[s] = r;
        f.spawn_chunk[s] = r.ChunkIndex;
        const float first = r.Params.ChunkSizeAndIndices[1], last = r.Params.ChunkSizeAndIndices[2];
        if (last >= first && last >= 0.0f) {      // (a NaN bound compares false: no lane can pass the per-lane test either)
            const double lo = first < 0.0f ? 0.0 : std::floor((double)first), hi = std::floor((double)last);
            const double max_unit = (double)(a.units_per_chunk - 1);
            f.spawn_unit_lo[s] = (int32_t)std::fmin(lo / 64.0, max_unit + 1.0);
            f.spawn_unit_hi[s] = (int32_t)std::fmin(hi / 64.0, max_unit);
        }
        f.spawn_count = s + 1;
    }
    if (d.SpawnCount == 0 && a.derived.noise.op >= 0 && a.derived.noise.classes == kNoiseBigClasses)
        memcpy(f.noise_big, &d.Spawns[0], sizeof(f.noise_big));       // fill_noise_fast put the 5 x 5 tables there
    f.chunk_bases = a.chunk_bases;
    f.live_counts = a.live_counts; f.zero_counts = a.zero_counts; f.zero_n = a.zero_n; f.host_counts = a.host_counts; f.count_seq = a.count_seq; f.count_buckets = a.count_buckets;
    f.stride = a.stride;
    f.upc_shift = a.upc_shift; f.unit_rotate = a.unit_rotate; f.total_padded = a.total_padded; f.total_units = a.unit_end - a.unit_begin;
    f.first_chunk = a.first_chunk; f.cs_shift = a.derived.cs_shift; f.flags = d.Flags;
    f.partial_count = a.partial_count;
    for (int k = 0; k < kMaxPartialChunks; k++) {
        f.partial_chunk[k] = (k < a.partial_count) ? a.partial_chunk[k] : -1;
        f.partial_units[k] = (k < a.partial_count) ? a.partial_units[k] : 0;
    }
    f.dt_s = a.derived.dt_s; f.bezier_codes = a.derived.bezier_codes; f.update_bits = a.derived.update_bits;
    f.sys = d.System; f.update = d.Update;
    f.rnd = a.rnd; f.rw = a.rw; f.rh = a.rh; f.inv_rw = a.derived.inv_rw; f.inv_rh = a.derived.inv_rh;
    return true;
}

static hipError_t launch_lean_step(const LeanStep& f, bool spawning, bool streaming, hipStream_t stream) {
    const int units_per_block = kStepThreads / 64;
    const dim3 grid((unsigned)((f.total_units + units_per_block - 1) / units_per_block), 1, 1), block(kStepThreads, 1, 1);
    // experiment switch: resident waves per SIMD capped through dynamic LDS (blocks per CU); 0 = no cap
    static const int occ = [] { const char* e = getenv("ILM_STEP_OCC"); return e ? atoi(e) : 0; }();
    const unsigned lds = (occ > 0) ? (unsigned)(160 * 1024 / occ - 64) : 0u;
    if (spawning) {
        if (streaming) hipLaunchKernelGGL((step_lean_kernel<true, true>), grid, block, lds, stream, f);
        else hipLaunchKernelGGL((step_lean_kernel<true, false>), grid, block, lds, stream, f);
    } else {
        if (streaming) hipLaunchKernelGGL((step_lean_kernel<false, true>), grid, block, lds, stream, f);
        else hipLaunchKernelGGL((step_lean_kernel<false, false>), grid, block, lds, stream, f);
    }
    return hipGetLastError();
}

// The collision kernels' first template argument: the field's format, plus kFieldSlice0 when the uniforms put every lookup in virtual
// slice 0 with a z weight of 0 (hlsl_math.hpp sample_distance_field<.., SLICE0>) -- what the reference's particle path binds.
constexpr int kFieldSlice0 = 2;
constexpr int kFieldCells0 = 4;      // ... plus the slice-0 cells (SdfView::cells0): one load per lookup (lean collision kernel only)
static_assert((ILM_SDF_UNORM16 | ILM_SDF_FP16) == 1, "the format is bit 0 of the collision kernels' first template argument");
// The slice-0 sampler returns the bilinear fetch of channel r where the general one forms lerp(lo, hi, 0) = fma(0, hi - lo, lo): equal
// for FINITE texels only, so it serves UNORM16 fields (every code is finite); an FP16 atlas uploaded through ilm_sdf_upload may hold
// inf / NaN (hi = inf gives NaN in the general form and in the oracle) and keeps the general sampler.  The uniforms whose fma(0, ., .)
// terms the slice-0 form drops (Packed1.x, .z, TextureSliceAndTexelSize.xy) must be finite for the same reason.
static bool field_is_slice0(const IlmDistanceFieldUniforms& df, int format) {
    static const int enabled = [] { const char* e = getenv("ILM_DF_SLICE0"); return e ? atoi(e) : 1; }();
    return enabled && (format == ILM_SDF_UNORM16) && (df.Packed1.y == 0.0f) && std::isfinite(df.Packed1.x) && std::isfinite(df.Packed1.z) &&
           std::isfinite(df.TextureSliceAndTexelSize.x) && std::isfinite(df.TextureSliceAndTexelSize.y);
}
// SdfView::cells0 from the atlas: cell (x0, yr) = channel r of the taps (x0, y0), (x1, y0), (x0, y1), (x1, y1) of a bilinear fetch whose
// upper-left tap is column x0 (already wrapped) of row yi = yr - 1 -- x1 = x0 + 1 with U WRAP, y0 = clamp(yi), y1 = the next row exactly
// when 0 <= yi < height - 1: the integer bookkeeping of sample_distance_field, done once per texel instead of once per lookup.
__global__ __launch_bounds__(256) void build_slice0_cells_kernel(const uint2* __restrict__ texels, int width, int height, uint2* __restrict__ cells) {
    const int x0 = (int)blockIdx.x * 256 + (int)threadIdx.x, yr = (int)blockIdx.y;
    if (x0 >= width) return;
    const int yi = yr - 1;
    const int y0 = min(max(yi, 0), height - 1);
    const int y1 = ((uint32_t)yi < (uint32_t)(height - 1)) ? y0 + 1 : y0;
    const int x1 = (x0 + 1 == width) ? 0 : x0 + 1;
    const uint32_t r00 = texels[(size_t)y0 * width + x0].x & 0xFFFFu, r10 = texels[(size_t)y0 * width + x1].x & 0xFFFFu;
    const uint32_t r01 = texels[(size_t)y1 * width + x0].x & 0xFFFFu, r11 = texels[(size_t)y1 * width + x1].x & 0xFFFFu;
    cells[(size_t)yr * width + x0] = make_uint2(r00 | (r10 << 16), r01 | (r11 << 16));
}
hipError_t launch_build_slice0_cells(const uint2* texels, int width, int height, void* cells, hipStream_t stream) {
    hipLaunchKernelGGL(build_slice0_cells_kernel, dim3((unsigned)((width + 255) / 256), (unsigned)(height + 1)), dim3(256), 0, stream, texels, width, height, static_cast<uint2*>(cells));
    return hipGetLastError();
}
static bool step_interpreter_forced();
bool step_wants_slice0_cells(const IlmStepDesc& d, int format) {
    const char* e = getenv("ILM_DF_CELLS0");            // A/B switch, read per step: 0 = the four-tap form
    if (e && atoi(e) == 0) return false;
    const char* lean = getenv("ILM_DF_LEAN");
    if ((lean && atoi(lean) == 0) || step_interpreter_forced()) return false;
    return d.UpdateMode == ILM_UPDATE_WITH_DISTANCE_FIELD && format == ILM_SDF_UNORM16 && field_is_slice0(d.DistanceField, format);
}

// The lean collision step.  K = units per wave: the more units a wave walks, the fuller its long passes run -- and the longer it lives, so
// a launch of few waves (cfg2: 16 384 units) takes the smaller K (tools/ab_collision.sh; ILM_DF_UNITS overrides).
template <int FMT, bool SPAWN>
static hipError_t launch_lean_df_variant(const LeanStepDf& f, int k, bool streaming, const dim3& grid, hipStream_t stream) {
    const dim3 block(kStepThreads, 1, 1);
    if (streaming) hipLaunchKernelGGL((step_lean_df_kernel<FMT, SPAWN, true, 4>), grid, block, 0, stream, f);
    else if (k == 4) hipLaunchKernelGGL((step_lean_df_kernel<FMT, SPAWN, false, 4>), grid, block, 0, stream, f);
    else if (k == 2) hipLaunchKernelGGL((step_lean_df_kernel<FMT, SPAWN, false, 2>), grid, block, 0, stream, f);
    else hipLaunchKernelGGL((step_lean_df_kernel<FMT, SPAWN, false, 1>), grid, block, 0, stream, f);
    return hipGetLastError();
}
static hipError_t launch_lean_df_step(LeanStepDf& f, const StepLaunch& a, bool spawning, hipStream_t stream) {
    const char* forced_env = getenv("ILM_DF_UNITS");                  // (read per launch: an A/B and test switch)
    const int forced_k = forced_env ? atoi(forced_env) : 0;
    const bool streaming = a.streaming != 0;
    int k = (forced_k == 1 || forced_k == 2 || forced_k == 4) ? forced_k : ((f.base.total_units >= 65536) ? 4 : 2);
    if (streaming) k = 4;
    const int upb = (kStepThreads / 64) * k;                            // units per block
    while (k > 1 && (a.units_per_chunk % ((kStepThreads / 64) * k)) != 0) k >>= 1;       // (units per chunk is a power of two >= 64: never taken)
    // the grid's bookkeeping for blocks of upb units (launch_step laid it out for blocks of kStepThreads / 64)
    f.base.total_padded = (f.base.total_units + upb - 1) / upb * upb;
    f.base.unit_rotate = f.base.unit_rotate / upb * upb;
    int buckets = 1;
    while (buckets * 2 <= kCountLines - 1 && (a.units_per_chunk / upb) % (buckets * 2) == 0) buckets *= 2;
    f.base.count_buckets = buckets;
    const dim3 grid((unsigned)(f.base.total_padded / upb), 1, 1);
    int fmt = (int)a.sdf.format | (field_is_slice0(a.desc.DistanceField, (int)a.sdf.format) ? kFieldSlice0 : 0);
    if (fmt == (ILM_SDF_UNORM16 | kFieldSlice0) && a.sdf.cells0 != nullptr) fmt |= kFieldCells0;
    switch (fmt) {
        case ILM_SDF_UNORM16 | kFieldSlice0 | kFieldCells0:
            return spawning ? launch_lean_df_variant<ILM_SDF_UNORM16 | kFieldSlice0 | kFieldCells0, true>(f, k, streaming, grid, stream)
                            : launch_lean_df_variant<ILM_SDF_UNORM16 | kFieldSlice0 | kFieldCells0, false>(f, k, streaming, grid, stream);
        case ILM_SDF_FP16: return spawning ? launch_lean_df_variant<ILM_SDF_FP16, true>(f, k, streaming, grid, stream) : launch_lean_df_variant<ILM_SDF_FP16, false>(f, k, streaming, grid, stream);
        case ILM_SDF_UNORM16: return spawning ? launch_lean_df_variant<ILM_SDF_UNORM16, true>(f, k, streaming, grid, stream) : launch_lean_df_variant<ILM_SDF_UNORM16, false>(f, k, streaming, grid, stream);
        case ILM_SDF_FP16 | kFieldSlice0: return spawning ? launch_lean_df_variant<ILM_SDF_FP16 | kFieldSlice0, true>(f, k, streaming, grid, stream)
                                                          : launch_lean_df_variant<ILM_SDF_FP16 | kFieldSlice0, false>(f, k, streaming, grid, stream);
        default: return spawning ? launch_lean_df_variant<ILM_SDF_UNORM16 | kFieldSlice0, true>(f, k, streaming, grid, stream)
                                 : launch_lean_df_variant<ILM_SDF_UNORM16 | kFieldSlice0, false>(f, k, streaming, grid, stream);
    }
}

// The extended variant carries the rarely used techniques (MatrixMultiply, SpatialNoise, the position-buffer and feedback spawners) so
// that the common variants do not pay their registers; it is always the spawning superset.
static bool needs_extended_variant(const StepLaunch& a) {
    if (a.op_mask & ((1u << ILM_OP_MATRIX_MULTIPLY) | (1u << ILM_OP_SPATIAL_NOISE))) return true;
    for (int s = 0; s < a.desc.SpawnCount; s++)
        if (a.desc.Spawns[s].Kind != ILM_SPAWN_INLINE) return true;
    return false;
}

// Collision variants: six waves per SIMD (79 VGPRs, no scratch) 37.5-38.3 us on bench.py's collision row against 38.6-39.4 for the
// allocator's own 86 VGPRs / five waves; seven (72 VGPRs, 24 B of scratch) 44.4, eight (64 VGPRs, 76 B) 55.2 (tools/ab_collision.sh)
#ifndef ILM_DF_MINW
#define ILM_DF_MINW 6
#endif
template <bool SPAWN>
static hipError_t launch_step_variant(const StepLaunch& a, hipStream_t stream) {
    const int units = a.unit_end - a.unit_begin;
    if (units <= 0) return hipSuccess;
    if (needs_extended_variant(a)) {
        const int upb = (kStepThreads / 64) * kUnitsPerWave;
        const dim3 g((unsigned)((units + upb - 1) / upb), 1, 1), b(kStepThreads, 1, 1);
        if (a.desc.UpdateMode == ILM_UPDATE_WITH_DISTANCE_FIELD) {
            switch ((int)a.sdf.format | (field_is_slice0(a.desc.DistanceField, (int)a.sdf.format) ? kFieldSlice0 : 0)) {
                case ILM_SDF_FP16: hipLaunchKernelGGL((step_kernel<ILM_SDF_FP16, true, true, 1, true>), g, b, 0, stream, a); break;
                case ILM_SDF_UNORM16: hipLaunchKernelGGL((step_kernel<ILM_SDF_UNORM16, true, true, 1, true>), g, b, 0, stream, a); break;
                case ILM_SDF_FP16 | kFieldSlice0: hipLaunchKernelGGL((step_kernel<ILM_SDF_FP16 | kFieldSlice0, true, true, 1, true>), g, b, 0, stream, a); break;
                default: hipLaunchKernelGGL((step_kernel<ILM_SDF_UNORM16 | kFieldSlice0, true, true, 1, true>), g, b, 0, stream, a); break;
            }
        } else {
            hipLaunchKernelGGL((step_kernel<ILM_SDF_UNORM16, false, true, 1, true>), g, b, 0, stream, a);
        }
        return hipGetLastError();
    }
    static int minw = -1;
    if (minw < 0) {
        const char* e = getenv("ILM_STEP_MINWAVES");
        minw = e ? atoi(e) : kDefaultStepMinWaves;
    }
    const int units_per_block = (kStepThreads / 64) * kUnitsPerWave;
    const dim3 grid((unsigned)((units + units_per_block - 1) / units_per_block), 1, 1), block(kStepThreads, 1, 1);
    if (a.desc.UpdateMode == ILM_UPDATE_WITH_DISTANCE_FIELD) {
        // waves per SIMD requested for the collision variants (ILM_DF_MINW; measured in docs/experiments.md 3.1)
        switch ((int)a.sdf.format | (field_is_slice0(a.desc.DistanceField, (int)a.sdf.format) ? kFieldSlice0 : 0)) {
            case ILM_SDF_FP16: hipLaunchKernelGGL((step_kernel<ILM_SDF_FP16, true, SPAWN, ILM_DF_MINW>), grid, block, 0, stream, a); break;
            case ILM_SDF_UNORM16: hipLaunchKernelGGL((step_kernel<ILM_SDF_UNORM16, true, SPAWN, ILM_DF_MINW>), grid, block, 0, stream, a); break;
            case ILM_SDF_FP16 | kFieldSlice0: hipLaunchKernelGGL((step_kernel<ILM_SDF_FP16 | kFieldSlice0, true, SPAWN, ILM_DF_MINW>), grid, block, 0, stream, a); break;
            default: hipLaunchKernelGGL((step_kernel<ILM_SDF_UNORM16 | kFieldSlice0, true, SPAWN, ILM_DF_MINW>), grid, block, 0, stream, a); break;
        }
    } else if (a.streaming) {
        hipLaunchKernelGGL((step_kernel<ILM_SDF_UNORM16, false, SPAWN, 1, false, true>), grid, block, 0, stream, a);
    } else if (SPAWN) {
        // (bounding this variant to 7 / 8 waves per SIMD -- 72 / 64 VGPRs with 8 / 24 bytes of scratch in the cold spawn path --
        // measured 3.5 % / 14 % SLOWER on cfg2: the step is VALU-issue-bound, not occupancy-bound; DESIGN.md "experiments")
        hipLaunchKernelGGL((step_kernel<ILM_SDF_UNORM16, false, SPAWN, 1>), grid, block, 0, stream, a);
    } else {
        switch (minw) {
            case 8: hipLaunchKernelGGL((step_kernel<ILM_SDF_UNORM16, false, false, 8>), grid, block, 0, stream, a); break;
            case 7: hipLaunchKernelGGL((step_kernel<ILM_SDF_UNORM16, false, false, 7>), grid, block, 0, stream, a); break;
            case 6: hipLaunchKernelGGL((step_kernel<ILM_SDF_UNORM16, false, false, 6>), grid, block, 0, stream, a); break;
            default: hipLaunchKernelGGL((step_kernel<ILM_SDF_UNORM16, false, false, 1>), grid, block, 0, stream, a); break;
        }
    }
    return hipGetLastError();
}

static std::atomic<int> g_step_interpreter{-1};      // (atomic: a process-wide switch read by every stepping thread)  -1: not decided yet (ILM_STEP_LEAN), 0: lean kernel where it applies, 1: interpreter always
static bool step_interpreter_forced() {
    if (g_step_interpreter < 0) {
        const char* e = getenv("ILM_STEP_LEAN");
        g_step_interpreter = (e && atoi(e) == 0) ? 1 : 0;
    }
    return g_step_interpreter != 0;
}
int set_step_interpreter(int on) {
    const int before = step_interpreter_forced() ? 1 : 0;
    g_step_interpreter = on ? 1 : 0;
    return before;
}

// One launch per ParticleSystem.Update: the unit range covers every chunk of the step; when spawn records are
// present the SPAWN variant runs (for every unit) and the grid is rotated to start at the first spawn range.
hipError_t launch_step(StepLaunch& a, hipStream_t stream) {
    a.units_per_chunk = a.span / 64;
    a.upc_shift = -1;
    for (int b = 0; b < 31; b++)
        if ((1 << b) == a.units_per_chunk) a.upc_shift = b;
    a.unit_begin = 0;
    a.unit_end = a.chunk_count * a.units_per_chunk;
    const int waves_per_block = (kStepThreads / 64) * kUnitsPerWave;   // units per block
    a.total_padded = (a.unit_end + waves_per_block - 1) / waves_per_block * waves_per_block;
    a.unit_rotate = 0;
    a.count_buckets = 1;
    while (a.count_buckets * 2 <= kCountLines - 1 && (a.units_per_chunk / waves_per_block) % (a.count_buckets * 2) == 0) a.count_buckets *= 2;
    bool spawning = false;
    for (int s = 0; s < a.desc.SpawnCount; s++) {
        const IlmSpawnRecord& r = a.desc.Spawns[s];
        if (r.ChunkIndex < a.first_chunk || r.ChunkIndex >= a.first_chunk + a.chunk_count)
            continue;
        if (r.Params.ChunkSizeAndIndices[2] < r.Params.ChunkSizeAndIndices[1])
            continue;
        if (!spawning) {
            const int unit = (r.ChunkIndex - a.first_chunk) * a.units_per_chunk + (int)r.Params.ChunkSizeAndIndices[1] / 64;
            a.unit_rotate = unit / waves_per_block * waves_per_block;
        }
        spawning = true;
    }
    if (!step_interpreter_forced() && a.unit_end > a.unit_begin) {
        if (a.desc.UpdateMode == ILM_UPDATE_WITH_DISTANCE_FIELD) {
            const char* lean_env = getenv("ILM_DF_LEAN");                 // A/B switch, read per launch: 0 = the interpreter
            const bool lean_df = !(lean_env && atoi(lean_env) == 0);
            LeanStepDf f;
            if (lean_df && build_lean_step(a, f.base)) {
                f.df = a.desc.DistanceField;
                f.sdf = a.sdf;
                return launch_lean_df_step(f, a, spawning, stream);
            }
        } else {
            LeanStep f;
            if (build_lean_step(a, f))
                return launch_lean_step(f, spawning, a.streaming != 0, stream);
        }
    }
    return spawning ? launch_step_variant<true>(a, stream) : launch_step_variant<false>(a, stream);
}

// ---------------------------------------------------------------------------------------------
// layout conversion: the reference API speaks AoS float4 (Spawn initializers, AutoReadback)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void aos_to_soa_kernel(const float4* __restrict__ src, float* __restrict__ plane0, int64_t stride,
                                                          int first_slot, int count) {
    const int i = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (i >= count) return;
    const float4 v = src[i];
    float* p = plane0 + first_slot + i;
    p[0] = v.x; p[stride] = v.y; p[2 * stride] = v.z; p[3 * stride] = v.w;
}
__global__ __launch_bounds__(256) void soa_to_aos_kernel(const float* __restrict__ plane0, int64_t stride, float4* __restrict__ dst,
                                                          int first_slot, int count) {
    const int i = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (i >= count) return;
    const float* p = plane0 + first_slot + i;
    dst[i] = mk4(p[0], p[stride], p[2 * stride], p[3 * stride]);
}
hipError_t launch_aos_to_soa(const float4* src, float* plane0, int64_t stride, int32_t first_slot, int32_t count, hipStream_t stream) {
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(aos_to_soa_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, src, plane0, stride, first_slot, count);
    return hipGetLastError();
}
hipError_t launch_soa_to_aos(const float* plane0, int64_t stride, float4* dst, int32_t first_slot, int32_t count, hipStream_t stream) {
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(soa_to_aos_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, plane0, stride, dst, first_slot, count);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// liveness: standalone count + ordered live-slot compaction
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void count_live_kernel(float* const* __restrict__ bases, int64_t stride, int slots, uint32_t* __restrict__ counts) {
    __shared__ uint32_t wave_live[4];
    const int chunk = (int)blockIdx.y;
    const float* life = bases[chunk] + 3 * stride;
    const int i0 = ((int)blockIdx.x * 256 + (int)threadIdx.x) * 4;
    const float4 l = *reinterpret_cast<const float4*>(life + i0);
    // only the chunk's ChunkSize^2 slots are particles: the padding up to the stride is not counted
    const uint32_t n = (uint32_t)__popcll(__ballot((i0 < slots) && (l.x > 0.0f))) + (uint32_t)__popcll(__ballot((i0 + 1 < slots) && (l.y > 0.0f))) +
                       (uint32_t)__popcll(__ballot((i0 + 2 < slots) && (l.z > 0.0f))) + (uint32_t)__popcll(__ballot((i0 + 3 < slots) && (l.w > 0.0f)));
    if ((threadIdx.x & 63) == 0) wave_live[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = wave_live[0] + wave_live[1] + wave_live[2] + wave_live[3];
        if (total != 0) atomicAdd(&counts[chunk * kCountStride], total);
    }
}
hipError_t launch_count_live(float* const* chunk_bases, int64_t stride, int32_t span, int32_t slots, int32_t chunk_count, uint32_t* counts, hipStream_t stream) {
    if (chunk_count <= 0) return hipSuccess;
    hipLaunchKernelGGL(count_live_kernel, dim3((unsigned)(span / 1024), (unsigned)chunk_count), dim3(256), 0, stream, chunk_bases, stride, slots, counts);
    return hipGetLastError();
}

// Single workgroup of 1024 threads walks the chunk in 1024-slot tiles keeping a
// running base, so the output is in ascending slot order (deterministic):
// per-wave ballot -> popcount prefix inside the wave, LDS scan across the 16 waves.
// Two launches over 1024-slot blocks (r04; one workgroup used to walk the whole chunk, 64 rounds of three barriers for a 256^2 chunk:
// 97 us): the blocks' live counts, then every block adds the counts in front of it and writes its slots at that base -- ballot +
// popcount prefix inside the block, so the list is in slot order.
__global__ __launch_bounds__(1024) void live_slots_count_kernel(const float* __restrict__ life, int slots, uint32_t* __restrict__ block_counts) {
    __shared__ uint32_t wave_counts[16];
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const int i = (int)blockIdx.x * 1024 + (int)threadIdx.x;
    const bool alive = (i < slots) && (life[i] > 0.0f);
    const unsigned long long mask = __ballot(alive);
    if (lane == 0) wave_counts[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < 16; w++) t += wave_counts[w];
        block_counts[blockIdx.x] = t;
    }
}
__global__ __launch_bounds__(1024) void live_slots_emit_kernel(const float* __restrict__ life, int slots, const uint32_t* __restrict__ block_counts,
                                                                uint32_t* __restrict__ out, uint32_t capacity, uint32_t* __restrict__ out_count) {
    __shared__ uint32_t wave_counts[16];
    __shared__ uint32_t partial[16];
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    // the live slots in the blocks in front of this one
    uint32_t before = 0;
    for (int b = (int)threadIdx.x; b < (int)blockIdx.x; b += 1024) before += block_counts[b];
    for (int off = 32; off > 0; off >>= 1) before += __shfl_down(before, off);
    if (lane == 0) partial[wave] = before;
    const int i = (int)blockIdx.x * 1024 + (int)threadIdx.x;
    const bool alive = (i < slots) && (life[i] > 0.0f);
    const unsigned long long mask = __ballot(alive);
    const uint32_t prefix = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) wave_counts[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < 16; w++) base += partial[w];
    uint32_t wave_base = base;
    for (int w = 0; w < wave; w++) wave_base += wave_counts[w];
    if (alive) {
        const uint32_t dst = wave_base + prefix;
        if (dst < capacity) out[dst] = (uint32_t)i;
    }
    if ((blockIdx.x == gridDim.x - 1) && (threadIdx.x == 0)) {
        uint32_t t = base;
        for (int w = 0; w < 16; w++) t += wave_counts[w];
        *out_count = t;
    }
}
hipError_t launch_live_slots(const float* life, int32_t slots, uint32_t* out_slots, uint32_t capacity, uint32_t* out_count, uint32_t* block_counts, hipStream_t stream) {
    const unsigned blocks = (unsigned)((slots + 1023) / 1024);
    if (blocks == 0) return hipMemsetAsync(out_count, 0, sizeof(uint32_t), stream);
    hipLaunchKernelGGL(live_slots_count_kernel, dim3(blocks), dim3(1024), 0, stream, life, slots, block_counts);
    hipLaunchKernelGGL(live_slots_emit_kernel, dim3(blocks), dim3(1024), 0, stream, life, slots, block_counts, out_slots, capacity, out_count);
    return hipGetLastError();
}

// ilm_debug_step_sdf_samples: switch the count on / off and fetch-and-clear it (the caller has synchronised the device)
hipError_t step_sdf_sample_counter(int enable, unsigned long long* out) {
    unsigned long long value = 0, zero = 0;
    hipError_t e = hipMemcpyFromSymbol(&value, HIP_SYMBOL(g_step_sdf_samples), sizeof(value));
    if (e != hipSuccess) return e;
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_step_sdf_samples), &zero, sizeof(zero));
    if (e != hipSuccess) return e;
    const int flag = enable ? 1 : 0;
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_step_count_sdf_samples), &flag, sizeof(flag));
    if (out) *out = value;
    return e;
}

}  // namespace ilm
