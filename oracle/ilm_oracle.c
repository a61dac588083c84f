/*
 * ilm_oracle.c -- CPU restatement of the reference HLSL (see ilm_oracle.h).
 * TEST INFRASTRUCTURE ONLY -- never linked into the product.  PARITY UNPINNED
 * (no executable reference, no reference golden vectors).
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -fopenmp -fPIC -shared
 */
#include "ilm_oracle.h"

#define _USE_MATH_DEFINES
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------------------
 * HLSL scalar / vector semantics
 * ------------------------------------------------------------------------- */
typedef struct { float x, y, z; } f3;
typedef IlmFloat4 f4;

#include "ilm_oracle_constants.h"

/* value of the reference constant `key` as this restatement uses it (ilm_oracle_constants.h); 0 when unknown */
int32_t orc_reference_constant(const char* key, double* out_value) {
    for (size_t i = 0; i < sizeof(orc_reference_constants) / sizeof(orc_reference_constants[0]); i++)
        if (strcmp(orc_reference_constants[i].key, key) == 0) { *out_value = orc_reference_constants[i].value; return 1; }
    return 0;
}
int32_t orc_reference_constant_count(void) { return (int32_t)(sizeof(orc_reference_constants) / sizeof(orc_reference_constants[0])); }
const char* orc_reference_constant_key(int32_t index) { return orc_reference_constants[index].key; }

static inline float h_sat(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
static inline float h_lerp(float a, float b, float t) { return a + (b - a) * t; }
static inline float h_sign(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
static inline float h_clamp(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

static inline f3 v3(float x, float y, float z) { f3 r = { x, y, z }; return r; }
static inline f4 v4(float x, float y, float z, float w) { f4 r = { x, y, z, w }; return r; }
static inline f3 v3add(f3 a, f3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 v3sub(f3 a, f3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 v3mul(f3 a, f3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline f3 v3scale(f3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline float v3dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float v3len(f3 a) { return sqrtf(v3dot(a, a)); }
static inline f3 v3norm(f3 a) { float l = v3len(a); return v3(a.x / l, a.y / l, a.z / l); }
static inline f3 v3cross(f3 a, f3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline f3 xyz(f4 a) { return v3(a.x, a.y, a.z); }
static inline f4 v4add(f4 a, f4 b) { return v4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline f4 v4sub(f4 a, f4 b) { return v4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
static inline f4 v4mul(f4 a, f4 b) { return v4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
static inline f4 v4scale(f4 a, float s) { return v4(a.x * s, a.y * s, a.z * s, a.w * s); }
static inline f4 v4lerp(f4 a, f4 b, float t) {
    return v4(h_lerp(a.x, b.x, t), h_lerp(a.y, b.y, t), h_lerp(a.z, b.z, t), h_lerp(a.w, b.w, t));
}

/* mul(float4(v,1), M) with XNA row-major M (SpawnerCommon.fxh:166,179) */
static inline f4 mul_point(f3 v, const IlmMatrix* M) {
    const float* m = M->m;
    f4 r;
    r.x = v.x * m[0] + v.y * m[4] + v.z * m[8]  + m[12];
    r.y = v.x * m[1] + v.y * m[5] + v.z * m[9]  + m[13];
    r.z = v.x * m[2] + v.y * m[6] + v.z * m[10] + m[14];
    r.w = v.x * m[3] + v.y * m[7] + v.z * m[11] + m[15];
    return r;
}

static inline int wrap_index(float t, int size) {
    /* D3D WRAP addressing of a POINT/LINEAR tap index */
    int i = (int)t;
    i %= size;
    if (i < 0) i += size;
    return i;
}

/* IEEE half -> float */
static inline float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else {
            exp = 127 - 15 + 1;
            while ((man & 0x400u) == 0) { man <<= 1; exp--; }
            man &= 0x3FFu;
            bits = sign | (exp << 23) | (man << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7F800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

/* float -> IEEE half, round to nearest even, any finite / infinite / NaN input (the lightmap's HalfVector4 blend emulation) */
static inline uint16_t float_to_half_any(float f) {
    union { float f; uint32_t u; } v = { f };
    const uint32_t sign = (v.u >> 16) & 0x8000u;
    uint32_t x = v.u & 0x7FFFFFFFu;
    if (x >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | ((x > 0x7F800000u) ? 0x0200u : 0u));      /* inf / NaN */
    if (x >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);                                              /* rounds to 65536 or more: inf */
    if (x < 0x38800000u) {                                                                                 /* below 2^-14: subnormal half (or zero) */
        union { float f; uint32_t u; } a = { 0 };
        a.u = x;
        const float r = nearbyintf(a.f * 16777216.0f);                                                     /* |f| * 2^24, ties to even */
        return (uint16_t)(sign | (uint32_t)r);
    }
    const uint32_t mant = x & 0x007FFFFFu, exp = (x >> 23) - 112u;
    uint32_t half = (exp << 10) | (mant >> 13);
    const uint32_t rem = mant & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (half & 1u))) half++;                                          /* may carry into the exponent: correct */
    return (uint16_t)(sign | half);
}
static inline float round_through_half(float f) { return half_to_float(float_to_half_any(f)); }

/* Lightmap blend model of the following orc_render_sphere_lights calls.  0: fp32 accumulation over the lights, one rounding when the
 * caller converts the frame (the parity model of this restatement).  1: the reference's render target -- a HalfVector4 surface the
 * ROP blends into light by light (LightingRenderer.cs:476-479, additive blend state :206): clear to half(Ambient), and after every
 * light dst = half(float(dst) + float(half(src))), the pixel shader's output being converted to the target format before the blend. */
static int g_orc_blend_fp16 = 0;
void orc_set_lightmap_blend(int32_t mode) { g_orc_blend_fp16 = (mode != 0); }

/* ---------------------------------------------------------------------------
 * ParticleCommon.fxh accessors (ParticleCommon.fxh:29-92)
 * ------------------------------------------------------------------------- */
static inline float sys_dt_seconds(const IlmParticleSystemUniforms* s) { return s->GlobalSettings.x / VELOCITY_CONSTANT_SCALE; }
static inline float sys_dt(const IlmParticleSystemUniforms* s) { return s->GlobalSettings.x; }
static inline float sys_friction(const IlmParticleSystemUniforms* s) { return s->GlobalSettings.y; }
static inline float sys_max_velocity(const IlmParticleSystemUniforms* s) { return s->GlobalSettings.z; }
static inline float sys_life_decay(const IlmParticleSystemUniforms* s) { return s->GlobalSettings.w; }

/* checkCategoryFilter, ParticleCommon.fxh:187-189 */
static inline int check_category_filter(float type, const float mm[2]) {
    return (type >= mm[0]) && (type <= mm[1]);
}

/* ---------------------------------------------------------------------------
 * RandomCommon.fxh:27-34 -- randomCustom with the POINT/WRAP sampler
 * ------------------------------------------------------------------------- */
static inline f4 random_custom(const f4* rnd, int rw, int rh,
                               float x, float y, const float offset[2], float rate_x, float rate_y) {
    /* RandomnessTexel = (1/807, 1/653), ParticleTransform.cs:255-258 */
    const float texel_x = 1.0f / (float)rw, texel_y = 1.0f / (float)rh;
    float u = ((x * rate_x) + offset[0]) * texel_x;
    float v = ((y * rate_y) + offset[1]) * texel_y;
    int tx = wrap_index(floorf(u * (float)rw), rw);
    int ty = wrap_index(floorf(v * (float)rh), rh);
    return rnd[ty * rw + tx];
}

/* ---------------------------------------------------------------------------
 * DistanceFunctionCommon.fxh -- area distance functions
 * ------------------------------------------------------------------------- */
/* qmul, DistanceFunctionCommon.fxh:16-21 */
static inline f4 qmul(f4 q1, f4 q2) {
    f3 a = v3scale(xyz(q2), q1.w);
    f3 b = v3scale(xyz(q1), q2.w);
    f3 c = v3cross(xyz(q1), xyz(q2));
    f3 s = v3add(v3add(a, b), c);
    return v4(s.x, s.y, s.z, q1.w * q2.w - v3dot(xyz(q1), xyz(q2)));
}
/* rotateLocalPosition, DistanceFunctionCommon.fxh:24-27.  The callers pass the
 * scalar AreaRotation, which HLSL promotes to float4(r,r,r,r) (FMA.fx:11,16-18). */
static inline f3 rotate_local_position(f3 p, f4 rotation) {
    f4 r_c = v4(rotation.x * -1.0f, rotation.y * -1.0f, rotation.z * -1.0f, rotation.w * 1.0f);
    return xyz(qmul(rotation, qmul(v4(p.x, p.y, p.z, 0.0f), r_c)));
}
static inline f3 v3abs(f3 a) { return v3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
static inline f3 v3max0(f3 a) { return v3(fmaxf(a.x, 0.0f), fmaxf(a.y, 0.0f), fmaxf(a.z, 0.0f)); }

/* opElongate, DistanceFunctionCommon.fxh:43-46 */
static inline f4 op_elongate(f3 p, f3 h) {
    f3 q = v3sub(v3abs(p), h);
    f3 m = v3max0(q);
    return v4(h_sign(p.x) * m.x, h_sign(p.y) * m.y, h_sign(p.z) * m.z,
              fminf(fmaxf(q.x, fmaxf(q.y, q.z)), 0.0f));
}
/* evaluateBox, DistanceFunctionCommon.fxh:48-63 */
static float evaluate_box(f3 wp, f3 center, f3 size, f4 rot) {
    f3 p = rotate_local_position(v3sub(wp, center), rot);
    f3 d = v3sub(v3abs(p), size);
    return fminf(fmaxf(d.x, fmaxf(d.y, d.z)), 0.0f) + v3len(v3max0(d));
}
/* evaluateSpheroid, DistanceFunctionCommon.fxh:65-75 */
static float evaluate_spheroid(f3 wp, f3 center, f3 size, f4 rot) {
    f3 p = rotate_local_position(v3sub(wp, center), rot);
    float min_size = fminf(size.x, fminf(size.y, size.z));
    f3 elong = v3(size.x - min_size, size.y - min_size, size.z - min_size);
    f4 w = op_elongate(p, elong);
    return w.w + (v3len(xyz(w)) - min_size);
}
/* sdEllipsoid_improvedV2 + evaluateEllipsoid, DistanceFunctionCommon.fxh:92-109 */
static float evaluate_ellipsoid(f3 wp, f3 center, f3 size, f4 rot) {
    f3 p = rotate_local_position(v3sub(wp, center), rot);
    f3 pr = v3(p.x / size.x, p.y / size.y, p.z / size.z);
    f3 prr = v3(p.x / (size.x * size.x), p.y / (size.y * size.y), p.z / (size.z * size.z));
    float k0 = v3len(pr), k1 = v3len(prr);
    return (k0 < 1.0f) ? (k0 - 1.0f) * fminf(fminf(size.x, size.y), size.z)
                       : k0 * (k0 - 1.0f) / k1;
}
/* sdCappedCylinder + evaluateCylinder, DistanceFunctionCommon.fxh:111-124 */
static float evaluate_cylinder(f3 wp, f3 center, f3 size, f4 rot) {
    f3 p = rotate_local_position(v3sub(wp, center), rot);
    float h = size.z, r = sqrtf(size.x * size.x + size.y * size.y);
    float dx = fabsf(sqrtf(p.x * p.x + p.y * p.y)) - r;
    float dy = fabsf(p.z) - h;
    float mx = fmaxf(dx, 0.0f), my = fmaxf(dy, 0.0f);
    return fminf(fmaxf(dx, dy), 0.0f) + sqrtf(mx * mx + my * my);
}
/* sdOctogonPrism + evaluateOctagon, DistanceFunctionCommon.fxh:141-168 */
static float sd_octogon_prism(f3 p, float r, float h) {
    const float kx = -0.9238795325f, ky = 0.3826834323f, kz = 0.4142135623f;
    p = v3abs(p);
    float d1 = fminf(kx * p.x + ky * p.y, 0.0f);
    p.x -= 2.0f * d1 * kx;
    p.y -= 2.0f * d1 * ky;
    float d2 = fminf(-kx * p.x + ky * p.y, 0.0f);
    p.x -= 2.0f * d2 * -kx;
    p.y -= 2.0f * d2 * ky;
    p.x -= h_clamp(p.x, -kz * r, kz * r);
    p.y -= r;
    float dx = sqrtf(p.x * p.x + p.y * p.y) * h_sign(p.y);
    float dy = p.z - h;
    float mx = fmaxf(dx, 0.0f), my = fmaxf(dy, 0.0f);
    return fminf(fmaxf(dx, dy), 0.0f) + sqrtf(mx * mx + my * my);
}
static float evaluate_octagon(f3 wp, f3 center, f3 size, f4 rot) {
    f3 p = rotate_local_position(v3sub(wp, center), rot);
    float min_size = fminf(size.x, size.y);
    f3 elong = v3(size.x - min_size, size.y - min_size, 0.0f);
    f4 w = op_elongate(p, elong);
    return w.w + sd_octogon_prism(xyz(w), min_size, size.z);
}
/* evaluateByTypeId, DistanceFunctionCommon.fxh:170-187 */
static float evaluate_by_type_id(int type_id, f3 wp, f3 center, f3 size, float rotation) {
    f4 rot = v4(rotation, rotation, rotation, rotation);
    switch (abs(type_id)) {
        case 1: return evaluate_ellipsoid(wp, center, size, rot);
        case 2: return evaluate_box(wp, center, size, rot);
        case 3: return evaluate_cylinder(wp, center, size, rot);
        case 4: return evaluate_spheroid(wp, center, size, rot);
        case 5: return evaluate_octagon(wp, center, size, rot);
        default: return 0.0f;
    }
}
float orc_evaluate_area(int32_t type_id, const float pos[3], const float center[3], const float size[3], float rotation) {
    return evaluate_by_type_id(type_id, v3(pos[0], pos[1], pos[2]), v3(center[0], center[1], center[2]),
                               v3(size[0], size[1], size[2]), rotation);
}

/* computeWeight, FMA.fx:15-20 == Noise.fx:21-26 */
static inline float compute_weight(const IlmAreaParams* a, f3 wp) {
    float distance = evaluate_by_type_id(a->AreaType, wp,
        v3(a->AreaCenter[0], a->AreaCenter[1], a->AreaCenter[2]),
        v3(a->AreaSize[0], a->AreaSize[1], a->AreaSize[2]), a->AreaRotation);
    return (1.0f - h_sat(distance / a->AreaFalloff)) * a->Strength;
}

/* ---------------------------------------------------------------------------
 * Gravity.fx:12-61
 * ------------------------------------------------------------------------- */
static void gravity_slot(f4* pos, f4* vel, const IlmParticleSystemUniforms* sys, const IlmGravityParams* p) {
    f4 new_position = *pos, old_velocity = *vel;
    if ((new_position.w <= 0.0f) || !check_category_filter(old_velocity.w, p->CategoryFilter))
        return; /* newVelocity = oldVelocity */

    f3 acceleration = v3(0, 0, 0);
    for (int i = 0; i < p->AttractorCount; i++) {
        f3 apos = v3(p->AttractorPositions[i][0], p->AttractorPositions[i][1], p->AttractorPositions[i][2]);
        const float* ars = p->AttractorRadiusesAndStrengths[i];
        f3 to_center = v3sub(apos, xyz(new_position));
        float attraction;
        if (ars[2] >= 0.5f) {
            float distance = v3len(to_center);
            attraction = 1.0f - h_sat(distance / ars[0]);
            if (ars[2] >= 1.5f)
                attraction *= attraction;
            attraction = attraction * sys_dt(sys) / VELOCITY_CONSTANT_SCALE;
        } else {
            float distance_squared = v3dot(to_center, to_center) - ars[0];
            distance_squared = fmaxf(distance_squared, 0.001f);
            attraction = 1.0f / distance_squared;
        }
        f3 n = v3norm(to_center);
        f3 new_accel = v3scale(v3scale(n, attraction), ars[1]);
        acceleration = v3add(acceleration, new_accel);
    }

    float maximum_acceleration = p->MaximumAcceleration * sys_dt(sys) / VELOCITY_CONSTANT_SCALE;
    float current_length = v3len(acceleration);
    if (current_length > maximum_acceleration)
        acceleration = v3scale(v3norm(acceleration), maximum_acceleration);

    /* min(getMaximumVelocity(), oldVelocity + acceleration): float4 + float3 truncates to float3 */
    float mv = sys_max_velocity(sys);
    vel->x = fminf(mv, old_velocity.x + acceleration.x);
    vel->y = fminf(mv, old_velocity.y + acceleration.y);
    vel->z = fminf(mv, old_velocity.z + acceleration.z);
    vel->w = old_velocity.w;
}

/* rows [y0, y1) of one chunk; the passes are slot-local, so a band of rows can be taken through the whole
 * pass list on its own (orc_step) without changing any result */
static void gravity_rows(IlmFloat4* pos, IlmFloat4* vel, int32_t chunk_size, int y0, int y1,
                         const IlmParticleSystemUniforms* sys, const IlmGravityParams* p) {
    for (int i = y0 * chunk_size; i < y1 * chunk_size; i++)
        gravity_slot(&pos[i], &vel[i], sys, p);
}

void orc_gravity(IlmFloat4* pos, IlmFloat4* vel, int32_t chunk_size,
                 const IlmParticleSystemUniforms* sys, const IlmGravityParams* p) {
    #pragma omp parallel for schedule(static)
    for (int y = 0; y < chunk_size; y++)
        gravity_rows(pos, vel, chunk_size, y, y + 1, sys, p);
}

/* ---------------------------------------------------------------------------
 * FMA.fx:22-51
 * ------------------------------------------------------------------------- */
static void fma_slot(f4* pos, f4* vel, const IlmParticleSystemUniforms* sys, const IlmFMAParams* p) {
    f4 old_position = *pos, old_velocity = *vel;
    if ((old_position.w <= 0.0f) || !check_category_filter(old_velocity.w, p->Area.CategoryFilter))
        return;
    float weight = compute_weight(&p->Area, xyz(old_position));
    float t = weight * sys_dt(sys) / p->TimeDivisor;
    *pos = v4lerp(old_position, v4add(v4mul(old_position, p->PositionMultiply), p->PositionAdd), t);
    *vel = v4lerp(old_velocity, v4add(v4mul(old_velocity, p->VelocityMultiply), p->VelocityAdd), t);
}

static void fma_rows(IlmFloat4* pos, IlmFloat4* vel, int32_t chunk_size, int y0, int y1,
                     const IlmParticleSystemUniforms* sys, const IlmFMAParams* p) {
    for (int i = y0 * chunk_size; i < y1 * chunk_size; i++)
        fma_slot(&pos[i], &vel[i], sys, p);
}

void orc_fma(IlmFloat4* pos, IlmFloat4* vel, int32_t chunk_size,
             const IlmParticleSystemUniforms* sys, const IlmFMAParams* p) {
    #pragma omp parallel for schedule(static)
    for (int y = 0; y < chunk_size; y++)
        fma_rows(pos, vel, chunk_size, y, y + 1, sys, p);
}

/* ---------------------------------------------------------------------------
 * Noise.fx:28-72
 * ------------------------------------------------------------------------- */
static inline f4 noise_shape(f4 r, f4 offset, f4 minimum, f4 scale) {
    f4 d = v4add(r, offset);
    d = v4(h_sign(d.x) * fmaxf(fabsf(d.x), minimum.x), h_sign(d.y) * fmaxf(fabsf(d.y), minimum.y),
           h_sign(d.z) * fmaxf(fabsf(d.z), minimum.z), h_sign(d.w) * fmaxf(fabsf(d.w), minimum.w));
    return v4mul(d, scale);
}

static void noise_slot(f4* pos, f4* vel, float x, float y, const f4* rnd, int rw, int rh,
                       const IlmParticleSystemUniforms* sys, const IlmNoiseParams* p) {
    f4 old_position = *pos, old_velocity = *vel;
    /* NOTE: no life check here (Noise.fx:40) -- dead slots go through the math */
    if (!check_category_filter(old_velocity.w, p->Area.CategoryFilter))
        return;

    float weight = compute_weight(&p->Area, xyz(old_position));
    float t = weight * sys_dt(sys) / p->TimeDivisor;

    /* rate = RandomnessTexel (quirk, Noise.fx:49-52) */
    const float rate_x = 1.0f / (float)rw, rate_y = 1.0f / (float)rh;
    f4 random_p1 = random_custom(rnd, rw, rh, x, y, p->RandomnessOffset, rate_x, rate_y);
    f4 random_p2 = random_custom(rnd, rw, rh, x, y, p->NextRandomnessOffset, rate_x, rate_y);
    f4 random_v1 = random_custom(rnd, rw, rh, x + 2.0f, y + 1.0f, p->RandomnessOffset, rate_x, rate_y);
    f4 random_v2 = random_custom(rnd, rw, rh, x + 2.0f, y + 1.0f, p->NextRandomnessOffset, rate_x, rate_y);

    f4 random_p = v4lerp(random_p1, random_p2, p->FrequencyLerp);
    f4 random_v = v4lerp(random_v1, random_v2, p->FrequencyLerp);

    f4 position_delta = noise_shape(random_p, p->PositionOffset, p->PositionMinimum, p->PositionScale);
    f4 velocity_delta = noise_shape(random_v, p->VelocityOffset, p->VelocityMinimum, p->VelocityScale);

    *pos = v4lerp(old_position, v4add(old_position, position_delta), t);
    f3 ov = xyz(old_velocity), nv;
    if (p->ReplaceOldVelocity != 0.0f) {
        nv = v3(h_lerp(ov.x, velocity_delta.x, weight), h_lerp(ov.y, velocity_delta.y, weight), h_lerp(ov.z, velocity_delta.z, weight));
    } else {
        nv = v3(h_lerp(ov.x, ov.x + velocity_delta.x, t), h_lerp(ov.y, ov.y + velocity_delta.y, t), h_lerp(ov.z, ov.z + velocity_delta.z, t));
    }
    f3 n = v3norm(ov);
    nv = v3add(nv, v3scale(n, velocity_delta.w));
    *vel = v4(nv.x, nv.y, nv.z, old_velocity.w);
}

static void noise_rows(IlmFloat4* pos, IlmFloat4* vel, int32_t chunk_size, int y0, int y1,
                       const IlmFloat4* rnd, int32_t rw, int32_t rh,
                       const IlmParticleSystemUniforms* sys, const IlmNoiseParams* p) {
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < chunk_size; x++)
            noise_slot(&pos[y * chunk_size + x], &vel[y * chunk_size + x], (float)x, (float)y, rnd, rw, rh, sys, p);
}

void orc_noise(IlmFloat4* pos, IlmFloat4* vel, int32_t chunk_size,
               const IlmFloat4* rnd, int32_t rw, int32_t rh,
               const IlmParticleSystemUniforms* sys, const IlmNoiseParams* p) {
    #pragma omp parallel for schedule(static)
    for (int y = 0; y < chunk_size; y++)
        noise_rows(pos, vel, chunk_size, y, y + 1, rnd, rw, rh, sys, p);
}

/* ---------------------------------------------------------------------------
 * SpawnerCommon.fxh + SpawnParticles.fx:10-30 (technique SpawnParticles)
 * ------------------------------------------------------------------------- */
/* generateRandomNormal3, SpawnerCommon.fxh:47-57 */
static inline f3 generate_random_normal3(float rx, float ry) {
    float phi = rx * H_PI * 2.0f;
    float costheta = (ry - 0.5f) * 2.0f;
    float theta = acosf(costheta);
    return v3(sinf(theta) * cosf(phi), sinf(theta) * sinf(phi), cosf(theta));
}

/* evaluateFormula, SpawnerCommon.fxh:59-104 */
static f4 evaluate_formula(f4 origin, f4 constant, f4 scale, f4 offset, f4 randomness, float type, const float axis_mask[3]) {
    f4 non_circular = v4mul(v4add(randomness, offset), scale);
    f4 type0 = v4add(constant, non_circular);

    unsigned itype = (unsigned)fabsf(floorf(type));
    switch (itype) {
        case 0:
        default:
            return type0;
        case 3:
        case 1: {
            f3 rn = generate_random_normal3(randomness.x, randomness.y);
            rn = v3norm(v3(rn.x * axis_mask[0], rn.y * axis_mask[1], rn.z * axis_mask[2]));
            f3 circular = v3(rn.x * randomness.z * scale.x, rn.y * randomness.z * scale.y, rn.z * randomness.z * scale.z);
            f3 result;
            if (itype == 3) {
                const float sqrt2 = 1.41421356237f;
                f3 edge = v3(fabsf(offset.x), fabsf(offset.y), fabsf(offset.z));
                result = v3(h_clamp(offset.x * rn.x * sqrt2, -edge.x, edge.x),
                            h_clamp(offset.y * rn.y * sqrt2, -edge.y, edge.y),
                            h_clamp(offset.z * rn.z * sqrt2, -edge.z, edge.z));
                result = v3add(result, v3add(xyz(constant), circular));
            } else {
                circular = v3add(circular, v3mul(rn, xyz(offset)));
                result = v3add(xyz(constant), circular);
            }
            return v4(result.x, result.y, result.z, type0.w);
        }
        case 2: {
            f3 distance = v3sub(xyz(constant), xyz(origin));
            float ldistance = v3len(distance);
            if (ldistance < 0.1f)
                return v4(0, 0, 0, constant.w);
            f3 direction = v3(distance.x / ldistance, distance.y / ldistance, distance.z / ldistance);
            f3 random_speed = v3(randomness.x * scale.x * direction.x, randomness.x * scale.y * direction.y, randomness.x * scale.z * direction.z);
            f3 fixed_speed = v3mul(xyz(offset), direction);
            f3 s = v3add(random_speed, fixed_speed);
            return v4(s.x, s.y, s.z, type0.w);
        }
    }
}

static void spawn_slot(f4* pos, f4* vel, f4* attr, float x, float y,
                       const f4* rnd, int rw, int rh, const IlmSpawnParams* p) {
    const float* csi = p->ChunkSizeAndIndices;
    /* Spawn_Stage1, SpawnerCommon.fxh:119-160 */
    float index = x + (y * csi[0]);
    if ((index < csi[1]) || (index > csi[2]))
        return; /* discard: target keeps its contents (RenderTargetUsage.PreserveContents, ParticleSystem.cs:106-111) */

    /* evaluateRandomForIndex, SpawnerCommon.fxh:106-117; random(xy) = randomCustom(xy, RandomnessOffset, 1) */
    f4 random1 = random_custom(rnd, rw, rh, fmodf(index, SP_RANDOM1_X_MODULUS), 0.0f + fmodf(index, SP_RANDOM1_Y_MODULUS), p->RandomnessOffset, 1.0f, 1.0f);
    f4 random2 = random_custom(rnd, rw, rh, fmodf(index, SP_RANDOM2_X_MODULUS), 1.0f + fmodf(index, SP_RANDOM2_Y_MODULUS), p->RandomnessOffset, 1.0f, 1.0f);
    f4 random3 = random_custom(rnd, rw, rh, fmodf(index, SP_RANDOM3_X_MODULUS), 2.0f + fmodf(index, SP_RANDOM3_Y_MODULUS), p->RandomnessOffset, 1.0f, 1.0f);
    if (p->AlignVelocityAndPosition != 0.0f) {
        random2.x = random1.x;
        random2.y = random1.y;
    }

    int index1, index2;
    float position_index_t;
    float relative_index = index - csi[1];
    if (p->PolygonRate > 0.05f) {
        float position_index_f = (relative_index / p->PolygonRate) + csi[3];
        float divisor = p->PositionConstantCount;
        float position_index_i;
        position_index_t = modff(position_index_f, &position_index_i);
        if (p->PolygonLoop != 0.0f) {
            index1 = (int)fmodf(position_index_i, divisor);
            index2 = (int)fmodf(position_index_i + 1.0f, divisor);
        } else {
            index1 = (int)fmodf(position_index_i, divisor);
            index2 = (int)fminf((float)(index1 + 1), divisor - 1.0f);
        }
    } else {
        index1 = index2 = (int)fmodf(relative_index + csi[3], p->PositionConstantCount);
        position_index_t = 0.0f;
    }

    /* PS_Spawn, SpawnParticles.fx:24-29 */
    f4 position1 = p->InlinePositionConstants[index1], position2 = p->InlinePositionConstants[index2];
    f4 position_constant = v4lerp(position1, position2, position_index_t);
    f4 towards_next = v4sub(position2, position1);

    /* Spawn_Stage2, SpawnerCommon.fxh:162-190 */
    const f4 zero = v4(0, 0, 0, 0);
    const f4* C = p->Configuration;
    f4 temp_position = evaluate_formula(zero, position_constant, C[0], C[1], random1, p->FormulaTypes[0], p->AxisMask);
    f4 new_position = mul_point(xyz(temp_position), &p->PositionMatrix);
    new_position.w = temp_position.w;

    f4 temp_velocity = evaluate_formula(temp_position, C[2], C[3], C[4], random2, p->FormulaTypes[1], p->AxisMask);
    f4 new_attributes = evaluate_formula(zero, C[5], C[6], C[7], random3, p->FormulaTypes[2], p->AxisMask);

    float towards_distance = sqrtf(towards_next.x * towards_next.x + towards_next.y * towards_next.y +
                                   towards_next.z * towards_next.z + towards_next.w * towards_next.w);
    if (towards_distance > 0.0001f) {
        /* scalars are promoted to float4; only .x of the result is used */
        f4 c8c = v4(C[8].x, C[8].x, C[8].x, C[8].x), c8s = v4(C[8].y, C[8].y, C[8].y, C[8].y),
           c8o = v4(C[8].z, C[8].z, C[8].z, C[8].z), r3w = v4(random3.w, random3.w, random3.w, random3.w);
        float towards_speed = evaluate_formula(zero, c8c, c8s, c8o, r3w, p->FormulaTypes[3], p->AxisMask).x;
        temp_velocity = v4add(temp_velocity, v4scale(v4(towards_next.x / towards_distance, towards_next.y / towards_distance,
                                                        towards_next.z / towards_distance, towards_next.w / towards_distance), towards_speed));
    }

    f4 new_velocity = mul_point(xyz(temp_velocity), &p->VelocityMatrix);
    new_velocity.w = temp_velocity.w;
    /* #if FNA nudge is off in the x86/XNA build (Illuminant.csproj:24-35) */

    if (new_attributes.w < p->AttributeDiscardThreshold)
        return; /* discard */

    *pos = new_position;
    *vel = new_velocity;
    *attr = new_attributes;
}

static void spawn_rows(IlmFloat4* pos, IlmFloat4* vel, IlmFloat4* attr, int32_t chunk_size, int y0, int y1,
                       const IlmFloat4* rnd, int32_t rw, int32_t rh, const IlmSpawnParams* p) {
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < chunk_size; x++) {
            int i = y * chunk_size + x;
            spawn_slot(&pos[i], &vel[i], &attr[i], (float)x, (float)y, rnd, rw, rh, p);
        }
}

void orc_spawn(IlmFloat4* pos, IlmFloat4* vel, IlmFloat4* attr, int32_t chunk_size,
               const IlmFloat4* rnd, int32_t rw, int32_t rh, const IlmSpawnParams* p) {
    #pragma omp parallel for schedule(static)
    for (int y = 0; y < chunk_size; y++)
        spawn_rows(pos, vel, attr, chunk_size, y, y + 1, rnd, rw, rh, p);
}

/* ---------------------------------------------------------------------------
 * Bezier.fxh:21-177
 * ------------------------------------------------------------------------- */
/* tForScaledBezier, Bezier.fxh:21-67 */
static float t_for_scaled_bezier(f4 range_and_count, float value, float* t_out) {
    float min_value = range_and_count.x, inv_divisor = range_and_count.y;
    unsigned mode = (unsigned)fabsf(range_and_count.w);
    int repeating = mode > 255, bouncing = mode > 511;

    float t = (value - min_value) * fabsf(inv_divisor);

    if (bouncing) {
        t *= 2.0f;
        if (inv_divisor < 0.0f)
            t = 2.0f - fmodf(t, 2.0f);
        else
            t = fmodf(t, 2.0f);
        if (t > 1.0f)
            t = 1.0f - (t - 1.0f);
    } else if (repeating) {
        if (inv_divisor < 0.0f)
            t = 1.0f - fmodf(t, 1.0f);
        else
            t = fmodf(t, 1.0f);
    } else {
        if (inv_divisor < 0.0f)
            t = 1.0f - h_sat(t);
        else
            t = h_sat(t);
    }

    switch (mode % 256) {
        default: break;
        case 1: t = sinf(t * H_PI * 0.5f); break;
        case 2: t = t * t; break;
    }
    *t_out = t;
    return range_and_count.z;
}

/* evaluateBezier1AtT, Bezier.fxh:69-99 */
float orc_bezier1(const IlmClampedBezier1* bz, float value) {
    float t;
    float count = t_for_scaled_bezier(bz->RangeAndCount, value, &t);
    float a = bz->ABCD.x, b = bz->ABCD.y, c = bz->ABCD.z, d = bz->ABCD.w;
    if (count <= 1.5f) return a;
    float ab = h_lerp(a, b, t);
    if (count <= 2.5f) return ab;
    if (count <= 3.5f) {
        if (t <= 0.0f) return a;
        else if (t >= 1.0f) return c;
        else return b;
    }
    float bc = h_lerp(b, c, t);
    float abbc = h_lerp(ab, bc, t);
    float cd = h_lerp(c, d, t);
    float bccd = h_lerp(bc, cd, t);
    return h_lerp(abbc, bccd, t);
}

/* evaluateBezier4AtT, Bezier.fxh:141-171 */
static f4 bezier4(const IlmClampedBezier4* bz, float value) {
    float t;
    float count = t_for_scaled_bezier(bz->RangeAndCount, value, &t);
    f4 a = bz->A, b = bz->B, c = bz->C, d = bz->D;
    if (count <= 1.5f) return a;
    f4 ab = v4lerp(a, b, t);
    if (count <= 2.5f) return ab;
    if (count <= 3.5f) {
        if (t <= 0.0f) return a;
        else if (t >= 1.0f) return c;
        else return b;
    }
    f4 bc = v4lerp(b, c, t);
    f4 abbc = v4lerp(ab, bc, t);
    f4 cd = v4lerp(c, d, t);
    f4 bccd = v4lerp(bc, cd, t);
    return v4lerp(abbc, bccd, t);
}
void orc_bezier4(const IlmClampedBezier4* b, float value, IlmFloat4* out) { *out = bezier4(b, value); }

/* ---------------------------------------------------------------------------
 * DistanceFieldCommon.fxh:264-353 -- encode/decode + sampleDistanceFieldEx
 * ------------------------------------------------------------------------- */

float orc_encode_distance(float distance, float max_encoded) { return DISTANCE_ZERO - (distance / max_encoded); }
float orc_decode_distance(float encoded, float max_encoded) { return (DISTANCE_ZERO - encoded) * max_encoded; }

static inline void sdf_texel(const OrcTexture* t, int x, int y, float out[4]) {
    const uint16_t* p = (const uint16_t*)t->texels + ((size_t)y * (size_t)t->width + (size_t)x) * 4;
    if (t->format == ILM_SDF_FP16) {
        for (int c = 0; c < 4; c++) out[c] = half_to_float(p[c]);
    } else {
        for (int c = 0; c < 4; c++) out[c] = (float)p[c] / 65535.0f;
    }
}

/* tex2Dlod on DistanceFieldTextureSampler: LINEAR min/mag, U WRAP, V CLAMP
 * (DistanceFieldCommon.fxh:273-281) */
/* Multiply-adds of the SDF sampler and of the cone-trace step are FUSED (one rounding), in exactly the places where the HIP path
 * issues v_fma_f32 / v_fmac_f32: the HLSL leaves this open (fxc emits mad / lrp for these expressions at will), so the restatement
 * fixes it once for both sides; fmaf is exact by definition, which keeps the two bit-identical. */
static inline float h_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
static inline float h_lerp_fused(float a, float b, float t) { return h_fma(t, b - a, a); }

static void sdf_sample_linear(const OrcTexture* t, float u, float v, float out[4]) {
    float x = h_fma(u, (float)t->width, -0.5f);
    float y = h_fma(v, (float)t->height, -0.5f);
    float x0f = floorf(x), y0f = floorf(y);
    float fx = x - x0f, fy = y - y0f;
    int x0 = wrap_index(x0f, t->width), x1 = wrap_index(x0f + 1.0f, t->width);
    int y0 = (int)y0f, y1 = (int)y0f + 1;
    if (y0 < 0) y0 = 0; if (y0 > t->height - 1) y0 = t->height - 1;
    if (y1 < 0) y1 = 0; if (y1 > t->height - 1) y1 = t->height - 1;
    float t00[4], t10[4], t01[4], t11[4];
    sdf_texel(t, x0, y0, t00); sdf_texel(t, x1, y0, t10);
    sdf_texel(t, x0, y1, t01); sdf_texel(t, x1, y1, t11);
    for (int c = 0; c < 4; c++) {
        float top = h_lerp_fused(t00[c], t10[c], fx);
        float bot = h_lerp_fused(t01[c], t11[c], fx);
        out[c] = h_lerp_fused(top, bot, fy);
    }
}

typedef struct { uint64_t samples; } SdfCounter;

/* sampleDistanceFieldEx, DistanceFieldCommon.fxh:313-353 */
static float sample_distance_field_ex(f3 position, const IlmDistanceFieldUniforms* df, const OrcTexture* sdf, SdfCounter* ctr) {
    if (ctr) ctr->samples++;
    position.z -= df->ConeAndMisc.y; /* getDistanceFieldZOffset */
    f3 extent = v3(df->Extent.x, df->Extent.y, df->Extent.z);
    f3 clamped = v3(h_clamp(position.x, 0.0f, extent.x), h_clamp(position.y, 0.0f, extent.y), h_clamp(position.z, 0.0f, extent.z));
    f3 dtv = v3(-fminf(position.x, 0.0f) + (fmaxf(position.x, extent.x) - extent.x),
                -fminf(position.y, 0.0f) + (fmaxf(position.y, extent.y) - extent.y),
                -fminf(position.z, 0.0f) + (fmaxf(position.z, extent.z) - extent.z));
    float distance_to_volume = sqrtf(h_fma(dtv.z, dtv.z, h_fma(dtv.y, dtv.y, dtv.x * dtv.x)));

    float slice_position = fminf(clamped.z, df->Packed1.z) * df->Packed1.y;
    float virtual_slice_index = floorf(slice_position);

    float texel_u = clamped.x * df->TextureSliceAndTexelSize.z;
    float texel_v = clamped.y * df->TextureSliceAndTexelSize.w;

    /* computeDistanceFieldSliceUv, DistanceFieldCommon.fxh:303-311 */
    float column_index = floorf(virtual_slice_index / 3.0f);
    float row_index = floorf(virtual_slice_index * df->Packed1.x);
    float u = h_fma(column_index, df->TextureSliceAndTexelSize.x, texel_u);
    float v = h_fma(row_index, df->TextureSliceAndTexelSize.y, texel_v);

    float packed[4];
    sdf_sample_linear(sdf, u, v, packed);

    float mask_pattern_index = fmodf(virtual_slice_index, 3.0f);
    float subslice = slice_position - virtual_slice_index, blended;
    if (mask_pattern_index >= 2.0f)
        blended = h_lerp_fused(packed[2], packed[3], subslice);
    else if (mask_pattern_index >= 1.0f)
        blended = h_lerp_fused(packed[1], packed[2], subslice);
    else
        blended = h_lerp_fused(packed[0], packed[1], subslice);

    /* decodeDistance(blended) + distanceToVolume */
    return h_fma(DISTANCE_ZERO - blended, df->Extent.w, distance_to_volume);
}

float orc_sample_distance_field(const float pos[3], const IlmDistanceFieldUniforms* df, const OrcTexture* sdf) {
    return sample_distance_field_ex(v3(pos[0], pos[1], pos[2]), df, sdf, NULL);
}

/* estimateNormal4, VisualizeCommon.fxh:44-63 with VISUALIZE_TEXEL (:8-15) */
static f3 estimate_normal4(f3 position, const IlmDistanceFieldUniforms* df, const OrcTexture* sdf) {
    f3 texel = v3(df->ConeAndMisc.w, df->StepAndMisc2.w, df->Extent.z / fmaxf(df->TextureSliceCount.w, 1.0f));
    static const float W[4][3] = { { 1, -1, -1 }, { -1, -1, 1 }, { -1, 1, -1 }, { 1, 1, 1 } };
    f3 result = v3(0, 0, 0);
    for (int i = 0; i < 4; i++) {
        f3 w = v3(W[i][0], W[i][1], W[i][2]);
        float s = sample_distance_field_ex(v3add(position, v3mul(w, texel)), df, sdf, NULL);
        result = v3add(result, v3scale(w, s));
    }
    return v3norm(result);
}

/* ---------------------------------------------------------------------------
 * UpdateCommon.fxh
 * ------------------------------------------------------------------------- */
/* applyFrictionAndMaximum, UpdateCommon.fxh:20-35 */
static f3 apply_friction_and_maximum(f3 velocity, const IlmParticleSystemUniforms* sys) {
    float l = v3len(velocity);
    if (l <= 0.001f)
        return v3(0, 0, 0);
    if (l > sys_max_velocity(sys))
        l = sys_max_velocity(sys);
    float friction = l * sys_friction(sys);
    l -= (friction * sys_dt_seconds(sys));
    l = h_clamp(l, 0.0f, sys_max_velocity(sys));
    return v3scale(v3norm(velocity), l);
}

/* readLifeRamp: LifeRampSampler POINT, U CLAMP, V WRAP (UpdateCommon.fxh:6-12,37-39) */
static f4 read_life_ramp(const f4* ramp, int w, int h, float u, float v) {
    if (!ramp || w <= 0 || h <= 0)
        return v4(1, 1, 1, 1);
    int tx = (int)floorf(u * (float)w);
    if (tx < 0) tx = 0; if (tx > w - 1) tx = w - 1;
    int ty = wrap_index(floorf(v * (float)h), h);
    return ramp[ty * w + tx];
}

/* getRotationForVelocity, UpdateCommon.fxh:81-94 */
static float rotation_for_velocity(f3 velocity) {
    if ((fabsf(velocity.x) < 0.01f) && (fabsf(velocity.y) < 0.01f))
        return 0.0f;
    float result = atan2f(velocity.y, velocity.x);
    if (result < 0.0f)
        result += 2.0f * H_PI;
    return result;
}

/* computeRenderData, UpdateCommon.fxh:96-117 */
static void compute_render_data(float vx, float vy, f4 position, f4 velocity, f4 attributes,
                                const IlmParticleSystemUniforms* sys, const IlmUpdateParams* p,
                                const f4* ramp, int ramp_w, int ramp_h, f4* render_color, f4* render_data) {
    if (position.w <= 0.0f) {
        *render_color = v4(0, 0, 0, 0);
        *render_data = v4(0, 0, 0, 0);
        return;
    }
    float index = vx + (vy * RD_INDEX_ROW_PITCH); /* FIXME in the reference: hard-coded 256, UpdateCommon.fxh:107 */
    float velocity_length = fmaxf(v3len(xyz(velocity)), 0.0001f);

    /* getRampedColorForLifeValueAndIndex, UpdateCommon.fxh:66-79 */
    f4 color = v4mul(bezier4(&p->ColorFromLife, position.w), bezier4(&p->ColorFromVelocity, velocity_length));
    if (p->LifeRampSettings.x != 0.0f) {
        float u = (position.w - p->LifeRampSettings.y) / p->LifeRampSettings.z;
        if (p->LifeRampSettings.x < 0.0f)
            u = 1.0f - h_sat(u);
        float v = index / p->LifeRampSettings.w;
        f4 ramped = v4mul(read_life_ramp(ramp, ramp_w, ramp_h, u, v), color);
        color = v4lerp(color, ramped, h_sat(fabsf(p->LifeRampSettings.x)));
    }

    f4 rc = v4mul(attributes, color);
    rc.w = h_sat(rc.w);
    rc.x *= rc.w; rc.y *= rc.w; rc.z *= rc.w;
    *render_color = rc;

    f4 rd;
    rd.x = orc_bezier1(&p->SizeFromLife, position.w) * orc_bezier1(&p->SizeFromVelocity, velocity_length);
    rd.y = (rotation_for_velocity(xyz(velocity)) * sys->AnimationRateAndRotationAndZToY.z) +
           ((position.w * p->RotationFromLifeAndIndex[0]) + (index * p->RotationFromLifeAndIndex[1]));
    rd.z = velocity_length;
    rd.w = velocity.w;
    *render_data = rd;
}

/* PS_Update, UpdateParticleSystem.fx:9-38 */
static void update_slot(f4* pos, f4* vel, const f4* attr, f4* rc, f4* rd, float x, float y,
                        const IlmParticleSystemUniforms* sys, const IlmUpdateParams* p,
                        const f4* ramp, int ramp_w, int ramp_h) {
    f4 old_position = *pos, old_velocity = *vel;
    const f4 zero = v4(0, 0, 0, 0);
    if (old_position.w <= 0.0f) { /* readStateOrDiscard: discard => cleared target survives */
        *pos = zero; *vel = zero; *rc = zero; *rd = zero;
        return;
    }
    f4 attributes = *attr;
    f3 velocity = apply_friction_and_maximum(xyz(old_velocity), sys);
    f3 scaled_velocity = v3scale(velocity, sys_dt_seconds(sys));
    float new_life = old_position.w - (sys_life_decay(sys) * sys_dt_seconds(sys));
    f4 new_position, new_velocity;
    if (new_life <= 0.0f) {
        new_position = zero;
        new_velocity = zero;
    } else {
        new_position = v4(old_position.x + scaled_velocity.x, old_position.y + scaled_velocity.y, old_position.z + scaled_velocity.z, new_life);
        new_velocity = v4(velocity.x, velocity.y, velocity.z, old_velocity.w);
    }
    compute_render_data(x, y, new_position, new_velocity, attributes, sys, p, ramp, ramp_w, ramp_h, rc, rd);
    *pos = new_position;
    *vel = new_velocity;
}

/* PS_Update, UpdateParticleSystemWithDistanceField.fx:29-147 */

static void update_df_slot(f4* pos, f4* vel, const f4* attr, f4* rc, f4* rd, float x, float y,
                           const IlmParticleSystemUniforms* sys, const IlmUpdateParams* p,
                           const f4* ramp, int ramp_w, int ramp_h,
                           const IlmDistanceFieldUniforms* df, const OrcTexture* sdf) {
    f4 old_position = *pos, old_velocity = *vel;
    const f4 zero = v4(0, 0, 0, 0);
    if (old_position.w <= 0.0f) {
        *pos = zero; *vel = zero; *rc = zero; *rd = zero;
        return;
    }
    f4 attributes = *attr;
    f4 new_velocity = zero;

    float new_life = old_position.w - (sys_life_decay(sys) * sys_dt_seconds(sys));
    if (new_life <= 0.0f) {
        *pos = zero; *vel = zero; *rc = zero; *rd = zero;
        return;
    }

    const float collision_distance = sys->CollisionSettings.z;
    f3 unit_vector = v3norm(xyz(old_velocity));
    f3 velocity = apply_friction_and_maximum(xyz(old_velocity), sys);

    int collided = 0, escaping = 0;
    f3 scaled_velocity = v3scale(velocity, sys_dt_seconds(sys));
    f3 old_xyz = xyz(old_position);
    f3 collision_position = v3(0, 0, 0), new_position = old_xyz;

    float initial_distance = sample_distance_field_ex(old_xyz, df, sdf, NULL);
    int was_colliding = initial_distance < collision_distance;
    float travel_distance = fmaxf(0.0f, fminf(initial_distance, v3len(scaled_velocity)));
    int step_count = DF_MAX_STEP_COUNT;
    if (was_colliding)
        step_count = 1;
    else if (travel_distance <= 0.001f)
        step_count = 0;

    for (int i = 0; i < step_count; i++) {
        f3 test_position = v3add(old_xyz, v3scale(unit_vector, travel_distance));
        float step_distance = sample_distance_field_ex(test_position, df, sdf, NULL);
        if (step_distance < collision_distance) {
            collided = 1;
            collision_position = test_position;
        }
        escaping = step_distance > initial_distance;

        if (collided && !escaping) {
            collision_position = test_position;
            float offset = h_clamp(step_distance + collision_distance, 0.05f, 16.0f);
            travel_distance = fmaxf(0.0f, travel_distance - offset);
        } else
            step_count = 0;

        if (travel_distance <= 0.001f)
            step_count = 0;
    }

    if (collided) {
        int bounce = old_velocity.w <= 0.0f;
        int redirect = was_colliding && !escaping;

        f3 normal = v3(0, 0, 0);
        if (bounce || redirect)
            normal = estimate_normal4(collision_position, df, sdf);

        float escape_speed = fminf(sys_max_velocity(sys), sys->CollisionSettings.x);

        if (redirect) {
            normal = v3mul(normal, v3(1, 1, 0)); /* ESCAPE_MASK */
            if (v3len(normal) < DF_NO_NORMAL_THRESHOLD) {
                float a = (x / 67.0f) + (y / 13.0f);
                normal = v3(sinf(a), cosf(a), 0.0f);
            }
            f3 escape_vector = v3norm(normal);
            f3 nv = v3scale(v3scale(escape_vector, escape_speed), DF_INITIAL_ESCAPE_SPEED);
            new_velocity = v4(nv.x, nv.y, nv.z, DF_BOUNCE_DELAY);
            f3 escape_delta = v3scale(nv, sys_dt_seconds(sys));
            new_position = v3add(old_xyz, escape_delta);
        } else if (bounce) {
            float d2 = 2.0f * v3dot(normal, unit_vector);
            f3 bounce_vector = v3scale(v3scale(v3sub(normal, unit_vector), d2), -1.0f);
            if (v3len(bounce_vector) < DF_NO_NORMAL_THRESHOLD)
                bounce_vector = v3scale(unit_vector, -1.0f);
            else
                bounce_vector = v3norm(bounce_vector);
            new_position = collision_position;
            float speed = fminf(sys_max_velocity(sys), v3len(velocity) * sys->CollisionSettings.y);
            f3 nv = v3scale(bounce_vector, speed);
            new_velocity = v4(nv.x, nv.y, nv.z, DF_BOUNCE_DELAY);
            new_life -= sys->CollisionSettings.w;
        } else {
            float current_speed = v3len(xyz(old_velocity));
            float new_speed = fmaxf(current_speed * DF_ESCAPE_SPEED_ACCELERATION, escape_speed);
            f3 nv = v3scale(unit_vector, new_speed);
            new_velocity = v4(nv.x, nv.y, nv.z, 0.0f); /* .w stays at its initial 0 (:36,:125) */
            new_position = v3add(old_xyz, v3scale(unit_vector, travel_distance));
        }
    } else {
        new_velocity = v4(velocity.x, velocity.y, velocity.z, fmaxf(old_velocity.w - 1.0f, 0.0f));
        new_position = v3add(old_xyz, v3scale(unit_vector, travel_distance));
    }

    if (new_life <= 0.0f) {
        new_position = v3(0, 0, 0);
        new_velocity = zero;
    }
    f4 result_position = v4(new_position.x, new_position.y, new_position.z, new_life);
    compute_render_data(x, y, result_position, new_velocity, attributes, sys, p, ramp, ramp_w, ramp_h, rc, rd);
    *pos = result_position;
    *vel = new_velocity;
}

static void update_rows(IlmFloat4* pos, IlmFloat4* vel, const IlmFloat4* attr,
                        IlmFloat4* render_color, IlmFloat4* render_data, int32_t chunk_size, int y0, int y1,
                        const IlmParticleSystemUniforms* sys, const IlmUpdateParams* p,
                        const IlmFloat4* life_ramp, int32_t ramp_w, int32_t ramp_h,
                        const IlmDistanceFieldUniforms* df, const OrcTexture* sdf) {
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < chunk_size; x++) {
            int i = y * chunk_size + x;
            if (df && sdf)
                update_df_slot(&pos[i], &vel[i], &attr[i], &render_color[i], &render_data[i], (float)x, (float)y,
                               sys, p, life_ramp, ramp_w, ramp_h, df, sdf);
            else
                update_slot(&pos[i], &vel[i], &attr[i], &render_color[i], &render_data[i], (float)x, (float)y,
                            sys, p, life_ramp, ramp_w, ramp_h);
        }
}

void orc_update(IlmFloat4* pos, IlmFloat4* vel, const IlmFloat4* attr,
                IlmFloat4* render_color, IlmFloat4* render_data, int32_t chunk_size,
                const IlmParticleSystemUniforms* sys, const IlmUpdateParams* p,
                const IlmFloat4* life_ramp, int32_t ramp_w, int32_t ramp_h,
                const IlmDistanceFieldUniforms* df, const OrcTexture* sdf) {
    #pragma omp parallel for schedule(static)
    for (int y = 0; y < chunk_size; y++)
        update_rows(pos, vel, attr, render_color, render_data, chunk_size, y, y + 1, sys, p, life_ramp, ramp_w, ramp_h, df, sdf);
}

/* PS_Erase, UpdateParticleSystem.fx:40-49 */
void orc_erase(IlmFloat4* pos, IlmFloat4* vel, IlmFloat4* render_color, IlmFloat4* render_data, int32_t chunk_size) {
    size_t n = (size_t)chunk_size * (size_t)chunk_size * sizeof(IlmFloat4);
    memset(pos, 0, n); memset(vel, 0, n); memset(render_color, 0, n); memset(render_data, 0, n);
}

/* CountLiveParticles.fx:5-40 + ProcessLivenessInfoData (ParticleEngine.cs:224-252):
 * each live slot adds 1/65535 to a 16-bit unorm channel, decoded as raw & 0xFFFF */
uint32_t orc_count_live(const IlmFloat4* pos, int32_t slots, int32_t saturate16) {
    uint32_t n = 0;
    for (int i = 0; i < slots; i++)
        if (pos[i].w > 0.0f) n++;
    if (saturate16 && n > LIVE_COUNT_SATURATION) n = LIVE_COUNT_SATURATION;
    return n;
}

#include "ilm_oracle_transforms.c"

/* ParticleSystem.Update pass order, ParticleSystem.cs:725-745 + UpdateChunk :791-856.
 * Every pass is slot-local (a slot reads only its own previous state, the randomness table and the SDF),
 * so the chunk table is cut into bands of rows and each band goes through spawn -> transforms -> update on
 * its own: same per-slot pass order as the reference's chunk-at-a-time draws, one OpenMP region per step. */
void orc_step(IlmFloat4** planes, int32_t chunk_count, int32_t chunk_size,
              const IlmFloat4* rnd, int32_t rw, int32_t rh,
              const IlmFloat4* life_ramp, int32_t ramp_w, int32_t ramp_h,
              const OrcTexture* sdf, const IlmStepDesc* desc, uint32_t* live_counts) {
    orc_step_ex(planes, chunk_count, chunk_size, rnd, rw, rh, life_ramp, ramp_w, ramp_h, sdf, desc, live_counts, NULL);
}

void orc_step_ex(IlmFloat4** planes, int32_t chunk_count, int32_t chunk_size,
                 const IlmFloat4* rnd, int32_t rw, int32_t rh,
                 const IlmFloat4* life_ramp, int32_t ramp_w, int32_t ramp_h,
                 const OrcTexture* sdf, const IlmStepDesc* desc, uint32_t* live_counts, const OrcStepExtras* ex) {
    static const OrcStepExtras no_extras;
    if (!ex) ex = &no_extras;
    /* SpatialNoise reads the Rgba64 copy of the randomness table */
    uint16_t* lp_owned = NULL;
    const uint16_t* lp = ex->low_precision_rnd;
    for (int o = 0; o < desc->OpCount && !lp; o++)
        if (desc->Ops[o].Type == ILM_OP_SPATIAL_NOISE) {
            lp_owned = (uint16_t*)malloc(sizeof(uint16_t) * 4 * (size_t)rw * (size_t)rh);
            orc_low_precision_randomness(rnd, rw * rh, lp_owned);
            lp = lp_owned;
        }
    int first = desc->FirstChunk, count = desc->ChunkCount;
    if (count < 0) { first = 0; count = chunk_count; }
    const int band = chunk_size >= 64 ? 16 : chunk_size;           /* rows per task */
    const int bands = (chunk_size + band - 1) / band;
    const int tasks = chunk_count * bands;
    #pragma omp parallel for schedule(dynamic, 1)
    for (int t = 0; t < tasks; t++) {
        const int c = t / bands;
        const int y0 = (t % bands) * band;
        const int y1 = (y0 + band < chunk_size) ? y0 + band : chunk_size;
        IlmFloat4 *pos = planes[c * 5 + 0], *vel = planes[c * 5 + 1], *attr = planes[c * 5 + 2],
                  *rc = planes[c * 5 + 3], *rd = planes[c * 5 + 4];
        /* spawners run first and may target chunks outside [first, first + count) */
        for (int s = 0; s < desc->SpawnCount; s++)
            if (desc->Spawns[s].ChunkIndex == c)
                spawn_record_rows(pos, vel, attr, chunk_size, y0, y1, rnd, rw, rh, &desc->Spawns[s], s, ex);
        if (c < first || c >= first + count)
            continue;
        for (int o = 0; o < desc->OpCount; o++) {
            const IlmTransformOp* op = &desc->Ops[o];
            switch (op->Type) {
                case ILM_OP_GRAVITY: gravity_rows(pos, vel, chunk_size, y0, y1, &desc->System, &op->u.Gravity); break;
                case ILM_OP_NOISE:   noise_rows(pos, vel, chunk_size, y0, y1, rnd, rw, rh, &desc->System, &op->u.Noise); break;
                case ILM_OP_FMA:     fma_rows(pos, vel, chunk_size, y0, y1, &desc->System, &op->u.FMA); break;
                case ILM_OP_MATRIX_MULTIPLY: matrix_multiply_rows(pos, vel, chunk_size, y0, y1, &desc->System, &op->u.MatrixMultiply); break;
                case ILM_OP_SPATIAL_NOISE:   spatial_noise_rows(pos, vel, chunk_size, y0, y1, lp, rw, rh, &desc->System, &op->u.SpatialNoise); break;
                default: break;
            }
        }
        switch (desc->UpdateMode) {
            case ILM_UPDATE_POSITIONS:
                update_rows(pos, vel, attr, rc, rd, chunk_size, y0, y1, &desc->System, &desc->Update, life_ramp, ramp_w, ramp_h, NULL, NULL);
                break;
            case ILM_UPDATE_WITH_DISTANCE_FIELD:
                update_rows(pos, vel, attr, rc, rd, chunk_size, y0, y1, &desc->System, &desc->Update, life_ramp, ramp_w, ramp_h, &desc->DistanceField, sdf);
                break;
            case ILM_UPDATE_ERASE: {   /* PS_Erase, UpdateParticleSystem.fx:40-49 */
                size_t o = (size_t)y0 * (size_t)chunk_size, n = (size_t)(y1 - y0) * (size_t)chunk_size * sizeof(IlmFloat4);
                memset(pos + o, 0, n); memset(vel + o, 0, n); memset(rc + o, 0, n); memset(rd + o, 0, n);
                break;
            }
            default: break;
        }
    }
    if (live_counts && (desc->Flags & ILM_STEP_COUNT_LIVE))
        for (int c = first; c < first + count && c < chunk_count; c++)
            live_counts[c] = orc_count_live(planes[c * 5 + 0], chunk_size * chunk_size, 0);
    free(lp_owned);
}

/* ---------------------------------------------------------------------------
 * Lighting: EnvironmentCommon.fxh, LightCommon.fxh, AOCommon.fxh, ConeTrace.fxh,
 * SphereLightCore.fxh, SphereLight.fx
 * ------------------------------------------------------------------------- */
/* decodeNormalSpherical, EnvironmentCommon.fxh:40-51 */
static f3 decode_normal_spherical(float ex, float ey) {
    if ((ex != 0.0f) || (ey != 0.0f)) {
        float ax = ex * 2.0f - 1.0f, ay = ey * 2.0f - 1.0f;
        float s = sinf(ax * H_PI), c = cosf(ax * H_PI);
        float phx = sqrtf(1.0f - ay * ay), phy = ay;
        return v3(c * phx, s * phx, phy);
    }
    return v3(0, 0, 0);
}

static void gbuffer_texel(const OrcTexture* g, int x, int y, float out[4]) {
    if (x < 0) x = 0; if (x > g->width - 1) x = g->width - 1;
    if (y < 0) y = 0; if (y > g->height - 1) y = g->height - 1;
    size_t i = ((size_t)y * (size_t)g->width + (size_t)x) * 4;
    if (g->format == ILM_GBUFFER_HALF4) {
        const uint16_t* p = (const uint16_t*)g->texels + i;
        for (int c = 0; c < 4; c++) out[c] = half_to_float(p[c]);
    } else {
        const float* p = (const float*)g->texels + i;
        for (int c = 0; c < 4; c++) out[c] = p[c];
    }
}

/* sampleGBuffer, LightCommon.fxh:58-144 */
static f3 sample_gbuffer(float spx, float spy, const IlmEnvironment* env, const OrcTexture* g,
                         f3* world_position, f3* normal, int* enable_shadows, int* fullbright) {
    *enable_shadows = 1;
    *fullbright = 0;
    f3 camera_position;
    const float vsx = env->GBufferTexelSizeAndMisc.z, vsy = env->GBufferTexelSizeAndMisc.w;
    const float rsx = env->ZAndScale.z, rsy = env->ZAndScale.w;
    const float maximum_z = env->ZAndScale.y, ground_z = env->ZAndScale.x;

    if (g && ((env->GBufferTexelSizeAndMisc.x != 0.0f) || (env->GBufferTexelSizeAndMisc.y != 0.0f))) {
        float sx = spx, sy = spy;
        if (env->GBufferViewportRelative != 0.0f) {
            sx /= vsx; sy /= vsy;
            sx += env->ViewportPosition[0]; sy += env->ViewportPosition[1];
        }
        float u = (sx + 0.5f) * env->GBufferTexelSizeAndMisc.x;
        float v = (sy + 0.5f) * env->GBufferTexelSizeAndMisc.y;
        float sample[4];
        /* POINT / CLAMP sampler (LightCommon.fxh:35-43) */
        gbuffer_texel(g, (int)floorf(u * (float)g->width), (int)floorf(v * (float)g->height), sample);

        float relative_y = sample[2];
        float world_z = sample[3];
        if (world_z < 0.0f) {
            world_z += 1.0f;
            world_z = -world_z;
            *enable_shadows = 0;
        } else if (world_z >= 9999.0f) {
            world_z = 0.0f;
            *enable_shadows = 0;
            *fullbright = 1;
        }
        world_z *= GBUFFER_Z_SCALE;
        world_z -= GBUFFER_Z_OFFSET;

        spx /= rsx; spy /= rsy;
        camera_position = v3(spx, spy, maximum_z + 0.01f);
        *world_position = v3((spx + 0.0f) / vsx + env->ViewportPosition[0],
                             (spy + relative_y) / vsy + env->ViewportPosition[1], world_z);
        if ((sample[0] != 0.0f) || (sample[1] != 0.0f))
            *normal = decode_normal_spherical(sample[0], sample[1]);
        else
            *normal = v3(0, 0, 0);
    } else {
        spx /= rsx; spy /= rsy;
        camera_position = v3(spx, spy, maximum_z + 0.01f);
        *world_position = v3(spx / vsx + env->ViewportPosition[0], spy / vsy + env->ViewportPosition[1], ground_z);
        *normal = v3(0, 0, 1);
    }
    return camera_position;
}

void orc_sample_gbuffer(float px, float py, const IlmEnvironment* env, const OrcTexture* gbuffer,
                        float world_pos[3], float normal[3], int32_t* enable_shadows, int32_t* fullbright, float camera_pos[3]) {
    f3 wp, n; int es, fb;
    f3 cam = sample_gbuffer(px, py, env, gbuffer, &wp, &n, &es, &fb);
    world_pos[0] = wp.x; world_pos[1] = wp.y; world_pos[2] = wp.z;
    normal[0] = n.x; normal[1] = n.y; normal[2] = n.z;
    camera_pos[0] = cam.x; camera_pos[1] = cam.y; camera_pos[2] = cam.z;
    *enable_shadows = es; *fullbright = fb;
}

/* checkShadowFilter, LightCommon.fxh:146-152 */
static int check_shadow_filter(float filter, int enable_shadows) {
    if (filter < 0.0f)
        return 0;
    return (filter > 0.5f) != (enable_shadows != 0);
}

/* computeNormalFactorEx, LightCommon.fxh:154-166 (DOT_OFFSET .15, RANGE .15, EXPONENT .85) */
static float compute_normal_factor(f3 light_normal, f3 n) {
    if ((n.x == 0.0f) && (n.y == 0.0f) && (n.z == 0.0f))
        return 1.0f;
    float d = v3dot(v3scale(light_normal, -1.0f), n);
    return powf(h_sat((d + LC_DOT_OFFSET) / LC_DOT_RAMP_RANGE), LC_DOT_EXPONENT);
}

/* computeSphereLightOpacity, LightCommon.fxh:174-214 */
static float compute_sphere_light_opacity(f3 shaded, f3 normal, f3 light_center, f4 light_properties,
                                          float y_distance_factor, const IlmEnvironment* env) {
    float light_radius = light_properties.x, light_ramp_length = light_properties.y, falloff_mode = light_properties.z;
    f3 distance3 = v3sub(shaded, light_center);
    distance3.y *= y_distance_factor;
    float distance = v3len(distance3);
    float distance_factor = 1.0f - h_sat((distance - light_radius) / light_ramp_length);

    float light_occlusion = env->ZToY.z;
    if (light_occlusion > 0.0f)
        distance_factor *= 1.0f - h_sat(distance3.z / light_occlusion);

    f3 light_normal = v3(distance3.x / distance, distance3.y / distance, distance3.z / distance);
    float normal_factor = compute_normal_factor(light_normal, normal);

    if (falloff_mode >= 2.0f) {
        distance_factor = 1.0f - h_sat(distance - light_radius);
        normal_factor = 1.0f;
    } else if (falloff_mode >= 1.0f) {
        distance_factor *= distance_factor;
    }
    return h_sat((normal_factor * distance_factor) + h_sat(light_radius - distance));
}

/* CalcSphereLightSpecularity, LightCommon.fxh:216-226 */
static float calc_sphere_light_specularity(f3 camera, f3 shaded, f3 normal, f3 light_center, float power) {
    f3 light_direction = v3sub(shaded, light_center);
    f3 h = v3norm(v3sub(v3norm(v3sub(camera, shaded)), light_direction));
    return powf(h_sat(v3dot(h, normal)), power);
}

/* computeAO, AOCommon.fxh:1-19 */
static float compute_ao(f3 shaded, f3 normal, f4 more, const IlmDistanceFieldUniforms* df, const OrcTexture* sdf, int visible, SdfCounter* ctr) {
    float ao_radius = more.x, ao_opacity = more.w;
    if ((ao_radius >= 0.5f) && (df->Extent.x > 0.0f) && visible && sdf) {
        float distance = sample_distance_field_ex(v3(shaded.x, shaded.y, shaded.z + normal.z * more.x), df, sdf, ctr);
        float clamped = h_clamp(distance, 0.0f, ao_radius);
        float result = 1.0f - h_sat(clamped / more.x);
        result *= result;
        result = 1.0f - result;
        return (1.0f - ao_opacity) + (result * ao_opacity);
    }
    return 1.0f;
}

/* coneTrace, ConeTrace.fxh:37-191 */
static float cone_trace(f3 light_center, float light_radius, float light_ramp, float growth, float falloff,
                        f3 shaded, const IlmDistanceFieldUniforms* df, const OrcTexture* sdf, int enable, SdfCounter* ctr) {
    (void)falloff;
    /* coneTraceInitialize, :37-50 (startAtEnd = false, startOffset = TRACE_INITIAL_OFFSET_PX) */
    f3 trace_vector = v3sub(light_center, shaded);
    float trace_length = v3len(trace_vector);
    f3 direction = v3(trace_vector.x / trace_length, trace_vector.y / trace_length, trace_vector.z / trace_length);
    float data_y = fmaxf(trace_length - light_radius, 1.0f);
    float data_x = CT_TRACE_INITIAL_OFFSET_PX;
    float data_z = 1.0f;

    /* createTraceConfig, :128-146 */
    float max_radius = h_clamp(light_radius, CT_MIN_CONE_RADIUS, df->ConeAndMisc.x);
    float ramp_length = fmaxf(light_ramp, 16.0f);
    float radius_growth_per_pixel = max_radius / ramp_length * growth;
    float cfg_x = max_radius, cfg_y = radius_growth_per_pixel, cfg_z = fmaxf(1.0f, df->Packed1.w);

    float steps_remaining = df->StepAndMisc2.x;
    float liveness = ((df->Extent.x > 0.0f) && enable && sdf) ? 1.0f : 0.0f;

    while (liveness > 0.0f) {
        steps_remaining -= 1.0f;
        /* coneTraceAdvance, :76-85 */
        f3 sp = v3(h_fma(direction.x, data_x, shaded.x), h_fma(direction.y, data_x, shaded.y), h_fma(direction.z, data_x, shaded.z));
        float sample = sample_distance_field_ex(sp, df, sdf, ctr);
        /* coneTraceStep, :52-74 */
        float local_sphere_radius = fminf(h_fma(cfg_y, data_x, CT_MIN_CONE_RADIUS), cfg_x);
        float local_visibility = ((sample + CT_HACK_DISTANCE_OFFSET) / local_sphere_radius);
        data_z = fminf(data_z, local_visibility);
        data_x += fmaxf(fabsf(sample) * df->StepAndMisc2.z, cfg_z);
        float step_liveness = h_sat(data_z - CT_FULLY_SHADOWED_THRESHOLD) * h_sat(data_y - data_x);
        liveness = steps_remaining * step_liveness;
    }

    float step_window_visibility = steps_remaining / CT_MAX_STEP_RAMP_WINDOW;
    float visibility = fminf(data_z, step_window_visibility);
    float final_result = powf(h_sat(h_sat(visibility - CT_FULLY_SHADOWED_THRESHOLD) / (CT_UNSHADOWED_THRESHOLD - CT_FULLY_SHADOWED_THRESHOLD)), df->ConeAndMisc.z);
    return enable ? final_result : 1.0f;
}

/* Raster footprint of one light: the 12-vertex cut-corner "sphere" quad
 * (FillSphereBuffer, LightingRenderer.cs:636-656) transformed by
 * SphereLightVertexShader (SphereLightCore.fxh:13-56).  A pixel is shaded for
 * the light iff its centre lies inside the union of three axis-aligned rects. */
static int light_covers_pixel(const IlmLightVertex* L, const IlmEnvironment* env, float cx, float cy) {
    const float cOne = 1.0f / 7.0f, mOne = 6.0f / 7.0f;
    float radius = L->LightProperties.x + L->LightProperties.y + 1.0f;
    float delta_y = radius - (radius / L->MoreLightProperties.z);
    float rx = radius, ry = radius - (delta_y / 2.0f);
    float tlx = L->LightPosition1.x - rx, tly = L->LightPosition1.y - ry;
    float brx = L->LightPosition1.x + rx, bry = L->LightPosition1.y + ry;
    float off = radius * env->ZToY.y + L->LightPosition1.z * env->ZToY.x;
    const float sx = env->GBufferTexelSizeAndMisc.z * env->ZAndScale.z, sy = env->GBufferTexelSizeAndMisc.w * env->ZAndScale.w;
    #define WX(w) ((h_lerp(tlx, brx, (w)) - env->ViewportPosition[0]) * sx)
    #define WY(w) (((h_lerp(tly, bry, (w)) - (((w) < 0.5f) ? off : 0.0f)) - env->ViewportPosition[1]) * sy)
    float x0 = WX(0.0f), x1 = WX(cOne), x2 = WX(mOne), x3 = WX(1.0f);
    float y0 = WY(0.0f), y1 = WY(cOne), y2 = WY(mOne), y3 = WY(1.0f);
    #undef WX
    #undef WY
    if ((cx >= x1) && (cx < x2) && (cy >= y0) && (cy < y3)) return 1;
    if ((cx >= x0) && (cx < x3) && (cy >= y1) && (cy < y2)) return 1;
    return 0;
}

/* The RampTexture of the light group being rendered (technique SphereLightWithDistanceRamp / SphereLightProbeWithDistanceRamp;
 * bound per LightTypeRenderStateKey, Illuminant/Lighting/LightingRenderer.cs:764-766).  NULL: techniques SphereLight / SphereLightProbe. */
static const IlmFloat4* g_light_ramp = NULL;
static int g_light_ramp_w = 0, g_light_ramp_h = 0;
void orc_set_light_ramp(const IlmFloat4* texels, int32_t width, int32_t height) {
    g_light_ramp = (texels && width > 0 && height > 0) ? texels : NULL;
    g_light_ramp_w = width; g_light_ramp_h = height;
}

/* SampleFromRamp2, RampCommon.fxh:4-21: tex2Dlod level 0, LINEAR, U CLAMP, V WRAP, texel centres at + 0.5 */
static f4 sample_from_ramp2(float u, float v) {
    const int w = g_light_ramp_w, h = g_light_ramp_h;
    const float sx = u * (float)w - 0.5f, sy = v * (float)h - 0.5f;
    float x0f = floorf(sx), y0f = floorf(sy);
    const float fx = sx - x0f, fy = sy - y0f;
    float x1f = x0f + 1.0f;
    if (!(x0f >= 0.0f)) x0f = 0.0f; if (x0f > (float)(w - 1)) x0f = (float)(w - 1);
    if (!(x1f >= 0.0f)) x1f = 0.0f; if (x1f > (float)(w - 1)) x1f = (float)(w - 1);
    const int x0 = (int)x0f, x1 = (int)x1f;
    const int y0 = wrap_index(y0f, h), y1 = wrap_index(y0f + 1.0f, h);
    return v4lerp(v4lerp(g_light_ramp[y0 * w + x0], g_light_ramp[y0 * w + x1], fx),
                  v4lerp(g_light_ramp[y1 * w + x0], g_light_ramp[y1 * w + x1], fx), fy);
}

/* SphereLightPixelEpilogue / ...WithRamp, SphereLightCore.fxh:83-119: the light's opacity per colour channel */
static f3 sphere_light_epilogue(float pre_trace_opacity, float cone_opacity, f3 distance_to_center, f4 even_more) {
    if (g_light_ramp == NULL) {
        const float o = pre_trace_opacity * cone_opacity;
        return v3(o, o, o);
    }
    const float angle = atan2f(distance_to_center.y, distance_to_center.x);
    const f4 rgb = sample_from_ramp2(pre_trace_opacity, (angle + even_more.z) * even_more.w);
    return v3(rgb.x * cone_opacity, rgb.y * cone_opacity, rgb.z * cone_opacity);
}

/* census hook (oracle/ilm_oracle_census.c, tools/open_ray_census.py): called for every traced pair after its march */
typedef struct OrcTraceEvent {
    int px, py, light;
    f3 start, light_center;
    float light_radius, light_ramp;
    uint64_t samples;        /* SDF samples this pair's cone trace took */
    float result;            /* coneTrace's return value */
} OrcTraceEvent;
typedef void (*OrcTraceHook)(void* user, const OrcTraceEvent* e);

/* One row of SphereLightPixelShader, SphereLight.fx:7-46, + additive blend onto the
 * ambient clear (LightingRenderer.cs:1013-1024) with fp32 accumulation */
static void sphere_lights_row(int py, const IlmLightVertex* lights, int32_t light_count,
                              const IlmEnvironment* env, const IlmDistanceFieldUniforms* df,
                              const OrcTexture* gbuffer, const OrcTexture* sdf, const float ambient[4],
                              IlmFloat4* lightmap, int32_t width, SdfCounter* ctr, uint64_t* pairs, uint64_t* traced,
                              OrcTraceHook hook, void* hook_user) {
    for (int px = 0; px < width; px++) {
        f4 acc = v4(ambient[0], ambient[1], ambient[2], ambient[3]);
        if (g_orc_blend_fp16) acc = v4(round_through_half(acc.x), round_through_half(acc.y), round_through_half(acc.z), round_through_half(acc.w));
        f3 shaded, normal;
        int enable_shadows, fullbright;
        f3 camera = sample_gbuffer((float)px, (float)py, env, gbuffer, &shaded, &normal, &enable_shadows, &fullbright);
        for (int li = 0; li < light_count; li++) {
            const IlmLightVertex* L = &lights[li];
            if (!light_covers_pixel(L, env, (float)px + 0.5f, (float)py + 0.5f))
                continue;
            (*pairs)++;
            if (fullbright || check_shadow_filter(L->EvenMoreLightProperties.x, enable_shadows))
                continue; /* discard */

            f4 light_properties = L->LightProperties;
            light_properties.w *= (float)enable_shadows;
            f4 more = L->MoreLightProperties;
            f3 light_center = xyz(L->LightPosition1);

            /* SphereLightPixelPrologue, SphereLightCore.fxh:58-81 */
            float distance_opacity = compute_sphere_light_opacity(shaded, normal, light_center, light_properties, more.z, env);
            int visible = (distance_opacity > 0.0f) && (shaded.x > -9999.0f);
            more.x *= fmaxf(0.0f, normal.z);
            if (!visible)
                continue; /* discard */

            /* SphereLightPixelCore, SphereLightCore.fxh:122-158 */
            float ao_opacity = compute_ao(shaded, normal, more, df, sdf, visible, ctr);
            float pre_trace_opacity = distance_opacity * ao_opacity;
            int trace_shadows = visible && (light_properties.w != 0.0f) && (pre_trace_opacity >= SL_SHADOW_OPACITY_THRESHOLD);
            if (trace_shadows) (*traced)++;
            f3 start = v3add(shaded, v3scale(normal, SL_SELF_OCCLUSION_HACK));
            const uint64_t samples_before = ctr->samples;
            float cone_opacity = cone_trace(light_center, light_properties.x, light_properties.y, 1.0f, more.y,
                                            start, df, sdf, trace_shadows, ctr);
            if (hook && trace_shadows) {
                OrcTraceEvent e = { px, py, li, start, light_center, light_properties.x, light_properties.y, ctr->samples - samples_before, cone_opacity };
                hook(hook_user, &e);
            }
            const f3 opacity = sphere_light_epilogue(pre_trace_opacity, cone_opacity, v3sub(shaded, light_center), L->EvenMoreLightProperties);

            float specularity = calc_sphere_light_specularity(camera, shaded, normal, light_center, L->Color2.w);
            const float cr = (L->Color1.x * L->Color1.w * opacity.x) + (L->Color2.x * specularity * opacity.x);
            const float cg = (L->Color1.y * L->Color1.w * opacity.y) + (L->Color2.y * specularity * opacity.y);
            const float cb = (L->Color1.z * L->Color1.w * opacity.z) + (L->Color2.z * specularity * opacity.z);
            if (g_orc_blend_fp16) {
                acc.x = round_through_half(acc.x + round_through_half(cr));
                acc.y = round_through_half(acc.y + round_through_half(cg));
                acc.z = round_through_half(acc.z + round_through_half(cb));
                acc.w = round_through_half(acc.w + 1.0f);
            } else {
                acc.x += cr;
                acc.y += cg;
                acc.z += cb;
                acc.w += 1.0f;
            }
        }
        if (lightmap) lightmap[(size_t)py * (size_t)width + (size_t)px] = acc;
    }
}

void orc_render_sphere_lights(const IlmLightVertex* lights, int32_t light_count,
                              const IlmEnvironment* env, const IlmDistanceFieldUniforms* df,
                              const OrcTexture* gbuffer, const OrcTexture* sdf,
                              const float ambient[4],
                              IlmFloat4* lightmap, int32_t width, int32_t height,
                              int32_t row_begin, int32_t row_end, IlmRenderStats* stats) {
    uint64_t total_samples = 0, total_pairs = 0, total_traced = 0;
    if (row_begin < 0) row_begin = 0;
    if (row_end > height) row_end = height;
    #pragma omp parallel for schedule(dynamic, 4) reduction(+:total_samples, total_pairs, total_traced)
    for (int py = row_begin; py < row_end; py++) {
        SdfCounter ctr = { 0 };
        uint64_t pairs = 0, traced = 0;
        sphere_lights_row(py, lights, light_count, env, df, gbuffer, sdf, ambient, lightmap, width, &ctr, &pairs, &traced, NULL, NULL);
        total_samples += ctr.samples;
        total_pairs += pairs;
        total_traced += traced;
    }
    if (stats) {
        stats->SdfSamples = total_samples;
        stats->PixelLightPairs = total_pairs;
        stats->TracedPairs = total_traced;
    }
}

#include "ilm_oracle_census.c"
#include "ilm_oracle_lights.c"
#include "ilm_oracle_output.c"

/* ---------------------------------------------------------------------------
 * Host-side integer / layout logic
 * ------------------------------------------------------------------------- */
/* Math.Round(double): banker's rounding (MidpointRounding.ToEven) */
static double cs_round(double v) { return nearbyint(v); }
static double cs_round_digits3(double v) { return nearbyint(v * 1000.0) / 1000.0; }

/* DistanceField ctor, SDF/DistanceField.cs:43-122 */
void orc_distance_field_layout(int32_t virtual_width, int32_t virtual_height, float virtual_depth,
                               int32_t requested_slice_count, double requested_resolution,
                               int32_t maximum_encoded_distance, OrcDistanceFieldLayout* out) {
    const int MaxSurfaceSize = 8192, PackedSliceCount = 3;
    if (requested_resolution < 0.05) requested_resolution = 0.05;
    else if (requested_resolution > 1) requested_resolution = 1;

    int candidate_w = (int)cs_round(virtual_width * requested_resolution);
    int candidate_h = (int)cs_round(virtual_height * requested_resolution);
    double frac_x = (double)virtual_width / candidate_w, frac_y = (double)virtual_height / candidate_h;
    double frac = (frac_x + frac_y) / 2;
    double resolution = cs_round_digits3(1.0 / frac);
    if (resolution < 0.05) resolution = 0.05;
    else if (resolution > 1) resolution = 1;

    int slice_w = (int)cs_round(virtual_width * resolution);
    int slice_h = (int)cs_round(virtual_height * resolution);
    int max_x = MaxSurfaceSize / slice_w, max_y = MaxSurfaceSize / slice_h;
    int max_slices = max_x * max_y * PackedSliceCount;

    int slice_count = requested_slice_count > 3 ? requested_slice_count : 3;
    slice_count = ((slice_count + 2) / 3) * 3;
    if (slice_count > max_slices) slice_count = max_slices;
    int physical = (int)ceilf(slice_count / (float)PackedSliceCount);

    int columns = max_x < physical ? max_x : physical;
    int rows_needed = (int)ceilf(physical / (float)max_x);
    if (rows_needed < 1) rows_needed = 1;
    int rows = max_y < rows_needed ? max_y : rows_needed;

    while ((rows < columns) && (rows < max_y)) {
        int new_rows = rows + 1;
        int new_cols = (int)ceilf(physical / (float)new_rows);
        if (new_rows > max_x) new_rows = max_x;
        if (new_cols > max_y) new_cols = max_y;
        if ((new_rows * new_cols) < physical) break;
        rows = new_rows;
        columns = new_cols;
    }

    out->virtual_width = virtual_width; out->virtual_height = virtual_height; out->virtual_depth = virtual_depth;
    out->resolution = resolution;
    out->slice_width = slice_w; out->slice_height = slice_h;
    out->slice_count = slice_count; out->physical_slice_count = physical;
    out->column_count = columns; out->row_count = rows;
    out->atlas_width = slice_w * columns; out->atlas_height = slice_h * rows;
    out->maximum_encoded_distance = maximum_encoded_distance;
}

/* Uniforms.DistanceField ctor (Uniforms.cs:90-110) + SetDistanceFieldParameters (LightingRenderer.cs:1916-1939) */
void orc_distance_field_uniforms(const OrcDistanceFieldLayout* l, int32_t valid_slice_count, float z_offset,
                                 float max_cone_radius, float occlusion_to_opacity_power, int32_t step_limit,
                                 float min_step_size, float long_step_factor, IlmDistanceFieldUniforms* out) {
    memset(out, 0, sizeof(*out));
    out->Extent = v4((float)l->virtual_width, (float)l->virtual_height, l->virtual_depth, (float)l->maximum_encoded_distance);
    float slice_z_size = l->virtual_depth / (float)l->slice_count;
    int valid = valid_slice_count < l->slice_count ? valid_slice_count : l->slice_count;
    out->TextureSliceCount = v4((float)l->column_count, (float)l->row_count, (float)valid * slice_z_size, (float)l->slice_count);
    out->TextureSliceAndTexelSize = v4(1.0f / (float)l->column_count, 1.0f / (float)l->row_count,
                                       1.0f / (float)(l->virtual_width * l->column_count),
                                       1.0f / (float)(l->virtual_height * l->row_count));
    out->ConeAndMisc = v4(max_cone_radius, z_offset, occlusion_to_opacity_power, (float)((double)l->virtual_width / l->slice_width));
    out->StepAndMisc2 = v4((float)step_limit, min_step_size, long_step_factor, (float)((double)l->virtual_height / l->slice_height));
    out->Packed1 = v4((float)((1.0f / fmaxf(0.0001f, out->TextureSliceCount.x)) * (1.0f / 3.0f)),
                      (float)((1.0f / fmaxf(0.0001f, out->Extent.z)) * out->TextureSliceCount.w),
                      out->TextureSliceCount.z, min_step_size);
}

/* SpawnerBase.BeginTick, ParticleSpawner.cs:152-189 (rng_draw = RNG.NextDouble()) */
int32_t orc_spawner_begin_tick(OrcSpawnerState* s, float min_rate, float max_rate, int32_t count_scale,
                               double rng_draw, double delta_time_seconds, int32_t maximum_total) {
    int32_t spawn_count;
    if (min_rate > max_rate)
        min_rate = max_rate;
    /* (maxRate - minRate) is a float subtraction; the rest is double */
    double current_rate = ((rng_draw * (double)(float)(max_rate - min_rate)) + (double)min_rate) * count_scale * delta_time_seconds;
    current_rate += s->rate_error;
    s->rate_error = 0;
    if (current_rate < 1) {
        s->rate_error = current_rate > 0 ? current_rate : 0;
        spawn_count = 0;
    } else {
        spawn_count = (int32_t)current_rate;
        s->rate_error = current_rate - spawn_count;
    }
    if (maximum_total >= 0) {
        int32_t scaled_total = maximum_total * count_scale;
        int32_t remaining = scaled_total - s->total_spawned;
        if (spawn_count > remaining) {
            spawn_count = remaining;
            s->rate_error = 0;
        }
    }
    return spawn_count;
}

/* SpawnerBase.EndTick, ParticleSpawner.cs:191-194 */
void orc_spawner_end_tick(OrcSpawnerState* s, int32_t requested, int32_t actual) {
    s->rate_error += requested - actual;
    s->total_spawned += actual;
}

#include "ilm_oracle_fields.c"
#include "ilm_oracle_gbuffer.c"

void orc_set_num_threads(int32_t n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int32_t orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
