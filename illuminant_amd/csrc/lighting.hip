// lighting.hip -- LightingRenderer sphere-light pass (SDF cone trace) for gfx950.
//
// The reference draws one instanced quad per light and lets the ROP add the
// results (Illuminant/Lighting/LightingRenderer.cs:1004-1169, technique
// SphereLight in Illuminant/Shaders/SphereLight.fx:7-46): every light re-reads
// the G-buffer and rounds through the lightmap format.  Here a 16x16 pixel tile
// is one workgroup: wave 0 bins the lights whose raster footprint touches the
// tile into an LDS list (wave64 ballot + popcount, light order preserved), then
// every thread decodes its G-buffer texel once and walks the list, accumulating
// in fp32 registers and storing the lightmap texel once.  The sum runs over eight
// index-defined parts of the light list combined as a tree, so that several
// workgroups can share a tile without changing a bit (r04: "the order of the sum"
// in sphere_lights_kernel; short launches end tapered, api.hip plan_light_split).
//
// Per-light constants (footprint rectangles, cone config, premultiplied colour)
// are prepared once per call by prepare_lights_kernel and fetched through the
// scalar cache (the list index is wave-uniform), so they cost no VGPRs.
// The cone trace is a chain of dependent SDF fetches: latency-bound, served by
// L1/L2/MALL (the 25 MB atlas is cache resident); no MFMA (no dense contraction).
#include "internal.hpp"

namespace ilm {

struct LightRec {
    float cx, cy, cz, radius;
    float ramp, falloff_mode, casts_shadows, ao_radius;
    float falloff_y, ao_opacity, shadow_filter, spec_power;
    float col_r, col_g, col_b, flags;           // Color1.rgb * Color1.a; kLightHasSpecular | kLightFastTrace | kLightFastDivide (as a float)
    float spec_r, spec_g, spec_b, ramp_rcp;     // refined_rcp(ramp) for the distance ramp's division when kLightFastDivide
    float fx0, fx1, fx2, fx3;                   // raster footprint (screen px), see light_covers
    float fy0, fy1, fy2, fy3;
    float cfg_x, cfg_y, ramp_offset, ramp_rate; // createTraceConfig: maxRadius, radiusGrowthPerPixel; EvenMoreLightProperties.zw
};
static_assert(sizeof(LightRec) == 128, "LightRec is one 128-byte record");
constexpr int kLightHasSpecular = 1;    // Color2.rgb != 0
constexpr int kLightFastTrace = 2;      // light + field admit the in-volume trace loop (uniform half of shade_light's test; see light_flags)
constexpr int kLightFastDivide = 4;     // radius and ramp lie in the operand range of the unscaled division

// What the in-volume trace loop asks of a light and the field, decided once per light (prepare kernels): the centre inside the table
// sampler's box with the margin for rounding, a cone radius of ordinary size, a sane encoded distance, a trace that does not
// overshoot the light (radius >= 0), a step budget the uniform counter can count.
ILM_DEV int light_flags(const LightRec& r, const TraceGate& g) {
    int f = 0;
    if ((r.spec_r != 0.0f) || (r.spec_g != 0.0f) || (r.spec_b != 0.0f)) f |= kLightHasSpecular;
    const bool inside = (r.cx >= g.x0) & (r.cx <= g.x1) & (r.cy >= g.y0) & (r.cy <= g.y1) & (r.cz >= g.z0) & (r.cz <= g.z1);
    if (inside && (g.field_ok != 0.0f) && (r.cfg_x >= 0x1p-60f) && (r.cfg_x <= 0x1p60f) && (r.radius >= 0.0f)) f |= kLightFastTrace;
    if ((r.ramp >= 0x1p-60f) && (r.ramp <= 0x1p60f) && (fabsf(r.radius) <= 0x1p59f)) f |= kLightFastDivide;
    return f;
}

// SphereLightVertexShader (SphereLightCore.fxh:13-56) over the 12-vertex cut-corner
// quad (FillSphereBuffer, LightingRenderer.cs:636-656) + createTraceConfig (ConeTrace.fxh:128-146)
__global__ __launch_bounds__(64) void prepare_lights_kernel(const IlmLightVertex* __restrict__ lights, int count, IlmEnvironment env,
                                                             float max_cone_radius, TraceGate gate, LightRec* __restrict__ out) {
    const int i = (int)blockIdx.x * 64 + (int)threadIdx.x;
    if (i >= count) return;
    const IlmLightVertex L = lights[i];
    LightRec r;
    r.cx = L.LightPosition1.x; r.cy = L.LightPosition1.y; r.cz = L.LightPosition1.z;
    r.radius = L.LightProperties.x; r.ramp = L.LightProperties.y; r.falloff_mode = L.LightProperties.z; r.casts_shadows = L.LightProperties.w;
    r.ao_radius = L.MoreLightProperties.x; r.falloff_y = L.MoreLightProperties.z; r.ao_opacity = L.MoreLightProperties.w;
    r.shadow_filter = L.EvenMoreLightProperties.x;
    r.col_r = L.Color1.x * L.Color1.w; r.col_g = L.Color1.y * L.Color1.w; r.col_b = L.Color1.z * L.Color1.w;
    r.spec_r = L.Color2.x; r.spec_g = L.Color2.y; r.spec_b = L.Color2.z; r.spec_power = L.Color2.w;

    const float cOne = 1.0f / 7.0f, mOne = 6.0f / 7.0f;
    const float radius = L.LightProperties.x + L.LightProperties.y + 1.0f;
    const float delta_y = radius - (radius / L.MoreLightProperties.z);
    const float rx = radius, ry = radius - (delta_y / 2.0f);
    const float tlx = r.cx - rx, tly = r.cy - ry, brx = r.cx + rx, bry = r.cy + ry;
    const float off = radius * env.ZToY.y + r.cz * env.ZToY.x;
    const float sx = env.GBufferTexelSizeAndMisc.z * env.ZAndScale.z, sy = env.GBufferTexelSizeAndMisc.w * env.ZAndScale.w;
    r.fx0 = (lerp(tlx, brx, 0.0f) - env.ViewportPosition[0]) * sx;
    r.fx1 = (lerp(tlx, brx, cOne) - env.ViewportPosition[0]) * sx;
    r.fx2 = (lerp(tlx, brx, mOne) - env.ViewportPosition[0]) * sx;
    r.fx3 = (lerp(tlx, brx, 1.0f) - env.ViewportPosition[0]) * sx;
    r.fy0 = ((lerp(tly, bry, 0.0f) - off) - env.ViewportPosition[1]) * sy;
    r.fy1 = ((lerp(tly, bry, cOne) - off) - env.ViewportPosition[1]) * sy;
    r.fy2 = ((lerp(tly, bry, mOne) - 0.0f) - env.ViewportPosition[1]) * sy;
    r.fy3 = ((lerp(tly, bry, 1.0f) - 0.0f) - env.ViewportPosition[1]) * sy;

    const float max_radius = clampf(r.radius, ref::kMinConeRadius, max_cone_radius);
    r.cfg_x = max_radius;
    r.cfg_y = max_radius / fmaxf(r.ramp, 16.0f) * 1.0f;  // getConeGrowthFactor() == 1 (DistanceFieldCommon.fxh:233-236)
    r.ramp_offset = L.EvenMoreLightProperties.z; r.ramp_rate = L.EvenMoreLightProperties.w;
    const int flags = light_flags(r, gate);
    r.flags = (float)flags;
    r.ramp_rcp = (flags & kLightFastDivide) ? refined_rcp(r.ramp) : 0.0f;
    out[i] = r;
}

// decodeNormalSpherical, EnvironmentCommon.fxh:40-51
ILM_DEV f3 decode_normal(float ex, float ey) {
    const float ax = ex * 2.0f - 1.0f, ay = ey * 2.0f - 1.0f;
    const float s = sinf(ax * kPi), c = cosf(ax * kPi);
    const float phx = sqrtf(1.0f - ay * ay);
    return mk3(c * phx, s * phx, ay);
}

struct Pixel {
    f3 shaded, normal;
    // the camera position (sampleGBuffer's third output, read by the specular term alone) is a function of the pixel's coordinates:
    // not kept per lane -- the pixel of lane l is (origin_x + (l & 7), origin_y + (l >> 3)), and shade_light forms it where a light has a
    // specular colour (the same operations on the same values)
    int origin_x, origin_y;
    bool enable_shadows, fullbright;
    // light-independent half of the in-volume trace test: the trace start (shaded + normal * SELF_OCCLUSION_HACK) lies in the table
    // sampler's box with the start-side margin (set by trace_start_inside once per pixel)
    bool start_inside;
};

// Samples lie on the segment start -> light centre, or (trace shorter than the minimum length 1) within 1 of start: the start must
// keep 1 + the rounding margin from the box faces, the light centre (TraceGate, per light) the rounding margin alone.
ILM_DEV bool trace_start_inside(const Pixel& P, const IlmDistanceFieldUniforms& df, const SdfView& sdf) {
    const f3 start = P.shaded + (P.normal * ref::kSelfOcclusionHack);
    const float mx = 1.0625f + df.Extent.x * 0x1p-16f, my = 1.0625f + df.Extent.y * 0x1p-16f, mz = 1.0625f + df.Extent.z * 0x1p-16f;
    return (start.x >= sdf.box_x0 + mx) & (start.x <= sdf.box_x1 - mx) & (start.y >= sdf.box_y0 + my) & (start.y <= sdf.box_y1 - my) &
           (start.z >= sdf.box_z0 + mz) & (start.z <= sdf.box_z1 - mz);
}

// sampleGBuffer, LightCommon.fxh:58-144
ILM_DEV Pixel sample_gbuffer(float spx, float spy, const IlmEnvironment& env, const GBufferView& g) {
    Pixel p;
    p.enable_shadows = true;
    p.fullbright = false;
    const float vsx = env.GBufferTexelSizeAndMisc.z, vsy = env.GBufferTexelSizeAndMisc.w;
    const float rsx = env.ZAndScale.z, rsy = env.ZAndScale.w;
    if (g.texels != nullptr && ((env.GBufferTexelSizeAndMisc.x != 0.0f) || (env.GBufferTexelSizeAndMisc.y != 0.0f))) {
        float sx = spx, sy = spy;
        if (env.GBufferViewportRelative != 0.0f) {
            sx /= vsx; sy /= vsy;
            sx += env.ViewportPosition[0]; sy += env.ViewportPosition[1];
        }
        const float u = (sx + 0.5f) * env.GBufferTexelSizeAndMisc.x;
        const float v = (sy + 0.5f) * env.GBufferTexelSizeAndMisc.y;
        const int tx = min(max((int)floorf(u * (float)g.width), 0), g.width - 1);
        const int ty = min(max((int)floorf(v * (float)g.height), 0), g.height - 1);
        float4 s;
        if (g.format == ILM_GBUFFER_HALF4) {
            const uint2 raw = reinterpret_cast<const uint2*>(g.texels)[(size_t)ty * (size_t)g.width + (size_t)tx];
            s = mk4(__half2float(__ushort_as_half((unsigned short)(raw.x & 0xFFFFu))), __half2float(__ushort_as_half((unsigned short)(raw.x >> 16))),
                    __half2float(__ushort_as_half((unsigned short)(raw.y & 0xFFFFu))), __half2float(__ushort_as_half((unsigned short)(raw.y >> 16))));
        } else {
            s = reinterpret_cast<const float4*>(g.texels)[(size_t)ty * (size_t)g.width + (size_t)tx];
        }
        const float relative_y = s.z;
        float world_z = s.w;
        if (world_z < 0.0f) {
            world_z += 1.0f;
            world_z = -world_z;
            p.enable_shadows = false;
        } else if (world_z >= 9999.0f) {
            world_z = 0.0f;
            p.enable_shadows = false;
            p.fullbright = true;
        }
        world_z *= ref::kGBufferZScale;
        world_z -= ref::kGBufferZOffset;
        spx /= rsx; spy /= rsy;
        p.shaded = mk3((spx + 0.0f) / vsx + env.ViewportPosition[0], (spy + relative_y) / vsy + env.ViewportPosition[1], world_z);
        if ((s.x != 0.0f) || (s.y != 0.0f))
            p.normal = decode_normal(s.x, s.y);
        else
            p.normal = mk3(0.0f, 0.0f, 0.0f);
    } else {
        spx /= rsx; spy /= rsy;
        p.shaded = mk3(spx / vsx + env.ViewportPosition[0], spy / vsy + env.ViewportPosition[1], env.ZAndScale.x);
        p.normal = mk3(0.0f, 0.0f, 1.0f);
    }
    return p;
}

// computeSphereLightOpacity + computeNormalFactor, LightCommon.fxh:154-214.
// SHARED: every divisor of this function lies in the operand range of the unscaled division (the caller's wave-uniform test):
// the three divisions by `distance` share one refined reciprocal, the division by the light's ramp uses the one its record carries,
// the division by DOT_RAMP_RANGE the constant's -- the same correctly rounded quotients as `/` (tests: ilm_debug_divide,
// ilm_debug_divide_by_constant over all 2^32 numerators), 9 + 9 + 6 + 6 instructions instead of five 12-instruction sequences.
constexpr float kDotRampRangeRcp = (float)(1.0 / (double)ref::kDotRampRange);
constexpr float kVisibilityRange = ref::kUnshadowedThreshold - ref::kFullyShadowedThreshold;
constexpr float kVisibilityRangeRcp = (float)(1.0 / (double)kVisibilityRange);
// FLAT (with SHARED only): every lane's normal has x = y = 0 (the ground plane, flat G-buffer texels; the caller's wave-uniform test).
// dot(-lightNormal, normal) is then ((-ln.x * 0) + (-ln.y * 0)) + (-ln.z * normal.z) = -ln.z * normal.z exactly -- the two vanishing
// products are signed zeros (all components are finite: SHARED's range test bounds `distance`, hence d3), and a zero added to the third
// product leaves it unchanged (the sign of a zero RESULT may differ, which the following + DOT_OFFSET erases) -- so the x and y
// components of lightNormal, two 6-instruction divisions, are never formed.
template <bool SHARED, bool FLAT = false>
ILM_DEV float sphere_light_opacity(f3 d3, float distance, f3 normal, const LightRec& L, float light_occlusion) {
    const float over = distance - L.radius;
    float distance_factor = 1.0f - sat(SHARED ? div_with_rcp(over, L.ramp, L.ramp_rcp) : (over / L.ramp));
    if (light_occlusion > 0.0f)
        distance_factor *= 1.0f - sat(d3.z / light_occlusion);
    float normal_factor = 1.0f;
    if ((normal.x != 0.0f) || (normal.y != 0.0f) || (normal.z != 0.0f)) {
        float d;
        if (SHARED && FLAT) {
            const float y = refined_rcp(distance);
            d = (div_with_rcp(d3.z, distance, y) * -1.0f) * normal.z;
        } else {
            f3 ln;
            if (SHARED) {
                const float y = refined_rcp(distance);
                ln = mk3(div_with_rcp(d3.x, distance, y), div_with_rcp(d3.y, distance, y), div_with_rcp(d3.z, distance, y));
            } else {
                ln = mk3(d3.x / distance, d3.y / distance, d3.z / distance);
            }
            d = dot3(ln * -1.0f, normal);
        }
        const float ramped = SHARED ? div_with_rcp(d + ref::kDotOffset, ref::kDotRampRange, kDotRampRangeRcp) : ((d + ref::kDotOffset) / ref::kDotRampRange);
        normal_factor = pow_pos(sat(ramped), ref::kDotExponent);
    }
    if (L.falloff_mode >= 2.0f) {
        distance_factor = 1.0f - sat(distance - L.radius);
        normal_factor = 1.0f;
    } else if (L.falloff_mode >= 1.0f) {
        distance_factor *= distance_factor;
    }
    return sat((normal_factor * distance_factor) + sat(L.radius - distance));
}

struct LightStats { unsigned long long samples = 0, pairs = 0, traced = 0; };

// The partial sums of a light split travel between workgroups (any XCD) as device-scope accesses: `sc1` stores are written through the
// XCD's L2, `sc1` loads are served past the CU's L1 -- valid without cache write-backs or invalidates when BOTH sides use them
// (MI355X_MICROARCH.md, "inter-workgroup visibility") -- as ONE 16-byte access per lane (a dword `sc1` store is a fabric write of its
// own, ~6 x the time per byte).  Written as instructions: the compiler has no 16-byte device-scope access to offer.  The store is not
// waited for here (the caller's `s_waitcnt vmcnt(0)` before the ticket is); the loads of a tile's sums are issued together and waited for.
ILM_DEV void store_partial(float4* where, float r, float g, float b, float a) {
    const f32x4 v = { r, g, b, a };
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(where), "v"(v) : "memory");
}
// K = 2 or 4 partial sums `stride` float4 apart, all requested before the first is waited for (for callers with registers to spare)
template <int K>
ILM_DEV void load_partials(const float4* first, size_t stride, float4 (&out)[K]) {
    f32x4 v[K];
    if constexpr (K == 2) {
        asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]) : "v"(first), "v"(first + stride) : "memory");
    } else if constexpr (K == 4) {
        asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\t"
                     "global_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                     : "v"(first), "v"(first + stride), "v"(first + 2 * stride), "v"(first + 3 * stride) : "memory");
    }
#pragma unroll
    for (int i = 0; i < K; i++) out[i] = mk4(v[i].x, v[i].y, v[i].z, v[i].w);
}
// the sums of a tile's K = 2, 4 or 8 members, `stride` float4 apart, combined as the upper levels of the tree over the parts:
// ((m0 + m1) + (m2 + m3)) + ((m4 + m5) + (m6 + m7)), left operand first as in close_part
ILM_DEV float4 combine_partials(const float4* first, size_t stride, int count) {
    auto add4 = [](const float4& l, const float4& r) { return mk4(l.x + r.x, l.y + r.y, l.z + r.z, l.w + r.w); };
    if (count == 2) { float4 m[2]; load_partials<2>(first, stride, m); return add4(m[0], m[1]); }
    if (count == 4) { float4 m[4]; load_partials<4>(first, stride, m); return add4(add4(m[0], m[1]), add4(m[2], m[3])); }
    // (eight: two requests of four -- 32 registers of sums at once push the kernel's allocation into scratch)
    float4 m[4], n[4];
    load_partials<4>(first, stride, m);
    const float4 lower = add4(add4(m[0], m[1]), add4(m[2], m[3]));
    load_partials<4>(first + 4 * stride, stride, n);
    return add4(lower, add4(add4(n[0], n[1]), add4(n[2], n[3])));
}
// What a trace needs besides the light: the field (general sampler) and, for the in-volume loop, its uniform constants and the
// workgroup's slice table in LDS (table == nullptr: no in-volume loop -- light probes, fields the table cannot describe)
struct TraceField {
    const IlmDistanceFieldUniforms& df;
    const TraceSdfView& sdf;
    const InsideConsts& inside;
    const SliceEntry* table;
};

// The cone-trace loop (coneTraceAdvance + coneTraceStep, ConeTrace.fxh:52-85).  FAST: every sample of every active lane lies inside the
// box of the field's in-volume sampler and the light's cone radius is of ordinary size (shade_light decides per wave): the table
// sampler, division without the range scaling, the step budget as a uniform counter.
template <int FMT, bool STATS, bool FAST>
ILM_DEV void cone_trace_loop(const f3& start, const f3& dir, float data_y, float cfg_z, float cone_growth, float cone_max_radius,
                             const TraceField& F, float& data_x, float& data_z, float& steps_remaining, bool alive, LightStats& st) {
    // liveness = stepsRemaining * (saturate(visibility - FULLY_SHADOWED) * saturate(length - position)) > 0 is, factor by factor,
    // stepsRemaining > 0 && visibility > FULLY_SHADOWED && length > position: the smallest positive factors (ulp(0.075), ulp(0.5),
    // 1) cannot underflow their product, and a NaN factor saturates to 0 and fails its compare alike.
    const float long_step = F.df.StepAndMisc2.z;
    if (FAST) {
        // Every lane enters with the same budget S (a uniform) and loses 1 per iteration: after k iterations a lane that is still
        // in the loop holds S - k, exactly (S < 2^24: subtracting 1 from a positive float towards zero is exact, and the last step,
        // which may cross zero, rounds once either way).  So `stepsRemaining > 0` is the scalar test k < S, and a lane's budget at its
        // exit is S - (iterations it ran).
        const float budget = steps_remaining;
        const int most = (int)ceilf(budget);          // iterations k = 1 .. with S - k > 0 beforehand: k < S
        int k = 0;
        float ran = 0.0f;
        constexpr float kVisibilityGuard = 1.00000095367431640625f;      // 1 + 2^-20
        float guard_z = data_z * kVisibilityGuard;
        bool lit = data_z > ref::kFullyShadowedThreshold;
        while (alive) {
            k++;
            ran = (float)k;
            const f3 sp = mk3(__builtin_fmaf(dir.x, data_x, start.x), __builtin_fmaf(dir.y, data_x, start.y), __builtin_fmaf(dir.z, data_x, start.z));
            const float s = sample_inside_table<FMT>(sp, F.inside, F.sdf, F.table);
            if (STATS) st.samples++;
            const float local_radius = __builtin_elementwise_minimum(__builtin_fmaf(cone_growth, data_x, ref::kMinConeRadius), cone_max_radius);
            // visibility = min(visibility, n / r) with n = distance + HACK_DISTANCE_OFFSET (coneTraceStep, ConeTrace.fxh:62-63).  The quotient
            // only matters when it is below the running minimum v, and away from obstacles it never is.  With g = fl(v (1 + 2^-20)), kept
            // beside v: n >= fl(g r) implies n / r > v (the two roundings lose at most 2^-23 relative, r > 0, v > FULLY_SHADOWED_THRESHOLD
            // while the lane is alive), hence RN(n / r) >= v and the minimum keeps v -- exactly.  When NO lane of the wave can lower its
            // visibility the 11-instruction division is skipped (a multiply and a compare decide); a NaN fails the test and takes the
            // division like everything else.  cfg5 10.65 -> 10.01 ms; cfg3 (four times the obstacle density, 2 x 2 texels per wave) unchanged.
            const float n = s + ref::kHackDistanceOffset;
            if (__builtin_amdgcn_ballot_w64(!(n >= guard_z * local_radius)) != 0ull) {
                const float local_visibility = div_no_scale(n, local_radius);
                asm("v_min_f32 %0, %0, %1" : "+v"(data_z) : "v"(local_visibility));
                guard_z = data_z * kVisibilityGuard;
                lit = data_z > ref::kFullyShadowedThreshold;       // (visibility changes here and nowhere else)
            }
            float step = fabsf(s) * long_step;
            asm("v_max_f32 %0, %0, %1" : "+v"(step) : "v"(cfg_z));
            data_x += step;
            alive = (k < most) & lit & (data_y > data_x);
        }
        steps_remaining = budget - ran;
        return;
    }
    while (alive) {
        steps_remaining -= 1.0f;
        const f3 sp = mk3(__builtin_fmaf(dir.x, data_x, start.x), __builtin_fmaf(dir.y, data_x, start.y), __builtin_fmaf(dir.z, data_x, start.z));
        const float s = sample_distance_field<FMT, false>(sp, F.df, F.sdf);
        if (STATS) st.samples++;
        // (both operands are finite: v_minimum3_f32 needs no canonicalising v_max in front of it, unlike IEEE minNum)
        const float local_radius = __builtin_elementwise_minimum(__builtin_fmaf(cone_growth, data_x, ref::kMinConeRadius), cone_max_radius);
        // (the same exact skip of the division as in the in-volume loop; a cone radius that is not positive -- the general path admits
        // any light -- always divides)
        const float n = s + ref::kHackDistanceOffset;
        if (__builtin_amdgcn_ballot_w64(!((n >= (data_z * 1.00000095367431640625f) * local_radius) & (local_radius > 0.0f) & (data_z > 0.0f))) != 0ull) {
            const float local_visibility = n / local_radius;
            // fminf / fmaxf as the bare instructions: the same minNum / maxNum result without the v_max x, x canonicalisation the
            // compiler puts in front of each loop-carried operand (no signalling NaN can reach them)
            asm("v_min_f32 %0, %0, %1" : "+v"(data_z) : "v"(local_visibility));
        }
        float step = fabsf(s) * long_step;
        asm("v_max_f32 %0, %0, %1" : "+v"(step) : "v"(cfg_z));
        data_x += step;
        alive = (steps_remaining > 0.0f) & (data_z > ref::kFullyShadowedThreshold) & (data_y > data_x);
    }
}

// One light on one shaded point: SphereLightPixelShader (SphereLight.fx:7-46) after the raster test.  Returns false when the shader
// discards (nothing is blended); otherwise the light's rgb contribution in (out_r, out_g, out_b).
// flat_normals: wave-uniform, every lane's normal has x = y = 0 (see sphere_light_opacity)
template <int FMT, bool STATS>
ILM_DEV bool shade_light(const Pixel& P, const LightRec& L, const IlmEnvironment& env, const TraceField& F,
                         bool have_sdf, const RampView& ramp, LightStats& st, float& out_r, float& out_g, float& out_b, bool flat_normals = false) {
    const IlmDistanceFieldUniforms& df = F.df;
    const SdfView& sdf = F.sdf;
    // checkShadowFilter, LightCommon.fxh:146-152
    const bool filtered = (L.shadow_filter < 0.0f) ? false : ((L.shadow_filter > 0.5f) != P.enable_shadows);
    if (P.fullbright || filtered)
        return false;

    const float casts = L.casts_shadows * (P.enable_shadows ? 1.0f : 0.0f);
    const int light_flags_i = (int)L.flags;
    f3 d3 = P.shaded - mk3(L.cx, L.cy, L.cz);
    d3.y *= L.falloff_y;
    const float distance = len3(d3);
    // the shared-reciprocal divisions need distance (a divisor, and with the radius the ramp's numerator) inside their operand range:
    // one wave-uniform test, the IEEE form otherwise
    const bool divisors_ordinary = (distance >= 0x1p-60f) & (distance <= 0x1p59f);
    const bool shared = (light_flags_i & kLightFastDivide) && __builtin_amdgcn_ballot_w64(!divisors_ordinary) == 0ull;
    const float distance_opacity = shared ? (flat_normals ? sphere_light_opacity<true, true>(d3, distance, P.normal, L, env.ZToY.z)
                                                          : sphere_light_opacity<true>(d3, distance, P.normal, L, env.ZToY.z))
                                          : sphere_light_opacity<false>(d3, distance, P.normal, L, env.ZToY.z);
    const bool visible = (distance_opacity > 0.0f) && (P.shaded.x > -9999.0f);
    if (!visible)
        return false;

    // computeAO, AOCommon.fxh:1-19 (aoRadius scaled by max(0, normal.z), SphereLightCore.fxh:78)
    float ao_opacity = 1.0f;
    const float ao_radius = L.ao_radius * fmaxf(0.0f, P.normal.z);
    if ((ao_radius >= 0.5f) && have_sdf) {
        const float distance = sample_distance_field<FMT>(mk3(P.shaded.x, P.shaded.y, P.shaded.z + P.normal.z * ao_radius), df, sdf);
        if (STATS) st.samples++;
        float r = 1.0f - sat(clampf(distance, 0.0f, ao_radius) / ao_radius);
        r *= r;
        r = 1.0f - r;
        ao_opacity = (1.0f - L.ao_opacity) + (r * L.ao_opacity);
    }
    const float pre_trace = distance_opacity * ao_opacity;

    // coneTrace, ConeTrace.fxh:148-191
    float cone_opacity = 1.0f;
    const bool trace = (casts != 0.0f) && (pre_trace >= ref::kShadowOpacityThreshold);
    if (trace) {
        if (STATS) st.traced++;
        f3 start = P.shaded + (P.normal * ref::kSelfOcclusionHack);
        const f3 tv = mk3(L.cx, L.cy, L.cz) - start;
        const float trace_length = len3(tv);
        const float data_y = fmaxf(trace_length - L.radius, 1.0f);
        float data_x = ref::kTraceInitialOffsetPx;
        float data_z = 1.0f;
        const float cfg_z = fmaxf(1.0f, df.Packed1.w);
        float steps_remaining = df.StepAndMisc2.x;
        const bool alive = have_sdf;
        // The light's cone configuration is read every iteration; with ~104 SGPRs live the compiler re-loads it from memory inside the
        // loop (s_load + s_waitcnt per sample).  Two VGPRs keep it resident.
        float cone_max_radius = L.cfg_x, cone_growth = L.cfg_y;
        asm volatile("" : "+v"(cone_max_radius), "+v"(cone_growth));
        // In-volume loop when, for every tracing lane of the wave, every sample lies in the box of the table sampler: the light's half of
        // the test was decided when its record was prepared (kLightFastTrace), the pixel's half once per pixel (start_inside); what is
        // left per pair is a trace of ordinary length (a divisor of the unscaled division; > 0 also means a finite direction).
        const bool pair_ok = P.start_inside & (trace_length >= 0x1p-60f);
        if ((F.table != nullptr) && (light_flags_i & kLightFastTrace) && __builtin_amdgcn_ballot_w64(!pair_ok) == 0ull) {
            const float y = refined_rcp(trace_length);
            const f3 dir = mk3(div_with_rcp(tv.x, trace_length, y), div_with_rcp(tv.y, trace_length, y), div_with_rcp(tv.z, trace_length, y));
            cone_trace_loop<FMT, STATS, true>(start, dir, data_y, cfg_z, cone_growth, cone_max_radius, F, data_x, data_z, steps_remaining, alive, st);
        } else {
            f3 dir = mk3(tv.x / trace_length, tv.y / trace_length, tv.z / trace_length);
            // The sampler treats a NaN coordinate as 0.  start + dir * x is NaN for every x exactly when start or dir is (x stays
            // finite), so the test is hoisted: such an axis becomes start = dir = 0 and the loop samples with CHECK_NAN = false.
            if ((start.x != start.x) || (dir.x != dir.x)) { start.x = 0.0f; dir.x = 0.0f; }
            if ((start.y != start.y) || (dir.y != dir.y)) { start.y = 0.0f; dir.y = 0.0f; }
            if ((start.z != start.z) || (dir.z != dir.z)) { start.z = 0.0f; dir.z = 0.0f; }
            cone_trace_loop<FMT, STATS, false>(start, dir, data_y, cfg_z, cone_growth, cone_max_radius, F, data_x, data_z, steps_remaining, alive, st);
        }
        const float visibility = fminf(data_z, steps_remaining / ref::kMaxStepRampWindow);
        // (the numerator is a saturate: always inside the unscaled division's range; the divisor is a constant)
        cone_opacity = pow_pos(sat(div_with_rcp(sat(visibility - ref::kFullyShadowedThreshold), kVisibilityRange, kVisibilityRangeRcp)), df.ConeAndMisc.z);
    }
    // SphereLightPixelEpilogue / ...WithRamp (SphereLightCore.fxh:83-119): with a ramp texture the opacity becomes a colour,
    // SampleFromRamp2(preTraceOpacity, (angle + rampOffset) * rampRate).rgb * coneOpacity -- tex2Dlod level 0, LINEAR, U CLAMP, V WRAP
    float opacity_r = pre_trace * cone_opacity, opacity_g = opacity_r, opacity_b = opacity_r;
    if (ramp.texels != nullptr) {
        const float angle = atan2f(P.shaded.y - L.cy, P.shaded.x - L.cx);
        const float u = pre_trace, v = (angle + L.ramp_offset) * L.ramp_rate;
        const int w = ramp.width, h = ramp.height;
        const float sx = u * (float)w - 0.5f, sy = v * (float)h - 0.5f;
        float x0f = floorf(sx);
        const float y0f = floorf(sy);
        const float fx = sx - x0f, fy = sy - y0f;
        float x1f = x0f + 1.0f;
        x0f = (x0f >= 0.0f) ? x0f : 0.0f; x0f = fminf(x0f, (float)(w - 1));
        x1f = (x1f >= 0.0f) ? x1f : 0.0f; x1f = fminf(x1f, (float)(w - 1));
        const int x0 = (int)x0f, x1 = (int)x1f;
        const int y0 = wrap_index(y0f, h), y1 = wrap_index(y0f + 1.0f, h);
        const float4 t00 = ramp.texels[y0 * w + x0], t10 = ramp.texels[y0 * w + x1], t01 = ramp.texels[y1 * w + x0], t11 = ramp.texels[y1 * w + x1];
        opacity_r = lerp(lerp(t00.x, t10.x, fx), lerp(t01.x, t11.x, fx), fy) * cone_opacity;
        opacity_g = lerp(lerp(t00.y, t10.y, fx), lerp(t01.y, t11.y, fx), fy) * cone_opacity;
        opacity_b = lerp(lerp(t00.z, t10.z, fx), lerp(t01.z, t11.z, fx), fy) * cone_opacity;
    }

    // SphereLightPixelShader epilogue, SphereLight.fx:37-45.  The specular term is
    // skipped when Color2.rgb == 0: it then contributes exactly 0 unless
    // pow() produced inf/NaN (negative SpecularPower), which the reference does not guard.
    float sr = 0.0f, sg = 0.0f, sb = 0.0f;
    if (light_flags_i & kLightHasSpecular) {
        const f3 light_direction = P.shaded - mk3(L.cx, L.cy, L.cz);
        const int l = (int)(threadIdx.x & 63u);
        const f3 camera = mk3((float)(P.origin_x + (l & 7)) / env.ZAndScale.z, (float)(P.origin_y + (l >> 3)) / env.ZAndScale.w, env.ZAndScale.y + 0.01f);
        const f3 h = norm3(norm3(camera - P.shaded) - light_direction);
        const float specularity = pow_pos(sat(dot3(h, P.normal)), L.spec_power);
        sr = L.spec_r * specularity * opacity_r;
        sg = L.spec_g * specularity * opacity_g;
        sb = L.spec_b * specularity * opacity_b;
    }
    out_r = (L.col_r * opacity_r) + sr;
    out_g = (L.col_g * opacity_g) + sg;
    out_b = (L.col_b * opacity_b) + sb;
    return true;
}

constexpr int kTile = kLightTile;                   // internal.hpp
constexpr int kLightThreads = kLightTileThreads;
constexpr int kListCapacity = (kTile == 16) ? 1024 : 256;

// Eight waves per SIMD (64 VGPRs; the fp16 kernel without scratch, the unorm16 one with 8 bytes outside the loop).  History, tools/ab_lib.sh:
// r02, table-driven sampler: five waves (81 VGPRs, the allocator's own need) cfg5 11.93 ms, six 11.26, seven 10.89, eight (48 bytes of
// scratch) 10.93.  r03, cell array + exact skips + tile groups: six 9.29, seven 8.82, eight 8.67 on cfg5 but 0.61 -> 0.66 on cfg3 (spills
// in the per-pair code).  With the launch descriptor read where it is needed (the light loop below) the spills are gone:
// seven 0.616 | 8.93, **eight 0.602 | 8.54**.
#ifndef ILM_LIGHT_WAVES
#define ILM_LIGHT_WAVES 8
#endif
// the wave-level skip of uncovered list entries (sphere_lights_kernel's walk): 1 = the wide-binning instantiations, 2 = all, 0 = none
#ifndef ILM_WAVE_SKIP
#define ILM_WAVE_SKIP 1
#endif
#ifndef ILM_LIGHT_SGPRS
#define ILM_LIGHT_SGPRS
#endif
// circle cull (sphere_lights_kernel): the particle-light instantiation (WIDE) and the sphere-light ones, switchable for A/B builds
#ifndef ILM_LIGHT_CIRCLE_CULL_WIDE
#define ILM_LIGHT_CIRCLE_CULL_WIDE 1
#endif
#ifndef ILM_LIGHT_CIRCLE_CULL
#define ILM_LIGHT_CIRCLE_CULL 0
#endif
#if ILM_LIGHT_WAVES > 0
#define ILM_LIGHT_OCCUPANCY __attribute__((amdgpu_waves_per_eu(ILM_LIGHT_WAVES, ILM_LIGHT_WAVES))) ILM_LIGHT_SGPRS
#else
#define ILM_LIGHT_OCCUPANCY
#endif
#ifdef ILM_LIGHT_TRACE     // EXPERIMENT (tools/light_trace_probe.py): per-wave start / end of the last launch (100 MHz clock), tile and XCC / CU / SIMD
__device__ unsigned long long g_light_trace[4 * 262144];
extern "C" int ilm_experiment_light_trace(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_light_trace), sizeof(unsigned long long) * (size_t)n);
}
#endif
// WIDE_BIN: the tile list is built by all the workgroup's waves, 256 lights per round (frames of particle lights: thousands per tile);
// otherwise by wave 0 alone, 64 per round, while the others wait -- a frame of a few hundred lights bins in 1-4 rounds either way, and
// the wide form's extra code costs the sphere-light frames 1-2 % through register allocation, so it is its own instantiation.
template <int FMT, bool STATS, bool WIDE_BIN>
__global__ __launch_bounds__(kLightThreads) ILM_LIGHT_OCCUPANCY void sphere_lights_kernel(const LightLaunch a, const LightRec* __restrict__ recs, int tiles_x, int tiles_y, int tile_count) {
    __shared__ uint16_t list[kListCapacity];
    __shared__ int list_count;
    __shared__ int bin_count[2][kLightThreads / 64];
    __shared__ SliceEntry slice_table[kMaxTableSlices];
    __shared__ float4 tree[3][kLightThreads];       // the part sums waiting for their right-hand neighbours (levels 0-2 of the tree over 8 parts)
    __shared__ float4 wave_box[kLightThreads / 64];  // circle cull: (min x, max x, min y, max y) of the shaded points of each wave's 8 x 8 pixels

    // The dispatcher places block b on XCD b % 8.  Which tiles an XCD gets decides both its L2 locality and its share of the work
    // (lights are not spread evenly): see light_tile_map() in api.hip for the measurements; groups of 6 x 6 tiles (tile_map 4) are the default.
#ifdef ILM_LIGHT_TRACE
    const unsigned long long trace_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    // Light split: consecutive blocks of an XCD serve one tile (they share its L2: cells, light records, the tile's partial sums);
    // b is the tile's block number as the maps below see it, `member` which of the tile's workgroups this is.
    // (the launch descriptor through a pointer the compiler cannot see through -- see the light loop: what is read here is not kept
    // in scalar registers across the kernel)
    typedef const LightLaunch __attribute__((address_space(4))) CLightLaunch;
    auto kernargs = []() -> const LightLaunch& {
        CLightLaunch* ap = (CLightLaunch*)__builtin_amdgcn_kernarg_segment_ptr();      // the descriptor is the kernel's FIRST parameter: offset 0
        asm volatile("" : "+s"(ap));
        return *(const LightLaunch*)ap;
    };
    // Which of an XCD's block slots this is, and what it serves: slots [0, taper[0]) are whole tiles, the next 2 (taper[1] - taper[0]) the
    // two members of a tile each, then four, then eight (tile_split): the launch ends in its smallest pieces.
    auto tile_split = [&](int& slot, int& member) -> int {
        const LightLaunch& A = kernargs();
        const int k = (int)blockIdx.x / 8;
        const int t0 = A.taper[0], t1 = A.taper[1], t2 = A.taper[2];
        const int b1 = t0 + 2 * (t1 - t0), b2 = b1 + 4 * (t2 - t1);
        if (k < t0) { slot = k; member = 0; return 1; }
        if (k < b1) { slot = t0 + (k - t0) / 2; member = (k - t0) % 2; return 2; }
        if (k < b2) { slot = t1 + (k - b1) / 4; member = (k - b1) % 4; return 4; }
        slot = t2 + (k - b2) / 8; member = (k - b2) % 8; return 8;
    };
    int member, nb, b;
    {
        int slot;
        (void)tile_split(slot, member);
        nb = a.taper_slots * 8;
        b = slot * 8 + ((int)blockIdx.x % 8);
    }
    const int per_xcd = nb / 8;
    int tile;
    if (a.tile_map == 1) {
        // tile rows dealt round-robin to the XCDs: row r -> XCD r % 8 (balances a frame whose lights cluster vertically)
        const int xcd = b % 8, k = b / 8;                  // k-th block of this XCD
        const int rows_here = (tiles_y - xcd + 7) / 8;     // rows r with r % 8 == xcd
        const int r_local = k / tiles_x, cx_ = k - r_local * tiles_x;
        tile = (r_local < rows_here) ? (r_local * 8 + xcd) * tiles_x + cx_ : tile_count;
    } else if (a.tile_map == 2) {
        tile = b;
    } else if (a.tile_map == 4) {
        // square groups of M x M tiles, dealt round-robin to the XCDs; an XCD walks its groups one after the other, so the tiles it
        // runs side by side lie side by side (api.hip light_tile_map has the measurements)
        const int M = a.tile_macro, MM = M * M;
        const int xcd = b % 8, k = b / 8;
        int g = (k / MM) * 8 + xcd;
        const int t = k % MM;
        const int mx = (tiles_x + M - 1) / M;
        if (a.group_order != nullptr && g < mx * ((tiles_y + M - 1) / M)) g = (int)a.group_order[g];
        const int ty = (g / mx) * M + t / M, tx = (g % mx) * M + t % M;
        tile = (tx < tiles_x && ty < tiles_y) ? ty * tiles_x + tx : tile_count;
    } else {
        tile = (b % 8) * per_xcd + (b / 8);
    }
    if (tile >= tile_count)
        return;
    // the in-volume sampler's per-slice table (hlsl_math.hpp): one entry per virtual slice, visible to the tile's waves after the
    // first barrier of the light loop below
#ifdef ILM_EXP_NO_TABLE            // EXPERIMENT (timing of the prologue only)
    const int table_n = 0;
#else
    const int table_n = a.sdf.table_slices;
#endif
    for (int i = (int)threadIdx.x; i < table_n; i += kLightThreads) slice_table[i] = make_slice_entry((uint32_t)i, a.df, a.sdf);
    const InsideConsts inside = make_inside_consts(a.df, a.sdf);
    const int tx0 = (tile % tiles_x) * kTile, ty0 = a.row_begin + (tile / tiles_x) * kTile;

    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const int px = tx0 + (wave & 1) * 8 + (lane & 7);
    const int py = ty0 + (wave >> 1) * 8 + (lane >> 3);
    const bool in_image = (px < a.width) && (py < a.row_end);

    Pixel P = sample_gbuffer((float)px, (float)py, a.env, a.gbuffer);
    P.origin_x = tx0 + (wave & 1) * 8; P.origin_y = ty0 + (wave >> 1) * 8;
    P.start_inside = (table_n > 0) && trace_start_inside(P, a.df, a.sdf);
    const float cxp = (float)px + 0.5f, cyp = (float)py + 0.5f;
    const bool have_sdf = (a.sdf.texels != nullptr) && (a.df.Extent.x > 0.0f);
    // Circle cull (r05).  A light adds nothing to a point whose distance from its centre (y scaled by the light's falloffY) is at least
    // radius + max(ramp, 1): computeSphereLightOpacity's distance factor and its saturate(radius - distance) term are both exactly 0
    // there (LightCommon.fxh:174-214; every falloff mode), the shader discards.  The raster footprint is a square (particle lights) or a
    // cut-corner square (sphere lights) around that circle: 21 % / 13 % of its pixels lie outside it.  Each wave publishes the xy bounding
    // box of the SHADED POINTS of its 8 x 8 pixels (whatever the G-buffer made of them: no assumption about ZToY or relativeY), and the
    // binning marks, per list entry, the waves whose box lies wholly outside the light's circle (4 bits of the 16-bit entry): those waves
    // skip the entry on a scalar test, an entry no wave needs is not listed.  Not in the statistics variant, whose pair count is the
    // raster footprint's (so that variant is also the reference the culled frames are held to, bit for bit).
    constexpr bool kCircleCull = !STATS && (kTile == 16) && (WIDE_BIN ? (ILM_LIGHT_CIRCLE_CULL_WIDE != 0) : (ILM_LIGHT_CIRCLE_CULL != 0));
    if constexpr (kCircleCull) {
        const bool finite_xy = (fabsf(P.shaded.x) <= 0x1p100f) && (fabsf(P.shaded.y) <= 0x1p100f);
        const float inf = __builtin_inff();
        // (a lane outside the image shades nothing: neutral; a lane whose point is not finite opens the box to everything)
        float lo_x = in_image ? (finite_xy ? P.shaded.x : -inf) : inf, hi_x = in_image ? (finite_xy ? P.shaded.x : inf) : -inf;
        float lo_y = in_image ? (finite_xy ? P.shaded.y : -inf) : inf, hi_y = in_image ? (finite_xy ? P.shaded.y : inf) : -inf;
        for (int off = 32; off > 0; off >>= 1) {
            lo_x = fminf(lo_x, __shfl_xor(lo_x, off)); hi_x = fmaxf(hi_x, __shfl_xor(hi_x, off));
            lo_y = fminf(lo_y, __shfl_xor(lo_y, off)); hi_y = fmaxf(hi_y, __shfl_xor(hi_y, off));
        }
        if (lane == 0) wave_box[wave] = mk4(lo_x, hi_x, lo_y, hi_y);      // read after the first barrier of the light loop
    }
    // the waves of the workgroup whose box lies wholly outside the circle of light R (bit w = wave w), by whoever bins R
    auto culled_waves = [&](const LightRec& R) -> int {
        if constexpr (!kCircleCull) return 0;
        const float reach = R.radius + fmaxf(R.ramp, 1.0f), fy = fabsf(R.falloff_y);
        // (a ramp that is not positive never fades; NaNs fail every comparison below: no cull)
        const bool cullable = (R.ramp > 0.0f) & (R.radius >= 0.0f) & (reach <= 0x1p60f) & (fy <= 0x1p60f);
        const float limit = reach * 1.0001f + 0.01f, limit2 = limit * limit;
        int mask = 0;
#pragma unroll 1      // (one box at a time: the pass runs with the pixel's whole state live, 64 registers)
        for (int w = 0; w < kLightThreads / 64; w++) {
            const float4 b = wave_box[w];
            const float dx = fmaxf(fmaxf(b.x - R.cx, R.cx - b.y), 0.0f), dy = fmaxf(fmaxf(b.z - R.cy, R.cy - b.w), 0.0f) * fy;
            if (cullable && (dx * dx + dy * dy >= limit2)) mask |= 1 << w;
        }
        return mask;
    };
    // Entry layout of the LDS list: the light's index within the batch, the four cull bits above it, bit 15 = the tile lies wholly inside
    // the footprint.  The wide binning with the cull (r06, BIG): a batch holds up to 4 096 lights -- 12 index bits, the cull bits above,
    // no bit 15 (the walk always takes the per-pixel footprint test) -- and ends early when another round of 256 lights might not fit the
    // list: 4 096 particle lights were four batches of 1 024, four lists of ten entries with the tile's waves meeting at the end of each;
    // one list of forty, one meeting: 0.991 -> 0.978 ms per frame, the same bits (tools/particle_lights_ab.py, profiles/r06_lane_queue_ab.txt).
    constexpr bool BIG = WIDE_BIN && kCircleCull;
    constexpr int kCullShift = BIG ? 12 : 10, kIndexMask = (1 << kCullShift) - 1;
    const int cull_bit = __builtin_amdgcn_readfirstlane(kCullShift + wave);      // (uniform by construction; said so, so that the walk's test is scalar)

    // What the lights are added to: the clear colour, or the lightmap's contents (additive blend onto an earlier pass of the same frame:
    // another light-type render state, LightingRenderer.cs:1100-1169).  Read when it is needed -- at the end, where the tile's sum is
    // added to it; in the fp16-per-light model at the start, where the chain of roundings begins.
    // The reference's lightmap is a HalfVector4 surface blended into by the ROP light after light (LightingRenderer.cs:476-479): in that
    // model the clear colour and every partial sum pass through fp16.  Off by default (fp32 accumulation, one rounding at the store).
    auto through_half = [](float v) { return __half2float(__float2half_rn(v)); };
    auto base_value = [&](const LightLaunch& A) -> float4 {
        float4 v = mk4(A.ambient[0], A.ambient[1], A.ambient[2], A.ambient[3]);
        if (A.accumulate != 0 && in_image) {
            const size_t o = (size_t)py * (size_t)A.width + (size_t)px;
            if (A.format == ILM_LIGHTMAP_FLOAT4) {
                v = reinterpret_cast<const float4*>(A.lightmap)[o];
            } else if (A.format == ILM_LIGHTMAP_HALF4) {
                const uint2 h = reinterpret_cast<const uint2*>(A.lightmap)[o];
                v = mk4(__half2float(__ushort_as_half((unsigned short)(h.x & 0xFFFFu))), __half2float(__ushort_as_half((unsigned short)(h.x >> 16))),
                        __half2float(__ushort_as_half((unsigned short)(h.y & 0xFFFFu))), __half2float(__ushort_as_half((unsigned short)(h.y >> 16))));
            } else {
                const uint32_t c = reinterpret_cast<const uint32_t*>(A.lightmap)[o];
                v = mk4((float)(c & 0xFFu) / 255.0f, (float)((c >> 8) & 0xFFu) / 255.0f, (float)((c >> 16) & 0xFFu) / 255.0f, (float)(c >> 24) / 255.0f);
            }
        }
        if (A.blend_fp16 != 0) v = mk4(through_half(v.x), through_half(v.y), through_half(v.z), through_half(v.w));
        return v;
    };
    const bool blend_fp16 = a.blend_fp16 != 0;
    // the running sum of the current part of the light list (the whole chain in the fp16-per-light model)
    float acc_r = 0.0f, acc_g = 0.0f, acc_b = 0.0f, acc_a = 0.0f;
    if (blend_fp16) { const float4 v = base_value(a); acc_r = v.x; acc_g = v.y; acc_b = v.z; acc_a = v.w; }
    // wave-uniform: every pixel of this wave has a flat normal (no G-buffer, or ground / top-face texels): the normal factor needs one
    // component of the light direction instead of three (sphere_light_opacity<.., FLAT>)
    const bool flat_normals = __builtin_amdgcn_ballot_w64((P.normal.x != 0.0f) | (P.normal.y != 0.0f)) == 0ull;
    LightStats st;
    const int light_count = (a.light_count_ptr != nullptr) ? __builtin_amdgcn_readfirstlane(*a.light_count_ptr) : a.light_count;

    // ---- the order of the sum -------------------------------------------------------------------------------------------------------
    // The reference adds the lights of a pixel in draw order through the ROP (LightingRenderer.cs:1149-1166 cuts the list into draws of
    // 128 instances; SphereLight.fx:42-45 is what each adds).  Here the launch's light list is cut, BY LIGHT INDEX, into kLightParts parts
    // (part p = lights [L p / 8, L (p + 1) / 8)); a pixel's contributions are summed part by part in light order, each part from zero,
    // the part sums are combined as a balanced binary tree ((q0 + q1) + (q2 + q3)) + ((q4 + q5) + (q6 + q7)), and the tree's root is
    // added to the base value.  That order depends on the light list and the pixel alone -- not on the tile grid (strips may start on
    // any row), nor on how many workgroups serve a tile: with `split` = K of them, member m bins and walks the lights of its 8 / K parts
    // only, its subtree's sum goes to memory, and the member that arrives last adds the K subtree sums (the tree's upper levels).
    // (The fp16-per-light model is one chain of roundings in light order: no parts, split 1.)
    int part, part_end;
    {
        int slot_, member_;
        const int parts_each = kLightParts / tile_split(slot_, member_);
        part = member * parts_each; part_end = part + parts_each;
    }
    const int light_lo = blend_fp16 ? 0 : (int)(((long long)light_count * part) / kLightParts);
    const int light_hi = blend_fp16 ? light_count : (int)(((long long)light_count * part_end) / kLightParts);
    const int part_first = part;
    // closes the current part: its sum joins the tree (a sum whose index in the member's range is odd has its left-hand neighbour waiting
    // one level down: left + right, then one level up), and the registers start the next part from zero
    auto close_part = [&]() {
        int r = part - part_first, level = 0;
        while (r & 1) {
            const float4 left = tree[level][threadIdx.x];
            acc_r = left.x + acc_r; acc_g = left.y + acc_g; acc_b = left.z + acc_b; acc_a = left.w + acc_a;
            level++; r >>= 1;
        }
        part++;
        if (part < part_end) {
            tree[level][threadIdx.x] = mk4(acc_r, acc_g, acc_b, acc_a);
            acc_r = 0.0f; acc_g = 0.0f; acc_b = 0.0f; acc_a = 0.0f;
        }       // else: the member's subtree is summed, its root stays in the registers
    };
    int part_bound = blend_fp16 ? 0x7FFFFFFF : (int)(((long long)light_count * (part + 1)) / kLightParts);   // first light of the next part

    for (int batch = light_lo, batch_step = kListCapacity; batch < light_hi; batch += batch_step) {
#ifdef ILM_EXP_NO_BIN              // EXPERIMENT (timing of the prologue only): no light is looked at
        int batch_n = 0;
#else
        int batch_n = min(BIG ? 4096 : kListCapacity, light_hi - batch);
#endif
        if constexpr (WIDE_BIN) {
        __syncthreads();                                        // the previous batch's list has been walked
        {
            // ordered compaction of the lights whose footprint bounding box touches the tile: all the workgroup's threads test one light
            // each per round (a frame of particle lights bins thousands per tile), the waves' counts meet in LDS, the list keeps light order
            const float tminx = (float)tx0 + 0.5f, tmaxx = (float)(tx0 + kTile - 1) + 0.5f;
            const float tminy = (float)ty0 + 0.5f, tmaxy = (float)(ty0 + kTile - 1) + 0.5f;
            constexpr int kWaves = kLightThreads / 64;
            int base = 0, l0 = 0;
            for (; l0 < batch_n && (!BIG || base + kLightThreads <= kListCapacity); l0 += kLightThreads) {
                const int li = l0 + (int)threadIdx.x;
                bool hit = false, whole = false;
                if (li < batch_n) {
                    const LightRec& R = recs[batch + li];
                    const float4 fx = *reinterpret_cast<const float4*>(&R.fx0), fy = *reinterpret_cast<const float4*>(&R.fy0);
                    hit = (fx.x <= tmaxx) && (fx.w > tminx) && (fy.x <= tmaxy) && (fy.w > tminy);
                    // the whole tile inside one of the footprint's two rectangles: every pixel centre passes the per-pixel test below (the
                    // same comparisons, taken on the tile's extreme centres), so the walk skips it for this entry (bit 15)
                    whole = ((tminx >= fx.y) && (tmaxx < fx.z) && (tminy >= fy.x) && (tmaxy < fy.w)) ||
                            ((tminx >= fx.x) && (tmaxx < fx.w) && (tminy >= fy.y) && (tmaxy < fy.z));
                }
                const unsigned long long m = __ballot(hit);
                const int round = (l0 / kLightThreads) & 1;     // two sets of counters: a wave may be a round ahead of the slowest reader
                if (lane == 0) bin_count[round][wave] = __popcll(m);
                __syncthreads();
                int before = 0, total = 0;
#pragma unroll
                for (int w = 0; w < kWaves; w++) { const int c = bin_count[round][w]; total += c; if (w < wave) before += c; }
                if (hit)
                    list[base + before + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)(li | ((whole && !BIG) ? 0x8000 : 0));
                base += total;
            }
            // (the batch ends where the binning stopped: `base` is the same in every thread, so is l0)
            if constexpr (BIG) { batch_n = min(batch_n, l0); batch_step = (batch_n > 0) ? batch_n : 4096; }
            if (threadIdx.x == 0) list_count = base;
        }
        __syncthreads();
        } else {
        __syncthreads();
        if (threadIdx.x == 0) list_count = 0;
        __syncthreads();
        if (wave == 0) {
            // ordered compaction of the lights whose footprint bounding box touches the tile
            const float tminx = (float)tx0 + 0.5f, tmaxx = (float)(tx0 + kTile - 1) + 0.5f;
            const float tminy = (float)ty0 + 0.5f, tmaxy = (float)(ty0 + kTile - 1) + 0.5f;
            int base = 0;
            for (int l0 = 0; l0 < batch_n; l0 += 64) {
                const int li = l0 + lane;
                bool hit = false, whole = false;
                if (li < batch_n) {
                    const LightRec& R = recs[batch + li];
                    hit = (R.fx0 <= tmaxx) && (R.fx3 > tminx) && (R.fy0 <= tmaxy) && (R.fy3 > tminy);
                    // the whole tile inside one of the footprint's two rectangles: every pixel centre passes the per-pixel test below (the
                    // same comparisons, taken on the tile's extreme centres), so the walk skips it for this entry (bit 15)
                    whole = ((tminx >= R.fx1) && (tmaxx < R.fx2) && (tminy >= R.fy0) && (tmaxy < R.fy3)) ||
                            ((tminx >= R.fx0) && (tmaxx < R.fx3) && (tminy >= R.fy1) && (tmaxy < R.fy2));
                }
                const unsigned long long m = __ballot(hit);
                if (hit)
                    list[base + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)(li | (whole ? 0x8000 : 0));
                base += __popcll(m);
            }
            if (lane == 0) list_count = base;
        }
        __syncthreads();
        }

        const int n = list_count;
        if constexpr (kCircleCull) {
            // the cull bits of the LISTED lights only (the binning looks at every light of the launch -- thousands of particle lights per
            // tile, of which a few dozen are listed: computing the bits there cost more than the skipped entries gave back)
            for (int t = (int)threadIdx.x; t < n; t += kLightThreads) {
                const int e = (int)list[t];
                list[t] = (uint16_t)(e | (culled_waves(recs[batch + (e & kIndexMask)]) << kCullShift));
            }
            __syncthreads();
        }

        for (int k = 0; k < n; k++) {
            // The launch descriptor lives in the kernarg segment.  Read through a pointer the compiler cannot see through, its fields are
            // fetched (scalar-cache hits) where a pair needs them instead of being hoisted out of the light loop and held in SGPRs across
            // the whole kernel: 32-42 scalar registers were spilled to vector lanes that way, now 2 -- by itself worth nothing (cfg5
            // 8.82 -> 8.93 ms), but the vector registers it frees are what lets EIGHT waves per SIMD run without spilling in the loop.
            // (the descriptor is the kernel's FIRST parameter: offset 0 of the segment)
            const LightLaunch& A = kernargs();
            const TraceField field = { A.df, A.sdf, inside, (table_n > 0) ? slice_table : nullptr };
            const IlmEnvironment& env_k = A.env;
            const RampView& ramp_k = A.ramp;
            const int entry = __builtin_amdgcn_readfirstlane((int)list[k]);
            const int li = entry & (kCircleCull ? kIndexMask : 0x7FFF);
            if (kCircleCull && ((entry >> cull_bit) & 1))      // this wave's points are outside the light's circle (see culled_waves): a scalar test
                continue;
            while (batch + li >= part_bound) {      // (never in the fp16-per-light model)
                close_part();
                part_bound = (int)(((long long)light_count * (part + 1)) / kLightParts);
            }
            // (r05: fetching the whole 128-byte record at once for the wide-binning walk -- two 64-byte scalar loads, one wait instead of
            // seven -- was measured a loss like r04's hot / cold record on cfg3: particle lights 1.035 / 1.055 -> 1.071 / 1.088 ms; the 32
            // scalar registers it holds spill 40 more to vector lanes)
            const LightRec& L = recs[batch + li];

            bool covered = in_image;
            if (BIG || (entry & 0x8000) == 0) {
                // raster footprint: pixel centre inside the cross-shaped quad
                // (all eight bounds fetched together and combined without short-circuits: as written with && / || the compiler issued
                // eight dependent scalar loads, each behind its own wait and branch)
                const float fx0 = L.fx0, fx1 = L.fx1, fx2 = L.fx2, fx3 = L.fx3, fy0 = L.fy0, fy1 = L.fy1, fy2 = L.fy2, fy3 = L.fy3;
                const bool tall = (cxp >= fx1) & (cxp < fx2) & (cyp >= fy0) & (cyp < fy3);
                const bool wide = (cxp >= fx0) & (cxp < fx3) & (cyp >= fy1) & (cyp < fy2);
                covered = in_image & (tall | wide);
            }
            // (r06) A wave NONE of whose lanes the entry covers leaves on a scalar branch.  Without it such a wave runs the head of the
            // pair code masked off -- the compiler only branches around blocks it finds long enough -- ~31 vector instructions per
            // list entry on the particle-light frame (the tile lists a light whose footprint misses this quadrant, and the cull's box
            // test did not catch it): 468 M -> 429 M vector instructions per launch, **0.979 -> 0.931 ms per frame**
            // (profiles/r06_particle_lights_wave_skip_ab.txt).  On the sphere-light instantiations (ILM_WAVE_SKIP=2) it changes nothing
            // -- cfg3 0.625 / 0.625 -> 0.625 / 0.622, cfg5 8.691 / 8.695 -> 8.701 / 8.704 ms -- their lists are short and their lights
            // large: the branch is paid by every entry and almost never taken.
            if ((WIDE_BIN ? (ILM_WAVE_SKIP >= 1) : (ILM_WAVE_SKIP >= 2)) && __builtin_amdgcn_ballot_w64(covered) == 0ull)
                continue;
            if (!covered)
                continue;
            if (STATS) st.pairs++;
            float cr, cg, cb;
            if (!shade_light<FMT, STATS>(P, L, env_k, field, have_sdf, ramp_k, st, cr, cg, cb, flat_normals))
                continue;
            if (blend_fp16) {      // dst = half(float(dst) + float(half(src))): the shader's output is converted to the target format, then blended
                acc_r = through_half(acc_r + through_half(cr));
                acc_g = through_half(acc_g + through_half(cg));
                acc_b = through_half(acc_b + through_half(cb));
                acc_a = through_half(acc_a + 1.0f);
            } else {
                acc_r += cr;
                acc_g += cg;
                acc_b += cb;
                acc_a += 1.0f;
            }
        }
    }
    if (!blend_fp16) {
        while (part < part_end) close_part();       // the parts behind the last listed light (empty ones add their zeros: the same additions whatever the split)
    }

    int slot_end, member_end;
    const int split = tile_split(slot_end, member_end);
    if (split > 1) {
        // The members of a tile meet at its tickets, wave by wave: the waves of a workgroup own a quadrant each, so quadrant q of member m
        // only ever needs quadrant q of the other members -- no barrier, every wave leaves when ITS part is delivered.  A wave writes its
        // subtree's sums (device-scope stores, sc1: written through this XCD's L2), waits until the write is acknowledged, draws the
        // quadrant's ticket (device-scope atomic); the one that draws the last reads the K sums back with device-scope loads -- no cache
        // write-back or invalidate is involved, the light pass's L2 contents (the field's cells) stay where they are.
#ifdef ILM_SPLIT_NO_COMBINE      // EXPERIMENT (timing only, wrong pixels): what the members cost without their meeting
        goto tile_done;
#endif
        // (scratch is indexed by the tile's place among the SPLIT block slots of the launch -- slot - taper[0] of this XCD -- not by the tile:
        // the untapered head of a launch needs none, api.hip plan_light_split sizes it so)
        const size_t split_slot = (size_t)(slot_end - kernargs().taper[0]) * 8u + (size_t)((int)blockIdx.x % 8);
        float4* tile_sums = kernargs().partials + split_slot * (size_t)kLightParts * (size_t)kLightThreads + threadIdx.x;
        store_partial(tile_sums + (size_t)member_end * (size_t)kLightThreads, acc_r, acc_g, acc_b, acc_a);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t drawn = 0;
        uint32_t* ticket = kernargs().tickets + (size_t)tile * (size_t)(kLightThreads / 64) + (size_t)wave;
        if (lane == 0) drawn = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        drawn = (uint32_t)__builtin_amdgcn_readfirstlane((int)drawn);
        if (drawn != (uint32_t)(split - 1))
            goto tile_done;
        if (lane == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // ready for the next launch
        asm volatile("" ::: "memory");
        const float4 v = combine_partials(tile_sums, (size_t)kLightThreads, split);
        acc_r = v.x; acc_g = v.y; acc_b = v.z; acc_a = v.w;
    }
    if (!blend_fp16) {
        const float4 v = base_value(kernargs());
        acc_r = v.x + acc_r; acc_g = v.y + acc_g; acc_b = v.z + acc_b; acc_a = v.w + acc_a;
    }

    if (in_image) {
        const size_t o = (size_t)py * (size_t)a.width + (size_t)px;
        // the texel, once; then wherever it goes: the lightmap, and in store mode the same offset of every other member's copy of the
        // frame (peer-mapped; 8 x 8 pixels of a wave are eight 64-byte runs of half4 texels)
        const float4 v32 = mk4(acc_r, acc_g, acc_b, acc_a);
        uint2 v16 = make_uint2(0u, 0u);
        uint32_t v8 = 0u;
        if (a.format == ILM_LIGHTMAP_HALF4) {
            v16.x = (uint32_t)__half_as_ushort(__float2half_rn(acc_r)) | ((uint32_t)__half_as_ushort(__float2half_rn(acc_g)) << 16);
            v16.y = (uint32_t)__half_as_ushort(__float2half_rn(acc_b)) | ((uint32_t)__half_as_ushort(__float2half_rn(acc_a)) << 16);
        } else if (a.format != ILM_LIGHTMAP_FLOAT4) {
            const uint32_t r = (uint32_t)rintf(sat(acc_r) * 255.0f), g = (uint32_t)rintf(sat(acc_g) * 255.0f);
            const uint32_t bl = (uint32_t)rintf(sat(acc_b) * 255.0f), al = (uint32_t)rintf(sat(acc_a) * 255.0f);
            v8 = r | (g << 8) | (bl << 16) | (al << 24);
        }
        auto store_texel = [&](void* base) {
            if (a.format == ILM_LIGHTMAP_FLOAT4) reinterpret_cast<float4*>(base)[o] = v32;
            else if (a.format == ILM_LIGHTMAP_HALF4) reinterpret_cast<uint2*>(base)[o] = v16;
            else reinterpret_cast<uint32_t*>(base)[o] = v8;
        };
        store_texel(a.lightmap);
        const LightLaunch& A = kernargs();
        const int mirror_count = A.mirror_count;
        for (int m = 0; m < mirror_count; m++) store_texel(A.mirrors[m]);
    }

tile_done:
#ifdef ILM_LIGHT_TRACE
    if (lane == 0) {
        const unsigned w = ((unsigned)blockIdx.x * (unsigned)(kLightThreads / 64) + (unsigned)wave) & 262143u;
        unsigned hw_id, xcc_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
        g_light_trace[4 * w] = trace_t0; g_light_trace[4 * w + 1] = __builtin_amdgcn_s_memrealtime();
        g_light_trace[4 * w + 2] = (unsigned long long)tile; g_light_trace[4 * w + 3] = ((unsigned long long)xcc_id << 32) | hw_id;
    }
#endif
    if (STATS) {
        // wave reduce, one atomic per wave and counter
        for (int off = 32; off > 0; off >>= 1) {
            st.samples += __shfl_down(st.samples, off);
            st.pairs += __shfl_down(st.pairs, off);
            st.traced += __shfl_down(st.traced, off);
        }
        if (lane == 0) {
            atomicAdd(&a.stats[0], st.samples);
            atomicAdd(&a.stats[1], st.pairs);
            atomicAdd(&a.stats[2], st.traced);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Particle lights -- ParticleLightVertexShader, ParticleLight.fx:16-83, for every slot below the chunk's quad count:
// a light record is emitted iff life > 0 and the un-premultiplied render colour x LightColor has alpha > 0.
// Two passes over 1024-slot blocks: counts, then an ordered emit (block base = sum of the counts before it; inside a
// block: wave64 ballot + popcount prefix, wave bases through LDS), so the records come out in chunk / slot order --
// the order the reference's instanced draw blends them in.
// ---------------------------------------------------------------------------------------------
constexpr int kPlBlock = 1024;

ILM_DEV bool particle_emits_light(const ParticleLightLaunch& a, int chunk, int slot, float4& position, float4& light_color) {
    const int quads = (a.quad_counts != nullptr) ? a.quad_counts[chunk] : a.slots;
    if (slot >= quads || slot >= a.slots)
        return false;
    const float* base = a.chunk_bases[chunk];
    const int64_t S = a.stride;
    position = mk4(base[slot], base[S + slot], base[2 * S + slot], base[3 * S + slot]);
    float4 rc = mk4(base[12 * S + slot], base[13 * S + slot], base[14 * S + slot], base[15 * S + slot]);   // Chunk.RenderColor
    if (rc.w > 0.0f) {   // unpremultiply, :43-45
        rc.x /= rc.w; rc.y /= rc.w; rc.z /= rc.w;
    }
    if (position.w <= 0.0f)
        return false;
    light_color = mul4(rc, ld4(a.params.LightColor));
    return light_color.w > 0.0f;
}

__global__ __launch_bounds__(kPlBlock) void particle_light_count_kernel(const ParticleLightLaunch a, int blocks_per_chunk) {
    __shared__ int wave_counts[kPlBlock / 64];
    const int chunk = (int)blockIdx.x / blocks_per_chunk, blk = (int)blockIdx.x - chunk * blocks_per_chunk;
    const int slot = blk * kPlBlock + (int)threadIdx.x;
    float4 pos, col;
    const bool emit = particle_emits_light(a, chunk, slot, pos, col);
    const unsigned long long m = __ballot(emit);
    if ((threadIdx.x & 63u) == 0u) wave_counts[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int n = 0;
        for (int w = 0; w < kPlBlock / 64; w++) n += wave_counts[w];
        a.block_counts[blockIdx.x] = n;
    }
}

__global__ __launch_bounds__(kPlBlock) void particle_light_emit_kernel(const ParticleLightLaunch a, int blocks_per_chunk) {
    __shared__ int wave_counts[kPlBlock / 64];
    __shared__ int partial[kPlBlock / 64];
    __shared__ int block_base;
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    // exclusive prefix of the block counts before this block (a few thousand ints at most)
    int sum = 0;
    for (int i = (int)threadIdx.x; i < (int)blockIdx.x; i += kPlBlock) sum += a.block_counts[i];
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    if (lane == 0) partial[wave] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        int b = 0;
        for (int w = 0; w < kPlBlock / 64; w++) b += partial[w];
        block_base = b;
        if (blockIdx.x == gridDim.x - 1)
            *a.out_count = min(b + a.block_counts[blockIdx.x], a.capacity);
    }
    const int chunk = (int)blockIdx.x / blocks_per_chunk, blk = (int)blockIdx.x - chunk * blocks_per_chunk;
    const int slot = blk * kPlBlock + (int)threadIdx.x;
    float4 pos, col;
    const bool emit = particle_emits_light(a, chunk, slot, pos, col);
    const unsigned long long m = __ballot(emit);
    if (lane == 0) wave_counts[wave] = __popcll(m);
    __syncthreads();
    if (!emit)
        return;
    int index = block_base + __popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; w++) index += wave_counts[w];
    if (index >= a.capacity)
        return;
    const IlmParticleLightParams& P = a.params;
    LightRec r;
    r.cx = pos.x; r.cy = pos.y; r.cz = pos.z;
    r.radius = P.LightProperties.x; r.ramp = P.LightProperties.y; r.falloff_mode = P.LightProperties.z; r.casts_shadows = P.LightProperties.w;
    r.ao_radius = P.MoreLightProperties.x; r.falloff_y = P.MoreLightProperties.z; r.ao_opacity = P.MoreLightProperties.w;
    r.shadow_filter = -1.0f;                                                   // ParticleLightPixelShader has no shadow filter
    r.col_r = col.x * col.w; r.col_g = col.y * col.w; r.col_b = col.z * col.w; // lightColor.rgb * lightColor.a, :113-116
    r.spec_r = P.LightSpecularColor.x; r.spec_g = P.LightSpecularColor.y; r.spec_b = P.LightSpecularColor.z; r.spec_power = P.LightSpecularColor.w;
    // the quad, :55-70: a plain rectangle (fx1 = fx0, fx2 = fx3 collapse the sphere light's cross shape onto it)
    const float radius = P.LightProperties.x + P.LightProperties.y + 1.0f;
    const float tlx = r.cx - radius, brx = r.cx + radius, bry = r.cy + radius;
    float tly = r.cy - radius;
    tly -= radius * a.env.ZToY.y;
    tly -= r.cz * a.env.ZToY.x;
    const float sx = a.env.GBufferTexelSizeAndMisc.z * a.env.ZAndScale.z, sy = a.env.GBufferTexelSizeAndMisc.w * a.env.ZAndScale.w;
    r.fx0 = r.fx1 = (tlx - a.env.ViewportPosition[0]) * sx;
    r.fx2 = r.fx3 = (brx - a.env.ViewportPosition[0]) * sx;
    r.fy0 = r.fy1 = (tly - a.env.ViewportPosition[1]) * sy;
    r.fy2 = r.fy3 = (bry - a.env.ViewportPosition[1]) * sy;
    const float max_radius = clampf(r.radius, ref::kMinConeRadius, a.max_cone_radius);
    r.cfg_x = max_radius;
    r.cfg_y = max_radius / fmaxf(r.ramp, 16.0f) * 1.0f;
    r.ramp_offset = r.ramp_rate = 0.0f;
    const int flags = light_flags(r, a.gate);
    r.flags = (float)flags;
    r.ramp_rcp = (flags & kLightFastDivide) ? refined_rcp(r.ramp) : 0.0f;
    reinterpret_cast<LightRec*>(a.recs)[index] = r;
}

hipError_t launch_prepare_particle_lights(const ParticleLightLaunch& a, hipStream_t stream) {
    const int blocks_per_chunk = (a.slots + kPlBlock - 1) / kPlBlock;
    const int blocks = a.chunk_count * blocks_per_chunk;
    if (blocks <= 0) return hipMemsetAsync(a.out_count, 0, sizeof(int32_t), stream);
    hipLaunchKernelGGL(particle_light_count_kernel, dim3(blocks), dim3(kPlBlock), 0, stream, a, blocks_per_chunk);
    hipLaunchKernelGGL(particle_light_emit_kernel, dim3(blocks), dim3(kPlBlock), 0, stream, a, blocks_per_chunk);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Light probes -- SphereLightProbePixelShader, SphereLightProbe.fx:19-44: one lane per probe walks every light record
// (uniform index -> scalar loads), fp32 accumulation in light order.
// ---------------------------------------------------------------------------------------------
// r04: one WAVE per (block of 64 probes, light) instead of one lane walking every light (256 probes under 64 lights were four waves
// tracing 64 lights one after the other: 1.2 ms of latency); the pair's contribution goes to scratch [light][probe] and a second kernel
// adds a probe's contributions in light order -- the same additions in the same order.  `pairs` = nullptr: the one-kernel form (scratch
// for probe_count x light_count contributions was not to be had).
template <int FMT>
ILM_DEV bool probe_light(const float4 pp, const float4 pn, LightRec L, const IlmEnvironment& env, const TraceField& field, bool have_sdf, const RampView& ramp,
                         float& cr, float& cg, float& cb) {
    Pixel P;
    P.shaded = xyz(pp); P.normal = xyz(pn); P.origin_x = 0; P.origin_y = 0;      // (no specular term: SphereLightProbe.fx:33-42)
    P.fullbright = false;
    P.start_inside = false;
    // lightProperties.w *= enableShadows (a float here, not the G-buffer's flag); no AO, no specular, no shadow filter (:33-42)
    L.casts_shadows *= pn.w;
    P.enable_shadows = true;
    L.ao_radius = 0.0f; L.ao_opacity = 0.0f;
    L.shadow_filter = -1.0f;
    L.flags = (float)((int)L.flags & kLightFastDivide);     // no specular; probes use the general trace loop (no slice table here)
    LightStats st;
    return shade_light<FMT, false>(P, L, env, field, have_sdf, ramp, st, cr, cg, cb);
}

template <int FMT>
__global__ __launch_bounds__(64) void light_probes_kernel(const LightRec* __restrict__ recs, int light_count,
                                                           const float4* __restrict__ probe_positions, const float4* __restrict__ probe_normals,
                                                           int probe_count, IlmEnvironment env, IlmDistanceFieldUniforms df, SdfView sdf,
                                                           RampView ramp, float4* __restrict__ values, float4* __restrict__ pairs) {
    const int i = (int)blockIdx.x * 64 + (int)threadIdx.x;
    const bool valid = i < probe_count;
    const float4 pp = valid ? probe_positions[i] : mk4(0.0f, 0.0f, 0.0f, 0.0f);
    const float4 pn = valid ? probe_normals[i] : mk4(0.0f, 0.0f, 0.0f, 0.0f);
    // sampleLightProbeBuffer, LightCommon.fxh:233-254
    const float probe_opacity = pp.w;
    const bool have_sdf = (sdf.texels != nullptr) && (df.Extent.x > 0.0f);
    TraceSdfView trace_sdf;                                       // probes are few: the general sampler, no cell array
    static_cast<SdfView&>(trace_sdf) = sdf;
    trace_sdf.cells = nullptr; trace_sdf.cells_bytes = 0; trace_sdf.slice_w = 0; trace_sdf.slice_h = 0;
    const InsideConsts inside = make_inside_consts(df, trace_sdf);
    const TraceField field = { df, trace_sdf, inside, nullptr };
    if (pairs != nullptr) {
        const int k = (int)blockIdx.y;                            // this wave's light
        float cr = 0.0f, cg = 0.0f, cb = 0.0f;
        bool lit = false;
        if (valid && probe_opacity > 0.0f)
            lit = probe_light<FMT>(pp, pn, recs[k], env, field, have_sdf, ramp, cr, cg, cb);
        if (valid)
            pairs[(size_t)k * (size_t)probe_count + (size_t)i] = lit ? mk4(cr * probe_opacity, cg * probe_opacity, cb * probe_opacity, 1.0f) : mk4(0.0f, 0.0f, 0.0f, 0.0f);
        return;
    }
    float acc_r = 0.0f, acc_g = 0.0f, acc_b = 0.0f, acc_a = 0.0f;
    for (int k = 0; k < light_count; k++) {
        const LightRec L = recs[k];
        if (!(valid && probe_opacity > 0.0f))
            continue;
        float cr, cg, cb;
        if (!probe_light<FMT>(pp, pn, L, env, field, have_sdf, ramp, cr, cg, cb))
            continue;
        acc_r += cr * probe_opacity;
        acc_g += cg * probe_opacity;
        acc_b += cb * probe_opacity;
        acc_a += 1.0f;
    }
    if (valid)
        values[i] = mk4(acc_r, acc_g, acc_b, acc_a);
}

// a probe's contributions, added in light order (w = 1: the pair was lit and its products are added; w = 0: nothing is)
__global__ __launch_bounds__(64) void light_probes_sum_kernel(const float4* __restrict__ pairs, int light_count, int probe_count, float4* __restrict__ values) {
    const int i = (int)blockIdx.x * 64 + (int)threadIdx.x;
    if (i >= probe_count) return;
    float acc_r = 0.0f, acc_g = 0.0f, acc_b = 0.0f, acc_a = 0.0f;
    // (eight loads in flight; the additions stay one after the other, in light order)
    for (int k0 = 0; k0 < light_count; k0 += 8) {
        float4 c[8];
#pragma unroll
        for (int u = 0; u < 8; u++) c[u] = pairs[(size_t)min(k0 + u, light_count - 1) * (size_t)probe_count + (size_t)i];
#pragma unroll
        for (int u = 0; u < 8; u++)
            if ((k0 + u < light_count) && (c[u].w != 0.0f)) { acc_r += c[u].x; acc_g += c[u].y; acc_b += c[u].z; acc_a += 1.0f; }
    }
    values[i] = mk4(acc_r, acc_g, acc_b, acc_a);
}

hipError_t launch_light_probes(const void* recs, int light_count, const float4* probe_positions, const float4* probe_normals, int probe_count,
                               const IlmEnvironment& env, const IlmDistanceFieldUniforms& df, const SdfView& sdf, const RampView& ramp, float4* values,
                               float4* pairs, hipStream_t stream) {
    if (probe_count <= 0) return hipSuccess;
    const bool by_pair = (pairs != nullptr) && (light_count > 1) && (light_count <= 65535);
    const dim3 grid((unsigned)((probe_count + 63) / 64), by_pair ? (unsigned)light_count : 1u), block(64);
    const LightRec* r = reinterpret_cast<const LightRec*>(recs);
    float4* p = by_pair ? pairs : nullptr;
    if (sdf.format == ILM_SDF_FP16)
        hipLaunchKernelGGL(light_probes_kernel<ILM_SDF_FP16>, grid, block, 0, stream, r, light_count, probe_positions, probe_normals, probe_count, env, df, sdf, ramp, values, p);
    else
        hipLaunchKernelGGL(light_probes_kernel<ILM_SDF_UNORM16>, grid, block, 0, stream, r, light_count, probe_positions, probe_normals, probe_count, env, df, sdf, ramp, values, p);
    if (by_pair)
        hipLaunchKernelGGL(light_probes_sum_kernel, dim3((unsigned)((probe_count + 63) / 64)), block, 0, stream, p, light_count, probe_count, values);
    return hipGetLastError();
}

template <int FMT>
__global__ __launch_bounds__(256) void sdf_sample_kernel(SdfView sdf, IlmDistanceFieldUniforms df, const float* __restrict__ positions, int count,
                                                          float* __restrict__ out) {
    const int i = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (i >= count) return;
    out[i] = sample_distance_field<FMT>(mk3(positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]), df, sdf);
}

// the cone trace's in-volume sampler on caller-supplied positions (diagnostic entry point ilm_debug_sdf_sample_inside): positions inside
// the sampler's box go through sample_inside_table exactly as the trace loop calls it (used = 1), the others through the general sampler
template <int FMT>
__global__ __launch_bounds__(256) void sdf_sample_inside_kernel(TraceSdfView sdf, IlmDistanceFieldUniforms df, const float* __restrict__ positions, int count,
                                                                 float* __restrict__ out, int32_t* __restrict__ used) {
    __shared__ SliceEntry slice_table[kMaxTableSlices];
    for (int i = (int)threadIdx.x; i < sdf.table_slices; i += 256) slice_table[i] = make_slice_entry((uint32_t)i, df, sdf);
    __syncthreads();
    const InsideConsts inside = make_inside_consts(df, sdf);
    const int i = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (i >= count) return;
    const f3 p = mk3(positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]);
    const bool in_box = (sdf.table_slices > 0) & (p.x >= sdf.box_x0) & (p.x <= sdf.box_x1) & (p.y >= sdf.box_y0) & (p.y <= sdf.box_y1) &
                        (p.z >= sdf.box_z0) & (p.z <= sdf.box_z1);
    used[i] = in_box ? 1 : 0;
    // positions outside the box keep the general sampler's value, which launch_sdf_sample_inside has already written to out[] (both
    // samplers in one kernel make the backend carry the typed loads' scalar operands through a divergent phi: "illegal VGPR to SGPR copy")
    // No divergent branch around the sampler either (its uniform operands -- tap row bases, the typed loads' resource -- are pinned to
    // scalar registers): lanes outside the box sample the box's centre and drop the result.
    const f3 q = in_box ? p : mk3(0.5f * (sdf.box_x0 + sdf.box_x1), 0.5f * (sdf.box_y0 + sdf.box_y1), 0.5f * (sdf.box_z0 + sdf.box_z1));
    const float value = (sdf.table_slices > 0) ? sample_inside_table<FMT>(q, inside, sdf, slice_table) : 0.0f;
    if (in_box) out[i] = value;
}

hipError_t launch_sdf_sample_inside(const TraceSdfView& sdf, const IlmDistanceFieldUniforms& df, const float* positions, int count, float* out, int32_t* used,
                                    hipStream_t stream) {
    if (count <= 0) return hipSuccess;
    const dim3 grid((unsigned)((count + 255) / 256)), block(256);
    const hipError_t general = launch_sdf_sample(sdf, df, positions, count, out, stream);
    if (general != hipSuccess) return general;
    if (sdf.format == ILM_SDF_FP16) hipLaunchKernelGGL(sdf_sample_inside_kernel<ILM_SDF_FP16>, grid, block, 0, stream, sdf, df, positions, count, out, used);
    else hipLaunchKernelGGL(sdf_sample_inside_kernel<ILM_SDF_UNORM16>, grid, block, 0, stream, sdf, df, positions, count, out, used);
    return hipGetLastError();
}

// The cell array of a field (SdfView::cells): cell (v, y, x) = the four taps (x, y), (x + 1, y), (x, y + 1), (x + 1, y + 1) of virtual
// slice v's grid, each as the 32-bit channel pair (slice v, slice v + 1) -- exactly the words sample_distance_field's taps would fetch
// from the atlas for a sample whose floor coordinates are (x, y) in that slice (sdf_pair_word).  The last column / row of a slice
// repeat their neighbour (never sampled: the table sampler's box keeps every tap inside the slice).  One thread per cell, 16-byte stores.
__global__ __launch_bounds__(256) void build_sdf_cells_kernel(const uint2* __restrict__ atlas, int atlas_w, int slice_w, int slice_h, int columns, int first_slice,
                                                               uint4* __restrict__ cells) {
    const int x = (int)blockIdx.x * 256 + (int)threadIdx.x, y = (int)blockIdx.y, v = first_slice + (int)blockIdx.z;
    if (x >= slice_w) return;
    const uint32_t third = (uint32_t)v / 3u, m = (uint32_t)v - 3u * third;
    const int col = (int)(third % (uint32_t)columns), row = (int)(third / (uint32_t)columns);
    const int ax0 = col * slice_w + x, ax1 = col * slice_w + min(x + 1, slice_w - 1);
    const size_t r0 = (size_t)(row * slice_h + y) * (size_t)atlas_w, r1 = (size_t)(row * slice_h + min(y + 1, slice_h - 1)) * (size_t)atlas_w;
    uint4 c;
    c.x = sdf_pair_word(atlas[r0 + ax0], m); c.y = sdf_pair_word(atlas[r0 + ax1], m);
    c.z = sdf_pair_word(atlas[r1 + ax0], m); c.w = sdf_pair_word(atlas[r1 + ax1], m);
    cells[((size_t)v * (size_t)slice_h + (size_t)y) * (size_t)slice_w + (size_t)x] = c;
}

// (re)builds the cells of virtual slices [first_slice, first_slice + slice_count)
hipError_t launch_build_sdf_cells(const TraceSdfView& sdf, void* cells, int first_slice, int slice_count, hipStream_t stream) {
    if (sdf.table_slices <= 0 || slice_count <= 0) return hipSuccess;
    const dim3 grid((unsigned)((sdf.slice_w + 255) / 256), (unsigned)sdf.slice_h, (unsigned)slice_count), block(256);
    hipLaunchKernelGGL(build_sdf_cells_kernel, grid, block, 0, stream, sdf.texels, sdf.width, sdf.slice_w, sdf.slice_h, sdf.columns, first_slice,
                       reinterpret_cast<uint4*>(cells));
    return hipGetLastError();
}

hipError_t launch_sdf_sample(const SdfView& sdf, const IlmDistanceFieldUniforms& df, const float* positions, int count, float* out, hipStream_t stream) {
    if (count <= 0) return hipSuccess;
    const dim3 grid((unsigned)((count + 255) / 256)), block(256);
    if (sdf.format == ILM_SDF_FP16) hipLaunchKernelGGL(sdf_sample_kernel<ILM_SDF_FP16>, grid, block, 0, stream, sdf, df, positions, count, out);
    else hipLaunchKernelGGL(sdf_sample_kernel<ILM_SDF_UNORM16>, grid, block, 0, stream, sdf, df, positions, count, out);
    return hipGetLastError();
}

// device scratch for the prepared light records, owned by the caller (api.hip)
// the light-side half of the in-volume trace test (see light_flags)
TraceGate make_trace_gate(const IlmDistanceFieldUniforms& df, const SdfView& sdf) {
    TraceGate g;
    const float mx = 0.0625f + df.Extent.x * 0x1p-16f, my = 0.0625f + df.Extent.y * 0x1p-16f, mz = 0.0625f + df.Extent.z * 0x1p-16f;
    g.x0 = sdf.box_x0 + mx; g.x1 = sdf.box_x1 - mx;
    g.y0 = sdf.box_y0 + my; g.y1 = sdf.box_y1 - my;
    g.z0 = sdf.box_z0 + mz; g.z1 = sdf.box_z1 - mz;
    const bool ok = (sdf.table_slices > 0) && (df.Extent.w > 0.0f) && (df.Extent.w <= 0x1p20f) && (df.Extent.x <= 0x1p20f) && (df.Extent.y <= 0x1p20f) &&
                    (df.Extent.z <= 0x1p20f) && (df.StepAndMisc2.x >= 0.0f) && (df.StepAndMisc2.x <= 0x1p23f);
    g.field_ok = ok ? 1.0f : 0.0f;
    return g;
}

hipError_t launch_prepare_lights(const IlmLightVertex* lights, int count, const IlmEnvironment& env, const IlmDistanceFieldUniforms& df,
                                 const SdfView& sdf, void* recs, hipStream_t stream) {
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(prepare_lights_kernel, dim3((count + 63) / 64), dim3(64), 0, stream, lights, count, env, df.ConeAndMisc.x, make_trace_gate(df, sdf),
                       reinterpret_cast<LightRec*>(recs));
    return hipGetLastError();
}

// block slots per XCD of a launch over `a`'s rows (each slot = one tile of the block -> tile map, padding included)
int light_block_slots(const LightLaunch& a) {
    const int rows = a.row_end - a.row_begin;
    if (rows <= 0 || a.width <= 0) return 0;
    const int tiles_x = (a.width + kTile - 1) / kTile, tiles_y = (rows + kTile - 1) / kTile;
    const int tile_count = tiles_x * tiles_y;
    int blocks = ((tile_count + 7) / 8) * 8;
    if (a.tile_map == 1) blocks = ((tiles_y + 7) / 8) * tiles_x * 8;   // every XCD gets ceil(rows / 8) rows' worth of blocks
    if (a.tile_map == 4) {
        const int M = a.tile_macro, groups = ((tiles_x + M - 1) / M) * ((tiles_y + M - 1) / M);
        blocks = ((groups + 7) / 8) * 8 * M * M;
    }
    return blocks / 8;
}

// workgroups the tile kernel is launched with for `a` (split and taper as planned; what SQ_WAVES / 4 of the launch counts)
int light_launch_blocks(const LightLaunch& a) {
    const int slots = light_block_slots(a);
    if (slots <= 0) return 0;
    if (a.split <= 1) return slots * 8;
    return (a.taper[0] + 2 * (a.taper[1] - a.taper[0]) + 4 * (a.taper[2] - a.taper[1]) + 8 * (slots - a.taper[2])) * 8;
}

hipError_t launch_sphere_lights_prepared(const LightLaunch& launch, const void* recs, hipStream_t stream) {
    const int rows = launch.row_end - launch.row_begin;
    if (rows <= 0 || launch.width <= 0) return hipSuccess;
    LightLaunch a = launch;
    const int tiles_x = (a.width + kTile - 1) / kTile, tiles_y = (rows + kTile - 1) / kTile;
    const int tile_count = tiles_x * tiles_y;
    const int slots = light_block_slots(a);
    if (a.split <= 1) { a.split = 1; a.taper[0] = a.taper[1] = a.taper[2] = slots; a.taper_slots = slots; }     // one workgroup per tile throughout
    const int t0 = a.taper[0], t1 = a.taper[1], t2 = a.taper[2];
    if (a.taper_slots != slots || t0 < 0 || t0 > t1 || t1 > t2 || t2 > slots) return hipErrorInvalidValue;
    const int per_xcd_blocks = t0 + 2 * (t1 - t0) + 4 * (t2 - t1) + 8 * (slots - t2);
    const LightRec* r = reinterpret_cast<const LightRec*>(recs);
    const bool stats = a.stats != nullptr;
    const bool fp16 = a.sdf.format == ILM_SDF_FP16;
    const dim3 grid(per_xcd_blocks * 8), block(kLightThreads);
#define ILM_LAUNCH_LIGHTS(F, S, W) hipLaunchKernelGGL((sphere_lights_kernel<F, S, W>), grid, block, 0, stream, a, r, tiles_x, tiles_y, tile_count)
    // device-side counts are particle lights: thousands
    const bool wide = (a.light_count_ptr != nullptr) || (a.light_count > 512);
    const int variant = (wide ? 4 : 0) | (fp16 ? 2 : 0) | (stats ? 1 : 0);
    switch (variant) {
        case 0: ILM_LAUNCH_LIGHTS(ILM_SDF_UNORM16, false, false); break;
        case 1: ILM_LAUNCH_LIGHTS(ILM_SDF_UNORM16, true, false); break;
        case 2: ILM_LAUNCH_LIGHTS(ILM_SDF_FP16, false, false); break;
        case 3: ILM_LAUNCH_LIGHTS(ILM_SDF_FP16, true, false); break;
        case 4: ILM_LAUNCH_LIGHTS(ILM_SDF_UNORM16, false, true); break;
        case 5: ILM_LAUNCH_LIGHTS(ILM_SDF_UNORM16, true, true); break;
        case 6: ILM_LAUNCH_LIGHTS(ILM_SDF_FP16, false, true); break;
        default: ILM_LAUNCH_LIGHTS(ILM_SDF_FP16, true, true); break;
    }
#undef ILM_LAUNCH_LIGHTS
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void divide_probe_kernel(const float* __restrict__ n, const float* __restrict__ d, int count,
                                                            float* __restrict__ out_fast, float* __restrict__ out_ieee) {
    const int i = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (i >= count) return;
    out_fast[i] = div_no_scale(n[i], d[i]);
    out_ieee[i] = n[i] / d[i];
}

// every one of the 2^32 float bit patterns as the numerator of a division by one constant: div_with_rcp with the constant's
// host-computed reciprocal against `/` (proof by exhaustion for the two constant divisors of shade_light)
__global__ __launch_bounds__(256) void divide_by_constant_kernel(float divisor, float reciprocal, unsigned long long* __restrict__ mismatches) {
    const uint32_t first = ((uint32_t)blockIdx.x * 256u + (uint32_t)threadIdx.x) << 12;      // 2^20 threads x 4096 numerators
    uint32_t bad_inside = 0, bad_outside = 0;
    for (uint32_t k = 0; k < 4096u; k++) {
        const float n = __uint_as_float(first + k);
        const float a = div_with_rcp(n, divisor, reciprocal), b = n / divisor;
        const bool differ = (__float_as_uint(a) != __float_as_uint(b)) && !((a != a) && (b != b));
        // the range the unscaled division is specified for: 2^-60 <= |n| <= 2^60, or zero / infinite / NaN
        const float m = fabsf(n);
        const bool admitted = !(m < 0x1p-60f && m > 0.0f) && !(m > 0x1p60f && m < __builtin_inff());
        bad_inside += (differ && admitted) ? 1u : 0u;
        bad_outside += (differ && !admitted) ? 1u : 0u;
    }
    if (bad_inside) atomicAdd(&mismatches[0], (unsigned long long)bad_inside);
    if (bad_outside) atomicAdd(&mismatches[1], (unsigned long long)bad_outside);
}
hipError_t launch_divide_by_constant(float divisor, float reciprocal, unsigned long long* mismatches, hipStream_t stream) {
    hipLaunchKernelGGL(divide_by_constant_kernel, dim3(4096), dim3(256), 0, stream, divisor, reciprocal, mismatches);
    return hipGetLastError();
}
// the (divisor, reciprocal) pairs shade_light uses
void light_constant_divisors(float out[2][2]) {
    out[0][0] = ref::kDotRampRange; out[0][1] = kDotRampRangeRcp;
    out[1][0] = kVisibilityRange; out[1][1] = kVisibilityRangeRcp;
}

hipError_t launch_divide_probe(const float* n, const float* d, int count, float* out_fast, float* out_ieee, hipStream_t stream) {
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(divide_probe_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, n, d, count, out_fast, out_ieee);
    return hipGetLastError();
}

}  // namespace ilm
