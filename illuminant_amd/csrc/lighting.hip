// lighting.hip -- LightingRenderer sphere-light pass (SDF cone trace) for gfx950.
//
// The reference draws one instanced quad per light and lets the ROP add the
// results (Illuminant/Lighting/LightingRenderer.cs:1004-1169, technique
// SphereLight in Illuminant/Shaders/SphereLight.fx:7-46): every light re-reads
// the G-buffer and rounds through the lightmap format.  Here a 16x16 pixel tile
// is one workgroup: wave 0 bins the lights whose raster footprint touches the
// tile into an LDS list (wave64 ballot + popcount, light order preserved), then
// every thread decodes its G-buffer texel once and walks the list, accumulating
// in fp32 registers and storing the lightmap texel once.
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
    float col_r, col_g, col_b, has_spec;        // Color1.rgb * Color1.a
    float spec_r, spec_g, spec_b, shadow_falloff;
    float fx0, fx1, fx2, fx3;                   // raster footprint (screen px), see light_covers
    float fy0, fy1, fy2, fy3;
    float cfg_x, cfg_y, _pad0, _pad1;           // createTraceConfig: maxRadius, radiusGrowthPerPixel
};
static_assert(sizeof(LightRec) == 128, "LightRec is one 128-byte record");

// SphereLightVertexShader (SphereLightCore.fxh:13-56) over the 12-vertex cut-corner
// quad (FillSphereBuffer, LightingRenderer.cs:636-656) + createTraceConfig (ConeTrace.fxh:128-146)
__global__ __launch_bounds__(64) void prepare_lights_kernel(const IlmLightVertex* __restrict__ lights, int count, IlmEnvironment env,
                                                             float max_cone_radius, LightRec* __restrict__ out) {
    const int i = (int)blockIdx.x * 64 + (int)threadIdx.x;
    if (i >= count) return;
    const IlmLightVertex L = lights[i];
    LightRec r;
    r.cx = L.LightPosition1.x; r.cy = L.LightPosition1.y; r.cz = L.LightPosition1.z;
    r.radius = L.LightProperties.x; r.ramp = L.LightProperties.y; r.falloff_mode = L.LightProperties.z; r.casts_shadows = L.LightProperties.w;
    r.ao_radius = L.MoreLightProperties.x; r.shadow_falloff = L.MoreLightProperties.y; r.falloff_y = L.MoreLightProperties.z; r.ao_opacity = L.MoreLightProperties.w;
    r.shadow_filter = L.EvenMoreLightProperties.x;
    r.col_r = L.Color1.x * L.Color1.w; r.col_g = L.Color1.y * L.Color1.w; r.col_b = L.Color1.z * L.Color1.w;
    r.spec_r = L.Color2.x; r.spec_g = L.Color2.y; r.spec_b = L.Color2.z; r.spec_power = L.Color2.w;
    r.has_spec = ((L.Color2.x != 0.0f) || (L.Color2.y != 0.0f) || (L.Color2.z != 0.0f)) ? 1.0f : 0.0f;

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

    const float max_radius = clampf(r.radius, 0.33f, max_cone_radius);
    r.cfg_x = max_radius;
    r.cfg_y = max_radius / fmaxf(r.ramp, 16.0f) * 1.0f;  // getConeGrowthFactor() == 1 (DistanceFieldCommon.fxh:233-236)
    r._pad0 = r._pad1 = 0.0f;
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
    f3 shaded, normal, camera;
    bool enable_shadows, fullbright;
};

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
        world_z *= 1024.0f;   // GBUFFER_Z_SCALE
        world_z -= 1024.0f;   // GBUFFER_Z_OFFSET
        spx /= rsx; spy /= rsy;
        p.camera = mk3(spx, spy, env.ZAndScale.y + 0.01f);
        p.shaded = mk3((spx + 0.0f) / vsx + env.ViewportPosition[0], (spy + relative_y) / vsy + env.ViewportPosition[1], world_z);
        if ((s.x != 0.0f) || (s.y != 0.0f))
            p.normal = decode_normal(s.x, s.y);
        else
            p.normal = mk3(0.0f, 0.0f, 0.0f);
    } else {
        spx /= rsx; spy /= rsy;
        p.camera = mk3(spx, spy, env.ZAndScale.y + 0.01f);
        p.shaded = mk3(spx / vsx + env.ViewportPosition[0], spy / vsy + env.ViewportPosition[1], env.ZAndScale.x);
        p.normal = mk3(0.0f, 0.0f, 1.0f);
    }
    return p;
}

// computeSphereLightOpacity + computeNormalFactor, LightCommon.fxh:154-214
ILM_DEV float sphere_light_opacity(f3 shaded, f3 normal, const LightRec& L, float light_occlusion) {
    f3 d3 = shaded - mk3(L.cx, L.cy, L.cz);
    d3.y *= L.falloff_y;
    const float distance = len3(d3);
    float distance_factor = 1.0f - sat((distance - L.radius) / L.ramp);
    if (light_occlusion > 0.0f)
        distance_factor *= 1.0f - sat(d3.z / light_occlusion);
    float normal_factor = 1.0f;
    if ((normal.x != 0.0f) || (normal.y != 0.0f) || (normal.z != 0.0f)) {
        const f3 ln = mk3(d3.x / distance, d3.y / distance, d3.z / distance);
        const float d = dot3(ln * -1.0f, normal);
        normal_factor = powf(sat((d + 0.15f) / 0.15f), 0.85f);   // DOT_OFFSET, DOT_RAMP_RANGE, DOT_EXPONENT
    }
    if (L.falloff_mode >= 2.0f) {
        distance_factor = 1.0f - sat(distance - L.radius);
        normal_factor = 1.0f;
    } else if (L.falloff_mode >= 1.0f) {
        distance_factor *= distance_factor;
    }
    return sat((normal_factor * distance_factor) + sat(L.radius - distance));
}

constexpr int kTile = 16;
constexpr int kListCapacity = 1024;

template <int FMT, bool STATS>
__global__ __launch_bounds__(256) void sphere_lights_kernel(const LightLaunch a, const LightRec* __restrict__ recs, int tiles_x, int tiles_y, int tile_count) {
    __shared__ uint16_t list[kListCapacity];
    __shared__ int list_count;

    // XCD-aware remap: the dispatcher places block b on XCD b % 8; give every XCD a
    // contiguous band of tiles so neighbouring tiles (which walk the same SDF texels)
    // share one L2.
    const int nb = (int)gridDim.x;
    const int per_xcd = nb / 8;
    const int b = (int)blockIdx.x;
    const int tile = (b % 8) * per_xcd + (b / 8);
    if (tile >= tile_count)
        return;
    const int tx0 = (tile % tiles_x) * kTile, ty0 = a.row_begin + (tile / tiles_x) * kTile;

    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const int px = tx0 + (wave & 1) * 8 + (lane & 7);
    const int py = ty0 + (wave >> 1) * 8 + (lane >> 3);
    const bool in_image = (px < a.width) && (py < a.row_end);

    const Pixel P = sample_gbuffer((float)px, (float)py, a.env, a.gbuffer);
    const float cxp = (float)px + 0.5f, cyp = (float)py + 0.5f;
    const bool have_sdf = (a.sdf.texels != nullptr) && (a.df.Extent.x > 0.0f);

    float acc_r = a.ambient[0], acc_g = a.ambient[1], acc_b = a.ambient[2], acc_a = a.ambient[3];
    unsigned long long n_samples = 0, n_pairs = 0, n_traced = 0;

    for (int batch = 0; batch < a.light_count; batch += kListCapacity) {
        const int batch_n = min(kListCapacity, a.light_count - batch);
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
                bool hit = false;
                if (li < batch_n) {
                    const LightRec& R = recs[batch + li];
                    hit = (R.fx0 <= tmaxx) && (R.fx3 > tminx) && (R.fy0 <= tmaxy) && (R.fy3 > tminy);
                }
                const unsigned long long m = __ballot(hit);
                if (hit)
                    list[base + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)li;
                base += __popcll(m);
            }
            if (lane == 0) list_count = base;
        }
        __syncthreads();
        const int n = list_count;

        for (int k = 0; k < n; k++) {
            const int li = __builtin_amdgcn_readfirstlane((int)list[k]);
            const LightRec& L = recs[batch + li];

            // raster footprint: pixel centre inside the cross-shaped quad
            const bool covered = in_image && (((cxp >= L.fx1) && (cxp < L.fx2) && (cyp >= L.fy0) && (cyp < L.fy3)) ||
                                              ((cxp >= L.fx0) && (cxp < L.fx3) && (cyp >= L.fy1) && (cyp < L.fy2)));
            if (!covered)
                continue;
            if (STATS) n_pairs++;
            // checkShadowFilter, LightCommon.fxh:146-152
            const bool filtered = (L.shadow_filter < 0.0f) ? false : ((L.shadow_filter > 0.5f) != P.enable_shadows);
            if (P.fullbright || filtered)
                continue;

            const float casts = L.casts_shadows * (P.enable_shadows ? 1.0f : 0.0f);
            const float distance_opacity = sphere_light_opacity(P.shaded, P.normal, L, a.env.ZToY.z);
            const bool visible = (distance_opacity > 0.0f) && (P.shaded.x > -9999.0f);
            if (!visible)
                continue;

            // computeAO, AOCommon.fxh:1-19 (aoRadius scaled by max(0, normal.z), SphereLightCore.fxh:78)
            float ao_opacity = 1.0f;
            const float ao_radius = L.ao_radius * fmaxf(0.0f, P.normal.z);
            if ((ao_radius >= 0.5f) && have_sdf) {
                const float distance = sample_distance_field<FMT>(mk3(P.shaded.x, P.shaded.y, P.shaded.z + P.normal.z * ao_radius), a.df, a.sdf);
                if (STATS) n_samples++;
                float r = 1.0f - sat(clampf(distance, 0.0f, ao_radius) / ao_radius);
                r *= r;
                r = 1.0f - r;
                ao_opacity = (1.0f - L.ao_opacity) + (r * L.ao_opacity);
            }
            const float pre_trace = distance_opacity * ao_opacity;

            // coneTrace, ConeTrace.fxh:148-191
            float cone_opacity = 1.0f;
            const bool trace = (casts != 0.0f) && (pre_trace >= (0.75f / 255.0f));
            if (trace) {
                if (STATS) n_traced++;
                const f3 start = P.shaded + (P.normal * 1.6f);   // SELF_OCCLUSION_HACK
                const f3 tv = mk3(L.cx, L.cy, L.cz) - start;
                const float trace_length = len3(tv);
                const f3 dir = mk3(tv.x / trace_length, tv.y / trace_length, tv.z / trace_length);
                const float data_y = fmaxf(trace_length - L.radius, 1.0f);
                float data_x = 0.5f;   // TRACE_INITIAL_OFFSET_PX
                float data_z = 1.0f;
                const float cfg_z = fmaxf(1.0f, a.df.Packed1.w);
                float steps_remaining = a.df.StepAndMisc2.x;
                float liveness = have_sdf ? 1.0f : 0.0f;
                while (liveness > 0.0f) {
                    steps_remaining -= 1.0f;
                    const float s = sample_distance_field<FMT>(start + (dir * data_x), a.df, a.sdf);
                    if (STATS) n_samples++;
                    const float local_radius = fminf((L.cfg_y * data_x) + 0.33f, L.cfg_x);   // MIN_CONE_RADIUS
                    data_z = fminf(data_z, (s + 1.5f) / local_radius);                        // HACK_DISTANCE_OFFSET
                    data_x += fmaxf(fabsf(s) * a.df.StepAndMisc2.z, cfg_z);
                    liveness = steps_remaining * (sat(data_z - 0.075f) * sat(data_y - data_x));
                }
                const float visibility = fminf(data_z, steps_remaining / 2.0f);               // MAX_STEP_RAMP_WINDOW
                cone_opacity = powf(sat(sat(visibility - 0.075f) / (0.95f - 0.075f)), a.df.ConeAndMisc.z);
            }
            const float opacity = pre_trace * cone_opacity;

            // SphereLightPixelShader epilogue, SphereLight.fx:37-45.  The specular term is
            // skipped when Color2.rgb == 0: it then contributes exactly 0 unless
            // pow() produced inf/NaN (negative SpecularPower), which the reference does not guard.
            float sr = 0.0f, sg = 0.0f, sb = 0.0f;
            if (L.has_spec != 0.0f) {
                const f3 light_direction = P.shaded - mk3(L.cx, L.cy, L.cz);
                const f3 h = norm3(norm3(P.camera - P.shaded) - light_direction);
                const float specularity = powf(sat(dot3(h, P.normal)), L.spec_power);
                sr = L.spec_r * specularity * opacity;
                sg = L.spec_g * specularity * opacity;
                sb = L.spec_b * specularity * opacity;
            }
            acc_r += (L.col_r * opacity) + sr;
            acc_g += (L.col_g * opacity) + sg;
            acc_b += (L.col_b * opacity) + sb;
            acc_a += 1.0f;
        }
    }

    if (in_image) {
        const size_t o = (size_t)py * (size_t)a.width + (size_t)px;
        if (a.format == ILM_LIGHTMAP_FLOAT4) {
            reinterpret_cast<float4*>(a.lightmap)[o] = mk4(acc_r, acc_g, acc_b, acc_a);
        } else if (a.format == ILM_LIGHTMAP_HALF4) {
            uint2 v;
            v.x = (uint32_t)__half_as_ushort(__float2half_rn(acc_r)) | ((uint32_t)__half_as_ushort(__float2half_rn(acc_g)) << 16);
            v.y = (uint32_t)__half_as_ushort(__float2half_rn(acc_b)) | ((uint32_t)__half_as_ushort(__float2half_rn(acc_a)) << 16);
            reinterpret_cast<uint2*>(a.lightmap)[o] = v;
        } else {
            const uint32_t r = (uint32_t)rintf(sat(acc_r) * 255.0f), g = (uint32_t)rintf(sat(acc_g) * 255.0f);
            const uint32_t bl = (uint32_t)rintf(sat(acc_b) * 255.0f), al = (uint32_t)rintf(sat(acc_a) * 255.0f);
            reinterpret_cast<uint32_t*>(a.lightmap)[o] = r | (g << 8) | (bl << 16) | (al << 24);
        }
    }

    if (STATS) {
        // wave reduce, one atomic per wave and counter
        for (int off = 32; off > 0; off >>= 1) {
            n_samples += __shfl_down(n_samples, off);
            n_pairs += __shfl_down(n_pairs, off);
            n_traced += __shfl_down(n_traced, off);
        }
        if (lane == 0) {
            atomicAdd(&a.stats[0], n_samples);
            atomicAdd(&a.stats[1], n_pairs);
            atomicAdd(&a.stats[2], n_traced);
        }
    }
}

template <int FMT>
__global__ __launch_bounds__(256) void sdf_sample_kernel(SdfView sdf, IlmDistanceFieldUniforms df, const float* __restrict__ positions, int count,
                                                          float* __restrict__ out) {
    const int i = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (i >= count) return;
    out[i] = sample_distance_field<FMT>(mk3(positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]), df, sdf);
}

hipError_t launch_sdf_sample(const SdfView& sdf, const IlmDistanceFieldUniforms& df, const float* positions, int count, float* out, hipStream_t stream) {
    if (count <= 0) return hipSuccess;
    const dim3 grid((unsigned)((count + 255) / 256)), block(256);
    if (sdf.format == ILM_SDF_FP16) hipLaunchKernelGGL(sdf_sample_kernel<ILM_SDF_FP16>, grid, block, 0, stream, sdf, df, positions, count, out);
    else hipLaunchKernelGGL(sdf_sample_kernel<ILM_SDF_UNORM16>, grid, block, 0, stream, sdf, df, positions, count, out);
    return hipGetLastError();
}

// device scratch for the prepared light records, owned by the caller (api.hip)
hipError_t launch_prepare_lights(const IlmLightVertex* lights, int count, const IlmEnvironment& env, float max_cone_radius,
                                 void* recs, hipStream_t stream) {
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(prepare_lights_kernel, dim3((count + 63) / 64), dim3(64), 0, stream, lights, count, env, max_cone_radius,
                       reinterpret_cast<LightRec*>(recs));
    return hipGetLastError();
}

hipError_t launch_sphere_lights_prepared(const LightLaunch& a, const void* recs, hipStream_t stream) {
    const int rows = a.row_end - a.row_begin;
    if (rows <= 0 || a.width <= 0) return hipSuccess;
    const int tiles_x = (a.width + kTile - 1) / kTile, tiles_y = (rows + kTile - 1) / kTile;
    const int tile_count = tiles_x * tiles_y;
    const int blocks = ((tile_count + 7) / 8) * 8;
    const LightRec* r = reinterpret_cast<const LightRec*>(recs);
    const bool stats = a.stats != nullptr;
    const bool fp16 = a.sdf.format == ILM_SDF_FP16;
    if (stats) {
        if (fp16) hipLaunchKernelGGL((sphere_lights_kernel<ILM_SDF_FP16, true>), dim3(blocks), dim3(256), 0, stream, a, r, tiles_x, tiles_y, tile_count);
        else hipLaunchKernelGGL((sphere_lights_kernel<ILM_SDF_UNORM16, true>), dim3(blocks), dim3(256), 0, stream, a, r, tiles_x, tiles_y, tile_count);
    } else {
        if (fp16) hipLaunchKernelGGL((sphere_lights_kernel<ILM_SDF_FP16, false>), dim3(blocks), dim3(256), 0, stream, a, r, tiles_x, tiles_y, tile_count);
        else hipLaunchKernelGGL((sphere_lights_kernel<ILM_SDF_UNORM16, false>), dim3(blocks), dim3(256), 0, stream, a, r, tiles_x, tiles_y, tile_count);
    }
    return hipGetLastError();
}

}  // namespace ilm
