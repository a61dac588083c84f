/*
 * ilm_oracle_lights.c -- CPU restatement of particle lights and light probes.
 * TEST INFRASTRUCTURE ONLY (see ilm_oracle.h).  PARITY UNPINNED.
 * Textually included by ilm_oracle.c after orc_render_sphere_lights (shares its static helpers).
 *
 * Third-party code outside the tree: StippleReject lives in Fracture's DitherCommon.fxh (no pinned version); only
 * StippleFactor >= 1 -- the default, "no particle is rejected" -- is restated.
 */

/* SphereLightPixelCore, SphereLightCore.fxh:122-158.  Returns 0 with *discarded = 1 when the shader discards. */
static float sphere_light_pixel_core_ex(f3 shaded, f3 normal, f3 light_center, f4 light_properties, f4 more,
                                        const IlmEnvironment* env, const IlmDistanceFieldUniforms* df, const OrcTexture* sdf,
                                        SdfCounter* ctr, uint64_t* traced, int* discarded, float* pre_trace, float* cone) {
    /* SphereLightPixelPrologue, :58-81 */
    float distance_opacity = compute_sphere_light_opacity(shaded, normal, light_center, light_properties, more.z, env);
    int visible = (distance_opacity > 0.0f) && (shaded.x > -9999.0f);
    more.x *= fmaxf(0.0f, normal.z);
    *pre_trace = *cone = 0.0f;
    if (!visible) {
        *discarded = 1;
        return 0.0f;
    }
    *discarded = 0;
    float ao_opacity = compute_ao(shaded, normal, more, df, sdf, visible, ctr);
    float pre_trace_opacity = distance_opacity * ao_opacity;
    int trace_shadows = visible && (light_properties.w != 0.0f) && (pre_trace_opacity >= SL_SHADOW_OPACITY_THRESHOLD);
    if (trace_shadows && traced) (*traced)++;
    f3 start = v3add(shaded, v3scale(normal, SL_SELF_OCCLUSION_HACK));
    float cone_opacity = cone_trace(light_center, light_properties.x, light_properties.y, 1.0f, more.y, start, df, sdf, trace_shadows, ctr);
    *pre_trace = pre_trace_opacity; *cone = cone_opacity;
    return pre_trace_opacity * cone_opacity;
}
static float sphere_light_pixel_core(f3 shaded, f3 normal, f3 light_center, f4 light_properties, f4 more,
                                     const IlmEnvironment* env, const IlmDistanceFieldUniforms* df, const OrcTexture* sdf,
                                     SdfCounter* ctr, uint64_t* traced, int* discarded) {
    float pre, cone;
    return sphere_light_pixel_core_ex(shaded, normal, light_center, light_properties, more, env, df, sdf, ctr, traced, discarded, &pre, &cone);
}

/* ParticleLightVertexShader's quad, ParticleLight.fx:16-83: a plain rectangle of half-size radius + ramp + 1 whose top edge is
 * raised by radius / ZToY + z * ZToY; pixel centre inside [left, right) x [top, bottom) */
static int particle_light_covers_pixel(f3 center, float radius_sum, const IlmEnvironment* env, float cx, float cy) {
    float tlx = center.x - radius_sum, tly = center.y - radius_sum, brx = center.x + radius_sum, bry = center.y + radius_sum;
    tly -= radius_sum * env->ZToY.y;
    tly -= center.z * env->ZToY.x;
    const float sx = env->GBufferTexelSizeAndMisc.z * env->ZAndScale.z, sy = env->GBufferTexelSizeAndMisc.w * env->ZAndScale.w;
    float x0 = (tlx - env->ViewportPosition[0]) * sx, x1 = (brx - env->ViewportPosition[0]) * sx;
    float y0 = (tly - env->ViewportPosition[1]) * sy, y1 = (bry - env->ViewportPosition[1]) * sy;
    return (cx >= x0) && (cx < x1) && (cy >= y0) && (cy < y1);
}

/* technique ParticleLight (ParticleLight.fx:16-118) for every chunk of a system, additively blended onto `lightmap`
 * (RenderLighting draws it as one more light-type render state, LightingRenderer.cs:1126-1141).  planes[c][0] = position+life,
 * planes[c][3] = Chunk.RenderColor; quad_counts[c] = min(ChunkMaximumCount, TotalSpawned + 1) (RenderChunk, ParticleSystem.cs:880). */
void orc_render_particle_lights(IlmFloat4** planes, int32_t chunk_count, const int32_t* quad_counts,
                                const IlmParticleLightParams* p,
                                const IlmEnvironment* env, const IlmDistanceFieldUniforms* df,
                                const OrcTexture* gbuffer, const OrcTexture* sdf,
                                IlmFloat4* lightmap, int32_t width, int32_t height,
                                int32_t row_begin, int32_t row_end, IlmRenderStats* stats) {
    /* the vertex shader's per-particle work, once: live particles with a visible colour, in chunk / slot order */
    size_t cap = 0;
    for (int c = 0; c < chunk_count; c++) cap += (size_t)quad_counts[c];
    f3* centers = (f3*)malloc(sizeof(f3) * (cap ? cap : 1));
    f4* colors = (f4*)malloc(sizeof(f4) * (cap ? cap : 1));
    size_t n = 0;
    for (int c = 0; c < chunk_count; c++) {
        const IlmFloat4* pos = planes[c * 5 + 0];
        const IlmFloat4* rc = planes[c * 5 + 3];
        for (int i = 0; i < quad_counts[c]; i++) {
            f4 render_color = rc[i];
            if (render_color.w > 0.0f) {    /* unpremultiply, :43-45 */
                render_color.x /= render_color.w; render_color.y /= render_color.w; render_color.z /= render_color.w;
            }
            if (pos[i].w <= 0.0f)
                continue;                   /* :48-53 */
            f4 light_color = v4mul(render_color, p->LightColor);
            if (light_color.w <= 0.0f)
                continue;                   /* :79-82 */
            centers[n] = xyz(pos[i]);
            colors[n] = light_color;
            n++;
        }
    }
    const float radius_sum = p->LightProperties.x + p->LightProperties.y + 1.0f;
    uint64_t total_samples = 0, total_pairs = 0, total_traced = 0;
    if (row_begin < 0) row_begin = 0;
    if (row_end > height) row_end = height;
    #pragma omp parallel for schedule(dynamic, 4) reduction(+:total_samples, total_pairs, total_traced)
    for (int py = row_begin; py < row_end; py++) {
        SdfCounter ctr = { 0 };
        for (int px = 0; px < width; px++) {
            f4 acc = lightmap[(size_t)py * (size_t)width + (size_t)px];
            f3 shaded, normal;
            int enable_shadows, fullbright;
            f3 camera = sample_gbuffer((float)px, (float)py, env, gbuffer, &shaded, &normal, &enable_shadows, &fullbright);
            for (size_t li = 0; li < n; li++) {
                if (!particle_light_covers_pixel(centers[li], radius_sum, env, (float)px + 0.5f, (float)py + 0.5f))
                    continue;
                total_pairs++;
                if (fullbright)
                    continue;               /* discard, :100-104 */
                f4 light_properties = p->LightProperties;
                light_properties.w *= (float)enable_shadows;
                int discarded;
                uint64_t traced = 0;
                float opacity = sphere_light_pixel_core(shaded, normal, centers[li], light_properties, p->MoreLightProperties,
                                                        env, df, sdf, &ctr, &traced, &discarded);
                total_traced += traced;
                if (discarded)
                    continue;
                float specularity = calc_sphere_light_specularity(camera, shaded, normal, centers[li], p->LightSpecularColor.w);
                acc.x += (colors[li].x * colors[li].w * opacity) + (p->LightSpecularColor.x * specularity * opacity);
                acc.y += (colors[li].y * colors[li].w * opacity) + (p->LightSpecularColor.y * specularity * opacity);
                acc.z += (colors[li].z * colors[li].w * opacity) + (p->LightSpecularColor.z * specularity * opacity);
                acc.w += 1.0f;
            }
            lightmap[(size_t)py * (size_t)width + (size_t)px] = acc;
        }
        total_samples += ctr.samples;
    }
    free(centers); free(colors);
    if (stats) {
        stats->SdfSamples = total_samples;
        stats->PixelLightPairs = total_pairs;
        stats->TracedPairs = total_traced;
    }
}

/* technique SphereLightProbe (SphereLightProbe.fx:19-44) over a list of probes: every light reaches every probe
 * (SphereLightProbeVertexShader covers the whole N x 1 target, :4-17); the target is cleared to transparent
 * (UpdateLightProbes, LightingRenderer.LightProbes.cs:49-86).  probe_positions[i] = (position, 1), probe_normals[i] =
 * (normal or 0, enableShadows) as UpdateLightProbeTexture packs them (:88-110). */
void orc_render_light_probes(const IlmLightVertex* lights, int32_t light_count,
                             const IlmFloat4* probe_positions, const IlmFloat4* probe_normals, int32_t probe_count,
                             const IlmEnvironment* env, const IlmDistanceFieldUniforms* df, const OrcTexture* sdf,
                             IlmFloat4* out_values) {
    #pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < probe_count; i++) {
        f4 acc = v4(0, 0, 0, 0);
        SdfCounter ctr = { 0 };
        /* sampleLightProbeBuffer, LightCommon.fxh:233-254 */
        const float probe_opacity = probe_positions[i].w;
        if (probe_opacity > 0.0f) {
            const f3 shaded = xyz(probe_positions[i]);
            const f3 normal = xyz(probe_normals[i]);
            const float enable_shadows = probe_normals[i].w;
            for (int li = 0; li < light_count; li++) {
                const IlmLightVertex* L = &lights[li];
                f4 light_properties = L->LightProperties;
                light_properties.w *= enable_shadows;
                f4 more = L->MoreLightProperties;
                more.x = more.w = 0.0f;             /* no AO on probes, SphereLightProbe.fx:36 */
                int discarded;
                float pre, cone;
                (void)sphere_light_pixel_core_ex(shaded, normal, xyz(L->LightPosition1), light_properties, more,
                                                 env, df, sdf, &ctr, NULL, &discarded, &pre, &cone);
                if (discarded)
                    continue;
                /* opacity *= SphereLightPixelCore(...), or rgb = opacity * SphereLightPixelCoreWithRamp(...): SphereLightProbe.fx:39-44,67-71 */
                const f3 core = sphere_light_epilogue(pre, cone, v3sub(shaded, xyz(L->LightPosition1)), L->EvenMoreLightProperties);
                acc.x += L->Color1.x * L->Color1.w * (probe_opacity * core.x);
                acc.y += L->Color1.y * L->Color1.w * (probe_opacity * core.y);
                acc.z += L->Color1.z * L->Color1.w * (probe_opacity * core.z);
                acc.w += 1.0f;
            }
        }
        out_values[i] = acc;
    }
}
