/*
 * ilm_oracle_output.c -- CPU restatement of the output-side callers: FillReadbackResult and the lightmap resolve.
 * TEST INFRASTRUCTURE ONLY (see ilm_oracle.h).  PARITY UNPINNED.
 * Textually included by ilm_oracle.c (shares its static helpers).
 */

/* FillReadbackResult, Illuminant/Particles/ParticleReadback.cs:73-167 -- this one is C# CPU code in the reference, restated line
 * by line (float / double mix as C# evaluates it).  Returns the number of records written (chunk order, slot order). */
int32_t orc_fill_readback_result(IlmFloat4** planes, int32_t chunk_count, const int32_t* element_counts, int32_t slots,
                                 const IlmReadbackParams* p, IlmReadbackDrawCall* out, int32_t capacity) {
    const float anim_abs_x = fabsf(p->AnimationRate[0]), anim_abs_y = fabsf(p->AnimationRate[1]);
    const float region_w = p->TextureRegion[2] - p->TextureRegion[0], region_h = p->TextureRegion[3] - p->TextureRegion[1];   /* texSize = region.Size */
    int frame_count_x = (int)(1.0f / region_w), frame_count_y = (int)(1.0f / region_h);
    if (frame_count_x < 1) frame_count_x = 1;
    if (frame_count_y < 1) frame_count_y = 1;
    const double max_angle_x = (2 * M_PI) / frame_count_x, max_angle_y = (2 * M_PI) / frame_count_y;
    const double vel_rotation = p->RotationFromVelocity ? 1.0 : 0.0;
    int32_t result = 0;
    for (int c = 0; c < chunk_count; c++) {
        const IlmFloat4* position_and_life = planes[c * 5 + 0];
        const IlmFloat4* render_color = planes[c * 5 + 3];
        const IlmFloat4* render_data = planes[c * 5 + 4];
        const int count = element_counts ? element_counts[c] : slots;
        for (int i = 0; i < count && i < slots; i++) {
            const f4 pl = position_and_life[i];
            const float life = pl.w;
            if (life <= 0.0f)
                continue;
            const f4 rd = render_data[i], rc = render_color[i];
            const float sz = rd.x;
            const float rot = fmodf(rd.y, (float)(2 * M_PI));
            IlmReadbackDrawCall dc;
            memset(&dc, 0, sizeof(dc));
            for (int k = 0; k < 4; k++) dc.TextureRegion[k] = p->TextureRegion[k];
            if ((frame_count_x > 1) || (frame_count_y > 1)) {
                float fx = floorf(anim_abs_x * life), fy = floorf(anim_abs_y * life);
                fy += (float)floor((double)rd.w);
                if (p->ColumnFromVelocity) fx += (float)nearbyint(rot / max_angle_x);     /* Math.Round: half to even */
                if (p->RowFromVelocity)    fy += (float)nearbyint(rot / max_angle_y);
                fx = fmodf(fmaxf(0.0f, fx), (float)frame_count_x);
                fy = h_clamp(fy, 0.0f, (float)(frame_count_y - 1));
                if (p->AnimationRate[0] < 0.0f) fx = frame_count_x - fx;
                if (p->AnimationRate[1] < 0.0f) fy = frame_count_y - fy;
                const float ox = fx * region_w, oy = fy * region_h;
                dc.TextureRegion[0] += ox; dc.TextureRegion[1] += oy; dc.TextureRegion[2] += ox; dc.TextureRegion[3] += oy;
            }
            dc.Position[0] = pl.x; dc.Position[1] = pl.y;
            if (p->SortedReadback)
                dc.SortOrder = pl.y + p->ZToY;
            dc.Scale[0] = p->Size[0] * sz; dc.Scale[1] = p->Size[1] * sz;
            dc.MultiplyColor[0] = (uint8_t)(int32_t)(rc.x * 255.0f);    /* (byte)(float): truncation */
            dc.MultiplyColor[1] = (uint8_t)(int32_t)(rc.y * 255.0f);
            dc.MultiplyColor[2] = (uint8_t)(int32_t)(rc.z * 255.0f);
            dc.MultiplyColor[3] = (uint8_t)(int32_t)(rc.w * 255.0f);
            dc.Rotation = (float)(vel_rotation * rot);
            if (result < capacity)
                out[result] = dc;
            result++;
        }
    }
    return result;
}

/* HDR.fxh:1-44 */
static f4 gamma_compress(f4 color, float offset, float middle_gray, float average_luminance, float maximum_luminance_squared) {
    f3 rgb = v3(fmaxf(color.x + offset, 0.0f), fmaxf(color.y + offset, 0.0f), fmaxf(color.z + offset, 0.0f));
    float result_luminance = rgb.x * 0.299f + rgb.y * 0.587f + rgb.z * 0.114f;
    float scaled_luminance = (result_luminance * middle_gray) / average_luminance;
    float compressed_luminance = (scaled_luminance * (1.0f + (scaled_luminance / maximum_luminance_squared))) / (1.0f + scaled_luminance);
    float rescale_factor = compressed_luminance / result_luminance;
    return v4(rgb.x * rescale_factor, rgb.y * rescale_factor, rgb.z * rescale_factor, color.w);
}
static float uncharted2_tonemap1(float value) {
    const float kA = 0.15f, kB = 0.50f, kC = 0.10f, kD = 0.20f, kE = 0.02f, kF = 0.30f;
    return ((value * (kA * value + kC * kB) + kD * kE) / (value * (kA * value + kB) + kD * kF)) - kE / kF;
}

/* LightingResolvePixelShader / GammaCompressedLightingResolvePixelShader / ToneMappedLightingResolvePixelShader, Resolve.fx:62-139,
 * with ResolveCommon (:25-40) at scale 1 (each output pixel reads its own lightmap texel); parameter clamps of
 * SetGammaCompressionParameters / SetToneMappingParameters (IlluminantMaterials.cs:81-137). */
void orc_resolve_lighting_with_albedo(const IlmFloat4* lightmap, const IlmFloat4* albedo, int32_t width, int32_t height, const IlmHDRConfiguration* hdr,
                                      IlmFloat4* out, int32_t row_begin, int32_t row_end);
void orc_resolve_lighting(const IlmFloat4* lightmap, int32_t width, int32_t height, const IlmHDRConfiguration* hdr,
                          IlmFloat4* out, int32_t row_begin, int32_t row_end) {
    orc_resolve_lighting_with_albedo(lightmap, NULL, width, height, hdr, out, row_begin, row_end);
}

/* ... and LightingResolveWithAlbedoPixelShader / GammaCompressed... / ToneMapped... (Resolve.fx:141-233) when `albedo` is given:
 * ResolveWithAlbedoCommon (:43-60) with AlbedoIsSRGB = 0, each output pixel reading its own texel of both textures. */
void orc_resolve_lighting_with_albedo(const IlmFloat4* lightmap, const IlmFloat4* albedo_texels, int32_t width, int32_t height, const IlmHDRConfiguration* hdr,
                                      IlmFloat4* out, int32_t row_begin, int32_t row_end) {
    const float min_v = 1.0f / 256.0f, max_v = 99999.0f;
    const float inverse_scale = (hdr->InverseScaleFactor != 0.0f) ? hdr->InverseScaleFactor : 1.0f;
    const float exposure = h_clamp(hdr->Exposure, min_v, max_v);
    const float white_point = (hdr->Mode == ILM_HDR_TONE_MAP) ? h_clamp(hdr->WhitePoint, min_v, max_v) : h_clamp(1.0f, min_v, max_v);
    const float gamma = h_clamp(hdr->Gamma, 0.1f, 4.0f);
    const float exposure_minus_one = exposure - 1.0f, gamma_minus_one = gamma - 1.0f;
    const float middle_gray = h_clamp(hdr->MiddleGray, 0.0f, max_v);
    const float average_luminance = h_clamp(hdr->AverageLuminance, min_v, max_v);
    const float maximum_luminance = h_clamp(hdr->MaximumLuminance, min_v, max_v);
    const float maximum_luminance_squared = maximum_luminance * maximum_luminance;
    if (row_begin < 0) row_begin = 0;
    if (row_end > height) row_end = height;
    #pragma omp parallel for schedule(static)
    for (int y = row_begin; y < row_end; y++)
        for (int x = 0; x < width; x++) {
            const f4 color = lightmap[(size_t)y * (size_t)width + (size_t)x];
            f4 r;
            if (albedo_texels) {
                /* ResolveWithAlbedoCommon: light *= InverseScaleFactor * 2; lerp(albedo.rgb, albedo.rgb * light.rgb, saturate(light.a)) */
                const f4 albedo = albedo_texels[(size_t)y * (size_t)width + (size_t)x];
                const float k = inverse_scale * 2.0f;
                const f4 light = v4(color.x * k, color.y * k, color.z * k, color.w * k);
                const float t = h_clamp(light.w, 0.0f, 1.0f);
                r = v4(albedo.x + (albedo.x * light.x - albedo.x) * t, albedo.y + (albedo.y * light.y - albedo.y) * t,
                       albedo.z + (albedo.z * light.z - albedo.z) * t, albedo.w);
            } else {
                r = v4(color.x * inverse_scale, color.y * inverse_scale, color.z * inverse_scale, 1.0f);   /* ResolveCommon */
            }
            if (hdr->Mode == ILM_HDR_GAMMA_COMPRESS) {
                r = gamma_compress(r, hdr->Offset, middle_gray, average_luminance, maximum_luminance_squared);
            } else if (hdr->Mode == ILM_HDR_TONE_MAP) {
                f3 pre = v3(fmaxf(0.0f, r.x + hdr->Offset) * (exposure_minus_one + 1.0f), fmaxf(0.0f, r.y + hdr->Offset) * (exposure_minus_one + 1.0f),
                            fmaxf(0.0f, r.z + hdr->Offset) * (exposure_minus_one + 1.0f));
                const float w = uncharted2_tonemap1(white_point);
                r = v4(uncharted2_tonemap1(pre.x) / w, uncharted2_tonemap1(pre.y) / w, uncharted2_tonemap1(pre.z) / w, r.w);
                r.x = powf(r.x, gamma_minus_one + 1.0f); r.y = powf(r.y, gamma_minus_one + 1.0f); r.z = powf(r.z, gamma_minus_one + 1.0f);
            } else {
                r.x = fmaxf(0.0f, r.x + hdr->Offset); r.y = fmaxf(0.0f, r.y + hdr->Offset); r.z = fmaxf(0.0f, r.z + hdr->Offset);
                r.x *= (exposure_minus_one + 1.0f); r.y *= (exposure_minus_one + 1.0f); r.z *= (exposure_minus_one + 1.0f);
                r.x = powf(r.x, gamma_minus_one + 1.0f); r.y = powf(r.y, gamma_minus_one + 1.0f); r.z = powf(r.z, gamma_minus_one + 1.0f);
            }
            out[(size_t)y * (size_t)width + (size_t)x] = r;
        }
}

/* ---- particle rasterisation: technique RasterizeParticlesNoTexture ------------------------------------------------------------
 * VS_PosVelAttr + PS_NoTexture, Illuminant/Shaders/RasterizeParticleSystem.fx:61-148,150-163,228-241, drawn one instanced quad per
 * slot in chunk / slot order (RenderChunk, Illuminant/Particles/ParticleSystem.cs:876-908) with the blend state of the caller.
 * The view transform (Fracture, not in the tree) is the default one of a render target: pixel = screen * ViewportScale with pixel
 * centres at + 0.5; a pixel belongs to the quad when its centre maps to unit coordinates in [-1, 1) x [-1, 1).  No depth buffer.
 * One sprite in pixel space: centre + the inverse of the affine map unit square -> pixels. */
typedef struct OrcSprite {
    float cx, cy;               /* centre, pixels */
    float i00, i01, i10, i11;   /* unit = I * (pixel - centre) */
    float ex, ey;               /* half extents of the bounding box, pixels */
    f4 color;                   /* RenderColor (x GlobalColor for NoTexture; the textured pixel shaders apply it after the texel) */
    float rounding;
    float frame_u, frame_v;     /* frameTexCoord: offset of the animation frame inside the sheet */
    float dither_frame;         /* floor(index % 4) of premultipliedToDithered, index = the slot (ParticleEngine.cs:476-478) */
    int live;
} OrcSprite;

/* HLSL round(): to nearest, ties to even */
static float hlsl_round(float x) { return (float)nearbyint((double)x); }

/* Dither64 of Fracture's DitherCommon.fxh (Squared/RenderLib/Shaders, outside the reference tree, no pinned version), which restates
 * the published function of J. Jimenez, "Next Generation Post Processing in Call of Duty: Advanced Warfare" (SIGGRAPH 2014):
 * frac(dot(float3(Pos.xy, FrameIndexMod4), uint3(33, 52, 25) / 64.0)).  Every term is a multiple of 1/64 below 2^18, so the sum is
 * exact in whatever order the dot product adds. */
static float dither64(float x, float y, float frame_index_mod4) {
    const float d = ((x * (33.0f / 64.0f)) + (y * (52.0f / 64.0f))) + (frame_index_mod4 * (25.0f / 64.0f));
    return d - floorf(d);
}

static OrcSprite raster_sprite(f4 position, f4 render_data, f4 render_color, const IlmRasterizeParams* p, int slot) {
    OrcSprite sp;
    memset(&sp, 0, sizeof(sp));
    sp.dither_frame = floorf(fmodf((float)slot, 4.0f));
    const float life = position.w;
    if (life <= 0.0f)                                     /* StippleReject: StippleFactor >= 1 rejects nothing */
        return sp;
    float angle = fmodf(render_data.y, (float)(2 * M_PI));
    float sx = render_data.x * p->SystemSize[0] * p->SizeFactorAndPosition.x;
    float sy = render_data.x * p->SystemSize[1] * p->SizeFactorAndPosition.y;
    const float zf = fmaxf(0.0f, 1.0f + (position.z * p->ZConfiguration.x));
    sx *= zf; sy *= zf;
    const float s = sinf(angle), c = cosf(angle);
    const float display_x = (position.x * p->Scale.x) + p->SizeFactorAndPosition.z;
    const float display_y = ((position.y - (position.z * p->ZToY)) * p->Scale.y) + p->SizeFactorAndPosition.w;
    sp.cx = (display_x - p->ViewportPosition[0]) * p->ViewportScale[0];
    sp.cy = (display_y - p->ViewportPosition[1]) * p->ViewportScale[1];
    /* rotatedCorner = (c ux sx - s uy sy, s ux sx + c uy sy) * Scale.xy, then * ViewportScale: pixel - centre = A * unit */
    const float kx = p->Scale.x * p->ViewportScale[0], ky = p->Scale.y * p->ViewportScale[1];
    const float a00 = (c * sx) * kx, a01 = -(s * sy) * kx;
    const float a10 = (s * sx) * ky, a11 = (c * sy) * ky;
    const float det = (a00 * a11) - (a01 * a10);
    if (!(fabsf(det) > 0.0f) || !isfinite(det) || !isfinite(sp.cx) || !isfinite(sp.cy))
        return sp;                                        /* degenerate quad: no pixel */
    sp.i00 = a11 / det;  sp.i01 = -a01 / det;
    sp.i10 = -a10 / det; sp.i11 = a00 / det;
    sp.ex = fabsf(a00) + fabsf(a01);
    sp.ey = fabsf(a10) + fabsf(a11);
    sp.color = (p->BitmapFilter == ILM_BITMAP_NONE) ? v4mul(render_color, p->GlobalColor) : render_color;
    sp.rounding = h_clamp(orc_bezier1(&p->RoundingPowerFromLife, life), 0.001f, 1.0f);
    if (p->BitmapFilter != ILM_BITMAP_NONE) {
        /* frame selection, RasterizeParticleSystem.fx:112-139 */
        const float tex_w = p->BitmapTextureRegion.z - p->BitmapTextureRegion.x, tex_h = p->BitmapTextureRegion.w - p->BitmapTextureRegion.y;
        const float count_x = floorf(1.0f / tex_w), count_y = floorf(1.0f / tex_h);
        float fx = floorf(fabsf(p->AnimationRate[0]) * life), fy = floorf(fabsf(p->AnimationRate[1]) * life);
        const float max_angle_x = (float)(2 * M_PI) / count_x, max_angle_y = (float)(2 * M_PI) / count_y;
        fy += floorf(render_data.w);
        if (p->RenderingOptions[2] != 0.0f) fx += hlsl_round(angle / max_angle_x);
        if (p->RenderingOptions[3] != 0.0f) fy += hlsl_round(angle / max_angle_y);
        fx = fmodf(fmaxf(fx, 0.0f), count_x);
        fy = h_clamp(fy, 0.0f, count_y - 1.0f);
        if (p->AnimationRate[0] < 0.0f) fx = (count_x - fx) - 1.0f;
        if (p->AnimationRate[1] < 0.0f) fy = (count_y - fy) - 1.0f;
        sp.frame_u = fx * tex_w; sp.frame_v = fy * tex_h;
    }
    sp.live = 1;
    return sp;
}

/* computeCircularAlpha, RasterizeParticleSystem.fx:150-163 */
static float raster_circular_alpha(float u, float v, float rounding, float rounded) {
    if (rounded == 0.0f)
        return 1.0f;
    const float distance = sqrtf((u * u) + (v * v));
    const float power = fmaxf(rounding, 0.01f);
    const float divisor = fmaxf(h_sat(1.0f - power), 0.001f);
    const float distance_from_edge = h_sat(distance - power) / divisor;
    return h_sat(1.0f - powf(distance_from_edge, power));
}

/* tex2D on a bitmap without mips: BitmapPointSampler (POINT, CLAMP) or BitmapSampler (LINEAR, CLAMP), texel centres at + 0.5 */
static f4 bitmap_fetch(const IlmFloat4* tex, int w, int h, float u, float v, int filter) {
    if (filter == ILM_BITMAP_POINT) {
        float xf = floorf(u * (float)w), yf = floorf(v * (float)h);
        if (!(xf >= 0.0f)) xf = 0.0f; if (xf > (float)(w - 1)) xf = (float)(w - 1);
        if (!(yf >= 0.0f)) yf = 0.0f; if (yf > (float)(h - 1)) yf = (float)(h - 1);
        return tex[(int)yf * w + (int)xf];
    }
    const float sx = u * (float)w - 0.5f, sy = v * (float)h - 0.5f;
    float x0f = floorf(sx), y0f = floorf(sy);
    const float fx = sx - x0f, fy = sy - y0f;
    float x1f = x0f + 1.0f, y1f = y0f + 1.0f;
    if (!(x0f >= 0.0f)) x0f = 0.0f; if (x0f > (float)(w - 1)) x0f = (float)(w - 1);
    if (!(x1f >= 0.0f)) x1f = 0.0f; if (x1f > (float)(w - 1)) x1f = (float)(w - 1);
    if (!(y0f >= 0.0f)) y0f = 0.0f; if (y0f > (float)(h - 1)) y0f = (float)(h - 1);
    if (!(y1f >= 0.0f)) y1f = 0.0f; if (y1f > (float)(h - 1)) y1f = (float)(h - 1);
    const int x0 = (int)x0f, x1 = (int)x1f, y0 = (int)y0f, y1 = (int)y1f;
    return v4lerp(v4lerp(tex[y0 * w + x0], tex[y0 * w + x1], fx), v4lerp(tex[y1 * w + x0], tex[y1 * w + x1], fx), fy);
}

/* image: width * height float4, blended in place.  stats (may be NULL): live quads, shaded pixels.
 * bitmap (bitmap_w x bitmap_h float4, one level) is read when p->BitmapFilter != ILM_BITMAP_NONE. */
void orc_render_particles_textured(IlmFloat4** planes, int32_t chunk_count, const int32_t* quad_counts, int32_t slots,
                                   const IlmRasterizeParams* p, const IlmFloat4* bitmap, int32_t bitmap_w, int32_t bitmap_h,
                                   IlmFloat4* image, int32_t width, int32_t height, uint64_t* stats);

void orc_render_particles(IlmFloat4** planes, int32_t chunk_count, const int32_t* quad_counts, int32_t slots,
                          const IlmRasterizeParams* p, IlmFloat4* image, int32_t width, int32_t height, uint64_t* stats) {
    orc_render_particles_textured(planes, chunk_count, quad_counts, slots, p, NULL, 0, 0, image, width, height, stats);
}

void orc_render_particles_textured(IlmFloat4** planes, int32_t chunk_count, const int32_t* quad_counts, int32_t slots,
                                   const IlmRasterizeParams* p, const IlmFloat4* bitmap, int32_t bitmap_w, int32_t bitmap_h,
                                   IlmFloat4* image, int32_t width, int32_t height, uint64_t* stats) {
    const int textured = (p->BitmapFilter != ILM_BITMAP_NONE) && bitmap && bitmap_w > 0 && bitmap_h > 0;
    const float region_x = p->BitmapTextureRegion.x, region_y = p->BitmapTextureRegion.y;
    const float region_w = p->BitmapTextureRegion.z - p->BitmapTextureRegion.x, region_h = p->BitmapTextureRegion.w - p->BitmapTextureRegion.y;
    uint64_t live = 0, shaded = 0;
    for (int c = 0; c < chunk_count; c++) {
        const int count = quad_counts ? quad_counts[c] : slots;
        for (int i = 0; i < count && i < slots; i++) {
            const OrcSprite sp = raster_sprite(planes[c * 5 + 0][i], planes[c * 5 + 4][i], planes[c * 5 + 3][i], p, i);
            if (!sp.live)
                continue;
            live++;
            /* pixel centres within the bounding box (one pixel of slack; the unit-square test decides) */
            float fx0 = floorf(sp.cx - sp.ex - 0.5f) - 1.0f, fx1 = ceilf(sp.cx + sp.ex - 0.5f) + 1.0f;
            float fy0 = floorf(sp.cy - sp.ey - 0.5f) - 1.0f, fy1 = ceilf(sp.cy + sp.ey - 0.5f) + 1.0f;
            if (fx0 < 0.0f) fx0 = 0.0f; if (fy0 < 0.0f) fy0 = 0.0f;
            if (fx1 > (float)(width - 1)) fx1 = (float)(width - 1);
            if (fy1 > (float)(height - 1)) fy1 = (float)(height - 1);
            if (!(fx0 <= fx1) || !(fy0 <= fy1))
                continue;
            for (int y = (int)fy0; y <= (int)fy1; y++)
                for (int x = (int)fx0; x <= (int)fx1; x++) {
                    const float dx = ((float)x + 0.5f) - sp.cx, dy = ((float)y + 0.5f) - sp.cy;
                    const float u = (sp.i00 * dx) + (sp.i01 * dy), v = (sp.i10 * dx) + (sp.i11 * dy);
                    if (!((u >= -1.0f) && (u < 1.0f) && (v >= -1.0f) && (v < 1.0f)))
                        continue;
                    const float alpha = raster_circular_alpha(u, v, sp.rounding, p->RenderingOptions[0]);
                    f4 result = sp.color;
                    if (textured && (result.w > 0.0f)) {       /* PS_Texture: `color.a > (1 / 512)`, an integer division */
                        /* texCoord = lerp(region.xy, region.zw, unit / 2 + 0.5) + frameTexCoord, interpolated over the quad */
                        const float tu = (region_x + (region_w * ((u / 2.0f) + 0.5f))) + sp.frame_u;
                        const float tv = (region_y + (region_h * ((v / 2.0f) + 0.5f))) + sp.frame_v;
                        result = v4mul(v4mul(result, bitmap_fetch(bitmap, bitmap_w, bitmap_h, tu, tv, p->BitmapFilter)), p->GlobalColor);
                    }
                    f4 src = v4scale(result, alpha);
                    if (p->RenderingOptions[1] >= 0.5f) {       /* premultipliedToDithered, RasterizeParticleSystem.fx:158-175; GET_VPOS = floor(vpos) */
                        const float discard_threshold = RASTER_DITHER_DISCARD_NUMERATOR / 255.0f;
                        if ((src.w <= dither64((float)x, (float)y, sp.dither_frame)) || (src.w <= discard_threshold)) {
                            src = v4(0, 0, 0, 0);
                        } else {
                            const float a = fmaxf(src.w, 0.0001f);
                            src = v4(src.x / a, src.y / a, src.z / a, 1.0f);
                        }
                    }
                    if (src.w <= 0.0f)                    /* `result.a <= (1 / 512)`: integer division, i.e. <= 0 */
                        continue;
                    shaded++;
                    f4* dst = &image[(size_t)y * (size_t)width + (size_t)x];
                    const float keep = (p->BlendMode == ILM_BLEND_ADDITIVE) ? 1.0f : (1.0f - src.w);
                    dst->x = src.x + (dst->x * keep); dst->y = src.y + (dst->y * keep);
                    dst->z = src.z + (dst->z * keep); dst->w = src.w + (dst->w * keep);
                }
        }
    }
    if (stats) { stats[0] = live; stats[1] = shaded; }
}
