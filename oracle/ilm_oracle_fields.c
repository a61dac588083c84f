/*
 * ilm_oracle_fields.c -- CPU restatement of the distance-field generation pass.
 * TEST INFRASTRUCTURE ONLY (see ilm_oracle.h).  PARITY UNPINNED.
 * Textually included at the end of ilm_oracle.c (it shares that file's static HLSL helpers).
 *
 * Follows RenderDistanceFieldPartition -> RenderDistanceFieldSliceTriplet
 * (Illuminant/Lighting/LightingRenderer.DistanceField.cs:80-152,415-464), the
 * techniques Box/Ellipsoid/Cylinder/Spheroid/Octagon (Illuminant/Shaders/DistanceFunction.fx:15-115),
 * DistanceToPolygon (Illuminant/Shaders/DistanceField.fx:33-115) and ClearDistanceField
 * (Illuminant/Shaders/ClearDistanceField.fx:30-44), blended with BlendFunction.Max
 * (Illuminant/LoadMaterials.cs:164-176).
 *
 * Third-party code outside the tree: sdPolygonInit / sdPolygonVertex live in Fracture's
 * Squared/RenderLib/Shaders/SDF2D.fxh (sibling repository, no pinned version,
 * Illuminant/Illuminant.csproj:99-105).  They are restated from the published algorithm the header
 * credits (Inigo Quilez, "2D distance functions", sdPolygon): squared distance to the closest edge and a
 * sign flipped by the even-odd crossing test.  TransformPosition (ViewTransformCommon.fxh, also Fracture) is
 * an orthographic projection; the raster rule used here is "pixel (i, j) is covered iff its centre
 * (i + 0.5, j + 0.5) lies in [left, right) x [top, bottom)", the same rule the light pass uses.
 */

/* SliceIndexToZ, LightingRenderer.DistanceField.cs:32-35 */
static float slice_index_to_z(const IlmDistanceFieldRenderDesc* d, int slice) {
    float denom = (float)d->SliceCount;
    if (denom < 1.0f) denom = 1.0f;
    float slice_z = (float)slice / denom;
    return (slice_z * d->VirtualDepth) + d->ZOffset;
}

/* evaluate* by LightObstructionType (LightObstruction.cs:10-16; the technique table
 * IlluminantMaterials.DistanceFunctionTypes is indexed by it, LoadMaterials.cs:154-162) */
static float evaluate_obstruction(int type, f3 wp, f3 center, f3 size, f4 rot) {
    switch (type) {
        case ILM_OBSTRUCTION_ELLIPSOID: return evaluate_ellipsoid(wp, center, size, rot);
        case ILM_OBSTRUCTION_BOX:       return evaluate_box(wp, center, size, rot);
        case ILM_OBSTRUCTION_CYLINDER:  return evaluate_cylinder(wp, center, size, rot);
        case ILM_OBSTRUCTION_SPHEROID:  return evaluate_spheroid(wp, center, size, rot);
        case ILM_OBSTRUCTION_OCTAGON:   return evaluate_octagon(wp, center, size, rot);
        default: return 0.0f;
    }
}

/* IQ sdPolygon, one edge (vi = edge end "b", vj = edge start "a" in loadEdge order, DistanceField.fx:41-45,82-88) */
static void sd_polygon_vertex(float px, float py, float vix, float viy, float vjx, float vjy, float* d, float* s) {
    float ex = vjx - vix, ey = vjy - viy;
    float wx = px - vix, wy = py - viy;
    float t = h_clamp((wx * ex + wy * ey) / (ex * ex + ey * ey), 0.0f, 1.0f);
    float bx = wx - ex * t, by = wy - ey * t;
    *d = fminf(*d, bx * bx + by * by);
    int c0 = py >= viy, c1 = py < vjy, c2 = (ex * wy) > (ey * wx);
    if ((c0 && c1 && c2) || (!c0 && !c1 && !c2))
        *s = -*s;
}

/* computeDistanceZ, DistanceField.fx:47-56 */
static float compute_distance_z(float slice_z, float z0, float z1) {
    if (slice_z >= z0) {
        if (slice_z <= z1)
            return fmaxf(slice_z - z1, z0 - slice_z);
        else
            return slice_z - z1;
    } else
        return z0 - slice_z;
}

/* finalEval, DistanceField.fx:58-73 (PolygonXyBias 1.5, :13) */
static float final_eval(float z, float z0, float z1, float dist_sq, float sign) {
    float distance_z = compute_distance_z(z, z0, z1);
    float distance_xy = (sqrtf(dist_sq) * sign) + 1.5f;
    if (distance_xy <= 0.0f) {
        if (distance_z <= 0.0f)
            return distance_xy + distance_z;
        else
            return distance_z;
    } else
        return fmaxf(distance_xy, 0.0f) + fmaxf(distance_z, 0.0f);
}

/* render-target write of one channel: saturate, then D3D float -> unorm16 (c * 65535 + 0.5, truncated), or the
 * IEEE half of the saturated value for the fp16 atlas */
static uint16_t float_to_half_rn(float f);
static uint16_t sdf_store_channel(float enc, int format) {
    float c = h_sat(enc);
    if (format == ILM_SDF_FP16)
        return float_to_half_rn(c);
    return (uint16_t)floorf(c * 65535.0f + 0.5f);
}

static uint16_t float_to_half_rn(float f) {
    /* round-to-nearest-even float -> half for f in [0, 1] (never overflows, may be subnormal) */
    union { float f; uint32_t u; } v = { f };
    uint32_t x = v.u & 0x7FFFFFFFu;
    if (x < 0x38800000u) {            /* below 2^-14: subnormal half */
        float scaled = f * 16777216.0f;                /* f * 2^24: the subnormal mantissa as a real number */
        float r = nearbyintf(scaled);
        return (uint16_t)r;
    }
    uint32_t mant = x & 0x007FFFFFu, exp = (x >> 23) - 112u;
    uint32_t half = (exp << 10) | (mant >> 13);
    uint32_t rem = mant & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (half & 1u))) half++;
    return (uint16_t)half;
}

void orc_render_distance_field_slices(uint16_t* atlas, int32_t format, const uint16_t* clear_source,
                                      const IlmDistanceFieldRenderDesc* d,
                                      const int32_t* first_virtual_slices, int32_t triplet_count,
                                      const IlmObstruction* obstructions, int32_t obstruction_count,
                                      const IlmHeightVolume* volumes, int32_t volume_count,
                                      const float* polygon_xy, int32_t polygon_vertex_count) {
    (void)polygon_vertex_count;
    const int SW = d->SliceWidth, SH = d->SliceHeight;
    const int atlas_w = SW * d->ColumnCount;
    const float max_enc = d->MaximumEncodedDistance;
    /* the orthographic view transform maps [0, VirtualWidth * ColumnCount] onto the atlas width
     * (RenderDistanceFieldSliceTriplet, :97-102): virtual units -> atlas pixels */
    const float px_per_unit_x = (float)SW / (float)d->VirtualWidth;
    const float px_per_unit_y = (float)SH / (float)d->VirtualHeight;

    for (int t = 0; t < triplet_count; t++) {
        const int first = first_virtual_slices[t];
        const int physical = first / 3;                                   /* PackedSliceCount, :452 */
        const int slice_x = (physical % d->ColumnCount) * SW;             /* :91-94 */
        const int slice_y = (physical / d->ColumnCount) * SH;
        const int slice_x_virtual = (physical % d->ColumnCount) * d->VirtualWidth;
        const int slice_y_virtual = (physical / d->ColumnCount) * d->VirtualHeight;
        const float vpx = -(float)slice_x_virtual, vpy = -(float)slice_y_virtual;   /* viewTransform.Position, :102 */
        float slice_z[4];
        for (int k = 0; k < 4; k++) slice_z[k] = slice_index_to_z(d, first + k);    /* :358-363 */

#pragma omp parallel for schedule(static)
        for (int j = 0; j < SH; j++) {
            for (int i = 0; i < SW; i++) {
                const int ax = slice_x + i, ay = slice_y + j;
                uint16_t* texel = atlas + ((size_t)ay * (size_t)atlas_w + (size_t)ax) * 4;
                /* ClearDistanceFieldSlice (:266-287): transparent, or the static texture's texel */
                uint16_t cleared[4] = { 0, 0, 0, 0 };
                if (clear_source) {
                    const uint16_t* src = clear_source + ((size_t)ay * (size_t)atlas_w + (size_t)ax) * 4;
                    for (int k = 0; k < 4; k++) cleared[k] = src[k];
                }
                /* getPositionXy, DistanceFunction.fx:28-31 (vpos = integer atlas pixel) */
                const float wx = ((float)ax * d->InvScaleFactorX) + vpx;
                const float wy = ((float)ay * d->InvScaleFactorY) + vpy;
                const float cxp = (float)i + 0.5f, cyp = (float)j + 0.5f;
                float acc[4] = { 0.0f, 0.0f, 0.0f, 0.0f };   /* saturate() at the target floors every write at 0 */

                for (int o = 0; o < obstruction_count; o++) {
                    const IlmObstruction* ob = &obstructions[o];
                    if (d->DynamicFlagFilter >= 0 && ((ob->IsDynamic != 0) != (d->DynamicFlagFilter != 0)))
                        continue;                                          /* BuildDistanceFieldDistanceFunctionBuffer, :321-322 */
                    /* DistanceFunctionVertexShader, DistanceFunction.fx:16-26 (FUNCTION_SIZE_HACK 1) */
                    const float msize = fmaxf(fmaxf(fabsf(ob->Size[0]), fabsf(ob->Size[1])), fabsf(ob->Size[2])) + max_enc + 4.0f;
                    const float x0 = (ob->Center[0] - msize) * px_per_unit_x, x1 = (ob->Center[0] + msize) * px_per_unit_x;
                    const float y0 = (ob->Center[1] - msize) * px_per_unit_y, y1 = (ob->Center[1] + msize) * px_per_unit_y;
                    if (!((cxp >= x0) && (cxp < x1) && (cyp >= y0) && (cyp < y1)))
                        continue;
                    const f3 center = v3(ob->Center[0], ob->Center[1], ob->Center[2]);
                    const f3 size = v3(ob->Size[0], ob->Size[1], ob->Size[2]);
                    const f4 rot = v4(ob->Orientation[0], ob->Orientation[1], ob->Orientation[2], ob->Orientation[3]);
                    for (int k = 0; k < 4; k++) {
                        const float dist = evaluate_obstruction(ob->Type, v3(wx, wy, slice_z[k]), center, size, rot);
                        acc[k] = fmaxf(acc[k], orc_encode_distance(dist, max_enc));
                    }
                }

                for (int v = 0; v < volume_count; v++) {
                    const IlmHeightVolume* hv = &volumes[v];
                    if (d->DynamicFlagFilter >= 0 && ((hv->IsDynamic != 0) != (d->DynamicFlagFilter != 0)))
                        continue;                                          /* :205-206 */
                    if (hv->VertexCount < 1)
                        continue;
                    const float* P = polygon_xy + 2 * (size_t)hv->FirstVertex;
                    /* hv.Bounds.Expand(DistanceLimit, DistanceLimit), :216 */
                    float bx0 = P[0], bx1 = P[0], by0 = P[1], by1 = P[1];
                    for (int e = 1; e < hv->VertexCount; e++) {
                        bx0 = fminf(bx0, P[2 * e]); bx1 = fmaxf(bx1, P[2 * e]);
                        by0 = fminf(by0, P[2 * e + 1]); by1 = fmaxf(by1, P[2 * e + 1]);
                    }
                    const float x0 = (bx0 - ILM_DISTANCE_LIMIT) * px_per_unit_x, x1 = (bx1 + ILM_DISTANCE_LIMIT) * px_per_unit_x;
                    const float y0 = (by0 - ILM_DISTANCE_LIMIT) * px_per_unit_y, y1 = (by1 + ILM_DISTANCE_LIMIT) * px_per_unit_y;
                    if (!((cxp >= x0) && (cxp < x1) && (cyp >= y0) && (cyp < y1)))
                        continue;
                    /* computeSliceDistances, DistanceField.fx:75-99: every edge (p[j], p[j+1 wrapped]) exactly once */
                    float dist_sq = 999999.0f, sign = 1.0f;
                    for (int e = 0; e < hv->VertexCount; e++) {
                        const int n = (e + 1 == hv->VertexCount) ? 0 : e + 1;
                        sd_polygon_vertex(wx, wy, P[2 * n], P[2 * n + 1], P[2 * e], P[2 * e + 1], &dist_sq, &sign);
                    }
                    const float z0 = hv->ZBase, z1 = hv->ZBase + hv->Height;   /* zRange, :217 */
                    for (int k = 0; k < 4; k++)
                        acc[k] = fmaxf(acc[k], orc_encode_distance(final_eval(slice_z[k], z0, z1, dist_sq, sign), max_enc));
                }

                for (int k = 0; k < 4; k++) {
                    const uint16_t code = sdf_store_channel(acc[k], format);
                    texel[k] = code > cleared[k] ? code : cleared[k];      /* BlendFunction.Max on the stored value */
                }
            }
        }
    }
}


/* ---------------------------------------------------------------------------
 * G-buffer generation, non-2.5D (LightingRenderer.GBuffer.cs:127-219, GBuffer.fx:7-70, GBufferShaderCommon.fxh:10-35)
 * ------------------------------------------------------------------------- */
/* encodeNormalSpherical, EnvironmentCommon.fxh:33-40 */
static void encode_normal_spherical(f3 n, float out[2]) {
    if (fabsf(n.x) < 0.0001f)
        n.x = 0.0001f;
    out[0] = ((atan2f(n.y, n.x) / H_PI) + 1.0f) * 0.5f;
    out[1] = (n.z + 1.0f) * 0.5f;
}

/* encodeGBufferSample, GBufferShaderCommon.fxh:10-35 (dead = false, fullbright = false on this path) */
static f4 encode_gbuffer_sample(f3 normal, float relative_y, float z, int enable_shadows) {
    float enc[2] = { 0.0f, 0.0f };
    if ((normal.x != 0.0f) || (normal.y != 0.0f) || (normal.z != 0.0f))
        encode_normal_spherical(normal, enc);
    float w = (((z + 1024.0f) / 1024.0f) * (enable_shadows ? 1.0f : -1.0f)) + (enable_shadows ? 0.0f : -1.0f);
    return v4(enc[0], enc[1], relative_y, w);
}

/* even-odd crossing test of a pixel centre against a polygon (coverage of any triangulation of its interior) */
static int point_in_polygon(float px, float py, const float* P, int count) {
    int inside = 0;
    for (int e = 0; e < count; e++) {
        const int n = (e + 1 == count) ? 0 : e + 1;
        const float ax = P[2 * e], ay = P[2 * e + 1], bx = P[2 * n], by = P[2 * n + 1];
        if ((ay > py) != (by > py)) {
            const float xi = ((bx - ax) * (py - ay)) / (by - ay) + ax;
            if (px < xi) inside = !inside;
        }
    }
    return inside;
}

/* out: width * height float4 texels.  volumes must already be ordered lowest to highest top (OrderBy(ZBase + Height), :210). */
void orc_render_gbuffer(IlmFloat4* out, int32_t width, int32_t height, const IlmGBufferRenderDesc* d,
                        const IlmHeightVolume* volumes, int32_t volume_count, const float* polygon_xy) {
    const float ground_z = d->GroundZ + (d->RenderGroundPlane ? 0.0f : GB_GROUND_LIFT);     /* RenderGroundPlane, :271-286 */
    #pragma omp parallel for schedule(static)
    for (int j = 0; j < height; j++)
        for (int i = 0; i < width; i++) {
            /* inverse of (position.xy - ViewportPosition) * ViewportScale (GroundPlaneVertexShader, GBuffer.fx:15) at the pixel centre */
            const float wx = ((float)i + 0.5f) / d->ViewportScale[0] + d->ViewportPosition[0];
            const float wy = ((float)j + 0.5f) / d->ViewportScale[1] + d->ViewportPosition[1];
            f4 texel = v4(0, 0, 0, 0);                                  /* ClearBatch(Color.Transparent), :147-150 */
            if (!(ground_z < d->GroundZ))                               /* GroundPlanePixelShader, GBuffer.fx:63-66 */
                texel = encode_gbuffer_sample(v3(0, 0, 1), 0.0f, ground_z, d->EnableGroundShadows);
            for (int v = 0; v < volume_count; v++) {
                const IlmHeightVolume* hv = &volumes[v];
                if (hv->VertexCount < 3)
                    continue;
                if (!point_in_polygon(wx, wy, polygon_xy + 2 * (size_t)hv->FirstVertex, hv->VertexCount))
                    continue;
                const float top = hv->ZBase + hv->Height;               /* Mesh3D: Position.Z = h2, HeightVolume.cs:115-122 */
                if (top < d->GroundZ)
                    continue;                                           /* discard */
                texel = encode_gbuffer_sample(v3(0, 0, 1), 0.0f, top, hv->TopFaceEnableShadows);
            }
            out[(size_t)j * (size_t)width + (size_t)i] = texel;
        }
}
