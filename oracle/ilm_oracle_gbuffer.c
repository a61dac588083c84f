/*
 * ilm_oracle_gbuffer.c -- CPU restatement of RenderGBuffer with the host's meshes: 2.5D height volumes and billboards.
 * TEST INFRASTRUCTURE ONLY (included by ilm_oracle.c).  PARITY UNPINNED: the reference draws these triangle lists with the
 * Direct3D 9 rasteriser, whose sub-pixel snapping and attribute interpolation are the hardware's; this file fixes them to the
 * published Direct3D rule (pixel centres, top-left fill rule, 8 sub-pixel bits) and says so in orc_render_gbuffer_meshes.
 *
 * Follows Illuminant/Lighting/LightingRenderer.GBuffer.cs:102-203 (RenderGBuffer, _SetupGBufferGroundPlane), :205-269 (volumes),
 * :271-299 (ground plane), :330-478 (billboards); Illuminant/Shaders/GBuffer.fx; GBufferBitmap.fx; GBufferShaderCommon.fxh.
 */

enum { GB_GROUND = 0, GB_TOP = 1, GB_FACE = 2, GB_MASK = 3, GB_GDATA = 4 };
#define GB_ATTRS 12

typedef struct {
    int32_t x[3], y[3];                 /* 1/256-pixel positions; v0 v1 v2 clockwise on a y-down screen (area > 0) */
    int kind, texture;
    float a[3][GB_ATTRS];
} GbPrim;

/* round-to-nearest onto the 1/256-pixel grid; positions are confined to +-2^30 so that edge functions fit 64 bits */
static int32_t gb_snap(float s) {
    double v = floor((double)s * 256.0 + 0.5);
    if (!(v > -1073741824.0)) v = -1073741824.0;
    if (v > 1073741824.0) v = 1073741824.0;
    return (int32_t)v;
}

static int64_t gb_edge(int32_t ax, int32_t ay, int32_t bx, int32_t by, int64_t px, int64_t py) {
    return ((int64_t)bx - ax) * (py - ay) - ((int64_t)by - ay) * (px - ax);
}

/* Direct3D top-left rule on a clockwise triangle (y down): a sample on an edge belongs to the triangle when the edge is a top
 * edge (horizontal, running left to right) or a left edge (running upwards) */
static int gb_edge_owns(int32_t ax, int32_t ay, int32_t bx, int32_t by) {
    return (by < ay) || ((by == ay) && (bx > ax));
}

/* coverage of pixel (i, j)'s centre and the barycentric weights of v1 and v2 */
static int gb_cover(const GbPrim* p, int i, int j, float* f1, float* f2) {
    const int64_t px = 256 * (int64_t)i + 128, py = 256 * (int64_t)j + 128;
    const int64_t w0 = gb_edge(p->x[1], p->y[1], p->x[2], p->y[2], px, py);
    const int64_t w1 = gb_edge(p->x[2], p->y[2], p->x[0], p->y[0], px, py);
    const int64_t w2 = gb_edge(p->x[0], p->y[0], p->x[1], p->y[1], px, py);
    if ((w0 < 0) || (w1 < 0) || (w2 < 0)) return 0;
    if ((w0 == 0) && !gb_edge_owns(p->x[1], p->y[1], p->x[2], p->y[2])) return 0;
    if ((w1 == 0) && !gb_edge_owns(p->x[2], p->y[2], p->x[0], p->y[0])) return 0;
    if ((w2 == 0) && !gb_edge_owns(p->x[0], p->y[0], p->x[1], p->y[1])) return 0;
    const double area = (double)(w0 + w1 + w2);
    *f1 = (float)((double)w1 / area);
    *f2 = (float)((double)w2 / area);
    return 1;
}

static float gb_lerp(const GbPrim* p, int k, float f1, float f2) {
    return (p->a[0][k] + (p->a[1][k] - p->a[0][k]) * f1) + (p->a[2][k] - p->a[0][k]) * f2;
}

/* finishes a triangle from three screen positions; returns 0 for a degenerate one.  A counter-clockwise triangle is drawn too
 * (RenderStates.ScissorOnly culls nothing): its second and third vertex change places. */
static int gb_finish(GbPrim* p, const float sx[3], const float sy[3]) {
    for (int k = 0; k < 3; k++) { p->x[k] = gb_snap(sx[k]); p->y[k] = gb_snap(sy[k]); }
    const int64_t area = gb_edge(p->x[0], p->y[0], p->x[1], p->y[1], p->x[2], p->y[2]);
    if (area == 0) return 0;
    if (area < 0) {
        int32_t t = p->x[1]; p->x[1] = p->x[2]; p->x[2] = t;
        t = p->y[1]; p->y[1] = p->y[2]; p->y[2] = t;
        for (int k = 0; k < GB_ATTRS; k++) { float f = p->a[1][k]; p->a[1][k] = p->a[2][k]; p->a[2][k] = f; }
    }
    return 1;
}

/* GroundPlaneVertexShader / HeightVolumeVertexShader / HeightVolumeFaceVertexShader, GBuffer.fx:7-55.
 * attributes: 0-2 worldPosition, 3-5 normal, 6 enableShadows, 7 result.z, 8 dead */
static int gb_volume_prim(GbPrim* p, int kind, const IlmHeightVolumeVertex* v[3], const IlmGBufferMeshDesc* d) {
    float sx[3], sy[3];
    p->kind = kind; p->texture = -1;
    for (int k = 0; k < 3; k++) {
        float x = v[k]->Position[0], y = v[k]->Position[1];
        const float z = v[k]->Position[2];
        memset(p->a[k], 0, sizeof(p->a[k]));
        p->a[k][0] = x; p->a[k][1] = y; p->a[k][2] = z;
        p->a[k][3] = v[k]->Normal[0]; p->a[k][4] = v[k]->Normal[1]; p->a[k][5] = v[k]->Normal[2];
        p->a[k][6] = v[k]->EnableShadows;
        if (kind == GB_GROUND) {
            p->a[k][7] = 0.0f;                                   /* result.z = 0, :16 */
            p->a[k][8] = (z < -9999.0f) ? 1.0f : 0.0f;           /* dead, :17 */
        } else {
            y -= d->ZToYMultiplier * z;                          /* position.y -= getZToYMultiplier() * position.z, :31,:49 */
            p->a[k][7] = z / d->DistanceFieldExtentZ;            /* result.z, :34,:52 */
        }
        sx[k] = (x - d->ViewportPosition[0]) * d->ViewportScale[0];
        sy[k] = (y - d->ViewportPosition[1]) * d->ViewportScale[1];
    }
    return gb_finish(p, sx, sy);
}

/* BillboardVertexShader, GBufferBitmap.fx:12-27.  attributes: 0-2 worldPosition, 3-5 normal, 6-7 texCoord, 8 screenPosition.y,
 * 9-10 dataScaleAndDynamicFlag.  The vertex declaration gives POSITION0 two floats (Vertices.cs:89), so position.z reads 0. */
static int gb_billboard_prim(GbPrim* p, int kind, int texture, const IlmBillboardVertex* v[3], const IlmGBufferMeshDesc* d) {
    float sx[3], sy[3];
    p->kind = kind; p->texture = texture;
    for (int k = 0; k < 3; k++) {
        memset(p->a[k], 0, sizeof(p->a[k]));
        for (int c = 0; c < 3; c++) {
            p->a[k][c] = v[k]->WorldPosition[c] + (d->SelfOcclusionHack * v[k]->Normal[c]);      /* :21 */
            p->a[k][3 + c] = v[k]->Normal[c];
        }
        p->a[k][6] = v[k]->TexCoord[0]; p->a[k][7] = v[k]->TexCoord[1];
        p->a[k][8] = v[k]->ScreenPosition[1];
        p->a[k][9] = v[k]->DataScaleAndDynamicFlag[0]; p->a[k][10] = v[k]->DataScaleAndDynamicFlag[1];
        sx[k] = (v[k]->ScreenPosition[0] - d->ViewportPosition[0]) * d->ViewportScale[0];
        sy[k] = (v[k]->ScreenPosition[1] - d->ViewportPosition[1]) * d->ViewportScale[1];
    }
    return gb_finish(p, sx, sy);
}

/* encodeGBufferSample, GBufferShaderCommon.fxh:10-35 (fullbright = false) */
static f4 gb_encode(f3 normal, float relative_y, float z, int dead, int enable_shadows) {
    if (dead)
        return v4(0.0f, 0.0f, -GB_DEAD_TEXEL, -GB_DEAD_TEXEL);
    return encode_gbuffer_sample(normal, relative_y, z, enable_shadows);
}

/* tex2D through the POINT / CLAMP sampler _SetTextureForGBufferBillboard installs (:301-307); an unbound stage reads (0, 0, 0, 1) */
static f4 gb_sample(const OrcTexture* t, float u, float v) {
    if (!t || !t->texels)
        return v4(0.0f, 0.0f, 0.0f, 1.0f);
    const float fx = floorf(u * (float)t->width), fy = floorf(v * (float)t->height);
    const int x = !(fx >= 0.0f) ? 0 : ((fx > (float)(t->width - 1)) ? t->width - 1 : (int)fx);
    const int y = !(fy >= 0.0f) ? 0 : ((fy > (float)(t->height - 1)) ? t->height - 1 : (int)fy);
    const size_t o = (size_t)y * (size_t)t->width + (size_t)x;
    if (t->format == ILM_LIGHTMAP_RGBA8) {
        const uint8_t* b = (const uint8_t*)t->texels + 4 * o;
        return v4((float)b[0] / 255.0f, (float)b[1] / 255.0f, (float)b[2] / 255.0f, (float)b[3] / 255.0f);
    }
    if (t->format == ILM_LIGHTMAP_HALF4) {
        const uint16_t* hh = (const uint16_t*)t->texels + 4 * o;
        return v4(half_to_float(hh[0]), half_to_float(hh[1]), half_to_float(hh[2]), half_to_float(hh[3]));
    }
    return ((const f4*)t->texels)[o];
}

/* the five pixel shaders; returns 0 when the fragment is discarded */
static int gb_shade(const GbPrim* p, float f1, float f2, const IlmGBufferMeshDesc* d, const OrcTexture* textures, f4* out) {
    const f3 wp = v3(gb_lerp(p, 0, f1, f2), gb_lerp(p, 1, f1, f2), gb_lerp(p, 2, f1, f2));
    const f3 n = v3(gb_lerp(p, 3, f1, f2), gb_lerp(p, 4, f1, f2), gb_lerp(p, 5, f1, f2));
    if (p->kind == GB_GROUND) {                                   /* GroundPlanePixelShader, GBuffer.fx:57-70 */
        if (wp.z < d->GroundZ) return 0;
        *out = gb_encode(v3(0, 0, 1), 0.0f, wp.z, gb_lerp(p, 8, f1, f2) != 0.0f, gb_lerp(p, 6, f1, f2) > 0.5f);
        return 1;
    }
    if ((p->kind == GB_TOP) || (p->kind == GB_FACE)) {            /* HeightVolumePixelShader :72-85, HeightVolumeFacePixelShader :87-103 */
        f3 bias = v3(0.0f, 0.0f, d->ZSelfOcclusionHack);
        if (p->kind == GB_FACE) {
            if (wp.z < d->GroundZ) return 0;
            bias = v3mul(v3(d->SelfOcclusionHack, d->SelfOcclusionHack, d->ZSelfOcclusionHack), n);
        }
        /* float2 expression assigned to a float: the x components (GetViewportScale().x, getEnvironmentRenderScale().x) */
        const float relative_y = (((wp.z * d->ZToYMultiplier) * d->ViewportScale[0]) / d->RenderScale[0]) + bias.y;
        *out = gb_encode(n, relative_y, wp.z + bias.z, 0, gb_lerp(p, 6, f1, f2) > 0.5f);
        return 1;
    }
    const f4 data = gb_sample((p->texture >= 0) ? &textures[p->texture] : NULL, gb_lerp(p, 6, f1, f2), gb_lerp(p, 7, f1, f2));
    const float data_scale = gb_lerp(p, 9, f1, f2);
    if (p->kind == GB_MASK) {                                     /* MaskBillboardPixelShader, GBufferBitmap.fx:29-59 */
        const float discard_threshold = GB_MASK_DISCARD_NUMERATOR / 255.0f;
        if ((data.w - discard_threshold) < 0.0f) return 0;        /* clip() */
        const float relative_y = (wp.y - gb_lerp(p, 8, f1, f2)) * data_scale;
        *out = v4((n.x / 2.0f) + 0.5f, (n.z / 2.0f) + 0.5f, relative_y,
                  ((wp.z + GBUFFER_Z_OFFSET) / GBUFFER_Z_SCALE) * gb_lerp(p, 10, f1, f2));
        return 1;
    }
    /* GDataBillboardPixelShader, GBufferBitmap.fx:61-113 */
    const float discard_threshold = GB_GDATA_DISCARD_NUMERATOR / 255.0f;
    if (data.w < discard_threshold) return 0;
    const float tx = (data.x - 0.5f) * 2.0f, ty = (data.y - 0.5f) * 2.0f;
    const float tz = sqrtf(1.0f - (tx * tx + ty * ty));
    /* tangent (1,0,0) * tx + bitangent (0,-1,0) * ty + normal (0,0,1) * tz, summed left to right per component */
    const f3 world_normal = v3((1.0f * tx + 0.0f * ty) + 0.0f * tz, (0.0f * tx + -1.0f * ty) + 0.0f * tz, (0.0f * tx + 0.0f * ty) + 1.0f * tz);
    const f3 result_normal = v3norm(world_normal);
    const float effective_z = wp.z + (data.z * data_scale);
    const float y_offset = effective_z * d->ZToYMultiplier;
    *out = gb_encode(result_normal, y_offset, effective_z, 0, 1);
    return 1;
}

static void gb_draw(IlmFloat4* out, uint32_t* depth, int32_t width, int32_t height, const GbPrim* p, const IlmGBufferMeshDesc* d,
                    const OrcTexture* textures) {
    int32_t x0 = p->x[0], x1 = p->x[0], y0 = p->y[0], y1 = p->y[0];
    for (int k = 1; k < 3; k++) {
        if (p->x[k] < x0) x0 = p->x[k];
        if (p->x[k] > x1) x1 = p->x[k];
        if (p->y[k] < y0) y0 = p->y[k];
        if (p->y[k] > y1) y1 = p->y[k];
    }
    /* pixel centres 256 i + 128 inside [x0, x1] */
    int64_t i0 = ((int64_t)x0 - 128 + 255) >> 8, i1 = ((int64_t)x1 - 128) >> 8;
    int64_t j0 = ((int64_t)y0 - 128 + 255) >> 8, j1 = ((int64_t)y1 - 128) >> 8;
    if (i0 < 0) i0 = 0;
    if (j0 < 0) j0 = 0;
    if (i1 > width - 1) i1 = width - 1;
    if (j1 > height - 1) j1 = height - 1;
    const int depth_tested = (p->kind == GB_TOP) || (p->kind == GB_FACE);
    #pragma omp parallel for schedule(static)
    for (int64_t j = j0; j <= j1; j++)
        for (int64_t i = i0; i <= i1; i++) {
            float f1, f2;
            if (!gb_cover(p, (int)i, (int)j, &f1, &f2))
                continue;
            /* clipped against the near / far plane (w = 1) -- volumes only: BillboardVertex's POSITION0 is a Vector2 (Vertices.cs:89), so
             * BillboardVertexShader's result.z = position.z / DistanceFieldExtent.z (GBufferBitmap.fx:12-27) is 0; attribute 7 of a
             * billboard is TexCoord.y, which the CLAMP sampler, not the clipper, brings back into range */
            const int is_billboard = (p->kind == GB_MASK) || (p->kind == GB_GDATA);
            const float z = is_billboard ? 0.0f : gb_lerp(p, 7, f1, f2);
            if (!((z >= 0.0f) && (z <= 1.0f)))
                continue;
            f4 texel;
            if (!gb_shade(p, f1, f2, d, textures, &texel))
                continue;
            const size_t o = (size_t)j * (size_t)width + (size_t)i;
            if (depth_tested) {
                /* DepthFormat.Depth24 (GBuffer.cs:37), CompareFunction.GreaterEqual with writes (LightingRenderer.cs:539-551) */
                const uint32_t d24 = (uint32_t)floor((double)z * 16777215.0 + 0.5);
                if (!(d24 >= depth[o]))
                    continue;
                depth[o] = d24;
            }
            out[o] = texel;
        }
}

/* RenderGBuffer in draw order (see ilm_gbuffer_render_meshes in include/illuminant_hip.h).  `textures`: one per run (texels NULL =
 * nothing bound), format ILM_LIGHTMAP_*.  Rasterisation rule (not the reference's text -- the hardware's): a triangle covers the
 * pixels whose centres lie inside it under the top-left rule, on positions rounded to 1/256 pixel; attributes vary linearly over
 * the snapped triangle as a0 + (a1 - a0) f1 + (a2 - a0) f2. */
void orc_render_gbuffer_meshes(IlmFloat4* out, int32_t width, int32_t height, const IlmGBufferMeshDesc* d,
                               const IlmHeightVolumeVertex* top, int32_t top_count,
                               const IlmHeightVolumeVertex* front, int32_t front_count,
                               const IlmBillboardVertex* billboards, int32_t billboard_vertex_count,
                               const IlmBillboardRun* runs, int32_t run_count, const OrcTexture* textures) {
    static const int quad_indices[6] = { 0, 1, 3, 1, 2, 3 };     /* QuadIndices, LightingRenderer.cs:421-423 */
    const size_t texels = (size_t)width * (size_t)height;
    uint32_t* depth = (uint32_t*)calloc(texels ? texels : 1, sizeof(uint32_t));        /* ClearBatch(Color.Transparent, clearZ: 0), :147-150 */
    memset(out, 0, texels * sizeof(IlmFloat4));
    GbPrim p;
    (void)billboard_vertex_count;

    /* RenderGroundPlane, :271-299 */
    {
        const float lift = d->RenderGroundPlane ? 0.0f : GB_GROUND_LIFT;
        const float gz = d->GroundZ + lift;
        IlmHeightVolumeVertex g[4];
        const float e = GB_GROUND_HALF_EXTENT;
        const float cx[4] = { -e, e, e, -e }, cy[4] = { -e, -e, e, e };
        for (int k = 0; k < 4; k++) {
            g[k].Position[0] = cx[k]; g[k].Position[1] = cy[k]; g[k].Position[2] = gz;
            g[k].Normal[0] = 0.0f; g[k].Normal[1] = 0.0f; g[k].Normal[2] = 1.0f;
            g[k].ZRange[0] = d->GroundZ; g[k].ZRange[1] = d->GroundZ;
            g[k].EnableShadows = d->EnableGroundShadows ? 1.0f : 0.0f;
        }
        for (int t = 0; t < 2; t++) {
            const IlmHeightVolumeVertex* v[3] = { &g[quad_indices[3 * t]], &g[quad_indices[3 * t + 1]], &g[quad_indices[3 * t + 2]] };
            if (gb_volume_prim(&p, GB_GROUND, v, d))
                gb_draw(out, depth, width, height, &p, d, textures);
        }
    }
    /* RenderTwoPointFiveDVolumes :221-269 (top batch on layer 3, front batch on layer 5) / RenderGBufferVolumes :205-219 */
    for (int t = 0; t + 2 < top_count; t += 3) {
        const IlmHeightVolumeVertex* v[3] = { &top[t], &top[t + 1], &top[t + 2] };
        if (gb_volume_prim(&p, d->TwoPointFiveD ? GB_TOP : GB_GROUND, v, d))
            gb_draw(out, depth, width, height, &p, d, textures);
    }
    for (int t = 0; d->TwoPointFiveD && (t + 2 < front_count); t += 3) {
        const IlmHeightVolumeVertex* v[3] = { &front[t], &front[t + 1], &front[t + 2] };
        if (gb_volume_prim(&p, GB_FACE, v, d))
            gb_draw(out, depth, width, height, &p, d, textures);
    }
    /* RenderGBufferBillboards :330-478: the mask batch sits one layer below the g-data batch */
    for (int type = ILM_BILLBOARD_MASK; type <= ILM_BILLBOARD_GBUFFER_DATA; type++)
        for (int r = 0; r < run_count; r++) {
            if (runs[r].Type != type)
                continue;
            for (int q = runs[r].FirstQuad; q < runs[r].FirstQuad + runs[r].QuadCount; q++)
                for (int t = 0; t < 2; t++) {
                    const IlmBillboardVertex* b = billboards + 4 * (size_t)q;
                    const IlmBillboardVertex* v[3] = { &b[quad_indices[3 * t]], &b[quad_indices[3 * t + 1]], &b[quad_indices[3 * t + 2]] };
                    if (gb_billboard_prim(&p, (type == ILM_BILLBOARD_MASK) ? GB_MASK : GB_GDATA, r, v, d))
                        gb_draw(out, depth, width, height, &p, d, textures);
                }
        }
    free(depth);
}
