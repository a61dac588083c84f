/*
 * ilm_oracle_transforms.c -- CPU restatement of the remaining particle techniques:
 * MatrixMultiply, SpatialNoise, SpawnParticlesFromPositionTexture, SpawnFeedbackParticles.
 * TEST INFRASTRUCTURE ONLY (see ilm_oracle.h).  PARITY UNPINNED.
 * Textually included by ilm_oracle.c before orc_step (it shares that file's static HLSL helpers).
 */

/* mul3, ParticleCommon.fxh:183-196 */
static f4 mul3(f4 old_value, const IlmMatrix* mat, float w) {
    f4 temp = mul_point(xyz(old_value), mat);
    f3 divided;
    if (w != 0.0f)
        divided = v3(temp.x / temp.w, temp.y / temp.w, temp.z / temp.w);
    else
        divided = xyz(temp);
    return v4(divided.x, divided.y, divided.z, old_value.w);
}

/* PS_MatrixMultiply, MatrixMultiply.fx:22-52 (computeWeight uses clamp(x, 0, 1), :14-20) */
static void matrix_multiply_slot(f4* pos, f4* vel, const IlmParticleSystemUniforms* sys, const IlmMatrixMultiplyParams* p) {
    f4 old_position = *pos, old_velocity = *vel;
    if ((old_position.w <= 0.0f) || !check_category_filter(old_velocity.w, p->Area.CategoryFilter))
        return;
    float time_scale = (p->TimeDivisor >= 0.0f) ? sys_dt(sys) / p->TimeDivisor : 1.0f;
    float distance = evaluate_by_type_id(p->Area.AreaType, xyz(old_position),
        v3(p->Area.AreaCenter[0], p->Area.AreaCenter[1], p->Area.AreaCenter[2]),
        v3(p->Area.AreaSize[0], p->Area.AreaSize[1], p->Area.AreaSize[2]), p->Area.AreaRotation);
    float w = ((1.0f - h_clamp(distance / p->Area.AreaFalloff, 0.0f, 1.0f)) * p->Area.Strength) * time_scale;
    *pos = v4lerp(old_position, mul3(old_position, &p->PositionMatrix, 1.0f), w);
    *vel = v4lerp(old_velocity, mul3(old_velocity, &p->VelocityMatrix, 0.0f), w);
}

static void matrix_multiply_rows(IlmFloat4* pos, IlmFloat4* vel, int32_t chunk_size, int y0, int y1,
                                 const IlmParticleSystemUniforms* sys, const IlmMatrixMultiplyParams* p) {
    for (int i = y0 * chunk_size; i < y1 * chunk_size; i++)
        matrix_multiply_slot(&pos[i], &vel[i], sys, p);
}

void orc_matrix_multiply(IlmFloat4* pos, IlmFloat4* vel, int32_t chunk_size,
                         const IlmParticleSystemUniforms* sys, const IlmMatrixMultiplyParams* p) {
    #pragma omp parallel for schedule(static)
    for (int y = 0; y < chunk_size; y++)
        matrix_multiply_rows(pos, vel, chunk_size, y, y + 1, sys, p);
}

/* The Rgba64 copy of the randomness table: new Rgba64(Vector4) (ParticleEngine.cs:536-538) = per channel
 * round(clamp(v, 0, 1) * 65535) with Math.Round's round-half-to-even (XNA PackUtils.PackUNorm). */
void orc_low_precision_randomness(const IlmFloat4* rnd, int32_t count, uint16_t* out) {
    const float* f = (const float*)rnd;
    for (int64_t i = 0; i < (int64_t)count * 4; i++)
        out[i] = (uint16_t)nearbyintf(h_clamp(f[i], 0.0f, 1.0f) * 65535.0f);
}

/* smoothRandomCustom, RandomCommon.fxh:36-39: LINEAR min/mag, WRAP on both axes, texel centres at +0.5 */
static f4 smooth_random_custom(const uint16_t* lp, int rw, int rh, float x, float y, const float offset[2], float rate_x, float rate_y) {
    const float texel_x = 1.0f / (float)rw, texel_y = 1.0f / (float)rh;   /* RandomnessTexel */
    float u = ((x * rate_x) + offset[0]) * texel_x;
    float v = ((y * rate_y) + offset[1]) * texel_y;
    float sx = u * (float)rw - 0.5f, sy = v * (float)rh - 0.5f;
    float x0f = floorf(sx), y0f = floorf(sy);
    float fx = sx - x0f, fy = sy - y0f;
    int x0 = wrap_index(x0f, rw), x1 = wrap_index(x0f + 1.0f, rw);
    int y0 = wrap_index(y0f, rh), y1 = wrap_index(y0f + 1.0f, rh);
    float r[4];
    for (int c = 0; c < 4; c++) {
        float t00 = (float)lp[((size_t)y0 * rw + x0) * 4 + c] / 65535.0f, t10 = (float)lp[((size_t)y0 * rw + x1) * 4 + c] / 65535.0f;
        float t01 = (float)lp[((size_t)y1 * rw + x0) * 4 + c] / 65535.0f, t11 = (float)lp[((size_t)y1 * rw + x1) * 4 + c] / 65535.0f;
        r[c] = h_lerp(h_lerp(t00, t10, fx), h_lerp(t01, t11, fx), fy);
    }
    return v4(r[0], r[1], r[2], r[3]);
}

/* PS_SpatialNoise, Noise.fx:74-116 */
static void spatial_noise_slot(f4* pos, f4* vel, const uint16_t* lp, int rw, int rh,
                               const IlmParticleSystemUniforms* sys, const IlmSpatialNoiseParams* sp) {
    const IlmNoiseParams* p = &sp->Noise;
    f4 old_position = *pos, old_velocity = *vel;
    if (!check_category_filter(old_velocity.w, p->Area.CategoryFilter))
        return;
    float weight = compute_weight(&p->Area, xyz(old_position));
    float t = weight * sys_dt(sys) / p->TimeDivisor;

    float rx = old_position.x, ry = old_position.y;
    f4 random_p1 = smooth_random_custom(lp, rw, rh, rx, ry, p->RandomnessOffset, sp->SpaceScale[0], sp->SpaceScale[1]);
    f4 random_p2 = smooth_random_custom(lp, rw, rh, rx, ry, p->NextRandomnessOffset, sp->SpaceScale[0], sp->SpaceScale[1]);
    f4 random_v1 = smooth_random_custom(lp, rw, rh, rx + 2.0f, ry + 1.0f, p->RandomnessOffset, sp->SpaceScale[0], sp->SpaceScale[1]);
    f4 random_v2 = smooth_random_custom(lp, rw, rh, rx + 2.0f, ry + 1.0f, p->NextRandomnessOffset, sp->SpaceScale[0], sp->SpaceScale[1]);
    f4 random_p = v4lerp(random_p1, random_p2, p->FrequencyLerp);
    f4 random_v = v4lerp(random_v1, random_v2, p->FrequencyLerp);

    f4 position_delta = v4mul(v4add(random_p, p->PositionOffset), p->PositionScale);
    f4 velocity_delta = v4mul(v4add(random_v, p->VelocityOffset), p->VelocityScale);

    *pos = v4lerp(old_position, v4add(old_position, position_delta), t);
    f3 ov = xyz(old_velocity), nv;
    if (p->ReplaceOldVelocity != 0.0f)
        nv = v3(h_lerp(ov.x, velocity_delta.x, weight), h_lerp(ov.y, velocity_delta.y, weight), h_lerp(ov.z, velocity_delta.z, weight));
    else
        nv = v3(h_lerp(ov.x, ov.x + velocity_delta.x, t), h_lerp(ov.y, ov.y + velocity_delta.y, t), h_lerp(ov.z, ov.z + velocity_delta.z, t));
    nv = v3add(nv, v3scale(v3norm(ov), velocity_delta.w));
    *vel = v4(nv.x, nv.y, nv.z, old_velocity.w);
}

static void spatial_noise_rows(IlmFloat4* pos, IlmFloat4* vel, int32_t chunk_size, int y0, int y1, const uint16_t* lp, int rw, int rh,
                               const IlmParticleSystemUniforms* sys, const IlmSpatialNoiseParams* p) {
    for (int i = y0 * chunk_size; i < y1 * chunk_size; i++)
        spatial_noise_slot(&pos[i], &vel[i], lp, rw, rh, sys, p);
}

void orc_spatial_noise(IlmFloat4* pos, IlmFloat4* vel, int32_t chunk_size, const uint16_t* low_precision_rnd, int32_t rw, int32_t rh,
                       const IlmParticleSystemUniforms* sys, const IlmSpatialNoiseParams* p) {
    #pragma omp parallel for schedule(static)
    for (int y = 0; y < chunk_size; y++)
        spatial_noise_rows(pos, vel, chunk_size, y, y + 1, low_precision_rnd, rw, rh, sys, p);
}

/* evaluateRandomForIndex, SpawnerCommon.fxh:106-117 */
static void evaluate_random_for_index(const f4* rnd, int rw, int rh, float index, const float offset[2], float align_velocity_and_position,
                                      f4* r1, f4* r2, f4* r3) {
    *r1 = random_custom(rnd, rw, rh, fmodf(index, SP_RANDOM1_X_MODULUS), 0.0f + fmodf(index, SP_RANDOM1_Y_MODULUS), offset, 1.0f, 1.0f);
    *r2 = random_custom(rnd, rw, rh, fmodf(index, SP_RANDOM2_X_MODULUS), 1.0f + fmodf(index, SP_RANDOM2_Y_MODULUS), offset, 1.0f, 1.0f);
    *r3 = random_custom(rnd, rw, rh, fmodf(index, SP_RANDOM3_X_MODULUS), 2.0f + fmodf(index, SP_RANDOM3_Y_MODULUS), offset, 1.0f, 1.0f);
    /* "The x and y element of random samples determines the normal", :114-116 -- part of evaluateRandomForIndex, so every spawner
     * technique that calls it (Spawn_Stage1, PS_SpawnFeedback SpawnParticles.fx:83, PS_SpawnPattern PatternSpawner.fx:63) aligns */
    if (align_velocity_and_position != 0.0f) { r2->x = r1->x; r2->y = r1->y; }
}

/* tex2Dlod(PositionConstantSampler, index * PositionConstantTexel.x): POINT, CLAMP, on the Spawner's PositionBuffer whose
 * width is the count rounded up to a multiple of 128 (ParticleSpawner.cs:301-314, SpawnParticles.fx:47-48) */
static f4 position_constant_fetch(const IlmFloat4* positions, int count, int index) {
    const int width = (count + 127) / 128 * 128;
    const float texel = 1.0f / (float)width;
    float u = (float)index * texel;
    int tx = (int)floorf(u * (float)width);
    if (tx < 0) tx = 0;
    if (tx > width - 1) tx = width - 1;
    if (tx >= count)
        return v4(0, 0, 0, 0);              /* the padding of Temp4 stays zero */
    return positions[tx];
}

/* PS_SpawnFromPositionTexture, SpawnParticles.fx:32-52: Spawn_Stage1 + Spawn_Stage2 around the texture fetch */
static void spawn_position_buffer_slot(f4* pos, f4* vel, f4* attr, float x, float y, const f4* rnd, int rw, int rh,
                                       const IlmSpawnParams* p, const IlmFloat4* positions, int position_count) {
    const float* csi = p->ChunkSizeAndIndices;
    float index = x + (y * csi[0]);
    if ((index < csi[1]) || (index > csi[2]))
        return;
    int index1, index2;
    float position_index_t;
    float relative_index = index - csi[1];
    if (p->PolygonRate > 0.05f) {
        float position_index_f = (relative_index / p->PolygonRate) + csi[3];
        float divisor = p->PositionConstantCount;
        float position_index_i;
        position_index_t = modff(position_index_f, &position_index_i);
        index1 = (int)fmodf(position_index_i, divisor);
        if (p->PolygonLoop != 0.0f)
            index2 = (int)fmodf(position_index_i + 1.0f, divisor);
        else
            index2 = (int)fminf((float)(index1 + 1), divisor - 1.0f);
    } else {
        index1 = index2 = (int)fmodf(relative_index + csi[3], p->PositionConstantCount);
        position_index_t = 0.0f;
    }
    /* Spawn_Stage2 (SpawnerCommon.fxh:162-190) on the two fetched constants */
    f4 random1, random2, random3;
    evaluate_random_for_index(rnd, rw, rh, index, p->RandomnessOffset, p->AlignVelocityAndPosition, &random1, &random2, &random3);
    f4 position1 = position_constant_fetch(positions, position_count, index1), position2 = position_constant_fetch(positions, position_count, index2);
    f4 position_constant = v4lerp(position1, position2, position_index_t);
    f4 towards_next = v4sub(position2, position1);
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
        f4 c8c = v4(C[8].x, C[8].x, C[8].x, C[8].x), c8s = v4(C[8].y, C[8].y, C[8].y, C[8].y),
           c8o = v4(C[8].z, C[8].z, C[8].z, C[8].z), r3w = v4(random3.w, random3.w, random3.w, random3.w);
        float towards_speed = evaluate_formula(zero, c8c, c8s, c8o, r3w, p->FormulaTypes[3], p->AxisMask).x;
        temp_velocity = v4add(temp_velocity, v4scale(v4(towards_next.x / towards_distance, towards_next.y / towards_distance,
                                                        towards_next.z / towards_distance, towards_next.w / towards_distance), towards_speed));
    }
    f4 new_velocity = mul_point(xyz(temp_velocity), &p->VelocityMatrix);
    new_velocity.w = temp_velocity.w;
    if (new_attributes.w < p->AttributeDiscardThreshold)
        return;
    *pos = new_position; *vel = new_velocity; *attr = new_attributes;
}

/* PS_SpawnFeedback, SpawnParticles.fx:54-118.  The source chunk has the target's chunk size
 * (SourceChunkSizeAndTexel = (size, 1/size, 1/size), ParticleTransform.cs:129-141). */
static void spawn_feedback_slot(f4* pos, f4* vel, f4* attr, float x, float y, const f4* rnd, int rw, int rh,
                                const IlmSpawnParams* p, const IlmFeedbackParams* fb,
                                const IlmFloat4* src_pos, const IlmFloat4* src_vel, const IlmFloat4* src_attr, int source_chunk_size) {
    const float* csi = p->ChunkSizeAndIndices;
    float index = x + (y * csi[0]);
    if ((index < csi[1]) || (index > csi[2]))
        return;
    const float size = (float)source_chunk_size, texel = 1.0f / (float)source_chunk_size;
    float source_index = ((index - csi[1]) / fb->InstanceMultiplier) + fb->FeedbackSourceIndex;
    float source_y;
    float source_x = modff(source_index / size, &source_y) * size;
    /* readStateUv: POINT / CLAMP at uv = sourceXy * texel */
    int tx = (int)floorf((source_x * texel) * size), ty = (int)floorf((source_y * texel) * size);
    if (tx < 0) tx = 0; if (tx > source_chunk_size - 1) tx = source_chunk_size - 1;
    if (ty < 0) ty = 0; if (ty > source_chunk_size - 1) ty = source_chunk_size - 1;
    const int si = ty * source_chunk_size + tx;
    f4 source_position = src_pos[si], source_velocity = src_vel[si], source_attributes = src_attr[si];
    if ((source_position.w <= fb->SourceLifeRange[0]) || (source_position.w >= fb->SourceLifeRange[1]))
        return;

    f4 random1, random2, random3;
    evaluate_random_for_index(rnd, rw, rh, index, p->RandomnessOffset, p->AlignVelocityAndPosition, &random1, &random2, &random3);

    const f4 zero = v4(0, 0, 0, 0);
    const f4* C = p->Configuration;
    f4 position_constant = p->InlinePositionConstants[0];
    if (fb->AlignPositionConstant != 0.0f) {
        position_constant.x += source_position.x; position_constant.y += source_position.y; position_constant.z += source_position.z;
    }
    f4 temp_position = evaluate_formula(zero, position_constant, C[0], C[1], random1, p->FormulaTypes[0], p->AxisMask);
    f4 attribute_constant = C[5];
    if (fb->MultiplyAttributeConstant != 0.0f)
        attribute_constant = v4mul(attribute_constant, source_attributes);
    f4 new_position = mul_point(xyz(temp_position), &p->PositionMatrix);
    new_position.w = temp_position.w;
    if (fb->MultiplyLife != 0.0f)
        new_position.w *= source_position.w;
    f4 temp_velocity = evaluate_formula(temp_position, C[2], C[3], C[4], random2, p->FormulaTypes[1], p->AxisMask);
    temp_velocity = v4add(temp_velocity, v4scale(source_velocity, fb->SourceVelocityFactor));
    f4 new_velocity = mul_point(xyz(temp_velocity), &p->VelocityMatrix);
    new_velocity.w = temp_velocity.w;
    f4 new_attributes = evaluate_formula(temp_position, attribute_constant, C[6], C[7], random3, p->FormulaTypes[2], p->AxisMask);
    if (new_attributes.w < p->AttributeDiscardThreshold)
        return;
    *pos = new_position; *vel = new_velocity; *attr = new_attributes;
}

/* tex2Dlod(PatternSampler, float4(uv, 0, lod)), PatternSpawner.fx:11-19,58-60: CLAMP addressing, LINEAR min / mag filter, POINT mip
 * filter.  The explicit LOD picks the nearest level, floor(lod + 0.5) clamped to the chain; bilinear weights come from
 * uv * size - 0.5 (texel centres at integer + 0.5).  The mip chain itself is an input (the reference's texture loader makes it). */
static f4 pattern_fetch(const IlmFloat4* tex, int w, int h, int levels, float u, float v, float lod) {
    int level = (int)floorf(lod + 0.5f);
    if (level < 0) level = 0;
    if (level > levels - 1) level = levels - 1;
    int lw = w, lh = h;
    for (int l = 0; l < level; l++) {
        tex += lw * lh;
        lw = lw >> 1; if (lw < 1) lw = 1;
        lh = lh >> 1; if (lh < 1) lh = 1;
    }
    float sx = u * (float)lw - 0.5f, sy = v * (float)lh - 0.5f;
    float x0f = floorf(sx), y0f = floorf(sy);
    float fx = sx - x0f, fy = sy - y0f;
    float x1f = x0f + 1.0f, y1f = y0f + 1.0f;
    if (x0f < 0.0f) x0f = 0.0f; if (x0f > (float)(lw - 1)) x0f = (float)(lw - 1);
    if (x1f < 0.0f) x1f = 0.0f; if (x1f > (float)(lw - 1)) x1f = (float)(lw - 1);
    if (y0f < 0.0f) y0f = 0.0f; if (y0f > (float)(lh - 1)) y0f = (float)(lh - 1);
    if (y1f < 0.0f) y1f = 0.0f; if (y1f > (float)(lh - 1)) y1f = (float)(lh - 1);
    const int x0 = (int)x0f, x1 = (int)x1f, y0 = (int)y0f, y1 = (int)y1f;
    return v4lerp(v4lerp(tex[y0 * lw + x0], tex[y0 * lw + x1], fx), v4lerp(tex[y1 * lw + x0], tex[y1 * lw + x1], fx), fy);
}

/* PS_SpawnPattern, PatternSpawner.fx:21-97 (the #if FNA nudge is off in the reference build, as in PS_Spawn) */
static void spawn_pattern_slot(f4* pos, f4* vel, f4* attr, float x, float y, const f4* rnd, int rw, int rh,
                               const IlmSpawnParams* p, const IlmPatternParams* pt, const IlmFloat4* tex, int tw, int th, int levels) {
    const float* csi = p->ChunkSizeAndIndices;
    float index = x + (y * csi[0]);
    if ((index < csi[1]) || (index > csi[2]))
        return;
    float relative_index = floorf(index - csi[1]);
    float particles_per_row = pt->StepWidthAndSizeScale[1];
    float ix = floorf(fmodf(relative_index, particles_per_row)), iy = floorf(relative_index / particles_per_row);
    iy += pt->YOffsetsAndCoordScale[0];
    float u = (ix * pt->StepWidthAndSizeScale[2]) + pt->TexelOffsetAndMipBias[0];
    float v = (iy * pt->StepWidthAndSizeScale[3]) + pt->TexelOffsetAndMipBias[1];
    v += pt->YOffsetsAndCoordScale[1];
    float position_x = ix * pt->YOffsetsAndCoordScale[2] + pt->CenteringOffset[0];
    float position_y = iy * pt->YOffsetsAndCoordScale[3] + pt->CenteringOffset[1];
    if ((u > 1.0f) || (v > 1.0f))
        return;
    f4 pattern_color = pattern_fetch(tex, tw, th, levels, u, v, pt->TexelOffsetAndMipBias[3]);

    f4 random1, random2, random3;
    evaluate_random_for_index(rnd, rw, rh, index, p->RandomnessOffset, p->AlignVelocityAndPosition, &random1, &random2, &random3);

    const f4 zero = v4(0, 0, 0, 0);
    const f4* C = p->Configuration;
    f4 temp_position = evaluate_formula(zero, p->InlinePositionConstants[0], C[0], C[1], random1, p->FormulaTypes[0], p->AxisMask);
    temp_position.x += position_x;
    temp_position.y += position_y;
    f4 attribute_constant = pattern_color;
    if (pt->MultiplyAttributeConstant != 0.0f)
        attribute_constant = v4mul(attribute_constant, C[5]);
    else
        attribute_constant = v4add(attribute_constant, C[5]);
    f4 new_position = mul_point(xyz(temp_position), &p->PositionMatrix);
    new_position.w = temp_position.w;
    f4 temp_velocity = evaluate_formula(temp_position, C[2], C[3], C[4], random2, p->FormulaTypes[1], p->AxisMask);
    f4 new_velocity = mul_point(xyz(temp_velocity), &p->VelocityMatrix);
    new_velocity.w = temp_velocity.w;
    f4 new_attributes = evaluate_formula(temp_position, attribute_constant, C[6], C[7], random3, p->FormulaTypes[2], p->AxisMask);
    if (new_attributes.w < p->AttributeDiscardThreshold)
        return;
    *pos = new_position; *vel = new_velocity; *attr = new_attributes;
}

/* spawn record dispatch over a band of rows (kinds: ILM_SPAWN_*) */
static void spawn_record_rows(IlmFloat4* pos, IlmFloat4* vel, IlmFloat4* attr, int32_t chunk_size, int y0, int y1,
                              const IlmFloat4* rnd, int32_t rw, int32_t rh, const IlmSpawnRecord* r, int slot, const OrcStepExtras* ex) {
    if (r->Kind == ILM_SPAWN_INLINE) {
        spawn_rows(pos, vel, attr, chunk_size, y0, y1, rnd, rw, rh, &r->Params);
        return;
    }
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < chunk_size; x++) {
            const int i = y * chunk_size + x;
            if (r->Kind == ILM_SPAWN_POSITION_BUFFER)
                spawn_position_buffer_slot(&pos[i], &vel[i], &attr[i], (float)x, (float)y, rnd, rw, rh, &r->Params,
                                           ex->spawn_positions[slot], ex->spawn_position_count[slot]);
            else if (r->Kind == ILM_SPAWN_FEEDBACK)
                spawn_feedback_slot(&pos[i], &vel[i], &attr[i], (float)x, (float)y, rnd, rw, rh, &r->Params, &r->Feedback,
                                    ex->source_pos[slot], ex->source_vel[slot], ex->source_attr[slot], chunk_size);
            else if (r->Kind == ILM_SPAWN_PATTERN)
                spawn_pattern_slot(&pos[i], &vel[i], &attr[i], (float)x, (float)y, rnd, rw, rh, &r->Params, &r->Pattern,
                                   ex->spawn_pattern[slot], ex->pattern_w[slot], ex->pattern_h[slot], ex->pattern_levels[slot]);
        }
}
