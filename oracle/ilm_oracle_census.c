/* ilm_oracle_census.c -- TEST / ANALYSIS INFRASTRUCTURE (included by ilm_oracle.c; never part of the product).
 *
 * How much of a lit frame's cone-trace work is spent on rays that provably see nothing?  (VERDICT r04 #2.)
 *
 * coneTrace (Illuminant/Shaders/ConeTrace.fxh:148-191) returns
 *     pow(saturate(saturate(min(visibility, stepsRemaining / MAX_STEP_RAMP_WINDOW) - FULLY_SHADOWED) / (UNSHADOWED - FULLY_SHADOWED)), power)
 * with visibility = min over the samples of (distance + HACK_DISTANCE_OFFSET) / min(growth * x + MIN_CONE_RADIUS, maxRadius)
 * (coneTraceStep, :52-74), starting from 1.  That is EXACTLY 1.0f whenever
 *   (V) every sample's quotient is >= c, c = 1 ("strict": the running minimum never leaves 1.0f) or c = 0.9501 ("loose":
 *       (v - 0.075f) / 0.875f >= 1 for every float v >= 0.9501f, so both saturates give 1 and pow(1, p) = 1), and
 *   (S) the march ends because it reached the light (x >= length) with at least MAX_STEP_RAMP_WINDOW = 2 steps of its budget left.
 * Both can be PROVEN without marching from a table of lower bounds of the field: D[b] = the smallest decoded distance of any texel
 * of brick b (B x B texels of the slice plane, plus one texel of apron on every side for the bilinear taps) over ALL slices (so the
 * z-lerp of any two slices is covered).  A sample whose xy lies in brick b returns d >= D[b] (every tap is in the brick + apron,
 * bilinear and slice lerps are convex; a margin covers the lerps' roundings; outside the volume the sampler ADDS a distance).
 * Along the segment start -> light, per brick crossed with x in [x_in, x_out]:
 *   (V)  D[b] + 1.5 >= c * min(growth * x_out + 0.33, maxRadius)                      (the radius grows with x: x_out is the worst case)
 *   (S)  samples in the brick <= ceil((x_out - x_in) / max(D[b] * longStep, minStep)) + 1  (every step is at least that long)
 *        and the sum over the bricks <= StepLimit - 2.
 * Bricks on the border of the slice are never open (their taps wrap into the neighbouring slice of the atlas, U WRAP), neither is a
 * ray that leaves the volume in xy.
 *
 * The census marches every traced pair as the oracle does (so the counts are the frame's own), evaluates the proof beside it, CHECKS
 * that every proven ray indeed returned exactly 1.0f, and accumulates what a kernel could skip: per pair, and per wave -- the light
 * kernel runs an 8 x 8 pixel square as one wave64 (lighting.hip:582-583), whose loop runs as long as its longest lane. */

typedef struct OrcOpenRayCensus {
    uint64_t traced_pairs, traced_samples;              /* every traced pair, the SDF samples of its march */
    uint64_t result_one_pairs, result_one_samples;      /* marches that returned exactly 1.0f: the ceiling of ANY exact early-out */
    uint64_t strict_pairs, strict_samples;              /* provable with c = 1 */
    uint64_t loose_pairs, loose_samples;                /* provable with c = 0.9501 */
    uint64_t violations;                                /* proven rays whose march did not return exactly 1.0f: must be 0 */
    uint64_t wave_count, wave_open;                     /* (8 x 8 square, light) waves with a traced lane; those ALL of whose traced lanes are proven (loose) */
    uint64_t wave_iterations, wave_iterations_left;     /* sum over waves of the longest lane's samples; the same with the proven lanes removed */
    uint64_t wave_open_samples;                         /* samples of the lanes of fully open waves */
    uint64_t dda_bricks;                                /* bricks visited by the proofs (what the test itself would cost) */
} OrcOpenRayCensus;

typedef struct {
    int bricks_x, bricks_y, brick_texels;
    float texels_per_unit_x, texels_per_unit_y;         /* slice texels per unit of the field's virtual xy */
    float extent_x, extent_y;
    float* min_distance;                                /* bricks_x * bricks_y; -INFINITY = never open */
} BrickTable;

static BrickTable build_brick_table(const IlmDistanceFieldUniforms* df, const OrcTexture* sdf, int brick_texels) {
    BrickTable t;
    const int cols = (int)df->TextureSliceCount.x, rows = (int)df->TextureSliceCount.y;
    const int slice_w = sdf->width / cols, slice_h = sdf->height / rows;
    const int slices = (int)df->TextureSliceCount.w;                       /* virtual slices */
    const int physical = (slices + 2) / 3;
    t.brick_texels = brick_texels;
    t.bricks_x = (slice_w + brick_texels - 1) / brick_texels;
    t.bricks_y = (slice_h + brick_texels - 1) / brick_texels;
    t.texels_per_unit_x = (float)slice_w / df->Extent.x;
    t.texels_per_unit_y = (float)slice_h / df->Extent.y;
    t.extent_x = df->Extent.x; t.extent_y = df->Extent.y;
    /* per slice-plane texel: the smallest distance over all physical slices and channels */
    float* column_min = (float*)malloc(sizeof(float) * (size_t)slice_w * (size_t)slice_h);
    #pragma omp parallel for schedule(static)
    for (int y = 0; y < slice_h; y++)
        for (int x = 0; x < slice_w; x++) {
            float lowest = INFINITY;
            for (int p = 0; p < physical; p++) {
                const int ax = (p % cols) * slice_w + x, ay = (p / cols) * slice_h + y;
                float c[4];
                sdf_texel(sdf, ax, ay, c);
                for (int k = 0; k < 4; k++) {
                    const float d = (DISTANCE_ZERO - c[k]) * df->Extent.w;
                    if (!(d >= lowest)) lowest = d;               /* a NaN texel makes the column NaN -> the brick closed */
                }
            }
            column_min[(size_t)y * (size_t)slice_w + (size_t)x] = lowest;
        }
    t.min_distance = (float*)malloc(sizeof(float) * (size_t)t.bricks_x * (size_t)t.bricks_y);
    for (int by = 0; by < t.bricks_y; by++)
        for (int bx = 0; bx < t.bricks_x; bx++) {
            float lowest = INFINITY;
            const int x0 = bx * brick_texels - 1, x1 = (bx + 1) * brick_texels, y0 = by * brick_texels - 1, y1 = (by + 1) * brick_texels;
            if (x0 < 0 || y0 < 0 || x1 > slice_w - 1 || y1 > slice_h - 1) {
                lowest = -INFINITY;                               /* border brick: taps wrap / clamp outside the slice */
            } else {
                for (int y = y0; y <= y1; y++)
                    for (int x = x0; x <= x1; x++) {
                        const float d = column_min[(size_t)y * (size_t)slice_w + (size_t)x];
                        if (!(d >= lowest)) lowest = d;
                    }
                if (!(lowest == lowest)) lowest = -INFINITY;
            }
            t.min_distance[(size_t)by * (size_t)t.bricks_x + (size_t)bx] = lowest;
        }
    free(column_min);
    return t;
}

/* 1 = the ray start -> light is provably open under criterion c (see the file comment); *bricks_visited counts the table lookups */
static int open_ray_proof(const BrickTable* t, const IlmDistanceFieldUniforms* df, f3 start, f3 light_center, float light_radius,
                          float light_ramp, double c, uint64_t* bricks_visited) {
    const double margin = 1e-3;
    const double tvx = (double)light_center.x - start.x, tvy = (double)light_center.y - start.y, tvz = (double)light_center.z - start.z;
    const double length = sqrt(tvx * tvx + tvy * tvy + tvz * tvz);
    if (!(length > 1e-6)) return 0;
    const double dx = tvx / length, dy = tvy / length;
    const double x_end = fmax(length - (double)light_radius, 1.0) + margin;       /* the last sample lies below data_y */
    const double x_begin = CT_TRACE_INITIAL_OFFSET_PX - margin;
    /* both ends of the sampled segment inside the volume in xy (a segment: everything between them too) */
    const double ax = start.x + dx * x_begin, ay = start.y + dy * x_begin, bx = start.x + dx * x_end, by = start.y + dy * x_end;
    if (!(ax > 0 && ax < t->extent_x && ay > 0 && ay < t->extent_y && bx > 0 && bx < t->extent_x && by > 0 && by < t->extent_y)) return 0;
    const double max_radius = h_clamp(light_radius, CT_MIN_CONE_RADIUS, df->ConeAndMisc.x);
    const double growth = max_radius / fmax((double)light_ramp, 16.0);
    const double min_step = fmax(1.0, (double)df->Packed1.w), long_step = (double)df->StepAndMisc2.z;
    const double budget = (double)df->StepAndMisc2.x - 2.0;
    /* 2D DDA over the bricks, in brick units */
    const double sx = t->texels_per_unit_x / t->brick_texels, sy = t->texels_per_unit_y / t->brick_texels;
    double x = x_begin, steps = 0.0;
    int guard = 0;
    while (x < x_end) {
        const double px_ = (start.x + dx * x) * sx, py_ = (start.y + dy * x) * sy;
        /* the brick of a point a hair inside the interval (so that a point ON a boundary belongs to the brick being entered) */
        const double probe = fmin(x + 1e-6 * (1.0 + fabs(x)), x_end);
        const int cx = (int)floor((start.x + dx * probe) * sx), cy = (int)floor((start.y + dy * probe) * sy);
        if (cx < 0 || cy < 0 || cx >= t->bricks_x || cy >= t->bricks_y) return 0;
        /* where the ray leaves this brick */
        double tx = INFINITY, ty = INFINITY;
        if (dx > 0) tx = ((cx + 1) - px_) / (dx * sx); else if (dx < 0) tx = (cx - px_) / (dx * sx);
        if (dy > 0) ty = ((cy + 1) - py_) / (dy * sy); else if (dy < 0) ty = (cy - py_) / (dy * sy);
        double x_out = x + fmax(fmin(tx, ty), 0.0);
        if (!(x_out > x)) x_out = x + 1e-6 * (1.0 + fabs(x));        /* (a corner: make progress) */
        if (x_out > x_end) x_out = x_end;
        if (bricks_visited) (*bricks_visited)++;
        const double D = (double)t->min_distance[(size_t)cy * (size_t)t->bricks_x + (size_t)cx] - margin;
        if (!(D > 0.0)) return 0;
        const double radius_here = fmin(growth * x_out + CT_MIN_CONE_RADIUS, max_radius);
        if (!(D + CT_HACK_DISTANCE_OFFSET >= c * radius_here + margin)) return 0;
        steps += ceil((x_out - x) / fmax(D * long_step, min_step)) + 1.0;
        if (steps > budget) return 0;
        x = x_out;
        if (++guard > 100000) return 0;
    }
    return 1;
}

typedef struct {
    const BrickTable* table;
    const IlmDistanceFieldUniforms* df;
    int squares_x, light_count, row0;
    /* per (square of this 8-row band, light): longest lane, longest unproven lane, traced lanes, proven lanes, samples */
    uint16_t* longest; uint16_t* longest_left; uint16_t* traced; uint16_t* proven; uint32_t* samples;
    OrcOpenRayCensus c;
} CensusBand;

static void census_hook(void* user, const OrcTraceEvent* e) {
    CensusBand* b = (CensusBand*)user;
    OrcOpenRayCensus* c = &b->c;
    c->traced_pairs++; c->traced_samples += e->samples;
    if (e->result == 1.0f) { c->result_one_pairs++; c->result_one_samples += e->samples; }
    const int strict = open_ray_proof(b->table, b->df, e->start, e->light_center, e->light_radius, e->light_ramp, 1.0, &c->dda_bricks);
    const int loose = strict || open_ray_proof(b->table, b->df, e->start, e->light_center, e->light_radius, e->light_ramp, 0.9501, NULL);
    if (strict) { c->strict_pairs++; c->strict_samples += e->samples; }
    if (loose) { c->loose_pairs++; c->loose_samples += e->samples; }
    if (loose && e->result != 1.0f) c->violations++;
    const size_t k = (size_t)(e->px >> 3) * (size_t)b->light_count + (size_t)e->light;
    const uint16_t n = (uint16_t)(e->samples > 65535 ? 65535 : e->samples);
    if (n > b->longest[k]) b->longest[k] = n;
    if (!loose && n > b->longest_left[k]) b->longest_left[k] = n;
    b->traced[k]++; b->samples[k] += (uint32_t)e->samples;
    if (loose) b->proven[k]++;
}

/* rows [row_begin, row_end) of the frame (row_begin a multiple of 8); the counters are ADDED to *out */
void orc_open_ray_census(const IlmLightVertex* lights, int32_t light_count, const IlmEnvironment* env, const IlmDistanceFieldUniforms* df,
                         const OrcTexture* gbuffer, const OrcTexture* sdf, int32_t width, int32_t height, int32_t row_begin, int32_t row_end,
                         int32_t brick_texels, OrcOpenRayCensus* out) {
    if (row_begin < 0) row_begin = 0;
    if (row_end > height) row_end = height;
    const BrickTable table = build_brick_table(df, sdf, brick_texels);
    const int squares_x = (width + 7) / 8, first = row_begin / 8, last = (row_end + 7) / 8;
    const float ambient[4] = { 0, 0, 0, 1 };
    OrcOpenRayCensus total;
    memset(&total, 0, sizeof(total));
    #pragma omp parallel
    {
        CensusBand b;
        memset(&b, 0, sizeof(b));
        b.table = &table; b.df = df; b.squares_x = squares_x; b.light_count = light_count;
        const size_t cells = (size_t)squares_x * (size_t)light_count;
        b.longest = (uint16_t*)malloc(cells * 2); b.longest_left = (uint16_t*)malloc(cells * 2);
        b.traced = (uint16_t*)malloc(cells * 2); b.proven = (uint16_t*)malloc(cells * 2); b.samples = (uint32_t*)malloc(cells * 4);
        #pragma omp for schedule(dynamic, 1)
        for (int sq = first; sq < last; sq++) {
            memset(b.longest, 0, cells * 2); memset(b.longest_left, 0, cells * 2);
            memset(b.traced, 0, cells * 2); memset(b.proven, 0, cells * 2); memset(b.samples, 0, cells * 4);
            for (int py = sq * 8 > row_begin ? sq * 8 : row_begin; py < (sq + 1) * 8 && py < row_end; py++) {
                SdfCounter ctr = { 0 };
                uint64_t pairs = 0, traced = 0;
                sphere_lights_row(py, lights, light_count, env, df, gbuffer, sdf, ambient, NULL, width, &ctr, &pairs, &traced, census_hook, &b);
            }
            for (size_t k = 0; k < cells; k++)
                if (b.traced[k]) {
                    b.c.wave_count++;
                    b.c.wave_iterations += b.longest[k];
                    b.c.wave_iterations_left += b.longest_left[k];
                    if (b.proven[k] == b.traced[k]) { b.c.wave_open++; b.c.wave_open_samples += b.samples[k]; }
                }
        }
        #pragma omp critical
        {
            uint64_t* dst = (uint64_t*)&total; const uint64_t* src = (const uint64_t*)&b.c;
            for (size_t i = 0; i < sizeof(total) / sizeof(uint64_t); i++) dst[i] += src[i];
        }
        free(b.longest); free(b.longest_left); free(b.traced); free(b.proven); free(b.samples);
    }
    {
        uint64_t* dst = (uint64_t*)out; const uint64_t* src = (const uint64_t*)&total;
        for (size_t i = 0; i < sizeof(total) / sizeof(uint64_t); i++) dst[i] += src[i];
    }
    free(table.min_distance);
}
