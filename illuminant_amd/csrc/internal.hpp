// internal.hpp -- launch descriptors shared between the kernels (particles.hip,
// lighting.hip) and the C-ABI implementation (api.hip).  Not part of the ABI.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/illuminant_hip.h"
#include "hlsl_math.hpp"

namespace ilm {

// Component planes of one chunk, SoA, `stride` floats apart:
//   0..3  x, y, z, life          (PositionAndLife)
//   4..7  vx, vy, vz, category   (Velocity)
//   8..11 attribute rgba         (Chunk.Color)
//  12..15 render color rgba      (Chunk.RenderColor)
//  16..19 render data            (Chunk.RenderData: size, rotation, speed, category)
constexpr int kComponents = 20;
constexpr int kSlotsPerThread = 4;         // widest variant; strides are padded for it
// 128-thread blocks: a block's slot is refilled when its slowest wave ends, and the spawning variant (74 VGPRs, 6 waves per SIMD) fills the
// CU more evenly in pairs of waves than in fours (tools/step_ab.py r02: cfg2 with the spawner 22.1 -> 20.9 us per step, others unchanged)
#ifndef ILM_STEP_THREADS
#define ILM_STEP_THREADS 128
#endif
constexpr int kStepThreads = ILM_STEP_THREADS;
constexpr int kSlotsPerBlock = 1024;       // strides are padded to this
#ifndef ILM_STEP_UNITS
#define ILM_STEP_UNITS 1
#endif
constexpr int kUnitsPerWave = ILM_STEP_UNITS;   // units (64 slots each) a wave loads up front and then processes in turn: 1, 2 or 4
constexpr int kDefaultStepMinWaves = 1;   // __launch_bounds__ min waves/SIMD of the main step variant (override: ILM_STEP_MINWAVES=6|7|8)

// Values every wave would otherwise recompute on the vector ALU from uniform inputs (gfx950 has no scalar float
// unit): filled on the host with the SAME IEEE single-precision operations, in the same order, as the per-slot
// code they replace, so the results are bit-identical (api.hip is compiled with -ffp-contract=off).
struct StepDerived {
    float dt_s;                  // getDeltaTimeSeconds: GlobalSettings.x / 1000            (ParticleCommon.fxh:54-56)
    float inv_rw, inv_rh;        // RandomnessTexel = 1 / (807, 653)                         (RandomCommon.fxh:12-15)
    int32_t cs_shift;            // log2(chunk_size) when it is a power of two, else -1
    int32_t noise_may_revive;    // some Noise op can change the life of a dead slot (see api.hip); 0 => dead slots skip the transforms
    uint32_t bezier_codes;       // the uniform decisions of the update pass's four curves (bezier.hpp): byte 0 ColorFromLife, 1 ColorFromVelocity,
                                 // 2 SizeFromLife, 3 SizeFromVelocity
    uint32_t update_bits;        // bit 0: LifeRampSettings.x != 0 (life ramp on)   bit 1: getVelocityRotation() == 0   bit 2: LifeRampSettings.x < 0
    // Noise.fx:49-52 samples the randomness table at uv = ((xy * RandomnessTexel) + offset) * RandomnessTexel: the slot coordinate is
    // scaled by the texel size TWICE, so the texel column only steps once per 807 slots along x (the row once per 653 along y) and
    // a whole wave of 64 consecutive slots almost always reads ONE texel in each of the four lookups -- positionDelta and
    // velocityDelta (Noise.fx:49-60) are then the same for all 64 slots.  The map coordinate -> texel index is monotone (a chain
    // of monotone float operations), i.e. a step function; the host finds its steps with the same IEEE operations, evaluates the
    // two deltas for every combination of runs (api.hip, fill_noise_fast) and the wave picks its pair with scalar integer code.
    // A wave with a step inside it keeps the per-slot lookups.
    struct NoiseFast {
        int32_t op;              // index of the (first) Noise op this describes; -1: none / not applicable (then nothing else is read)
        int32_t classes;         // classes per axis of the delta tables: 3 (tables below) or 5 (tables in the unused spawn records
                                 // of the descriptor: chunk sizes above the table size, 807 x 653, step more often; see kNoiseBigTable)
        int32_t yb[4];           // rows where a sample's texel row steps (INT32_MAX: no such step): y class = #(yb <= row)
        uint32_t wcode[16];      // per 64-slot column w of a chunk row: bit 6 = usable, bits 0-2 = x class of the (x, y) samples,
                                 // bits 3-5 = x class of the (x + 2, y + 1) samples
        IlmFloat4 position[3][3];   // positionDelta [y class of row][x class]          (classes == 3)
        IlmFloat4 velocity[3][3];   // velocityDelta [y class of row + 1][x class]
    } noise;
    struct Op {
        int32_t area_none;       // AreaType outside 1..5: evaluateByTypeId returns 0 => weight == Strength exactly
        float   t;               // Noise / FMA with area_none: weight * dtMs / TimeDivisor
        float   max_accel;       // Gravity: MaximumAcceleration * dtMs / 1000
        int32_t _pad;
    } op[ILM_MAX_OPS];
};

// With 5 classes per axis the two delta tables are 2 x 25 float4 = 800 bytes: more than the kernarg block has left, but a launch
// without spawn records does not use desc.Spawns (2 x 544 bytes), so they live there: positionDelta[5][5] then velocityDelta[5][5].
constexpr int kNoiseBigClasses = 5;
static_assert(sizeof(IlmSpawnRecord) * ILM_MAX_SPAWNS >= 2 * kNoiseBigClasses * kNoiseBigClasses * sizeof(IlmFloat4), "noise tables fit the spawn records");

constexpr int kMaxPartialChunks = 4;
struct StepLaunch {
    IlmStepDesc desc;
    StepDerived derived;
    float* const* chunk_bases;   // device table, one base pointer per chunk
    int64_t stride;              // floats between component planes: span + the engine's plane padding (api.hip, ilm_engine_create)
    int32_t span;                // slots rounded up to kSlotsPerBlock: the units a launch walks per chunk are span / 64
    int32_t chunk_size;
    int32_t slots;               // chunk_size^2: lanes at or past it are stride padding (the stride is rounded up to 1024) and take no part
    int32_t first_chunk, chunk_count;
    uint32_t op_mask;            // bit t set when an op of type t is present
    int32_t streaming;           // != 0: the chunks of this launch do not fit the Infinity Cache -> non-temporal plane accesses
    const float4* rnd; int32_t rw, rh;
    const uint2* rnd_lp;         // the Rgba64 copy of the randomness table (SpatialNoise; ParticleEngine.cs:508-540)
    const float4* ramp; int32_t ramp_w, ramp_h;
    // extended spawn kinds (per spawn record slot): the Spawner's PositionBuffer and the feedback source chunk
    const float4* spawn_positions[ILM_MAX_SPAWNS]; int32_t spawn_position_count[ILM_MAX_SPAWNS];
    const float* source_base[ILM_MAX_SPAWNS];
    // PatternSpawner texture: mip levels back to back (ilm_system_set_spawn_pattern)
    const float4* spawn_pattern[ILM_MAX_SPAWNS]; int32_t pattern_w[ILM_MAX_SPAWNS], pattern_h[ILM_MAX_SPAWNS], pattern_levels[ILM_MAX_SPAWNS];
    SdfView sdf;
    // chunks whose tail has never been written (api.hip, System::used): units >= partial_units[i] of chunk partial_chunk[i]
    // hold only zeros and are skipped; chunks not listed are processed whole
    int32_t partial_count; int32_t partial_chunk[kMaxPartialChunks]; int32_t partial_units[kMaxPartialChunks];
    // ILM_STEP_COUNT_LIVE: one 64-bit counter per chunk at index chunk * kCountStride (its own 128-byte line), all zero on entry:
    // low word = live particles, high word = blocks of the chunk that have reported.  The block that completes a chunk publishes
    // {count_seq, count} with ONE system-scope store into page-locked host memory, where ilm_system_poll_counts reads it: no copy,
    // no event, no second stream (a record / wait pair on the stepping stream cost two ~7 us bubbles per counting step).
    // A chunk owns kCountLines lines: line 0 collects the chunk, lines 1.. are buckets that the chunk's blocks are dealt over
    // (block b of the chunk -> bucket b mod count_buckets), so that the ~4000 blocks of a 1024^2 chunk, which are resident at the
    // same time, do not queue on one address; the block that completes a bucket carries its sum to line 0.
    unsigned long long* live_counts;
    unsigned long long* zero_counts;     // the other counter region: zeroed by this launch for the next counting step
    unsigned long long* host_counts;     // device address of the page-locked table, one word per chunk: count_seq << 32 | count
    uint32_t count_seq;
    int32_t zero_n;                      // lines of the other region (capacity in chunks x kCountLines)
    int32_t count_buckets;               // power of two <= kCountLines - 1 that divides the blocks per chunk
    // Work is cut into units of one wave (64 consecutive slots); unit = chunk_rel * units_per_chunk + segment.
    // Filled by launch_step.
    int32_t units_per_chunk;     // stride / 64
    int32_t upc_shift;           // log2(units_per_chunk) when it is a power of two, else -1
    int32_t unit_begin, unit_end;        // global unit range [begin, end) of this launch
    int32_t unit_rotate, total_padded;   // block b starts at unit (b * 4 + unit_rotate) mod total_padded
};

static_assert(sizeof(StepLaunch) <= 4096, "StepLaunch travels in the kernarg segment (4 KB)");

// per-chunk live counters are kCountStride 64-bit words apart (one 128-byte line each)
constexpr int kCountStride = 16;     // in 64-bit words
constexpr int kCountLines = 17;      // per chunk in the step kernels' counter regions: the chunk's line + 16 bucket lines
hipError_t launch_step(StepLaunch& a, hipStream_t stream);
hipError_t step_sdf_sample_counter(int enable, unsigned long long* out);   // ilm_debug_step_sdf_samples
// the slice-0 cells of a UNORM16 field (SdfView::cells0): does this step's collision update use them, and their build
bool step_wants_slice0_cells(const IlmStepDesc& d, int format);
hipError_t launch_build_slice0_cells(const uint2* texels, int width, int height, void* cells, hipStream_t stream);
int set_step_interpreter(int on);     // ilm_debug_step_interpreter: returns the previous setting
int set_step_streams(int n);          // ilm_debug_step_streams: 1 keeps every step on the context stream, 2 (default) lets large steps use two; returns the previous setting
// the context's stream for work that is not a particle step: ordered after everything the context's second stepping stream holds (api.hip)
hipStream_t ctx_stream_joined(IlmHandle ctx);
// Tracing ranges (the reference brackets every transform and light batch with RenderTrace.Marker,
// Illuminant/Particles/ParticleSystem.cs:464-469, Illuminant/Lighting/LightingRenderer.cs:1123-1124).  ILM_TRACE=1: every data-path entry
// point of the C ABI pushes a named roctx range for its duration (host side: what a call queues; the kernels of a range carry its
// correlation in a rocprofv3 --marker-trace --kernel-trace run).  The roctx library is bound at run time like RCCL (dlopen of
// librocprofiler-sdk-roctx.so.1, then libroctx64.so.4); off (the default) an entry point pays one predictable branch.
struct TraceApi { int (*push)(const char*) = nullptr; int (*pop)() = nullptr; bool on = false; };
const TraceApi& trace_api();
struct TraceRange {
    bool on;
    explicit TraceRange(const char* name) : on(trace_api().on) { if (on) (void)trace_api().push(name); }
    ~TraceRange() { if (on) (void)trace_api().pop(); }
    TraceRange(const TraceRange&) = delete;
    TraceRange& operator=(const TraceRange&) = delete;
};
#define ILM_TRACE_RANGE(name) ::ilm::TraceRange ilm_trace_range_(name)
// store-mode exchange of a group lightmap: the light passes into `lightmap` also store at the same offsets of `count` other buffers (0: off)
int32_t lightmap_set_mirrors(IlmHandle lightmap, void* const* buffers, int count);

// AoS float4 (device staging) <-> one SoA plane group (4 consecutive components)
hipError_t launch_aos_to_soa(const float4* src, float* plane0, int64_t stride, int32_t first_slot, int32_t count, hipStream_t stream);
hipError_t launch_soa_to_aos(const float* plane0, int64_t stride, float4* dst, int32_t first_slot, int32_t count, hipStream_t stream);

// standalone liveness count over the life plane of each chunk (CountLiveParticles.fx)
hipError_t launch_count_live(float* const* chunk_bases, int64_t stride, int32_t span, int32_t slots, int32_t chunk_count, uint32_t* counts, hipStream_t stream);
// ordered live-slot compaction of one chunk (ballot + prefix sum); *out_count is a device counter
hipError_t launch_live_slots(const float* life, int32_t slots, uint32_t* out_slots, uint32_t capacity, uint32_t* out_count, uint32_t* block_counts, hipStream_t stream);

struct GBufferView {
    const void* texels;
    int32_t width, height, format;
};

// RampTexture of the light group being rendered (technique SphereLightWithDistanceRamp, RampCommon.fxh); texels == nullptr => none
struct RampView { const float4* texels; int32_t width, height; };

struct LightLaunch {
    const IlmLightVertex* lights;   // device
    int32_t light_count;
    IlmEnvironment env;
    IlmDistanceFieldUniforms df;
    GBufferView gbuffer;            // texels == nullptr => none
    TraceSdfView sdf;               // texels == nullptr => none
    float ambient[4];
    void* lightmap; int32_t width, height, format;
    int32_t row_begin, row_end;
    unsigned long long* stats;      // device, 3 counters, or nullptr
    const int32_t* light_count_ptr; // device: when non-null the record count is read from here (particle lights are counted on the device)
    int32_t tile_map;               // block -> tile mapping: 0 contiguous band per XCD, 1 tile rows round-robin over the XCDs, 2 identity,
                                    // 4 groups of tile_macro x tile_macro tiles round-robin over the XCDs
    int32_t accumulate;             // != 0: start from the lightmap's contents instead of `ambient` (additive blend onto an earlier pass)
    int32_t blend_fp16;             // != 0: the reference's HalfVector4 render target -- round through fp16 after every light (ilm_ctx_set_lightmap_blend)
    RampView ramp;
    int32_t tile_macro;             // tile_map 4: edge of the square groups of tiles dealt round-robin to the XCDs
    // Light split (lighting.hip, "parts"): `split` workgroups serve one tile, each walking kLightParts / split consecutive parts of the
    // tile's light list; their per-part sums meet in `partials` (float4 per pixel, tile-major, part, thread) and the workgroup that
    // draws the tile's last ticket adds them up in part order.  split == 1: one workgroup per tile, nothing leaves the registers / LDS.
    const uint16_t* group_order;    // tile_map 4: the groups in the order they are dealt out (heaviest first), or nullptr: row-major
    int32_t split;                  // the largest number of workgroups per tile in this launch (1: no split anywhere)
    // Tapered split: of the `taper_slots` tiles an XCD is dealt (block slots, padding included), the first taper[0] are served by one
    // workgroup each, those up to taper[1] by two, up to taper[2] by four, the rest by eight -- the work that starts last comes in the
    // smallest pieces (guided self-scheduling, laid out in the grid).  No taper: taper[0..2] = taper_slots.
    int32_t taper[3], taper_slots;
    float4* partials;               // device scratch of the context: tile_count * kLightParts * 256 float4 (split > 1)
    uint32_t* tickets;              // device, one per tile, zero between launches (the last arriver resets its tile's)
    // Store-mode exchange of a group lightmap (ILM_GATHER_STORE, group.hip): the lightmap's texel is ALSO stored at the same offset of
    // `mirror_count` other buffers -- the other members' copies of the frame, peer-mapped over xGMI -- so the strips need no gather phase.
    void* const* mirrors;           // device array of mirror_count base pointers (nullptr / 0: an ordinary lightmap)
    int32_t mirror_count;
};
// A tile's light list is summed in kLightParts consecutive parts (entries [p n / 8, (p + 1) n / 8) of a list of n), each part from zero in
// light order, the parts added onto the clear colour in part order: the sum's bits depend on the tile and its list only, not on how
// many workgroups computed the parts.
constexpr int kLightParts = 8;
// Tile edge of the light pass in pixels: 16 = four waves per workgroup (one 8 x 8 quadrant each), 8 = one wave per workgroup (EXPERIMENT, -DILM_LIGHT_TILE=8)
#ifndef ILM_LIGHT_TILE
#define ILM_LIGHT_TILE 16
#endif
constexpr int kLightTile = ILM_LIGHT_TILE;
constexpr int kLightTileThreads = (kLightTile / 8) * (kLightTile / 8) * 64;

constexpr size_t kLightRecBytes = 128;   // sizeof(LightRec) in lighting.hip
// What the in-volume trace loop asks of a light and the field, decided once per light by the prepare kernels (lighting.hip,
// light_flags): the centre inside the table sampler's box (SdfView) with the margin for rounding, and a field whose encoded distance,
// extent and step budget are in range.
struct TraceGate {
    float x0, x1, y0, y1, z0, z1;       // box of SdfView shrunk by the light-side margin; empty when the field has no table
    float field_ok;
};
TraceGate make_trace_gate(const IlmDistanceFieldUniforms& df, const SdfView& sdf);
// per-call preparation of the light records (footprint, cone config, trace flags) into `recs` (device, count * kLightRecBytes)
hipError_t launch_prepare_lights(const IlmLightVertex* lights, int count, const IlmEnvironment& env, const IlmDistanceFieldUniforms& df,
                                 const SdfView& sdf, void* recs, hipStream_t stream);
hipError_t launch_sphere_lights_prepared(const LightLaunch& a, const void* recs, hipStream_t stream);
int light_launch_blocks(const LightLaunch& a);    // workgroups of the tile kernel's launch for a (split / taper included)
int light_block_slots(const LightLaunch& a);      // block slots per XCD of the tile kernel's launch over a's rows

// Particle lights (ParticleLight.fx): ordered device-side compaction of the live, visible particles of every chunk into light
// records.  block_counts: one int per 1024-slot block of every chunk (scratch); *out_count receives the record count.
struct ParticleLightLaunch {
    float* const* chunk_bases; int64_t stride; int32_t chunk_count, slots;
    const int32_t* quad_counts;     // device, per chunk; nullptr => every slot
    IlmParticleLightParams params;
    IlmEnvironment env;
    float max_cone_radius;
    TraceGate gate;                 // make_trace_gate of the frame's field
    int32_t* block_counts;          // chunk_count * blocks_per_chunk
    void* recs; int32_t capacity;   // LightRec array
    int32_t* out_count;
};
hipError_t launch_prepare_particle_lights(const ParticleLightLaunch& a, hipStream_t stream);
// Light probes (SphereLightProbe.fx): every prepared light record on every probe; values = float4 per probe (device)
hipError_t launch_light_probes(const void* recs, int light_count, const float4* probe_positions, const float4* probe_normals, int probe_count,
                               const IlmEnvironment& env, const IlmDistanceFieldUniforms& df, const SdfView& sdf, const RampView& ramp, float4* values, float4* pairs,
                               hipStream_t stream);
// sampleDistanceFieldEx at `count` positions (xyz triples) -- diagnostic entry point ilm_sdf_sample
hipError_t launch_stamp16(void* where, const uint32_t stamp[4], hipStream_t stream);      // 16 bytes stored by a kernel (output.hip)
hipError_t launch_sdf_sample(const SdfView& sdf, const IlmDistanceFieldUniforms& df, const float* positions, int count, float* out, hipStream_t stream);
// fills the cell array of the table sampler (sdf.table_slices x sdf.slice_h x sdf.slice_w cells of 16 bytes) from the atlas
hipError_t launch_build_sdf_cells(const TraceSdfView& sdf, void* cells, int first_slice, int slice_count, hipStream_t stream);
hipError_t launch_sdf_sample_inside(const TraceSdfView& sdf, const IlmDistanceFieldUniforms& df, const float* positions, int count, float* out, int32_t* used,
                                    hipStream_t stream);
hipError_t launch_divide_by_constant(float divisor, float reciprocal, unsigned long long* mismatches, hipStream_t stream);
void light_constant_divisors(float out[2][2]);
hipError_t launch_divide_probe(const float* n, const float* d, int count, float* out_fast, float* out_ieee, hipStream_t stream);

// ---- distance-field generation (fields.hip) ---------------------------------------------------------------
// One obstruction as the kernel reads it: the DistanceFunctionVertex (Vertices.cs:105-141) plus its quad in slice
// pixels, computed on the host with the oracle's operations (api.hip is compiled with -ffp-contract=off).
struct FieldObstruction {
    float cx, cy, cz; int32_t type;
    float sx, sy, sz; int32_t _pad;       // flags (api.hip): bit 0 the orientation is the identity quaternion (the rotation is skipped), bit 1 sizes and centre of ordinary magnitude
    float qx, qy, qz, qw;
    float x0, x1, y0, y1;       // raster bounds of DistanceFunctionVertexShader's quad, slice-local pixels
    // Exact culling (fields.hip): with e = |world position - centre|, the distance function is provably >= (e - cull_radius) / cull_inv_scale
    // (api.hip, fill_cull_bound: bounding sphere of the shape, safety margins for the float evaluation folded in).  An obstruction whose
    // bound cannot beat a texel's current maximum is not evaluated; cull_radius = +infinity switches the test off.
    float cull_inv_scale, cull_radius;
    float _pad2[2];
};
static_assert(sizeof(FieldObstruction) == 80, "FieldObstruction is one 80-byte record");
struct FieldVolume {
    int32_t first_vertex, vertex_count;
    float z0, z1;               // zRange = (ZBase, ZBase + Height)
    float x0, x1, y0, y1;       // hv.Bounds.Expand(DistanceLimit) in slice-local pixels
    float cx, cy, radius;       // a circle (world units) that holds the polygon, radius rounded up: the culling bound of fields.hip
    float _pad;
};
struct FieldLaunch {
    uint2* atlas; const uint2* clear_source; int32_t atlas_w;
    const int32_t* first_slices; int32_t triplet_count;     // device
    const FieldObstruction* obstructions; int32_t obstruction_count;
    const FieldVolume* volumes; int32_t volume_count;
    const float2* polygon_xy;
    int32_t slice_w, slice_h, columns, virtual_w, virtual_h;
    float slice_count_f, virtual_depth, z_offset, max_encoded, inv_scale_x, inv_scale_y;
};
hipError_t launch_render_slices(const FieldLaunch& a, int format, hipStream_t stream);

// G-buffer generation (fields.hip): volumes sorted by top height on the host, bounds of each polygon for the early reject
struct GBufferVolume {
    int32_t first_vertex, vertex_count;
    float top; int32_t enable_shadows;
    float x0, x1, y0, y1;
};
struct GBufferLaunch {
    void* texels; int32_t width, height, format;
    IlmGBufferRenderDesc desc;
    const GBufferVolume* volumes; int32_t volume_count;
    const float2* polygon_xy;
};
hipError_t launch_render_gbuffer(const GBufferLaunch& a, hipStream_t stream);

// G-buffer from the host's meshes (gbuffer.hip): one record per triangle in draw order, written by the setup kernel and read
// wave-uniformly by the raster kernel
constexpr int kGBufferAttrs = 11;
// what the setup kernel works on (one triangle in draw order) ...
struct GBufferPrimSetup {
    int32_t x[3], y[3];           // 1/256-pixel positions, clockwise on the y-down screen
    int32_t i0, i1, j0, j1;       // pixels whose centres the bounding box holds (inclusive; empty for a degenerate triangle)
    int32_t kind, texture;        // pixel shader; index into the launch's texture table or -1
    float a[3][kGBufferAttrs];    // per-vertex attributes
    int32_t flat;                 // bit k: attribute k is the same finite number at the three vertices (interpolation returns a[0][k] + 0)
    float enc_x, enc_y;           // the encoded normal of a ground / top / front-face triangle whose normal is flat
};
// ... and what it leaves for the raster kernel's scalar loads: per attribute (a0, a1 - a0, a2 - a0) side by side -- one 16-byte load and
// no subtraction per attribute and covering triangle (the differences are the same IEEE operations, taken once)
struct GBufferPrim {
    float4 attr[kGBufferAttrs];
    int32_t flat; float enc_x, enc_y; int32_t _pad;
};
struct GBufferTex { const void* texels; int32_t width, height, format, _pad; };
struct GBufferMeshLaunch {
    void* texels; int32_t width, height, format;
    IlmGBufferMeshDesc desc;
    const IlmHeightVolumeVertex* top; int32_t top_triangles;
    const IlmHeightVolumeVertex* front; int32_t front_triangles;
    const IlmBillboardVertex* billboards;
    const int4* quads;            // (quad, texture, kind, -) per billboard quad in draw order
    const GBufferTex* textures;   // device copy (read by the raster kernel)
    const GBufferTex* textures_in; int32_t texture_count;   // where the setup kernel copies it from (the pinned slot), or nullptr
    GBufferPrim* prims; int32_t prim_count;
    int4* bounds;                 // (i0, i1, j0, j1) per record: what the binning passes read
    int4* verts;                  // (x0, y0, x1, y1), (x2, y2, kind, texture) per record: what a wave's coverage test reads
    int32_t block_shift, block_cols, block_rows;   // coarse bins: blocks of (1 << block_shift)^2 pixels, a multiple of the 16 x 16 tile
    int32_t* block_count;         // [block_cols * block_rows]
    int32_t* block_list;          // [block_cols * block_rows][prim_count] record indices in draw order
};
constexpr size_t kGBufferBlockListBudget = (size_t)256 << 20;   // bytes of block lists per frame before the blocks grow
hipError_t launch_gbuffer_meshes(const GBufferMeshLaunch& a, hipStream_t stream);

// ---- output side (output.hip) ---------------------------------------------------------------------------------
struct ReadbackLaunch {
    float* const* chunk_bases; int64_t stride; int32_t chunk_count, slots;
    const int32_t* element_counts;      // device, per chunk; nullptr => every slot
    IlmReadbackParams params;
    // derived on the host exactly as FillReadbackResult does before its loop (ParticleReadback.cs:100-112)
    float region_w, region_h; int32_t frame_count_x, frame_count_y; double max_angle_x, max_angle_y;
    int32_t* block_counts;
    IlmReadbackDrawCall* out; int32_t capacity;
    int32_t* out_count;
};
hipError_t launch_readback(const ReadbackLaunch& a, hipStream_t stream);

struct ResolveLaunch {
    const void* src; int32_t src_format;
    void* dst; int32_t dst_format;
    const void* albedo; int32_t albedo_format;      // the ...WithAlbedo techniques: the albedo texture, texel for texel (NULL: plain resolve)
    int32_t width, row_begin, row_end, mode;
    // uniform reciprocals are taken on the host (one rounding each; the per-pixel divisions they replace cost ~10 VALU instructions)
    float inverse_scale, offset, exposure_minus_one, gamma_minus_one, middle_gray, inv_average_luminance, inv_maximum_luminance_squared, inv_white;
};
hipError_t launch_resolve(const ResolveLaunch& a, hipStream_t stream);

// ---- particle rasterisation (raster.hip) ------------------------------------------------------------------------
struct Sprite;
struct RasterLaunch {
    float* const* chunk_bases; int64_t stride; int32_t chunk_count, slots, total_slots;
    const int32_t* quad_counts;         // device, per chunk; nullptr => every slot
    IlmRasterizeParams params;
    void* target; int32_t format, width, height, tiles_x, tiles_y;
    int32_t count_shaded;               // != 0: count the fragments that were blended (statistics)
    const float4* bitmap; int32_t bitmap_w, bitmap_h;      // Appearance.Texture, one level (ilm_system_set_bitmap)
    // scratch, filled by render_particles
    Sprite* sprites; uint32_t* counts; uint32_t* offsets; uint2* rects;
    unsigned long long* keys; unsigned long long* sorted_keys; int64_t pair_count;
    unsigned long long* stats;          // [0] live quads, [2] shaded pixels
    // per tile (tile_count + 1 entries where scanned): first key of the tile's run, number of segments the run is cut into, first
    // work item of the tile, first partial-result slot of the tile (tiles with more than one segment only)
    uint32_t* tile_begin; uint32_t* tile_segments; uint32_t* tile_first_item; uint32_t* tile_multi; uint32_t* tile_first_partial;
    float* partials;                    // per partial slot: 256 x (premultiplied colour sum rgba, transmittance)
    int32_t work_items;                 // upper bound of the work items (grid of the shading kernel)
};
// device buffers the rasteriser keeps between calls (owned by the context)
struct RasterScratch {
    void* sprites = nullptr; void* counts = nullptr; void* offsets = nullptr; void* keys = nullptr; void* sorted_keys = nullptr;
    void* temp = nullptr; void* stats = nullptr; void* tiles = nullptr; void* partials = nullptr; void* rects = nullptr;
    size_t sprites_cap = 0, counts_cap = 0, offsets_cap = 0, keys_cap = 0, sorted_cap = 0, temp_cap = 0, tiles_cap = 0, partials_cap = 0, rects_cap = 0;
    unsigned long long* host_words = nullptr;    // pinned, device-visible: [0] the frame's pair count, [1..4] its statistics (written by kernels, read after a stream sync)
};
hipError_t render_particles(RasterLaunch& a, RasterScratch& s, hipStream_t stream, unsigned long long out_stats[3], bool* too_many);
void free_raster_scratch(RasterScratch& s);
hipError_t launch_clear_target(void* texels, int format, size_t n, float4 color, hipStream_t stream);

// ---- shared with group.hip (the multi-device layer sits on the C ABI of api.hip, inside the same library) -------------------------
// thread-local error text + return code, as every entry point reports failures
int32_t api_fail(int32_t code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// the table of live handles (api.hip): objects of other translation units register under their own magic
constexpr uint32_t kMagicGroup = 0x494C4752u, kMagicGroupLightmap = 0x494C474Cu;
IlmHandle handle_register(const void* object, uint32_t magic);
bool handle_is_live(IlmHandle h, uint32_t magic);
void handle_retire(const void* object);
int ctx_child_count(IlmHandle ctx);        // live objects of a context (-1: not a context)
// one chunk of a system for the chunk exchange (api.hip): base of component 0, stride between component planes (floats), chunk size, owning context
int32_t system_chunk_view(IlmHandle system, int chunk, bool written, float** out_base, int64_t* out_stride, int32_t* out_chunk_size, IlmHandle* out_ctx);

}  // namespace ilm
