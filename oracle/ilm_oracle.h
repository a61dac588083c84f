/*
 * ilm_oracle.h -- CPU restatement of the reference's HLSL for the two hot paths.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or
 * executed by the product path (illuminant_amd/, libilluminant_hip.so,
 * libilluminant_host.so).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker / the timed
 * CPU baseline.
 *
 * PARITY UNPINNED: the reference (sq/Illuminant, C# + HLSL) ships no tests,
 * golden vectors or fixtures for these paths, and neither .NET nor fxc exists
 * in the build container, so this restatement cannot be checked against the
 * reference's own output.  It is pinned only by the substitute known-answer
 * tests listed in DESIGN.md (values hand-derived from the cited reference
 * lines; tests/test_oracle_kat.py).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference checkout).  HLSL semantics are made explicit:
 *   saturate(x) = min(max(x,0),1) with NaN -> 0;  lerp(a,b,t) = a + (b-a)*t;
 *   float % = fmodf (truncating);  normalize(v) = v / sqrt(dot(v,v));
 *   POINT sample = floor(u*W) with WRAP = positive modulo, CLAMP = clamp;
 *   LINEAR sample = texel centres at +0.5, fp32 weights;
 *   unorm16 -> float = v / 65535.  No FMA contraction (-ffp-contract=off).
 */
#ifndef ILM_ORACLE_H
#define ILM_ORACLE_H

#include "../include/illuminant_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* the numbers this restatement takes from the reference's text, by the reference's names (ilm_oracle_constants.h);
 * tests/test_reference_pin.py compares them with the values extracted from the reference sources */
int32_t orc_reference_constant(const char* key, double* out_value);
int32_t orc_reference_constant_count(void);
const char* orc_reference_constant_key(int32_t index);

typedef struct OrcTexture {
    const void* texels;
    int32_t width, height;
    int32_t format;          /* ILM_SDF_* / ILM_GBUFFER_* */
} OrcTexture;

/* particles: AoS float4 planes of one chunk (chunk_size^2 slots), updated in place */
void orc_spawn(IlmFloat4* pos, IlmFloat4* vel, IlmFloat4* attr, int32_t chunk_size,
               const IlmFloat4* rnd, int32_t rw, int32_t rh, const IlmSpawnParams* p);
void orc_gravity(IlmFloat4* pos, IlmFloat4* vel, int32_t chunk_size,
                 const IlmParticleSystemUniforms* sys, const IlmGravityParams* p);
void orc_noise(IlmFloat4* pos, IlmFloat4* vel, int32_t chunk_size,
               const IlmFloat4* rnd, int32_t rw, int32_t rh,
               const IlmParticleSystemUniforms* sys, const IlmNoiseParams* p);
void orc_fma(IlmFloat4* pos, IlmFloat4* vel, int32_t chunk_size,
             const IlmParticleSystemUniforms* sys, const IlmFMAParams* p);
void orc_update(IlmFloat4* pos, IlmFloat4* vel, const IlmFloat4* attr,
                IlmFloat4* render_color, IlmFloat4* render_data, int32_t chunk_size,
                const IlmParticleSystemUniforms* sys, const IlmUpdateParams* p,
                const IlmFloat4* life_ramp, int32_t ramp_w, int32_t ramp_h,
                const IlmDistanceFieldUniforms* df, const OrcTexture* sdf /* both NULL => UpdatePositions */);
void orc_erase(IlmFloat4* pos, IlmFloat4* vel, IlmFloat4* render_color, IlmFloat4* render_data, int32_t chunk_size);
uint32_t orc_count_live(const IlmFloat4* pos, int32_t slots, int32_t saturate16);

/* one ParticleSystem.Update over a table of chunks; planes[c][0..4] = pos, vel, attr, rc, rd */
void orc_step(IlmFloat4** planes, int32_t chunk_count, int32_t chunk_size,
              const IlmFloat4* rnd, int32_t rw, int32_t rh,
              const IlmFloat4* life_ramp, int32_t ramp_w, int32_t ramp_h,
              const OrcTexture* sdf, const IlmStepDesc* desc, uint32_t* live_counts /* may be NULL */);

/* remaining particle techniques, ilm_oracle_transforms.c */
void orc_matrix_multiply(IlmFloat4* pos, IlmFloat4* vel, int32_t chunk_size,
                         const IlmParticleSystemUniforms* sys, const IlmMatrixMultiplyParams* p);
void orc_low_precision_randomness(const IlmFloat4* rnd, int32_t count, uint16_t* out /* count * 4 */);
void orc_spatial_noise(IlmFloat4* pos, IlmFloat4* vel, int32_t chunk_size, const uint16_t* low_precision_rnd, int32_t rw, int32_t rh,
                       const IlmParticleSystemUniforms* sys, const IlmSpatialNoiseParams* p);
/* what orc_step cannot find in the descriptor: the position lists of ILM_SPAWN_POSITION_BUFFER records, the textures of
 * ILM_SPAWN_PATTERN records and the planes of the
 * source chunk of ILM_SPAWN_FEEDBACK records, per spawn record slot */
typedef struct OrcStepExtras {
    const IlmFloat4* spawn_positions[ILM_MAX_SPAWNS];
    int32_t          spawn_position_count[ILM_MAX_SPAWNS];
    const IlmFloat4* source_pos[ILM_MAX_SPAWNS];
    const IlmFloat4* source_vel[ILM_MAX_SPAWNS];
    const IlmFloat4* source_attr[ILM_MAX_SPAWNS];
    const uint16_t*  low_precision_rnd;       /* optional: the Rgba64 randomness copy (computed per call when NULL) */
    const IlmFloat4* spawn_pattern[ILM_MAX_SPAWNS];       /* ILM_SPAWN_PATTERN: mip levels back to back */
    int32_t          pattern_w[ILM_MAX_SPAWNS], pattern_h[ILM_MAX_SPAWNS], pattern_levels[ILM_MAX_SPAWNS];
} OrcStepExtras;
void orc_step_ex(IlmFloat4** planes, int32_t chunk_count, int32_t chunk_size,
                 const IlmFloat4* rnd, int32_t rw, int32_t rh,
                 const IlmFloat4* life_ramp, int32_t ramp_w, int32_t ramp_h,
                 const OrcTexture* sdf, const IlmStepDesc* desc, uint32_t* live_counts, const OrcStepExtras* extras);

/* bezier / distance field primitives exposed for known-answer tests */
float orc_bezier1(const IlmClampedBezier1* b, float value);
void  orc_bezier4(const IlmClampedBezier4* b, float value, IlmFloat4* out);
float orc_sample_distance_field(const float pos[3], const IlmDistanceFieldUniforms* df, const OrcTexture* sdf);
float orc_encode_distance(float distance, float max_encoded);
float orc_decode_distance(float encoded, float max_encoded);
float orc_evaluate_area(int32_t type_id, const float pos[3], const float center[3], const float size[3], float rotation);

/* lighting */
/* RampTexture of the light group rendered by the following orc_render_sphere_lights / orc_render_light_probes calls (width * height
 * float4, kept by reference); NULL unbinds (techniques without a distance ramp) */
void orc_set_light_ramp(const IlmFloat4* texels, int32_t width, int32_t height);
/* 0: fp32 accumulation over the lights (default); 1: the reference's HalfVector4 lightmap, rounded by the ROP after every light
 * (LightingRenderer.cs:476-479) -- see ilm_ctx_set_lightmap_blend */
void orc_set_lightmap_blend(int32_t mode);
void orc_sample_gbuffer(float px, float py, const IlmEnvironment* env, const OrcTexture* gbuffer,
                        float world_pos[3], float normal[3], int32_t* enable_shadows, int32_t* fullbright, float camera_pos[3]);
void orc_render_sphere_lights(const IlmLightVertex* lights, int32_t light_count,
                              const IlmEnvironment* env, const IlmDistanceFieldUniforms* df,
                              const OrcTexture* gbuffer, const OrcTexture* sdf,
                              const float ambient[4],
                              IlmFloat4* lightmap, int32_t width, int32_t height,
                              int32_t row_begin, int32_t row_end, IlmRenderStats* stats);

/* particle lights and light probes, ilm_oracle_lights.c; the lightmap is accumulated into (additive blend) */
void orc_render_particle_lights(IlmFloat4** planes, int32_t chunk_count, const int32_t* quad_counts,
                                const IlmParticleLightParams* p,
                                const IlmEnvironment* env, const IlmDistanceFieldUniforms* df,
                                const OrcTexture* gbuffer, const OrcTexture* sdf,
                                IlmFloat4* lightmap, int32_t width, int32_t height,
                                int32_t row_begin, int32_t row_end, IlmRenderStats* stats);
void orc_render_light_probes(const IlmLightVertex* lights, int32_t light_count,
                             const IlmFloat4* probe_positions, const IlmFloat4* probe_normals, int32_t probe_count,
                             const IlmEnvironment* env, const IlmDistanceFieldUniforms* df, const OrcTexture* sdf,
                             IlmFloat4* out_values);

/* output side, ilm_oracle_output.c */
int32_t orc_fill_readback_result(IlmFloat4** planes, int32_t chunk_count, const int32_t* element_counts, int32_t slots,
                                 const IlmReadbackParams* p, IlmReadbackDrawCall* out, int32_t capacity);
void orc_resolve_lighting(const IlmFloat4* lightmap, int32_t width, int32_t height, const IlmHDRConfiguration* hdr,
                          IlmFloat4* out, int32_t row_begin, int32_t row_end);
/* technique RasterizeParticlesNoTexture: blends the live particles of the chunks onto image (width * height float4) in chunk / slot
 * order; stats (may be NULL): [0] live quads, [1] shaded pixels */
void orc_render_particles(IlmFloat4** planes, int32_t chunk_count, const int32_t* quad_counts, int32_t slots,
                          const IlmRasterizeParams* p, IlmFloat4* image, int32_t width, int32_t height, uint64_t* stats);
/* techniques RasterizeParticlesTexturePoint / TextureLinear on a bitmap without mips (p->BitmapFilter) */
void orc_render_particles_textured(IlmFloat4** planes, int32_t chunk_count, const int32_t* quad_counts, int32_t slots,
                                   const IlmRasterizeParams* p, const IlmFloat4* bitmap, int32_t bitmap_w, int32_t bitmap_h,
                                   IlmFloat4* image, int32_t width, int32_t height, uint64_t* stats);

/* host-side integer/layout logic */
typedef struct OrcDistanceFieldLayout {
    int32_t virtual_width, virtual_height;
    float   virtual_depth;
    double  resolution;
    int32_t slice_width, slice_height, slice_count, physical_slice_count;
    int32_t column_count, row_count, atlas_width, atlas_height;
    int32_t maximum_encoded_distance;
} OrcDistanceFieldLayout;
void orc_distance_field_layout(int32_t virtual_width, int32_t virtual_height, float virtual_depth,
                               int32_t requested_slice_count, double requested_resolution,
                               int32_t maximum_encoded_distance, OrcDistanceFieldLayout* out);
void orc_distance_field_uniforms(const OrcDistanceFieldLayout* l, int32_t valid_slice_count, float z_offset,
                                 float max_cone_radius, float occlusion_to_opacity_power, int32_t step_limit,
                                 float min_step_size, float long_step_factor, IlmDistanceFieldUniforms* out);

/* SpawnerBase.BeginTick / EndTick + RunSpawner slot allocation */
typedef struct OrcSpawnerState {
    double  rate_error;
    int32_t total_spawned;
} OrcSpawnerState;
int32_t orc_spawner_begin_tick(OrcSpawnerState* s, float min_rate, float max_rate, int32_t count_scale,
                               double rng_draw, double delta_time_seconds, int32_t maximum_total /* <0 => none */);
void    orc_spawner_end_tick(OrcSpawnerState* s, int32_t requested, int32_t actual);

/* distance-field generation, ilm_oracle_fields.c: renders the listed slice triplets into `atlas`
 * (RGBA16 texels of SliceWidth*ColumnCount x SliceHeight*RowCount); clear_source may be NULL */
void orc_render_distance_field_slices(uint16_t* atlas, int32_t format, const uint16_t* clear_source,
                                      const IlmDistanceFieldRenderDesc* desc,
                                      const int32_t* first_virtual_slices, int32_t triplet_count,
                                      const IlmObstruction* obstructions, int32_t obstruction_count,
                                      const IlmHeightVolume* volumes, int32_t volume_count,
                                      const float* polygon_xy, int32_t polygon_vertex_count);

void orc_render_gbuffer(IlmFloat4* out, int32_t width, int32_t height, const IlmGBufferRenderDesc* desc,
                        const IlmHeightVolume* volumes, int32_t volume_count, const float* polygon_xy);

/* RenderGBuffer with the host's meshes (2.5D top / front faces under the depth test, billboards): ilm_oracle_gbuffer.c.
 * textures: one per run, texels NULL = nothing bound, format ILM_LIGHTMAP_*. */
void orc_render_gbuffer_meshes(IlmFloat4* out, int32_t width, int32_t height, const IlmGBufferMeshDesc* desc,
                               const IlmHeightVolumeVertex* top, int32_t top_count,
                               const IlmHeightVolumeVertex* front, int32_t front_count,
                               const IlmBillboardVertex* billboards, int32_t billboard_vertex_count,
                               const IlmBillboardRun* runs, int32_t run_count, const OrcTexture* textures);

int32_t orc_num_threads(void);
void    orc_set_num_threads(int32_t n);

#ifdef __cplusplus
}
#endif
#endif
