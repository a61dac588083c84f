/*
 * illuminant_hip.h -- C ABI of libilluminant_hip.so
 *
 * The drop-in boundary for the two data-parallel hot paths of sq/Illuminant on
 * MI355X (gfx950): the ParticleEngine per-chunk state update and the
 * LightingRenderer sphere-light SDF cone trace.  Everything here is plain C:
 * POD structs whose byte layout is the reference's own uniform / vertex
 * structs (so C# can pass them with `ref` / `fixed`), opaque handles for
 * device memory, int32 return codes (0 = ok).
 *
 * All file:line citations are relative to the reference checkout
 * (sq/Illuminant @ 2025-08-29).  The reference interface each entry point
 * replaces is cited on the entry point.  INTEGRATION.md shows the P/Invoke
 * declarations a maintainer of the reference would add.
 *
 * Threading: a context owns one HIP stream; calls on one context are
 * asynchronous and ordered; callers serialise calls per context (this is what
 * the reference's `lock (_UpdateParameterPool)` / issue-thread ordering gives,
 * Illuminant/Particles/ParticleSystem.cs:681).  Different contexts may be driven
 * from different threads; the ilm_debug_* switches are process-wide (atomic) and
 * change the behaviour of every context.  Host output buffers are valid
 * after the call returns (download / count calls synchronise the stream).
 */
#ifndef ILLUMINANT_HIP_H
#define ILLUMINANT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 9 (r06): + ilm_group_gather_chunks (the sharded particle state made whole on every member: Pos+Life for global consumers, Pos+Life
 * and RenderColor for particle lights across ranks).  Nothing removed or changed in layout.
 * 8 (r05): + ilm_group_lightmap_store_mode, ILM_GATHER_STORE, ilm_debug_last_light_launch, ilm_ctx_create_sibling,
 * ILM_GATHER_ASYNC + ilm_group_lightmap_wait.
 * 7 (r04): + ilm_ctx_set_light_split, ilm_sdf_mark_dirty, ilm_sdf_trace_info / IlmSdfTraceInfo; ilm_group_lightmap_set_strips became a
 * collective with one process per GPU.  Nothing was removed or changed in layout since 6. */
#define ILM_ABI_VERSION 9

/* ---- return codes ------------------------------------------------------ */
#define ILM_OK                    0
#define ILM_ERR_INVALID_ARGUMENT  (-1)
#define ILM_ERR_INVALID_HANDLE    (-2)
#define ILM_ERR_OUT_OF_RANGE      (-3)
#define ILM_ERR_TOO_MANY          (-4)   /* e.g. >16 attractors: Transforms.cs:348-349 */
#define ILM_ERR_NO_DEVICE         (-5)
#define ILM_ERR_STATE             (-6)   /* e.g. DF update without a bound field: ParticleSystem.cs:835-836 */
/* positive values are hipError_t values passed through unchanged */

/* Opaque object handles (contexts, engines, systems, textures).  Every entry point looks its handles up in a table of live
 * objects first: a destroyed, foreign or garbage value returns ILM_ERR_INVALID_HANDLE, it is never dereferenced.  0 is never
 * a valid handle.  A create call that fails releases everything it had allocated and leaves *out = 0.  Parents outlive children:
 * destroying a context (engine) that still has live objects (systems) returns ILM_ERR_STATE and destroys nothing. */
typedef uint64_t IlmHandle;

/* ---- POD mirrors of the reference's uniform structs -------------------- */

typedef struct IlmFloat4 { float x, y, z, w; } IlmFloat4;

/* XNA Matrix, row-major M11..M44; shaders use mul(rowVector, M)
 * (Illuminant/Shaders/SpawnerCommon.fxh:166,179). */
typedef struct IlmMatrix { float m[16]; } IlmMatrix;

/* Uniforms.ParticleSystem -- Illuminant/Uniforms.cs:197-236,
 * accessors Illuminant/Shaders/ParticleCommon.fxh:29-92. */
typedef struct IlmParticleSystemUniforms {
    IlmFloat4 GlobalSettings;     /* deltaTimeMilliseconds, friction, maximumVelocity, lifeDecayRate */
    IlmFloat4 CollisionSettings;  /* escapeVelocity, bounceVelocityMultiplier, collisionDistance, collisionLifePenalty */
    IlmFloat4 TexelAndSize;       /* 1/ChunkSize, 1/ChunkSize, Size.X, Size.Y */
    IlmFloat4 AnimationRateAndRotationAndZToY; /* rate_x, rate_y, velocityRotation, zToY */
} IlmParticleSystemUniforms;

/* Uniforms.ClampedBezier1 / ClampedBezier4 -- Illuminant/Bezier.cs:433-441,588-599;
 * shader side Illuminant/Shaders/Bezier.fxh:6-19. */
typedef struct IlmClampedBezier1 { IlmFloat4 RangeAndCount, ABCD; } IlmClampedBezier1;
typedef struct IlmClampedBezier4 { IlmFloat4 RangeAndCount, A, B, C, D; } IlmClampedBezier4;

/* Uniforms.DistanceField -- Illuminant/Uniforms.cs:79-88 -- followed by the
 * separately-bound DistanceFieldPacked1 (Illuminant/Lighting/LightingRenderer.cs:1933-1939,
 * Illuminant/Shaders/DistanceFieldCommon.fxh:189-206). */
typedef struct IlmDistanceFieldUniforms {
    IlmFloat4 ConeAndMisc;              /* MaxConeRadius, DistanceFieldZOffset, OcclusionToOpacityPower, InvScaleFactorX */
    IlmFloat4 TextureSliceAndTexelSize; /* 1/cols, 1/rows, 1/(virtualW*cols), 1/(virtualH*rows) */
    IlmFloat4 StepAndMisc2;             /* StepLimit, MinimumLength, LongStepFactor, InvScaleFactorY */
    IlmFloat4 TextureSliceCount;        /* cols, rows, validZ, sliceCount */
    IlmFloat4 Extent;                   /* virtualW, virtualH, virtualDepth, maximumEncodedDistance */
    IlmFloat4 Packed1;                  /* 1/(3*cols), sliceCount/extentZ, validZ, minStepSize */
} IlmDistanceFieldUniforms;

/* Uniforms.Environment -- Illuminant/Uniforms.cs:14-24 -- plus the G-buffer
 * uniforms bound by SetGBufferParameters (Illuminant/Lighting/LightingRenderer.GBuffer.cs:520-534)
 * and the view-transform values the pixel shader reads through
 * GetViewportPosition()/GetViewportScale() (Illuminant/Shaders/LightCommon.fxh:27-33). */
typedef struct IlmEnvironment {
    IlmFloat4 ZAndScale;          /* GroundZ, MaximumZ, RenderScale.X, RenderScale.Y */
    IlmFloat4 ZToY;               /* ZToYMultiplier, InvZToYMultiplier, LightOcclusion, unused */
    IlmFloat4 GBufferTexelSizeAndMisc; /* 1/gbufW, 1/gbufH, ViewportScaleX, ViewportScaleY; xy==0 => no G-buffer */
    float     ViewportPosition[2];
    float     GBufferViewportRelative;
    float     _pad0;
} IlmEnvironment;

/* LightVertex, 8 x float4, Pack=4 -- Illuminant/Vertices.cs:10-39; filled by
 * RenderSphereLightSource, Illuminant/Lighting/LightingRenderer.cs:1193-1219. */
typedef struct IlmLightVertex {
    IlmFloat4 LightPosition1, LightPosition2, LightPosition3;
    IlmFloat4 LightProperties;         /* radius, rampLength, rampMode, castsShadows */
    IlmFloat4 MoreLightProperties;     /* aoRadius, shadowDistanceFalloff, falloffYFactor, aoOpacity */
    IlmFloat4 EvenMoreLightProperties; /* shadowFilter, 0, rampOffset, rampRate */
    IlmFloat4 Color1;                  /* rgb, a*opacity*intensityScale */
    IlmFloat4 Color2;                  /* specular rgb, specular power */
} IlmLightVertex;

/* ---- particle transform parameters ------------------------------------- */

/* ParticleAreaTransform.SetParameters -- Illuminant/Particles/ParticleTransform.cs:294-318. */
typedef struct IlmAreaParams {
    int32_t AreaType;             /* 0 none, 1 ellipsoid, 2 box, 3 cylinder, 4 spheroid, 5 octagon */
    float   Strength;
    float   AreaFalloff;
    float   AreaRotation;
    float   AreaCenter[3];  float _pad0;
    float   AreaSize[3];    float _pad1;
    float   CategoryFilter[2];    /* default (-9999, 9999) */
    float   _pad2[2];
} IlmAreaParams;

#define ILM_MAX_ATTRACTORS 16     /* Illuminant/Shaders/Gravity.fx:3 */

/* Gravity.SetParameters -- Illuminant/Particles/Transforms.cs:347-365; shader
 * Illuminant/Shaders/Gravity.fx:5-10.  NOTE: Gravity derives from ParticleTransform,
 * not ParticleAreaTransform, so the reference never binds CategoryFilter for it
 * and the effect default (0,0) applies; the field is explicit here. */
typedef struct IlmGravityParams {
    int32_t AttractorCount;
    float   MaximumAcceleration;
    float   CategoryFilter[2];
    float   AttractorPositions[ILM_MAX_ATTRACTORS][3];
    float   AttractorRadiusesAndStrengths[ILM_MAX_ATTRACTORS][3]; /* radius, strength, type */
} IlmGravityParams;

/* FMA.SetParameters -- Illuminant/Particles/Transforms.cs:38-45; shader Illuminant/Shaders/FMA.fx:4-13. */
typedef struct IlmFMAParams {
    IlmAreaParams Area;
    float     TimeDivisor;  float _pad[3];
    IlmFloat4 PositionAdd, PositionMultiply;
    IlmFloat4 VelocityAdd, VelocityMultiply;
} IlmFMAParams;

/* Noise.SetParameters -- Illuminant/Particles/Transforms.cs:243-268; shader Illuminant/Shaders/Noise.fx:5-19. */
typedef struct IlmNoiseParams {
    IlmAreaParams Area;
    float     TimeDivisor;
    float     FrequencyLerp;
    float     ReplaceOldVelocity;
    float     _pad;
    float     RandomnessOffset[2];
    float     NextRandomnessOffset[2];
    IlmFloat4 PositionOffset, PositionMinimum, PositionScale;
    IlmFloat4 VelocityOffset, VelocityMinimum, VelocityScale;
} IlmNoiseParams;

#define ILM_MAX_INLINE_POSITION_CONSTANTS 4   /* Illuminant/Shaders/SpawnerCommon.fxh:1 */

/* SpawnerBase.SetParameters + Spawner.SetParameters --
 * Illuminant/Particles/ParticleSpawner.cs:200-256,376-403; shader uniforms
 * Illuminant/Shaders/SpawnerCommon.fxh:3-15. */
typedef struct IlmSpawnParams {
    float     ChunkSizeAndIndices[4];   /* chunkSize, first, last, positionIndexOffset */
    IlmFloat4 Configuration[9];
    float     FormulaTypes[4];
    IlmMatrix PositionMatrix, VelocityMatrix;
    float     AxisMask[3];
    float     AlignVelocityAndPosition;
    float     RandomnessOffset[2];
    float     AttributeDiscardThreshold;  /* AlphaDiscardThreshold / 255 */
    float     PolygonRate;
    float     PolygonLoop;
    float     PositionConstantCount;
    float     _pad[2];
    IlmFloat4 InlinePositionConstants[ILM_MAX_INLINE_POSITION_CONSTANTS];
} IlmSpawnParams;

/* Everything SetSystemUniforms (Illuminant/Particles/ParticleSystem.cs:547-575) and
 * UpdateHandler._BeforeDraw (Illuminant/Particles/ParticleTransform.cs:84-168) bind
 * for the Update pass. */
typedef struct IlmUpdateParams {
    IlmClampedBezier4 ColorFromLife, ColorFromVelocity;
    IlmClampedBezier1 SizeFromLife, SizeFromVelocity;
    float     RotationFromLifeAndIndex[2];   /* radians */
    float     _pad[2];
    IlmFloat4 LifeRampSettings;              /* strength, min, divisor, indexDivisor; x==0 => off */
} IlmUpdateParams;

/* MatrixMultiply.SetParameters -- Illuminant/Particles/Transforms.cs:61-66; shader
 * Illuminant/Shaders/MatrixMultiply.fx:4-52 (mul3: Illuminant/Shaders/ParticleCommon.fxh:183-196). */
typedef struct IlmMatrixMultiplyParams {
    IlmAreaParams Area;
    float     TimeDivisor;  float _pad[3];     /* 1000 / CyclesPerSecond, or -1 => timeScale 1 */
    IlmMatrix PositionMatrix, VelocityMatrix;
} IlmMatrixMultiplyParams;

/* SpatialNoise.SetParameters -- Illuminant/Particles/Transforms.cs:275-300 on top of Noise's; shader
 * Illuminant/Shaders/Noise.fx:17,74-116 (smoothRandomCustom: bilinear, WRAP, on the Rgba64 copy of the
 * randomness table, Illuminant/Shaders/RandomCommon.fxh:7-15,36-39, Illuminant/Particles/ParticleEngine.cs:508-540). */
typedef struct IlmSpatialNoiseParams {
    IlmNoiseParams Noise;            /* PositionMinimum / VelocityMinimum are not read by this technique */
    float     SpaceScale[2];         /* 1 / SpaceScale.X, 1 / SpaceScale.Y */
    float     _pad[2];
} IlmSpatialNoiseParams;

enum {
    ILM_OP_GRAVITY         = 1,   /* technique Gravity,        Illuminant/Shaders/Gravity.fx:12-61        */
    ILM_OP_NOISE           = 2,   /* technique Noise,          Illuminant/Shaders/Noise.fx:28-72          */
    ILM_OP_FMA             = 3,   /* technique FMA,            Illuminant/Shaders/FMA.fx:15-51            */
    ILM_OP_MATRIX_MULTIPLY = 4,   /* technique MatrixMultiply, Illuminant/Shaders/MatrixMultiply.fx:22-52 */
    ILM_OP_SPATIAL_NOISE   = 5    /* technique SpatialNoise,   Illuminant/Shaders/Noise.fx:74-116         */
};

typedef struct IlmTransformOp {
    int32_t Type;
    int32_t _pad[3];
    union {
        IlmGravityParams        Gravity;
        IlmNoiseParams          Noise;
        IlmFMAParams            FMA;
        IlmMatrixMultiplyParams MatrixMultiply;
        IlmSpatialNoiseParams   SpatialNoise;
    } u;
} IlmTransformOp;

enum {
    ILM_UPDATE_NONE                = 0,  /* transforms only */
    ILM_UPDATE_POSITIONS           = 1,  /* technique UpdatePositions, UpdateParticleSystem.fx:9-38 */
    ILM_UPDATE_WITH_DISTANCE_FIELD = 2,  /* technique UpdateWithDistanceField, UpdateParticleSystemWithDistanceField.fx:29-147 */
    ILM_UPDATE_ERASE               = 3   /* technique Erase, UpdateParticleSystem.fx:40-49 */
};

#define ILM_MAX_OPS    4
#define ILM_MAX_SPAWNS 2

#define ILM_STEP_COUNT_LIVE  1u   /* also produce per-chunk live counts (CountLiveParticles.fx) */

enum {
    ILM_SPAWN_INLINE           = 0,  /* technique SpawnParticles (<= 4 inline positions), SpawnParticles.fx:10-30 */
    ILM_SPAWN_POSITION_BUFFER  = 1,  /* technique SpawnParticlesFromPositionTexture, SpawnParticles.fx:32-52: positions from
                                        the list bound with ilm_system_set_spawn_positions */
    ILM_SPAWN_FEEDBACK         = 2,  /* technique SpawnFeedbackParticles, SpawnParticles.fx:54-118: one source particle of
                                        another system's chunk per InstanceMultiplier new particles */
    ILM_SPAWN_PATTERN          = 3   /* technique SpawnPatternParticles, PatternSpawner.fx:21-97: one particle per Divisor x Divisor
                                        block of the texture bound with ilm_system_set_spawn_pattern */
};

/* FeedbackSpawner.SetParameters (Illuminant/Particles/SpecialSpawners.cs:411-427) + the source chunk bound by
 * UpdateHandler._BeforeDraw (Illuminant/Particles/ParticleTransform.cs:129-141, SourceChunkSizeAndTexel). */
typedef struct IlmFeedbackParams {
    IlmHandle SourceSystem;            /* system that owns the source chunk (same engine chunk size) */
    int32_t   SourceChunkIndex;        /* index in the source system's chunk table */
    float     FeedbackSourceIndex;     /* first source slot */
    float     InstanceMultiplier;
    float     SourceVelocityFactor;
    float     AlignPositionConstant, MultiplyLife, MultiplyAttributeConstant;
    float     SourceLifeRange[2];
    float     _pad;
} IlmFeedbackParams;

/* PatternSpawner.SetParameters (Illuminant/Particles/SpecialSpawners.cs:208-256); uniforms of
 * Illuminant/Shaders/PatternSpawner.fx:6-9. */
typedef struct IlmPatternParams {
    float StepWidthAndSizeScale[4];    /* Divisor, ParticlesPerRow, Divisor / texWidth, Divisor / texHeight */
    float YOffsetsAndCoordScale[4];    /* currentRow, currentRow * Divisor / texHeight, Divisor, Divisor */
    float TexelOffsetAndMipBias[4];    /* -0.5 / texWidth + baseX, -0.5 / texHeight + baseY, 0, log2(Divisor) + MipBiasBase */
    float CenteringOffset[2];          /* DirectTextureSize * -0.5 */
    float MultiplyAttributeConstant;   /* != 0: pattern colour * Configuration[5], else + */
    float _pad;
} IlmPatternParams;

typedef struct IlmSpawnRecord {
    int32_t        ChunkIndex;    /* index in the system's chunk table */
    int32_t        Kind;          /* ILM_SPAWN_* */
    int32_t        _pad[2];
    IlmSpawnParams Params;
    IlmFeedbackParams Feedback;   /* read when Kind == ILM_SPAWN_FEEDBACK */
    IlmPatternParams  Pattern;    /* read when Kind == ILM_SPAWN_PATTERN */
} IlmSpawnRecord;

/* One ParticleSystem.Update's worth of GPU work (Illuminant/Particles/ParticleSystem.cs:725-745):
 * spawns first, then for every chunk in [FirstChunk, FirstChunk+ChunkCount) each
 * transform in order, then the update pass.  Executed as ONE kernel launch with
 * every pass applied in registers; pass-ordering semantics are preserved per slot. */
typedef struct IlmStepDesc {
    int32_t  FirstChunk, ChunkCount;   /* ChunkCount < 0 => all chunks */
    int32_t  OpCount, SpawnCount;
    int32_t  UpdateMode;
    uint32_t Flags;
    int32_t  _pad[2];
    IlmParticleSystemUniforms System;
    IlmUpdateParams           Update;
    IlmDistanceFieldUniforms  DistanceField;  /* used when UpdateMode == ILM_UPDATE_WITH_DISTANCE_FIELD */
    IlmTransformOp            Ops[ILM_MAX_OPS];
    IlmSpawnRecord            Spawns[ILM_MAX_SPAWNS];
} IlmStepDesc;

/* ---- formats ------------------------------------------------------------ */
enum {
    ILM_SDF_UNORM16 = 0,  /* SurfaceFormat.Rgba64, Illuminant/SDF/DistanceField.cs:22 */
    ILM_SDF_FP16    = 1   /* same atlas, channels stored as IEEE half of the encoded value */
};
enum {
    ILM_GBUFFER_FLOAT4 = 0, /* SurfaceFormat.Vector4     (Illuminant/GBuffer.cs:30-38) */
    ILM_GBUFFER_HALF4  = 1  /* SurfaceFormat.HalfVector4 */
};
enum {
    ILM_LIGHTMAP_FLOAT4 = 0, /* fp32 accumulation written unrounded (parity format) */
    ILM_LIGHTMAP_HALF4  = 1, /* HalfVector4 lightmap, Illuminant/Lighting/LightingRenderer.cs:476-479 */
    ILM_LIGHTMAP_RGBA8  = 2  /* SurfaceFormat.Color lightmap (low quality) */
};

/* Which particle attribute planes to move in upload / download. */
enum {
    ILM_PLANE_POSITION     = 0,  /* PositionAndLife  (xyz, life)      ParticleSystem.cs:73-146 */
    ILM_PLANE_VELOCITY     = 1,  /* Velocity         (xyz, category)  */
    ILM_PLANE_ATTRIBUTES   = 2,  /* Chunk.Color      (rgba)           ParticleSystem.cs:159 */
    ILM_PLANE_RENDER_COLOR = 3,  /* Chunk.RenderColor (premultiplied) ParticleSystem.cs:160 */
    ILM_PLANE_RENDER_DATA  = 4   /* Chunk.RenderData (size, rotation, speed, category) ParticleSystem.cs:161 */
};

/* ---- library / context -------------------------------------------------- */

int32_t     ilm_abi_version(void);
/* Last error text for the calling thread ("" if none). */
const char* ilm_last_error(void);
/* Number of visible HIP devices (0 when there is no GPU; never fails). */
int32_t     ilm_device_count(void);

/* The numbers the kernels take from the reference's text, by the reference's own names ("ConeTrace.fxh:FULLY_SHADOWED_THRESHOLD",
 * "SpawnerCommon.fxh:randomOffset1.x modulus", ...; csrc/reference_constants.hpp).  Diagnostic, needs no GPU: tests/test_reference_pin.py
 * compares every entry with the values tools/pin_reference_constants.py extracts from the reference sources. */
int32_t     ilm_debug_reference_constant(const char* key, double* out_value);
int32_t     ilm_debug_reference_constant_count(void);
const char* ilm_debug_reference_constant_key(int32_t index);

/* One context per GPU.  Replaces the GraphicsDevice/RenderCoordinator the
 * reference threads through ParticleEngine (Illuminant/Particles/ParticleEngine.cs:95-141)
 * and LightingRenderer (Illuminant/Lighting/LightingRenderer.cs:486-560). */
int32_t ilm_ctx_create(int32_t device_id, IlmHandle* out_ctx);
/* A second (third, ...) context on the device of `ctx` for FRAMES IN FLIGHT (r05).  The reference keeps a ring of lightmaps
 * (BufferRing, Illuminant/Lighting/LightingRenderer.cs:472-485) because frame N + 1 is built while frame N renders; here a launch on a
 * context's stream starts when the previous launch of that stream has drained, so a host that wants the next frame's first waves to fill
 * the slots the previous frame's tail leaves empty renders alternate frames on sibling contexts: each has its own stream, scratch and
 * lightmaps, and -- unlike unrelated contexts -- the light passes of one (ilm_render_sphere_lights, ilm_render_particle_lights) may READ
 * the distance fields and G-buffers of another.  The library orders those reads against the owner's writes (uploads, ilm_sdf_render_slices,
 * ilm_gbuffer_render*, the field's cell rebuild -- all of which run on the OWNER's stream) with events, nothing blocks the host; writes
 * the library cannot see (ilm_sdf_device_ptr) are the host's to order.  Sibling contexts that share objects are driven from one thread.
 * Destroy like any context. */
int32_t ilm_ctx_create_sibling(IlmHandle ctx, IlmHandle* out_ctx);
int32_t ilm_ctx_destroy(IlmHandle ctx);
/* Block until all work queued on the context has finished. */
int32_t ilm_ctx_sync(IlmHandle ctx);
/* Raw hipStream_t of the context (for interop with RCCL / torch streams). */
int32_t ilm_ctx_stream(IlmHandle ctx, void** out_stream);

/* GPU-side timers around a region of queued work (hipEvent pair on the context
 * stream); elapsed time in milliseconds is returned by ilm_timer_stop after it
 * synchronises on the stop event. */
int32_t ilm_timer_start(IlmHandle ctx);
int32_t ilm_timer_stop(IlmHandle ctx, float* out_ms);

/* ---- particle engine ----------------------------------------------------- */

/* ParticleEngine ctor: chunk size + the 807x653 float4 randomness table
 * (Illuminant/Particles/ParticleEngine.cs:45-46,495-544).  The reference fills
 * it from an unseeded RNG, so it is an explicit input here.  `randomness` is a
 * host pointer to width*height float4 texels, row-major. */
#define ILM_RANDOMNESS_WIDTH  807
#define ILM_RANDOMNESS_HEIGHT 653
int32_t ilm_engine_create(IlmHandle ctx, int32_t chunk_size,
                          const IlmFloat4* randomness, int32_t rand_width, int32_t rand_height,
                          IlmHandle* out_engine);
int32_t ilm_engine_destroy(IlmHandle engine);

/* ParticleSystem: a table of chunks of chunk_size^2 slots each
 * (Illuminant/Particles/ParticleSystem.cs:148-240).  State is stored SoA, one
 * buffer set per chunk, updated in place (the reference's Previous/Current
 * ping-pong exists only to satisfy render-target hazards,
 * ParticleSystem.cs:577-616). */
int32_t ilm_system_create(IlmHandle engine, IlmHandle* out_system);
int32_t ilm_system_destroy(IlmHandle system);
/* CreateChunk (ParticleSystem.cs:349-384): appends a zero-filled chunk, returns its table index. */
int32_t ilm_system_add_chunk(IlmHandle system, int32_t* out_chunk_index);
/* Reap (ParticleLiveness.cs:120-129): removes the chunk at `chunk_index`; later chunks shift down by one. */
int32_t ilm_system_remove_chunk(IlmHandle system, int32_t chunk_index);
int32_t ilm_system_chunk_count(IlmHandle system, int32_t* out_count);

/* ParticleSystem.Spawn(initializers) upload path (ParticleSpawning.cs:13-113) /
 * AutoReadback (ParticleReadback.cs:21-71).  Host buffers are AoS float4,
 * `count` slots starting at slot `first_slot`. */
int32_t ilm_chunk_upload(IlmHandle system, int32_t chunk_index, int32_t plane,
                         const IlmFloat4* src, int32_t first_slot, int32_t count);
int32_t ilm_chunk_download(IlmHandle system, int32_t chunk_index, int32_t plane,
                           IlmFloat4* dst, int32_t first_slot, int32_t count);
/* Device pointer of one SoA component array of a chunk (component 0..19 =
 * x,y,z,life, vx,vy,vz,category, attr rgba, renderColor rgba, renderData xyzw);
 * `out_stride_floats` is the distance between component arrays.  For zero-copy
 * consumers (RCCL all-gather of positions, renderers). */
int32_t ilm_chunk_device_ptr(IlmHandle system, int32_t chunk_index, int32_t component,
                             void** out_ptr, int64_t* out_stride_floats);

/* Bind the distance field particles collide with
 * (ParticleCollision.DistanceField, ParticleConfiguration.cs:20-24); 0 unbinds. */
int32_t ilm_system_set_distance_field(IlmHandle system, IlmHandle sdf);
/* Optional life ramp texture (ParticleSystem.cs:911-941): width*height float4, POINT, U clamp / V wrap. */
int32_t ilm_system_set_life_ramp(IlmHandle system, const IlmFloat4* texels, int32_t width, int32_t height);

/* The Spawner's PositionBuffer texture (Illuminant/Particles/ParticleSpawner.cs:301-353): `count` (xyz, life) position
 * constants for spawn record `spawn_slot` (0 .. ILM_MAX_SPAWNS-1) of this system, used by records of kind
 * ILM_SPAWN_POSITION_BUFFER.  The reference pads the texture width to a multiple of 128 and addresses it with
 * index * (1 / width), POINT / CLAMP; the same arithmetic is applied here.  count == 0 releases the list. */
int32_t ilm_system_set_spawn_positions(IlmHandle system, int32_t spawn_slot, const IlmFloat4* positions, int32_t count);

/* The PatternSpawner's texture (Illuminant/Particles/SpecialSpawners.cs:19-22,255; sampler PatternSampler,
 * Illuminant/Shaders/PatternSpawner.fx:11-19: CLAMP, LINEAR min/mag, POINT mip) for spawn record `spawn_slot`, used by records of
 * kind ILM_SPAWN_PATTERN.  `texels` holds `levels` mip levels back to back, level l being max(1, width >> l) x max(1, height >> l)
 * float4 texels, row-major (the reference's mip chain comes from its texture loader, which is outside the tree: it is an explicit
 * input here; levels == 1 is a texture without mips).  tex2Dlod's explicit LOD picks level clamp(floor(lod + 0.5), 0, levels - 1).
 * levels == 0 releases the texture. */
int32_t ilm_system_set_spawn_pattern(IlmHandle system, int32_t spawn_slot, const IlmFloat4* texels,
                                     int32_t width, int32_t height, int32_t levels);

/* The hot path.  Replaces RunSpawner + the UpdateChunk loop of
 * ParticleSystem.Update (ParticleSystem.cs:725-745, 791-856): every RunTransform
 * draw for every chunk becomes one launch. */
int32_t ilm_system_step(IlmHandle system, const IlmStepDesc* desc);

/* Single-pass entry points, one per reference technique (LoadMaterials.cs:387-522);
 * chunk_index < 0 => every chunk.  Thin wrappers over ilm_system_step. */
int32_t ilm_spawn  (IlmHandle system, int32_t chunk_index, const IlmParticleSystemUniforms* sys, const IlmSpawnParams* p);
int32_t ilm_gravity(IlmHandle system, int32_t chunk_index, const IlmParticleSystemUniforms* sys, const IlmGravityParams* p);
int32_t ilm_noise  (IlmHandle system, int32_t chunk_index, const IlmParticleSystemUniforms* sys, const IlmNoiseParams* p);
int32_t ilm_fma    (IlmHandle system, int32_t chunk_index, const IlmParticleSystemUniforms* sys, const IlmFMAParams* p);
int32_t ilm_matrix_multiply(IlmHandle system, int32_t chunk_index, const IlmParticleSystemUniforms* sys, const IlmMatrixMultiplyParams* p);
int32_t ilm_spatial_noise  (IlmHandle system, int32_t chunk_index, const IlmParticleSystemUniforms* sys, const IlmSpatialNoiseParams* p);
int32_t ilm_update (IlmHandle system, int32_t chunk_index, const IlmParticleSystemUniforms* sys, const IlmUpdateParams* p,
                    const IlmDistanceFieldUniforms* df /* NULL => UpdatePositions */);
int32_t ilm_erase  (IlmHandle system, int32_t chunk_index);

/* Liveness (CountLiveParticles.fx:5-40 + ProcessLivenessInfoData, ParticleEngine.cs:224-252):
 * out_counts[i] = #{slots of chunk i with life > 0}; when saturate16 != 0 the
 * value is min(count, 65535) exactly as the reference's 16-bit additive target decodes. */
int32_t ilm_system_live_counts(IlmHandle system, uint32_t* out_counts, int32_t capacity, int32_t saturate16);

/* Counts produced inside the last ilm_system_step that had ILM_STEP_COUNT_LIVE set
 * (the ballot/popcount reduction fused into the update kernel); synchronises. */
int32_t ilm_system_step_counts(IlmHandle system, uint32_t* out_counts, int32_t capacity, int32_t saturate16);

/* Non-blocking form (the reference reads liveness back a frame or more later through
 * LivenessDataReadbackWorkItem, ParticleWorkItems.cs:98-135): *out_ready = 1 and the counts are
 * written when the last counting step has finished on the GPU, else *out_ready = 0. */
int32_t ilm_system_poll_counts(IlmHandle system, uint32_t* out_counts, int32_t capacity, int32_t saturate16, int32_t* out_ready);

/* Ordered live-slot list of one chunk (wave64 ballot + prefix sum): writes the
 * ascending slot indices with life > 0 to `out_slots` (host, capacity entries)
 * and their number to out_count.  Consumer: particle lights (ParticleLight.fx). */
int32_t ilm_chunk_live_slots(IlmHandle system, int32_t chunk_index, uint32_t* out_slots, int32_t capacity, int32_t* out_count);

/* ---- lighting ------------------------------------------------------------- */

/* DistanceField atlas (Illuminant/SDF/DistanceField.cs:43-122): width x height
 * texels of 4 x 16 bit, row-major; `texels` is a host pointer (DistanceField.Load
 * layout, DistanceField.cs:178-213). */
int32_t ilm_sdf_create(IlmHandle ctx, int32_t atlas_width, int32_t atlas_height, int32_t format, IlmHandle* out_sdf);
int32_t ilm_sdf_upload(IlmHandle sdf, const uint16_t* texels);
/* sampleDistanceFieldEx (Illuminant/Shaders/DistanceFieldCommon.fxh:313-353) evaluated on the device at `count`
 * world positions (host array of xyz triples); writes `count` distances to out_distances (host) and synchronises.
 * Not on the reference's call path: it exposes the very sampler the particle collision and the cone trace use, so a
 * host (or a test) can query the field the kernels see. */
int32_t ilm_sdf_sample(IlmHandle sdf, const IlmDistanceFieldUniforms* df, const float* positions, int32_t count, float* out_distances);

/* Diagnostic, like ilm_sdf_sample: the cone trace samples positions that lie well inside the field through a second, table-driven form of
 * sampleDistanceFieldEx (csrc/hlsl_math.hpp, sample_inside_table) that reads the field's cell array (built from the atlas on demand and
 * rebuilt when the atlas changes).  This evaluates `count` positions the way the trace loop does:
 * out_used_table[i] = 1 where position i met that form's precondition and went through it, 0 where the general form was used.  A test
 * holds both bit-equal to the CPU restatement. */
int32_t ilm_debug_sdf_sample_inside(IlmHandle sdf, const IlmDistanceFieldUniforms* df, const float* positions, int32_t count,
                                    float* out_distances, int32_t* out_used_table);

/* Diagnostic, like ilm_sdf_sample: the cone trace's in-volume loop divides (distance + HACK_DISTANCE_OFFSET) by the cone radius
 * (ConeTrace.fxh:62) with an instruction sequence that skips the IEEE division's range scaling.  This evaluates that sequence
 * (`out_fast`) and the plain IEEE division (`out_ieee`) for `count` operand pairs on the device, so a test can hold them
 * bit-equal over the operand range the kernel admits (2^-60 <= |d| <= 2^60, |n| <= 2^60 or zero / infinite / NaN). */
int32_t ilm_debug_divide(IlmHandle ctx, const float* numerators, const float* denominators, int32_t count, float* out_fast, float* out_ieee);
/* The light pass divides by two constants -- DOT_RAMP_RANGE (LightCommon.fxh:6) and UNSHADOWED_THRESHOLD - FULLY_SHADOWED_THRESHOLD
 * (ConeTrace.fxh:19-20, :186) -- with the constants' reciprocals folded in at compile time.  This runs, for each of them, ALL 2^32
 * float bit patterns as the numerator through that sequence and through the IEEE division and counts the results that differ:
 * out_mismatches[2 i] for numerators inside the range the sequence is specified for (2^-60 <= |n| <= 2^60, zero, infinite, NaN -- the
 * kernel's numerators are saturates and sums of unit-vector components), out_mismatches[2 i + 1] outside it, for divisor
 * out_divisors[i]; *out_count divisors, out_mismatches holds 2 * capacity entries.  A test requires zero inside. */
int32_t ilm_debug_divide_by_constants(IlmHandle ctx, float* out_divisors, uint64_t* out_mismatches, int32_t capacity, int32_t* out_count);
/* ilm_system_step runs a step of the common shape (power-of-two chunk size >= 64, UpdatePositions, Gravity / area-less Noise and FMA,
 * inline spawners) through a kernel specialised for it and every other step through the kernel that interprets the descriptor; the
 * per-slot arithmetic is the same.  interpreter != 0 forces the interpreting kernel for every later step of this process (0: the
 * default choice) so that a test can hold the two bit-equal; returns the previous setting.  Environment: ILM_STEP_LEAN=0. */
int32_t ilm_debug_step_interpreter(int32_t interpreter);
/* A step over at least two chunks and half a million slots puts the second half of its chunk range on a second stream of the context
 * (chunks never interact, ParticleSystem.cs:743-745; every other entry point waits for both streams before it touches anything).
 * streams == 1 keeps every later step of this process on the context stream, 2 restores the default; returns the previous setting.
 * Environment: ILM_STEP_STREAMS=1.  A context whose stream was handed out by ilm_ctx_stream never splits. */
int32_t ilm_debug_step_streams(int32_t streams);
/* Diagnostic: the collision update (UpdateParticleSystemWithDistanceField.fx:29-147) is priced per sampleDistanceFieldEx call -- the
 * field lookup at the particle, one per step of its sweep, four for a bounce / redirect normal (VisualizeCommon.fxh:44-63).  enable != 0
 * makes every later ilm_system_step of this process that runs the collision update count those calls on the device; the call returns
 * the count accumulated since the previous call in *out_samples (may be NULL) and clears it.  Synchronises the context.  The count
 * costs one atomic per particle while it is on: for measurement runs only. */
int32_t ilm_debug_step_sdf_samples(IlmHandle ctx, int32_t enable, uint64_t* out_samples);
/* The grid of the context's last light-pass launch (sphere or particle lights): workgroups of 256 threads (four waves each; exit-only
 * workgroups of partial tile groups included -- what a profiler's wave count sees), the largest number of workgroups per tile (light
 * split; 1 = none) and the edge of the tile groups dealt to the XCDs (0: another block -> tile map).  Measurement scripts price per-wave
 * counters with it instead of re-deriving the launch (r05). */
int32_t ilm_debug_last_light_launch(IlmHandle ctx, int32_t* out_workgroups, int32_t* out_split, int32_t* out_tile_macro);
int32_t ilm_sdf_destroy(IlmHandle sdf);
/* DistanceField.Save (Illuminant/SDF/DistanceField.cs:178-194): the atlas bytes, 8 per texel, row-major. */
int32_t ilm_sdf_download(IlmHandle sdf, uint16_t* texels);
/* The atlas in device memory, for callers that write it themselves (a torch tensor view, another library's kernel).  The light passes
 * read the field through a cell array derived from the atlas (docs/experiments.md 2): once the pointer has been handed out the library cannot know
 * when the atlas changes and re-derives ALL cells before every light pass (138 MB on a 512 x 512 x 33 field) -- until the caller takes
 * over the bookkeeping with ilm_sdf_mark_dirty. */
int32_t ilm_sdf_device_ptr(IlmHandle sdf, void** out_ptr);
/* "I have written virtual slices [first_virtual_slice, first_virtual_slice + slice_count) of the atlas" (slice_count 0: all of it).  The
 * next light pass re-derives the cells of those slices and their lower neighbour only -- the reference regenerates
 * MaximumFieldUpdatesPerFrame = 1 slice triplet per frame (Illuminant/Lighting/LightingRenderer.Configuration.cs:91,
 * LightingRenderer.DistanceField.cs:415-464) -- and from this call on the field is no longer re-derived before every pass:
 * the caller reports its writes.  ilm_sdf_upload and ilm_sdf_render_slices do this bookkeeping themselves. */
int32_t ilm_sdf_mark_dirty(IlmHandle sdf, int32_t first_virtual_slice, int32_t slice_count);
/* What the light passes do with this field: the bytes of its cell array (0: none was built), how many virtual slices the last light pass
 * re-derived and has a table for (TableSlices 0: the pass used the general sampler -- uniforms that do not tile the atlas, more than 256
 * slices, cells past 2 GiB, or no memory for them: same results, about half the speed), and the running totals. */
typedef struct IlmSdfTraceInfo {
    uint64_t CellBytes;
    uint64_t CellRebuilds;          /* light passes that re-derived at least one slice */
    uint64_t CellSlicesRebuilt;     /* virtual slices re-derived in total */
    int32_t  LastRebuiltSlices;
    int32_t  TableSlices;
    int32_t  RebuiltEveryFrame;     /* 1: the device pointer was handed out and ilm_sdf_mark_dirty has not been called since */
    int32_t  Reserved;
} IlmSdfTraceInfo;
int32_t ilm_sdf_trace_info(IlmHandle sdf, IlmSdfTraceInfo* out);

/* ---- distance field generation ------------------------------ */

/* LightObstruction (Illuminant/Lighting/LightObstruction.cs:10-140): the DistanceFunctionVertex it packs
 * (Center, Size, Orientation quaternion -- Illuminant/Vertices.cs:105-141) + Type + IsDynamic. */
enum {
    ILM_OBSTRUCTION_ELLIPSOID = 0, ILM_OBSTRUCTION_BOX = 1, ILM_OBSTRUCTION_CYLINDER = 2,
    ILM_OBSTRUCTION_SPHEROID = 3, ILM_OBSTRUCTION_OCTAGON = 4
};
typedef struct IlmObstruction {
    float   Center[3];      int32_t Type;        /* LightObstructionType */
    float   Size[3];        int32_t IsDynamic;
    float   Orientation[4];                      /* quaternion x, y, z, w */
} IlmObstruction;

/* HeightVolumeBase (Illuminant/SDF/HeightVolume.cs:14-80) as RenderDistanceFieldHeightVolumes consumes it
 * (Illuminant/Lighting/LightingRenderer.DistanceField.cs:185-266): a polygon (a vertex range in the shared
 * xy array handed to the call) extruded over [ZBase, ZBase + Height]. */
typedef struct IlmHeightVolume {
    int32_t FirstVertex, VertexCount;
    float   ZBase, Height;
    int32_t IsDynamic;
    int32_t TopFaceEnableShadows;   /* HeightVolumeBase.TopFaceEnableShadows (:18), read by the G-buffer pass only */
    int32_t _pad[2];
} IlmHeightVolume;

#define ILM_DISTANCE_LIMIT 520.0f   /* LightingRenderer.DistanceLimit, Illuminant/Lighting/LightingRenderer.cs:316 */

/* What RenderDistanceFieldSliceTriplet binds for a group of slice triplets
 * (Illuminant/Lighting/LightingRenderer.DistanceField.cs:80-152): the atlas layout of the DistanceField
 * constructor (Illuminant/SDF/DistanceField.cs:43-122) and SliceIndexToZ's inputs (:32-35). */
typedef struct IlmDistanceFieldRenderDesc {
    int32_t VirtualWidth, VirtualHeight;
    float   VirtualDepth, ZOffset;
    int32_t SliceWidth, SliceHeight, SliceCount, ColumnCount;
    int32_t RowCount;
    float   MaximumEncodedDistance;
    float   InvScaleFactorX, InvScaleFactorY;    /* VirtualWidth / SliceWidth, VirtualHeight / SliceHeight (Uniforms.cs:108-109) */
    int32_t DynamicFlagFilter;                   /* -1: every obstruction (plain DistanceField); 0: static only; 1: dynamic only */
    int32_t _pad[3];
} IlmDistanceFieldRenderDesc;

/* RenderDistanceFieldPartition's inner loop (LightingRenderer.DistanceField.cs:415-464) for `triplet_count` slice
 * triplets at once: for each first virtual slice s = first_virtual_slices[i] (a multiple of 3) the physical slice
 * s / 3 is cleared -- to zero, or to the texels of `clear_source` (the DynamicDistanceField's static texture,
 * ClearDistanceField.fx:30-44) -- and every obstruction / height volume passing the dynamic-flag filter is
 * rasterised into it with the analytic distance functions of DistanceFunction.fx:33-115 (quad of half-size
 * max|Size| + MaximumEncodedDistance + 4, :24-25) / DistanceField.fx:75-115, encoded (DistanceFieldCommon.fxh:264-266)
 * and MAX-blended (LoadMaterials.cs:164-176); texel RGBA = virtual slices s .. s+3.
 * `obstructions`, `volumes`, `polygon_xy` (x,y pairs) and `first_virtual_slices` are host arrays.  Asynchronous. */
int32_t ilm_sdf_render_slices(IlmHandle sdf, IlmHandle clear_source, const IlmDistanceFieldRenderDesc* desc,
                              const int32_t* first_virtual_slices, int32_t triplet_count,
                              const IlmObstruction* obstructions, int32_t obstruction_count,
                              const IlmHeightVolume* volumes, int32_t volume_count,
                              const float* polygon_xy, int32_t polygon_vertex_count);

/* G-buffer (Illuminant/GBuffer.cs): width x height texels (encNormal.xy, relativeY, encodedZ). */
int32_t ilm_gbuffer_create(IlmHandle ctx, int32_t width, int32_t height, int32_t format, IlmHandle* out_gbuffer);
int32_t ilm_gbuffer_upload(IlmHandle gbuffer, const void* texels);
int32_t ilm_gbuffer_destroy(IlmHandle gbuffer);
int32_t ilm_gbuffer_download(IlmHandle gbuffer, void* texels);

/* What RenderGBuffer binds in its non-2.5D form (Illuminant/Lighting/LightingRenderer.GBuffer.cs:127-203): the view transform
 * (Position = the pending field viewport position, Scale = viewportScale * RenderScale, :131-135) and the ground plane's inputs
 * (RenderGroundPlane, :271-299). */
typedef struct IlmGBufferRenderDesc {
    float   ViewportPosition[2];
    float   ViewportScale[2];
    float   GroundZ;
    int32_t RenderGroundPlane;        /* Configuration.RenderGroundPlane: false lifts the plane by 99999 (:280-286) */
    int32_t EnableGroundShadows;      /* Environment.EnableGroundShadows */
    int32_t _pad;
} IlmGBufferRenderDesc;

/* RenderGBuffer without TwoPointFiveD, billboards or user content: clear to transparent, the ground plane
 * (technique GroundPlane, Illuminant/Shaders/GBuffer.fx:7-19,57-70) and the top face of every height volume drawn with the same
 * material from the lowest to the highest (RenderGBufferVolumes, :205-219), encoded by encodeGBufferSample
 * (Illuminant/Shaders/GBufferShaderCommon.fxh:10-35).  The top-face meshes come from Fracture's Geometry.Triangulate
 * (SDF/HeightVolume.cs:126-133), which is outside the tree; a triangulation covers exactly the polygon's interior, so coverage
 * is decided per pixel centre with an even-odd point-in-polygon test instead.  Host arrays; asynchronous. */
int32_t ilm_gbuffer_render(IlmHandle gbuffer, const IlmGBufferRenderDesc* desc,
                           const IlmHeightVolume* volumes, int32_t volume_count,
                           const float* polygon_xy, int32_t polygon_vertex_count);

/* ---- RenderGBuffer with the host's own meshes: 2.5D (top + front faces under a depth test) and billboards ------------------
 * The reference's host builds triangle lists -- HeightVolume.Mesh3D / GetFrontFaceMesh3D (Illuminant/SDF/HeightVolume.cs:106-224)
 * and the billboard quads of RenderGBufferBillboards (Illuminant/Lighting/LightingRenderer.GBuffer.cs:330-478) -- and hands them
 * to the GPU; those builders stay host code, the arrays cross the boundary as they are. */

/* HeightVolumeVertex, Illuminant/Vertices.cs:41-46 (Sequential, Pack = 4: 36 bytes) */
typedef struct IlmHeightVolumeVertex {
    float Position[3];
    float Normal[3];
    float ZRange[2];
    float EnableShadows;
} IlmHeightVolumeVertex;

/* BillboardVertex, Illuminant/Vertices.cs:75-81 (Sequential, Pack = 4: 48 bytes) */
typedef struct IlmBillboardVertex {
    float ScreenPosition[2];
    float TexCoord[2];
    float WorldPosition[3];
    float Normal[3];
    float DataScaleAndDynamicFlag[2];
} IlmBillboardVertex;

enum {
    ILM_BILLBOARD_MASK         = 0,  /* BillboardType.Mask        -> technique MaskBillboard  (GBufferBitmap.fx:29-59,115-122) */
    ILM_BILLBOARD_GBUFFER_DATA = 1   /* BillboardType.GBufferData -> technique GDataBillboard (GBufferBitmap.fx:61-113,124-131) */
};

/* One PrimitiveDrawCall of RenderGBufferBillboards (LightingRenderer.GBuffer.cs:392-409): a run of quads sharing a texture and a
 * type.  Texture = a lightmap-class texture of the same context (ilm_lightmap_create + ilm_lightmap_upload; RGBA8 = SurfaceFormat.Color,
 * float4 / half4 accepted), sampled POINT / CLAMP (_SetTextureForGBufferBillboard, :301-307); 0 = no texture bound, which Direct3D 9
 * samples as (0, 0, 0, 1) -- "the mask is an opaque rectangle" (Billboard.cs:93). */
typedef struct IlmBillboardRun {
    IlmHandle Texture;
    int32_t   FirstQuad, QuadCount;   /* quads [FirstQuad, FirstQuad + QuadCount) of the vertex array, 4 vertices each (TL, TR, BR, BL) */
    int32_t   Type;                   /* ILM_BILLBOARD_* */
    int32_t   _pad;
} IlmBillboardRun;

/* What RenderGBuffer and _SetupGBufferGroundPlane bind (LightingRenderer.GBuffer.cs:102-157): the view transform, the Environment
 * uniforms the three techniques read (Uniforms.cs:15-24: GroundZ, ZToYMultiplier, RenderScale), DistanceFieldExtent.z and the two
 * self-occlusion hacks (ComputeSelfOcclusionHack / ComputeZSelfOcclusionHack, :62-80 -- host arithmetic, passed in). */
typedef struct IlmGBufferMeshDesc {
    float   ViewportPosition[2];
    float   ViewportScale[2];
    float   GroundZ;
    float   ZToYMultiplier;
    float   RenderScale[2];
    float   DistanceFieldExtentZ;
    float   SelfOcclusionHack;
    float   ZSelfOcclusionHack;
    int32_t TwoPointFiveD;            /* Configuration.TwoPointFiveD */
    int32_t RenderGroundPlane;
    int32_t EnableGroundShadows;
    int32_t _pad[2];
} IlmGBufferMeshDesc;

/* RenderGBuffer (LightingRenderer.GBuffer.cs:127-203) in draw order -- clear (colour 0, depth 0); the ground plane; then
 *   TwoPointFiveD = 1: every triangle of `top` with technique HeightVolume, then every triangle of `front` with technique
 *     HeightVolumeFace (GBuffer.fx:21-55,72-103), both under DepthBufferFunction GreaterEqual with depth writes on a 24-bit depth
 *     buffer, depth = z / DistanceFieldExtent.z (LightingRenderer.cs:539-551, GBuffer.cs:30-38; RenderTwoPointFiveDVolumes, :221-269 --
 *     the caller concatenates the volumes' meshes in its OrderByDescending(ZBase + Height) order);
 *   TwoPointFiveD = 0: every triangle of `top` with the ground plane's technique in the caller's OrderBy(ZBase + Height) order
 *     (RenderGBufferVolumes, :205-219); `front` must be empty;
 * then the billboard runs: all Mask runs in array order, then all GBufferData runs (layerIndex, layerIndex + 1, :371-392), no
 * depth test.  Triangles cover a pixel when its centre is inside under the top-left rule on positions snapped to 1/256 pixel
 * (both windings: CullMode.None); vertex attributes are interpolated with the barycentric weights of the snapped triangle;
 * fragments whose depth leaves [0, 1] are clipped.  Host arrays; asynchronous. */
int32_t ilm_gbuffer_render_meshes(IlmHandle gbuffer, const IlmGBufferMeshDesc* desc,
                                  const IlmHeightVolumeVertex* top_vertices, int32_t top_vertex_count,
                                  const IlmHeightVolumeVertex* front_vertices, int32_t front_vertex_count,
                                  const IlmBillboardVertex* billboard_vertices, int32_t billboard_vertex_count,
                                  const IlmBillboardRun* runs, int32_t run_count);

/* Lightmap render target (BufferRing of lightmaps, LightingRenderer.cs:472-485).
 * If external_device_ptr != NULL the lightmap aliases caller-owned device memory
 * of width*height*bytes_per_texel bytes (e.g. a torch tensor that RCCL gathers). */
int32_t ilm_lightmap_create(IlmHandle ctx, int32_t width, int32_t height, int32_t format,
                            void* external_device_ptr, IlmHandle* out_lightmap);
int32_t ilm_lightmap_download(IlmHandle lightmap, void* dst, int32_t first_row, int32_t row_count);
/* Texture2D.SetData on rows [first_row, first_row + row_count) in the lightmap's own texel format (the albedo texture of the with-albedo
 * resolve is such a texture). */
int32_t ilm_lightmap_upload(IlmHandle lightmap, const void* src, int32_t first_row, int32_t row_count);
int32_t ilm_lightmap_device_ptr(IlmHandle lightmap, void** out_ptr);
int32_t ilm_lightmap_destroy(IlmHandle lightmap);

typedef struct IlmRenderStats {
    uint64_t SdfSamples;       /* sampleDistanceFieldEx calls executed (AO + cone-trace steps) */
    uint64_t PixelLightPairs;  /* pixel x light pairs inside a light's raster footprint */
    uint64_t TracedPairs;      /* pairs that ran the cone trace */
} IlmRenderStats;

/* The sphere-light pass of LightingRenderer.RenderLighting
 * (Illuminant/Lighting/LightingRenderer.cs:1004-1169 + technique SphereLight,
 * Illuminant/Shaders/SphereLight.fx:7-46): lightmap[row_begin..row_end) =
 * ambient + sum over lights, in light order.  gbuffer == 0 => ground plane
 * (LightCommon.fxh:130-141); sdf == 0 => no distance field.  `lights` is a host
 * array.  stats may be NULL; when non-NULL the instrumented (counting) kernel
 * variant runs and the call synchronises.
 * ambient == NULL: the lights are ADDED to what the lightmap holds -- a further light group of the same frame (the reference draws
 * one batch group per LightTypeRenderStateKey, e.g. per ramp texture, onto the same target, LightingRenderer.cs:1004-1169).
 * With a ramp texture bound (ilm_ctx_set_light_ramp) the technique is SphereLightWithDistanceRamp (SphereLight.fx:48-86,
 * SphereLightPixelEpilogueWithRamp, SphereLightCore.fxh:99-119). */
int32_t ilm_render_sphere_lights(IlmHandle ctx,
                                 const IlmLightVertex* lights, int32_t light_count,
                                 const IlmEnvironment* env,
                                 const IlmDistanceFieldUniforms* df,
                                 IlmHandle gbuffer, IlmHandle sdf,
                                 const float ambient[4],
                                 IlmHandle lightmap, int32_t row_begin, int32_t row_end,
                                 IlmRenderStats* stats);

/* LightSource.TextureRef / Configuration.DefaultRampTexture of the light group rendered by the FOLLOWING ilm_render_sphere_lights and
 * ilm_render_light_probes calls (bound per group by _LightBatchSetup, Illuminant/Lighting/LightingRenderer.cs:764-766; sampler
 * RampTextureSampler, Illuminant/Shaders/RampCommon.fxh:4-21: tex2Dlod level 0, LINEAR, U CLAMP, V WRAP): width * height float4 texels.
 * width == 0 -- or a 1 x 1 texture, which the reference treats as none (:822-827) -- selects the techniques without a ramp. */
int32_t ilm_ctx_set_light_ramp(IlmHandle ctx, const IlmFloat4* texels, int32_t width, int32_t height);

/* How the light passes of this context sum the lights of a pixel.
 *   ILM_BLEND_FP32_ACCUMULATE (default): fp32 registers over all lights in light order, one rounding when the texel is stored.
 *   ILM_BLEND_FP16_PER_LIGHT: the reference's render target.  Its lightmap is a HalfVector4 surface (LightingRenderer.cs:476-479) that
 *     the ROP blends into additively, one light quad after the other: the clear colour and every partial sum are fp16 values.  The
 *     model: dst = half(float(dst) + float(half(src))) per light and channel, round to nearest even (Direct3D converts the shader's
 *     output to the target format, then blends).  Same light order, same shader arithmetic; only the rounding of the sum differs.
 * 1e-4 parity with the HLSL path is defined against the fp32 form; this mode exists to MEASURE how far the frame the
 * reference's hardware path displays lies from it (docs/experiments.md 3.2; tests/test_lighting_gpu.py holds it bit-equal to the oracle's model). */
#define ILM_BLEND_FP32_ACCUMULATE 0
#define ILM_BLEND_FP16_PER_LIGHT 1
int32_t ilm_ctx_set_lightmap_blend(IlmHandle ctx, int32_t mode);

/* How many workgroups serve one 16 x 16 tile of this context's sphere-light launches ("light split").  The reference cuts a frame's
 * light list into draws of 128 instances and the ROP adds them (Illuminant/Lighting/LightingRenderer.cs:1149-1166,
 * Illuminant/Shaders/SphereLight.fx:42-45): the sum does not care who shaded which light.  Here a tile's list is summed in 8 fixed
 * parts -- each part in light order, the parts onto the clear colour in part order -- and `workgroups` = 1, 2, 4 or 8 says how many
 * workgroups share those parts; the lightmap's bits do not depend on it.  0 (default) = chosen per launch: 1 for launches that fill the
 * device several times over (whole frames); short ones (one rank's strip of a frame split over 8 GPUs, a small target), which would
 * otherwise last two wave lifetimes whatever their share of the work, are TAPERED -- the tiles that start first are served whole, later
 * ones by 2, then 4, the last by 8 workgroups, so that the launch drains in its smallest pieces (docs/experiments.md 3.2).  A non-zero value
 * serves every tile by that many.  Launches in the ILM_BLEND_FP16_PER_LIGHT model, of fewer than 16 or more than 1 024 lights, or of
 * particle lights always use 1. */
int32_t ilm_ctx_set_light_split(IlmHandle ctx, int32_t workgroups);

/* ---- particle lights and light probes --------------------------------------------------------- */

/* What _ParticleLightBatchSetup binds for a ParticleLightSource (Illuminant/Lighting/LightingRenderer.cs:769-790,
 * Illuminant/Lighting/LightSource.cs:466-505) + the StippleFactor ParticleSystem.Render adds
 * (Illuminant/Particles/ParticleSystem.cs:1023). */
typedef struct IlmParticleLightParams {
    IlmFloat4 LightProperties;      /* Template.Radius, RampLength, RampMode, CastsShadows && DistanceField */
    IlmFloat4 MoreLightProperties;  /* aoRadius (0 when aoOpacity <= .001), shadowDistanceFalloff | -99999, falloffYFactor, saturate(aoOpacity) */
    IlmFloat4 LightColor;           /* Template.Color */
    IlmFloat4 LightSpecularColor;   /* Template.SpecularColor, SpecularPower */
    float     StippleFactor;        /* must be >= 1: StippleReject is Fracture code outside the tree (DitherCommon.fxh) */
    float     _pad[3];
} IlmParticleLightParams;

/* technique ParticleLight (Illuminant/Shaders/ParticleLight.fx:16-118): one sphere light per live particle of `system`
 * (position from PositionAndLife, colour = un-premultiplied Chunk.RenderColor x LightColor), additively blended onto what the
 * lightmap already holds -- RenderLighting draws it as one more light-type render state after the clear
 * (LightingRenderer.cs:1126-1141).  quad_counts[i] = min(ChunkMaximumCount, chunk.TotalSpawned + 1) slots of chunk i take part
 * (RenderChunk, ParticleSystem.cs:880); NULL => every slot.  The live particles are compacted on the device in chunk / slot
 * order (wave64 ballot + prefix sum) into the same light records the sphere-light pass uses; nothing is read back. */
int32_t ilm_render_particle_lights(IlmHandle ctx, IlmHandle system, const int32_t* quad_counts, int32_t chunk_count,
                                   const IlmParticleLightParams* params,
                                   const IlmEnvironment* env, const IlmDistanceFieldUniforms* df,
                                   IlmHandle gbuffer, IlmHandle sdf,
                                   IlmHandle lightmap, int32_t row_begin, int32_t row_end,
                                   IlmRenderStats* stats);

/* technique SphereLightProbe (Illuminant/Shaders/SphereLightProbe.fx:19-44) for `probe_count` probes
 * (UpdateLightProbes, Illuminant/Lighting/LightingRenderer.LightProbes.cs:49-110): probe_positions[i] = (position, 1),
 * probe_normals[i] = (normal or 0, enableShadows); out_values[i] = sum over lights of color.rgb * color.a * opacity, alpha =
 * number of contributing lights.  Host arrays; synchronises (the reference reads the values back, :112-150). */
int32_t ilm_render_light_probes(IlmHandle ctx, const IlmLightVertex* lights, int32_t light_count,
                                const IlmFloat4* probe_positions, const IlmFloat4* probe_normals, int32_t probe_count,
                                const IlmEnvironment* env, const IlmDistanceFieldUniforms* df, IlmHandle sdf,
                                IlmFloat4* out_values);

/* ---- output side: particle read-back and lightmap resolve -------------------------------------- */

/* The members of Fracture's BitmapDrawCall that FillReadbackResult writes per live particle
 * (Illuminant/Particles/ParticleReadback.cs:73-167).  BitmapDrawCall itself is outside the reference tree. */
typedef struct IlmReadbackDrawCall {
    float   Position[2];        /* pAndL.X, pAndL.Y */
    float   Scale[2];           /* pSize * renderData.x */
    float   TextureRegion[4];   /* TopLeft.xy, BottomRight.xy (animation frame applied) */
    float   Rotation;           /* RotationFromVelocity ? renderData.y % 2 pi : 0 */
    float   SortOrder;          /* SortedReadback ? pAndL.Y + ZToY : 0 */
    uint8_t MultiplyColor[4];   /* (byte)(renderColor * 255), R G B A */
    int32_t _pad;
} IlmReadbackDrawCall;

/* What FillReadbackResult reads from ParticleSystemConfiguration / ParticleAppearance (:80-115): pSize and the texture region are
 * resolved by the caller (they depend on the texture's size, which lives on the C# side). */
typedef struct IlmReadbackParams {
    float   Size[2];            /* pSize */
    float   TextureRegion[4];   /* region: TopLeft.xy, BottomRight.xy; Bounds.Unit = (0,0,1,1) without a texture */
    float   AnimationRate[2];
    float   ZToY;
    int32_t ColumnFromVelocity, RowFromVelocity, RotationFromVelocity, SortedReadback;
    int32_t _pad;
} IlmReadbackParams;

/* MaybePerformReadback + FillReadbackResult (ParticleReadback.cs:21-167): instead of reading three whole float4 planes per chunk
 * back and filtering on the CPU, the live particles are compacted on the device (chunk / slot order, wave64 ballot + prefix sum)
 * straight into draw-call records and only those cross PCIe.  element_counts[i] = ceil(TotalSpawned / ChunkSize) * ChunkSize slots
 * of chunk i are examined (:57-58); NULL => every slot.  Writes at most `capacity` records and the total to *out_count;
 * synchronises. */
int32_t ilm_system_readback(IlmHandle system, const int32_t* element_counts, int32_t chunk_count, const IlmReadbackParams* params,
                            IlmReadbackDrawCall* out, int32_t capacity, int32_t* out_count);
/* The same read-back without the last copy: the records stay in a page-locked host buffer owned by the system's context (the
 * reference keeps a pooled ReadbackResultBuffer too, ParticleReadback.cs:40-41) and *out_records points at it; valid until the next
 * read-back on that context or its destruction.  Capacity is every examined slot.  NULL when nothing is alive.  Synchronises. */
int32_t ilm_system_readback_view(IlmHandle system, const int32_t* element_counts, int32_t chunk_count, const IlmReadbackParams* params,
                                 const IlmReadbackDrawCall** out_records, int32_t* out_count);

enum { ILM_HDR_NONE = 0, ILM_HDR_GAMMA_COMPRESS = 1, ILM_HDR_TONE_MAP = 2 };   /* HDRMode, LightingRenderer.HDR.cs:254-258 */

/* ---- particle rasterisation (techniques RasterizeParticlesNoTexture / TexturePoint / TextureLinear) -------------- */

enum {
    ILM_BLEND_ALPHA    = 0,  /* BlendState.AlphaBlend on premultiplied colour: dst = src + dst * (1 - src.a) */
    ILM_BLEND_ADDITIVE = 1   /* BlendState.Additive-like (One, One): dst = src + dst */
};
enum {
    ILM_BITMAP_NONE   = 0,   /* technique RasterizeParticlesNoTexture */
    ILM_BITMAP_POINT  = 1,   /* technique RasterizeParticlesTexturePoint  (BitmapPointSampler: POINT, CLAMP) */
    ILM_BITMAP_LINEAR = 2    /* technique RasterizeParticlesTextureLinear (BitmapSampler: LINEAR, CLAMP) */
};

/* What ParticleSystem.Render binds for the rasterise techniques: Uniforms.RasterizeParticleSystem (Illuminant/Uniforms.cs:238-290),
 * RoundingPowerFromLife, RenderingOptions, StippleFactor (Illuminant/Particles/ParticleSystem.cs:943-1041, :254-271) and the bits of
 * Uniforms.ParticleSystem the vertex shader reads (TexelAndSize.zw, ZToY).  The view transform -- Fracture's ViewTransformCommon.fxh,
 * outside the tree -- is reduced to the default one of a render target: pixel = (display - ViewportPosition) * ViewportScale,
 * pixel centres at +0.5; the depth formula is carried but unused (no depth buffer). */
typedef struct IlmRasterizeParams {
    IlmFloat4 GlobalColor;             /* Color.Global, premultiplied (Uniforms.cs:279-283) */
    IlmFloat4 BitmapTextureRegion;     /* (offset, offset + size) / texture size; (0, 0, 1, 1) without a texture */
    IlmFloat4 SizeFactorAndPosition;   /* (SizePx / 2 when RelativeSize and a texture is bound, else (1, 1); origin.xy) */
    IlmFloat4 Scale;                   /* (scale.xy, 0, 0) */
    IlmFloat4 ZFormula;
    IlmFloat4 ZConfiguration;          /* (SizeFromZ, 0, 0, 0) */
    IlmClampedBezier1 RoundingPowerFromLife;
    float     RenderingOptions[4];     /* Rounded, DitheredOpacity (premultipliedToDithered, RasterizeParticleSystem.fx:158-175, over Dither64 -- Fracture code outside the
                                      * tree, restated from the function it publishes: Jimenez, SIGGRAPH 2014), column / row (of the frame sheet) from velocity */
    float     SystemSize[2];           /* System.TexelAndSize.zw = Configuration.Size */
    float     ZToY;
    float     StippleFactor;           /* must be >= 1 (StippleReject is Fracture code) */
    float     ViewportScale[2], ViewportPosition[2];
    int32_t   BlendMode;               /* ILM_BLEND_* */
    int32_t   BitmapFilter;            /* ILM_BITMAP_*: which technique; the bitmap is the one bound with ilm_system_set_bitmap */
    float     AnimationRate[2];        /* System.AnimationRateAndRotationAndZToY.xy = 1 / Appearance.AnimationRate (0 when that is 0) */
} IlmRasterizeParams;

/* ParticleSystem.Render with technique RasterizeParticlesNoTexture (Illuminant/Shaders/RasterizeParticleSystem.fx:61-260,
 * Illuminant/Particles/ParticleSystem.cs:876-1041): every live particle of the first chunk_count chunks becomes a rotated quad
 * (VS_PosVelAttr) shaded by PS_NoTexture (colour x GlobalColor x computeCircularAlpha, discard at alpha <= 0 -- the shader's
 * `1 / 512` is an integer division) and blended onto `target` (a lightmap object used as the render target) IN CHUNK / SLOT ORDER,
 * as the instanced draws of RenderChunk do.  quad_counts[i] = min(ChunkMaximumCount, TotalSpawned + 1) slots of chunk i take part
 * (NULL => every slot).  A pixel belongs to a quad when its centre maps to unit coordinates in [-1, 1) x [-1, 1) (the top-left rule
 * of an axis-aligned quad, carried along with the rotation).  Sorted on the device by (16 x 16 tile, slot); nothing is read back.
 * out_stats (may be NULL): [0] live quads, [1] (quad, tile) pairs, [2] shaded pixels (fragments not discarded). */
int32_t ilm_render_particles(IlmHandle system, const int32_t* quad_counts, int32_t chunk_count, const IlmRasterizeParams* params,
                             IlmHandle target, uint64_t* out_stats);
/* Appearance.Texture for the textured techniques (PS_Texture / PS_TexturePoint, RasterizeParticleSystem.fx:191-226; frame selection
 * of VS_PosVelAttr, :112-139): width * height float4 texels, row-major, ONE level -- the reference samples with a mip filter whose
 * level comes from screen-space derivatives (hardware-defined) over a chain its texture loader builds; a bitmap without mips is
 * the case that is defined by the shader text alone.  The fragment is colour x texel x GlobalColor x computeCircularAlpha.
 * width == 0 releases the bitmap. */
int32_t ilm_system_set_bitmap(IlmHandle system, const IlmFloat4* texels, int32_t width, int32_t height);
/* Fill a lightmap / render target with one colour (the Clear before ParticleSystem.Render). */
int32_t ilm_lightmap_clear(IlmHandle lightmap, const float rgba[4]);

/* HDRConfiguration (Illuminant/Lighting/LightingRenderer.HDR.cs:198-252) as LightingResolveHandler._Before binds it
 * (Illuminant/Lighting/LightingRenderer.cs:1463-1520; clamps of IlluminantMaterials.cs:81-137 are applied by the library). */
typedef struct IlmHDRConfiguration {
    int32_t Mode;
    float   InverseScaleFactor;     /* 0 => 1 */
    float   Offset, Exposure, Gamma;
    float   MiddleGray, AverageLuminance, MaximumLuminance;   /* GammaCompression */
    float   WhitePoint;                                       /* ToneMapping */
    int32_t ResolveToSRGB;          /* must be 0: pLinearToPSRGB is Fracture code (sRGBCommon.fxh, outside the tree) */
    int32_t DitheringStrength;      /* must be 0: ApplyDither is Fracture code (DitherCommon.fxh) */
    int32_t AlbedoIsSRGB;           /* with-albedo resolve only; must be 0: pSRGBToPLinear is Fracture code (sRGBCommon.fxh) */
} IlmHDRConfiguration;

/* RenderedLighting.Resolve without albedo, 1:1 (techniques ScreenSpaceLightingResolve / GammaCompressedLightingResolve /
 * ToneMappedLightingResolve, Illuminant/Shaders/Resolve.fx:25-139 + HDR.fxh): dst[row_begin..row_end) = tone-mapped src, alpha 1.
 * src and dst are lightmap handles of the same width (any formats; dst RGBA8 is the back-buffer case).  Heights may differ -- a group
 * member's lightmap (ilm_group_lightmap_member) carries padding rows below the frame, a back buffer does not -- as long as the rows
 * resolved exist in every texture involved. */
int32_t ilm_resolve_lighting(IlmHandle src_lightmap, IlmHandle dst_lightmap, const IlmHDRConfiguration* hdr,
                             int32_t row_begin, int32_t row_end);
/* RenderedLighting.Resolve WITH albedo (LightingRenderer.ResolveLighting with `albedo != null`, Illuminant/Lighting/LightingRenderer.cs:1537-1580;
 * techniques ScreenSpaceLightingResolveWithAlbedo / GammaCompressed... / ToneMapped..., Illuminant/Shaders/Resolve.fx:43-60,141-233), 1:1:
 * light = src * (InverseScaleFactor * 2); rgb = lerp(albedo.rgb, albedo.rgb * light.rgb, saturate(light.a)); alpha = albedo.a; then the
 * HDR mode as above.  `albedo` is a texture the size of the lightmap held in a lightmap object (RGBA8 = SurfaceFormat.Color is the
 * usual case; upload with ilm_lightmap_upload); albedo == 0 is ilm_resolve_lighting.  Not bound: AlbedoIsSRGB, the LUT-blended technique. */
int32_t ilm_resolve_lighting_with_albedo(IlmHandle src_lightmap, IlmHandle albedo, IlmHandle dst_lightmap, const IlmHDRConfiguration* hdr,
                                         int32_t row_begin, int32_t row_end);

/* ---- multi-device groups (SURVEY 8e / 8b "ilm_ctx_create(device_ids, n)") --------------------------------------------------------
 *
 * The reference renders on one GraphicsDevice.  A group is the MI355X-native extension that keeps its call sites unchanged while
 * the work spreads over the GPUs of a node: the caller of LightingRenderer.RenderLighting (Illuminant/Lighting/LightingRenderer.cs:917-923)
 * still gets ONE composited lightmap per frame (:1004-1010), the caller of ParticleSystem.Update (Illuminant/Particles/ParticleSystem.cs:630-761)
 * still sees one liveness table.  Two shapes, same entry points afterwards:
 *   - in-process: one host process drives n devices (ilm_group_create) -- the shape of a C# host;
 *   - one process per GPU: every process creates its member with the same 128-byte id (ilm_group_unique_id on rank 0, handed to the
 *     others by the launcher) -- the shape of `torchrun` / MPI jobs.
 * Each member owns a context (ilm_group_ctx): replicated inputs -- the distance-field atlas, G-buffers, light ramps, randomness tables --
 * are ordinary per-context objects the host creates in a loop over the members (the atlas is 25 MB; generating it per device is
 * cheaper than broadcasting it).  Ranks number the members of the whole group: rank = first_rank + local index.
 *
 * Lighting: the frame is cut into `world` strips of whole 16-row tile bands, equal slots of R rows (world * R >= height); every member
 * renders its strip into its own full-frame buffer and the strips are all-gathered IN PLACE, so every device ends with the composited
 * frame.  Particles: chunks never interact (ParticleSystem.cs:743-745), chunk c lives on rank c % world as that member's chunk
 * c / world; the per-step data path has no collective, only the per-chunk live counts are gathered (ilm_group_live_counts). */

enum {
    ILM_GATHER_NONE = 0,  /* strips stay where they were rendered (a host that reads every strip back itself) */
    ILM_GATHER_PEER = 1,  /* in-process groups: every member pushes its strip to the n - 1 others with hipMemcpyPeerAsync -- one
                             transfer per xGMI link, all links of the full mesh busy at once */
    ILM_GATHER_RCCL = 2,  /* ncclAllGather on the members' context streams (RCCL over xGMI; the only exchange between processes) */
    ILM_GATHER_ASYNC = 0x100, /* flag, OR'ed onto ILM_GATHER_PEER / ILM_GATHER_RCCL (r05): the exchange runs on a second stream of every member,
                             behind the strip just queued, and the member's context stream goes on -- with the next frame's strip into ANOTHER
                             group lightmap (a ring of two, the reference's BufferRing).  ilm_group_lightmap_wait orders later work behind it. */
    ILM_GATHER_STORE = 3  /* (r05; in-process groups through peer access, groups that span processes through IPC-mapped buffers): no copy phase at all -- the light kernel's final store writes every texel of a member's
                             strip at the same offset of EVERY member's copy of the frame (the others' buffers peer-mapped over xGMI:
                             n stores of 8 B per pixel), so the exchange overlaps the strip; what remains of the gather is a fence */
};

/* In-process group over `n` devices (ids may repeat: several members on one device, which is how the exchange paths are tested on a
 * one-GPU box; RCCL itself refuses duplicate devices).  Creates one context per member. */
int32_t ilm_group_create(const int32_t* device_ids, int32_t n, IlmHandle* out_group);
/* One-process-per-GPU group: rank 0 obtains `id` (128 bytes) from ilm_group_unique_id and the launcher hands it to every rank; all
 * `world` processes then call ilm_group_create_rank collectively (it creates the RCCL communicator). */
int32_t ilm_group_unique_id(void* out_id128);
int32_t ilm_group_create_rank(int32_t device_id, int32_t rank, int32_t world, const void* id128, IlmHandle* out_group);
/* Destroys the member contexts too; ILM_ERR_STATE while group lightmaps or objects of the member contexts are alive. */
int32_t ilm_group_destroy(IlmHandle group);
/* out_local = members in this process, out_world = members of the whole group, out_first_rank = rank of local member 0,
 * out_comm_ranks = the rank count the RCCL communicator itself reports (ncclCommCount; 0 while no communicator exists). */
int32_t ilm_group_info(IlmHandle group, int32_t* out_local, int32_t* out_world, int32_t* out_first_rank, int32_t* out_comm_ranks);
int32_t ilm_group_ctx(IlmHandle group, int32_t local_index, IlmHandle* out_ctx);
int32_t ilm_group_sync(IlmHandle group);

/* In-place all-gather of device buffers: buffers[i] (on local member i's device) holds world * bytes_per_rank bytes and rank r's
 * slot starts at r * bytes_per_rank; on return (stream-ordered on each member's context stream) every buffer holds every slot.
 * This is the primitive under ilm_group_lightmap_gather; a host uses it directly for the optional Pos+Life all-gather of SURVEY 8e
 * (planes from ilm_chunk_device_ptr) when a global consumer exists. */
int32_t ilm_group_all_gather(IlmHandle group, void* const* buffers, uint64_t bytes_per_rank, int32_t gather);

/* Small HOST payloads across the group (timings, counters; <= 1 MiB per rank): local = n_local * bytes_per_rank bytes, one slot per
 * local member in member order; out_all receives world * bytes_per_rank bytes, rank r's slot at r * bytes_per_rank, identical on every
 * process.  Waits for the members' queued work first, so it is also the barrier a host brackets a timed region with. */
int32_t ilm_group_host_all_gather(IlmHandle group, const void* local, void* out_all, uint32_t bytes_per_rank);

/* The composited lightmap of a group: one buffer of world * R rows per local member (the frame is its first `height` rows). */
int32_t ilm_group_lightmap_create(IlmHandle group, int32_t width, int32_t height, int32_t format, IlmHandle* out_group_lightmap);
/* The plain lightmap object (aliasing the member's buffer) that per-context calls -- particle lights, resolve, download -- take. */
int32_t ilm_group_lightmap_member(IlmHandle group_lightmap, int32_t local_index, IlmHandle* out_lightmap);
/* Rows [*out_row_begin, *out_row_end) of the frame that rank `rank` renders, and the slot height R. */
int32_t ilm_group_lightmap_strip(IlmHandle group_lightmap, int32_t rank, int32_t* out_row_begin, int32_t* out_row_end, int32_t* out_slot_rows);
/* Replaces the equal slots by other strips: row_begins[r], row_ends[r] for every rank r of the group (world entries each), contiguous in
 * rank order, whole 16-row tile bands, covering [0, height) -- cost-balanced strips when the lights are unevenly spread (SURVEY 8e;
 * the reference has one device and no such notion).  Every process of the group must install the same table: with one process per GPU
 * the call is ALWAYS a COLLECTIVE (r05) -- every rank, also one that was handed a malformed table and one that resets to the equal slots
 * with NULL, NULL, enters the same 8-byte host all-gather with a hash of its argument, and all ranks succeed or fail together: they return
 * ILM_ERR_STATE (ILM_ERR_INVALID_ARGUMENT on the rank whose own table is malformed) and keep the table they had when the arguments differ
 * or any of them is malformed (each rank sizes its sends and receives from its own copy).  Unequal strips are
 * exchanged range by range at their true rows (ILM_GATHER_PEER: peer copies; ILM_GATHER_RCCL: one group of ncclSend / ncclRecv, one
 * transfer per xGMI link and direction) instead of by the single in-place all-gather.  NULL, NULL restores the equal slots. */
int32_t ilm_group_lightmap_set_strips(IlmHandle group_lightmap, const int32_t* row_begins, const int32_t* row_ends);
int32_t ilm_group_lightmap_gather(IlmHandle group_lightmap, int32_t gather);
/* After ilm_group_lightmap_gather(..., mode | ILM_GATHER_ASYNC): every member's context stream waits for that exchange (stream-ordered,
 * nothing blocks the host).  Call it before anything reads the composited frame and before the lightmap is rendered into again
 * (ilm_group_render_sphere_lights does it itself); a no-op when no asynchronous exchange is pending.  A host keeps two group lightmaps
 * and alternates: strip N + 1 is rendered while strip N travels.  ilm_group_sync / ilm_group_host_all_gather drain the exchange streams too. */
int32_t ilm_group_lightmap_wait(IlmHandle group_lightmap);
/* Arms (enable != 0) or disarms the store-mode exchange: while armed, EVERY light pass into a member's lightmap (ilm_render_sphere_lights,
 * ilm_render_particle_lights through ilm_group_lightmap_member's handle) also stores its texels into the other members' copies of the frame, and
 * ilm_group_lightmap_gather(ILM_GATHER_STORE) is the fence that orders each member's later readers behind the other members' passes --
 * call it after the strips of a frame, and again in front of the next frame's strips when readers of the old frame may still be queued
 * (a pass overwrites the other members' copies as it runs).  ilm_group_render_sphere_lights(..., ILM_GATHER_STORE) does all of this
 * itself for one call.  In-process groups need peer access between their devices (ILM_ERR_STATE otherwise).  For a group that spans
 * processes the call is a COLLECTIVE (arming and disarming alike; destroying an armed lightmap disarms it): every rank exports its buffer
 * as an IPC handle (hipIpcGetMemHandle), the handles are all-gathered, every rank maps the others' (hipIpcOpenMemHandle) -- and either
 * every rank arms or none does (ILM_ERR_STATE on all); the fence of ILM_GATHER_STORE is then an 8-byte collective on the context streams.
 * The mappings are made once per group lightmap, PROVEN before anybody arms (every rank writes a stamp through each of its mappings
 * and finds every other rank's stamp in its own buffer; the bytes underneath are restored) and kept until the GROUP is destroyed --
 * disarming, re-arming and destroying the lightmap unmap nothing (unmapping and re-exporting inside one process was measured to
 * resolve handles to the wrong buffer on ROCm 7.2: DESIGN.md section 5).
 * The table follows the BUFFER: a lightmap object the host made around a member's texels on the member's context (ilm_lightmap_create
 * with external_device_ptr = ilm_lightmap_device_ptr(member)) is mirrored exactly like the member handle itself; such an object on ANOTHER
 * context (a sibling's) is refused by the light passes with ILM_ERR_STATE while the mode is armed (its stream is outside the fence).
 * ONLY the light passes are mirrored: ilm_lightmap_clear, ilm_lightmap_upload, ilm_render_particles and ilm_resolve_lighting into an armed
 * member (or an alias of its texels) write THAT member's copy alone -- the members' frames stay equal only if every member makes the same
 * call (a host that clears or uploads per frame does it on every member, as it creates the replicated inputs).
 * Synchronises the members' streams.  The reference has one device and one lightmap
 * (Illuminant/Lighting/LightingRenderer.cs:1004-1010): every member still ends with that one composited frame. */
int32_t ilm_group_lightmap_store_mode(IlmHandle group_lightmap, int32_t enable);
int32_t ilm_group_lightmap_destroy(IlmHandle group_lightmap);

/* ilm_render_sphere_lights for the whole group: every local member renders its strip (same lights, environment and uniforms;
 * gbuffers / sdfs = one handle per local member, objects of that member's context, NULL / 0 as in the single-device call), then the
 * strips are gathered.  stats (may be NULL) = sums over the local members. */
int32_t ilm_group_render_sphere_lights(IlmHandle group, const IlmLightVertex* lights, int32_t light_count,
                                       const IlmEnvironment* env, const IlmDistanceFieldUniforms* df,
                                       const IlmHandle* gbuffers, const IlmHandle* sdfs, const float ambient[4],
                                       IlmHandle group_lightmap, int32_t gather, IlmRenderStats* stats);

/* Liveness table of a sharded particle system: systems[i] = local member i's system (chunk c of the table = chunk c / world of rank
 * c % world); out_counts[c] = live count of chunk c from each member's last counting step, identical on every process (integers,
 * bit-exact).  One small RCCL all-gather when the group spans processes, none otherwise.  Synchronises. */
int32_t ilm_group_live_counts(IlmHandle group, const IlmHandle* systems, int32_t total_chunks, uint32_t* out_counts, int32_t capacity,
                              int32_t saturate16);

/* The sharded particle state made whole ("optional all-gather of Pos+Life, 16 B/slot, only when a global consumer
 * exists -- particle lights, host readback").  sources[i] = local member i's system (chunk c of the table = chunk c / world of rank
 * c % world, as for ilm_group_live_counts); gathered[i] = an ordinary system on the SAME member's context whose engine has the same
 * chunk size and which holds total_chunks chunks (ilm_system_add_chunk).  After the call (stream-ordered on the members' context
 * streams, nothing blocks the host) components [first_component, first_component + component_count) of chunk c of every gathered
 * system are those of chunk c of the table: 0..3 = Pos+Life (ParticleSystem.cs:73-146 PositionAndLife), 12..15 = RenderColor (what
 * ParticleLight.fx:16-83 reads beside the position).  The planes of a chunk are contiguous, so a chunk travels as one range from the
 * owner's planes into the destination's planes, no packing: ILM_GATHER_PEER (in-process groups) = hipMemcpyPeerAsync to each other
 * member; ILM_GATHER_RCCL = one group of ncclSend / ncclRecv per call; ILM_GATHER_NONE = only each member's own chunks are copied over.
 * Every other component of the gathered system is left as it is.  Any consumer that takes a system handle then sees the whole table in
 * chunk order: ilm_render_particle_lights(ctx, gathered, ...) lights a member's strip with EVERY rank's particles, in the order -- and
 * therefore with the bits -- of the one-context frame.  A collective when the group spans processes: every rank calls it with the same
 * total_chunks, components and mode. */
int32_t ilm_group_gather_chunks(IlmHandle group, const IlmHandle* sources, const IlmHandle* gathered, int32_t total_chunks,
                                int32_t first_component, int32_t component_count, int32_t gather);


#ifdef __cplusplus
}
#endif
#endif /* ILLUMINANT_HIP_H */
