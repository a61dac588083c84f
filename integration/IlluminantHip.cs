// IlluminantHip.cs -- P/Invoke layer of libilluminant_hip.so for sq/Illuminant (drop into Illuminant/Native/).
// GENERATED from include/illuminant_hip.h by tools/gen_csharp_binding.py -- do not edit; the header carries the documentation
// and the reference file:line each entry point replaces.  ABI version 9.
//
// Vector4 / Matrix are XNA's; LightVertex is Illuminant/Vertices.cs:10-39; the Uniforms.* structs of the reference
// (Uniforms.cs:14-24,79-88,197-236; Bezier.cs:433-441,588-599) have the byte layout of the Ilm* mirrors below and can be passed
// with a pointer cast.  Every call returns 0 or an error code: IlluminantHip.Check turns it into the reference's exception types.
using System;
using System.Runtime.InteropServices;
using Microsoft.Xna.Framework;

namespace Squared.Illuminant.Native {
    public sealed class IlluminantHipException : Exception {
        public readonly int Code;
        public IlluminantHipException (int code, string message) : base(message) { Code = code; }
    }

    public static class IlmConstants {
        public const int ABI_VERSION = 9;
        public const int BLEND_FP16_PER_LIGHT = 1;
        public const int BLEND_FP32_ACCUMULATE = 0;
        public const int ERR_INVALID_ARGUMENT = -1;
        public const int ERR_INVALID_HANDLE = -2;
        public const int ERR_NO_DEVICE = -5;
        public const int ERR_OUT_OF_RANGE = -3;
        public const int ERR_STATE = -6;
        public const int ERR_TOO_MANY = -4;
        public const int MAX_ATTRACTORS = 16;
        public const int MAX_INLINE_POSITION_CONSTANTS = 4;
        public const int MAX_OPS = 4;
        public const int MAX_SPAWNS = 2;
        public const int OK = 0;
        public const int RANDOMNESS_HEIGHT = 653;
        public const int RANDOMNESS_WIDTH = 807;
        public const int STEP_COUNT_LIVE = 1;
        public const float DISTANCE_LIMIT = 520.0f;
        public const int OP_GRAVITY = 1;
        public const int OP_NOISE = 2;
        public const int OP_FMA = 3;
        public const int OP_MATRIX_MULTIPLY = 4;
        public const int OP_SPATIAL_NOISE = 5;
        public const int UPDATE_NONE = 0;
        public const int UPDATE_POSITIONS = 1;
        public const int UPDATE_WITH_DISTANCE_FIELD = 2;
        public const int UPDATE_ERASE = 3;
        public const int SPAWN_INLINE = 0;
        public const int SPAWN_POSITION_BUFFER = 1;
        public const int SPAWN_FEEDBACK = 2;
        public const int SPAWN_PATTERN = 3;
        public const int SDF_UNORM16 = 0;
        public const int SDF_FP16 = 1;
        public const int GBUFFER_FLOAT4 = 0;
        public const int GBUFFER_HALF4 = 1;
        public const int LIGHTMAP_FLOAT4 = 0;
        public const int LIGHTMAP_HALF4 = 1;
        public const int LIGHTMAP_RGBA8 = 2;
        public const int PLANE_POSITION = 0;
        public const int PLANE_VELOCITY = 1;
        public const int PLANE_ATTRIBUTES = 2;
        public const int PLANE_RENDER_COLOR = 3;
        public const int PLANE_RENDER_DATA = 4;
        public const int OBSTRUCTION_ELLIPSOID = 0;
        public const int OBSTRUCTION_BOX = 1;
        public const int OBSTRUCTION_CYLINDER = 2;
        public const int OBSTRUCTION_SPHEROID = 3;
        public const int OBSTRUCTION_OCTAGON = 4;
        public const int BILLBOARD_MASK = 0;
        public const int BILLBOARD_GBUFFER_DATA = 1;
        public const int HDR_NONE = 0;
        public const int HDR_GAMMA_COMPRESS = 1;
        public const int HDR_TONE_MAP = 2;
        public const int BLEND_ALPHA = 0;
        public const int BLEND_ADDITIVE = 1;
        public const int BITMAP_NONE = 0;
        public const int BITMAP_POINT = 1;
        public const int BITMAP_LINEAR = 2;
        public const int GATHER_NONE = 0;
        public const int GATHER_PEER = 1;
        public const int GATHER_RCCL = 2;
        public const int GATHER_ASYNC = 256;
        public const int GATHER_STORE = 3;
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 64)]
    public struct IlmParticleSystemUniforms {
        public Vector4 GlobalSettings;
        public Vector4 CollisionSettings;
        public Vector4 TexelAndSize;
        public Vector4 AnimationRateAndRotationAndZToY;
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 32)]
    public struct IlmClampedBezier1 {
        public Vector4 RangeAndCount;
        public Vector4 ABCD;
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 80)]
    public struct IlmClampedBezier4 {
        public Vector4 RangeAndCount;
        public Vector4 A;
        public Vector4 B;
        public Vector4 C;
        public Vector4 D;
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 96)]
    public struct IlmDistanceFieldUniforms {
        public Vector4 ConeAndMisc;
        public Vector4 TextureSliceAndTexelSize;
        public Vector4 StepAndMisc2;
        public Vector4 TextureSliceCount;
        public Vector4 Extent;
        public Vector4 Packed1;
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 64)]
    public unsafe struct IlmEnvironment {
        public Vector4 ZAndScale;
        public Vector4 ZToY;
        public Vector4 GBufferTexelSizeAndMisc;
        public fixed float ViewportPosition[2];
        public float GBufferViewportRelative;
        public float _pad0;
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 64)]
    public unsafe struct IlmAreaParams {
        public int AreaType;
        public float Strength;
        public float AreaFalloff;
        public float AreaRotation;
        public fixed float AreaCenter[3];
        public float _pad0;
        public fixed float AreaSize[3];
        public float _pad1;
        public fixed float CategoryFilter[2];
        public fixed float _pad2[2];
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 400)]
    public unsafe struct IlmGravityParams {
        public int AttractorCount;
        public float MaximumAcceleration;
        public fixed float CategoryFilter[2];
        public fixed float AttractorPositions[48];
        public fixed float AttractorRadiusesAndStrengths[48];
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 144)]
    public unsafe struct IlmFMAParams {
        public IlmAreaParams Area;
        public float TimeDivisor;
        public fixed float _pad[3];
        public Vector4 PositionAdd;
        public Vector4 PositionMultiply;
        public Vector4 VelocityAdd;
        public Vector4 VelocityMultiply;
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 192)]
    public unsafe struct IlmNoiseParams {
        public IlmAreaParams Area;
        public float TimeDivisor;
        public float FrequencyLerp;
        public float ReplaceOldVelocity;
        public float _pad;
        public fixed float RandomnessOffset[2];
        public fixed float NextRandomnessOffset[2];
        public Vector4 PositionOffset;
        public Vector4 PositionMinimum;
        public Vector4 PositionScale;
        public Vector4 VelocityOffset;
        public Vector4 VelocityMinimum;
        public Vector4 VelocityScale;
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 416)]
    public unsafe struct IlmSpawnParams {
        public fixed float ChunkSizeAndIndices[4];
        public fixed float Configuration[36];   // Vector4[9]
        public fixed float FormulaTypes[4];
        public Matrix PositionMatrix;
        public Matrix VelocityMatrix;
        public fixed float AxisMask[3];
        public float AlignVelocityAndPosition;
        public fixed float RandomnessOffset[2];
        public float AttributeDiscardThreshold;
        public float PolygonRate;
        public float PolygonLoop;
        public float PositionConstantCount;
        public fixed float _pad[2];
        public fixed float InlinePositionConstants[16];   // Vector4[4]
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 256)]
    public unsafe struct IlmUpdateParams {
        public IlmClampedBezier4 ColorFromLife;
        public IlmClampedBezier4 ColorFromVelocity;
        public IlmClampedBezier1 SizeFromLife;
        public IlmClampedBezier1 SizeFromVelocity;
        public fixed float RotationFromLifeAndIndex[2];
        public fixed float _pad[2];
        public Vector4 LifeRampSettings;
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 208)]
    public unsafe struct IlmMatrixMultiplyParams {
        public IlmAreaParams Area;
        public float TimeDivisor;
        public fixed float _pad[3];
        public Matrix PositionMatrix;
        public Matrix VelocityMatrix;
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 208)]
    public unsafe struct IlmSpatialNoiseParams {
        public IlmNoiseParams Noise;
        public fixed float SpaceScale[2];
        public fixed float _pad[2];
    }

    [StructLayout(LayoutKind.Explicit, Size = 416)]
    public unsafe struct IlmTransformOp {
        [FieldOffset(0)] public int Type;
        [FieldOffset(4)] public fixed int _pad[3];
        [FieldOffset(16)] public IlmGravityParams Gravity;
        [FieldOffset(16)] public IlmNoiseParams Noise;
        [FieldOffset(16)] public IlmFMAParams FMA;
        [FieldOffset(16)] public IlmMatrixMultiplyParams MatrixMultiply;
        [FieldOffset(16)] public IlmSpatialNoiseParams SpatialNoise;
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 48)]
    public unsafe struct IlmFeedbackParams {
        public ulong SourceSystem;
        public int SourceChunkIndex;
        public float FeedbackSourceIndex;
        public float InstanceMultiplier;
        public float SourceVelocityFactor;
        public float AlignPositionConstant;
        public float MultiplyLife;
        public float MultiplyAttributeConstant;
        public fixed float SourceLifeRange[2];
        public float _pad;
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 64)]
    public unsafe struct IlmPatternParams {
        public fixed float StepWidthAndSizeScale[4];
        public fixed float YOffsetsAndCoordScale[4];
        public fixed float TexelOffsetAndMipBias[4];
        public fixed float CenteringOffset[2];
        public float MultiplyAttributeConstant;
        public float _pad;
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 544)]
    public unsafe struct IlmSpawnRecord {
        public int ChunkIndex;
        public int Kind;
        public fixed int _pad[2];
        public IlmSpawnParams Params;
        public IlmFeedbackParams Feedback;
        public IlmPatternParams Pattern;
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 3200)]
    public unsafe struct IlmStepDesc {
        public int FirstChunk;
        public int ChunkCount;
        public int OpCount;
        public int SpawnCount;
        public int UpdateMode;
        public uint Flags;
        public fixed int _pad[2];
        public IlmParticleSystemUniforms System;
        public IlmUpdateParams Update;
        public IlmDistanceFieldUniforms DistanceField;
        public IlmTransformOp Ops0;
        public IlmTransformOp Ops1;
        public IlmTransformOp Ops2;
        public IlmTransformOp Ops3;
        public IlmSpawnRecord Spawns0;
        public IlmSpawnRecord Spawns1;
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 40)]
    public struct IlmSdfTraceInfo {
        public ulong CellBytes;
        public ulong CellRebuilds;
        public ulong CellSlicesRebuilt;
        public int LastRebuiltSlices;
        public int TableSlices;
        public int RebuiltEveryFrame;
        public int Reserved;
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 48)]
    public unsafe struct IlmObstruction {
        public fixed float Center[3];
        public int Type;
        public fixed float Size[3];
        public int IsDynamic;
        public fixed float Orientation[4];
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 32)]
    public unsafe struct IlmHeightVolume {
        public int FirstVertex;
        public int VertexCount;
        public float ZBase;
        public float Height;
        public int IsDynamic;
        public int TopFaceEnableShadows;
        public fixed int _pad[2];
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 64)]
    public unsafe struct IlmDistanceFieldRenderDesc {
        public int VirtualWidth;
        public int VirtualHeight;
        public float VirtualDepth;
        public float ZOffset;
        public int SliceWidth;
        public int SliceHeight;
        public int SliceCount;
        public int ColumnCount;
        public int RowCount;
        public float MaximumEncodedDistance;
        public float InvScaleFactorX;
        public float InvScaleFactorY;
        public int DynamicFlagFilter;
        public fixed int _pad[3];
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 32)]
    public unsafe struct IlmGBufferRenderDesc {
        public fixed float ViewportPosition[2];
        public fixed float ViewportScale[2];
        public float GroundZ;
        public int RenderGroundPlane;
        public int EnableGroundShadows;
        public int _pad;
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 36)]
    public unsafe struct IlmHeightVolumeVertex {
        public fixed float Position[3];
        public fixed float Normal[3];
        public fixed float ZRange[2];
        public float EnableShadows;
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 48)]
    public unsafe struct IlmBillboardVertex {
        public fixed float ScreenPosition[2];
        public fixed float TexCoord[2];
        public fixed float WorldPosition[3];
        public fixed float Normal[3];
        public fixed float DataScaleAndDynamicFlag[2];
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 24)]
    public struct IlmBillboardRun {
        public ulong Texture;
        public int FirstQuad;
        public int QuadCount;
        public int Type;
        public int _pad;
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 64)]
    public unsafe struct IlmGBufferMeshDesc {
        public fixed float ViewportPosition[2];
        public fixed float ViewportScale[2];
        public float GroundZ;
        public float ZToYMultiplier;
        public fixed float RenderScale[2];
        public float DistanceFieldExtentZ;
        public float SelfOcclusionHack;
        public float ZSelfOcclusionHack;
        public int TwoPointFiveD;
        public int RenderGroundPlane;
        public int EnableGroundShadows;
        public fixed int _pad[2];
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 24)]
    public struct IlmRenderStats {
        public ulong SdfSamples;
        public ulong PixelLightPairs;
        public ulong TracedPairs;
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 80)]
    public unsafe struct IlmParticleLightParams {
        public Vector4 LightProperties;
        public Vector4 MoreLightProperties;
        public Vector4 LightColor;
        public Vector4 LightSpecularColor;
        public float StippleFactor;
        public fixed float _pad[3];
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 48)]
    public unsafe struct IlmReadbackDrawCall {
        public fixed float Position[2];
        public fixed float Scale[2];
        public fixed float TextureRegion[4];
        public float Rotation;
        public float SortOrder;
        public fixed byte MultiplyColor[4];
        public int _pad;
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 56)]
    public unsafe struct IlmReadbackParams {
        public fixed float Size[2];
        public fixed float TextureRegion[4];
        public fixed float AnimationRate[2];
        public float ZToY;
        public int ColumnFromVelocity;
        public int RowFromVelocity;
        public int RotationFromVelocity;
        public int SortedReadback;
        public int _pad;
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 192)]
    public unsafe struct IlmRasterizeParams {
        public Vector4 GlobalColor;
        public Vector4 BitmapTextureRegion;
        public Vector4 SizeFactorAndPosition;
        public Vector4 Scale;
        public Vector4 ZFormula;
        public Vector4 ZConfiguration;
        public IlmClampedBezier1 RoundingPowerFromLife;
        public fixed float RenderingOptions[4];
        public fixed float SystemSize[2];
        public float ZToY;
        public float StippleFactor;
        public fixed float ViewportScale[2];
        public fixed float ViewportPosition[2];
        public int BlendMode;
        public int BitmapFilter;
        public fixed float AnimationRate[2];
    }

    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = 48)]
    public struct IlmHDRConfiguration {
        public int Mode;
        public float InverseScaleFactor;
        public float Offset;
        public float Exposure;
        public float Gamma;
        public float MiddleGray;
        public float AverageLuminance;
        public float MaximumLuminance;
        public float WhitePoint;
        public int ResolveToSRGB;
        public int DitheringStrength;
        public int AlbedoIsSRGB;
    }

    internal static unsafe class IlluminantHip {
        const string Lib = "illuminant_hip";             // libilluminant_hip.so next to the game's assemblies

        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_abi_version ();
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern IntPtr ilm_last_error ();
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_device_count ();
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_debug_reference_constant (byte* key, double* outValue);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_debug_reference_constant_count ();
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern IntPtr ilm_debug_reference_constant_key (int index);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_ctx_create (int deviceId, ulong* outCtx);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_ctx_create_sibling (ulong ctx, ulong* outCtx);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_ctx_destroy (ulong ctx);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_ctx_sync (ulong ctx);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_ctx_stream (ulong ctx, void** outStream);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_timer_start (ulong ctx);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_timer_stop (ulong ctx, float* outMs);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_engine_create (ulong ctx, int chunkSize, Vector4* randomness, int randWidth, int randHeight, ulong* outEngine);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_engine_destroy (ulong engine);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_system_create (ulong engine, ulong* outSystem);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_system_destroy (ulong system);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_system_add_chunk (ulong system, int* outChunkIndex);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_system_remove_chunk (ulong system, int chunkIndex);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_system_chunk_count (ulong system, int* outCount);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_chunk_upload (ulong system, int chunkIndex, int plane, Vector4* src, int firstSlot, int count);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_chunk_download (ulong system, int chunkIndex, int plane, Vector4* dst, int firstSlot, int count);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_chunk_device_ptr (ulong system, int chunkIndex, int component, void** outPtr, long* outStrideFloats);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_system_set_distance_field (ulong system, ulong sdf);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_system_set_life_ramp (ulong system, Vector4* texels, int width, int height);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_system_set_spawn_positions (ulong system, int spawnSlot, Vector4* positions, int count);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_system_set_spawn_pattern (ulong system, int spawnSlot, Vector4* texels, int width, int height, int levels);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_system_step (ulong system, IlmStepDesc* desc);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_spawn (ulong system, int chunkIndex, IlmParticleSystemUniforms* sys, IlmSpawnParams* p);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_gravity (ulong system, int chunkIndex, IlmParticleSystemUniforms* sys, IlmGravityParams* p);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_noise (ulong system, int chunkIndex, IlmParticleSystemUniforms* sys, IlmNoiseParams* p);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_fma (ulong system, int chunkIndex, IlmParticleSystemUniforms* sys, IlmFMAParams* p);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_matrix_multiply (ulong system, int chunkIndex, IlmParticleSystemUniforms* sys, IlmMatrixMultiplyParams* p);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_spatial_noise (ulong system, int chunkIndex, IlmParticleSystemUniforms* sys, IlmSpatialNoiseParams* p);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_update (ulong system, int chunkIndex, IlmParticleSystemUniforms* sys, IlmUpdateParams* p, IlmDistanceFieldUniforms* df);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_erase (ulong system, int chunkIndex);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_system_live_counts (ulong system, uint* outCounts, int capacity, int saturate16);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_system_step_counts (ulong system, uint* outCounts, int capacity, int saturate16);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_system_poll_counts (ulong system, uint* outCounts, int capacity, int saturate16, int* outReady);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_chunk_live_slots (ulong system, int chunkIndex, uint* outSlots, int capacity, int* outCount);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_sdf_create (ulong ctx, int atlasWidth, int atlasHeight, int format, ulong* outSdf);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_sdf_upload (ulong sdf, ushort* texels);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_sdf_sample (ulong sdf, IlmDistanceFieldUniforms* df, float* positions, int count, float* outDistances);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_debug_sdf_sample_inside (ulong sdf, IlmDistanceFieldUniforms* df, float* positions, int count, float* outDistances, int* outUsedTable);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_debug_divide (ulong ctx, float* numerators, float* denominators, int count, float* outFast, float* outIeee);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_debug_divide_by_constants (ulong ctx, float* outDivisors, ulong* outMismatches, int capacity, int* outCount);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_debug_step_interpreter (int interpreter);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_debug_step_streams (int streams);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_debug_step_sdf_samples (ulong ctx, int enable, ulong* outSamples);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_debug_last_light_launch (ulong ctx, int* outWorkgroups, int* outSplit, int* outTileMacro);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_sdf_destroy (ulong sdf);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_sdf_download (ulong sdf, ushort* texels);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_sdf_device_ptr (ulong sdf, void** outPtr);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_sdf_mark_dirty (ulong sdf, int firstVirtualSlice, int sliceCount);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_sdf_trace_info (ulong sdf, IlmSdfTraceInfo* @out);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_sdf_render_slices (ulong sdf, ulong clearSource, IlmDistanceFieldRenderDesc* desc, int* firstVirtualSlices, int tripletCount, IlmObstruction* obstructions, int obstructionCount, IlmHeightVolume* volumes, int volumeCount, float* polygonXy, int polygonVertexCount);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_gbuffer_create (ulong ctx, int width, int height, int format, ulong* outGbuffer);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_gbuffer_upload (ulong gbuffer, void* texels);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_gbuffer_destroy (ulong gbuffer);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_gbuffer_download (ulong gbuffer, void* texels);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_gbuffer_render (ulong gbuffer, IlmGBufferRenderDesc* desc, IlmHeightVolume* volumes, int volumeCount, float* polygonXy, int polygonVertexCount);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_gbuffer_render_meshes (ulong gbuffer, IlmGBufferMeshDesc* desc, IlmHeightVolumeVertex* topVertices, int topVertexCount, IlmHeightVolumeVertex* frontVertices, int frontVertexCount, IlmBillboardVertex* billboardVertices, int billboardVertexCount, IlmBillboardRun* runs, int runCount);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_lightmap_create (ulong ctx, int width, int height, int format, void* externalDevicePtr, ulong* outLightmap);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_lightmap_download (ulong lightmap, void* dst, int firstRow, int rowCount);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_lightmap_upload (ulong lightmap, void* src, int firstRow, int rowCount);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_lightmap_device_ptr (ulong lightmap, void** outPtr);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_lightmap_destroy (ulong lightmap);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_render_sphere_lights (ulong ctx, LightVertex* lights, int lightCount, IlmEnvironment* env, IlmDistanceFieldUniforms* df, ulong gbuffer, ulong sdf, float* ambient, ulong lightmap, int rowBegin, int rowEnd, IlmRenderStats* stats);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_ctx_set_light_ramp (ulong ctx, Vector4* texels, int width, int height);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_ctx_set_lightmap_blend (ulong ctx, int mode);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_ctx_set_light_split (ulong ctx, int workgroups);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_render_particle_lights (ulong ctx, ulong system, int* quadCounts, int chunkCount, IlmParticleLightParams* @params, IlmEnvironment* env, IlmDistanceFieldUniforms* df, ulong gbuffer, ulong sdf, ulong lightmap, int rowBegin, int rowEnd, IlmRenderStats* stats);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_render_light_probes (ulong ctx, LightVertex* lights, int lightCount, Vector4* probePositions, Vector4* probeNormals, int probeCount, IlmEnvironment* env, IlmDistanceFieldUniforms* df, ulong sdf, Vector4* outValues);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_system_readback (ulong system, int* elementCounts, int chunkCount, IlmReadbackParams* @params, IlmReadbackDrawCall* @out, int capacity, int* outCount);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_system_readback_view (ulong system, int* elementCounts, int chunkCount, IlmReadbackParams* @params, IlmReadbackDrawCall** outRecords, int* outCount);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_render_particles (ulong system, int* quadCounts, int chunkCount, IlmRasterizeParams* @params, ulong target, ulong* outStats);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_system_set_bitmap (ulong system, Vector4* texels, int width, int height);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_lightmap_clear (ulong lightmap, float* rgba);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_resolve_lighting (ulong srcLightmap, ulong dstLightmap, IlmHDRConfiguration* hdr, int rowBegin, int rowEnd);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_resolve_lighting_with_albedo (ulong srcLightmap, ulong albedo, ulong dstLightmap, IlmHDRConfiguration* hdr, int rowBegin, int rowEnd);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_group_create (int* deviceIds, int n, ulong* outGroup);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_group_unique_id (void* outId128);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_group_create_rank (int deviceId, int rank, int world, void* id128, ulong* outGroup);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_group_destroy (ulong group);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_group_info (ulong group, int* outLocal, int* outWorld, int* outFirstRank, int* outCommRanks);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_group_ctx (ulong group, int localIndex, ulong* outCtx);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_group_sync (ulong group);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_group_all_gather (ulong group, void* const* buffers, ulong bytesPerRank, int gather);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_group_host_all_gather (ulong group, void* local, void* outAll, uint bytesPerRank);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_group_lightmap_create (ulong group, int width, int height, int format, ulong* outGroupLightmap);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_group_lightmap_member (ulong groupLightmap, int localIndex, ulong* outLightmap);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_group_lightmap_strip (ulong groupLightmap, int rank, int* outRowBegin, int* outRowEnd, int* outSlotRows);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_group_lightmap_set_strips (ulong groupLightmap, int* rowBegins, int* rowEnds);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_group_lightmap_gather (ulong groupLightmap, int gather);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_group_lightmap_wait (ulong groupLightmap);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_group_lightmap_store_mode (ulong groupLightmap, int enable);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_group_lightmap_destroy (ulong groupLightmap);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_group_render_sphere_lights (ulong group, LightVertex* lights, int lightCount, IlmEnvironment* env, IlmDistanceFieldUniforms* df, ulong* gbuffers, ulong* sdfs, float* ambient, ulong groupLightmap, int gather, IlmRenderStats* stats);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_group_live_counts (ulong group, ulong* systems, int totalChunks, uint* outCounts, int capacity, int saturate16);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int ilm_group_gather_chunks (ulong group, ulong* sources, ulong* gathered, int totalChunks, int firstComponent, int componentCount, int gather);

        public static void Check (int code) {
            if (code == 0) return;
            var msg = Marshal.PtrToStringAnsi(ilm_last_error());
            // the reference's own exception types for the conditions it checks itself
            if (code == IlmConstants.ERR_TOO_MANY) throw new InvalidOperationException(msg);   // "Maximum number of attractors per instance is 16" (Transforms.cs:348-349)
            if (code == IlmConstants.ERR_STATE) throw new InvalidOperationException(msg);      // distance field update without a field (ParticleSystem.cs:835-836)
            throw new IlluminantHipException(code, msg);
        }
    }
}
