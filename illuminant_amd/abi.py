"""ctypes mirror of include/illuminant_hip.h (POD structs, constants).

Byte layouts equal the reference's own uniform / vertex structs
(Illuminant/Uniforms.cs, Illuminant/Bezier.cs:433-441,588-599,
Illuminant/Vertices.cs:10-39); tests/test_abi_layout.py checks the sizes and
offsets against the C header by compiling a probe.
"""
import ctypes as C

ABI_VERSION = 9

OK = 0
ERR_INVALID_ARGUMENT = -1
ERR_INVALID_HANDLE = -2
ERR_OUT_OF_RANGE = -3
ERR_TOO_MANY = -4
ERR_NO_DEVICE = -5
ERR_STATE = -6

MAX_ATTRACTORS = 16
MAX_INLINE_POSITION_CONSTANTS = 4
MAX_OPS = 4
MAX_SPAWNS = 2
RANDOMNESS_WIDTH = 807
RANDOMNESS_HEIGHT = 653

OP_GRAVITY, OP_NOISE, OP_FMA, OP_MATRIX_MULTIPLY, OP_SPATIAL_NOISE = 1, 2, 3, 4, 5
SPAWN_INLINE, SPAWN_POSITION_BUFFER, SPAWN_FEEDBACK, SPAWN_PATTERN = 0, 1, 2, 3
UPDATE_NONE, UPDATE_POSITIONS, UPDATE_WITH_DISTANCE_FIELD, UPDATE_ERASE = 0, 1, 2, 3
STEP_COUNT_LIVE = 1

SDF_UNORM16, SDF_FP16 = 0, 1
GBUFFER_FLOAT4, GBUFFER_HALF4 = 0, 1
LIGHTMAP_FLOAT4, LIGHTMAP_HALF4, LIGHTMAP_RGBA8 = 0, 1, 2
PLANE_POSITION, PLANE_VELOCITY, PLANE_ATTRIBUTES, PLANE_RENDER_COLOR, PLANE_RENDER_DATA = 0, 1, 2, 3, 4

Handle = C.c_uint64
f32 = C.c_float
i32 = C.c_int32
u32 = C.c_uint32


class Float4(C.Structure):
    _fields_ = [("x", f32), ("y", f32), ("z", f32), ("w", f32)]

    def __init__(self, x=0.0, y=0.0, z=0.0, w=0.0):
        super().__init__(float(x), float(y), float(z), float(w))

    def tuple(self):
        return (self.x, self.y, self.z, self.w)


def f4(*v):
    if len(v) == 1 and hasattr(v[0], "__len__"):
        v = tuple(v[0])
    v = tuple(v) + (0.0,) * (4 - len(v))
    return Float4(*v[:4])


class Matrix(C.Structure):
    _fields_ = [("m", f32 * 16)]

    @staticmethod
    def identity():
        r = Matrix()
        for i in (0, 5, 10, 15):
            r.m[i] = 1.0
        return r

    @staticmethod
    def from_rows(rows):
        r = Matrix()
        flat = [float(x) for row in rows for x in row]
        for i in range(16):
            r.m[i] = flat[i]
        return r


class ParticleSystemUniforms(C.Structure):
    _fields_ = [("GlobalSettings", Float4), ("CollisionSettings", Float4),
                ("TexelAndSize", Float4), ("AnimationRateAndRotationAndZToY", Float4)]


class ClampedBezier1(C.Structure):
    _fields_ = [("RangeAndCount", Float4), ("ABCD", Float4)]

    @staticmethod
    def one():
        # ClampedBezier1.One, Bezier.cs:434-437
        return ClampedBezier1(f4(0, 1, 1, 0), f4(1, 1, 1, 1))

    @staticmethod
    def constant(v):
        # new ClampedBezier1(new BezierF(v)): one control point, Bezier.cs:439-460
        return ClampedBezier1(f4(0, 1, 1, 0), f4(v, v, v, v))

    @staticmethod
    def linear(a, b, lo, hi):
        # two control points a -> b over [lo, hi] (count 2, clamped range: RangeAndCount = (min, 1 / (max - min), count, mode))
        return ClampedBezier1(f4(lo, 1.0 / max(hi - lo, 1e-6), 2, 0), f4(a, b, b, b))


class ClampedBezier4(C.Structure):
    _fields_ = [("RangeAndCount", Float4), ("A", Float4), ("B", Float4), ("C", Float4), ("D", Float4)]

    @staticmethod
    def one():
        # ClampedBezier4.One, Bezier.cs:589-595
        o = f4(1, 1, 1, 1)
        return ClampedBezier4(f4(0, 1, 1, 0), o, f4(1, 1, 1, 1), f4(1, 1, 1, 1), f4(1, 1, 1, 1))


class DistanceFieldUniforms(C.Structure):
    _fields_ = [("ConeAndMisc", Float4), ("TextureSliceAndTexelSize", Float4), ("StepAndMisc2", Float4),
                ("TextureSliceCount", Float4), ("Extent", Float4), ("Packed1", Float4)]


class Environment(C.Structure):
    _fields_ = [("ZAndScale", Float4), ("ZToY", Float4), ("GBufferTexelSizeAndMisc", Float4),
                ("ViewportPosition", f32 * 2), ("GBufferViewportRelative", f32), ("_pad0", f32)]


class SdfTraceInfo(C.Structure):
    _fields_ = [("CellBytes", C.c_uint64), ("CellRebuilds", C.c_uint64), ("CellSlicesRebuilt", C.c_uint64),
                ("LastRebuiltSlices", i32), ("TableSlices", i32), ("RebuiltEveryFrame", i32), ("Reserved", i32)]


class LightVertex(C.Structure):
    _fields_ = [("LightPosition1", Float4), ("LightPosition2", Float4), ("LightPosition3", Float4),
                ("LightProperties", Float4), ("MoreLightProperties", Float4), ("EvenMoreLightProperties", Float4),
                ("Color1", Float4), ("Color2", Float4)]


class AreaParams(C.Structure):
    _fields_ = [("AreaType", i32), ("Strength", f32), ("AreaFalloff", f32), ("AreaRotation", f32),
                ("AreaCenter", f32 * 3), ("_pad0", f32), ("AreaSize", f32 * 3), ("_pad1", f32),
                ("CategoryFilter", f32 * 2), ("_pad2", f32 * 2)]


class GravityParams(C.Structure):
    _fields_ = [("AttractorCount", i32), ("MaximumAcceleration", f32), ("CategoryFilter", f32 * 2),
                ("AttractorPositions", (f32 * 3) * MAX_ATTRACTORS),
                ("AttractorRadiusesAndStrengths", (f32 * 3) * MAX_ATTRACTORS)]


class FMAParams(C.Structure):
    _fields_ = [("Area", AreaParams), ("TimeDivisor", f32), ("_pad", f32 * 3),
                ("PositionAdd", Float4), ("PositionMultiply", Float4),
                ("VelocityAdd", Float4), ("VelocityMultiply", Float4)]


class NoiseParams(C.Structure):
    _fields_ = [("Area", AreaParams), ("TimeDivisor", f32), ("FrequencyLerp", f32),
                ("ReplaceOldVelocity", f32), ("_pad", f32),
                ("RandomnessOffset", f32 * 2), ("NextRandomnessOffset", f32 * 2),
                ("PositionOffset", Float4), ("PositionMinimum", Float4), ("PositionScale", Float4),
                ("VelocityOffset", Float4), ("VelocityMinimum", Float4), ("VelocityScale", Float4)]


class SpawnParams(C.Structure):
    _fields_ = [("ChunkSizeAndIndices", f32 * 4), ("Configuration", Float4 * 9), ("FormulaTypes", f32 * 4),
                ("PositionMatrix", Matrix), ("VelocityMatrix", Matrix),
                ("AxisMask", f32 * 3), ("AlignVelocityAndPosition", f32),
                ("RandomnessOffset", f32 * 2), ("AttributeDiscardThreshold", f32), ("PolygonRate", f32),
                ("PolygonLoop", f32), ("PositionConstantCount", f32), ("_pad", f32 * 2),
                ("InlinePositionConstants", Float4 * MAX_INLINE_POSITION_CONSTANTS)]


class UpdateParams(C.Structure):
    _fields_ = [("ColorFromLife", ClampedBezier4), ("ColorFromVelocity", ClampedBezier4),
                ("SizeFromLife", ClampedBezier1), ("SizeFromVelocity", ClampedBezier1),
                ("RotationFromLifeAndIndex", f32 * 2), ("_pad", f32 * 2), ("LifeRampSettings", Float4)]

    @staticmethod
    def default():
        u = UpdateParams()
        u.ColorFromLife = ClampedBezier4.one()
        u.ColorFromVelocity = ClampedBezier4.one()
        u.SizeFromLife = ClampedBezier1.one()
        u.SizeFromVelocity = ClampedBezier1.one()
        u.LifeRampSettings = f4(0, 0, 1, 1)  # ParticleSystem.cs:937-939
        return u


class MatrixMultiplyParams(C.Structure):
    _fields_ = [("Area", AreaParams), ("TimeDivisor", f32), ("_pad", f32 * 3), ("PositionMatrix", Matrix), ("VelocityMatrix", Matrix)]


class SpatialNoiseParams(C.Structure):
    _fields_ = [("Noise", NoiseParams), ("SpaceScale", f32 * 2), ("_pad", f32 * 2)]


class FeedbackParams(C.Structure):
    _fields_ = [("SourceSystem", Handle), ("SourceChunkIndex", i32), ("FeedbackSourceIndex", f32), ("InstanceMultiplier", f32),
                ("SourceVelocityFactor", f32), ("AlignPositionConstant", f32), ("MultiplyLife", f32), ("MultiplyAttributeConstant", f32),
                ("SourceLifeRange", f32 * 2), ("_pad", f32)]


class PatternParams(C.Structure):
    _fields_ = [("StepWidthAndSizeScale", f32 * 4), ("YOffsetsAndCoordScale", f32 * 4), ("TexelOffsetAndMipBias", f32 * 4),
                ("CenteringOffset", f32 * 2), ("MultiplyAttributeConstant", f32), ("_pad", f32)]


BLEND_ALPHA, BLEND_ADDITIVE = 0, 1
BITMAP_NONE, BITMAP_POINT, BITMAP_LINEAR = 0, 1, 2


class RasterizeParams(C.Structure):
    _fields_ = [("GlobalColor", Float4), ("BitmapTextureRegion", Float4), ("SizeFactorAndPosition", Float4), ("Scale", Float4),
                ("ZFormula", Float4), ("ZConfiguration", Float4), ("RoundingPowerFromLife", ClampedBezier1),
                ("RenderingOptions", f32 * 4), ("SystemSize", f32 * 2), ("ZToY", f32), ("StippleFactor", f32),
                ("ViewportScale", f32 * 2), ("ViewportPosition", f32 * 2), ("BlendMode", i32), ("BitmapFilter", i32),
                ("AnimationRate", f32 * 2)]


class _OpUnion(C.Union):
    _fields_ = [("Gravity", GravityParams), ("Noise", NoiseParams), ("FMA", FMAParams),
                ("MatrixMultiply", MatrixMultiplyParams), ("SpatialNoise", SpatialNoiseParams)]


class TransformOp(C.Structure):
    _fields_ = [("Type", i32), ("_pad", i32 * 3), ("u", _OpUnion)]


class SpawnRecord(C.Structure):
    _fields_ = [("ChunkIndex", i32), ("Kind", i32), ("_pad", i32 * 2), ("Params", SpawnParams), ("Feedback", FeedbackParams),
                ("Pattern", PatternParams)]


class StepDesc(C.Structure):
    _fields_ = [("FirstChunk", i32), ("ChunkCount", i32), ("OpCount", i32), ("SpawnCount", i32),
                ("UpdateMode", i32), ("Flags", u32), ("_pad", i32 * 2),
                ("System", ParticleSystemUniforms), ("Update", UpdateParams),
                ("DistanceField", DistanceFieldUniforms),
                ("Ops", TransformOp * MAX_OPS), ("Spawns", SpawnRecord * MAX_SPAWNS)]


OBSTRUCTION_ELLIPSOID, OBSTRUCTION_BOX, OBSTRUCTION_CYLINDER, OBSTRUCTION_SPHEROID, OBSTRUCTION_OCTAGON = 0, 1, 2, 3, 4
DISTANCE_LIMIT = 520.0


class Obstruction(C.Structure):
    _fields_ = [("Center", f32 * 3), ("Type", i32), ("Size", f32 * 3), ("IsDynamic", i32), ("Orientation", f32 * 4)]


class HeightVolume(C.Structure):
    _fields_ = [("FirstVertex", i32), ("VertexCount", i32), ("ZBase", f32), ("Height", f32),
                ("IsDynamic", i32), ("TopFaceEnableShadows", i32), ("_pad", i32 * 2)]


class GBufferRenderDesc(C.Structure):
    _fields_ = [("ViewportPosition", f32 * 2), ("ViewportScale", f32 * 2), ("GroundZ", f32), ("RenderGroundPlane", i32),
                ("EnableGroundShadows", i32), ("_pad", i32)]


class HeightVolumeVertex(C.Structure):
    """HeightVolumeVertex, Illuminant/Vertices.cs:41-46 (Pack = 4)"""
    _fields_ = [("Position", f32 * 3), ("Normal", f32 * 3), ("ZRange", f32 * 2), ("EnableShadows", f32)]


class BillboardVertex(C.Structure):
    """BillboardVertex, Illuminant/Vertices.cs:75-81 (Pack = 4)"""
    _fields_ = [("ScreenPosition", f32 * 2), ("TexCoord", f32 * 2), ("WorldPosition", f32 * 3), ("Normal", f32 * 3),
                ("DataScaleAndDynamicFlag", f32 * 2)]


BILLBOARD_MASK, BILLBOARD_GBUFFER_DATA = 0, 1


class BillboardRun(C.Structure):
    _fields_ = [("Texture", Handle), ("FirstQuad", i32), ("QuadCount", i32), ("Type", i32), ("_pad", i32)]


class GBufferMeshDesc(C.Structure):
    _fields_ = [("ViewportPosition", f32 * 2), ("ViewportScale", f32 * 2), ("GroundZ", f32), ("ZToYMultiplier", f32),
                ("RenderScale", f32 * 2), ("DistanceFieldExtentZ", f32), ("SelfOcclusionHack", f32), ("ZSelfOcclusionHack", f32),
                ("TwoPointFiveD", i32), ("RenderGroundPlane", i32), ("EnableGroundShadows", i32), ("_pad", i32 * 2)]


class DistanceFieldRenderDesc(C.Structure):
    _fields_ = [("VirtualWidth", i32), ("VirtualHeight", i32), ("VirtualDepth", f32), ("ZOffset", f32),
                ("SliceWidth", i32), ("SliceHeight", i32), ("SliceCount", i32), ("ColumnCount", i32),
                ("RowCount", i32), ("MaximumEncodedDistance", f32), ("InvScaleFactorX", f32), ("InvScaleFactorY", f32),
                ("DynamicFlagFilter", i32), ("_pad", i32 * 3)]


class ParticleLightParams(C.Structure):
    _fields_ = [("LightProperties", Float4), ("MoreLightProperties", Float4), ("LightColor", Float4), ("LightSpecularColor", Float4),
                ("StippleFactor", f32), ("_pad", f32 * 3)]


HDR_NONE, HDR_GAMMA_COMPRESS, HDR_TONE_MAP = 0, 1, 2


class ReadbackDrawCall(C.Structure):
    _fields_ = [("Position", f32 * 2), ("Scale", f32 * 2), ("TextureRegion", f32 * 4), ("Rotation", f32), ("SortOrder", f32),
                ("MultiplyColor", C.c_uint8 * 4), ("_pad", i32)]


class ReadbackParams(C.Structure):
    _fields_ = [("Size", f32 * 2), ("TextureRegion", f32 * 4), ("AnimationRate", f32 * 2), ("ZToY", f32),
                ("ColumnFromVelocity", i32), ("RowFromVelocity", i32), ("RotationFromVelocity", i32), ("SortedReadback", i32), ("_pad", i32)]


class HDRConfiguration(C.Structure):
    _fields_ = [("Mode", i32), ("InverseScaleFactor", f32), ("Offset", f32), ("Exposure", f32), ("Gamma", f32),
                ("MiddleGray", f32), ("AverageLuminance", f32), ("MaximumLuminance", f32), ("WhitePoint", f32),
                ("ResolveToSRGB", i32), ("DitheringStrength", i32), ("AlbedoIsSRGB", i32)]


class RenderStats(C.Structure):
    _fields_ = [("SdfSamples", C.c_uint64), ("PixelLightPairs", C.c_uint64), ("TracedPairs", C.c_uint64)]


# expected sizes (bytes) -- checked against the C header in tests
EXPECTED_SIZES = {
    "IlmFloat4": (Float4, 16), "IlmMatrix": (Matrix, 64),
    "IlmParticleSystemUniforms": (ParticleSystemUniforms, 64),
    "IlmClampedBezier1": (ClampedBezier1, 32), "IlmClampedBezier4": (ClampedBezier4, 80),
    "IlmDistanceFieldUniforms": (DistanceFieldUniforms, 96), "IlmEnvironment": (Environment, 64),
    "IlmLightVertex": (LightVertex, 128), "IlmAreaParams": (AreaParams, 64),
    "IlmGravityParams": (GravityParams, 400), "IlmFMAParams": (FMAParams, 144),
    "IlmNoiseParams": (NoiseParams, 192), "IlmSpawnParams": (SpawnParams, 416),
    "IlmUpdateParams": (UpdateParams, 256), "IlmTransformOp": (TransformOp, 416),
    "IlmSpawnRecord": (SpawnRecord, 544), "IlmStepDesc": (StepDesc, 3200),
    "IlmMatrixMultiplyParams": (MatrixMultiplyParams, 208), "IlmSpatialNoiseParams": (SpatialNoiseParams, 208),
    "IlmFeedbackParams": (FeedbackParams, 48), "IlmPatternParams": (PatternParams, 64),
    "IlmRenderStats": (RenderStats, 24),
    "IlmParticleLightParams": (ParticleLightParams, 80),
    "IlmReadbackDrawCall": (ReadbackDrawCall, 48), "IlmReadbackParams": (ReadbackParams, 56), "IlmHDRConfiguration": (HDRConfiguration, 48),
    "IlmGBufferRenderDesc": (GBufferRenderDesc, 32), "IlmRasterizeParams": (RasterizeParams, 192),
    "IlmObstruction": (Obstruction, 48), "IlmHeightVolume": (HeightVolume, 32),
    "IlmDistanceFieldRenderDesc": (DistanceFieldRenderDesc, 64),
    "IlmHeightVolumeVertex": (HeightVolumeVertex, 36), "IlmBillboardVertex": (BillboardVertex, 48),
    "IlmBillboardRun": (BillboardRun, 24), "IlmGBufferMeshDesc": (GBufferMeshDesc, 64),
}
