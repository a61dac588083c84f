"""What the reference's C# host puts into the uniform blocks and light vertices the two hot paths read, written from the reference's
text alone -- NOT from illuminant_amd/scenes.py, whose builders every GPU test uses ("Next" #4b: the second reading took
its uniforms from the same builders as oracle and kernels, so a misreading of what goes into DistanceFieldPacked1 or
TextureSliceAndTexelSize was shared by all three).  tests/golden/second_reading.py builds its inputs with THIS file;
tests/test_reference_uniforms.py (CPU) holds scenes.py's builders to it byte for byte.

Only the containers come from the package: illuminant_amd.abi's ctypes mirrors of the C ABI structs, whose field order
tests/test_reference_pin.py pins to Uniforms.cs / Vertices.cs.  C# arithmetic is reproduced by type: `float` expressions in numpy
float32 operation by operation, `double` in Python floats, integer division as integer division, Math.Round as round-half-to-even,
Math.Ceiling of a float quotient on the float32 quotient.
"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from illuminant_amd import abi  # noqa: E402

f32 = np.float32


class ReferenceDistanceField:
    """Squared.Illuminant.DistanceField's constructor, Illuminant/SDF/DistanceField.cs:43-122 (MaxSurfaceSize 8192 :19,
    LightingRenderer.PackedSliceCount 3), and GetExtent4 :145-152."""

    def __init__(self, virtualWidth, virtualHeight, virtualDepth, requestedSliceCount, requestedResolution=1.0, maximumEncodedDistance=128):
        self.VirtualWidth, self.VirtualHeight, self.VirtualDepth = int(virtualWidth), int(virtualHeight), f32(virtualDepth)
        self.MaximumEncodedDistance = int(maximumEncodedDistance)
        if requestedResolution < 0.05:                      # :56-59
            requestedResolution = 0.05
        elif requestedResolution > 1:
            requestedResolution = 1
        candidateSliceWidth = int(round(self.VirtualWidth * requestedResolution))          # (int)Math.Round(double): half to even, :61-62
        candidateSliceHeight = int(round(self.VirtualHeight * requestedResolution))
        fracX = self.VirtualWidth / candidateSliceWidth                                     # doubles, :64-66
        fracY = self.VirtualHeight / candidateSliceHeight
        frac = (fracX + fracY) / 2
        resolution = round(1.0 / frac, 3)                                                   # Math.Round(x, 3), :68
        if resolution < 0.05:
            resolution = 0.05
        elif resolution > 1:
            resolution = 1
        self.Resolution = resolution
        self.SliceWidth = int(round(self.VirtualWidth * self.Resolution))                   # :76-77
        self.SliceHeight = int(round(self.VirtualHeight * self.Resolution))
        maxSlicesX = 8192 // self.SliceWidth                                                # int / int, :79-81
        maxSlicesY = 8192 // self.SliceHeight
        maxSlices = maxSlicesX * maxSlicesY * 3
        sliceCount = max(3, int(requestedSliceCount))                                       # :83-84
        sliceCount = ((sliceCount + 2) // 3) * 3
        self.SliceCount = min(sliceCount, maxSlices)
        self.PhysicalSliceCount = int(math.ceil(f32(self.SliceCount) / f32(3)))             # Math.Ceiling(SliceCount / (float)3), :87
        self.ColumnCount = min(maxSlicesX, self.PhysicalSliceCount)                         # :91-92
        self.RowCount = min(maxSlicesY, max(int(math.ceil(f32(self.PhysicalSliceCount) / f32(maxSlicesX))), 1))
        while self.RowCount < self.ColumnCount and self.RowCount < maxSlicesY:              # :96-109
            newRowCount = self.RowCount + 1
            newColumnCount = int(math.ceil(f32(self.PhysicalSliceCount) / f32(newRowCount)))
            if newRowCount > maxSlicesX:
                newRowCount = maxSlicesX
            if newColumnCount > maxSlicesY:
                newColumnCount = maxSlicesY
            if newRowCount * newColumnCount < self.PhysicalSliceCount:
                break
            self.RowCount, self.ColumnCount = newRowCount, newColumnCount
        self.TextureWidth, self.TextureHeight = self.SliceWidth * self.ColumnCount, self.SliceHeight * self.RowCount     # :111-116
        self.ValidSliceCount = self.SliceCount        # a fully generated field (SliceInfo.ValidSliceCount, IsFullyGenerated :124-129)
        self.ZOffset = f32(0)


def distance_field_uniforms(df, MaxConeRadius=24.0, OcclusionToOpacityPower=1.0, MaxStepCount=64, MinStepSize=3.0, LongStepFactor=1.0,
                            set_packed1=True):
    """new Uniforms.DistanceField(df) (Illuminant/Uniforms.cs:90-110) with the initialiser of SetDistanceFieldParameters
    (Illuminant/Lighting/LightingRenderer.cs:1916-1923; the quality defaults are RendererQualitySettings', LightingRenderer.Configuration.cs:
    254-291) and the DistanceFieldPacked1 it sets beside the block (:1932-1939).  set_packed1=False: the particle path, which binds the block
    through the same struct but never sets Packed1 (it stays zero)."""
    u = abi.DistanceFieldUniforms()
    u.Extent = abi.f4(df.VirtualWidth, df.VirtualHeight, df.VirtualDepth, df.MaximumEncodedDistance)                 # GetExtent4
    sliceZSize = f32(df.VirtualDepth) / f32(df.SliceCount)                                                             # float / int -> float, :93
    u.TextureSliceCount = abi.f4(df.ColumnCount, df.RowCount, f32(min(df.ValidSliceCount, df.SliceCount)) * sliceZSize, df.SliceCount)   # :94-98
    u.TextureSliceAndTexelSize = abi.f4(f32(1) / f32(df.ColumnCount), f32(1) / f32(df.RowCount),                      # :99-103
                                        f32(1) / f32(df.VirtualWidth * df.ColumnCount), f32(1) / f32(df.VirtualHeight * df.RowCount))
    inv_x = f32(df.VirtualWidth / df.SliceWidth)                  # (float)((double)VirtualWidth / SliceWidth), :105
    inv_y = f32(df.VirtualHeight / df.SliceHeight)                # :106
    # _ConeAndMisc = (MaxConeRadius, DistanceFieldZOffset, OcclusionToOpacityPower, InvScaleFactorX): accessors :136-176
    u.ConeAndMisc = abi.f4(f32(MaxConeRadius), f32(df.ZOffset), f32(OcclusionToOpacityPower), inv_x)
    # _StepAndMisc2 = (StepLimit, MinimumLength, LongStepFactor, InvScaleFactorY): accessors :112-134, :178-186; the ctor's LongStepFactor 1 is overwritten
    u.StepAndMisc2 = abi.f4(f32(int(MaxStepCount)), f32(MinStepSize), f32(LongStepFactor), inv_y)
    if set_packed1:
        tsc_x, tsc_w, tsc_z, ext_z = f32(u.TextureSliceCount.x), f32(u.TextureSliceCount.w), f32(u.TextureSliceCount.z), f32(u.Extent.z)
        u.Packed1 = abi.f4((f32(1.0) / max(f32(0.0001), tsc_x)) * (f32(1.0) / f32(3.0)),          # float arithmetic (the FIXME above it says so)
                           (f32(1.0) / max(f32(0.0001), ext_z)) * tsc_w, tsc_z, f32(MinStepSize))
    else:
        u.Packed1 = abi.f4(0, 0, 0, 0)
    return u


def particle_system_uniforms(ChunkSize, deltaTimeSeconds, Size=(1.0, 1.0), Friction=0.0, MaximumVelocity=9999.0, LifeDecayPerSecond=1.0,
                             Collision=None, AnimationRate=(0.0, 0.0), RotationFromVelocity=False, ZToY=0.0):
    """new Uniforms.ParticleSystem(Engine, Configuration, deltaTimeSeconds), Illuminant/Uniforms.cs:207-234 (VelocityConstantScale 1000 :199).
    Collision = (EscapeVelocity, BounceVelocityMultiplier, Distance, LifePenalty) or None."""
    u = abi.ParticleSystemUniforms()
    u.TexelAndSize = abi.f4(f32(1) / f32(int(ChunkSize)), f32(1) / f32(int(ChunkSize)), f32(Size[0]), f32(Size[1]))
    u.GlobalSettings = abi.f4(f32(float(deltaTimeSeconds) * 1000), f32(Friction), f32(MaximumVelocity), f32(LifeDecayPerSecond))     # (float)(double * int)
    u.CollisionSettings = abi.f4(*[f32(v) for v in Collision]) if Collision is not None else abi.f4(0, 0, 0, 0)
    ax, ay = f32(AnimationRate[0]), f32(AnimationRate[1])
    u.AnimationRateAndRotationAndZToY = abi.f4(f32(1.0) / ax if ax != 0 else f32(0), f32(1.0) / ay if ay != 0 else f32(0),
                                               f32(1) if RotationFromVelocity else f32(0), f32(ZToY))
    return u


def sphere_light_vertex(Position, Radius, RampLength, Color=(1, 1, 1, 1), Opacity=1.0, intensityScale=1.0, RampMode=0, CastsShadows=True,
                        have_distance_field=True, AmbientOcclusionRadius=0.0, AmbientOcclusionOpacity=1.0, FalloffYFactor=1.0,
                        ShadowDistanceFalloff=None, ShadowFilter=-1, SpecularColor=(0, 0, 0), SpecularPower=1.0, RampOffsetAndRate=(0.0, 1.0)):
    """RenderSphereLightSource, Illuminant/Lighting/LightingRenderer.cs:1193-1219; RampOffsetForGPU / RampRateForGPU, Illuminant/Lighting/
    LightSource.cs:97-98.  (A light with Opacity <= 0 is not drawn at all, :1195-1196.)"""
    v = abi.LightVertex()
    v.LightPosition1 = v.LightPosition2 = v.LightPosition3 = abi.f4(f32(Position[0]), f32(Position[1]), f32(Position[2]), 0)
    # color.W *= (lightSource.Opacity * intensityScale): float * float, then float *= float
    v.Color1 = abi.f4(f32(Color[0]), f32(Color[1]), f32(Color[2]), f32(Color[3]) * (f32(Opacity) * f32(intensityScale)))
    v.Color2 = abi.f4(f32(SpecularColor[0]), f32(SpecularColor[1]), f32(SpecularColor[2]), f32(SpecularPower))
    v.LightProperties = abi.f4(f32(Radius), f32(RampLength), f32(int(RampMode)), f32(1) if (CastsShadows and have_distance_field) else f32(0))
    v.MoreLightProperties = abi.f4(f32(AmbientOcclusionRadius), f32(-99999) if ShadowDistanceFalloff is None else f32(ShadowDistanceFalloff),
                                   f32(FalloffYFactor), f32(AmbientOcclusionOpacity))
    v.EvenMoreLightProperties = abi.f4(f32(int(ShadowFilter)), 0, f32(-math.pi) + f32(RampOffsetAndRate[0]),
                                       f32(1.0 / (math.pi * 2) * float(RampOffsetAndRate[1])))
    return v


def environment_uniforms(GroundZ=0.0, MaximumZ=128.0, ZToYMultiplier=0.0, TwoPointFiveD=False, LightOcclusion=0.0, RenderScale=(1.0, 1.0),
                         gbuffer_size=None, ViewportScale=(1.0, 1.0), ViewportPosition=(0.0, 0.0), GBufferViewportRelative=False):
    """ComputeUniforms (Illuminant/Lighting/LightingRenderer.cs:691-701; the Environment block's accessors Illuminant/Uniforms.cs:14-75:
    _ZAndScale = (GroundZ, MaximumZ, RenderScale.xy), _ZToY = (ZToYMultiplier, 1 / ZToYMultiplier or 0, LightOcclusion, 0)) and
    SetGBufferParameters (Illuminant/Lighting/LightingRenderer.GBuffer.cs:520-534: the G-buffer's InverseSize or zeros, ViewportScale)."""
    e = abi.Environment()
    z_to_y = f32(ZToYMultiplier) if TwoPointFiveD else f32(0)
    e.ZAndScale = abi.f4(f32(GroundZ), f32(MaximumZ), f32(RenderScale[0]), f32(RenderScale[1]))
    e.ZToY = abi.f4(z_to_y, f32(0) if abs(float(z_to_y)) <= 0.0001 else f32(1.0) / z_to_y, f32(LightOcclusion), 0)
    if gbuffer_size is not None:
        e.GBufferTexelSizeAndMisc = abi.f4(f32(1.0) / f32(gbuffer_size[0]), f32(1.0) / f32(gbuffer_size[1]), f32(ViewportScale[0]), f32(ViewportScale[1]))
    else:
        e.GBufferTexelSizeAndMisc = abi.f4(0, 0, f32(ViewportScale[0]), f32(ViewportScale[1]))
    e.ViewportPosition[0], e.ViewportPosition[1] = f32(ViewportPosition[0]), f32(ViewportPosition[1])
    e.GBufferViewportRelative = 1.0 if GBufferViewportRelative else 0.0
    return e
