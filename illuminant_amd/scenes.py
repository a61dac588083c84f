"""Deterministic synthetic scenes of the shapes BASELINE.json names.

Host-side input synthesis only (numpy): randomness tables, particle initial
state, analytic SDF atlases, light lists and the parameter structs of each
config.  Nothing here is on the hot path; the GPU work goes through
illuminant_amd.native (the C ABI).  The generator is SplitMix64 -> uniform
float, so seeds alone reproduce every input on the GPU box.
"""
import ctypes as C
import math

import numpy as np

from . import abi

_MASK = (1 << 64) - 1


def splitmix64(seed, n):
    """n uint64 draws of SplitMix64 seeded with `seed` (vectorised)."""
    idx = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) + np.uint64(seed & _MASK)
    z = idx
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def uniform(seed, shape, lo=0.0, hi=1.0):
    """float32 uniforms in [lo, hi) from the top 24 bits of SplitMix64."""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        bits = splitmix64(seed, n)
    u = (bits >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return (lo + u * (hi - lo)).astype(np.float32).reshape(shape)


def randomness_table(seed, width=abi.RANDOMNESS_WIDTH, height=abi.RANDOMNESS_HEIGHT):
    """The 807x653 float4 table (ParticleEngine.cs:495-544); unseeded in the reference, so explicit here."""
    return uniform(seed, (height, width, 4))


# ---- particle-side parameter builders ------------------------------------------------------------------

def system_uniforms(chunk_size, dt_seconds=1.0 / 60, friction=0.0, max_velocity=9999.0, life_decay=1.0,
                    collision=(128.0, 0.0, 0.33, 0.0), size=(1.0, 1.0), rotation_from_velocity=False, z_to_y=0.0):
    """Uniforms.ParticleSystem ctor, Uniforms.cs:207-235."""
    u = abi.ParticleSystemUniforms()
    u.GlobalSettings = abi.f4(np.float32(dt_seconds * 1000.0), friction, max_velocity, life_decay)
    u.CollisionSettings = abi.f4(*collision)
    u.TexelAndSize = abi.f4(1.0 / chunk_size, 1.0 / chunk_size, size[0], size[1])
    u.AnimationRateAndRotationAndZToY = abi.f4(0, 0, 1.0 if rotation_from_velocity else 0.0, z_to_y)
    return u


def area_none(strength=1.0, category_filter=(-9999.0, 9999.0)):
    """ParticleAreaTransform.SetParameters with Area == null (ParticleTransform.cs:294-318).
    AreaFalloff is left at the effect default 0: distance/falloff = 0/0 = NaN, saturate(NaN) = 0."""
    a = abi.AreaParams()
    a.AreaType = 0
    a.Strength = strength
    a.AreaFalloff = 0.0
    a.CategoryFilter[0], a.CategoryFilter[1] = category_filter
    return a


def area(type_id, center, size, falloff=1.0, rotation=0.0, strength=1.0, category_filter=(-9999.0, 9999.0)):
    a = abi.AreaParams()
    a.AreaType = type_id
    a.Strength = strength
    a.AreaFalloff = max(1.0, falloff)   # ParticleTransform.cs:301-302
    a.AreaRotation = rotation
    for i in range(3):
        a.AreaCenter[i] = center[i]
        a.AreaSize[i] = size[i]
    a.CategoryFilter[0], a.CategoryFilter[1] = category_filter
    return a


def gravity_params(attractors, maximum_acceleration=8.0, category_filter=(0.0, 0.0)):
    """attractors: [(position xyz, radius, strength, type)] ; Transforms.cs:347-365.
    category_filter defaults to the effect default (0, 0): the reference never binds it for Gravity."""
    g = abi.GravityParams()
    g.AttractorCount = len(attractors)
    g.MaximumAcceleration = maximum_acceleration
    g.CategoryFilter[0], g.CategoryFilter[1] = category_filter
    for i, (pos, radius, strength, typ) in enumerate(attractors[:abi.MAX_ATTRACTORS]):
        for k in range(3):
            g.AttractorPositions[i][k] = pos[k]
        g.AttractorRadiusesAndStrengths[i][0] = radius
        g.AttractorRadiusesAndStrengths[i][1] = strength
        g.AttractorRadiusesAndStrengths[i][2] = float(typ)
    return g


def fma_params(area_params, cycles_per_second=10.0, position_add=(0, 0, 0), position_multiply=(1, 1, 1),
               velocity_add=(0, 0, 0), velocity_multiply=(1, 1, 1)):
    """FMA.SetParameters, Transforms.cs:38-45."""
    p = abi.FMAParams()
    p.Area = area_params
    p.TimeDivisor = (1000.0 / cycles_per_second) if cycles_per_second is not None else -1.0
    p.PositionAdd = abi.f4(*position_add, 0)
    p.PositionMultiply = abi.f4(*position_multiply, 1)
    p.VelocityAdd = abi.f4(*velocity_add, 0)
    p.VelocityMultiply = abi.f4(*velocity_multiply, 1)
    return p


def noise_params(area_params, offsets, next_offsets, frequency_lerp, cycles_per_second=10.0, replace_old_velocity=True,
                 position=((-0.5,) * 4, (0,) * 4, (0,) * 4), velocity=((-0.5,) * 3, (0,) * 3, (1,) * 3), speed=(-0.5, 0.0, 0.0)):
    """Noise.SetParameters, Transforms.cs:243-268; defaults = Noise ctor (:192-204).
    offsets = (CurrentU*253, CurrentV*127), next_offsets likewise."""
    p = abi.NoiseParams()
    p.Area = area_params
    p.TimeDivisor = (1000.0 / cycles_per_second) if cycles_per_second is not None else -1.0
    p.FrequencyLerp = frequency_lerp
    p.ReplaceOldVelocity = 1.0 if replace_old_velocity else 0.0
    p.RandomnessOffset[0], p.RandomnessOffset[1] = offsets
    p.NextRandomnessOffset[0], p.NextRandomnessOffset[1] = next_offsets
    p.PositionOffset, p.PositionMinimum, p.PositionScale = abi.f4(position[0]), abi.f4(position[1]), abi.f4(position[2])
    p.VelocityOffset = abi.f4(*velocity[0], speed[0])
    p.VelocityMinimum = abi.f4(*velocity[1], speed[1])
    p.VelocityScale = abi.f4(*velocity[2], speed[2])
    return p


def matrix_multiply_params(area_params, position_matrix=None, velocity_matrix=None, cycles_per_second=10.0):
    """MatrixMultiply.SetParameters, Transforms.cs:61-66."""
    p = abi.MatrixMultiplyParams()
    p.Area = area_params
    p.TimeDivisor = (1000.0 / cycles_per_second) if cycles_per_second is not None else -1.0
    p.PositionMatrix = position_matrix if position_matrix is not None else abi.Matrix.identity()
    p.VelocityMatrix = velocity_matrix if velocity_matrix is not None else abi.Matrix.identity()
    return p


def spatial_noise_params(noise, space_scale=(1.0, 1.0)):
    """SpatialNoise.SetParameters, Transforms.cs:288-293: SpaceScale uniform = 1 / scale."""
    p = abi.SpatialNoiseParams()
    p.Noise = noise
    p.SpaceScale[0] = float(np.float32(1.0) / np.float32(space_scale[0]))
    p.SpaceScale[1] = float(np.float32(1.0) / np.float32(space_scale[1]))
    return p


def position_buffer_spawn_params(chunk_size, first, last, total_spawned, randomness_offset, positions, life_constant=1.0, **kw):
    """Spawner with more than 4 positions (technique SpawnParticlesFromPositionTexture): returns (SpawnParams, (n, 4) position list).
    positions: [(x, y, z)] including the spawner's own Position.Constant first (ParticleSpawner.cs:319-353)."""
    polygon_rate = kw.get("polygon_rate")
    polygon_loop = kw.get("polygon_loop", True)
    base = dict(kw)
    base["position"] = (tuple(positions[0]),) + tuple(kw.get("position", ((0, 0, 0), (1, 1, 1), (0, 0, 0), FORMULA_SPHERICAL))[1:])
    base["life"] = (life_constant,) + tuple(kw.get("life", (1.0, 0.0, 0.0))[1:])
    p = spawn_params(chunk_size, first, last, total_spawned, randomness_offset, **base)
    count = len(positions)
    rate = float(polygon_rate) if polygon_rate is not None else 0.0
    if rate >= 1:
        c = count - 1 if (not polygon_loop and count > 1) else count
        w = math.fmod(np.float32(total_spawned / rate), float(c))
    else:
        w = total_spawned % count
    p.ChunkSizeAndIndices[3] = w
    p.PositionConstantCount = float(count)
    buf = np.zeros((count, 4), np.float32)
    buf[:, :3] = np.asarray(positions, np.float32)
    buf[:, 3] = life_constant
    return p, buf


def feedback_params(source_system_handle, source_chunk_index, source_index, instance_multiplier=1, source_velocity_factor=0.0,
                    align_position_constant=True, multiply_life=False, multiply_color_constant=False, source_life_range=(0.0, 9999.0)):
    """FeedbackSpawner.SetParameters, SpecialSpawners.cs:411-427 (defaults :262-300)."""
    f = abi.FeedbackParams()
    f.SourceSystem = source_system_handle
    f.SourceChunkIndex = source_chunk_index
    f.FeedbackSourceIndex = float(source_index)
    f.InstanceMultiplier = float(instance_multiplier)
    f.SourceVelocityFactor = source_velocity_factor
    f.AlignPositionConstant = 1.0 if align_position_constant else 0.0
    f.MultiplyLife = 1.0 if multiply_life else 0.0
    f.MultiplyAttributeConstant = 1.0 if multiply_color_constant else 0.0
    f.SourceLifeRange[0], f.SourceLifeRange[1] = source_life_range
    return f


def rasterize_params(size=(1.0, 1.0), global_color=(1.0, 1.0, 1.0, 1.0), origin=(0.0, 0.0), scale=(1.0, 1.0), size_from_z=0.0, z_to_y=0.0,
                     rounded=False, rounding_power=None, viewport_scale=(1.0, 1.0), viewport_position=(0.0, 0.0), blend=abi.BLEND_ALPHA,
                     z_formula=(0.0, 0.0, 0.0, 0.0), stipple_factor=1.0, texture_size=None, offset_px=(0.0, 0.0), size_px=None,
                     relative_size=True, bilinear=True, animation_rate=(0.0, 0.0), column_from_velocity=False, row_from_velocity=False, dithered_opacity=False):
    """Uniforms.RasterizeParticleSystem (Uniforms.cs:238-290) + what ParticleSystem.Render / RenderHandler._BeforeDraw add
    (ParticleSystem.cs:254-271, 963-1032).  global_color is Color.Global (NOT premultiplied: the ctor does that); rounding_power: a
    ClampedBezier1 or None for the constant 0.8 (ParticleAppearance.RoundingPowerFromLife default).  texture_size = (w, h) of
    Appearance.Texture or None (technique NoTexture); offset_px / size_px = the frame rectangle; animation_rate = Appearance.AnimationRate
    (the uniform holds its reciprocal, Uniforms.cs:230-234)."""
    f = np.float32
    p = abi.RasterizeParams()
    gc = np.asarray(global_color, np.float32)
    p.GlobalColor = abi.f4(gc[0] * gc[3], gc[1] * gc[3], gc[2] * gc[3], gc[3])
    if texture_size is not None:
        tw, th = f(texture_size[0]), f(texture_size[1])
        sw, sh = (f(size_px[0]), f(size_px[1])) if size_px is not None else (tw, th)
        ox, oy = f(offset_px[0]) / tw, f(offset_px[1]) / th
        p.BitmapTextureRegion = abi.f4(ox, oy, ox + sw / tw, oy + sh / th)
        p.SizeFactorAndPosition = abi.f4(sw * f(0.5), sh * f(0.5), origin[0], origin[1]) if relative_size else abi.f4(1, 1, origin[0], origin[1])
        p.BitmapFilter = abi.BITMAP_LINEAR if bilinear else abi.BITMAP_POINT
    else:
        p.BitmapTextureRegion = abi.f4(0, 0, 1, 1)
        p.SizeFactorAndPosition = abi.f4(1, 1, origin[0], origin[1])
        p.BitmapFilter = abi.BITMAP_NONE
    p.Scale = abi.f4(scale[0], scale[1], 0, 0)
    p.ZFormula = abi.f4(*z_formula)
    p.ZConfiguration = abi.f4(size_from_z, 0, 0, 0)
    if rounding_power is None:
        rounding_power = abi.ClampedBezier1.constant(0.8)
    p.RoundingPowerFromLife = rounding_power
    p.RenderingOptions[:] = [1.0 if rounded else 0.0, 1.0 if dithered_opacity else 0.0, 1.0 if column_from_velocity else 0.0, 1.0 if row_from_velocity else 0.0]
    p.SystemSize[:] = list(size)
    p.ZToY = z_to_y
    p.StippleFactor = stipple_factor
    p.ViewportScale[:] = list(viewport_scale)
    p.ViewportPosition[:] = list(viewport_position)
    p.BlendMode = blend
    p.AnimationRate[:] = [float(f(1.0) / f(animation_rate[0])) if animation_rate[0] != 0 else 0.0,
                          float(f(1.0) / f(animation_rate[1])) if animation_rate[1] != 0 else 0.0]
    return p


def next_power_of_two(v):
    """Arithmetic.NextPowerOfTwo (Fracture, not in the tree): the smallest power of two >= v; 0 for v <= 0 (PatternSpawner.BeginTick treats
    a non-positive particle count as "nothing to spawn", SpecialSpawners.cs:189-194)."""
    v = int(v)
    if v <= 0:
        return 0
    return 1 << (v - 1).bit_length()


def pattern_direct_texture_size(tex_w, tex_h, top_left_px=None, size_px=None):
    """PatternSpawner.DirectTextureSize, SpecialSpawners.cs:76-97 (the top-left corner is subtracted even from an explicit size)."""
    w, h = np.float32(tex_w), np.float32(tex_h)
    if size_px is not None:
        if size_px[0] > 0:
            w = np.float32(size_px[0])
        if size_px[1] > 0:
            h = np.float32(size_px[1])
    if top_left_px is not None:
        w, h = np.float32(w - np.float32(top_left_px[0])), np.float32(h - np.float32(top_left_px[1]))
    return w, h


def pattern_counts(tex_w, tex_h, divisor=1, top_left_px=None, size_px=None):
    """(ParticlesPerRow, RowsPerInstance) of a PatternSpawner, SpecialSpawners.cs:111-127: integer divisions, then NextPowerOfTwo."""
    w, h = pattern_direct_texture_size(tex_w, tex_h, top_left_px, size_px)
    return next_power_of_two(int(w) // divisor), next_power_of_two(int(h) // divisor)


def pattern_params(tex_w, tex_h, divisor=1, current_row=0, top_left_px=None, size_px=None, mip_bias_base=-0.5,
                   multiply_color_constant=True):
    """PatternSpawner.SetParameters, SpecialSpawners.cs:208-256, in float32.  current_row = 0 for WholeSpawn, else
    RowsSpawned % RowsPerInstance.  `(currentRow * Divisor) / tex.Height` is an INTEGER division in the reference (:237)."""
    f = np.float32
    per_row, _rows = pattern_counts(tex_w, tex_h, divisor, top_left_px, size_px)
    p = abi.PatternParams()
    p.StepWidthAndSizeScale[:] = [f(divisor), f(per_row), f(divisor) / f(tex_w), f(divisor) / f(tex_h)]
    base_x = base_y = f(0)
    if top_left_px is not None:
        base_x, base_y = f(top_left_px[0]) / f(tex_w), f(top_left_px[1]) / f(tex_h)
    p.YOffsetsAndCoordScale[:] = [f(current_row), f((current_row * divisor) // tex_h), f(divisor), f(divisor)]
    p.TexelOffsetAndMipBias[:] = [f(-0.5) / f(tex_w) + base_x, f(-0.5) / f(tex_h) + base_y, f(0),
                                  f(math.log(divisor) / math.log(2.0)) + f(mip_bias_base)]
    dw, dh = pattern_direct_texture_size(tex_w, tex_h, top_left_px, size_px)
    p.CenteringOffset[:] = [dw * f(-0.5), dh * f(-0.5)]
    p.MultiplyAttributeConstant = 1.0 if multiply_color_constant else 0.0
    return p


def pattern_mip_chain(texels, levels=None):
    """A synthetic mip chain for tests and benchmarks: 2x2 box filter (edge texels repeated for odd sizes).  The reference's chain comes
    from its texture loader, which is not in the tree; the C ABI takes the levels as given."""
    a = np.ascontiguousarray(texels, dtype=np.float32)
    out = [a]
    while (levels is None and (a.shape[0] > 1 or a.shape[1] > 1)) or (levels is not None and len(out) < levels):
        h, w = a.shape[0], a.shape[1]
        nh, nw = max(1, h >> 1), max(1, w >> 1)
        ys0 = np.minimum(np.arange(nh) * 2, h - 1); ys1 = np.minimum(ys0 + 1, h - 1)
        xs0 = np.minimum(np.arange(nw) * 2, w - 1); xs1 = np.minimum(xs0 + 1, w - 1)
        a = ((a[ys0][:, xs0] + a[ys0][:, xs1] + a[ys1][:, xs0] + a[ys1][:, xs1]) * np.float32(0.25)).astype(np.float32)
        out.append(a)
    return out


FORMULA_LINEAR, FORMULA_SPHERICAL, FORMULA_TOWARDS, FORMULA_RECTANGULAR = 0, 1, 2, 3


def spawn_params(chunk_size, first, last, total_spawned, randomness_offset,
                 position=((0, 0, 0), (1, 1, 1), (0, 0, 0), FORMULA_SPHERICAL),
                 velocity=((0, 0, 0), (1, 1, 1), (0, 0, 0), FORMULA_SPHERICAL),
                 life=(1.0, 0.0, 0.0), category=(0.0, 0.0, 0.0),
                 color=((1, 1, 1, 1), (0, 0, 0, 0), (0, 0, 0, 0)), color_type=FORMULA_LINEAR,
                 additional_positions=(), polygon_rate=None, polygon_loop=True, polygon_speed=(0.0, 0.0, 0.0),
                 axis_mask=(1, 1, 1), align=False, alpha_discard_threshold=1.0,
                 position_matrix=None, velocity_matrix=None):
    """SpawnerBase.SetParameters + Spawner.SetParameters/GetChunkSizeAndIndices/BeginTick,
    ParticleSpawner.cs:200-256, 319-403.  formula tuples are (constant, random scale, offset[, type])."""
    p = abi.SpawnParams()
    count = 1 + len(additional_positions)
    assert count <= abi.MAX_INLINE_POSITION_CONSTANTS
    rate = float(polygon_rate) if polygon_rate is not None else 0.0
    if rate >= 1:
        c = count - 1 if (not polygon_loop and count > 1) else count
        w = math.fmod(np.float32(total_spawned / rate), float(c))
    else:
        w = total_spawned % count
    p.ChunkSizeAndIndices[:] = [chunk_size, first, last, w]
    p.Configuration[0] = abi.f4(*position[1], life[1])
    p.Configuration[1] = abi.f4(*position[2], life[2])
    p.Configuration[2] = abi.f4(*velocity[0], category[0])
    p.Configuration[3] = abi.f4(*velocity[1], category[1])
    p.Configuration[4] = abi.f4(*velocity[2], category[2])
    p.Configuration[5] = abi.f4(*color[0])
    p.Configuration[6] = abi.f4(*color[1])
    p.Configuration[7] = abi.f4(*color[2])
    p.Configuration[8] = abi.f4(polygon_speed[0], polygon_speed[1], polygon_speed[2], 0)
    p.FormulaTypes[:] = [float(position[3]), float(velocity[3]), float(color_type), 0.0]
    p.PositionMatrix = position_matrix if position_matrix is not None else abi.Matrix.identity()
    p.VelocityMatrix = velocity_matrix if velocity_matrix is not None else abi.Matrix.identity()
    p.AxisMask[:] = axis_mask
    circular = position[3] in (FORMULA_SPHERICAL, FORMULA_RECTANGULAR) and velocity[3] in (FORMULA_SPHERICAL, FORMULA_RECTANGULAR)
    p.AlignVelocityAndPosition = 1.0 if (align and circular) else 0.0
    p.RandomnessOffset[0], p.RandomnessOffset[1] = randomness_offset
    p.AttributeDiscardThreshold = np.float32(alpha_discard_threshold / 255.0)
    p.PolygonRate = rate
    p.PolygonLoop = 1.0 if polygon_loop else 0.0
    p.PositionConstantCount = float(count)
    p.InlinePositionConstants[0] = abi.f4(*position[0], life[0])
    for i, ap in enumerate(additional_positions):
        p.InlinePositionConstants[i + 1] = abi.f4(*ap, life[0])
    return p


def make_particles(seed, n, pos_lo=(0, 0, 0), pos_hi=(256, 256, 32), vel=60.0, life=(1.0, 6.0), dead_fraction=0.0,
                   categories=(0.0,)):
    """AoS float4 position/velocity/attribute planes of n slots."""
    pos = np.zeros((n, 4), np.float32)
    velo = np.zeros((n, 4), np.float32)
    for k in range(3):
        pos[:, k] = uniform(seed + 11 * k + 1, (n,), pos_lo[k], pos_hi[k])
    pos[:, 3] = uniform(seed + 41, (n,), life[0], life[1])
    velo[:, 0] = uniform(seed + 43, (n,), -vel, vel)
    velo[:, 1] = uniform(seed + 47, (n,), -vel, vel)
    velo[:, 2] = uniform(seed + 53, (n,), -vel * 0.25, vel * 0.25)
    cat = uniform(seed + 59, (n,), 0.0, float(len(categories)))
    velo[:, 3] = np.asarray(categories, np.float32)[np.minimum(cat.astype(np.int64), len(categories) - 1)]
    attr = np.ones((n, 4), np.float32)
    attr[:, :3] = uniform(seed + 61, (n, 3), 0.09, 0.39)
    if dead_fraction > 0:
        dead = uniform(seed + 67, (n,)) < dead_fraction
        pos[dead] = 0
        velo[dead] = 0
    return pos, velo, attr


# ---- distance field ------------------------------------------------------------------------------------

class DistanceFieldLayout:
    """Host mirror of the DistanceField ctor's atlas layout math (SDF/DistanceField.cs:43-122)."""
    MAX_SURFACE_SIZE = 8192
    PACKED_SLICE_COUNT = 3

    def __init__(self, virtual_width, virtual_height, virtual_depth, requested_slice_count, requested_resolution=1.0,
                 maximum_encoded_distance=128):
        self.virtual_width, self.virtual_height, self.virtual_depth = virtual_width, virtual_height, float(virtual_depth)
        self.maximum_encoded_distance = maximum_encoded_distance
        rr = min(max(requested_resolution, 0.05), 1.0)
        cw = int(round(virtual_width * rr))     # Python round == Math.Round (banker's)
        ch = int(round(virtual_height * rr))
        frac = (virtual_width / cw + virtual_height / ch) / 2
        res = round(1.0 / frac, 3)
        self.resolution = min(max(res, 0.05), 1.0)
        self.slice_width = int(round(virtual_width * self.resolution))
        self.slice_height = int(round(virtual_height * self.resolution))
        max_x = self.MAX_SURFACE_SIZE // self.slice_width
        max_y = self.MAX_SURFACE_SIZE // self.slice_height
        max_slices = max_x * max_y * self.PACKED_SLICE_COUNT
        sc = max(3, requested_slice_count)
        sc = ((sc + 2) // 3) * 3
        self.slice_count = min(sc, max_slices)
        self.physical_slice_count = int(math.ceil(np.float32(self.slice_count) / np.float32(3)))
        cols = min(max_x, self.physical_slice_count)
        rows = min(max_y, max(int(math.ceil(np.float32(self.physical_slice_count) / np.float32(max_x))), 1))
        while rows < cols and rows < max_y:
            nr = rows + 1
            nc = int(math.ceil(np.float32(self.physical_slice_count) / np.float32(nr)))
            nr = min(nr, max_x)
            nc = min(nc, max_y)
            if nr * nc < self.physical_slice_count:
                break
            rows, cols = nr, nc
        self.column_count, self.row_count = cols, rows
        self.atlas_width, self.atlas_height = self.slice_width * cols, self.slice_height * rows

    def slice_index_to_z(self, s, z_offset=0.0):
        """SliceIndexToZ, LightingRenderer.DistanceField.cs:32-35."""
        return float(np.float32(np.float32(s) / np.float32(max(1, self.slice_count))) * np.float32(self.virtual_depth) + np.float32(z_offset))

    def uniforms(self, valid_slice_count=None, z_offset=0.0, max_cone_radius=24.0, power=1.0, step_limit=64,
                 min_step_size=3.0, long_step_factor=1.0, packed1=True):
        """Uniforms.DistanceField ctor (Uniforms.cs:90-110) + SetDistanceFieldParameters
        (LightingRenderer.cs:1916-1939).  packed1=False leaves DistanceFieldPacked1 at zero, which is what the
        reference's particle path does (only LightingRenderer ever sets it)."""
        f = np.float32
        u = abi.DistanceFieldUniforms()
        valid = self.slice_count if valid_slice_count is None else min(valid_slice_count, self.slice_count)
        slice_z = f(self.virtual_depth) / f(self.slice_count)
        u.Extent = abi.f4(self.virtual_width, self.virtual_height, self.virtual_depth, self.maximum_encoded_distance)
        u.TextureSliceCount = abi.f4(self.column_count, self.row_count, f(valid) * slice_z, self.slice_count)
        u.TextureSliceAndTexelSize = abi.f4(f(1) / f(self.column_count), f(1) / f(self.row_count),
                                            f(1) / f(self.virtual_width * self.column_count),
                                            f(1) / f(self.virtual_height * self.row_count))
        u.ConeAndMisc = abi.f4(max_cone_radius, z_offset, power, f(self.virtual_width / self.slice_width))
        u.StepAndMisc2 = abi.f4(step_limit, min_step_size, long_step_factor, f(self.virtual_height / self.slice_height))
        if packed1:
            u.Packed1 = abi.f4((f(1) / max(f(0.0001), f(u.TextureSliceCount.x))) * (f(1) / f(3)),
                               (f(1) / max(f(0.0001), f(u.Extent.z))) * f(u.TextureSliceCount.w),
                               u.TextureSliceCount.z, min_step_size)
        return u


def _sd_ellipsoid(px, py, pz, size):
    k0 = np.sqrt((px / size[0]) ** 2 + (py / size[1]) ** 2 + (pz / size[2]) ** 2)
    k1 = np.sqrt((px / size[0] ** 2) ** 2 + (py / size[1] ** 2) ** 2 + (pz / size[2] ** 2) ** 2)
    with np.errstate(divide="ignore", invalid="ignore"):
        outer = k0 * (k0 - 1.0) / k1
    return np.where(k0 < 1.0, (k0 - 1.0) * min(size), outer)


def _sd_box(px, py, pz, size):
    dx, dy, dz = np.abs(px) - size[0], np.abs(py) - size[1], np.abs(pz) - size[2]
    outside = np.sqrt(np.maximum(dx, 0) ** 2 + np.maximum(dy, 0) ** 2 + np.maximum(dz, 0) ** 2)
    return np.minimum(np.maximum(dx, np.maximum(dy, dz)), 0.0) + outside


def _sd_cylinder(px, py, pz, size):
    r = math.hypot(size[0], size[1])
    dx = np.sqrt(px * px + py * py) - r
    dy = np.abs(pz) - size[2]
    return np.minimum(np.maximum(dx, dy), 0.0) + np.sqrt(np.maximum(dx, 0) ** 2 + np.maximum(dy, 0) ** 2)


_SD = {1: _sd_ellipsoid, 2: _sd_box, 3: _sd_cylinder}


def build_sdf_atlas(layout, obstacles, z_offset=0.0, fmt=abi.SDF_UNORM16):
    """Rasterise analytic obstructions [(type 1|2|3, center xyz, size xyz)] into the packed atlas the way
    RenderDistanceFieldSliceTriplet does (LightingRenderer.DistanceField.cs:80-152, DistanceFunction.fx:33-80):
    clear to encoded 0, MAX-blend encodeDistance(d) of every obstruction; texel RGBA = slices 3p..3p+3.
    Returns (H, W, 4) uint16."""
    L = layout
    enc = np.zeros((L.atlas_height, L.atlas_width, 4), np.float32)
    sx = L.virtual_width / L.slice_width
    sy = L.virtual_height / L.slice_height
    maxd = float(L.maximum_encoded_distance)
    for p in range(L.physical_slice_count):
        ox = (p % L.column_count) * L.slice_width
        oy = (p // L.column_count) * L.slice_height
        zs = [L.slice_index_to_z(3 * p + c, z_offset) for c in range(4)]
        for (typ, center, size) in obstacles:
            # DistanceFunctionVertexShader's quad (DistanceFunction.fx:16-26): half-size max|size| + maxDistance + 4 around the
            # centre; texel i is covered iff its centre i + 0.5 lies in [left, right) (in slice pixels)
            reach = max(size) + maxd + 4
            x0 = max(int(math.ceil((center[0] - reach) / sx - 0.5)), 0)
            x1 = min(int(math.ceil((center[0] + reach) / sx - 0.5)), L.slice_width)
            y0 = max(int(math.ceil((center[1] - reach) / sy - 0.5)), 0)
            y1 = min(int(math.ceil((center[1] + reach) / sy - 0.5)), L.slice_height)
            if x0 >= x1 or y0 >= y1:
                continue
            wx = (np.arange(x0, x1, dtype=np.float32) * np.float32(sx))[None, :] - np.float32(center[0])
            wy = (np.arange(y0, y1, dtype=np.float32) * np.float32(sy))[:, None] - np.float32(center[1])
            for c in range(4):
                d = _SD[typ](wx, wy, np.float32(zs[c] - center[2]), size)
                e = np.float32(192.0 / 255.0) - d.astype(np.float32) / np.float32(maxd)
                view = enc[oy + y0:oy + y1, ox + x0:ox + x1, c]
                np.maximum(view, e, out=view)
    enc = np.clip(enc, 0.0, 1.0)
    if fmt == abi.SDF_FP16:
        return enc.astype(np.float16).view(np.uint16)
    return np.rint(enc * 65535.0).astype(np.uint16)


def render_desc(layout, z_offset=0.0, dynamic_flag_filter=-1):
    """IlmDistanceFieldRenderDesc of a layout: what RenderDistanceFieldSliceTriplet binds
    (LightingRenderer.DistanceField.cs:80-152; inverse scale factors as in Uniforms.cs:108-109)."""
    d = abi.DistanceFieldRenderDesc()
    d.VirtualWidth, d.VirtualHeight, d.VirtualDepth, d.ZOffset = layout.virtual_width, layout.virtual_height, layout.virtual_depth, z_offset
    d.SliceWidth, d.SliceHeight, d.SliceCount = layout.slice_width, layout.slice_height, layout.slice_count
    d.ColumnCount, d.RowCount = layout.column_count, layout.row_count
    d.MaximumEncodedDistance = float(layout.maximum_encoded_distance)
    d.InvScaleFactorX = float(np.float32(layout.virtual_width / layout.slice_width))
    d.InvScaleFactorY = float(np.float32(layout.virtual_height / layout.slice_height))
    d.DynamicFlagFilter = dynamic_flag_filter
    return d


def gbuffer_render_desc(ground_z=0.0, viewport_position=(0.0, 0.0), viewport_scale=(1.0, 1.0), render_ground_plane=True, enable_ground_shadows=True):
    """What RenderGBuffer binds (LightingRenderer.GBuffer.cs:127-203)."""
    d = abi.GBufferRenderDesc()
    d.ViewportPosition[:] = viewport_position
    d.ViewportScale[:] = viewport_scale
    d.GroundZ = ground_z
    d.RenderGroundPlane = int(render_ground_plane)
    d.EnableGroundShadows = int(enable_ground_shadows)
    return d


def gbuffer_mesh_desc(ground_z=0.0, viewport_position=(0.0, 0.0), viewport_scale=(1.0, 1.0), z_to_y=0.0, render_scale=(1.0, 1.0),
                      extent_z=128.0, self_occlusion_hack=0.0, z_self_occlusion_hack=0.0, two_point_five_d=True,
                      render_ground_plane=True, enable_ground_shadows=True):
    """What RenderGBuffer / _SetupGBufferGroundPlane bind (LightingRenderer.GBuffer.cs:102-157)."""
    d = abi.GBufferMeshDesc()
    d.ViewportPosition[:] = viewport_position
    d.ViewportScale[:] = viewport_scale
    d.GroundZ, d.ZToYMultiplier = ground_z, z_to_y
    d.RenderScale[:] = render_scale
    d.DistanceFieldExtentZ, d.SelfOcclusionHack, d.ZSelfOcclusionHack = extent_z, self_occlusion_hack, z_self_occlusion_hack
    d.TwoPointFiveD, d.RenderGroundPlane, d.EnableGroundShadows = int(two_point_five_d), int(render_ground_plane), int(enable_ground_shadows)
    return d


def self_occlusion_hacks(resolution, virtual_depth, slice_count):
    """ComputeSelfOcclusionHack / ComputeZSelfOcclusionHack, LightingRenderer.GBuffer.cs:62-80 (C# double / float arithmetic)."""
    ratio_bias = max((1.0 / resolution) - 1.0, 0.0)
    so = np.float32(0.5 + (ratio_bias ** 1.5) * 0.05)
    slice_size = np.float32(virtual_depth) / np.float32(slice_count)
    return float(so), float(max(np.float32(slice_size * np.float32(0.525)), np.float32(1.0)))


def _segments_intersect(a0, a1, b0, b1):
    """Proper-or-touching intersection of segments a and b (the role Geometry.LineIntersectPolygon's per-edge test plays in the
    back-face hack of GetFrontFaceMesh3D; Fracture's exact routine is outside the reference tree)."""
    la, lb = (a1[0] - a0[0], a1[1] - a0[1]), (b1[0] - b0[0], b1[1] - b0[1])
    d = la[0] * lb[1] - la[1] * lb[0]
    if d == 0.0:
        return False
    dx, dy = a0[0] - b0[0], a0[1] - b0[1]
    r = (dy * lb[0] - dx * lb[1]) / d
    t = (dy * la[0] - dx * la[1]) / d
    return (0.0 <= r <= 1.0) and (0.0 <= t <= 1.0)


def top_face_mesh(polygon, z_base, height, enable_shadows=True):
    """SimpleHeightVolume.Mesh3D (HeightVolume.cs:106-133): a triangulation of the polygon at z = ZBase + Height with normal +z.
    The reference triangulates with Fracture's Geometry.Triangulate; any triangulation covers the same interior -- this one clips
    ears.  Returns (3 t, 9) float32 rows of HeightVolumeVertex."""
    pts = [(float(np.float32(x)), float(np.float32(y))) for (x, y) in polygon]
    area2 = sum(pts[i][0] * pts[(i + 1) % len(pts)][1] - pts[(i + 1) % len(pts)][0] * pts[i][1] for i in range(len(pts)))
    idx = list(range(len(pts))) if area2 > 0 else list(range(len(pts) - 1, -1, -1))

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])
    tris = []
    guard = 0
    while len(idx) > 3 and guard < 10000:
        guard += 1
        for k in range(len(idx)):
            i0, i1, i2 = idx[k - 1], idx[k], idx[(k + 1) % len(idx)]
            a, b, c = pts[i0], pts[i1], pts[i2]
            if cross(a, b, c) <= 0:
                continue
            if any(cross(a, b, pts[j]) >= 0 and cross(b, c, pts[j]) >= 0 and cross(c, a, pts[j]) >= 0 for j in idx if j not in (i0, i1, i2)):
                continue
            tris.append((i0, i1, i2))
            del idx[k]
            break
        else:
            break
    if len(idx) == 3:
        tris.append(tuple(idx))
    h1, h2 = np.float32(z_base), np.float32(z_base) + np.float32(height)
    rows = [[pts[i][0], pts[i][1], h2, 0.0, 0.0, 1.0, h1, h2, 1.0 if enable_shadows else 0.0] for t in tris for i in t]
    return np.asarray(rows, np.float32).reshape(-1, 9)


def front_face_mesh(polygon, z_base, height, enable_shadows=True):
    """SimpleHeightVolume.GetFrontFaceMesh3D (HeightVolume.cs:135-224): one quad per polygon edge from the top to the base, edges
    whose start or end has polygon below it culled, normals as that function derives them.  (3 t, 9) float32 rows."""
    pts = [(float(np.float32(x)), float(np.float32(y))) for (x, y) in polygon]
    n = len(pts)
    h1, h2 = float(np.float32(z_base)), float(np.float32(z_base) + np.float32(height))
    es = 1.0 if enable_shadows else 0.0

    def hits(p):
        s0, s1 = (p[0], float(np.float32(p[1]) + np.float32(0.1))), (p[0], float(np.float32(p[1]) + np.float32(999.0)))
        return any(_segments_intersect(s0, s1, pts[e], pts[(e + 1) % n]) for e in range(n))

    def left_normal(d):
        # Vector2.PerpendicularLeft() = (y, -x); Vector3.Normalize in float
        v = np.asarray([d[1], -d[0], 0.0], np.float32)
        l = np.sqrt(np.float32(v[0] * v[0] + v[1] * v[1]))
        return [float(v[0] / l), float(v[1] / l), 0.0]
    rows = []
    for j in range(n):
        prior, a, b = pts[(j - 1) % n], pts[j], pts[(j + 1) % n]
        if hits(a) or hits(b):
            continue
        if a[1] == b[1]:
            an = bn = [0.0, 1.0, 0.0]
        else:
            an = [0.0, 0.0, 0.0] if a == prior else left_normal((a[0] - prior[0], a[1] - prior[1]))
            bn = [0.0, 0.0, 0.0] if b == a else left_normal((b[0] - a[0], b[1] - a[1]))
        a_top, a_bot, b_top, b_bot = [a[0], a[1], h2], [a[0], a[1], h1], [b[0], b[1], h2], [b[0], b[1], h1]
        for pos, nrm in ((a_top, an), (b_top, bn), (a_bot, an), (b_top, bn), (b_bot, bn), (a_bot, an)):
            rows.append(pos + nrm + [h1, h2, es])
    return np.asarray(rows, np.float32).reshape(-1, 9)


CYLINDER_NORMAL_FACTOR = 0.9      # LightingRenderer.GBuffer.cs:467-468 (pinned by tests/test_reference_pin.py)


def billboard_vertices(billboards, ground_z=0.0, z_to_y=0.0):
    """The vertex loop of RenderGBufferBillboards (LightingRenderer.GBuffer.cs:411-475) for billboards given as dicts with the
    Billboard struct's fields (Billboard.cs:9-86): screen_bounds ((x1, y1), (x2, y2)) | None, world_bounds ((x, y, z), (x, y, z)) |
    None, world_elevation, world_offset, normal, cylinder_factor, data_scale, static_lighting_only, type, texture_bounds.
    Returns (4 n, 12) float32 rows of BillboardVertex (TL, TR, BR, BL per billboard), in the order given (the caller sorts)."""
    f = np.float32
    rows = []
    for b in billboards:
        sb = b.get("screen_bounds")
        wbv = b.get("world_bounds")
        kind = b.get("type", abi.BILLBOARD_MASK)
        normal1 = [f(c) for c in b.get("normal", (0.0, 1.0, 0.0))]
        normal2 = list(normal1)
        ds = b.get("data_scale")
        dsf = (f(1.0) if ds is None else f(ds), f(-1.0) if b.get("static_lighting_only") else f(1.0))
        sbv = [[f(0), f(0)], [f(0), f(0)]] if sb is None else [[f(sb[0][0]), f(sb[0][1])], [f(sb[1][0]), f(sb[1][1])]]
        if wbv is not None:
            wb = [[f(c) for c in wbv[0]], [f(c) for c in wbv[1]]]
        else:
            base_z = f(ground_z) * f(1)
            x1, x2, y = sbv[0][0], sbv[1][0], sbv[1][1]
            we = b.get("world_elevation")
            h = f(sbv[1][1] - sbv[0][1]) if we is None else f(we)
            z_scale = f(h / f(z_to_y)) if z_to_y > 0 else f(0)
            if kind == abi.BILLBOARD_GBUFFER_DATA:
                base_z = f(0) if we is None else f(we)
                z_scale = f(0)
            wb = [[x1, y, f(base_z + z_scale)], [x2, y, base_z]]
        if sb is None and wbv is not None:
            sbv = [[wb[0][0], f(wb[0][1] - f(wb[0][2] * f(z_to_y)))], [wb[1][0], f(wb[1][1] - f(wb[1][2] * f(z_to_y)))]]
            max_z = max(wb[0][2], wb[1][2])
            wb[0][2] = wb[1][2] = max_z
        off = [f(c) for c in b.get("world_offset", (0.0, 0.0, 0.0))]
        wb = [[f(wb[0][c] + off[c]) for c in range(3)], [f(wb[1][c] + off[c]) for c in range(3)]]
        cf = f(b.get("cylinder_factor", 0.0))
        if abs(cf) >= f(0.001):
            normal1[0] = f(0) - f(f(CYLINDER_NORMAL_FACTOR) * cf)
            normal2[0] = f(0) + f(f(CYLINDER_NORMAL_FACTOR) * cf)
        tb = b.get("texture_bounds", ((0.0, 0.0), (1.0, 1.0)))
        (tl, br) = ((f(tb[0][0]), f(tb[0][1])), (f(tb[1][0]), f(tb[1][1])))
        (sx1, sy1), (sx2, sy2) = sbv
        rows.append([sx1, sy1, tl[0], tl[1], wb[0][0], wb[0][1], wb[0][2], *normal1, *dsf])
        rows.append([sx2, sy1, br[0], tl[1], wb[1][0], wb[0][1], wb[0][2], *normal2, *dsf])
        rows.append([sx2, sy2, br[0], br[1], wb[1][0], wb[1][1], wb[1][2], *normal2, *dsf])
        rows.append([sx1, sy2, tl[0], br[1], wb[0][0], wb[1][1], wb[1][2], *normal1, *dsf])
    return np.asarray(rows, np.float32).reshape(-1, 12)


def obstruction_array(obstacles):
    """[(LightObstructionType 0..4, center xyz, size xyz[, rotation about z in radians[, is_dynamic]])] ->
    ctypes array of abi.Obstruction; Orientation = Quaternion.CreateFromAxisAngle(UnitZ, rotation)
    (LightObstruction.Rotation setter, LightObstruction.cs:94-103)."""
    arr = (abi.Obstruction * len(obstacles))()
    for i, ob in enumerate(obstacles):
        typ, center, size = ob[0], ob[1], ob[2]
        rot = float(ob[3]) if len(ob) > 3 else 0.0
        o = arr[i]
        o.Type = int(typ)
        o.IsDynamic = int(bool(ob[4])) if len(ob) > 4 else 0
        for k in range(3):
            o.Center[k] = float(center[k])
            o.Size[k] = float(size[k])
        half = np.float32(rot) * np.float32(0.5)
        o.Orientation[0] = 0.0
        o.Orientation[1] = 0.0
        o.Orientation[2] = float(np.float32(math.sin(float(half))))
        o.Orientation[3] = float(np.float32(math.cos(float(half))))
    return arr


def height_volume_arrays(volumes):
    """[(polygon [(x, y), ...], z_base, height[, is_dynamic])] -> (ctypes array of abi.HeightVolume, (n, 2) float32 vertices)."""
    arr = (abi.HeightVolume * len(volumes))()
    verts = []
    for i, hv in enumerate(volumes):
        poly = hv[0]
        arr[i].FirstVertex, arr[i].VertexCount = len(verts), len(poly)
        arr[i].ZBase, arr[i].Height = float(hv[1]), float(hv[2])
        arr[i].IsDynamic = int(bool(hv[3])) if len(hv) > 3 else 1     # HeightVolumeBase.IsDynamic defaults to true, HeightVolume.cs:23
        arr[i].TopFaceEnableShadows = int(bool(hv[4])) if len(hv) > 4 else 1   # :18
        verts.extend((float(x), float(y)) for (x, y) in poly)
    return arr, np.asarray(verts, dtype=np.float32).reshape(-1, 2)


def random_obstructions(seed, n, extent, size_lo=12.0, size_hi=70.0, z_hi=64.0, types=(0, 1, 2, 3, 4), rotate=True, dynamic_fraction=0.0):
    """n random LightObstructions of every type with rotations, for obstruction_array()."""
    cx = uniform(seed + 1, (n,), 0, extent[0])
    cy = uniform(seed + 2, (n,), 0, extent[1])
    cz = uniform(seed + 5, (n,), 0, z_hi * 0.5)
    sz = uniform(seed + 3, (n, 3), size_lo, size_hi)
    ty = uniform(seed + 4, (n,))
    ro = uniform(seed + 6, (n,), -math.pi, math.pi)
    dy = uniform(seed + 7, (n,))
    out = []
    for i in range(n):
        t = types[min(int(ty[i] * len(types)), len(types) - 1)]
        out.append((t, (float(cx[i]), float(cy[i]), float(cz[i])), (float(sz[i, 0]), float(sz[i, 1]), float(min(sz[i, 2], z_hi))),
                    float(ro[i]) if rotate else 0.0, bool(dy[i] < dynamic_fraction)))
    return out


def random_obstacles(seed, n, extent, size_lo=12.0, size_hi=70.0, z_hi=64.0):
    cx = uniform(seed + 1, (n,), 0, extent[0])
    cy = uniform(seed + 2, (n,), 0, extent[1])
    sz = uniform(seed + 3, (n, 3), size_lo, size_hi)
    typ = (uniform(seed + 4, (n,)) < 0.5)
    out = []
    for i in range(n):
        h = float(min(sz[i, 2], z_hi))
        out.append((1 if typ[i] else 2, (float(cx[i]), float(cy[i]), 0.0), (float(sz[i, 0]), float(sz[i, 1]), h)))
    return out


def simple_particles_obstacles(width=256.0, height=256.0):
    """4 cylinders + 4 wall boxes in the pattern of TestGame Scenes/SimpleParticles.cs:255-284, scaled to the field."""
    obs = []
    for (fx, fy) in ((0.25, 0.25), (0.75, 0.25), (0.25, 0.75), (0.75, 0.75)):
        obs.append((3, (width * fx, height * fy, 0.0), (width * 0.045, width * 0.045, 48.0)))
    t = 6.0
    obs.append((2, (width / 2, 0.0, 0.0), (width / 2, t, 64.0)))
    obs.append((2, (width / 2, height, 0.0), (width / 2, t, 64.0)))
    obs.append((2, (0.0, height / 2, 0.0), (t, height / 2, 64.0)))
    obs.append((2, (width, height / 2, 0.0), (t, height / 2, 64.0)))
    return obs


# ---- lighting ---------------------------------------------------------------------------------------------

def environment(ground_z=0.0, maximum_z=128.0, z_to_y=0.0, light_occlusion=0.0, render_scale=(1.0, 1.0),
                gbuffer_size=None, viewport_scale=(1.0, 1.0), viewport_position=(0.0, 0.0), viewport_relative=False):
    """ComputeUniforms (LightingRenderer.cs:691-701) + SetGBufferParameters (LightingRenderer.GBuffer.cs:520-534)."""
    e = abi.Environment()
    e.ZAndScale = abi.f4(ground_z, maximum_z, render_scale[0], render_scale[1])
    inv = 0.0 if abs(z_to_y) <= 0.0001 else 1.0 / z_to_y
    e.ZToY = abi.f4(z_to_y, inv, light_occlusion, 0)
    if gbuffer_size is not None:
        e.GBufferTexelSizeAndMisc = abi.f4(1.0 / gbuffer_size[0], 1.0 / gbuffer_size[1], viewport_scale[0], viewport_scale[1])
    else:
        e.GBufferTexelSizeAndMisc = abi.f4(0, 0, viewport_scale[0], viewport_scale[1])
    e.ViewportPosition[0], e.ViewportPosition[1] = viewport_position
    e.GBufferViewportRelative = 1.0 if viewport_relative else 0.0
    return e


def sphere_light(position, radius, ramp_length, color=(1, 1, 1, 1), opacity=1.0, intensity_scale=1.0, ramp_mode=0,
                 casts_shadows=True, have_distance_field=True, ao_radius=0.0, ao_opacity=1.0, falloff_y=1.0,
                 shadow_distance_falloff=None, shadow_filter=-1, specular=(0, 0, 0), specular_power=1.0, ramp_offset=0.0, ramp_rate=1.0):
    """RenderSphereLightSource, LightingRenderer.cs:1193-1219; EvenMore.zw = RampOffsetForGPU, RampRateForGPU (LightSource.cs:97-98)."""
    v = abi.LightVertex()
    v.LightPosition1 = v.LightPosition2 = v.LightPosition3 = abi.f4(position[0], position[1], position[2], 0)
    # color.W *= (lightSource.Opacity * intensityScale): float * float, then float *= float (found by tests/test_reference_uniforms.py: the
    # product of the two factors had been formed in double)
    v.Color1 = abi.f4(color[0], color[1], color[2], np.float32(color[3]) * (np.float32(opacity) * np.float32(intensity_scale)))
    v.Color2 = abi.f4(specular[0], specular[1], specular[2], specular_power)
    v.LightProperties = abi.f4(radius, ramp_length, float(ramp_mode), 1.0 if (casts_shadows and have_distance_field) else 0.0)
    v.MoreLightProperties = abi.f4(ao_radius, -99999.0 if shadow_distance_falloff is None else shadow_distance_falloff,
                                   falloff_y, ao_opacity)
    v.EvenMoreLightProperties = abi.f4(float(shadow_filter), 0, np.float32(-math.pi) + np.float32(ramp_offset),
                                       np.float32(1.0 / (math.pi * 2) * ramp_rate))
    return v


def random_lights(seed, n, width, height, z=(8.0, 64.0), radius=24.0, ramp=(200.0, 550.0), **kw):
    xs = uniform(seed + 1, (n,), 0, width)
    ys = uniform(seed + 2, (n,), 0, height)
    zs = uniform(seed + 3, (n,), z[0], z[1])
    rs = uniform(seed + 4, (n,), ramp[0], ramp[1])
    col = uniform(seed + 5, (n, 3), 0.2, 1.0)
    arr = (abi.LightVertex * n)()
    for i in range(n):
        arr[i] = sphere_light((xs[i], ys[i], zs[i]), radius, rs[i], color=(col[i, 0], col[i, 1], col[i, 2], 1.0), **kw)
    return arr


def ground_plane_gbuffer(width, height, fmt=abi.GBUFFER_FLOAT4):
    """G-buffer of a bare ground plane: texel (0.5, 1.0, 0, 1.0) <=> normal (0,0,1), z = 0
    (GBufferShaderCommon.fxh:21-33 with encodeNormalSpherical, EnvironmentCommon.fxh:34-38)."""
    g = np.empty((height, width, 4), np.float32)
    g[...] = (0.5, 1.0, 0.0, 1.0)
    if fmt == abi.GBUFFER_HALF4:
        return g.astype(np.float16).view(np.uint16)
    return g


def encode_gbuffer(normal, relative_y, z, enable_shadows=True, fullbright=False):
    """encodeGBufferSample, GBufferShaderCommon.fxh:10-35 (numpy, per texel arrays allowed)."""
    nx = np.where(np.abs(normal[..., 0]) < 0.0001, 0.0001, normal[..., 0])
    ex = (np.arctan2(normal[..., 1], nx) / np.pi + 1.0) * 0.5
    ey = (normal[..., 2] + 1.0) * 0.5
    sign = 1.0 if enable_shadows else -1.0
    w = ((z + 1024.0) / 1024.0) * sign + (0.0 if enable_shadows else -1.0)
    if fullbright:
        w = np.full_like(ex, 99999.0)
    return np.stack([ex, ey, np.broadcast_to(relative_y, ex.shape), np.broadcast_to(w, ex.shape)], axis=-1).astype(np.float32)
