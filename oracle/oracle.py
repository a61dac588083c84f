"""ctypes front end of the CPU oracle (oracle/libilm_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by illuminant_amd/.  PARITY UNPINNED (see
oracle/ilm_oracle.h and DESIGN.md).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from illuminant_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libilm_oracle.so")
_lib = None


def build(force=False):
    """Compile the C restatement with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
            for f in ("ilm_oracle.c", "ilm_oracle_fields.c", "ilm_oracle_gbuffer.c", "ilm_oracle_transforms.c", "ilm_oracle_lights.c", "ilm_oracle_output.c", "ilm_oracle_census.c", "ilm_oracle.h")):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


class Texture(C.Structure):
    _fields_ = [("texels", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32), ("format", C.c_int32)]


class DistanceFieldLayout(C.Structure):
    _fields_ = [("virtual_width", C.c_int32), ("virtual_height", C.c_int32), ("virtual_depth", C.c_float),
                ("resolution", C.c_double),
                ("slice_width", C.c_int32), ("slice_height", C.c_int32), ("slice_count", C.c_int32),
                ("physical_slice_count", C.c_int32), ("column_count", C.c_int32), ("row_count", C.c_int32),
                ("atlas_width", C.c_int32), ("atlas_height", C.c_int32), ("maximum_encoded_distance", C.c_int32)]


class SpawnerState(C.Structure):
    _fields_ = [("rate_error", C.c_double), ("total_spawned", C.c_int32)]


def usable_cpus():
    """Cores this process may actually run on: the affinity mask capped by the cgroup CPU quota.

    A GPU box can show hundreds of logical cores while the container is limited to a few; an OpenMP team sized
    by the former spends its time in oversubscribed barriers."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def lib():
    global _lib
    if _lib is None:
        build()
        os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_set_num_threads(C.c_int32(int(os.environ.get("ILM_ORACLE_THREADS", usable_cpus()))))
        _lib.orc_bezier1.restype = C.c_float
        _lib.orc_bezier1.argtypes = [C.c_void_p, C.c_float]
        _lib.orc_sample_distance_field.restype = C.c_float
        _lib.orc_encode_distance.restype = C.c_float
        _lib.orc_encode_distance.argtypes = [C.c_float, C.c_float]
        _lib.orc_decode_distance.restype = C.c_float
        _lib.orc_decode_distance.argtypes = [C.c_float, C.c_float]
        _lib.orc_evaluate_area.restype = C.c_float
        _lib.orc_count_live.restype = C.c_uint32
        _lib.orc_spawner_begin_tick.restype = C.c_int32
        _lib.orc_spawner_begin_tick.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int32, C.c_double, C.c_double, C.c_int32]
        _lib.orc_distance_field_layout.argtypes = [C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_double, C.c_int32, C.c_void_p]
        _lib.orc_distance_field_uniforms.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_int32,
                                                     C.c_float, C.c_float, C.c_void_p]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f4(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"] and a.shape[-1] == 4
    return _p(a)


def make_texture(arr, fmt):
    """arr: (H, W, 4) uint16 (SDF / half G-buffer) or float32 (G-buffer)."""
    assert arr.flags["C_CONTIGUOUS"] and arr.ndim == 3 and arr.shape[2] == 4
    t = Texture(arr.ctypes.data, arr.shape[1], arr.shape[0], fmt)
    t._keep = arr
    return t


def num_threads():
    return int(lib().orc_num_threads())


# ---- particles: planes are (slots, 4) float32 arrays, updated in place ---------------------------

def spawn(pos, vel, attr, chunk_size, rnd, p):
    lib().orc_spawn(_f4(pos), _f4(vel), _f4(attr), chunk_size, _f4(rnd), rnd.shape[1], rnd.shape[0], C.byref(p))


def gravity(pos, vel, chunk_size, sys, p):
    lib().orc_gravity(_f4(pos), _f4(vel), chunk_size, C.byref(sys), C.byref(p))


def noise(pos, vel, chunk_size, rnd, sys, p):
    lib().orc_noise(_f4(pos), _f4(vel), chunk_size, _f4(rnd), rnd.shape[1], rnd.shape[0], C.byref(sys), C.byref(p))


def fma(pos, vel, chunk_size, sys, p):
    lib().orc_fma(_f4(pos), _f4(vel), chunk_size, C.byref(sys), C.byref(p))


def update(pos, vel, attr, rc, rd, chunk_size, sys, p, life_ramp=None, df=None, sdf=None):
    rw = rh = 0
    if life_ramp is not None:
        rh, rw = life_ramp.shape[0], life_ramp.shape[1]
    lib().orc_update(_f4(pos), _f4(vel), _f4(attr), _f4(rc), _f4(rd), chunk_size, C.byref(sys), C.byref(p),
                     _p(life_ramp), rw, rh,
                     C.byref(df) if df is not None else None, C.byref(sdf) if sdf is not None else None)


def matrix_multiply(pos, vel, chunk_size, sys, p):
    lib().orc_matrix_multiply(_f4(pos), _f4(vel), chunk_size, C.byref(sys), C.byref(p))


def low_precision_randomness(rnd):
    """The Rgba64 copy of the randomness table (ParticleEngine.cs:536-538): (H, W, 4) uint16."""
    out = np.empty(rnd.shape, dtype=np.uint16)
    lib().orc_low_precision_randomness(_f4(rnd), C.c_int32(rnd.shape[0] * rnd.shape[1]), _p(out))
    return out


def spatial_noise(pos, vel, chunk_size, rnd, sys, p):
    lp = low_precision_randomness(rnd)
    lib().orc_spatial_noise(_f4(pos), _f4(vel), chunk_size, _p(lp), rnd.shape[1], rnd.shape[0], C.byref(sys), C.byref(p))


class StepExtras(C.Structure):
    _fields_ = [("spawn_positions", C.c_void_p * abi.MAX_SPAWNS), ("spawn_position_count", C.c_int32 * abi.MAX_SPAWNS),
                ("source_pos", C.c_void_p * abi.MAX_SPAWNS), ("source_vel", C.c_void_p * abi.MAX_SPAWNS),
                ("source_attr", C.c_void_p * abi.MAX_SPAWNS), ("low_precision_rnd", C.c_void_p),
                ("spawn_pattern", C.c_void_p * abi.MAX_SPAWNS), ("pattern_w", C.c_int32 * abi.MAX_SPAWNS),
                ("pattern_h", C.c_int32 * abi.MAX_SPAWNS), ("pattern_levels", C.c_int32 * abi.MAX_SPAWNS)]


def erase(pos, vel, rc, rd, chunk_size):
    lib().orc_erase(_f4(pos), _f4(vel), _f4(rc), _f4(rd), chunk_size)


def reference_constants():
    """{reference key: value} of every number the restatement takes from the reference's text (ilm_oracle_constants.h)."""
    l = lib()
    l.orc_reference_constant.argtypes = [C.c_char_p, C.POINTER(C.c_double)]
    l.orc_reference_constant_key.restype = C.c_char_p
    out = {}
    for i in range(l.orc_reference_constant_count()):
        key = l.orc_reference_constant_key(i)
        v = C.c_double()
        assert l.orc_reference_constant(key, C.byref(v)) == 1
        out[key.decode()] = float(v.value)
    return out


def count_live(pos, saturate16=False):
    return int(lib().orc_count_live(_f4(pos), pos.shape[0], 1 if saturate16 else 0))


def step(chunks, chunk_size, rnd, desc, life_ramp=None, sdf=None, want_counts=False, spawn_positions=None, feedback_sources=None,
         spawn_patterns=None):
    """chunks: list of dicts/tuples of 5 planes (pos, vel, attr, rc, rd) per chunk.
    spawn_positions: {spawn slot: (n, 4) float32} for ILM_SPAWN_POSITION_BUFFER records;
    feedback_sources: {spawn slot: (pos, vel, attr) planes of the source chunk} for ILM_SPAWN_FEEDBACK records;
    spawn_patterns: {spawn slot: [level 0 (h, w, 4) float32, level 1, ...]} for ILM_SPAWN_PATTERN records."""
    n = len(chunks)
    ptrs = (C.c_void_p * (n * 5))()
    for c, planes in enumerate(chunks):
        for k in range(5):
            ptrs[c * 5 + k] = _f4(planes[k]).value
    counts = np.zeros(n, dtype=np.uint32) if want_counts else None
    rw = rh = 0
    if life_ramp is not None:
        rh, rw = life_ramp.shape[0], life_ramp.shape[1]
    ex = StepExtras()
    keep = []
    for slot, a in (spawn_positions or {}).items():
        a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 4)
        keep.append(a)
        ex.spawn_positions[slot] = a.ctypes.data
        ex.spawn_position_count[slot] = a.shape[0]
    for slot, levels in (spawn_patterns or {}).items():
        flat = np.ascontiguousarray(np.concatenate([np.asarray(l, dtype=np.float32).reshape(-1, 4) for l in levels]))
        keep.append(flat)
        ex.spawn_pattern[slot] = flat.ctypes.data
        ex.pattern_h[slot], ex.pattern_w[slot], ex.pattern_levels[slot] = levels[0].shape[0], levels[0].shape[1], len(levels)
    for slot, (sp, sv, sa) in (feedback_sources or {}).items():
        ex.source_pos[slot], ex.source_vel[slot], ex.source_attr[slot] = _f4(sp).value, _f4(sv).value, _f4(sa).value
    lib().orc_step_ex(ptrs, n, chunk_size, _f4(rnd), rnd.shape[1], rnd.shape[0], _p(life_ramp), rw, rh,
                      C.byref(sdf) if sdf is not None else None, C.byref(desc), _p(counts), C.byref(ex))
    return counts


# ---- primitives ------------------------------------------------------------------------------------

def bezier1(b, value):
    return float(lib().orc_bezier1(C.byref(b), C.c_float(value)))


def bezier4(b, value):
    out = abi.Float4()
    lib().orc_bezier4(C.byref(b), C.c_float(value), C.byref(out))
    return out.tuple()


def sample_distance_field(pos, df, sdf):
    a = (C.c_float * 3)(*[float(x) for x in pos])
    return float(lib().orc_sample_distance_field(a, C.byref(df), C.byref(sdf)))


def encode_distance(d, max_encoded):
    return float(lib().orc_encode_distance(d, max_encoded))


def decode_distance(e, max_encoded):
    return float(lib().orc_decode_distance(e, max_encoded))


def evaluate_area(type_id, pos, center, size, rotation):
    mk = lambda v: (C.c_float * 3)(*[float(x) for x in v])
    return float(lib().orc_evaluate_area(C.c_int32(type_id), mk(pos), mk(center), mk(size), C.c_float(rotation)))


# ---- lighting --------------------------------------------------------------------------------------

def sample_gbuffer(px, py, env, gbuffer=None):
    wp = (C.c_float * 3)(); n = (C.c_float * 3)(); cam = (C.c_float * 3)()
    es = C.c_int32(); fb = C.c_int32()
    lib().orc_sample_gbuffer(C.c_float(px), C.c_float(py), C.byref(env),
                             C.byref(gbuffer) if gbuffer is not None else None,
                             wp, n, C.byref(es), C.byref(fb), cam)
    return tuple(wp), tuple(n), bool(es.value), bool(fb.value), tuple(cam)


_ramp_keepalive = [None]


def set_light_ramp(texels):
    """orc_set_light_ramp: (h, w, 4) float32 RampTexture of the light group rendered by the following render_sphere_lights /
    render_light_probes calls; None unbinds."""
    # a 1 x 1 ramp is no ramp at all: GetLightRenderState, LightingRenderer.cs:822-827 (host logic, restated here)
    if texels is None or (np.asarray(texels).shape[0] == 1 and np.asarray(texels).shape[1] == 1):
        lib().orc_set_light_ramp(None, C.c_int32(0), C.c_int32(0))
        _ramp_keepalive[0] = None
        return
    a = np.ascontiguousarray(texels, dtype=np.float32)
    _ramp_keepalive[0] = a
    lib().orc_set_light_ramp(_f4(a), C.c_int32(a.shape[1]), C.c_int32(a.shape[0]))


def render_sphere_lights(lights, env, df, gbuffer, sdf, ambient, width, height, row_begin=0, row_end=None, want_stats=False, blend_fp16=False):
    """lights: ctypes array of abi.LightVertex.  Returns (lightmap (H, W, 4) float32, stats|None).
    blend_fp16: the reference's HalfVector4 render target, rounded after every light (orc_set_lightmap_blend)."""
    if row_end is None:
        row_end = height
    lib().orc_set_lightmap_blend(C.c_int32(1 if blend_fp16 else 0))
    out = np.zeros((height, width, 4), dtype=np.float32)
    amb = (C.c_float * 4)(*[float(x) for x in ambient])
    stats = abi.RenderStats() if want_stats else None
    lib().orc_render_sphere_lights(lights, len(lights), C.byref(env), C.byref(df),
                                   C.byref(gbuffer) if gbuffer is not None else None,
                                   C.byref(sdf) if sdf is not None else None,
                                   amb, _f4(out), width, height, row_begin, row_end,
                                   C.byref(stats) if stats is not None else None)
    lib().orc_set_lightmap_blend(C.c_int32(0))
    return out, stats


class OpenRayCensus(C.Structure):
    """OrcOpenRayCensus, oracle/ilm_oracle_census.c"""
    _fields_ = [(n, C.c_uint64) for n in ("traced_pairs", "traced_samples", "result_one_pairs", "result_one_samples", "strict_pairs", "strict_samples",
                                          "loose_pairs", "loose_samples", "violations", "wave_count", "wave_open", "wave_iterations", "wave_iterations_left",
                                          "wave_open_samples", "dda_bricks")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def open_ray_census(lights, env, df, gbuffer, sdf, width, height, row_begin=0, row_end=None, brick_texels=8, census=None):
    """orc_open_ray_census: marches every traced pair of rows [row_begin, row_end) like render_sphere_lights and counts the rays a table of
    per-brick minimum distances PROVES open (analysis only: tools/open_ray_census.py).  Adds to `census` (a new one when None)."""
    if row_end is None:
        row_end = height
    census = census if census is not None else OpenRayCensus()
    n = len(lights) if lights is not None else 0
    lib().orc_open_ray_census(C.cast(lights, C.c_void_p) if n else None, n, C.byref(env), C.byref(df),
                              C.byref(gbuffer) if gbuffer is not None else None, C.byref(sdf), width, height, row_begin, row_end, brick_texels,
                              C.byref(census))
    return census


def render_distance_field_slices(atlas, fmt, desc, first_virtual_slices, obstructions=None, volumes=None, polygon_xy=None,
                                 clear_source=None):
    """orc_render_distance_field_slices: renders the listed slice triplets into `atlas` ((H, W, 4) uint16, in place)."""
    assert atlas.dtype == np.uint16 and atlas.flags["C_CONTIGUOUS"] and atlas.shape[2] == 4
    sl = np.ascontiguousarray(first_virtual_slices, dtype=np.int32)
    no = len(obstructions) if obstructions is not None else 0
    nv = len(volumes) if volumes is not None else 0
    poly = np.ascontiguousarray(polygon_xy, dtype=np.float32).reshape(-1, 2) if polygon_xy is not None else np.zeros((0, 2), np.float32)
    if clear_source is not None:
        assert clear_source.dtype == np.uint16 and clear_source.shape == atlas.shape and clear_source.flags["C_CONTIGUOUS"]
    lib().orc_render_distance_field_slices(_p(atlas), C.c_int32(fmt), _p(clear_source), C.byref(desc), _p(sl), C.c_int32(sl.shape[0]),
                                           obstructions if no else None, C.c_int32(no), volumes if nv else None, C.c_int32(nv),
                                           _p(poly) if poly.shape[0] else None, C.c_int32(poly.shape[0]))
    return atlas


def render_particle_lights(chunks, quad_counts, params, env, df, gbuffer, sdf, lightmap, row_begin=0, row_end=None, want_stats=False):
    """orc_render_particle_lights: chunks = list of 5 planes per chunk; lightmap (H, W, 4) float32 is accumulated into, in place."""
    n = len(chunks)
    ptrs = (C.c_void_p * (n * 5))()
    for c, planes in enumerate(chunks):
        for k in range(5):
            ptrs[c * 5 + k] = _f4(planes[k]).value
    q = np.ascontiguousarray(quad_counts, dtype=np.int32)
    h, w = lightmap.shape[0], lightmap.shape[1]
    if row_end is None:
        row_end = h
    stats = abi.RenderStats() if want_stats else None
    lib().orc_render_particle_lights(ptrs, C.c_int32(n), _p(q), C.byref(params), C.byref(env), C.byref(df),
                                     C.byref(gbuffer) if gbuffer is not None else None, C.byref(sdf) if sdf is not None else None,
                                     _f4(lightmap), C.c_int32(w), C.c_int32(h), C.c_int32(row_begin), C.c_int32(row_end),
                                     C.byref(stats) if stats is not None else None)
    return stats


def render_light_probes(lights, probe_positions, probe_normals, env, df, sdf):
    pp = np.ascontiguousarray(probe_positions, dtype=np.float32).reshape(-1, 4)
    pn = np.ascontiguousarray(probe_normals, dtype=np.float32).reshape(-1, 4)
    out = np.zeros_like(pp)
    lib().orc_render_light_probes(lights, C.c_int32(len(lights) if lights is not None else 0), _f4(pp), _f4(pn), C.c_int32(pp.shape[0]),
                                  C.byref(env), C.byref(df), C.byref(sdf) if sdf is not None else None, _f4(out))
    return out


def fill_readback_result(chunks, params, element_counts=None, capacity=None):
    """orc_fill_readback_result: (ctypes array of abi.ReadbackDrawCall, total)."""
    n = len(chunks)
    slots = chunks[0][0].shape[0]
    ptrs = (C.c_void_p * (n * 5))()
    for c, planes in enumerate(chunks):
        for k in range(5):
            ptrs[c * 5 + k] = _f4(planes[k]).value
    e = np.ascontiguousarray(element_counts, dtype=np.int32) if element_counts is not None else None
    if capacity is None:
        capacity = n * slots
    out = (abi.ReadbackDrawCall * capacity)()
    lib().orc_fill_readback_result.restype = C.c_int32
    total = lib().orc_fill_readback_result(ptrs, C.c_int32(n), _p(e), C.c_int32(slots), C.byref(params), out, C.c_int32(capacity))
    return out, int(total)


def render_particles(chunks, params, width, height, quad_counts=None, image=None, bitmap=None):
    """orc_render_particles(_textured): (image (h, w, 4) float32 blended in place / created black, (live quads, shaded pixels)).
    bitmap: (h, w, 4) float32 sprite sheet for params.BitmapFilter != BITMAP_NONE."""
    n = len(chunks)
    slots = chunks[0][0].shape[0]
    ptrs = (C.c_void_p * (n * 5))()
    for c, planes in enumerate(chunks):
        for k in range(5):
            ptrs[c * 5 + k] = _f4(planes[k]).value
    q = np.ascontiguousarray(quad_counts, dtype=np.int32) if quad_counts is not None else None
    if image is None:
        image = np.zeros((height, width, 4), np.float32)
    stats = (C.c_uint64 * 2)()
    bm = np.ascontiguousarray(bitmap, dtype=np.float32) if bitmap is not None else None
    lib().orc_render_particles_textured(ptrs, C.c_int32(n), _p(q), C.c_int32(slots), C.byref(params), _p(bm),
                                        C.c_int32(bm.shape[1] if bm is not None else 0), C.c_int32(bm.shape[0] if bm is not None else 0),
                                        _f4(image), C.c_int32(width), C.c_int32(height), stats)
    return image, (int(stats[0]), int(stats[1]))


def resolve_lighting(lightmap, hdr, row_begin=0, row_end=None, albedo=None):
    """albedo: (h, w, 4) float32 texture for the ...WithAlbedo techniques (None: the plain resolve)."""
    h, w = lightmap.shape[0], lightmap.shape[1]
    out = np.zeros_like(lightmap)
    if albedo is None:
        lib().orc_resolve_lighting(_f4(lightmap), C.c_int32(w), C.c_int32(h), C.byref(hdr), _f4(out), C.c_int32(row_begin),
                                   C.c_int32(h if row_end is None else row_end))
    else:
        albedo = np.ascontiguousarray(albedo, dtype=np.float32)
        assert albedo.shape == lightmap.shape
        lib().orc_resolve_lighting_with_albedo(_f4(lightmap), _f4(albedo), C.c_int32(w), C.c_int32(h), C.byref(hdr), _f4(out), C.c_int32(row_begin),
                                               C.c_int32(h if row_end is None else row_end))
    return out


def render_gbuffer(width, height, desc, volumes=None, polygon_xy=None):
    """orc_render_gbuffer: (H, W, 4) float32; volumes are sorted by top height here as RenderGBufferVolumes does."""
    out = np.zeros((height, width, 4), np.float32)
    nv = len(volumes) if volumes is not None else 0
    order = sorted(range(nv), key=lambda i: np.float32(volumes[i].ZBase) + np.float32(volumes[i].Height))
    sorted_vols = (abi.HeightVolume * max(nv, 1))()
    for k, i in enumerate(order):
        C.memmove(C.byref(sorted_vols[k]), C.byref(volumes[i]), C.sizeof(abi.HeightVolume))
    poly = np.ascontiguousarray(polygon_xy, dtype=np.float32).reshape(-1, 2) if polygon_xy is not None else np.zeros((1, 2), np.float32)
    lib().orc_render_gbuffer(_f4(out), C.c_int32(width), C.c_int32(height), C.byref(desc), sorted_vols, C.c_int32(nv), _p(poly))
    return out


def render_gbuffer_meshes(width, height, desc, top=None, front=None, billboards=None, runs=(), textures=()):
    """orc_render_gbuffer_meshes: (H, W, 4) float32.  top / front: (n, 9) float32 rows of HeightVolumeVertex; billboards: (4 q, 12)
    float32 rows of BillboardVertex; runs: (first_quad, quad_count, type) per run; textures: one (h, w, 4) array (uint8 = Color,
    float16, float32) or None per run."""
    out = np.zeros((height, width, 4), np.float32)

    def rows(a, n):
        return np.ascontiguousarray(a, dtype=np.float32).reshape(-1, n) if a is not None else np.zeros((0, n), np.float32)
    top, front, bb = rows(top, 9), rows(front, 9), rows(billboards, 12)
    nr = len(runs)
    c_runs = (abi.BillboardRun * max(nr, 1))()
    c_tex = (Texture * max(nr, 1))()
    keep = []
    for r, (first, count, kind) in enumerate(runs):
        c_runs[r].FirstQuad, c_runs[r].QuadCount, c_runs[r].Type = first, count, kind
        t = textures[r] if r < len(textures) else None
        if t is not None:
            t = np.ascontiguousarray(t)
            keep.append(t)
            fmt = {np.dtype(np.uint8): abi.LIGHTMAP_RGBA8, np.dtype(np.float16): abi.LIGHTMAP_HALF4, np.dtype(np.float32): abi.LIGHTMAP_FLOAT4}[t.dtype]
            c_tex[r] = Texture(t.ctypes.data, t.shape[1], t.shape[0], fmt)
    lib().orc_render_gbuffer_meshes(_f4(out), C.c_int32(width), C.c_int32(height), C.byref(desc),
                                    _p(top) if len(top) else None, C.c_int32(len(top)), _p(front) if len(front) else None, C.c_int32(len(front)),
                                    _p(bb) if len(bb) else None, C.c_int32(len(bb)), c_runs, C.c_int32(nr), c_tex)
    return out


# ---- host logic -------------------------------------------------------------------------------------

def distance_field_layout(vw, vh, vdepth, requested_slices, resolution=1.0, max_encoded=128):
    out = DistanceFieldLayout()
    lib().orc_distance_field_layout(vw, vh, vdepth, requested_slices, resolution, max_encoded, C.byref(out))
    return out


def distance_field_uniforms(layout, valid_slice_count=None, z_offset=0.0, max_cone_radius=24.0, power=1.0,
                            step_limit=64, min_step_size=3.0, long_step_factor=1.0):
    out = abi.DistanceFieldUniforms()
    if valid_slice_count is None:
        valid_slice_count = layout.slice_count
    lib().orc_distance_field_uniforms(C.byref(layout), valid_slice_count, z_offset, max_cone_radius, power,
                                      step_limit, min_step_size, long_step_factor, C.byref(out))
    return out


def spawner_begin_tick(state, min_rate, max_rate, count_scale, rng_draw, dt, maximum_total=-1):
    return int(lib().orc_spawner_begin_tick(C.byref(state), min_rate, max_rate, count_scale, rng_draw, dt, maximum_total))


def spawner_end_tick(state, requested, actual):
    lib().orc_spawner_end_tick(C.byref(state), requested, actual)
