"""ctypes binding of libilluminant_hip.so (include/illuminant_hip.h).

This is the only way Python reaches the kernels: through the C ABI, exactly as
the C# P/Invoke layer of INTEGRATION.md would.  There is no CPU fallback: if
the library is missing or no GPU is visible, calls raise.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# ILM_HIP_LIB: a variant build of the same library for on-box A/B runs (tools/ab_build.sh).  Scripts that also drive the host mirror
# (bench.py) point LD_LIBRARY_PATH at the same directory: mirror and binding must share ONE copy (two copies = two handle registries)
LIB_PATH = os.environ.get("ILM_HIP_LIB") or os.path.join(_HERE, "lib", "libilluminant_hip.so")

_lib = None


class IlluminantError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("illuminant_hip error %d: %s" % (code, message))
        self.code = code


# every symbol include/illuminant_hip.h declares: name -> (restype, argtypes)
_H = abi.Handle
_P = C.c_void_p
_I = C.c_int32
SYMBOLS = {
    "ilm_abi_version": (_I, []),
    "ilm_last_error": (C.c_char_p, []),
    "ilm_device_count": (_I, []),
    "ilm_debug_reference_constant": (_I, [C.c_char_p, C.POINTER(C.c_double)]),
    "ilm_debug_reference_constant_count": (_I, []),
    "ilm_debug_reference_constant_key": (C.c_char_p, [_I]),
    "ilm_ctx_create": (_I, [_I, C.POINTER(_H)]),
    "ilm_ctx_create_sibling": (_I, [_H, C.POINTER(_H)]),
    "ilm_ctx_destroy": (_I, [_H]),
    "ilm_ctx_sync": (_I, [_H]),
    "ilm_ctx_stream": (_I, [_H, C.POINTER(_P)]),
    "ilm_timer_start": (_I, [_H]),
    "ilm_timer_stop": (_I, [_H, C.POINTER(C.c_float)]),
    "ilm_engine_create": (_I, [_H, _I, _P, _I, _I, C.POINTER(_H)]),
    "ilm_engine_destroy": (_I, [_H]),
    "ilm_system_create": (_I, [_H, C.POINTER(_H)]),
    "ilm_system_destroy": (_I, [_H]),
    "ilm_system_add_chunk": (_I, [_H, C.POINTER(_I)]),
    "ilm_system_remove_chunk": (_I, [_H, _I]),
    "ilm_system_chunk_count": (_I, [_H, C.POINTER(_I)]),
    "ilm_chunk_upload": (_I, [_H, _I, _I, _P, _I, _I]),
    "ilm_chunk_download": (_I, [_H, _I, _I, _P, _I, _I]),
    "ilm_chunk_device_ptr": (_I, [_H, _I, _I, C.POINTER(_P), C.POINTER(C.c_int64)]),
    "ilm_system_set_distance_field": (_I, [_H, _H]),
    "ilm_system_set_life_ramp": (_I, [_H, _P, _I, _I]),
    "ilm_system_step": (_I, [_H, _P]),
    "ilm_spawn": (_I, [_H, _I, _P, _P]),
    "ilm_gravity": (_I, [_H, _I, _P, _P]),
    "ilm_noise": (_I, [_H, _I, _P, _P]),
    "ilm_fma": (_I, [_H, _I, _P, _P]),
    "ilm_matrix_multiply": (_I, [_H, _I, _P, _P]),
    "ilm_spatial_noise": (_I, [_H, _I, _P, _P]),
    "ilm_system_set_spawn_positions": (_I, [_H, _I, _P, _I]),
    "ilm_system_set_spawn_pattern": (_I, [_H, _I, _P, _I, _I, _I]),
    "ilm_update": (_I, [_H, _I, _P, _P, _P]),
    "ilm_erase": (_I, [_H, _I]),
    "ilm_system_live_counts": (_I, [_H, _P, _I, _I]),
    "ilm_system_step_counts": (_I, [_H, _P, _I, _I]),
    "ilm_system_poll_counts": (_I, [_H, _P, _I, _I, C.POINTER(_I)]),
    "ilm_chunk_live_slots": (_I, [_H, _I, _P, _I, C.POINTER(_I)]),
    "ilm_sdf_create": (_I, [_H, _I, _I, _I, C.POINTER(_H)]),
    "ilm_sdf_upload": (_I, [_H, _P]),
    "ilm_sdf_sample": (_I, [_H, _P, _P, _I, _P]),
    "ilm_debug_sdf_sample_inside": (_I, [_H, _P, _P, _I, _P, _P]),
    "ilm_debug_divide": (_I, [_H, _P, _P, _I, _P, _P]),
    "ilm_debug_step_interpreter": (_I, [_I]),
    "ilm_debug_step_streams": (_I, [_I]),
    "ilm_debug_step_sdf_samples": (_I, [_H, _I, C.POINTER(C.c_uint64)]),
    "ilm_debug_last_light_launch": (_I, [_H, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ilm_debug_divide_by_constants": (_I, [_H, _P, _P, _I, C.POINTER(_I)]),
    "ilm_sdf_destroy": (_I, [_H]),
    "ilm_sdf_download": (_I, [_H, _P]),
    "ilm_sdf_device_ptr": (_I, [_H, C.POINTER(_P)]),
    "ilm_sdf_mark_dirty": (_I, [_H, _I, _I]),
    "ilm_sdf_trace_info": (_I, [_H, C.POINTER(abi.SdfTraceInfo)]),
    "ilm_sdf_render_slices": (_I, [_H, _H, _P, _P, _I, _P, _I, _P, _I, _P, _I]),
    "ilm_gbuffer_create": (_I, [_H, _I, _I, _I, C.POINTER(_H)]),
    "ilm_gbuffer_upload": (_I, [_H, _P]),
    "ilm_gbuffer_destroy": (_I, [_H]),
    "ilm_gbuffer_download": (_I, [_H, _P]),
    "ilm_gbuffer_render": (_I, [_H, _P, _P, _I, _P, _I]),
    "ilm_gbuffer_render_meshes": (_I, [_H, _P, _P, _I, _P, _I, _P, _I, _P, _I]),
    "ilm_lightmap_create": (_I, [_H, _I, _I, _I, _P, C.POINTER(_H)]),
    "ilm_lightmap_download": (_I, [_H, _P, _I, _I]),
    "ilm_lightmap_device_ptr": (_I, [_H, C.POINTER(_P)]),
    "ilm_lightmap_destroy": (_I, [_H]),
    "ilm_render_sphere_lights": (_I, [_H, _P, _I, _P, _P, _H, _H, _P, _H, _I, _I, _P]),
    "ilm_render_particle_lights": (_I, [_H, _H, _P, _I, _P, _P, _P, _H, _H, _H, _I, _I, _P]),
    "ilm_render_light_probes": (_I, [_H, _P, _I, _P, _P, _I, _P, _P, _H, _P]),
    "ilm_system_readback": (_I, [_H, _P, _I, _P, _P, _I, C.POINTER(_I)]),
    "ilm_system_readback_view": (_I, [_H, _P, _I, _P, C.POINTER(_P), C.POINTER(_I)]),
    "ilm_render_particles": (_I, [_H, _P, _I, _P, _H, _P]),
    "ilm_lightmap_clear": (_I, [_H, _P]),
    "ilm_ctx_set_light_ramp": (_I, [_H, _P, _I, _I]),
    "ilm_ctx_set_lightmap_blend": (_I, [_H, _I]),
    "ilm_ctx_set_light_split": (_I, [_H, _I]),
    "ilm_system_set_bitmap": (_I, [_H, _P, _I, _I]),
    "ilm_resolve_lighting": (_I, [_H, _H, _P, _I, _I]),
    "ilm_resolve_lighting_with_albedo": (_I, [_H, _H, _H, _P, _I, _I]),
    "ilm_lightmap_upload": (_I, [_H, _P, _I, _I]),
    "ilm_group_create": (_I, [_P, _I, C.POINTER(_H)]),
    "ilm_group_unique_id": (_I, [_P]),
    "ilm_group_create_rank": (_I, [_I, _I, _I, _P, C.POINTER(_H)]),
    "ilm_group_destroy": (_I, [_H]),
    "ilm_group_info": (_I, [_H, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "ilm_group_ctx": (_I, [_H, _I, C.POINTER(_H)]),
    "ilm_group_sync": (_I, [_H]),
    "ilm_group_all_gather": (_I, [_H, _P, C.c_uint64, _I]),
    "ilm_group_host_all_gather": (_I, [_H, _P, _P, C.c_uint32]),
    "ilm_group_lightmap_create": (_I, [_H, _I, _I, _I, C.POINTER(_H)]),
    "ilm_group_lightmap_member": (_I, [_H, _I, C.POINTER(_H)]),
    "ilm_group_lightmap_strip": (_I, [_H, _I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "ilm_group_lightmap_set_strips": (_I, [_H, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ilm_group_lightmap_gather": (_I, [_H, _I]),
    "ilm_group_lightmap_store_mode": (_I, [_H, _I]),
    "ilm_group_lightmap_wait": (_I, [_H]),
    "ilm_group_lightmap_destroy": (_I, [_H]),
    "ilm_group_render_sphere_lights": (_I, [_H, _P, _I, _P, _P, _P, _P, _P, _H, _I, _P]),
    "ilm_group_live_counts": (_I, [_H, _P, _I, _P, _I, _I]),
    "ilm_group_gather_chunks": (_I, [_H, _P, _P, _I, _I, _I, _I]),
}


def lib():
    """Load the shared library (fails loudly when it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IlluminantError(abi.ERR_NO_DEVICE, "%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                  "(there is no CPU fallback)" % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(code):
    if code != 0:
        raise IlluminantError(code, lib().ilm_last_error().decode("utf-8", "replace"))


def reference_constants():
    """{reference key: value} of every number the kernels take from the reference's text (csrc/reference_constants.hpp); no GPU needed."""
    out = {}
    for i in range(lib().ilm_debug_reference_constant_count()):
        key = lib().ilm_debug_reference_constant_key(i)
        v = C.c_double()
        check(lib().ilm_debug_reference_constant(key, C.byref(v)))
        out[key.decode()] = float(v.value)
    return out


def device_count():
    return int(lib().ilm_device_count())


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _byref(s):
    return C.cast(C.byref(s), C.c_void_p) if s is not None else None


class Context:
    """One GPU + one HIP stream (ilm_ctx_*)."""

    def __init__(self, device_id=0, borrowed_handle=None):
        """borrowed_handle: a context owned by someone else (a group member, ilm_group_ctx): close() leaves it alone."""
        self.owned = borrowed_handle is None
        if borrowed_handle is not None:
            self.handle = abi.Handle(int(borrowed_handle))
        else:
            self.handle = abi.Handle(0)
            check(lib().ilm_ctx_create(device_id, C.byref(self.handle)))
        self.device_id = device_id

    def sync(self):
        check(lib().ilm_ctx_sync(self.handle))

    def sibling(self):
        """ilm_ctx_create_sibling: another context on this device for a further frame in flight (own stream, scratch, lightmaps); its light
        passes may read this context's distance fields and G-buffers."""
        c = Context.__new__(Context)
        c.owned, c.device_id = True, self.device_id
        c.handle = abi.Handle(0)
        check(lib().ilm_ctx_create_sibling(self.handle, C.byref(c.handle)))
        return c

    def stream(self):
        p = C.c_void_p()
        check(lib().ilm_ctx_stream(self.handle, C.byref(p)))
        return p.value

    def set_light_ramp(self, texels):
        """ilm_ctx_set_light_ramp: (h, w, 4) float32 RampTexture of the light group rendered next; None unbinds."""
        if texels is None:
            check(lib().ilm_ctx_set_light_ramp(self.handle, None, 0, 0))
            return
        a = np.ascontiguousarray(texels, dtype=np.float32)
        check(lib().ilm_ctx_set_light_ramp(self.handle, _ptr(a), a.shape[1], a.shape[0]))

    def set_lightmap_blend(self, fp16_per_light):
        """ilm_ctx_set_lightmap_blend: True = the reference's HalfVector4 render target (rounded through fp16 after every light)."""
        check(lib().ilm_ctx_set_lightmap_blend(self.handle, 1 if fp16_per_light else 0))

    def set_light_split(self, workgroups):
        """ilm_ctx_set_light_split: workgroups per tile of the sphere-light launches (0 = chosen per launch, 1 / 2 / 4 / 8)."""
        check(lib().ilm_ctx_set_light_split(self.handle, int(workgroups)))

    def last_light_launch(self):
        """ilm_debug_last_light_launch: (workgroups, largest split, tile-group edge) of the context's last light-pass launch."""
        wg, sp, mc = C.c_int32(), C.c_int32(), C.c_int32()
        check(lib().ilm_debug_last_light_launch(self.handle, C.byref(wg), C.byref(sp), C.byref(mc)))
        return int(wg.value), int(sp.value), int(mc.value)

    def debug_divide(self, numerators, denominators):
        """ilm_debug_divide: (the cone trace's unscaled division, the IEEE division) of the operand pairs, both evaluated on the device."""
        n = np.ascontiguousarray(numerators, dtype=np.float32).ravel()
        d = np.ascontiguousarray(denominators, dtype=np.float32).ravel()
        assert n.shape == d.shape
        fast = np.empty_like(n); ieee = np.empty_like(n)
        check(lib().ilm_debug_divide(self.handle, _ptr(n), _ptr(d), n.shape[0], _ptr(fast), _ptr(ieee)))
        return fast, ieee

    def debug_divide_by_constants(self):
        """ilm_debug_divide_by_constants: [(divisor, mismatches among the numerators inside the admitted range, mismatches outside it)] over all 2^32 numerators."""
        d = np.zeros(4, np.float32); bad = np.zeros(8, np.uint64); n = C.c_int32()
        check(lib().ilm_debug_divide_by_constants(self.handle, _ptr(d), _ptr(bad), 4, C.byref(n)))
        return [(float(d[i]), int(bad[2 * i]), int(bad[2 * i + 1])) for i in range(n.value)]

    def timer_start(self):
        check(lib().ilm_timer_start(self.handle))

    def timer_stop(self):
        ms = C.c_float()
        check(lib().ilm_timer_stop(self.handle, C.byref(ms)))
        return float(ms.value)

    def close(self):
        if self.handle.value and self.owned:
            lib().ilm_ctx_destroy(self.handle)
        self.handle = abi.Handle(0)


class Engine:
    """ilm_engine_*: chunk size + randomness table."""

    def __init__(self, ctx, chunk_size, randomness):
        rnd = np.ascontiguousarray(randomness, dtype=np.float32)
        assert rnd.ndim == 3 and rnd.shape[2] == 4
        self.ctx = ctx
        self.chunk_size = int(chunk_size)
        self.slots = self.chunk_size * self.chunk_size
        self.randomness = rnd
        self.handle = abi.Handle(0)
        check(lib().ilm_engine_create(ctx.handle, chunk_size, _ptr(rnd), rnd.shape[1], rnd.shape[0], C.byref(self.handle)))

    def close(self):
        if self.handle.value:
            lib().ilm_engine_destroy(self.handle)
            self.handle = abi.Handle(0)


class System:
    """ilm_system_*: a table of chunks updated by ilm_system_step."""

    def __init__(self, engine):
        self.engine = engine
        self.handle = abi.Handle(0)
        check(lib().ilm_system_create(engine.handle, C.byref(self.handle)))

    def add_chunk(self):
        i = C.c_int32()
        check(lib().ilm_system_add_chunk(self.handle, C.byref(i)))
        return int(i.value)

    def remove_chunk(self, index):
        check(lib().ilm_system_remove_chunk(self.handle, index))

    def chunk_count(self):
        n = C.c_int32()
        check(lib().ilm_system_chunk_count(self.handle, C.byref(n)))
        return int(n.value)

    def upload(self, chunk, plane, data, first_slot=0):
        a = np.ascontiguousarray(data, dtype=np.float32).reshape(-1, 4)
        check(lib().ilm_chunk_upload(self.handle, chunk, plane, _ptr(a), first_slot, a.shape[0]))

    def download(self, chunk, plane, first_slot=0, count=None):
        if count is None:
            count = self.engine.slots - first_slot
        out = np.empty((count, 4), dtype=np.float32)
        check(lib().ilm_chunk_download(self.handle, chunk, plane, _ptr(out), first_slot, count))
        return out

    def device_ptr(self, chunk, component):
        p = C.c_void_p()
        stride = C.c_int64()
        check(lib().ilm_chunk_device_ptr(self.handle, chunk, component, C.byref(p), C.byref(stride)))
        return p.value, int(stride.value)

    def set_distance_field(self, sdf):
        check(lib().ilm_system_set_distance_field(self.handle, sdf.handle if sdf is not None else abi.Handle(0)))
        self._sdf = sdf

    def set_life_ramp(self, texels):
        if texels is None:
            check(lib().ilm_system_set_life_ramp(self.handle, None, 0, 0))
            return
        a = np.ascontiguousarray(texels, dtype=np.float32)
        assert a.ndim == 3 and a.shape[2] == 4
        check(lib().ilm_system_set_life_ramp(self.handle, _ptr(a), a.shape[1], a.shape[0]))

    def step(self, desc):
        check(lib().ilm_system_step(self.handle, _byref(desc)))

    def spawn(self, chunk, sys, p):
        check(lib().ilm_spawn(self.handle, chunk, _byref(sys), _byref(p)))

    def gravity(self, chunk, sys, p):
        check(lib().ilm_gravity(self.handle, chunk, _byref(sys), _byref(p)))

    def noise(self, chunk, sys, p):
        check(lib().ilm_noise(self.handle, chunk, _byref(sys), _byref(p)))

    def fma(self, chunk, sys, p):
        check(lib().ilm_fma(self.handle, chunk, _byref(sys), _byref(p)))

    def matrix_multiply(self, chunk, sys, p):
        check(lib().ilm_matrix_multiply(self.handle, chunk, _byref(sys), _byref(p)))

    def spatial_noise(self, chunk, sys, p):
        check(lib().ilm_spatial_noise(self.handle, chunk, _byref(sys), _byref(p)))

    def set_spawn_positions(self, slot, positions):
        """ilm_system_set_spawn_positions: (n, 4) float32 (xyz, life) position constants of spawn record `slot`; None releases."""
        if positions is None:
            check(lib().ilm_system_set_spawn_positions(self.handle, slot, None, 0))
            return
        a = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 4)
        check(lib().ilm_system_set_spawn_positions(self.handle, slot, _ptr(a), a.shape[0]))

    def set_bitmap(self, texels):
        """ilm_system_set_bitmap: (h, w, 4) float32 sprite sheet for the textured rasterise techniques; None releases."""
        if texels is None:
            check(lib().ilm_system_set_bitmap(self.handle, None, 0, 0))
            return
        a = np.ascontiguousarray(texels, dtype=np.float32)
        check(lib().ilm_system_set_bitmap(self.handle, _ptr(a), a.shape[1], a.shape[0]))

    def set_spawn_pattern(self, slot, levels):
        """ilm_system_set_spawn_pattern: `levels` = [level 0 (h, w, 4) float32, level 1 (max(1, h >> 1), max(1, w >> 1), 4), ...] of the
        PatternSpawner texture of spawn record `slot`; None releases."""
        if not levels:
            check(lib().ilm_system_set_spawn_pattern(self.handle, slot, None, 0, 0, 0))
            return
        h, w = levels[0].shape[0], levels[0].shape[1]
        for l, a in enumerate(levels):
            if a.shape[:2] != (max(1, h >> l), max(1, w >> l)):
                raise ValueError("mip level %d is %s, expected %s" % (l, a.shape[:2], (max(1, h >> l), max(1, w >> l))))
        flat = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.float32).reshape(-1, 4) for a in levels]))
        check(lib().ilm_system_set_spawn_pattern(self.handle, slot, _ptr(flat), w, h, len(levels)))

    def update(self, chunk, sys, p, df=None):
        check(lib().ilm_update(self.handle, chunk, _byref(sys), _byref(p), _byref(df)))

    def erase(self, chunk):
        check(lib().ilm_erase(self.handle, chunk))

    def live_counts(self, saturate16=False):
        n = self.chunk_count()
        out = np.zeros(max(n, 1), dtype=np.uint32)
        check(lib().ilm_system_live_counts(self.handle, _ptr(out), out.shape[0], 1 if saturate16 else 0))
        return out[:n]

    def step_counts(self, saturate16=False):
        n = self.chunk_count()
        out = np.zeros(max(n, 1), dtype=np.uint32)
        check(lib().ilm_system_step_counts(self.handle, _ptr(out), out.shape[0], 1 if saturate16 else 0))
        return out[:n]

    def poll_counts(self, saturate16=False):
        """Non-blocking: returns the counts of the last counting step, or None while the GPU is still busy."""
        n = self.chunk_count()
        out = np.zeros(max(n, 1), dtype=np.uint32)
        ready = C.c_int32()
        check(lib().ilm_system_poll_counts(self.handle, _ptr(out), out.shape[0], 1 if saturate16 else 0, C.byref(ready)))
        return out[:n] if ready.value else None

    def live_slots(self, chunk):
        out = np.zeros(self.engine.slots, dtype=np.uint32)
        n = C.c_int32()
        check(lib().ilm_chunk_live_slots(self.handle, chunk, _ptr(out), out.shape[0], C.byref(n)))
        return out[:n.value]

    def readback(self, params, element_counts=None, chunk_count=None, capacity=None):
        """ilm_system_readback: (array of abi.ReadbackDrawCall for the live particles in chunk / slot order, total live count)."""
        if chunk_count is None:
            chunk_count = self.chunk_count()
        if capacity is None:
            capacity = max(1, chunk_count * self.engine.slots)
        e = np.ascontiguousarray(element_counts, dtype=np.int32) if element_counts is not None else None
        out = (abi.ReadbackDrawCall * capacity)()
        n = C.c_int32()
        check(lib().ilm_system_readback(self.handle, _ptr(e) if e is not None else None, chunk_count, _byref(params),
                                        C.cast(out, C.c_void_p), capacity, C.byref(n)))
        return out, int(n.value)

    def readback_view(self, params, element_counts=None, chunk_count=None):
        """ilm_system_readback_view: (n, 12) float32 view of the draw-call records in the context's pinned buffer (valid until the next
        read-back on the context) -- no host copy."""
        if chunk_count is None:
            chunk_count = self.chunk_count()
        e = np.ascontiguousarray(element_counts, dtype=np.int32) if element_counts is not None else None
        ptr = C.c_void_p()
        n = C.c_int32()
        check(lib().ilm_system_readback_view(self.handle, _ptr(e) if e is not None else None, chunk_count, _byref(params),
                                             C.byref(ptr), C.byref(n)))
        if n.value == 0:
            return np.zeros((0, C.sizeof(abi.ReadbackDrawCall) // 4), np.float32)
        buf = (C.c_float * (n.value * (C.sizeof(abi.ReadbackDrawCall) // 4))).from_address(ptr.value)
        return np.frombuffer(buf, dtype=np.float32).reshape(n.value, -1)

    def close(self):
        if self.handle.value:
            lib().ilm_system_destroy(self.handle)
            self.handle = abi.Handle(0)


class DistanceFieldTexture:
    """ilm_sdf_*: the RGBA16 atlas on the device."""

    def __init__(self, ctx, texels=None, fmt=abi.SDF_UNORM16, size=None):
        """texels: (H, W, 4) uint16 atlas to upload, or None with size=(width, height) for an empty (zeroed) atlas."""
        self.ctx = ctx
        self.format = fmt
        self.handle = abi.Handle(0)
        if texels is None:
            self.width, self.height = int(size[0]), int(size[1])
            check(lib().ilm_sdf_create(ctx.handle, self.width, self.height, fmt, C.byref(self.handle)))
            return
        a = np.ascontiguousarray(texels, dtype=np.uint16)
        assert a.ndim == 3 and a.shape[2] == 4
        self.height, self.width = a.shape[0], a.shape[1]
        check(lib().ilm_sdf_create(ctx.handle, self.width, self.height, fmt, C.byref(self.handle)))
        check(lib().ilm_sdf_upload(self.handle, _ptr(a)))

    def upload(self, texels):
        """ilm_sdf_upload: replaces the atlas ((H, W, 4) uint16)."""
        a = np.ascontiguousarray(texels, dtype=np.uint16)
        assert a.shape == (self.height, self.width, 4)
        check(lib().ilm_sdf_upload(self.handle, _ptr(a)))

    def device_ptr(self):
        """ilm_sdf_device_ptr: the atlas' device address (from then on the library assumes the caller may write it)."""
        p = C.c_void_p()
        check(lib().ilm_sdf_device_ptr(self.handle, C.byref(p)))
        return p.value

    def mark_dirty(self, first_virtual_slice=0, slice_count=0):
        """ilm_sdf_mark_dirty: the caller wrote those virtual slices through device_ptr() (slice_count 0: the whole atlas)."""
        check(lib().ilm_sdf_mark_dirty(self.handle, first_virtual_slice, slice_count))

    def trace_info(self):
        """ilm_sdf_trace_info"""
        info = abi.SdfTraceInfo()
        check(lib().ilm_sdf_trace_info(self.handle, C.byref(info)))
        return info

    def download(self):
        """ilm_sdf_download: the atlas as (H, W, 4) uint16."""
        out = np.empty((self.height, self.width, 4), dtype=np.uint16)
        check(lib().ilm_sdf_download(self.handle, _ptr(out)))
        return out

    def render_slices(self, desc, first_virtual_slices, obstructions=None, volumes=None, polygon_xy=None, clear_source=None):
        """ilm_sdf_render_slices.  obstructions / volumes: ctypes arrays of abi.Obstruction / abi.HeightVolume (or None);
        polygon_xy: (n, 2) float32."""
        sl = np.ascontiguousarray(first_virtual_slices, dtype=np.int32)
        no = len(obstructions) if obstructions is not None else 0
        nv = len(volumes) if volumes is not None else 0
        poly = np.ascontiguousarray(polygon_xy, dtype=np.float32).reshape(-1, 2) if polygon_xy is not None else np.zeros((0, 2), np.float32)
        check(lib().ilm_sdf_render_slices(
            self.handle, clear_source.handle if clear_source is not None else abi.Handle(0), _byref(desc),
            _ptr(sl), sl.shape[0], C.cast(obstructions, C.c_void_p) if no else None, no,
            C.cast(volumes, C.c_void_p) if nv else None, nv, _ptr(poly) if poly.shape[0] else None, poly.shape[0]))

    def sample(self, df, positions):
        """ilm_sdf_sample: distances at (n, 3) float32 world positions."""
        positions = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        out = np.empty(positions.shape[0], dtype=np.float32)
        check(lib().ilm_sdf_sample(self.handle, _byref(df), _ptr(positions), positions.shape[0], _ptr(out)))
        return out

    def sample_inside(self, df, positions):
        """ilm_debug_sdf_sample_inside: (distances, used_table) -- the cone trace's table-driven in-volume sampler where its precondition holds."""
        positions = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        out = np.empty(positions.shape[0], dtype=np.float32)
        used = np.zeros(positions.shape[0], dtype=np.int32)
        check(lib().ilm_debug_sdf_sample_inside(self.handle, _byref(df), _ptr(positions), positions.shape[0], _ptr(out), _ptr(used)))
        return out, used.astype(bool)

    def upload(self, texels):
        a = np.ascontiguousarray(texels, dtype=np.uint16)
        assert a.shape == (self.height, self.width, 4)
        check(lib().ilm_sdf_upload(self.handle, _ptr(a)))

    def close(self):
        if self.handle.value:
            lib().ilm_sdf_destroy(self.handle)
            self.handle = abi.Handle(0)


class GBufferTexture:
    def __init__(self, ctx, texels=None, fmt=abi.GBUFFER_FLOAT4, size=None):
        """texels: (H, W, 4) array to upload, or None with size=(width, height) for an empty G-buffer to render into."""
        dt = np.float32 if fmt == abi.GBUFFER_FLOAT4 else np.uint16
        self.ctx = ctx
        self.format = fmt
        self.handle = abi.Handle(0)
        if texels is None:
            self.width, self.height = int(size[0]), int(size[1])
            check(lib().ilm_gbuffer_create(ctx.handle, self.width, self.height, fmt, C.byref(self.handle)))
            return
        a = np.ascontiguousarray(texels, dtype=dt)
        assert a.ndim == 3 and a.shape[2] == 4
        self.height, self.width = a.shape[0], a.shape[1]
        check(lib().ilm_gbuffer_create(ctx.handle, self.width, self.height, fmt, C.byref(self.handle)))
        check(lib().ilm_gbuffer_upload(self.handle, _ptr(a)))

    def render(self, desc, volumes=None, polygon_xy=None):
        """ilm_gbuffer_render: ground plane + height-volume top faces."""
        nv = len(volumes) if volumes is not None else 0
        poly = np.ascontiguousarray(polygon_xy, dtype=np.float32).reshape(-1, 2) if polygon_xy is not None else np.zeros((0, 2), np.float32)
        check(lib().ilm_gbuffer_render(self.handle, _byref(desc), C.cast(volumes, C.c_void_p) if nv else None, nv,
                                       _ptr(poly) if poly.shape[0] else None, poly.shape[0]))

    def render_meshes(self, desc, top=None, front=None, billboards=None, runs=()):
        """ilm_gbuffer_render_meshes.  top / front: (n, 9) float32 rows of HeightVolumeVertex; billboards: (4 q, 12) float32 rows of
        BillboardVertex; runs: (texture Lightmap | None, first_quad, quad_count, type) per run."""
        def rows(a, n):
            return np.ascontiguousarray(a, dtype=np.float32).reshape(-1, n) if a is not None else np.zeros((0, n), np.float32)
        top, front, bb = rows(top, 9), rows(front, 9), rows(billboards, 12)
        c_runs = (abi.BillboardRun * max(len(runs), 1))()
        for r, (tex, first, count, kind) in enumerate(runs):
            c_runs[r].Texture = tex.handle.value if tex is not None else 0
            c_runs[r].FirstQuad, c_runs[r].QuadCount, c_runs[r].Type = first, count, kind
        check(lib().ilm_gbuffer_render_meshes(self.handle, _byref(desc), _ptr(top) if len(top) else None, len(top),
                                              _ptr(front) if len(front) else None, len(front), _ptr(bb) if len(bb) else None, len(bb),
                                              C.cast(c_runs, C.c_void_p) if len(runs) else None, len(runs)))

    def download(self):
        dt = np.float32 if self.format == abi.GBUFFER_FLOAT4 else np.uint16
        out = np.empty((self.height, self.width, 4), dtype=dt)
        check(lib().ilm_gbuffer_download(self.handle, _ptr(out)))
        return out

    def close(self):
        if self.handle.value:
            lib().ilm_gbuffer_destroy(self.handle)
            self.handle = abi.Handle(0)


_LM_DTYPE = {abi.LIGHTMAP_FLOAT4: (np.float32, 4), abi.LIGHTMAP_HALF4: (np.float16, 4), abi.LIGHTMAP_RGBA8: (np.uint8, 4)}


class Lightmap:
    def __init__(self, ctx, width, height, fmt=abi.LIGHTMAP_FLOAT4, external_ptr=None, borrowed_handle=None):
        """borrowed_handle: a lightmap owned by someone else (a group lightmap's member): close() leaves it alone."""
        self.ctx = ctx
        self.width, self.height, self.format = width, height, fmt
        self.owned = borrowed_handle is None
        if borrowed_handle is not None:
            self.handle = abi.Handle(int(borrowed_handle))
            return
        self.handle = abi.Handle(0)
        check(lib().ilm_lightmap_create(ctx.handle, width, height, fmt, external_ptr, C.byref(self.handle)))

    def download(self, first_row=0, row_count=None):
        if row_count is None:
            row_count = self.height - first_row
        dt, ch = _LM_DTYPE[self.format]
        out = np.empty((row_count, self.width, ch), dtype=dt)
        check(lib().ilm_lightmap_download(self.handle, _ptr(out), first_row, row_count))
        return out

    def upload(self, texels, first_row=0):
        """ilm_lightmap_upload: rows of texels in the lightmap's own format (float32 x 4, float16 x 4 or uint8 x 4)."""
        dt, ch = _LM_DTYPE[self.format]
        a = np.ascontiguousarray(texels, dtype=dt).reshape(-1, self.width, ch)
        check(lib().ilm_lightmap_upload(self.handle, _ptr(a), first_row, a.shape[0]))

    def device_ptr(self):
        p = C.c_void_p()
        check(lib().ilm_lightmap_device_ptr(self.handle, C.byref(p)))
        return p.value

    def clear(self, rgba=(0.0, 0.0, 0.0, 0.0)):
        """ilm_lightmap_clear"""
        c = (C.c_float * 4)(*rgba)
        check(lib().ilm_lightmap_clear(self.handle, c))

    def close(self):
        if self.handle.value and self.owned:
            lib().ilm_lightmap_destroy(self.handle)
        self.handle = abi.Handle(0)


GATHER_NONE, GATHER_PEER, GATHER_RCCL, GATHER_STORE = 0, 1, 2, 3
GATHER_ASYNC = 0x100        # flag: the exchange on the members' second streams (GroupLightmap.wait orders later work behind it)


class Group:
    """ilm_group_*: the members of a multi-device group that live in this process.

    Group([0, 1, ...])                                  one process drives the listed devices
    Group.rank(device, rank, world, unique_id_bytes)    one process per GPU (unique_id from Group.unique_id() on rank 0)
    """

    def __init__(self, device_ids=None, _handle=None):
        if _handle is not None:
            self.handle = _handle
        else:
            ids = (C.c_int32 * len(device_ids))(*[int(d) for d in device_ids])
            self.handle = abi.Handle(0)
            check(lib().ilm_group_create(C.cast(ids, C.c_void_p), len(device_ids), C.byref(self.handle)))
        n, w, f, cr = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        check(lib().ilm_group_info(self.handle, C.byref(n), C.byref(w), C.byref(f), C.byref(cr)))
        self.n_local, self.world, self.first_rank = int(n.value), int(w.value), int(f.value)
        self.contexts = []
        for i in range(self.n_local):
            h = abi.Handle(0)
            check(lib().ilm_group_ctx(self.handle, i, C.byref(h)))
            self.contexts.append(Context(borrowed_handle=h.value))

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        check(lib().ilm_group_unique_id(C.cast(buf, C.c_void_p)))
        return bytes(buf)

    @classmethod
    def rank(cls, device_id, rank, world, unique_id):
        assert len(unique_id) == 128
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        h = abi.Handle(0)
        check(lib().ilm_group_create_rank(device_id, rank, world, C.cast(buf, C.c_void_p), C.byref(h)))
        return cls(_handle=h)

    def comm_ranks(self):
        """The rank count the RCCL communicator reports (0 while none has been created)."""
        cr = C.c_int32()
        check(lib().ilm_group_info(self.handle, None, None, None, C.byref(cr)))
        return int(cr.value)

    def sync(self):
        check(lib().ilm_group_sync(self.handle))

    def all_gather(self, device_ptrs, bytes_per_rank, gather):
        ptrs = (C.c_void_p * self.n_local)(*[int(p) for p in device_ptrs])
        check(lib().ilm_group_all_gather(self.handle, C.cast(ptrs, C.c_void_p), int(bytes_per_rank), gather))

    def host_all_gather(self, local_bytes):
        """ilm_group_host_all_gather: `local_bytes` = this process's slots (n_local equal slots); returns the `world` slots as a list of
        bytes objects.  Also a barrier: every member's queued work has finished when it returns."""
        assert len(local_bytes) % self.n_local == 0
        per = len(local_bytes) // self.n_local
        src = (C.c_uint8 * len(local_bytes)).from_buffer_copy(local_bytes)
        dst = (C.c_uint8 * (per * self.world))()
        check(lib().ilm_group_host_all_gather(self.handle, C.cast(src, C.c_void_p), C.cast(dst, C.c_void_p), per))
        raw = bytes(dst)
        return [raw[r * per:(r + 1) * per] for r in range(self.world)]

    def live_counts(self, systems, total_chunks, saturate16=False):
        hs = (abi.Handle * self.n_local)(*[s.handle.value for s in systems])
        out = np.zeros(max(total_chunks, 1), dtype=np.uint32)
        check(lib().ilm_group_live_counts(self.handle, C.cast(hs, C.c_void_p), total_chunks, _ptr(out), out.shape[0], 1 if saturate16 else 0))
        return out[:total_chunks]

    def gather_chunks(self, sources, gathered, total_chunks, first_component=0, component_count=4, gather=GATHER_RCCL):
        """ilm_group_gather_chunks: components [first, first + count) of every chunk of the sharded table land in chunk c of every local
        member's `gathered` system (Pos+Life by default).  sources / gathered: one System (or anything with .handle, or a raw handle) per
        local member.  Stream-ordered; a collective when the group spans processes."""
        def handles(xs):
            return (abi.Handle * self.n_local)(*[int(getattr(getattr(x, "handle", x), "value", getattr(x, "handle", x))) for x in xs])
        check(lib().ilm_group_gather_chunks(self.handle, C.cast(handles(sources), C.c_void_p), C.cast(handles(gathered), C.c_void_p), int(total_chunks),
                                            int(first_component), int(component_count), int(gather)))

    def render_sphere_lights(self, lights, env, df, gbuffers, sdfs, ambient, group_lightmap, gather=GATHER_PEER, want_stats=False):
        """ilm_group_render_sphere_lights: every local member renders its strip, then the strips are gathered in place."""
        n = len(lights) if lights is not None else 0
        amb = (C.c_float * 4)(*[float(x) for x in ambient]) if ambient is not None else None
        gb = (abi.Handle * self.n_local)(*[(g.handle.value if g is not None else 0) for g in gbuffers]) if gbuffers is not None else None
        sd = (abi.Handle * self.n_local)(*[(s.handle.value if s is not None else 0) for s in sdfs]) if sdfs is not None else None
        stats = abi.RenderStats() if want_stats else None
        check(lib().ilm_group_render_sphere_lights(
            self.handle, C.cast(lights, C.c_void_p) if n else None, n, _byref(env), _byref(df),
            C.cast(gb, C.c_void_p) if gb is not None else None, C.cast(sd, C.c_void_p) if sd is not None else None,
            C.cast(amb, C.c_void_p) if amb is not None else None, group_lightmap.handle, gather, _byref(stats)))
        return stats

    def close(self):
        if self.handle.value:
            check(lib().ilm_group_destroy(self.handle))
            self.handle = abi.Handle(0)
            self.contexts = []


class GroupLightmap:
    """ilm_group_lightmap_*: the composited lightmap of a group (one full-frame buffer per local member)."""

    def __init__(self, group, width, height, fmt=abi.LIGHTMAP_FLOAT4):
        self.group, self.width, self.height, self.format = group, width, height, fmt
        self.handle = abi.Handle(0)
        check(lib().ilm_group_lightmap_create(group.handle, width, height, fmt, C.byref(self.handle)))
        b, e, r = C.c_int32(), C.c_int32(), C.c_int32()
        self.strips = []
        for rank in range(group.world):
            check(lib().ilm_group_lightmap_strip(self.handle, rank, C.byref(b), C.byref(e), C.byref(r)))
            self.strips.append((int(b.value), int(e.value)))
        self.slot_rows = int(r.value)
        self.members = []
        for i in range(group.n_local):
            h = abi.Handle(0)
            check(lib().ilm_group_lightmap_member(self.handle, i, C.byref(h)))
            self.members.append(Lightmap(group.contexts[i], width, self.slot_rows * group.world, fmt, borrowed_handle=h.value))

    def set_strips(self, strips=None):
        """ilm_group_lightmap_set_strips: other contiguous strips of whole tile bands, one (begin, end) per rank (None: the equal slots)."""
        if strips is None:
            check(lib().ilm_group_lightmap_set_strips(self.handle, None, None))
        else:
            n = self.group.world
            assert len(strips) == n
            b = (C.c_int32 * n)(*[int(s[0]) for s in strips])
            e = (C.c_int32 * n)(*[int(s[1]) for s in strips])
            check(lib().ilm_group_lightmap_set_strips(self.handle, b, e))
        bb, ee, r = C.c_int32(), C.c_int32(), C.c_int32()
        self.strips = []
        for rank in range(self.group.world):
            check(lib().ilm_group_lightmap_strip(self.handle, rank, C.byref(bb), C.byref(ee), C.byref(r)))
            self.strips.append((int(bb.value), int(ee.value)))

    def gather(self, gather):
        check(lib().ilm_group_lightmap_gather(self.handle, gather))

    def wait(self):
        """ilm_group_lightmap_wait: the members' context streams wait for the asynchronous exchange queued last (no host block)."""
        check(lib().ilm_group_lightmap_wait(self.handle))

    def store_mode(self, enable=True):
        """ilm_group_lightmap_store_mode: while armed, every light pass into a member's lightmap also stores into the other members' copies
        of the frame; gather(GATHER_STORE) is then the fence that replaces the exchange.  For a group that spans processes the call is a
        COLLECTIVE (arming and disarming): the ranks' buffers are mapped through IPC handles, proven, and every rank arms or none."""
        check(lib().ilm_group_lightmap_store_mode(self.handle, 1 if enable else 0))

    def download(self, local_index=0):
        """The frame (first `height` rows) as local member `local_index` holds it."""
        return self.members[local_index].download(0, self.height)

    def close(self):
        if self.handle.value:
            check(lib().ilm_group_lightmap_destroy(self.handle))
            self.handle = abi.Handle(0)
            self.members = []


def render_particles(system, params, target, quad_counts=None, chunk_count=None, want_stats=False):
    """ilm_render_particles: blends the live particles of `system` onto the lightmap `target` in chunk / slot order.
    Returns (live quads, (quad, tile) pairs, shaded pixels) when want_stats."""
    if chunk_count is None:
        chunk_count = system.chunk_count()
    q = np.ascontiguousarray(quad_counts, dtype=np.int32) if quad_counts is not None else None
    stats = (C.c_uint64 * 3)() if want_stats else None
    check(lib().ilm_render_particles(system.handle, _ptr(q) if q is not None else None, chunk_count, _byref(params), target.handle,
                                     C.cast(stats, C.c_void_p) if stats is not None else None))
    return tuple(int(x) for x in stats) if want_stats else None


def render_sphere_lights(ctx, lights, env, df, gbuffer, sdf, ambient, lightmap, row_begin=0, row_end=None, want_stats=False):
    """ilm_render_sphere_lights.  lights: ctypes array of abi.LightVertex (or None for zero lights); ambient None = add this light
    group to what the lightmap already holds."""
    if row_end is None:
        row_end = lightmap.height
    n = len(lights) if lights is not None else 0
    amb = (C.c_float * 4)(*[float(x) for x in ambient]) if ambient is not None else None
    stats = abi.RenderStats() if want_stats else None
    check(lib().ilm_render_sphere_lights(
        ctx.handle, C.cast(lights, C.c_void_p) if n else None, n, _byref(env), _byref(df),
        gbuffer.handle if gbuffer is not None else abi.Handle(0),
        sdf.handle if sdf is not None else abi.Handle(0),
        C.cast(amb, C.c_void_p) if amb is not None else None, lightmap.handle, row_begin, row_end, _byref(stats)))
    return stats


def render_particle_lights(ctx, system, params, env, df, gbuffer, sdf, lightmap, quad_counts=None, chunk_count=None, row_begin=0, row_end=None,
                           want_stats=False):
    """ilm_render_particle_lights: one sphere light per live particle of `system`, added onto the lightmap's contents."""
    if row_end is None:
        row_end = lightmap.height
    if chunk_count is None:
        chunk_count = system.chunk_count()
    q = np.ascontiguousarray(quad_counts, dtype=np.int32) if quad_counts is not None else None
    stats = abi.RenderStats() if want_stats else None
    check(lib().ilm_render_particle_lights(
        ctx.handle, system.handle, _ptr(q) if q is not None else None, chunk_count, _byref(params), _byref(env), _byref(df),
        gbuffer.handle if gbuffer is not None else abi.Handle(0), sdf.handle if sdf is not None else abi.Handle(0),
        lightmap.handle, row_begin, row_end, _byref(stats)))
    return stats


def render_light_probes(ctx, lights, probe_positions, probe_normals, env, df, sdf):
    """ilm_render_light_probes: (n, 4) float32 probe values (rgb sum, contributing light count)."""
    n = len(lights) if lights is not None else 0
    pp = np.ascontiguousarray(probe_positions, dtype=np.float32).reshape(-1, 4)
    pn = np.ascontiguousarray(probe_normals, dtype=np.float32).reshape(-1, 4)
    assert pp.shape == pn.shape
    out = np.zeros_like(pp)
    check(lib().ilm_render_light_probes(ctx.handle, C.cast(lights, C.c_void_p) if n else None, n, _ptr(pp), _ptr(pn), pp.shape[0],
                                        _byref(env), _byref(df), sdf.handle if sdf is not None else abi.Handle(0), _ptr(out)))
    return out


def resolve_lighting(src, dst, hdr, row_begin=0, row_end=None, albedo=None):
    """ilm_resolve_lighting[_with_albedo]: tone-map lightmap `src` into `dst` (same size, any formats); albedo: a Lightmap holding the albedo texture."""
    if row_end is None:
        row_end = min(src.height, dst.height) if albedo is None else min(src.height, dst.height, albedo.height)
    if albedo is None:
        check(lib().ilm_resolve_lighting(src.handle, dst.handle, _byref(hdr), row_begin, row_end))
    else:
        check(lib().ilm_resolve_lighting_with_albedo(src.handle, albedo.handle, dst.handle, _byref(hdr), row_begin, row_end))
