"""Closed-form cases of tests/golden/rasterize.json run through a backend (CPU oracle or the HIP path)."""
import json
import os

import numpy as np

from illuminant_amd import abi, scenes
from tests.util import assert_close

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CS = 16


def load_cases():
    with open(os.path.join(GOLDEN, "rasterize.json")) as f:
        return json.load(f)["cases"]


def chunk_from_particles(particles, n=CS * CS):
    """particles: [{slot, position[4], size, rotation, color[4]}] -> the five planes of a 16 x 16 chunk"""
    planes = [np.zeros((n, 4), np.float32) for _ in range(5)]
    for q in particles:
        i = q["slot"]
        planes[0][i] = q["position"]
        planes[3][i] = q["color"]                       # RenderColor (premultiplied)
        planes[4][i] = [q["size"], q["rotation"], 0.0, q.get("row", 0.0)]
    return planes


class OracleBackend:
    def __init__(self, oracle):
        self.orc = oracle

    def render(self, chunks, params, width, height, clear, bitmap=None):
        image = np.zeros((height, width, 4), np.float32)
        image[:] = np.asarray(clear, np.float32)
        image, stats = self.orc.render_particles(chunks, params, width, height, image=image, bitmap=bitmap)
        return image, stats


class GpuBackend:
    def __init__(self, ctx):
        from illuminant_amd import native
        self.native, self.ctx = native, ctx

    def render(self, chunks, params, width, height, clear, bitmap=None):
        native = self.native
        eng = native.Engine(self.ctx, int(round(chunks[0][0].shape[0] ** 0.5)), scenes.randomness_table(7))
        sysm = native.System(eng)
        for c, planes in enumerate(chunks):
            sysm.add_chunk()
            for pl, k in ((abi.PLANE_POSITION, 0), (abi.PLANE_RENDER_COLOR, 3), (abi.PLANE_RENDER_DATA, 4)):
                sysm.upload(c, pl, planes[k])
        if bitmap is not None:
            sysm.set_bitmap(bitmap)
        lm = native.Lightmap(self.ctx, width, height, abi.LIGHTMAP_FLOAT4)
        lm.clear(clear)
        live, pairs, shaded = native.render_particles(sysm, params, lm, want_stats=True)
        image = lm.download()
        lm.close(); sysm.close(); eng.close()
        return image, (live, shaded)


def check_case(case, backend):
    w, h = case["width"], case["height"]
    chunks = [chunk_from_particles(case["particles"])]
    rp = case.get("params", {})
    params = scenes.rasterize_params(size=tuple(rp.get("size", (1.0, 1.0))), global_color=tuple(rp.get("global_color", (1, 1, 1, 1))),
                                     origin=tuple(rp.get("origin", (0, 0))), scale=tuple(rp.get("scale", (1, 1))),
                                     size_from_z=rp.get("size_from_z", 0.0), z_to_y=rp.get("z_to_y", 0.0), rounded=rp.get("rounded", False),
                                     rounding_power=abi.ClampedBezier1.constant(rp["rounding"]) if "rounding" in rp else None,
                                     viewport_scale=tuple(rp.get("viewport_scale", (1, 1))), viewport_position=tuple(rp.get("viewport_position", (0, 0))),
                                     blend=abi.BLEND_ADDITIVE if rp.get("additive") else abi.BLEND_ALPHA,
                                     texture_size=(len(case["bitmap"][0]), len(case["bitmap"])) if "bitmap" in case else None,
                                     offset_px=tuple(rp.get("offset_px", (0, 0))), size_px=tuple(rp["size_px"]) if "size_px" in rp else None,
                                     relative_size=rp.get("relative_size", True), bilinear=rp.get("bilinear", True),
                                     animation_rate=tuple(rp.get("animation_rate", (0, 0))),
                                     column_from_velocity=rp.get("column_from_velocity", False), row_from_velocity=rp.get("row_from_velocity", False))
    bitmap = np.asarray(case["bitmap"], np.float32) if "bitmap" in case else None
    image, (live, shaded) = backend.render(chunks, params, w, h, case.get("clear", [0, 0, 0, 0]), bitmap=bitmap)
    assert live == case["live_quads"], case["name"]
    if "shaded_pixels" in case:
        assert shaded == case["shaded_pixels"], case["name"]
    for e in case["pixels"]:
        assert_close(image[e["y"], e["x"]], e["rgba"], "%s: pixel (%d, %d)" % (case["name"], e["x"], e["y"]), rtol=1e-5, atol=1e-6)
    if "covered" in case:
        got = {(x, y) for y in range(h) for x in range(w) if not np.array_equal(image[y, x], np.asarray(case.get("clear", [0, 0, 0, 0]), np.float32))}
        assert got == {tuple(p) for p in case["covered"]}, case["name"]
