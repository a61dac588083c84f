"""Build container only (no GPU): the CPU oracle over the WHOLE cfg3 and cfg5 frames, once, band by band (16 rows = one tile row).

Round 3's full-size oracle evidence was a 24-row crop of cfg3 and an 8-row crop of cfg5 -- 0.4 % of cfg5's 6.9 G cone-trace samples
(ConeTrace.fxh:148-191); the rest of the frames was covered by strip invariance and additivity only.  This script runs
oracle.render_sphere_lights over every band of both frames -- the frames bench.py TIMES since r05 (same light seeds, the ground-plane
G-buffer bound as Vector4 texels, both fields generated from the 256 obstructions of seed 11 by the ORACLE's field pass, which is
bit-identical to the device's, tests/test_fields_gpu.py; r04's cfg3 fixture used the numpy rasteriser scenes.build_sdf_atlas) -- and
stores, per band, what a GPU test can hold the kernel to without the oracle's pixels travelling:
  * SDF-sample / pixel.light-pair / traced-pair counts (integers: exact),
  * the CRC-32 of the band's alpha plane (1 + the number of lights that contribute to each pixel: exact integers in fp32),
  * float64 means of r, g, b over the band, and
  * `probe` texels: every 97th pixel of the band's rows 3 and 11, rgb as float32 (pointwise 1e-4 criterion).
Output: tests/golden/full_frame_bands.json (~0.5 MB).      python tests/golden/make_full_frame_bands.py [cfg3|cfg5|both]
"""
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from illuminant_amd import abi, scenes  # noqa: E402
from oracle import oracle as orc  # noqa: E402

BAND = 16
PROBE_ROWS = (3, 11)
PROBE_STEP = 97


def scene(name):
    """(width, height, uniforms, lights, atlas, sdf format, G-buffer array, G-buffer format) exactly as tests/test_properties_gpu.py builds them"""
    if name == "cfg3":
        w, h = 1920, 1080
        layout = scenes.DistanceFieldLayout(2048, 2048, 128.0, 32, 0.25)
        obstacles = scenes.random_obstacles(11, 256, (2048, 2048))
        atlas = np.zeros((layout.atlas_height, layout.atlas_width, 4), np.uint16)
        atlas = orc.render_distance_field_slices(atlas, abi.SDF_UNORM16, scenes.render_desc(layout), list(range(0, layout.slice_count, 3)),
                                                 scenes.obstruction_array([(t - 1, c, s) for (t, c, s) in obstacles]))
        dfu = layout.uniforms(power=0.7, min_step_size=1.0, long_step_factor=0.5)
        lights = scenes.random_lights(12, 64, w, h)
        return w, h, dfu, lights, atlas, abi.SDF_UNORM16, scenes.ground_plane_gbuffer(w, h, abi.GBUFFER_FLOAT4), abi.GBUFFER_FLOAT4
    w, h = 3840, 2160
    layout = scenes.DistanceFieldLayout(4096, 4096, 128.0, 32, 0.125, 128)
    obstacles = scenes.random_obstacles(11, 256, (4096, 4096))
    atlas = np.zeros((layout.atlas_height, layout.atlas_width, 4), np.uint16)
    atlas = orc.render_distance_field_slices(atlas, abi.SDF_FP16, scenes.render_desc(layout), list(range(0, layout.slice_count, 3)),
                                             scenes.obstruction_array([(t - 1, c, s) for (t, c, s) in obstacles]))
    dfu = layout.uniforms(max_cone_radius=24.0, power=0.7, step_limit=64, min_step_size=1.0, long_step_factor=0.5)
    lights = scenes.random_lights(13, 256, w, h, z=(8.0, 64.0), radius=24.0, ramp=(400.0, 1100.0))
    return w, h, dfu, lights, atlas, abi.SDF_FP16, scenes.ground_plane_gbuffer(w, h, abi.GBUFFER_FLOAT4), abi.GBUFFER_FLOAT4


def bands_of(name):
    w, h, dfu, lights, atlas, sfmt, garr, gfmt = scene(name)
    env = scenes.environment(gbuffer_size=(w, h))
    ambient = (0.05, 0.05, 0.05, 1.0)
    tex, gtex = orc.make_texture(atlas, sfmt), orc.make_texture(garr, gfmt)
    out = []
    t0 = time.time()
    for b0 in range(0, h, BAND):
        b1 = min(h, b0 + BAND)
        img, st = orc.render_sphere_lights(lights, env, dfu, gtex, tex, ambient, w, h, row_begin=b0, row_end=b1, want_stats=True)
        band = np.ascontiguousarray(img[b0:b1], np.float32)
        probes = [[int(b0 + r), int(x)] + [float(v) for v in band[r, x, :3]] for r in PROBE_ROWS if b0 + r < b1 for x in range(PROBE_STEP // 2, w, PROBE_STEP)]
        out.append({"rows": [b0, b1], "sdf_samples": int(st.SdfSamples), "pairs": int(st.PixelLightPairs), "traced": int(st.TracedPairs),
                    "alpha_crc32": zlib.crc32(np.ascontiguousarray(band[..., 3]).tobytes()) & 0xFFFFFFFF,
                    "mean_rgb": [float(band[..., c].astype(np.float64).mean()) for c in range(3)], "probes": probes})
        if (b0 // BAND) % 10 == 0:
            print("%s rows %d: %.0f s" % (name, b0, time.time() - t0), flush=True)
    return {"width": w, "height": h, "band_rows": BAND, "probe_rows": list(PROBE_ROWS), "probe_step": PROBE_STEP,
            "sdf_samples": sum(b["sdf_samples"] for b in out), "pairs": sum(b["pairs"] for b in out), "traced": sum(b["traced"] for b in out),
            "bands": out}


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    path = os.path.join(ROOT, "tests", "golden", "full_frame_bands.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    for name in (("cfg3", "cfg5") if which == "both" else (which,)):
        data[name] = bands_of(name)
    data["_generator"] = "tests/golden/make_full_frame_bands.py (oracle/ilm_oracle.c over every 16-row band of the frames of tests/test_properties_gpu.py)"
    json.dump(data, open(path, "w"), separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")
