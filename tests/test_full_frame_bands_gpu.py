"""The WHOLE cfg3 and cfg5 frames against the CPU oracle, band by band (tests/golden/full_frame_bands.json, written in the build container by
tests/golden/make_full_frame_bands.py: the oracle over every 16-row band of both frames, the ground-plane G-buffer bound).

Per band the kernel -- through the C ABI, with the statistics variant -- must reproduce the oracle's SDF-sample, pixel.light-pair and
traced-pair counts exactly (ConeTrace.fxh:148-191 runs 6.9 G times per cfg5 frame: every one of them is counted here), the alpha plane
(1 + the lights that contribute to each pixel) bit for bit (CRC-32), the band's mean colour to 1e-5 relative, and the probe texels (two
rows per band, every 97th pixel: 10 800 texels of cfg5, 2 720 of cfg3) to the suite's pointwise criterion."""
import json
import os
import zlib

import numpy as np
import pytest

from illuminant_amd import abi, native, scenes
from tests.util import assert_close

pytestmark = pytest.mark.gpu
FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "full_frame_bands.json")


def frame(ctx, name):
    """The frames bench.py times (r05): both fields generated on the device from the 256 obstructions of seed 11, the ground plane as Vector4
    texels, the lights of seed 12 / 13."""
    if name == "cfg3":
        w, h = 1920, 1080
        layout = scenes.DistanceFieldLayout(2048, 2048, 128.0, 32, 0.25)
        obstacles = scenes.random_obstacles(11, 256, (2048, 2048))
        sfmt = abi.SDF_UNORM16
        dfu = layout.uniforms(power=0.7, min_step_size=1.0, long_step_factor=0.5)
        lights = scenes.random_lights(12, 64, w, h)
    else:
        w, h = 3840, 2160
        layout = scenes.DistanceFieldLayout(4096, 4096, 128.0, 32, 0.125, 128)
        obstacles = scenes.random_obstacles(11, 256, (4096, 4096))
        sfmt = abi.SDF_FP16
        dfu = layout.uniforms(max_cone_radius=24.0, power=0.7, step_limit=64, min_step_size=1.0, long_step_factor=0.5)
        lights = scenes.random_lights(13, 256, w, h, z=(8.0, 64.0), radius=24.0, ramp=(400.0, 1100.0))
    sdf = native.DistanceFieldTexture(ctx, None, sfmt, size=(layout.atlas_width, layout.atlas_height))
    sdf.render_slices(scenes.render_desc(layout), list(range(0, layout.slice_count, 3)), scenes.obstruction_array([(t - 1, c, s) for (t, c, s) in obstacles]))
    gfmt = abi.GBUFFER_FLOAT4
    gb = native.GBufferTexture(ctx, scenes.ground_plane_gbuffer(w, h, gfmt), gfmt)
    return w, h, dfu, lights, sdf, gb


@pytest.mark.parametrize("name", ["cfg3", "cfg5"])
def test_every_band_of_the_full_frame_matches_the_oracle(ctx, name):
    golden = json.load(open(FIXTURE))[name]
    w, h, dfu, lights, sdf, gb = frame(ctx, name)
    assert (w, h) == (golden["width"], golden["height"])
    env = scenes.environment(gbuffer_size=(w, h))
    ambient = (0.05, 0.05, 0.05, 1.0)
    lm = native.Lightmap(ctx, w, h, abi.LIGHTMAP_FLOAT4)
    totals = [0, 0, 0]
    try:
        for band in golden["bands"]:
            b0, b1 = band["rows"]
            st = native.render_sphere_lights(ctx, lights, env, dfu, gb, sdf, ambient, lm, b0, b1, want_stats=True)
            got = (st.SdfSamples, st.PixelLightPairs, st.TracedPairs)
            assert got == (band["sdf_samples"], band["pairs"], band["traced"]), "%s rows %d-%d" % (name, b0, b1)
            totals = [t + g for t, g in zip(totals, got)]
        img = lm.download()
        probe_got, probe_want = [], []
        for band in golden["bands"]:
            b0, b1 = band["rows"]
            rows = img[b0:b1]
            assert (zlib.crc32(np.ascontiguousarray(rows[..., 3]).tobytes()) & 0xFFFFFFFF) == band["alpha_crc32"], "%s rows %d-%d: alpha plane" % (name, b0, b1)
            mean = [float(rows[..., c].astype(np.float64).mean()) for c in range(3)]
            assert np.allclose(mean, band["mean_rgb"], rtol=1e-5, atol=1e-7), "%s rows %d-%d: mean colour %s vs %s" % (name, b0, b1, mean, band["mean_rgb"])
            for (y, x, r, g, b) in band["probes"]:
                probe_got.append(img[int(y), int(x), :3]); probe_want.append((r, g, b))
        assert len(probe_got) > 2000
        assert_close(np.asarray(probe_got, np.float32), np.asarray(probe_want, np.float32), "%s probe texels vs the oracle" % name)
        assert tuple(totals) == (golden["sdf_samples"], golden["pairs"], golden["traced"])
    finally:
        for x in (lm, gb, sdf):
            x.close()
