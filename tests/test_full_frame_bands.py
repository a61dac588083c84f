"""CPU: tests/golden/full_frame_bands.json is what the oracle says -- one band of cfg3 is recomputed here (the whole fixture takes five minutes:
tests/golden/make_full_frame_bands.py), and the fixture's totals are the sums of its bands."""
import json
import os
import sys
import zlib

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN)


def test_a_band_of_cfg3_is_reproduced_by_the_oracle(oracle):
    import make_full_frame_bands as gen
    from illuminant_amd import scenes
    doc = json.load(open(os.path.join(GOLDEN, "full_frame_bands.json")))
    for name in ("cfg3", "cfg5"):
        d = doc[name]
        assert len(d["bands"]) == (d["height"] + 15) // 16
        assert d["sdf_samples"] == sum(b["sdf_samples"] for b in d["bands"]) and d["pairs"] == sum(b["pairs"] for b in d["bands"])
        assert [b["rows"][0] for b in d["bands"]] == list(range(0, d["height"], 16))
    w, h, dfu, lights, atlas, sfmt, garr, gfmt = gen.scene("cfg3")
    band = doc["cfg3"]["bands"][33]
    b0, b1 = band["rows"]
    img, st = oracle.render_sphere_lights(lights, scenes.environment(gbuffer_size=(w, h)), dfu, oracle.make_texture(garr, gfmt), oracle.make_texture(atlas, sfmt),
                                          (0.05, 0.05, 0.05, 1.0), w, h, row_begin=b0, row_end=b1, want_stats=True)
    assert (st.SdfSamples, st.PixelLightPairs, st.TracedPairs) == (band["sdf_samples"], band["pairs"], band["traced"])
    rows = np.ascontiguousarray(img[b0:b1], np.float32)
    assert (zlib.crc32(np.ascontiguousarray(rows[..., 3]).tobytes()) & 0xFFFFFFFF) == band["alpha_crc32"]
    for (y, x, r, g, b) in band["probes"][:20]:
        assert tuple(float(v) for v in rows[int(y) - b0, int(x), :3]) == (r, g, b)
