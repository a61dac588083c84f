"""The closed-form cases of tests/golden/output_ext.json run through a backend (oracle or HIP path)."""
import json
import os

import numpy as np

from illuminant_amd import abi
from tests.util import assert_close

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CS = 8


def load_cases():
    with open(os.path.join(GOLDEN, "output_ext.json")) as f:
        return json.load(f)["cases"]


def readback_params(size=(1.0, 1.0), region=(0.0, 0.0, 1.0, 1.0), animation_rate=(0.0, 0.0), z_to_y=0.0, column_from_velocity=False,
                    row_from_velocity=False, rotation_from_velocity=False, sorted=False):
    p = abi.ReadbackParams()
    p.Size[:] = size
    p.TextureRegion[:] = region
    p.AnimationRate[:] = animation_rate
    p.ZToY = z_to_y
    p.ColumnFromVelocity, p.RowFromVelocity = int(column_from_velocity), int(row_from_velocity)
    p.RotationFromVelocity, p.SortedReadback = int(rotation_from_velocity), int(sorted)
    return p


def hdr_configuration(mode=0, inverse_scale=1.0, offset=0.0, exposure=1.0, gamma=1.0, middle_gray=0.0, average_luminance=0.0,
                      maximum_luminance=0.0, white_point=1.0):
    h = abi.HDRConfiguration()
    h.Mode, h.InverseScaleFactor, h.Offset, h.Exposure, h.Gamma = mode, inverse_scale, offset, exposure, gamma
    h.MiddleGray, h.AverageLuminance, h.MaximumLuminance, h.WhitePoint = middle_gray, average_luminance, maximum_luminance, white_point
    return h


def records_to_rows(records, count):
    return [(tuple(r.Position), tuple(r.Scale), tuple(r.TextureRegion), r.Rotation, r.SortOrder, tuple(r.MultiplyColor)) for r in list(records)[:count]]


class OracleBackend:
    def __init__(self, oracle):
        self.orc = oracle

    def readback(self, chunk, params):
        recs, n = self.orc.fill_readback_result([chunk], params)
        return records_to_rows(recs, n)

    def resolve(self, lightmap, hdr, albedo=None):
        return self.orc.resolve_lighting(lightmap, hdr, albedo=albedo)


class GpuBackend:
    def __init__(self, ctx):
        from illuminant_amd import native, scenes
        self.native, self.scenes, self.ctx = native, scenes, ctx

    def readback(self, chunk, params):
        native = self.native
        eng = native.Engine(self.ctx, CS, self.scenes.randomness_table(7))
        sysm = native.System(eng)
        sysm.add_chunk()
        for plane, k in ((abi.PLANE_POSITION, 0), (abi.PLANE_RENDER_COLOR, 3), (abi.PLANE_RENDER_DATA, 4)):
            sysm.upload(0, plane, chunk[k])
        recs, n = sysm.readback(params)
        rows = records_to_rows(recs, n)
        sysm.close(); eng.close()
        return rows

    def resolve(self, lightmap, hdr, albedo=None):
        native = self.native
        h, w = lightmap.shape[:2]
        src = native.Lightmap(self.ctx, w, h, abi.LIGHTMAP_FLOAT4)
        # seed: a zero-light pass clears to `ambient`; the fixture lightmaps are constant
        native.render_sphere_lights(self.ctx, None, self.scenes.environment(), abi.DistanceFieldUniforms(), None, None, tuple(lightmap[0, 0]), src)
        dst = native.Lightmap(self.ctx, w, h, abi.LIGHTMAP_FLOAT4)
        tex = None
        if albedo is not None:
            tex = native.Lightmap(self.ctx, w, h, abi.LIGHTMAP_FLOAT4)
            tex.upload(albedo)
        native.resolve_lighting(src, dst, hdr, albedo=tex)
        out = dst.download()
        src.close(); dst.close()
        if tex is not None:
            tex.close()
        return out


def check_case(case, backend):
    if case["kind"] == "readback":
        n = CS * CS
        chunk = [np.zeros((n, 4), np.float32) for _ in range(5)]
        for p in case["particles"]:
            chunk[0][p["slot"]] = p["position"]
            chunk[3][p["slot"]] = p["render_color"]
            chunk[4][p["slot"]] = p["render_data"]
        q = case["params"]
        params = readback_params(q["size"], q["region"], q["animation_rate"], q["z_to_y"], q["column_from_velocity"], q["row_from_velocity"],
                                 q["rotation_from_velocity"], q["sorted"])
        rows = backend.readback(chunk, params)
        assert len(rows) == len(case["expected"])
        for got, e in zip(rows, case["expected"]):
            assert_close(got[0], e["position"], "position", rtol=1e-6)
            assert_close(got[1], e["scale"], "scale", rtol=1e-6)
            assert_close(got[2], e["region"], "texture region", rtol=1e-6)
            assert_close([got[3]], [e["rotation"]], "rotation", rtol=1e-6)
            assert_close([got[4]], [e["sort_order"]], "sort order", rtol=1e-6)
            assert list(got[5]) == e["color"]
    elif case["kind"] == "resolve":
        lm = np.zeros((4, 8, 4), np.float32)
        lm[:] = np.asarray(case["texel"], np.float32)
        h = case["hdr"]
        hdr = hdr_configuration(h["mode"], h.get("inverse_scale", 1.0), h.get("offset", 0.0), h.get("exposure", 1.0), h.get("gamma", 1.0),
                                h.get("middle_gray", 0.0), h.get("average_luminance", 0.0), h.get("maximum_luminance", 0.0), h.get("white_point", 1.0))
        albedo = None
        if "albedo" in case:
            albedo = np.zeros((4, 8, 4), np.float32)
            albedo[:] = np.asarray(case["albedo"], np.float32)
        out = backend.resolve(lm, hdr, albedo)
        assert_close(out[2, 5], case["expected"], "resolved texel", rtol=2e-5)
    else:
        raise AssertionError(case["kind"])
