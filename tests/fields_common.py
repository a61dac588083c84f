"""Shared scene builders of the distance-field generation tests (CPU oracle and HIP path run the same cases)."""
import json
import os

import numpy as np

from illuminant_amd import abi, scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_generation_fixture():
    with open(os.path.join(GOLDEN, "distance_field_generation.json")) as f:
        return json.load(f)


def fixture_layout(doc):
    fd = doc["field"]
    return scenes.DistanceFieldLayout(fd["virtual_width"], fd["virtual_height"], fd["virtual_depth"], fd["requested_slices"],
                                      fd["resolution"], int(fd["maximum_encoded_distance"]))


def fixture_case_inputs(case):
    """(obstruction array | None, volumes | None, polygon | None) of one fixture case."""
    if case["kind"] == "obstruction":
        o = case["obstruction"]
        return scenes.obstruction_array([(o["type"], o["center"], o["size"], o["rotation"])]), None, None
    v = case["volume"]
    vols, poly = scenes.height_volume_arrays([(v["polygon"], v["z_base"], v["height"])])
    return None, vols, poly


def texel_of(atlas, layout, x, y, virtual_slice):
    """Stored code of virtual slice `virtual_slice` at slice-local texel (x, y): channel s % 3 of physical slice s // 3."""
    p, c = virtual_slice // 3, virtual_slice % 3
    ox = (p % layout.column_count) * layout.slice_width
    oy = (p // layout.column_count) * layout.slice_height
    return int(atlas[oy + y, ox + x, c])


def all_triplets(layout):
    return list(range(0, layout.slice_count, 3))


def mixed_scene(seed=21, n=24, extent=(256, 192), depth=96.0, slices=12, resolution=0.5, dynamic_fraction=0.0):
    """Every obstruction type with rotations + two height volumes (one concave)."""
    layout = scenes.DistanceFieldLayout(extent[0], extent[1], depth, slices, resolution, 128)
    obs = scenes.random_obstructions(seed, n, extent, size_lo=6.0, size_hi=40.0, z_hi=48.0, dynamic_fraction=dynamic_fraction)
    # a big rotated box whose corners poke out of DistanceFunctionVertexShader's quad (the cut is part of the semantics)
    obs.append((abi.OBSTRUCTION_BOX, (extent[0] * 0.5, extent[1] * 0.5, 10.0), (90.0, 90.0, 90.0), 0.78539816, False))
    volumes = [
        ([(20.0, 30.0), (90.0, 24.0), (100.0, 80.0), (60.0, 50.0), (24.0, 90.0)], 0.0, 30.0, True),    # concave
        ([(150.0, 100.0), (230.0, 110.0), (200.0, 170.0)], 8.0, 20.0, False),
    ]
    return layout, obs, volumes
