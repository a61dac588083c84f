"""Pins the oracle's restatement of the remaining particle techniques (oracle/ilm_oracle_transforms.c) on the
hand-derived closed forms of tests/golden/transforms_ext.json.  No GPU."""
import numpy as np
import pytest

from illuminant_amd import abi, scenes
from tests import transforms_common as tc


@pytest.mark.parametrize("index", range(len(tc.load_cases())))
def test_closed_form_case(oracle, index):
    tc.check_case(tc.load_cases()[index], tc.OracleBackend(oracle))


def test_low_precision_table_is_round_half_even_unorm16(oracle):
    """new Rgba64(Vector4) (ParticleEngine.cs:536-538): round(clamp(v) * 65535) with ties to even."""
    rnd = np.zeros((1, 4, 4), np.float32)
    rnd[0, 0] = [0.0, 1.0, 0.5, 2.0]                                    # 0.5 * 65535 = 32767.5 -> 32768 (even), clamp(2) -> 65535
    rnd[0, 1] = [-1.0, 1.5 / 65535.0, 2.5 / 65535.0, 0.25]              # ties: 1.5 -> 2, 2.5 -> 2
    lp = oracle.low_precision_randomness(rnd)
    assert lp[0, 0].tolist() == [0, 65535, 32768, 65535]
    assert lp[0, 1].tolist() == [0, 2, 2, 16384]


def test_spatial_noise_is_bilinear_in_position(oracle):
    """On a table that is a linear ramp in x the bilinear sample is linear in the particle's x (away from the wrap seam)."""
    rnd = np.zeros((abi.RANDOMNESS_HEIGHT, abi.RANDOMNESS_WIDTH, 4), np.float32)
    rnd[..., :] = (np.arange(abi.RANDOMNESS_WIDTH, dtype=np.float32) / 1024.0)[None, :, None]
    n = tc.CS * tc.CS
    pos = np.zeros((n, 4), np.float32)
    pos[:, 0] = 100.0 + np.arange(n) * 0.37
    pos[:, 3] = 1.0
    vel = np.zeros((n, 4), np.float32)
    vel[:, 0] = 1.0
    d = tc.base_desc(1.0 / 60.0)
    d.OpCount = 1
    d.Ops[0].Type = abi.OP_SPATIAL_NOISE
    noise = scenes.noise_params(scenes.area_none(1.0), (0.0, 0.0), (0.0, 0.0), 0.0, 10.0, True,
                                position=((0,) * 4, (0,) * 4, (6.0, 0, 0, 0)), velocity=((0,) * 3, (0,) * 3, (0,) * 3), speed=(0, 0, 0))
    d.Ops[0].u.SpatialNoise = scenes.spatial_noise_params(noise, (1.0, 1.0))
    out = tc.OracleBackend(oracle).run(d, rnd, (pos, vel, np.ones((n, 4), np.float32)))
    # sample coordinate = x texels, centres at +0.5 => value(x) = (x - 0.5) / 1024; delta.x = value * 6, applied with t = 1/6
    want = pos[:, 0] + ((pos[:, 0] - 0.5) / 1024.0)
    assert np.allclose(out[0][:, 0], want, rtol=0, atol=2e-3)
