"""The CPU oracle's particle rasteriser against the closed-form fixture tests/golden/rasterize.json."""
import pytest

from tests import raster_common as rc


@pytest.mark.parametrize("index", range(len(rc.load_cases())))
def test_closed_form_case(oracle, index):
    rc.check_case(rc.load_cases()[index], rc.OracleBackend(oracle))


def test_dithered_opacity_closed_form(oracle):
    """premultipliedToDithered (RasterizeParticleSystem.fx:158-175): a fragment whose alpha is <= Dither64(vpos, floor(index % 4)) or
    <= 6/255 vanishes, every other one becomes (rgb / alpha, 1).  Dither64 = frac((33 x + 52 y + 25 f) / 64) (Fracture's
    DitherCommon.fxh, outside the tree: the function Jimenez published, SIGGRAPH 2014), vpos = the integer pixel."""
    import numpy as np
    from illuminant_amd import abi, scenes
    from tests.raster_common import chunk_from_particles
    w, h = 24, 20
    for slot in (0, 1, 6, 255):
        for alpha in (0.5, 0.02, 1.0):
            color = [0.2 * alpha, 0.4 * alpha, 0.6 * alpha, alpha]        # premultiplied
            chunks = [chunk_from_particles([dict(slot=slot, position=[12.0, 10.0, 0.0, 1.0], size=8.0, rotation=0.0, color=color)])]
            params = scenes.rasterize_params(dithered_opacity=True)
            image = np.zeros((h, w, 4), np.float32)
            image, (live, shaded) = oracle.render_particles(chunks, params, w, h, image=image)
            jj, ii = np.mgrid[0:h, 0:w]
            d64 = np.mod((33 * ii + 52 * jj + 25 * (slot % 4)) / 64.0, 1.0)
            inside = (ii >= 4) & (ii < 20) & (jj >= 2) & (jj < 18)
            keep = inside & (alpha > d64) & (alpha > 6.0 / 255.0)
            assert live == 1 and shaded == int(keep.sum())
            want = np.zeros((h, w, 4), np.float32)
            want[keep] = [0.2, 0.4, 0.6, 1.0]
            assert np.allclose(image, want, rtol=0, atol=1e-6), (slot, alpha)
            if alpha == 0.5:
                assert 0.4 < keep.sum() / inside.sum() < 0.6       # an ordered dither: half the pixels at half opacity
            if alpha == 0.02:
                assert keep.sum() == 0                             # below 6 / 255: plain invisible
