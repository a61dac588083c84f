"""The CPU oracle's particle rasteriser against the closed-form fixture tests/golden/rasterize.json."""
import pytest

from tests import raster_common as rc


@pytest.mark.parametrize("index", range(len(rc.load_cases())))
def test_closed_form_case(oracle, index):
    rc.check_case(rc.load_cases()[index], rc.OracleBackend(oracle))
