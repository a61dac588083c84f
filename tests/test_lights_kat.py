"""Pins the oracle's particle-light / light-probe restatement (oracle/ilm_oracle_lights.c) on the hand-derived
closed forms of tests/golden/lights_ext.json.  No GPU."""
import pytest

from tests import lights_common as lc


@pytest.mark.parametrize("index", range(len(lc.load_cases())))
def test_closed_form_case(oracle, index):
    lc.check_case(lc.load_cases()[index], lc.OracleBackend(oracle))
