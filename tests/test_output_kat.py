"""Pins the oracle's FillReadbackResult / lightmap resolve restatement (oracle/ilm_oracle_output.c) on the hand-evaluated
cases of tests/golden/output_ext.json.  No GPU."""
import pytest

from tests import output_common as oc


@pytest.mark.parametrize("index", range(len(oc.load_cases())))
def test_closed_form_case(oracle, index):
    oc.check_case(oc.load_cases()[index], oc.OracleBackend(oracle))
