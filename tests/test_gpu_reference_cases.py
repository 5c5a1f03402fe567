"""The reference's unit tests for the hot path, run against the HIP library through its C-ABI."""
import pytest

import ref_cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,fn", ref_cases.ALL_CASES, ids=[n for n, _ in ref_cases.ALL_CASES])
def test_reference_case_hip(name, fn):
    fn(ref_cases.HipImpl())
