"""Pins the CPU oracle against every known-answer test the reference holds for the hot path."""
import pytest

import ref_cases


@pytest.mark.parametrize("algo", [0, 1], ids=["bruteforce", "grid"])
@pytest.mark.parametrize("name,fn", ref_cases.ALL_CASES, ids=[n for n, _ in ref_cases.ALL_CASES])
def test_reference_case_oracle(oracle, name, fn, algo):
    fn(ref_cases.OracleImpl(oracle, algo))
