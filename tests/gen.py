"""Synthetic clouds used by the tests: the generators live in the package (pointclouds_rs.synth)
so that the benchmark uses exactly the same data."""
from pointclouds_rs.synth import *  # noqa: F401,F403
from pointclouds_rs.synth import F  # noqa: F401
