"""The two lane-per-query search kernels that return neighbour identities (one walk with tagged
keys, and the two-walk kernel it hands crowded cubes and tie runs to) must give the same lists,
masks and normals, bit for bit, on clouds far larger and more varied than the CPU oracle covers.
The kernel is chosen once per process (PCRS_ONE_WALK), so the comparison runs
scripts/diff_onewalk.py in two child processes, one after the other."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(one_walk, **more):
    env = dict(os.environ)
    env["PCRS_ONE_WALK"] = one_walk
    env.update(more)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "diff_onewalk.py")],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout.splitlines()


def test_one_walk_and_two_walk_kernels_agree():
    a = _run("1")
    b = _run("0")
    assert len(a) == len(b) and len(a) > 150
    for la, lb in zip(a, b):
        assert la == lb


def test_results_do_not_depend_on_the_grid():
    """Neighbour sets are exact whatever the cell size, so the same checksums must come out under
    other grid geometries (the volume aim for every cloud; smaller and larger cells)."""
    a = _run("1")
    for more in ({"PCRS_GRID_DIM": "0"}, {"PCRS_GRID_OCC": "5"}, {"PCRS_GRID_OCC": "12"}):
        b = _run("1", **more)
        assert a == b, more
