"""The two example drivers (the reference's examples/driver_numpyVector.py and unittests/test_stateFollowingHO.py
shapes on the device backend) run end to end as a user would start them."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *args):
    p = subprocess.run([sys.executable, os.path.join(REPO, "examples", script), *args], capture_output=True, text=True,
                       timeout=600, cwd=REPO)
    assert p.returncode == 0, p.stderr[-2000:]
    return p.stdout


def test_driver_example_finds_the_eigenvalue_next_to_sigma(hip):
    out = _run("driver_hipVector.py")
    got = float(re.search(r"Eigenvalue nearest to sigma\s*::\s*([-\d.e+]+)", out).group(1))
    want = float(re.search(r"Actual eigenvalue nearest to sigma::\s*([-\d.e+]+)", out).group(1))
    assert abs(got - want) <= 1e-4 * abs(want)            # the reference example's own accuracy (inner solves at 1e-4)
    assert re.search(r"'isConverged': (np\.)?True", out), out


def test_state_following_example_runs(hip):
    out = _run("stateFollowingHO.py")
    m = re.search(r"followed state:\s*([-\d.e+]+)\s*reference:\s*([-\d.e+]+)\s*converged:\s*(\w+)", out)
    assert m, out
    assert m.group(3) == "True" and abs(float(m.group(1)) - float(m.group(2))) <= 1e-6 * abs(float(m.group(2)))
