"""examples/closed_loop.py: the reference's example loops (examples/example_point_mass.py:88-101 and its siblings) on the drop-in class, run as a user would run
them; at a tight tolerance the trajectory must be the one the reference classes themselves produce (tests/golden/traj_*.npz)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(300)
@pytest.mark.parametrize('system', ['point_mass', 'cart_pole', 'quadcopter'])
def test_example_loop_reproduces_the_reference_trajectory(system):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'examples', 'closed_loop.py'), system, '--eps', '1e-10', '--device-loop', '64'],
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    m = re.search(r'largest distance to the reference .*: x ([0-9.e+-]+), u ([0-9.e+-]+)', r.stdout)
    assert m, r.stdout
    assert float(m.group(1)) <= 1e-6 and float(m.group(2)) <= 1e-6, r.stdout
    assert '0 solves not "solved"' in r.stdout
    d = re.search(r'device loop: 64 controllers x (\d+) steps .* (\d+) of (\d+) solves "solved"', r.stdout)
    assert d and d.group(2) == d.group(3), r.stdout


@pytest.mark.timeout(300)
def test_output_feedback_example_reproduces_the_reference_run():
    """examples/closed_loop_kalman.py: the reference's Kalman example (Np = 200) on the drop-in classes against the reference classes' own recorded run."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'examples', 'closed_loop_kalman.py'), '--eps', '1e-10', '--device-loop', '32'], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    m = re.search(r'largest distance to the reference .*: x ([0-9.e+-]+), u ([0-9.e+-]+)', r.stdout)
    assert m, r.stdout
    assert float(m.group(1)) <= 1e-6 and float(m.group(2)) <= 1e-6, r.stdout
    d = re.search(r'device loop .*: 32 controllers x (\d+) steps .* (\d+) of (\d+) solves "solved"', r.stdout)
    assert d and d.group(2) == d.group(3), r.stdout
