"""The example scripts run end to end on the GPU and reach a sensible fidelity in a short budget."""
import contextlib
import io
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'examples'))

pytestmark = pytest.mark.gpu


def test_qubit_pi_pulse_example():
    import qubit_pi_pulse
    with contextlib.redirect_stdout(io.StringIO()):
        f = qubit_pi_pulse.main(iterations=150, quiet=True)
    assert f > 0.99


def test_transmon_state_transfer_example():
    import transmon_state_transfer
    with contextlib.redirect_stdout(io.StringIO()):
        f = transmon_state_transfer.main(iterations=200, restarts=4, quiet=True)
    assert f > 0.9


def test_two_transmon_cz_example():
    """One control set, n = 9, forbidden levels + dwdt: the latency mode with the affine backward half, end to end through Grape()."""
    import two_transmon_cz
    with contextlib.redirect_stdout(io.StringIO()):
        f = two_transmon_cz.main(iterations=400, quiet=True)
    assert f > 0.999


def test_three_transmon_cz_example():
    """n = 27, six controls, 19 forbidden levels + dwdt: a batch of 16 restarts on the batch kernels (active strips 7 of 8, affine costate), and one
    control set in the latency mode -- end to end through Grape(), checked by re-simulating the returned pulse with exact propagators."""
    import three_transmon_cz
    with contextlib.redirect_stdout(io.StringIO()):
        f16 = three_transmon_cz.main(iterations=800, restarts=16, quiet=True)
        f1 = three_transmon_cz.main(iterations=400, restarts=1, quiet=True)
    assert f16 > 0.995 and f1 > 0.99
