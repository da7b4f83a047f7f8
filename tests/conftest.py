import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run by the driver with -m gpu)")


def se3_err(A, B):
    """(translation error [m], rotation error [rad]) between two 4x4 transforms."""
    import numpy as np
    E = np.linalg.inv(np.asarray(A, np.float64)) @ np.asarray(B, np.float64)
    c = min(1.0, max(-1.0, (np.trace(E[:3, :3]) - 1.0) / 2.0))
    # small-angle-safe rotation error
    w = np.array([E[2, 1] - E[1, 2], E[0, 2] - E[2, 0], E[1, 0] - E[0, 1]]) / 2.0
    ang = float(np.arctan2(np.linalg.norm(w), c))
    return float(np.linalg.norm(E[:3, 3])), ang


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
