import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ref_trace():
    """Golden vectors captured from the unmodified reference ppo.py (oracle/capture_ppo_trace.py)."""
    import numpy as np

    with np.load(os.path.join(ROOT, "tests", "golden", "ppo_ref_trace.npz")) as z:
        return {k: z[k] for k in z.files}  # materialise once: NpzFile re-inflates on every __getitem__
