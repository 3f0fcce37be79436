"""pytest configuration: the `gpu` marker, import paths, fixtures for the golden vectors."""
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "tests"))
sys.path.insert(0, str(REPO / "n-body-sim_amd"))

GOLDEN = REPO / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    # our own files; allow_pickle stays False
    with np.load(GOLDEN / f"{name}.npz") as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_pos_err(r, r_ref):
    """max_i |dr_i| / |r_i| with vector norms (component-wise is ill-conditioned)."""
    r = np.asarray(r, dtype=np.float64)
    r_ref = np.asarray(r_ref, dtype=np.float64)
    num = np.linalg.norm(r - r_ref, axis=-1)
    den = np.linalg.norm(r_ref, axis=-1)
    den = np.where(den == 0, 1.0, den)
    return float(np.max(num / den))
