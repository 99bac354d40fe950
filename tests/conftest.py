import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs the compiled reference under oracle/_ref (this container only)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    import harness as H

    def load(eqset):
        return np.load(os.path.join(H.GOLDEN_DIR, f"{H.EQ_NAMES[eqset]}_small.npz"))
    return load
