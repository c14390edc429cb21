import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Multi-process tests start their ranks from a fork server that is launched HERE, before any test has touched the GPU:
    # a process that has initialised HIP must never exec (the GPU boxes refuse it), and fork-without-exec of such a process
    # would hand the children a stale HIP state.
    import multiprocessing
    import multiprocessing.forkserver as forkserver
    try:
        multiprocessing.set_forkserver_preload(["torch"])
        forkserver.ensure_running()
    except Exception:  # noqa: BLE001 - platforms without forkserver: the multi-process tests fall back to spawn
        pass


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    import torch

    def load(name):
        with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
            return {k: torch.from_numpy(z[k]) for k in z.files}
    return load
