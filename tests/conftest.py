import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle (torch-CPU fp32 / fp64 through oneDNN) is the checker of every parity test.  On a many-core host oneDNN
    # oversubscribes itself on this graph: the config-4 oracle step at B = 32 takes 43.7 s with 128 threads, 15.7 s with 64
    # and 9.4 s with 32 on the GPU box's 2 x 64-core host (round 6, scripts-level probe) -- the same effect bench.py's
    # cpu_baseline sweep shows.  Cap the checker's threads; results are compared with tolerances that do not depend on it.
    import torch
    torch.set_num_threads(max(1, min(32, torch.get_num_threads())))


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
