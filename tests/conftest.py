import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Two HIP runtimes live in a GPU test process: the system ROCm libc8p_rx.so links and the one bundled in the torch
    wheel.  torch must initialise its runtime BEFORE the engine creates a context with the other one (the other order
    has made torch report "No HIP GPUs are available"), so do it once, up front, whenever a GPU test is about to run."""
    if any(it.get_closest_marker("gpu") for it in items):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
                torch.zeros(1, device="cuda")
        except Exception:
            pass
