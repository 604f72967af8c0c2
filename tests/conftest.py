import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Two HIP runtimes live in a GPU test process: the one bundled with torch and /opt/rocm's, which libtsgpu.so links. torch's must
    # initialise FIRST (observed on the GPU box: after libtsgpu.so has opened the device, torch.cuda's lazy init reports "No HIP GPUs
    # are available"); tests that hand CUDA tensors to the library (shard merge, synthetic corpora built on the device) depend on it.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
            torch.zeros(1, device="cuda")
    except Exception:   # no torch / no GPU: the CPU tier does not need it
        pass
