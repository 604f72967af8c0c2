import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the SIMT emulator aborts when the threads of a workgroup meet at different textual barriers (tests/test_hipemu_selftest.py)
    os.environ.setdefault("HIPEMU_STRICT_BARRIERS", "1")
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


def _gpu_tier(config):
    """True when the run selects the GPU tier (-m gpu / -m "gpu and ..."), not the CPU tier (-m "not gpu")"""
    m = (config.getoption("markexpr") or "").replace(" ", "")
    return "gpu" in m and "notgpu" not in m


def _emulator_mapped():
    try:
        with open("/proc/self/maps") as f:
            return sorted({ln.split()[-1] for ln in f if "libtsgpu_emu" in ln})
    except OSError:
        return []


@pytest.fixture(autouse=True)
def _no_emulator_in_gpu_tier(request):
    """GPU-tier guard (round-2 leak: a module-scoped fixture resolved the CPU emulator inside a -m gpu test): under -m gpu no
    test may finish with tests/hipemu's libtsgpu_emu*.so mapped into the process — it fails the test that mapped it."""
    yield
    if _gpu_tier(request.config) and request.node.get_closest_marker("gpu"):
        mapped = _emulator_mapped()
        assert not mapped, "GPU-tier test ran with the CPU emulator mapped: %s" % mapped


def pytest_sessionfinish(session, exitstatus):
    if _gpu_tier(session.config):
        mapped = _emulator_mapped()
        if mapped:
            session.exitstatus = 1
            sys.stderr.write("\nFAILED: -m gpu session has the CPU emulator mapped: %s\n" % mapped)
