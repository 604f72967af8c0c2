"""The SIMT emulator's own checks (tests/hipemu/hip/hip_runtime.h), without which "the kernels pass on the emulator" would say less than it does:
threads of a workgroup that meet at DIFFERENT textual __syncthreads() — the signature of a race on the way into barrier-carrying code, which a
cooperative schedule would otherwise hide and the hardware only sometimes shows — are reported, and fatal under HIPEMU_STRICT_BARRIERS=1
(tests/conftest.py sets it for the whole CPU tier)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = os.path.join(str(tmp_path), "selftest_barriers")
    cxx = "/opt/rocm/lib/llvm/bin/clang++" if os.path.exists("/opt/rocm/lib/llvm/bin/clang++") else "g++"
    subprocess.check_call([cxx, "-x", "c++", "-std=c++17", "-O1", "-DTSGPU_HIP_EMU=1", "-Wno-unused-value", "-I", os.path.join(ROOT, "tests", "hipemu"),
                           "-o", exe, os.path.join(ROOT, "tests", "hipemu", "selftest_barriers.cpp"), "-lpthread"])
    return exe


def test_emulator_reports_threads_meeting_at_different_barriers(tmp_path):
    exe = _build(tmp_path)
    env = dict(os.environ, HIPEMU_STRICT_BARRIERS="1")
    clean = subprocess.run([exe, "clean"], capture_output=True, text=True, env=env)
    assert clean.returncode == 0 and clean.stdout.startswith("done") and "DIFFERENT" not in clean.stderr, clean.stderr
    race = subprocess.run([exe, "race"], capture_output=True, text=True, env=env)
    assert race.returncode != 0 and "DIFFERENT __syncthreads() call sites" in race.stderr, (race.returncode, race.stderr)
    # without the strict switch the race is reported and the run continues
    env.pop("HIPEMU_STRICT_BARRIERS")
    lax = subprocess.run([exe, "race"], capture_output=True, text=True, env=env)
    assert "DIFFERENT __syncthreads() call sites" in lax.stderr
