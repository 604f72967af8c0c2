"""The CPU oracle against the reference's own known-answer tests.

oracle/golden_tests.cpp restates, check by check, the reference's unit tests for the hot path
(sorted_array/array, posting_list, or_iterator, match_score, topster, the end-to-end text_match
constants, vector distances, rank fusion). This test builds and runs that binary.
"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_reproduces_reference_known_answers():
    from oracle import oracle_py
    oracle_py.build()
    exe = os.path.join(ROOT, "oracle", "_build", "golden_tests")
    p = subprocess.run([exe, os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-4000:] + p.stderr[-2000:]
    assert "0 failed" in p.stdout
