"""oracle/match_window.h (restated Match) vs the REFERENCE's own include/match_score.h.

oracle/_ref/libref_match.so is compiled from the reference header where it lies (oracle/Makefile `ref`).
It exists in the build container and travels with the snapshot; where it is missing the comparison is
skipped and the committed golden vectors (tests/golden/match_vectors.npz, generated from _ref by
tests/golden/make_match_vectors.py) are used instead.
"""
import os
import numpy as np
import pytest
from oracle import oracle_py as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _random_case(rng, n_tokens, max_pos, max_len):
    lens, pos, last = [], [], []
    for _ in range(n_tokens):
        n = int(rng.integers(1, max_len + 1))
        p = np.sort(rng.choice(max_pos, size=min(n, max_pos), replace=False)).astype(np.uint16)
        lens.append(p.size); pos.append(p); last.append(int(rng.integers(0, 4) == 0))
    return np.concatenate(pos), np.array(lens, np.uint32), np.array(last, np.uint8)


def _run(fn, pos, lens, last, check_exact):
    out = np.zeros(4, np.uint8)
    fn(pos.ctypes.data, lens.ctypes.data, last.ctypes.data, lens.size, check_exact, out.ctypes.data)
    return tuple(int(x) for x in out)


def test_match_restatement_equals_reference_header_on_random_inputs():
    R = O.ref_match_lib()
    if R is None:
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    L = O.lib()
    rng = np.random.default_rng(11)
    n = 0
    for n_tokens in (2, 3, 4, 5, 8, 10, 12):
        for max_pos, max_len in ((6, 3), (40, 4), (300, 6), (65535, 5)):
            for _ in range(300):
                pos, lens, last = _random_case(rng, n_tokens, max_pos, max_len)
                for ce in (0, 1):
                    assert _run(L.orc_match, pos, lens, last, ce) == _run(R.ref_match, pos, lens, last, ce), (pos, lens, last, ce)
                    n += 1
    assert n > 10000


def test_match_duplicate_query_tokens_and_wraparound():
    """same token twice in the query (identical position lists) and uint16 wrap-around (match_score.h:164-167)"""
    R = O.ref_match_lib()
    L = O.lib()
    cases = [
        (np.array([0, 1, 0, 1], np.uint16), np.array([2, 2], np.uint32), np.array([1, 1], np.uint8)),
        (np.array([5, 5, 5], np.uint16), np.array([1, 1, 1], np.uint32), np.array([0, 0, 0], np.uint8)),
        (np.array([65530, 3, 65531, 4], np.uint16), np.array([2, 2], np.uint32), np.array([0, 0], np.uint8)),
    ]
    for pos, lens, last in cases:
        for ce in (0, 1):
            got = _run(L.orc_match, pos, lens, last, ce)
            if R is not None:
                assert got == _run(R.ref_match, pos, lens, last, ce)
    # MatchScoreV2 pins (test/match_score_test.cpp:30-45): words 4, distance 3
    pos = np.array([25, 26, 11, 18, 24, 60, 14, 27, 63], np.uint16)
    assert _run(L.orc_match, pos, np.array([1, 1, 4, 3], np.uint32), np.zeros(4, np.uint8), 0)[:2] == (4, 3)


def test_match_golden_vectors_committed():
    """fixtures generated from the reference header; lets the GPU box (no _ref rebuild) still pin the oracle"""
    path = os.path.join(ROOT, "tests", "golden", "match_vectors.npz")
    z = np.load(path)
    L = O.lib()
    off = 0
    loff = 0
    for i in range(z["n_tokens"].size):
        nt = int(z["n_tokens"][i])
        lens = z["lens"][loff:loff + nt]
        tot = int(lens.sum())
        pos = np.ascontiguousarray(z["positions"][off:off + tot])
        last = np.ascontiguousarray(z["last"][loff:loff + nt])
        got = _run(L.orc_match, pos, np.ascontiguousarray(lens), last, int(z["check_exact"][i]))
        assert got == tuple(int(x) for x in z["expect"][i])
        off += tot
        loff += nt


def test_match_score_packing_equals_reference():
    R = O.ref_match_lib()
    L = O.lib()
    rng = np.random.default_rng(5)
    for _ in range(2000):
        a = [int(rng.integers(0, 11)), int(rng.integers(0, 11)), int(rng.integers(0, 256)), int(rng.integers(0, 2)),
             int(rng.integers(0, 20)), int(rng.integers(0, 11)), int(rng.integers(0, 2))]
        got = L.orc_match_score(*a)
        if R is not None:
            assert got == R.ref_match_score(*a)
    # 3 words, 3 unique, cost 0, distance 2, no exact, max_offset 255 (unused), synonym 1
    assert L.orc_match_score(3, 2, 255, 0, 0, 3, 1) == 0x303ff620001
