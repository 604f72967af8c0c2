"""Generates tests/golden/match_vectors.npz by running the REFERENCE's include/match_score.h
(through oracle/_ref/libref_match.so, built by oracle/Makefile from /root/reference) on seeded random
token-position lists. Runs only in the build container; the .npz is committed."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O

R = O.ref_match_lib()
assert R is not None, "build oracle/_ref first (make -C oracle ref)"
rng = np.random.default_rng(2024)
P, LENS, LAST, NT, CE, EXP = [], [], [], [], [], []
for n_tokens in (2, 3, 3, 3, 4, 6, 10, 11):
    for max_pos, max_len in ((8, 3), (33, 4), (400, 6)):
        for _ in range(40):
            lens, pos, last = [], [], []
            for _t in range(n_tokens):
                n = int(rng.integers(1, max_len + 1))
                p = np.sort(rng.choice(max_pos, size=min(n, max_pos), replace=False)).astype(np.uint16)
                lens.append(p.size); pos.append(p); last.append(int(rng.integers(0, 4) == 0))
            pos = np.concatenate(pos); lens = np.array(lens, np.uint32); last = np.array(last, np.uint8)
            for ce in (0, 1):
                out = np.zeros(4, np.uint8)
                R.ref_match(pos.ctypes.data, lens.ctypes.data, last.ctypes.data, n_tokens, ce, out.ctypes.data)
                P.append(pos); LENS.append(lens); LAST.append(last); NT.append(n_tokens); CE.append(ce); EXP.append(out.copy())
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "match_vectors.npz"),
                    positions=np.concatenate(P), lens=np.concatenate(LENS), last=np.concatenate(LAST),
                    n_tokens=np.array(NT, np.uint32), check_exact=np.array(CE, np.uint8), expect=np.stack(EXP))
print("wrote", len(NT), "cases")
