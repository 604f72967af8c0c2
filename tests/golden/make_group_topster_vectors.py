"""Writes tests/golden/group_topster_vectors.json from oracle/_ref/libref_topster.so — the reference's OWN include/topster.h, loglogbeta.h and
wyhash_v5.h compiled where they lie (oracle/Makefile `ref`; needs /root/reference, i.e. the build container). The fixture lets the oracle's
restatement (oracle/group_topster.h) be pinned where _ref cannot be built. Run: python tests/golden/make_group_topster_vectors.py"""
import json
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # noqa: E402

R = O.ref_topster_lib()
assert R is not None, "oracle/_ref/libref_topster.so missing: run `make -C oracle ref` where /root/reference exists"
rng = np.random.default_rng(77)
out = {"hash_wy": [], "hash_combine": [], "streams": []}
for v in [0, 1, 7, 9, 10, 99, 100, 12345, 4294967295, 4294967296, 2**53, 10**15, 10**16 - 1, 10**19, 2**64 - 1] + [int(x) for x in rng.integers(0, 2**63, 40)]:
    s = str(v)
    out["hash_wy"].append([s, str(R.ref_hash_wy(s.encode(), len(s)))])
for _ in range(40):
    a, b = int(rng.integers(0, 2**63)) * 2 + 1, int(rng.integers(0, 2**32))
    out["hash_combine"].append([str(a), str(b), str(R.ref_hash_combine(a, b))])
for cap, distinct, first_pass, n, n_groups, sr in [(5, 2, True, 60, 12, 3), (5, 2, False, 60, 12, 3), (16, 3, True, 400, 90, 4), (16, 3, False, 400, 90, 4),
                                                    (250, 1, True, 900, 600, 50), (250, 4, False, 900, 600, 50), (1, 1, True, 20, 5, 2), (3, 9, False, 50, 2, 1)]:
    keys = rng.permutation(n * 3)[:n].astype(np.uint64)
    dk = rng.integers(0, n_groups, n).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(1)
    sc = rng.integers(-sr, sr + 1, (n, 3)).astype(np.int64)
    ret, gsize, rdk, rkeys, rsc, rcount = O.ref_group_topster_run(R, cap, distinct, first_pass, keys, dk, sc)
    out["streams"].append({"capacity": cap, "distinct": distinct, "first_pass": first_pass, "keys": [int(x) for x in keys], "dkeys": [str(int(x)) for x in dk],
                           "scores": [int(x) for x in sc.ravel()], "ret": [int(x) for x in ret], "group_size": [int(x) for x in gsize],
                           "distinct_key": [str(int(x)) for x in rdk], "out_keys": [int(x) for x in rkeys], "groups_count": rcount})
with open(os.path.join(ROOT, "tests", "golden", "group_topster_vectors.json"), "w") as f:
    json.dump(out, f)
print("wrote", len(out["streams"]), "streams,", len(out["hash_wy"]), "hashes")
