"""N>1 path on CPU: the PRODUCT's rank-form exchange (tsgpu_group_create_rank_host -> tsgpu_group_keyword_search_batch /
_vec_knn_batch / _hybrid_search_batch) across PROCESSES on the SIMT-emulator build, the wire = torch.distributed gloo on 127.0.0.1
through the two host-collective callbacks: slice exchange and literal all-gather, an uneven cut and an EMPTY shard, padded query
slices, filters / curated hits, the replicas form, and the agreement step (one rank's bad argument fails the call on every rank
instead of hanging the others). Every rank compares with the unsharded oracle bit for bit (tests/dist_worker.py). SURVEY §8e."""
import os
import socket
import subprocess
import sys

import pytest

from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_ranks(world, lib, extra_env=None, timeout=900):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TSGPU_WORKER_LIB=lib,
                   GLOO_SOCKET_IFNAME="lo", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout)[0])
    finally:
        for p in procs:                       # a rank stuck in a collective must not outlive the test
            if p.poll() is None:
                p.kill()
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    for r, o in enumerate(outs):
        assert "rank %d: DIST_OK" % r in o, o


@pytest.mark.parametrize("world", [2, 8])          # (three ranks: the `-m gpu` tier, tests/test_gpu_dist.py)
def test_rank_form_group_over_gloo_equals_unsharded_oracle(world):
    """world 8 = the node BASELINE config 5 names: eight doc-range shards (uneven cut; a cut with an EMPTY shard), five queries over eight ranks (three ranks
    merge an empty, padded query slice), pruned and full exchange, slice exchange and literal all-gather, the replicas form, the agreement step"""
    run_ranks(world, H.emu_lib_path(), timeout=1500)
