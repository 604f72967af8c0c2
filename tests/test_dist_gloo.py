"""N>1 path on CPU: two processes (gloo, 127.0.0.1), doc-range shards, all-gather of per-shard top-K, exact merge,
hybrid fusion after the merge — the code bench.py runs on RCCL (typesense_amd/dist.py), against the unsharded oracle."""
import os
import socket
import subprocess
import sys

from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_shards_allgather_merge_equals_unsharded_oracle():
    lib = H.emu_lib_path()
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TSGPU_EMU_LIB=lib,
                   GLOO_SOCKET_IFNAME="lo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "DIST_OK" in outs[0], outs[0]
