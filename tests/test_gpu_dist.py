"""`-m gpu`, two tests. (1) tests/dist_worker.py with the REAL libtsgpu.so: two and three ranks share the one MI355X and drive the product's
rank-form exchange (tsgpu_group_create_rank_host: bounds, bound-pruned and full packed blocks, slice / all-gather exchanges, merge kernels,
replicas form, the agreement step) over gloo callbacks — every rank equals the unsharded oracle bit for bit. (2) the N>1 path of bench.py on ONE MI355X — two and EIGHT ranks share the device (collectives through gloo; the measured
configuration is RCCL, one GPU per rank), the 2M-doc collection is cut into two doc-range shards of the SAME corpus, and the
bench's own in-run equality checks (merged shard results == the unsharded collection: keyword top-100 + counts, k-NN labels +
distance bits, fused hybrid scores) must report zero mismatches. BASELINE config 5 / SURVEY §8(e)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
def test_n_rank_shards_equal_unsharded_collection(world):
    """world 8 = the exact command line the driver's SCALE run uses (`bench.py --gpus 8` under torch.distributed.run), rehearsed with the eight ranks sharing the
    one MI355X and the group's exchange over the HOST transport: the first 8-GPU run is not the first execution of the 8-rank bench path"""
    env = dict(os.environ, TSGPU_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(29613 + world),
           os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", "--n-docs", "2000000", "--batch", "1000", "--vec-batch", "64",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == world and r["scaling"] == "strong" and r["distributed"]["world_size"] == world and r["distributed"]["mode"] == "shards"
    assert r["shard_parity"]["checked"] >= 256 and r["shard_parity"]["mismatches"] == 0, r["shard_parity"]
    assert r["vector"]["shard_parity"]["mismatches"] == 0, r["vector"]["shard_parity"]
    assert r["hybrid"]["shard_parity"]["mismatches"] == 0, r["hybrid"]["shard_parity"]
    assert r["replicas"]["value"] > 0 and r["value"] > 0
    assert r["distributed"]["group_transport"] == "host" and "tsgpu_group" in r["config"]["parallelism"], r["distributed"]
    assert r["exchange_check"]["pruned_equals_full_exchange"] is True, r.get("exchange_check")      # bound-pruned exchange == the full top-k exchange
    assert r["exchange_check"]["hit_exchange_bytes_per_gpu"] > 0
    assert r["exchange_check"]["own_slice_delivery_equals_full_result"] is True, r.get("exchange_check")      # the timed form: every rank delivers the slice it merged
    assert r["candidate_combinations_sharded"]["shard_parity"]["mismatches"] == 0 and r["candidate_combinations_sharded"]["shard_parity"]["checked"] >= 8, r.get("candidate_combinations_sharded")
    assert r["wildcard_sharded"]["shard_parity"]["mismatches"] == 0 and r["wildcard_sharded"]["shard_parity"]["checked"] == 4, r.get("wildcard_sharded")      # q = * over the doc-range shards
    assert r["group_by_sharded"]["shard_parity"]["mismatches"] == 0 and r["group_by_sharded"]["shard_parity"]["checked"] >= 16, r.get("group_by_sharded")      # group_by, both passes, keyed exchange


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3, 8])
def test_rank_form_group_on_the_real_library(world):
    from tests import helpers as H
    from tests.test_dist_gloo import run_ranks
    run_ranks(world, H.gpu_lib_path(), extra_env={"TSGPU_WORKER_DOCS": "30000", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
