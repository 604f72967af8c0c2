"""GPU experiment: HNSW traversal on a knn-heuristic graph (typesense_amd/hnsw_synth.py) — build time, q/s by batch, recall vs exact.
usage: [EF=100,400] [PARITY=64] python tools/exp_hnsw.py [n_rows] [dim] [batches]
PARITY=n: also load rows + graph into the CPU oracle and compare labels, order and distance bits of the first n queries."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import typesense_amd as T  # noqa: E402
from typesense_amd import _lib as B, synth, hnsw_synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
batches = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "256,1024,4096").split(",")]
k, M = 100, 16
EFS = [int(x) for x in os.environ.get("EF", "100").split(",")]
g = T.GpuIndex(0, os.environ.get("TSGPU_LIB") or None)
g.vec_create(1, dim, B.METRIC_IP, n)
DATA = os.environ.get("DATA", "latent")
mk = (lambda m, sd: synth.latent_vectors(m, dim, seed=sd, device="cuda")) if DATA == "latent" else (lambda m, sd: synth.random_vectors(m, dim, seed=sd, device="cuda", normalize=True))
X = mk(n, 3)
lab = torch.arange(n, dtype=torch.int64, device="cuda")
g.vec_upsert_device(1, lab.data_ptr(), X.data_ptr(), n)
torch.cuda.synchronize()
t0 = time.perf_counter()
graph = hnsw_synth.build_graph(torch, g, 1, X, M=M, K0=int(os.environ.get("K0", "64")), seed=100, batch=1024, log=lambda m: print(m, flush=True))
torch.cuda.synchronize()
print("graph built in %.1f s: maxlevel %d, mean level-0 degree %.1f, upper lists %d" % (time.perf_counter() - t0, graph["maxlevel"], graph["link0"][:, 0].mean(), graph["upper_links"].shape[0]), flush=True)
Q = mk(max(batches), 4)
for gib, nq, ef in [(g_, b_, e_) for g_ in [int(x) for x in os.environ.get("TAG_GIB", "16").split(",")] for e_ in EFS for b_ in batches]:
    if (nq, ef) == (batches[0], EFS[0]):
        g.set_option("hnsw_visited_max_gib", gib)
        torch.cuda.empty_cache()
        print("HBM free before load: %.1f GB" % (torch.cuda.mem_get_info()[0] / 1e9), flush=True)
        t0 = time.perf_counter(); g.vec_hnsw_load(1, graph); print("visited tags <= %d GiB: load %.2f s" % (gib, time.perf_counter() - t0), flush=True)
    d = torch.zeros((nq, k), dtype=torch.float32, device="cuda"); l = torch.zeros((nq, k), dtype=torch.int64, device="cuda"); c = torch.zeros(nq, dtype=torch.int32, device="cuda")
    de = torch.zeros_like(d); le = torch.zeros_like(l); ce = torch.zeros_like(c)
    for _ in range(2):
        g.vec_hnsw_search_batch_raw(1, Q.data_ptr(), B.MEM_DEVICE, nq, k, ef, d.data_ptr(), l.data_ptr(), c.data_ptr(), B.MEM_DEVICE)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        g.vec_hnsw_search_batch_raw(1, Q.data_ptr(), B.MEM_DEVICE, nq, k, ef, d.data_ptr(), l.data_ptr(), c.data_ptr(), B.MEM_DEVICE)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    g.vec_knn_batch_raw(1, Q.data_ptr(), B.MEM_DEVICE, nq, k, de.data_ptr(), le.data_ptr(), ce.data_ptr(), B.MEM_DEVICE)
    torch.cuda.synchronize()
    hits = 0
    lc, lec = l.cpu().numpy(), le.cpu().numpy()
    for i in range(min(nq, 256)):
        hits += len(set(lc[i].tolist()) & set(lec[i].tolist()))
    print("B=%d ef=%d: %.3f ms/batch  %.0f q/s  recall@%d %.4f (first %d queries)  overflowed %d  expansions/query %.0f distances/query %.0f" % (nq, ef, dt * 1e3, nq / dt, k, hits / (min(nq, 256) * k), min(nq, 256), int((c.cpu().numpy() == 0xFFFFFFFF).sum()), g.counter("hnsw_last_expansions") / nq, g.counter("hnsw_last_distances") / nq), flush=True)
npar = int(os.environ.get("PARITY", "0"))
if npar:
    from oracle import oracle_py as O
    t0 = time.perf_counter()
    orc = O.OracleIndex(1, 1)
    orc.vec_init(dim, O.METRIC_IP)
    for a in range(0, n, 1 << 18):
        b = min(n, a + (1 << 18))
        orc.vec_add(np.arange(a, b, dtype=np.uint32), X[a:b].cpu().numpy())
    orc.hnsw_import(graph)
    print("oracle load + import %.0f s" % (time.perf_counter() - t0), flush=True)
    Qh = Q[:npar].cpu().numpy()
    ncpu = os.cpu_count() or 1
    for ef in EFS:
        d = torch.zeros((npar, k), dtype=torch.float32, device="cuda"); l = torch.zeros((npar, k), dtype=torch.int64, device="cuda"); c = torch.zeros(npar, dtype=torch.int32, device="cuda")
        g.vec_hnsw_search_batch_raw(1, Q.data_ptr(), B.MEM_DEVICE, npar, k, ef, d.data_ptr(), l.data_ptr(), c.data_ptr(), B.MEM_DEVICE)
        torch.cuda.synchronize()
        gd, gl, gc = d.cpu().numpy(), l.cpu().numpy(), c.cpu().numpy()
        t0 = time.perf_counter()
        od, ol, oc = orc.hnsw_search_batch(Qh, k, ef, threads=ncpu)
        wall = time.perf_counter() - t0
        bad = 0
        for i in range(npar):
            m = int(oc[i])
            if not (gc[i] == m and np.array_equal(gl[i, :m].astype(np.uint64), ol[i, :m]) and np.array_equal(gd[i, :m].view(np.uint32), od[i, :m].view(np.uint32))):
                bad += 1
        print("parity ef=%d: %d queries vs the oracle's traversal of the same graph, mismatches %d (oracle %.0f q/s on %d threads)" % (ef, npar, bad, npar / wall, ncpu), flush=True)
    orc.close()
g.close()
