"""GPU experiment: what a plain streaming read achieves on this box (yardstick for the scan kernels)."""
import time, torch
n = 10_000_000 * 768
x = torch.empty(n, dtype=torch.bfloat16, device="cuda").normal_()
xi = x.view(torch.int32)
for name, fn in (("sum(int32 view)", lambda: xi.sum()), ("max(bf16)", lambda: x.max()), ("amax int32", lambda: xi.amax()), ("copy 7.68GB", lambda: xi[: n // 4].clone())):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    nbytes = n * 2 if "copy" not in name else n // 4 * 4 * 2
    print("%s: %.3f ms  %.2f TB/s" % (name, dt * 1e3, nbytes / dt / 1e12), flush=True)
