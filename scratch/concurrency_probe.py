"""How many kernels of different HIP streams execute at once: n streams each run one 1-ms single-workgroup spin kernel."""
import torch, time
torch.cuda.init()
streams = [torch.cuda.Stream() for _ in range(8)]
cyc = 2_000_000
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in streams[:n]:
        with torch.cuda.stream(s):
            torch.cuda._sleep(cyc)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
run(1)
one = min(run(1) for _ in range(3))
for n in (1, 2, 3, 4, 5, 6, 8):
    t = min(run(n) for _ in range(3))
    print(f"{n} streams x 1 spin kernel: {t:.2f} ms = {t / one:.2f} x one kernel")
