"""Host link probe: (1) back-to-back pinned H2D copies on an idle GPU by size and stream count; (2) the same copies issued next to a resident
4-stream cfg-2 forward loop with NO dependency between copies and forwards: does the copy engine keep its rate while the chip is busy?"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
dev = torch.device("cuda:0")
MB = 1 << 20
def rate(size, nstreams, reps):
    src = [torch.empty(size, dtype=torch.uint8).pin_memory() for _ in range(nstreams)]
    dst = [torch.empty(size, dtype=torch.uint8, device=dev) for _ in range(nstreams)]
    sts = [torch.cuda.Stream() for _ in range(nstreams)]
    for i in range(2 * nstreams):
        with torch.cuda.stream(sts[i % nstreams]): dst[i % nstreams].copy_(src[i % nstreams], non_blocking=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(reps):
        with torch.cuda.stream(sts[i % nstreams]): dst[i % nstreams].copy_(src[i % nstreams], non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return size * reps / dt / 1e9, dt / reps * 1e6
print("idle GPU: size MB, streams -> GB/s, us per copy")
for size in (8.5, 17, 34, 68, 133):
    for ns in (1, 2, 4):
        g, us = rate(int(size * MB), ns, max(8, int(2000 / size)))
        print(f"  {size:6.1f} MB x {ns} streams: {g:6.1f} GB/s  {us:8.1f} us")
# (2) next to the forward loop
from tests.gpu_util import build_engine
synth = importlib.import_module("vognet-pytorch_amd.synth")
bench = importlib.import_module("bench")
eng, cfg, sd, batch, c, devb = build_engine("full/cfg2_ragged")
slots = [eng.make_slot(devb, graph=True) for _ in range(4)]
sts = [torch.cuda.Stream() for _ in range(4)]
def loop(steps, copy_every, size, ncs):
    src = [torch.empty(size, dtype=torch.uint8).pin_memory() for _ in range(ncs)]
    dst = [torch.empty(size, dtype=torch.uint8, device=dev) for _ in range(ncs)]
    cs = [torch.cuda.Stream() for _ in range(ncs)]
    k = 0
    def run(n):
        nonlocal k
        for i in range(n):
            if copy_every and i % copy_every == 0:
                with torch.cuda.stream(cs[k % ncs]): dst[k % ncs].copy_(src[k % ncs], non_blocking=True)
                k += 1
            slots[i % 4].launch(sts[i % 4])
    run(80); torch.cuda.synchronize(); t0 = time.perf_counter(); run(steps); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6
print("forward loop alone: %.1f us/step" % loop(2000, 0, 1024, 1))
for g in (1, 2, 4, 8):
    for ncs in (1, 2):
        us = loop(2000, g, int(8.5 * MB) * g, ncs)
        print(f"forward loop + independent copies of {g} x 8.5 MB every {g} steps on {ncs} copy streams: {us:6.1f} us/step = {8.5 * MB / us / 1e3:5.1f} GB/s")
