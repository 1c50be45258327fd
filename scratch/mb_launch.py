"""How fast can the GPU retire trivially small kernels from replayed graphs? (dispatch floor)"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.gpu_util import build_engine
eng, cfg, sd, batch, c, dev = build_engine("small/vog_spat")
for nst in (1, 2, 4, 8):
    slots = [eng.make_slot(dev, graph=True) for _ in range(nst)]
    streams = [torch.cuda.Stream() for _ in range(nst)]
    for i in range(40): slots[i % nst].launch(streams[i % nst])
    torch.cuda.synchronize()
    K = 800
    t0 = time.perf_counter()
    for i in range(K): slots[i % nst].launch(streams[i % nst])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K * 1e6
    print(f"streams {nst}: {dt:.1f} us per tiny forward")
# host-only cost of hipGraphLaunch: launch K graphs of a slot without sync in between is what we did; report host issue time
slot = slots[0]
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(200): slot.launch(streams[0])
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"host issue time per graph launch: {(t1-t0)/200*1e6:.1f} us")
