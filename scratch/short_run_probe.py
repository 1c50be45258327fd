"""Where the fixed ~156 us of a short timed region (K = 20 steps on 4 streams) go: host issue times of every launch and
the device-side start / end of every stream's sequence (events), for the issue orders tried."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
eng_mod = importlib.import_module("vognet-pytorch_amd.engine")
synth = importlib.import_module("vognet-pytorch_amd.synth")
ec = importlib.import_module("vognet-pytorch_amd.extended_config")
w = B.WORKLOADS["cfg2"]; cfg = B.make_cfg(w); nppf0 = ec.num_prop_per_frm(cfg)
comm = {"vocab_size": B.VOCAB, "detect_size": 431, "itod": {}, "wtoi": {"UNK": 1}, "num_prop_per_frm": nppf0}
eng = eng_mod.VogEngine(cfg, comm); eng.load_state_dict(synth.init_state_dict(cfg, B.VOCAB, seed=1))
NS = 4
streams = [torch.cuda.Stream() for _ in range(NS)]
slots = [eng.make_slot({k: torch.from_numpy(v) for k, v in synth.make_batch(w["conc"], w["B"], nppf0, vocab_size=B.VOCAB, seed=2000 + s).items()}, graph=True)
         for s in range(NS)]
def spin_us(us):
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e6 < us:
        pass


def run(K, order="rr", stagger=0.0):
    torch.cuda.synchronize()
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(NS)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(NS)]
    host = []
    t0 = time.perf_counter()
    seq = [i % NS for i in range(K)] if order == "rr" else [s for s in range(NS) for _ in range(K // NS)]
    first = set()
    for i, u in enumerate(seq):
        if u not in first:
            if first and stagger:
                spin_us(stagger)
            ev0[u].record(streams[u]); first.add(u)
        slots[u].launch(streams[u])
        host.append((time.perf_counter() - t0) * 1e6)
    for u in range(NS):
        ev1[u].record(streams[u])
    t_issue = (time.perf_counter() - t0) * 1e6
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e6
    st = [ev0[0].elapsed_time(ev0[u]) * 1e3 for u in range(NS)]
    en = [ev0[0].elapsed_time(ev1[u]) * 1e3 for u in range(NS)]
    return dt, t_issue, host, st, en
for stg in (0, 15, 30, 50, 70):
    res = []
    for rep in range(5):
        for _ in range(30): run(20)
        res.append(run(20, "rr", stg)[0])
    print(f"stagger {stg} us between the first launches: K=20 total {min(res):.0f} .. {max(res):.0f} us, median {sorted(res)[2]:.0f}")
for K in (20,):
    for order in ("rr",):
        for _ in range(30): run(20)
        dt, ti, host, st, en = run(K, order)
        print(f"K={K} {order}: total {dt:.0f} us ({dt/K:.1f}/step), host issued all after {ti:.0f} us; first 4 launches issued at {[round(h) for h in host[:4]]}; "
              f"stream starts {[round(x) for x in st]} ends {[round(x) for x in en]}")
