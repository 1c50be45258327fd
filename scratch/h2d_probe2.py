"""Why does the host-fed cfg-2 loop run at 235 us/step when independent copies + forwards run at 156 (link bound)? The fed loop (graph reads
the device staging buffer; copy engine on copy streams) with its two dependencies toggled."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from tests.gpu_util import build_engine
dls = importlib.import_module("vognet-pytorch_amd.dat_loader_simple")
synth = importlib.import_module("vognet-pytorch_amd.synth")
eng, cfg, sd, batch, c, devb = build_engine("full/cfg2_ragged")
B = batch["num_cmp_msk"].shape[0]
asm = dls.DeviceBatchAssembler(cfg, {"num_prop_per_frm": c["nppf0"]})
lang_keys = ("srl_arg_words_ind", "srl_arg_word_mask", "srl_arg_word_mask_len", "srl_arg_words_capture", "srl_arg_inds_msk", "num_cmp_msk")
it = synth.make_items(B, 4, c["nppf0"], seed=3)
NS = 4
def build(via, n_extra=0):
    slots = [eng.make_slot(devb, graph=True) for _ in range(NS)]
    stg = [dls.PackedStaging({**{k: it[k] for k in dls.FWD_KEYS}, **{k: batch[k] for k in lang_keys}}, n_dev=1) for _ in range(NS)]
    for u in range(NS):
        slots[u].feed_from(stg[u], asm, via=via)
    return slots, stg
sts = [torch.cuda.Stream() for _ in range(NS)]
def loop(slots, stg, steps, wait_ready, wait_free, ncs, copy=True, ahead=0):
    cs = [torch.cuda.Stream() for _ in range(ncs)]
    ready = [None] * NS; free = [None] * NS
    def step(i):
        u = i % NS
        if copy:
            c_ = cs[u % ncs]
            if wait_free and free[u] is not None: c_.wait_event(free[u])
            with torch.cuda.stream(c_):
                stg[u].dbuf.copy_(stg[u].hbuf, non_blocking=True)
                ev = torch.cuda.Event(); ev.record(c_)
            if wait_ready: sts[u].wait_event(ev)
        slots[u].launch(sts[u])
        if copy and wait_free:
            ev2 = torch.cuda.Event(); ev2.record(sts[u]); free[u] = ev2
    for i in range(80): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): step(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6
slots, stg = build("device")
print("fed graph (device form), no copies at all: %.1f us/step" % loop(slots, stg, 2000, False, False, 1, copy=False))
for ncs in (1, 2, 4):
    for wr, wf in ((False, False), (True, False), (False, True), (True, True)):
        print(f"copy streams {ncs}  forward waits for its copy: {wr!s:5}  copy waits for the buffer's previous forward: {wf!s:5} -> {loop(slots, stg, 2000, wr, wf, ncs):6.1f} us/step")
