"""Generate the committed golden fixtures from the REFERENCE itself.

TEST INFRASTRUCTURE; runs only in the build container (needs /root/reference,
imported through `oracle/ref_import.py`). For every case of `oracle/cases.py`:
regenerate inputs + weights from their seeds, run the reference model class
(CPU, fp32, eval, no_grad) and the reference `Evaluator*.get_out_results_boxes`,
and store OUTPUTS ONLY (+ stage outputs captured with forward hooks, + SHA-256 of
the generated inputs / weights) in `tests/golden/<case>.npz`.

    python -m oracle.make_golden            # all cases
    python -m oracle.make_golden small/     # name prefix
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

from oracle import cases, ref_import


def _hooks(mdl, store):
    hs = []

    def grab(name):
        def fn(_m, _i, o):
            store[name] = o.detach().clone()
        return fn
    for attr, name in (("prop_encoder", "st_prop_enc"), ("seg_encoder", "st_seg_enc"),
                       ("srl_arg_words_out_enc", "st_argvec_premask"),
                       ("obj_txf", "st_obj_out"), ("mult_txf", "st_mul_out")):
        if hasattr(mdl, attr):
            hs.append(getattr(mdl, attr).register_forward_hook(grab(name)))
    return hs


def make(name: str):
    cfg, sd, batch, c = cases.build(name)
    torch.set_num_threads(8)
    mdl = ref_import.build_model(cfg, c["vocab"], c["nppf0"], sd)
    store = {}
    if name.startswith("small/"):        # stage tensors only where they are KB-sized
        _hooks(mdl, store)
    inp = {k: torch.from_numpy(v).clone() for k, v in batch.items()}
    t0 = time.time()
    with torch.no_grad():
        out = mdl(inp)
        evl = ref_import.build_evaluator(cfg, c["nppf0"])
        inp2 = {k: torch.from_numpy(v).clone() for k, v in batch.items()}
        pr = evl.get_out_results_boxes(out, inp2)
    dt = time.time() - t0
    rec = {k: v.detach().contiguous().numpy() for k, v in out.items()}
    rec.update({k: pr[k].contiguous().numpy() for k in ("boxes", "scores", "indexs")})
    rec.update({k: v.numpy() for k, v in store.items()})
    rec["sha_inputs"] = np.array(cases.digest(batch))
    rec["sha_weights"] = np.array(cases.digest(sd))
    rec["ref_seconds"] = np.array(dt, np.float32)
    path = cases.golden_path(name)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, **rec)
    print(f"{name:40s} {dt:7.2f}s  {os.path.getsize(path)/1024:8.1f} KB")


def main(argv):
    if not ref_import.available():
        raise SystemExit("reference tree not present; goldens are generated in the build container")
    pref = argv[1] if len(argv) > 1 else ""
    for name in cases.CASES:
        if name.startswith(pref):
            make(name)


if __name__ == "__main__":
    main(sys.argv)
