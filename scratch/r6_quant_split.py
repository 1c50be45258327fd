"""Round 6: which scopes need hi + lo f16 operands? Operand-rounding model on the oracle (CPU).
   python scratch/r6_quant_split.py [qk-scale ...]   (cfg-2 shape, wq / wk x scale, pe x 4, perturbed LayerNorm)"""
import sys, copy, torch
sys.path.insert(0, ".")
from oracle import cases
from oracle import vog_oracle as vo
from scratch.r5_quant_envelope import scheme

h = torch.float16
SP = "split"
S = [("f16 everywhere", {"tx": h}),
     ("split enc, tx.proj, tx.qk", {"tx": h, "enc": SP, "enc.lang": h, "tx.proj": SP, "tx.qk": SP}),
     ("split tx.proj, tx.qk (enc f16)", {"tx": h, "tx.proj": SP, "tx.qk": SP}),
     ("split enc,proj,qk + wo,ffn", {"tx": h, "enc": SP, "enc.lang": h, "tx.proj": SP, "tx.qk": SP, "tx.wo": SP, "tx.ffn": SP}),
     ("split enc,proj,qk + p,v", {"tx": h, "enc": SP, "enc.lang": h, "tx.proj": SP, "tx.qk": SP, "tx.p": SP, "tx.v": SP}),
     ("split enc,proj,qk + enc.lang", {"tx": h, "enc": SP, "tx.proj": SP, "tx.qk": SP}),
     ("split enc,proj,qk + lstm", {"tx": h, "enc": SP, "tx.proj": SP, "tx.qk": SP, "lstm": SP}),
     ]

def run(base, qk, schemes, layers=None):
    c = copy.deepcopy(cases.CASES[base])
    c["sharp"] = (float(qk), 4.0)
    cases.CASES["_tmp"] = c
    cfg, sd, batch, cc = cases.build("_tmp")
    oc = vo.OracleCfg.from_cfg(cfg, cc["vocab"], cc["nppf0"])
    sdt, inp = vo.to_torch(sd), vo.to_torch(batch)
    torch.set_num_threads(16)
    with torch.no_grad():
        o = vo.forward(oc, sdt, inp)
        ev = o["mdl_outs_eval"]
        print(f"{base} x{qk}: logits std {o['mdl_outs'].std():.3f}", flush=True)
        for label, m in schemes:
            o2 = vo.forward(oc, sdt, inp, quant=scheme(m))
            nz = ev != 0
            rel = ((o2["mdl_outs_eval"] - ev).abs() / ev.abs().clamp(min=1e-6))[nz].max().item()
            print(f"   {label:36s} eval rel {rel:.2e}", flush=True)

if __name__ == "__main__":
    base = "full/cfg2_sharp16"
    args = sys.argv[1:] or ["16", "24", "32"]
    if args[0].startswith("full/"):
        base, args = args[0], args[1:]
    for s in args:
        run(base, float(s), S)
