"""Round 5: operand-rounding model of the 16-bit path on the sharpened cases (CPU, oracle only).
   python scratch/r5_quant_envelope.py [case ...]"""
import sys, torch
sys.path.insert(0, ".")
from oracle import cases
from oracle import vog_oracle as vo

def scheme(m, rest=torch.float16):
    def q(scope, x):
        t = m.get(scope, m.get(scope.split(".")[0], rest))
        if t is None:
            return x
        if t == "split":       # hi + lo f16: ~2^-22
            hi = x.to(torch.float16).float()
            return hi + (x - hi).to(torch.float16).float()
        return x.to(t).to(torch.float32)
    return q

def run(name, schemes):
    cfg, sd, batch, c = cases.build(name)
    oc = vo.OracleCfg.from_cfg(cfg, c["vocab"], c["nppf0"])
    sdt, inp = vo.to_torch(sd), vo.to_torch(batch)
    torch.set_num_threads(8)
    with torch.no_grad():
        o = vo.forward(oc, sdt, inp, keep_stages=False)
        ref = vo.pred_head(oc, o, inp)["scores"]
        ev = o["mdl_outs_eval"]
        print(f"{name}: logits std {o['mdl_outs'].std():.3f} scores min {ref[ref>0].min():.2e} max {ref.max():.3f}")
        for label, m in schemes:
            o2 = vo.forward(oc, sdt, inp, quant=scheme(m))
            got = vo.pred_head(oc, o2, inp)["scores"]
            nz = ref > 0
            rel = ((got - ref).abs() / ref.clamp(min=1e-6))[nz].max().item()
            nz2 = ev != 0
            rel2 = ((o2["mdl_outs_eval"] - ev).abs() / ev.abs().clamp(min=1e-6))[nz2].max().item()
            dl = (o2["mdl_outs"] - o["mdl_outs"]).abs().max().item()
            print(f"   {label:34s} scores rel {rel:.2e}  eval rel {rel2:.2e}  logit abs {dl:.2e}")

bf, h = torch.bfloat16, torch.float16
S = [("bf16 tx", {"tx": bf}), ("f16 tx", {"tx": h}),
     ("f16 tx, qk split", {"tx": h, "tx.qk": "split"}),
     ("f16 tx, qk+proj exact", {"tx": h, "tx.qk": None, "tx.proj": None}),
     ("f16 tx, p exact", {"tx": h, "tx.p": None}),
     ("f16 tx, qk,proj,p,v exact", {"tx": h, "tx.qk": None, "tx.proj": None, "tx.p": None, "tx.v": None}),
     ("tx exact (enc/lstm/head f16)", {"tx": None}),
     ("only tx f16 (rest exact)", {"tx": h, "enc": None, "lstm": None, "head": None}),
     ]
if __name__ == "__main__":
  for n in (sys.argv[1:] or ["full/cfg2_vog_spat_gt5_bs4", "full/cfg2_sharp8", "full/cfg2_sharp16", "full/cfg2_relu_heavy"]):
      run(n, S)
