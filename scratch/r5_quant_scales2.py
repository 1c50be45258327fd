import sys, torch, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "scratch")
from oracle import cases
from oracle import vog_oracle as vo
from r5_quant_envelope import scheme, bf, h
PLANS = [("enc,proj split", {"tx": h, "enc": "split", "tx.proj": "split"}),
         ("enc,qk split", {"tx": h, "enc": "split", "tx.qk": "split"}),
         ("enc,proj,qk split", {"tx": h, "enc": "split", "tx.proj": "split", "tx.qk": "split"}),
         ("enc,proj,qk,p,v split", {"tx": h, "enc": "split", "tx.proj": "split", "tx.qk": "split", "tx.p": "split", "tx.v": "split"}),
         ("enc,tx split", {"tx": "split", "enc": "split"}),
         ("enc,tx,head split", {"tx": "split", "enc": "split", "head": "split"}),
         ("all split", {"tx": "split", "enc": "split", "head": "split", "lstm": "split"})]
torch.set_num_threads(8)
for scale in (16, 24, 32):
  for dseed in (52, 61):
    cases.CASES["tmp"] = cases._case(cases._SPAT2, B=4, ragged=True, dseed=dseed, perturb_ln=True, sharp=(float(scale), 4.0))
    cfg, sd, batch, c = cases.build("tmp")
    oc = vo.OracleCfg.from_cfg(cfg, c["vocab"], c["nppf0"])
    sdt, inp = vo.to_torch(sd), vo.to_torch(batch)
    with torch.no_grad():
        o = vo.forward(oc, sdt, inp); ev = o["mdl_outs_eval"]; nz = ev != 0
        row = []
        for label, m in PLANS:
            o2 = vo.forward(oc, sdt, inp, quant=scheme(m))
            row.append(((o2["mdl_outs_eval"] - ev).abs() / ev.abs().clamp(min=1e-6))[nz].max().item())
    print(f"x{scale:<3d} d{dseed} | " + " | ".join(f"{l} {r:.2e}" for (l, _), r in zip(PLANS, row)), flush=True)
