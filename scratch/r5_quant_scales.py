"""Envelope: largest wq/wk scale that holds 1e-3 per precision plan (operand-rounding model on the oracle)."""
import sys, torch, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "scratch")
from oracle import cases
from oracle import vog_oracle as vo
from r5_quant_envelope import scheme, bf, h
E=None
PLANS = [("bf16 tx", {"tx": bf}), ("f16", {"tx": h}), ("f16 + enc split", {"tx": h, "enc": "split"}),
         ("f16 + enc,qk,proj split", {"tx": h, "enc": "split", "tx.qk": "split", "tx.proj": "split"}),
         ("f16 + enc,qk,proj,head split", {"tx": h, "enc": "split", "tx.qk": "split", "tx.proj": "split", "head": "split"})]
torch.set_num_threads(8)
for scale in (1, 4, 8, 12, 16, 24, 32):
    cases.CASES["tmp"] = cases._case(cases._SPAT2, B=4, ragged=True, dseed=52, perturb_ln=True, sharp=(float(scale), 4.0))
    cfg, sd, batch, c = cases.build("tmp")
    oc = vo.OracleCfg.from_cfg(cfg, c["vocab"], c["nppf0"])
    sdt, inp = vo.to_torch(sd), vo.to_torch(batch)
    # sharpness statistic
    sig = 0
    for k in sd:
        if k.endswith("wq.weight"):
            wq, wk = sd[k], sd[k.replace("wq", "wk")]
            d = wq.shape[0]; nh = 3
            for ch in np.array_split(np.arange(d), nh):
                sig = max(sig, np.linalg.norm(wq[ch].T @ wk[ch]) / np.sqrt(d))
    with torch.no_grad():
        o = vo.forward(oc, sdt, inp); ev = o["mdl_outs_eval"]; nz = ev != 0
        row = []
        for label, m in PLANS:
            o2 = vo.forward(oc, sdt, inp, quant=scheme(m))
            row.append(((o2["mdl_outs_eval"] - ev).abs() / ev.abs().clamp(min=1e-6))[nz].max().item())
    print(f"x{scale:<3d} sigma {sig:7.2f} | " + " | ".join(f"{l} {r:.2e}" for (l, _), r in zip(PLANS, row)), flush=True)
