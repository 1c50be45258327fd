"""Round 6: hi + lo per stack - which tail needs it? (obj_tx: d = 512 / 256, mul_tx: 768 / 384)"""
import sys, copy, torch
sys.path.insert(0, ".")
from oracle import cases
from oracle import vog_oracle as vo

h = torch.float16
def sp(x):
    hi = x.to(h).float()
    return hi + (x - hi).to(h).float()

def scheme(split_scopes):
    """split_scopes: set of 'enc', 'proj', 'qk', 'obj.wo', 'obj.ffn', 'mul.wo', 'mul.ffn', 'p', 'v'"""
    def q(scope, x):
        if scope == "enc":
            return sp(x) if "enc" in split_scopes else x.to(h).float()
        if scope.startswith("tx."):
            sub = scope[3:]
            if sub in ("wo", "ffn"):
                stack = "obj" if (x.shape[-1] in (512, 256) and x.shape[0] != 768 and x.shape[0] != 384) else "mul"
                if x.dim() == 2 and x.shape[0] in (512, 256) and x.shape[1] in (512, 256):
                    stack = "obj"
                elif x.dim() == 2:
                    stack = "mul"
                return sp(x) if f"{stack}.{sub}" in split_scopes else x.to(h).float()
            return sp(x) if sub in split_scopes else x.to(h).float()
        return x.to(h).float()
    return q

base = {"enc", "proj", "qk"}
S = [("split enc,proj,qk", base),
     ("+ obj.wo", base | {"obj.wo"}),
     ("+ obj.ffn", base | {"obj.ffn"}),
     ("+ obj.wo, obj.ffn", base | {"obj.wo", "obj.ffn"}),
     ("+ mul.wo, mul.ffn", base | {"mul.wo", "mul.ffn"}),
     ("+ all four", base | {"obj.wo", "obj.ffn", "mul.wo", "mul.ffn"})]

def run(basecase, qk):
    c = copy.deepcopy(cases.CASES[basecase])
    c["sharp"] = (float(qk), 4.0)
    cases.CASES["_tmp"] = c
    cfg, sd, batch, cc = cases.build("_tmp")
    oc = vo.OracleCfg.from_cfg(cfg, cc["vocab"], cc["nppf0"])
    sdt, inp = vo.to_torch(sd), vo.to_torch(batch)
    torch.set_num_threads(16)
    with torch.no_grad():
        o = vo.forward(oc, sdt, inp)
        ev = o["mdl_outs_eval"]
        print(f"{basecase} x{qk}", flush=True)
        for label, m in S:
            o2 = vo.forward(oc, sdt, inp, quant=scheme(m))
            nz = ev != 0
            r = ((o2["mdl_outs_eval"] - ev).abs() / ev.abs().clamp(min=1e-6))[nz]
            print(f"   {label:28s} eval rel max {r.max().item():.2e}  p99.9 {r.quantile(0.999).item():.2e}", flush=True)

if __name__ == "__main__":
    for s in sys.argv[1:] or ["16", "24", "32"]:
        run("full/cfg2_sharp16", float(s))
