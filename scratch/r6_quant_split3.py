"""Round 6: hi + lo plan on 3-layer stacks (every tail but the last mul_tx one feeds another attention layer).
   The oracle's hooks do not tell layers apart, so 'all tails split' is an upper bound of what the plan does (the last
   mul_tx tail stays f16 in the kernels); 'no tail split' the lower one."""
import sys, copy, torch
sys.path.insert(0, ".")
from oracle import cases
from oracle import vog_oracle as vo
h = torch.float16
def sp(x):
    hi = x.to(h).float()
    return hi + (x - hi).to(h).float()
def scheme(split):
    def q(scope, x):
        key = scope if scope in split else scope.split(".")[0] if scope.split(".")[0] in split else None
        return sp(x) if key else x.to(h).float()
    return q
S = [("f16", set()),
     ("split enc,proj,qk", {"enc", "tx.proj", "tx.qk"}),
     ("split enc,proj,qk,wo,ffn", {"enc", "tx.proj", "tx.qk", "tx.wo", "tx.ffn"}),
     ("split enc + all tx", {"enc", "tx"})]
def run(base, qk):
    c = copy.deepcopy(cases.CASES[base]); c["sharp"] = (float(qk), 4.0); cases.CASES["_tmp"] = c
    cfg, sd, batch, cc = cases.build("_tmp")
    oc = vo.OracleCfg.from_cfg(cfg, cc["vocab"], cc["nppf0"])
    sdt, inp = vo.to_torch(sd), vo.to_torch(batch)
    torch.set_num_threads(16)
    with torch.no_grad():
        o = vo.forward(oc, sdt, inp); ev = o["mdl_outs_eval"]
        print(f"{base} x{qk}", flush=True)
        for label, m in S:
            o2 = vo.forward(oc, sdt, inp, quant=scheme(m))
            nz = ev != 0
            r = ((o2["mdl_outs_eval"] - ev).abs() / ev.abs().clamp(min=1e-6))[nz]
            print(f"   {label:28s} eval rel max {r.max().item():.2e}", flush=True)
for s_ in sys.argv[1:] or ["8", "12", "16", "24"]:
    run("full/vog_spat_3layers_sharp8", float(s_))
