"""PMC csv (FETCH_SIZE pass, WRITE_SIZE pass) -> profiles/pmc_traffic.json for bench.py.

usage: pmc_to_json.py <fetch_csv> <write_csv> <workload> <forwards> <out_json>
`forwards` = number of whole forwards the traced command ran (steps + warmup); kernels launched
more often than that by the per-kernel timing loops are averaged per launch, which is what
bench.py wants. FETCH_SIZE is doubled (gfx950: the counter tallies 128-B requests at 64 B,
MI355X_MICROARCH.md "HBM"); WRITE_SIZE is used as reported (uncalibrated). Both are in KiB.
"""
import collections, csv, json, re, sys

fetch_csv, write_csv, workload, forwards, out = sys.argv[1:6]
forwards = int(forwards)


def load(path):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        n = re.sub(r"^void ", "", r["Kernel_Name"]).replace("vog::", "")
        n = re.sub(r"\(.*\)$", "", n)
        acc[(n, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return acc


f, w = load(fetch_csv), load(write_csv)
# step name <- (kernel, grid) at the cfg2 shapes (bench.py --workload cfg2)
STEP = {
    ("lstm_step_kernel<F16>", 131072): ("lstm_step", 24),
    ("lstm_layer_kernel<F16, 32>", 16384): ("lstm_layer", 2),
    ("gemm_pipe<BF16, 64, 64, 2, 0>", 193536): ("mul_wo", 1),      # also mul_ffn2 (same grid): averaged
    ("gemm_skinny<F16, false, 8, 1>", 131072): ("lstm_ih1", 1),
    ("gemm_skinny<F16, true, 8, 1>", 131072): ("lstm_ih0", 1),
    ("attn_sb_kernel<BF16, 8>", 122880): ("mul_attn", 1),
    ("gemm_pipe<BF16, 64, 64, 2, 1>", 119808): ("mul_pv", 1),
    ("gemm_pipe<BF16, 64, 64, 2, 1>", 89856): ("obj_qkv", 1),
    ("gemm_pipe<BF16, 64, 64, 2, 0>", 96768): ("mul_ffn1", 1),
    ("attn_frag_kernel<BF16, 6>", 21504): ("obj_attn", 1),
}
kern = {}
total = 0.0
for key, vals in f.items():
    if not key[0].startswith(("lstm", "gemm", "attn", "layernorm", "splitk", "argvec", "score", "pred", "vis_prep",
                              "lang_prep", "vislang", "qkv_combine", "cast2", "box_u", "srl_gather", "predcmp")):
        continue
    fb = 2.0 * 1024 * sum(vals) / len(vals)
    wv = w.get(key, [0.0])
    wb = 1024 * sum(wv) / len(wv)
    # launches per forward: kernels in the timing loops have extra launches; take the integer
    # nearest to launches / forwards but at least 1 for graph-resident kernels
    # launches per forward: bench.py's per-kernel timing loops add 103 launches per timed step
    # that uses this (kernel, grid); the rest are `forwards` whole forwards
    lpf = None
    for k in range(0, 8):
        rest = len(vals) - 103 * k
        if rest > 0 and rest % forwards == 0:
            lpf = rest // forwards
            break
    if lpf is None:
        lpf = max(1, round(len(vals) / forwards))
    total += (fb + wb) * lpf
    if key in STEP:
        kern[STEP[key][0]] = {"bytes_per_launch": fb + wb, "fetch_bytes": fb, "write_bytes": wb,
                              "kernel": key[0], "grid_threads": key[1]}
doc = {}
try:
    doc = json.load(open(out))
except Exception:
    pass
doc[workload] = {"source": "profiles/round1_pmc_traffic.md (rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE; FETCH x2 gfx950 correction)",
                 "bytes_per_forward": total, "kernels": kern}
json.dump(doc, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps(doc[workload], indent=1)[:1500])
