#!/bin/bash
# busy-CU time per kernel of one forward (1 stream): SQ_BUSY_CU_CYCLES / 2100 = CU*us   usage: prof_cu.sh [pair] [tag]
R=$PWD; O=$R/gpurun_out/cu; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --pmc SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/p${2:-} -o r -- python $R/scratch/prof_forward.py ${WL:-cfg2} 20 ${1:-1} > $O/p.log 2>&1
cd $R; f=$(find $O/p${2:-} -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import collections, csv, re, sys
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"^void ", "", r["Kernel_Name"]).replace("vog::", ""); n = re.sub(r"\(.*\)$", "", n)
    if n.startswith(("at::", "__amd")): continue
    acc[(n[:70], r["Grid_Size"])].append(float(r["Counter_Value"]))
tot = 0
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if len(v) < 10: continue            # once-per-checkpoint kernels (the gate-table GEMM of vog_ctx_finalize)
    per_fwd = sum(v) / 20 / 2100.0
    tot += per_fwd
    print(f"{per_fwd:8.0f} CU*us/forward  x{len(v)//20}  {k[0]} [{k[1]}]")
print(f"{tot:8.0f} total -> {tot/256:.1f} us of the whole chip per forward")
PY
