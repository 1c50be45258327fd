#!/bin/bash
# per-kernel durations of eager forwards of one workload (kernel trace only): kt_forward.sh <cfgN> [n] [pair]
R=$PWD; O=$R/gpurun_out/ktf; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o r -- python $R/scratch/prof_forward.py ${1:-cfg2} ${2:-10} ${3:-0} > $O/log 2>&1
cd $R; f=$(find $O -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import collections, csv, re, sys
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"^void ", "", r["Kernel_Name"]).replace("vog::", ""); n = re.sub(r"\(.*\)$", "", n)
    if n.startswith(("at::", "__amd")): continue
    acc.setdefault((n[:70], r["Grid_Size_X"]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
tot = 0
for k, v in acc.items():
    v = v[len(v) // 2:]
    m = sum(v) / len(v); tot += m
    print(f"{m:9.1f} us  {k[0]} [{k[1]}]")
print(f"{tot:9.1f} us sum")
PY
