#!/bin/bash
# per-kernel durations under load (4 graph streams) vs alone (1 stream): kernel trace only, no counters
# usage: gpurun -- bash scratch/prof_load.sh [tag]   (env passes through to bench.py)
R=$PWD; O=$R/gpurun_out/load${1:-}; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for s in 4 1; do
  timeout 240 rocprofv3 --kernel-trace --output-format csv -d $O/s$s -o r -- python $R/bench.py --throughput-only --streams $s --steps 400 --warmup 40 > $O/s$s.log 2>&1
done
cd $R
python - $O <<'PY'
import csv, collections, re, sys, glob
res = {}
for s in (1, 4):
    f = glob.glob(f"{sys.argv[1]}/s{s}/**/*kernel_trace.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = re.sub(r"^void ", "", r["Kernel_Name"]).replace("vog::", ""); n = re.sub(r"\(.*\)$", "", n)
        if n.startswith(("at::", "__amd")): continue
        acc[(n[:64], r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
    res[s] = {k: (sum(v[len(v)//4:]) / len(v[len(v)//4:]), len(v)) for k, v in acc.items()}
t1 = t4 = 0
for k in sorted(res[1], key=lambda k: -res[4].get(k, (0, 0))[0]):
    a, n = res[1][k]; b = res[4].get(k, (0, 0))[0]
    if n < 100: continue
    t1 += a; t4 += b
    print(f"{a:7.1f} -> {b:7.1f} us  {k[0]} [{k[1]}]")
print(f"{t1:7.1f} -> {t4:7.1f} sum")
PY
tail -1 $O/s4.log; tail -1 $O/s1.log
