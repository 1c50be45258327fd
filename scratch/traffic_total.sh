#!/bin/bash
# HBM-side bytes per forward of the DEFAULT paired forward: FETCH_SIZE / WRITE_SIZE passes
# usage: gpurun -- bash scratch/traffic_total.sh
R=$PWD; O=$R/gpurun_out/traffic; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
export VOG_PERF_EXPERIMENTS=1
for q in 0; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/q${q}_$c -o r -- python $R/scratch/prof_forward.py cfg2 20 1 > $O/q${q}_$c.log 2>&1
  done
done
cd $R
python - $O <<'PY'
import csv, glob, sys, collections, re
for q in (0,):
    tot = 0; per = collections.defaultdict(float); cnt = collections.Counter()
    for c, mul in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):     # FETCH_SIZE x2: gfx950 correction (guide)
        f = glob.glob(f"{sys.argv[1]}/q{q}_{c}/**/*counter_collection.csv", recursive=True)
        if not f: print("missing", q, c); continue
        for r in csv.DictReader(open(f[0])):
            n = re.sub(r"^void ", "", r["Kernel_Name"]).replace("vog::", ""); n = re.sub(r"\(.*\)$", "", n)
            if n.startswith(("at::", "__amd")): continue
            b = float(r["Counter_Value"]) * 1024 * mul / 20         # KB units -> bytes, per forward
            per[n[:60]] += b; cnt[n[:60]] += 1
    for k in [k for k in per if cnt[k] < 20]: del per[k]            # once-per-checkpoint kernels (the gate-table GEMM)
    tot = sum(per.values())
    print(f"default forward: {tot/1e6:.1f} MB per forward (paired launches, stream-form encoders)")
    for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:8]:
        print(f"     {v/1e6:7.1f} MB  {k}")
PY
echo "throughput: $(python bench.py --throughput-only | tail -1)"
