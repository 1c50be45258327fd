"""rocprofv3 --kernel-trace CSV -> launch-boundary gaps per queue: for every dispatch, the time between the END of the
previous dispatch on the SAME queue and its own START, grouped by (previous kernel -> this kernel). Steady state only
(the densest stretch of dispatches). Usage: gap_analysis.py <dir with *kernel_trace.csv> [label]"""
import collections
import csv
import glob
import re
import sys


def short(n):
    n = re.sub(r"^void ", "", n).replace("vog::", "")
    n = re.sub(r"\(.*\)$", "", n)
    n = re.sub(r"pair_kernel<(\w+)<[^>]*>\s*,\s*(\w+)<.*", r"pair<\1,\2>", n)
    return n[:44]


f = glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if n.startswith(("at::", "__amd", "void at::")):
        continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], short(n) + " [" + r["Grid_Size_X"] + "]"))
rows.sort()
# steady state: drop the first and last 15 % of the dispatches
k = len(rows) * 15 // 100
rows = rows[k:len(rows) - k]
byq = collections.defaultdict(list)
for s, e, q, n in rows:
    byq[q].append((s, e, n))
gaps = collections.defaultdict(list)
tot_gap = tot_dur = 0
nfw = 0
for q, ev in byq.items():
    for i in range(1, len(ev)):
        g = (ev[i][0] - ev[i - 1][1]) / 1000.0
        if g > 200:          # a pause of the issuing loop, not a boundary
            continue
        gaps[(ev[i - 1][2], ev[i][2])].append(g)
        tot_gap += g
        tot_dur += (ev[i][1] - ev[i][0]) / 1000.0
        nfw += ev[i][2].startswith("prep_fused")
print(f"# {sys.argv[2] if len(sys.argv) > 2 else ''}: {len(rows)} dispatches on {len(byq)} queues, {nfw} forwards; per forward: "
      f"{tot_dur / max(nfw, 1):.1f} us in kernels, {tot_gap / max(nfw, 1):.1f} us between them\n")
print("| previous kernel -> kernel | n | mean gap us | p50 | p90 |")
print("|---|---:|---:|---:|---:|")
for (a, b), v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    if len(v) < 20:
        continue
    v.sort()
    print(f"| `{a}` -> `{b}` | {len(v)} | {sum(v) / len(v):.2f} | {v[len(v) // 2]:.2f} | {v[len(v) * 9 // 10]:.2f} |")
