"""rocprofv3 --pmc <counters> --kernel-trace csv -> per-kernel averages of every counter collected.
usage: python scratch/pmc_kernels.py <counter_collection.csv> [kernel-name substring ...]"""
import collections, csv, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"^void ", "", r["Kernel_Name"]).replace("vog::", "")
    n = re.sub(r"\(.*\)$", "", n)[:60]
    if len(sys.argv) > 2 and not any(s in n for s in sys.argv[2:]):
        continue
    acc[(n, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for v in acc.values() for c in v})
print("| kernel | grid | n | " + " | ".join(names) + " |")
print("|---|---:|---:|" + "---:|" * len(names))
for k, v in sorted(acc.items()):
    n = max(len(x) for x in v.values())
    print(f"| `{k[0]}` | {k[1]} | {n} | " + " | ".join(f"{sum(v[c]) / max(1, len(v[c])):.0f}" if c in v else "-" for c in names) + " |")
