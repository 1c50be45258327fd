"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE csv -> per-(kernel, grid) average KiB per launch."""
import csv, re, sys, collections
def load(path):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        n = re.sub(r"^void ", "", r["Kernel_Name"]).replace("vog::", "")
        n = re.sub(r"\(.*\)$", "", n)[:70]
        acc[(n, r["Grid_Size"])].append(float(r["Counter_Value"]))
    return acc
f = load(sys.argv[1]); w = load(sys.argv[2])
print("| kernel | grid | launches | FETCH_SIZE KB (raw) | x2 gfx950 corr. MB | WRITE_SIZE KB (raw) |")
print("|---|---:|---:|---:|---:|---:|")
for k in sorted(f, key=lambda k: -sum(f[k])):
    fa = sum(f[k]) / len(f[k]); wa = sum(w.get(k, [0])) / max(1, len(w.get(k, [0])))
    print(f"| `{k[0]}` | {k[1]} | {len(f[k])} | {fa:.0f} | {2*fa/1024:.2f} | {wa:.0f} |")
