"""rocprofv3 results.db -> per-kernel stats table (markdown).

Rows are grouped by (kernel name, grid, workgroup): the same template instantiation is
launched for several GEMMs of one forward (Wo, FFN1, FFN2 ...), and the grid tells them apart.
Also prints the cross-queue concurrency histogram (fraction of the traced window with
0/1/2/.. kernels executing) when more than one queue was used.
"""
import re
import sqlite3
import sys

db = sys.argv[1]
c = sqlite3.connect(db)
rows = c.execute(
    "select name, grid_x, grid_y, grid_z, workgroup_x, count(*), sum(duration), avg(duration), min(duration), "
    "max(duration), max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels "
    "group by name, grid_x, grid_y, grid_z, workgroup_x order by sum(duration) desc").fetchall()
tot = sum(r[6] for r in rows)
print(f"total kernel time {tot/1e6:.3f} ms over {sum(r[5] for r in rows)} dispatches\n")
print("| kernel | grid (threads) | wg | calls | total us | avg us | min us | max us | % | vgpr+agpr | lds |")
print("|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|")


def short(n):
    n = re.sub(r"^void ", "", n)
    n = n.replace("vog::", "").replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*\)$", "", n)
    return n[:90]


for r in rows:
    print(f"| `{short(r[0])}` | {r[1]}x{r[2]}x{r[3]} | {r[4]} | {r[5]} | {r[6]/1e3:.1f} | {r[7]/1e3:.2f} | "
          f"{r[8]/1e3:.2f} | {r[9]/1e3:.2f} | {100*r[6]/tot:.1f} | {r[10]}+{r[11]} | {r[12]} |")

ev = c.execute("select start, end, queue_id from kernels order by start").fetchall()
queues = sorted({e[2] for e in ev})
if len(queues) > 1:
    pts = sorted([(s, 1) for s, _, _ in ev] + [(e, -1) for _, e, _ in ev])
    hist, cur, last = {}, 0, pts[0][0]
    for t, d in pts:
        hist[cur] = hist.get(cur, 0) + (t - last)
        cur += d
        last = t
    win = pts[-1][0] - pts[0][0]
    print(f"\nqueues {queues}; window {win/1e3:.1f} us; sum(durations)/window = {tot/win:.3f}")
    print("kernels executing at once: " + ", ".join(f"{k}: {100*v/win:.1f}%" for k, v in sorted(hist.items())))
