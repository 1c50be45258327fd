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
    def histogram(evs):
        pts = sorted([(s, 1) for s, _, _ in evs] + [(e, -1) for _, e, _ in evs])
        hist, cur, last = {}, 0, pts[0][0]
        for t, d in pts:
            hist[cur] = hist.get(cur, 0) + (t - last)
            cur += d
            last = t
        return hist, pts[-1][0] - pts[0][0]
    hist, win = histogram(ev)
    print(f"\nqueues {queues}; window {win/1e3:.1f} us; sum(durations)/window = {tot/win:.3f}")
    print("kernels executing at once (whole trace, incl. start-up, warm-up and the per-kernel timing passes): "
          + ", ".join(f"{k}: {100*v/win:.1f}%" for k, v in sorted(hist.items())))
    # the timed region: the densest contiguous stretch of dispatches (no gap longer than 1 ms), i.e.
    # the K steps issued round-robin over the streams
    best, lo = (0, 0, 0), 0
    for i in range(1, len(ev) + 1):
        if i == len(ev) or ev[i][0] - max(e[1] for e in ev[max(lo, i - 8):i]) > 1_000_000:
            if i - lo > best[0]:
                best = (i - lo, lo, i)
            lo = i
    _, a, b = best
    core = ev[a:b]
    k = len(core) // 10                      # drop the ramp at both ends
    core = core[k:len(core) - k] if len(core) > 50 else core
    hist, win = histogram(core)
    dur = sum(e - s for s, e, _ in core)
    print(f"steady state ({len(core)} dispatches, {win/1e3:.1f} us): sum(durations)/window = {dur/win:.3f}; kernels executing at once: "
          + ", ".join(f"{k}: {100*v/win:.1f}%" for k, v in sorted(hist.items())))
