"""rocprofv3 results.db -> per-kernel stats table (markdown)."""
import sqlite3, sys
db = sys.argv[1]
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"total kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches\n")
print("| kernel | calls | total us | avg us | min us | max us | % |")
print("|---|---:|---:|---:|---:|---:|---:|")
for r in rows:
    print(f"| `{r[0][:100]}` | {r[1]} | {r[2]/1e3:.1f} | {r[3]/1e3:.2f} | {r[4]/1e3:.2f} | {r[5]/1e3:.2f} | {100*r[2]/tot:.1f} |")
