"""Summarise a rocprofv3 rocpd database: python scratch/prof_sum.py gpurun_out/prof_x/x_results.db [top]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
top = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"total {tot/1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches")
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
for n, c, t, mn, mx in rows[:top]:
    print(f'"{n}",{c},{t},{t/c:.1f},{100*t/tot:.2f},{mn},{mx}')
