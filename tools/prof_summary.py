#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database: per-kernel count / total / avg duration (us)."""
import sqlite3, sys, subprocess, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
q = """select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start), avg(d.grid_size_x), avg(d.workgroup_size_x)
       from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name order by 3 desc"""
rows = cur.execute(q).fetchall()
tot = sum(r[2] for r in rows)
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
print(f"{'kernel':90s} {'calls':>7s} {'total_ms':>9s} {'avg_us':>8s} {'min_us':>8s} {'max_us':>8s} {'%':>5s} {'grid':>7s}")
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("fq3::", "").replace("unsigned short", "bf16").replace("void ", ""))
    print(f"{n[:90]:90s} {r[1]:7d} {r[2]/1e6:9.3f} {r[2]/r[1]/1e3:8.2f} {r[3]/1e3:8.2f} {r[4]/1e3:8.2f} {100*r[2]/tot:5.1f} {int(r[5]/max(r[6],1)):7d}")
print(f"total kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
span = cur.execute("select min(start), max(end) from rocpd_kernel_dispatch").fetchone()
print(f"first->last dispatch span {(span[1]-span[0])/1e6:.3f} ms")
