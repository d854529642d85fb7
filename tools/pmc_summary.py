#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc run (rocpd sqlite): per-kernel sum / mean of each counter."""
import sqlite3, sys, subprocess, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(rocpd_pmc_event)")]
icol = [r[1] for r in cur.execute("pragma table_info(rocpd_info_pmc)")]
print("# pmc_event cols:", cols)
print("# info_pmc cols:", icol)
q = """select s.kernel_name, p.name, count(*), sum(e.value) from rocpd_pmc_event e
       join rocpd_info_pmc p on e.pmc_id = p.id
       join rocpd_kernel_dispatch d on e.event_id = d.event_id
       join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name, p.name order by 4 desc"""
try:
    rows = cur.execute(q).fetchall()
except Exception as ex:
    print("query failed:", ex)
    for t in ("rocpd_pmc_event", "rocpd_info_pmc", "rocpd_kernel_dispatch"):
        print(t, cur.execute(f"select * from {t} limit 2").fetchall())
    sys.exit(0)
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
tot = {}
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n.replace("fq3::", "").replace("unsigned short", "bf16").replace("void ", ""))
    print(f"{n[:70]:70s} {r[1]:12s} dispatches {r[2]:7d} sum {r[3]:16.1f} mean {r[3]/r[2]:12.2f}")
    tot[r[1]] = tot.get(r[1], 0) + r[3]
print("totals:", tot)
