#!/usr/bin/env python3
"""The last N kernel dispatches of a rocprofv3 rocpd database in launch order: index, duration us, grid, kernel name.
usage: prof_timeline.py <db> [N=230]"""
import re, sqlite3, subprocess, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
N = int(sys.argv[2]) if len(sys.argv) > 2 else 230
q = """select d.start, d.end, d.grid_size_x / d.workgroup_size_x, d.grid_size_y / d.workgroup_size_y, d.grid_size_z / d.workgroup_size_z, s.kernel_name
       from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"""
rows = cur.execute(q).fetchall()[-N:]
names = subprocess.run(["c++filt"], input="\n".join(r[5] for r in rows), capture_output=True, text=True).stdout.split("\n")
t0 = rows[0][0]
for i, (r, n) in enumerate(zip(rows, names)):
    n = re.sub(r"\(.*", "", n.replace("fq3::", "").replace("unsigned short", "bf16").replace("void ", ""))
    print(f"{i:4d} t={(r[0] - t0) / 1e3:10.1f} us  {(r[1] - r[0]) / 1e3:9.2f} us  grid {int(r[2])} x {int(r[3])} x {int(r[4])}  {n[:80]}")
print(f"span {(rows[-1][1] - t0) / 1e6:.3f} ms, kernel time {sum(r[1] - r[0] for r in rows) / 1e6:.3f} ms")
