#!/usr/bin/env python3
"""List the individual dispatches of kernels matching a substring from a rocprofv3 rocpd database, in launch order:
usage: prof_dispatches.py <db> <substring> [first] [count]   -> index, duration us, grid (workgroups x, y, z)"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
pat = sys.argv[2]; first = int(sys.argv[3]) if len(sys.argv) > 3 else 0; count = int(sys.argv[4]) if len(sys.argv) > 4 else 100
q = """select d.start, d.end, d.grid_size_x / d.workgroup_size_x, d.grid_size_y / d.workgroup_size_y, d.grid_size_z / d.workgroup_size_z, s.kernel_name
       from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"""
rows = [r for r in cur.execute(q).fetchall() if pat in r[5]]
for i, r in enumerate(rows[first:first + count]):
    print(f"{first + i:4d} {(r[1] - r[0]) / 1e3:9.2f} us  grid {int(r[2])} x {int(r[3])} x {int(r[4])}")
