#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc run (rocpd sqlite): per-kernel sum / mean of each counter.
usage: pmc_summary.py <db> [--from <kernel-name substring>]
--from: count only the dispatches from the first one whose kernel name contains the substring on (a decode frame's first kernel:
frame_begin_kernel / frame_begin_batch_kernel) -- whatever ran before it (model set-up, the lanes' prefills, which share the
weight-stationary GEMM kernels with the frame) is not frame traffic -- and print the per-frame FETCH_SIZE figure (KB x 1024 x 2: the
gfx950 correction of MI355X_MICROARCH.md, 64 B counted per 128-B fabric request)."""
import sqlite3, sys, subprocess, re
args = sys.argv[1:]
first = None
if "--from" in args:
    i = args.index("--from"); first = args[i + 1]; del args[i:i + 2]
db = sqlite3.connect(args[0]); cur = db.cursor()
dcols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
order = next((c for c in ("dispatch_id", "start", "id") if c in dcols), "event_id")
print("# dispatch order key:", order)
q = f"""select s.kernel_name, p.name, d.{order}, e.value from rocpd_pmc_event e
        join rocpd_info_pmc p on e.pmc_id = p.id
        join rocpd_kernel_dispatch d on e.event_id = d.event_id
        join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.{order}"""
try:
    rows = cur.execute(q).fetchall()
except Exception as ex:
    print("query failed:", ex)
    for t in ("rocpd_pmc_event", "rocpd_info_pmc", "rocpd_kernel_dispatch"):
        print(t, cur.execute(f"select * from {t} limit 2").fetchall())
    sys.exit(0)
dropped = 0
if first is not None:
    start = next((i for i, r in enumerate(rows) if first in r[0]), None)
    if start is None:
        print(f"# no dispatch of a kernel named *{first}*: nothing to summarise"); sys.exit(0)
    dropped, rows = start, rows[start:]
    print(f"# dispatches counted from the first *{first}* on ({dropped} counter rows in front of it dropped)")
acc = {}
for name, ctr, _k, v in rows:
    n, s = acc.get((name, ctr), (0, 0.0))
    acc[(name, ctr)] = (n + 1, s + float(v))
items = sorted(acc.items(), key=lambda kv: -kv[1][1])
names = subprocess.run(["c++filt"], input="\n".join(k[0] for k, _v in items), capture_output=True, text=True).stdout.split("\n")
tot = {}
for ((raw, ctr), (n, s)), nm in zip(items, names):
    nm = re.sub(r"\(.*", "", nm.replace("fq3::", "").replace("unsigned short", "bf16").replace("void ", ""))
    print(f"{nm[:70]:70s} {ctr:12s} dispatches {n:7d} sum {s:16.1f} mean {s / n:12.2f}")
    tot[ctr] = tot.get(ctr, 0) + s
print("totals:", tot)
if first is not None and "FETCH_SIZE" in tot:
    frames = sum(n for (raw, ctr), (n, _s) in acc.items() if ctr == "FETCH_SIZE" and first in raw)
    if frames:
        print(f"frames: {frames}; FETCH_SIZE per frame: {tot['FETCH_SIZE'] / frames:.1f} KB -> x 1024 x 2 (gfx950) = {2048.0 * tot['FETCH_SIZE'] / frames / 1e9:.3f} GB per frame")
