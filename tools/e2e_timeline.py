#!/usr/bin/env python3
"""Where a batched end-to-end run spends its wall time, from a rocprofv3 --kernel-trace rocpd database of tools/batch_e2e_bench.py:
the lock-step frames of the LAST run are delimited by sample_talker_batch_kernel (the last launch of a frame); for every frame the
tool splits the interval since the previous frame's end into time the decode queue was busy and time it sat idle, and lists what ran
on the other queues meanwhile.  usage: e2e_timeline.py <db> [frames_in_last_run=400]"""
import re, sqlite3, subprocess, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
NF = int(sys.argv[2]) if len(sys.argv) > 2 else 400
cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
qcol = "d.queue_id" if "queue_id" in cols else "0"
scol = "d.stream_id" if "stream_id" in cols else "0"
rows = cur.execute(f"""select d.start, d.end, s.kernel_name, {qcol}, {scol} from rocpd_kernel_dispatch d
                       join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start""").fetchall()
uniq = sorted(set(r[2] for r in rows))
dem = dict(zip(uniq, subprocess.run(["c++filt"], input="\n".join(uniq), capture_output=True, text=True).stdout.split("\n")))
short = lambda n: re.sub(r"\(.*", "", dem[n].replace("(anonymous namespace)::", "").replace("fq3::", "").replace("unsigned short", "bf16").replace("void ", ""))[:60]
ends = [i for i, r in enumerate(rows) if "sample_talker_batch_kernel" in r[2]]
if len(ends) < NF + 1:
    NF = len(ends) - 1
first, last = ends[-NF - 1], ends[-1]
dq = rows[last][3], rows[last][4]                                  # the decode queue / stream
t_begin, t_end = rows[first][1], rows[last][1]
print(f"# {len(rows)} dispatches; last run: {NF} lock-step frames in {(t_end - t_begin) / 1e6:.1f} ms; decode queue/stream {dq}")
# work before the first frame of the run (staging of the first wave) and after the last frame (tail vocoding): bounded by the
# neighbouring runs' frames, so only the tail is reported here
tail = [r for r in rows[last + 1:]]
if tail:
    print(f"# after the last frame: {len(tail)} dispatches, span {(tail[-1][1] - t_end) / 1e6:.1f} ms, kernel time {sum(r[1] - r[0] for r in tail) / 1e6:.1f} ms")
fr = []
j = first + 1
bounds = [rows[i][1] for i in ends[-NF - 1:]]
k = first + 1
for f in range(NF):
    a, b = bounds[f], bounds[f + 1]
    busy = other = 0
    names = collections.Counter()
    while k < len(rows) and rows[k][0] < b:
        r = rows[k]
        if (r[3], r[4]) == dq:
            busy += min(r[1], b) - max(r[0], a)
        else:
            other += r[1] - r[0]
            names[short(r[2])] += r[1] - r[0]
        k += 1
    fr.append((b - a, busy, other, names))
durs = sorted(x[0] for x in fr)
med = durs[len(durs) // 2]
print(f"# frame interval: median {med / 1e3:.1f} us, p10 {durs[len(durs) // 10] / 1e3:.1f}, p90 {durs[9 * len(durs) // 10] / 1e3:.1f}, max {durs[-1] / 1e3:.1f}; "
      f"sum {sum(durs) / 1e6:.1f} ms = {NF} x median {NF * med / 1e6:.1f} ms + excess {(sum(durs) - NF * med) / 1e6:.1f} ms")
idle = sum(x[0] - x[1] for x in fr)
print(f"# decode queue idle inside the frame intervals: {idle / 1e6:.1f} ms; busy {sum(x[1] for x in fr) / 1e6:.1f} ms; other queues' kernel time meanwhile {sum(x[2] for x in fr) / 1e6:.1f} ms")
quiet = [x for x in fr if x[2] == 0]
loud = [x for x in fr if x[2] > 0]
if quiet:
    print(f"# frames with nothing else on the GPU: {len(quiet)}, mean interval {sum(x[0] for x in quiet) / len(quiet) / 1e3:.1f} us (busy {sum(x[1] for x in quiet) / len(quiet) / 1e3:.1f})")
if loud:
    print(f"# frames overlapped by other queues:  {len(loud)}, mean interval {sum(x[0] for x in loud) / len(loud) / 1e3:.1f} us (busy {sum(x[1] for x in loud) / len(loud) / 1e3:.1f}, other kernels {sum(x[2] for x in loud) / len(loud) / 1e3:.1f})")
tot = collections.Counter()
for x in fr:
    tot.update(x[3])
for n, t in tot.most_common(12):
    print(f"#   other-queue kernel {n:60s} {t / 1e6:8.2f} ms")
print("# frame, interval us, decode-queue busy us, other queues' kernel us  (every frame whose interval exceeds 1.3 x median, at most 60)")
shown = 0
for i, x in enumerate(fr):
    if x[0] > 1.3 * med and shown < 60:
        top = ", ".join(f"{n} {t / 1e3:.0f}" for n, t in x[3].most_common(2))
        print(f"{i:4d} {x[0] / 1e3:9.1f} {x[1] / 1e3:9.1f} {x[2] / 1e3:9.1f}  {top}")
        shown += 1
