#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> per-kernel duration statistics split by whether a dispatch SHARED the GPU with another one.

bench.py's default layout runs even and odd steps on two main streams (+ the side streams' small kernels), so inside the timed
region a k_decode usually overlaps the other stream's k_emit and its duration is that of a kernel on part of the machine; the
per-kernel pass behind the timed region launches one kernel at a time.  `rocprofv3 --stats` averages both kinds together.
This splits them with the dispatch timestamps of the same trace:
    alone      no other dispatch of a STREAMING kernel (k_histogram / k_emit / k_decode) overlaps it in time
    shared     at least one does
usage: python scratch/trace_alone.py <kernel_trace.csv> [out.csv]
"""
import csv
import sys

STREAMING = ("k_histogram", "k_emit", "k_decode")
rows = list(csv.DictReader(open(sys.argv[1])))
d = []
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ghf::", "").strip()
    d.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
d.sort()
big = [(s, e, n) for s, e, n in d if any(n.startswith(k) for k in STREAMING)]
out = {}
for i, (s, e, n) in enumerate(big):
    shared = False
    j = i - 1
    while j >= 0 and big[j][0] > s - 50_000_000:  # (look back 50 ms: far more than any kernel here lasts)
        if big[j][1] > s:
            shared = True
            break
        j -= 1
    if not shared and i + 1 < len(big) and big[i + 1][0] < e:
        shared = True
    out.setdefault((n, "shared" if shared else "alone"), []).append(e - s)
lines = [["kernel", "dispatches_were", "calls", "avg_ns", "min_ns", "median_ns", "max_ns"]]
for (n, kind), v in sorted(out.items()):
    v.sort()
    lines.append([n, kind, len(v), round(sum(v) / len(v), 1), v[0], v[len(v) // 2], v[-1]])
w = csv.writer(open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout)
w.writerows(lines)
