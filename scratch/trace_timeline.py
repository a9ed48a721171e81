#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV of a SHORT bench run -> the last `steps` steps' streaming kernels on one time axis: when each
started and ended, how long no streaming kernel (k_histogram / k_emit / k_decode) was running.
usage: python scratch/trace_timeline.py <kernel_trace.csv> <steps>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
ev = []
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ghf::", "").strip()
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
ev.sort()
dec = [e for e in ev if e[2] == "k_decode"]
# the timed region = the last `steps` decodes in front of the per-kernel pass (20 decodes, each alone) -> find by count from the start:
# warm-up decodes come first, then `steps` timed ones
import os
warm = int(os.environ.get("WARMUP", "5"))
first_timed_decode = dec[warm]
last_timed_decode = dec[warm + steps - 1]
# the region starts with the first k_histogram after the last warm-up decode ended
t_begin = min(e[0] for e in ev if e[2] == "k_histogram" and e[0] > dec[warm - 1][1])
t_end = last_timed_decode[1]
big = [e for e in ev if e[2] in ("k_histogram", "k_emit", "k_decode") and e[0] >= t_begin and e[1] <= t_end]
print("timed region %.3f ms, %d streaming kernels" % ((t_end - t_begin) / 1e6, len(big)))
# union of busy intervals
busy = 0; cur_s, cur_e = None, None; gaps = []
for s, e, n in sorted(big):
    if cur_s is None: cur_s, cur_e = s, e
    elif s <= cur_e: cur_e = max(cur_e, e)
    else:
        gaps.append((cur_e - t_begin, s - cur_e)); busy += cur_e - cur_s; cur_s, cur_e = s, e
busy += cur_e - cur_s
print("a streaming kernel was running %.3f ms; gaps: %d, total %.3f ms" % (busy / 1e6, len(gaps), sum(g for _, g in gaps) / 1e6))
for at, g in sorted(gaps, key=lambda x: -x[1])[:12]:
    print("  gap of %.1f us at %.3f ms" % (g / 1e3, at / 1e6))
for s, e, n in big[:40]:
    print("%8.3f %8.3f %7.1f us  %s" % ((s - t_begin) / 1e6, (e - t_begin) / 1e6, (e - s) / 1e3, n))
print("...")
for s, e, n in big[-24:]:
    print("%8.3f %8.3f %7.1f us  %s" % ((s - t_begin) / 1e6, (e - t_begin) / 1e6, (e - s) / 1e3, n))
