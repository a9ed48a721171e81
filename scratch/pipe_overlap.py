"""Summarise a rocprofv3 --kernel-trace --memory-copy-trace run of ghf_tool: how much of the time the three engines
(H2D copies, kernels, D2H copies) were busy at the same time.  usage: pipe_overlap.py <enc dir> <dec dir> <MiB>"""
import csv
import glob
import os
import sys


def load(d):
    ker, cp = [], []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ker.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            cp.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"]))
    return ker, cp


def union(iv):
    iv = sorted((a, b) for a, b in iv if b > a)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def length(u):
    return sum(b - a for a, b in u)


def inter(u, v):
    i = j = 0
    out = []
    while i < len(u) and j < len(v):
        a, b = max(u[i][0], v[j][0]), min(u[i][1], v[j][1])
        if b > a:
            out.append([a, b])
        if u[i][1] < v[j][1]:
            i += 1
        else:
            j += 1
    return out


def report(name, d, mib):
    ker, cp = load(d)
    big = [c for c in cp if c[1] - c[0] >= 50000]  # the pieces (>= 50 us on the bus); scalars and tables are not what keeps it busy
    h2d = union((a, b) for a, b, di in big if "HOST_TO_DEVICE" in di)
    d2h = union((a, b) for a, b, di in big if "DEVICE_TO_HOST" in di)
    k = union((a, b) for a, b, _ in ker)
    if not ker:
        print(name, ": no kernel records in", d)
        return
    t0 = min([a for a, _, _ in ker] + [a for a, _, _ in cp])
    t1 = max([b for _, b, _ in ker] + [b for _, b, _ in cp])
    ms = lambda x: x / 1e6
    print("== %s: %d MiB zipf, one file-to-file run (GPU-side trace: first to last GPU activity %.1f ms)" % (name, mib, ms(t1 - t0)))
    print("   H2D pieces busy %.1f ms (%d copies) | kernels busy %.1f ms (%d launches) | D2H pieces busy %.1f ms (%d copies)"
          % (ms(length(h2d)), sum(1 for c in big if "HOST_TO_DEVICE" in c[2]), ms(length(k)), len(ker), ms(length(d2h)),
             sum(1 for c in big if "DEVICE_TO_HOST" in c[2])))
    span = lambda u: ms(u[-1][1] - u[0][0]) if u else 0.0
    print("   first to last: H2D pieces %.1f ms, kernels %.1f ms, D2H pieces %.1f ms" % (span(h2d), span(k), span(d2h)))
    kh, kd, hd = inter(k, h2d), inter(k, d2h), inter(h2d, d2h)
    print("   kernels running while an H2D piece is in flight: %.1f ms = %.0f %% of kernel time" % (ms(length(kh)), 100.0 * length(kh) / max(1, length(k))))
    print("   kernels running while a D2H piece is in flight:  %.1f ms = %.0f %% of kernel time" % (ms(length(kd)), 100.0 * length(kd) / max(1, length(k))))
    print("   H2D and D2H pieces in flight together:           %.1f ms" % ms(length(hd)))
    by = {}
    for a, b, n in ker:
        e = by.setdefault(n, [0, 0])
        e[0] += 1
        e[1] += b - a
    for n, (c, t) in sorted(by.items(), key=lambda x: -x[1][1])[:8]:
        print("     %-28s %6d launches %9.3f ms" % (n[:28], c, ms(t)))


if __name__ == "__main__":
    report("compress", sys.argv[1], int(sys.argv[3]))
    report("decompress", sys.argv[2], int(sys.argv[3]))
