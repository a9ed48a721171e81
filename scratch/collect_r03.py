"""gpurun_out/<tag>/ (written by scratch/profile_r03.sh on the GPU box) -> profiles/<out>/: the bench line, the
rocprofv3 kernel-stats summaries, one table per input size with every PMC counter averaged per kernel and the HBM
bytes derived from them, the pipeline trace summaries, the file-to-file rates; and profiles/traffic.json.
usage: python scratch/collect_r03.py r03 r03_a"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
tag, out = sys.argv[1], sys.argv[2]
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles", out)
os.makedirs(dst, exist_ok=True)


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("ghf::", "")
    return n.strip()


shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, "bench.json"))
for mib, name in ((256, "256MiB"), (4096, "4GiB")):
    for f in sorted(glob.glob(os.path.join(src, "stats%d" % mib, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime):
        shutil.copy(f, os.path.join(dst, "kernel_stats_%s.csv" % name))
traffic = {"_note": "HBM bytes per launch from the rocprofv3 PMC passes in profiles/%s/pmc_*.csv (uniform bytes): "
                    "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 -- FETCH_SIZE doubled as MI355X_MICROARCH.md's HBM section "
                    "prescribes for wide coalesced reads on gfx950, both counters in KiB" % out}
for mib, name in ((256, "256MiB"), (4096, "4GiB")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, "pmc%d_p*" % mib, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if "k_" not in k:
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for k in acc for c in acc[k]})
    with open(os.path.join(dst, "pmc_%s.csv" % name), "w") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "launches_seen"] + counters + ["HBM_bytes_per_launch=(2*FETCH_SIZE+WRITE_SIZE)*1024"])
        for k in sorted(acc):
            row = [k, max(len(v) for v in acc[k].values())]
            for c in counters:
                v = acc[k].get(c)
                row.append("%.6g" % (sum(v) / len(v)) if v else "")
            hb = ""
            if "FETCH_SIZE" in acc[k] and "WRITE_SIZE" in acc[k]:
                f_, w_ = acc[k]["FETCH_SIZE"], acc[k]["WRITE_SIZE"]
                hb = int((2 * sum(f_) / len(f_) + sum(w_) / len(w_)) * 1024)
                if k in ("k_histogram", "k_emit", "k_decode"):
                    traffic["%s_uniform_%dMiB" % (k, mib)] = hb
            row.append(hb)
            w.writerow(row)
json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
for kind in ("uniform", "zipf", "sym16"):
    for f in sorted(glob.glob(os.path.join(src, "foreign_%s" % kind, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime):
        shutil.copy(f, os.path.join(dst, "foreign_decode_kernel_stats_%s.csv" % kind))
for f in ("file_perf.json", "membench.txt", "bench_256MiB_zipf.json", "bench_256MiB_sym16.json", "bench_4GiB_uniform.json", "bench_4GiB_zipf.json", "pipe_trace_reuse_uniform.log",
          "pipe_trace_reuse_zipf.log"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
print(open(os.path.join(ROOT, "profiles", "traffic.json")).read())
