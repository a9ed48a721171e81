"""gpurun_out/r4f/ (written by scratch/final_r04.sh on the GPU box) -> profiles/r04/: the -m gpu suite's and smoke()'s last
lines, the driver's bench command under four look-ahead ramps, one PMC table for the Zipf and 16-symbol inputs (HBM bytes
derived as in collect_r03.py: (2 x FETCH_SIZE + WRITE_SIZE) x 1024, both counters in KiB), the soaks' last lines.
usage: python scratch/collect_final_r04.py"""
import collections
import csv
import glob
import json
import os

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
src, dst = os.path.join(ROOT, "gpurun_out", "r4f"), os.path.join(ROOT, "profiles", "r04")
SYMS = 268435456  # symbols per launch of the 256 MiB bench


def short(name):
    return name.split("(")[0].replace("void ", "").replace("ghf::", "").strip()


def tail(path, n):
    if not os.path.exists(path):
        return ["(missing)"]
    lines = [l.rstrip() for l in open(path, errors="replace") if l.strip()]
    return lines[-n:]


with open(os.path.join(dst, "final_call_summary.txt"), "w") as fh:
    fh.write("scratch/final_r04.sh on one MI355X box (the tree as committed; library sha256 in the bench lines)\n\n")
    for name, n in (("gpu_tests.log", 3), ("smoke.log", 1), ("soak_cabi.log", 2), ("soak_k6.log", 2)):
        fh.write("== %s\n" % name)
        for l in tail(os.path.join(src, name), n):
            fh.write("   %s\n" % l)
    fh.write("\n== python3 bench.py --steps 20 --warmup 5 (the driver's command), GHF_BENCH_RAMP0 = fronts enqueued before the first emit\n")
    rows = []
    for f in sorted(glob.glob(os.path.join(src, "bench20_ramp*_*.json"))):
        line = [l for l in open(f, errors="replace") if l.startswith("{")]
        if not line:
            fh.write("   %s: no JSON line\n" % os.path.basename(f))
            continue
        d = json.loads(line[-1])
        r0 = os.path.basename(f).split("_")[1].replace("ramp", "")
        rows.append((int(r0), d["value"], d["ms_per_step"]))
        fh.write("   RAMP0=%s  %8.1f GB/s  %.4f ms/step  decode alone %.4f ms  lib %s\n" % (r0, d["value"], d["ms_per_step"], d["stage_ms_alone"]["decode"], d["library"]["sha256"][:12]))
    by = collections.defaultdict(list)
    for r0, v, _ in rows:
        by[r0].append(v)
    for r0 in sorted(by):
        fh.write("   RAMP0=%d mean %.1f GB/s over %d runs\n" % (r0, sum(by[r0]) / len(by[r0]), len(by[r0])))

acc = {}
for kind in ("zipf", "sym16"):
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, "pmc_%s_p*" % kind, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k in ("k_histogram", "k_emit", "k_decode"):
                a[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    acc[kind] = a
counters = sorted({c for kind in acc for k in acc[kind] for c in acc[kind][k]})
if counters:
    with open(os.path.join(dst, "pmc_256MiB_zipf_sym16.csv"), "w") as fh:
        w = csv.writer(fh)
        w.writerow(["input", "kernel"] + counters + ["HBM_bytes_per_launch", "VALU_per_symbol", "LDS_conflict_over_active", "WAIT_ANY_over_WAVE_CYCLES"])
        for kind in ("zipf", "sym16"):
            for k in sorted(acc[kind]):
                m = {c: sum(v) / len(v) for c, v in acc[kind][k].items()}
                row = [kind, k] + ["%.6g" % m[c] if c in m else "" for c in counters]
                row.append(int((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024) if "FETCH_SIZE" in m and "WRITE_SIZE" in m else "")
                # SQ_INSTS_VALU counts wave instructions: x 64 lanes / symbols
                row.append("%.2f" % (m["SQ_INSTS_VALU"] * 64 / SYMS) if "SQ_INSTS_VALU" in m else "")
                row.append("%.3f" % (m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]) if m.get("SQ_LDS_IDX_ACTIVE") else "")
                row.append("%.2f" % (m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]) if m.get("SQ_WAVE_CYCLES") else "")
                w.writerow(row)
    print(open(os.path.join(dst, "pmc_256MiB_zipf_sym16.csv")).read())
print(open(os.path.join(dst, "final_call_summary.txt")).read())
