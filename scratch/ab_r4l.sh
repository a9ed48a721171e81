#!/bin/bash
# K7 with its first span requested before the table copy (scratch/exp/libghf_k7pro.so) against the shipped library
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4l
mkdir -p $O
cd $R
timeout -k 10 400 python3 scratch/k_ab.py --mib 256,4096 --kinds uniform,zipf,sym16 --reps 60 shipped k7pro=scratch/exp/libghf_k7pro.so shipped2 k7pro2=scratch/exp/libghf_k7pro.so > $O/k_ab.txt 2> $O/k_ab.err || { tail -5 $O/k_ab.err; exit 1; }
for rep in 1 2 3; do
  timeout -k 10 200 python3 bench.py --steps 200 --no-cpu-baseline --no-configs > $O/b256_shipped_$rep.json 2> $O/b256_shipped_$rep.err || exit 1
  timeout -k 10 200 python3 scratch/bench_with_lib2.py scratch/exp/libghf_k7pro.so --steps 200 --no-cpu-baseline --no-configs > $O/b256_k7pro_$rep.json 2> $O/b256_k7pro_$rep.err || exit 1
done
python3 - <<'P'
import json, glob
for l in open("gpurun_out/r4l/k_ab.txt"):
    if l.startswith("{"):
        d = json.loads(l)
        print("%-10s" % d["arm"], "  ".join("%s %.4f" % (k, v["decode_ms"]) for k, v in d.items() if k != "arm"), " ok" if all(v["ok"] for k, v in d.items() if k != "arm") else " WRONG")
for f in sorted(glob.glob("gpurun_out/r4l/b*.json")):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    a = d["stage_ms_alone"]
    print("%-22s %7.1f GB/s  %.4f ms/step  alone K1 %.4f K5 %.4f K7 %.4f" % (f.split("/")[-1][:-5], d["value"], d["ms_per_step"], a["histogram"], a["emit"], a["decode"]))
P
