#!/bin/bash
# the driver's 20-step command: K1 on the main stream (default) or on its own, 3 or 7 fronts before the first emit; 200 steps too
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4j
mkdir -p $O
cd $R
for rep in 1 2 3; do
  for cfg in "0 3" "1 3" "1 7" "0 7" "1 5"; do
    set -- $cfg
    GHF_BENCH_K1_STREAM=$1 GHF_BENCH_RAMP0=$2 timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $O/b20_k1s$1_ramp$2_$rep.json 2> $O/b20_k1s$1_ramp$2_$rep.err || exit 1
  done
done
for cfg in "0 3" "1 3" "1 7"; do
  set -- $cfg
  GHF_BENCH_K1_STREAM=$1 GHF_BENCH_RAMP0=$2 timeout -k 10 200 python3 bench.py --steps 200 --no-cpu-baseline --no-configs > $O/b200_k1s$1_ramp$2.json 2> $O/b200_k1s$1_ramp$2.err || exit 1
done
python3 - <<'P'
import json, glob, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r4j/b*.json")):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    key = f.split("/")[-1].rsplit("_", 1)[0] if "b20_" in f else f.split("/")[-1][:-5]
    acc[key].append(d["value"])
for k in sorted(acc):
    print("%-22s %s  mean %.1f" % (k, " ".join("%.1f" % v for v in acc[k]), sum(acc[k]) / len(acc[k])))
P
