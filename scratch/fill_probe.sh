#!/bin/bash
# what a run pays once: 20-step and 200-step runs per input kind (K2 takes 0.26 ms on uniform / Zipf codes, 0.046 on 16 symbols)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4fill
mkdir -p $O
cd $R
for kind in uniform sym16 zipf; do
  for rep in 1 2 3; do
    timeout -k 10 200 python3 bench.py --kind $kind --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $O/${kind}_20_$rep.json 2> $O/${kind}_20_$rep.err || exit 1
  done
  timeout -k 10 200 python3 bench.py --kind $kind --steps 200 --no-cpu-baseline --no-configs > $O/${kind}_200_1.json 2> $O/${kind}_200_1.err || exit 1
  timeout -k 10 200 python3 bench.py --kind $kind --steps 60 --warmup 5 --no-cpu-baseline --no-configs > $O/${kind}_60_1.json 2> $O/${kind}_60_1.err || exit 1
done
python3 - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4fill/*.json")):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print("%-18s steps %3d  %.4f ms/step  total %.3f ms  %7.1f GB/s  K2 alone %.3f" % (f.split("/")[-1][:-5], d["steps"], d["ms_per_step"], d["ms_per_step"] * d["steps"], d["value"], d["stage_ms_alone"]["build_code"]))
P
