#!/bin/bash
# A/B of the K1 / K5 prologue-epilogue changes, then the -m gpu suite and the bench of the new library
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4h
mkdir -p $O
cd $R
timeout -k 10 400 python3 scratch/k_ab.py --mib 256,4096 --kinds uniform,zipf --reps 50 old=scratch/exp/libghf_k1pro.so new old2=scratch/exp/libghf_k1pro.so new2 > $O/k5pro.txt 2> $O/k5pro.err || exit 1
cat $O/k5pro.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err || exit 1
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $O/bench20_a.json 2> $O/bench20_a.err || exit 1
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $O/bench20_b.json 2> $O/bench20_b.err || exit 1
python3 - <<'P'
import json
for f in ("bench.json","bench20_a.json","bench20_b.json"):
    d=json.loads([l for l in open("gpurun_out/r4h/"+f) if l.startswith("{")][-1])
    print(f, d["value"], d["ms_per_step"], d["stage_ms_alone"], d["roofline"]["frac"])
P
