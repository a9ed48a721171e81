#!/bin/bash
# K5 with non-temporal input loads (scratch/exp/libghf_k5nt.so) against the shipped library: the pipelined bench, alternating
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4k
mkdir -p $O
cd $R
for rep in 1 2 3; do
  timeout -k 10 200 python3 bench.py --steps 200 --no-cpu-baseline --no-configs > $O/b256_shipped_$rep.json 2> $O/b256_shipped_$rep.err || exit 1
  timeout -k 10 200 python3 scratch/bench_with_lib2.py scratch/exp/libghf_k5nt.so --steps 200 --no-cpu-baseline --no-configs > $O/b256_k5nt_$rep.json 2> $O/b256_k5nt_$rep.err || exit 1
done
for kind in zipf sym16; do
  timeout -k 10 200 python3 bench.py --steps 200 --kind $kind --no-cpu-baseline --no-configs > $O/b256${kind}_shipped_1.json 2> $O/b256${kind}_shipped_1.err || exit 1
  timeout -k 10 200 python3 scratch/bench_with_lib2.py scratch/exp/libghf_k5nt.so --steps 200 --kind $kind --no-cpu-baseline --no-configs > $O/b256${kind}_k5nt_1.json 2> $O/b256${kind}_k5nt_1.err || exit 1
done
for rep in 1 2; do
  timeout -k 10 300 python3 bench.py --mib 4096 --steps 40 --warmup 2 --no-cpu-baseline --no-configs > $O/b4096_shipped_$rep.json 2> $O/b4096_shipped_$rep.err || exit 1
  timeout -k 10 300 python3 scratch/bench_with_lib2.py scratch/exp/libghf_k5nt.so --mib 4096 --steps 40 --warmup 2 --no-cpu-baseline --no-configs > $O/b4096_k5nt_$rep.json 2> $O/b4096_k5nt_$rep.err || exit 1
done
python3 - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4k/b*.json")):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    a = d["stage_ms_alone"]; s = d["stage_ms"]
    print("%-28s %7.1f GB/s  %.4f ms/step  alone K1 %.4f K5 %.4f K7 %.4f  in-pipeline K5 %.4f K7 %.4f  lib %s" % (f.split("/")[-1][:-5], d["value"], d["ms_per_step"], a["histogram"], a["emit"], a["decode"], s["emit"], s["decode"], d["library"]["sha256"][:8]))
P
