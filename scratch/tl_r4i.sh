#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4i
mkdir -p $O
cd /tmp
for r0 in 3 7; do
  GHF_BENCH_RAMP0=$r0 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/tl_ramp$r0 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-verify > $O/tl_ramp$r0.json 2> $O/tl_ramp$r0.err || exit 1
  f=$(find $O/tl_ramp$r0 -name '*kernel_trace.csv' | head -1)
  WARMUP=5 python3 $R/scratch/trace_timeline.py $f 20 > $O/timeline_ramp$r0.txt 2>&1
  grep -o '"value": [0-9.]*' $O/tl_ramp$r0.json
  head -3 $O/timeline_ramp$r0.txt
done
find $O -name '*.db' -delete
