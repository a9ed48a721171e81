#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench for the Zipf and 16-symbol inputs at 256 MiB and 4 GiB (BASELINE configs 3 and 5)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04s
mkdir -p $O
cd /tmp
for KIND in zipf sym16; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats256_$KIND -- python3 $R/bench.py --kind $KIND --steps 100 --warmup 3 --no-cpu-baseline --no-configs > $O/stats256_$KIND.log 2> $O/stats256_$KIND.err || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats4096_$KIND -- python3 $R/bench.py --kind $KIND --mib 4096 --steps 20 --warmup 2 --no-cpu-baseline --no-configs > $O/stats4096_$KIND.log 2> $O/stats4096_$KIND.err || exit 1
done
find $O -name '*.db' -delete
for d in $O/stats*/; do f=$(find $d -name '*kernel_trace.csv' | head -1); echo "== $d"; python3 $R/scratch/trace_alone.py $f $d/alone_vs_shared.csv | tail -7; done
find $O -name '*kernel_trace.csv' -size +2M -delete
