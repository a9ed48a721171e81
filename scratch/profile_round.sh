#!/bin/bash
# run on the GPU box: bench JSON + rocprofv3 kernel stats + PMC traffic passes -> gpurun_out/<tag>/
TAG=$1; shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp
python3 $R/bench.py "$@" > $O/bench.json 2> $O/bench.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 100 --warmup 3 --no-cpu-baseline "$@" > $O/stats.log 2>&1
for PMC in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $O/pmc_$PMC -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-verify "$@" > $O/pmc_$PMC.log 2>&1
done
cat $O/bench.json
