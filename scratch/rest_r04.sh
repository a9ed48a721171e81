#!/bin/bash
# the rest of profiles/r04 from the round's last library (parts B and C of profile_r04.sh without the PMC passes already taken), the
# fixed k_ab.py on the shipped library (per-launch events), then three bounded soaks.  A step killed at its limit ends the call.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04m
mkdir -p $O
step() {
  local t=$1 out=$2
  shift 2
  echo "[$(date +%H:%M:%S)] $*"
  timeout -k 10 $t "$@" > $out 2> ${out%.*}.err
  local rc=$?
  echo "   rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at its limit"; exit 1; fi
  return $rc
}
cd $R
step 400 $O/k_ab_fixed.txt python3 scratch/k_ab.py --mib 256,4096 --kinds uniform,zipf,sym16 --reps 40 shipped shipped_again
cat $O/k_ab_fixed.txt
cd /tmp
for KIND in uniform zipf sym16; do
  step 300 $O/foreign_$KIND.log rocprofv3 --kernel-trace --stats --output-format csv -d $O/foreign_$KIND -- python3 $R/scratch/foreign_prof.py $KIND
done
cd $R
step 300 $O/foreign_times.txt python3 scratch/foreign_time.py
GHF_BENCH_MAINS=1 step 200 $O/bench_one_main_stream.json python3 bench.py --steps 200 --no-configs --no-cpu-baseline
GHF_BENCH_NSIDE=2 step 200 $O/bench_two_side_streams.json python3 bench.py --steps 200 --no-configs --no-cpu-baseline
GHF_BENCH_NSIDE=1 step 200 $O/bench_one_side_stream.json python3 bench.py --steps 200 --no-configs --no-cpu-baseline
step 300 $O/membench.txt ./scratch/membench
step 600 $O/file_perf.log python3 scratch/file_perf.py 4 uniform zipf; cp gpurun_out/file_perf.json $O/file_perf.json
step 300 $O/pipe_trace_reuse_uniform.log python3 scratch/file_trace_reuse.py uniform
step 300 $O/pipe_trace_reuse_zipf.log python3 scratch/file_trace_reuse.py zipf
step 200 $O/soak_cabi.log python3 scratch/soak.py 120 5051
step 200 $O/soak_k6.log python3 scratch/k6_soak.py 90 5052
step 200 $O/soak_host.log python3 scratch/host_soak.py 90 5053
find $O -name '*kernel_trace.csv' -size +6M -delete
find $O -name '*.db' -delete
tail -2 $O/soak_cabi.log $O/soak_k6.log $O/soak_host.log
du -sh $O
