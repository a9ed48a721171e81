#!/bin/bash
# scratch/profile_r04_final.sh -- the round's last library (K1 / K5 launch edges): what profiles/r04/ holds per library is
# collected again -- bench line, rocprofv3 kernel stats of the same command, the four PMC passes at 256 MiB and 4 GiB, the other
# inputs' and sizes' bench lines, the driver's 20-step command.  (K6, the file pipeline and the micro-benchmarks are untouched
# by that change: their files stay.)  Every run keeps its stderr; a run killed at its limit ends the call.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04n
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
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
SQ2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
cd $R
step 500 $O/bench.json python3 bench.py || exit 1
cd /tmp
step 400 $O/stats256.log rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats256 -- python3 $R/bench.py --steps 100 --warmup 3 --no-cpu-baseline --no-configs
step 400 $O/stats4096.log rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats4096 -- python3 $R/bench.py --mib 4096 --steps 20 --warmup 2 --no-cpu-baseline --no-configs
for MIB in 256 4096; do
  i=0
  for PMC in "FETCH_SIZE" "WRITE_SIZE" "$SQ1" "$SQ2"; do
    i=$((i+1))
    step 300 $O/pmc${MIB}_p$i.log rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $O/pmc${MIB}_p$i -- python3 $R/bench.py --mib $MIB --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-configs
  done
done
cd $R
for KIND in zipf sym16; do
  step 300 $O/bench_256MiB_$KIND.json python3 bench.py --steps 100 --kind $KIND --no-configs --no-cpu-baseline
done
for KIND in uniform zipf; do
  step 300 $O/bench_4GiB_$KIND.json python3 bench.py --mib 4096 --steps 40 --warmup 2 --kind $KIND --no-configs --no-cpu-baseline
done
step 200 $O/bench_20_steps.json python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline
find $O -name '*kernel_trace.csv' -size +6M -delete
find $O -name '*.db' -delete
du -sh $O
