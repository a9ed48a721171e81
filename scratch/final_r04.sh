#!/bin/bash
# scratch/final_r04.sh -- the round's closing GPU call: the -m gpu suite and smoke() of the tree as committed, the driver's
# bench command under four look-ahead ramps, the PMC passes of the Zipf and 16-symbol inputs (profiles/r04 had uniform only),
# two bounded randomised soaks.  Every run keeps its stderr beside its output; a step that is killed at its limit ends the call.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4f
mkdir -p $O
step() {  # step <seconds> <log> <command...>
  local t=$1 log=$2
  shift 2
  echo "[$(date +%H:%M:%S)] $*"
  timeout -k 10 $t "$@" > $log 2>&1
  local rc=$?
  echo "   rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
    echo "killed at its limit: no further GPU step in this call"
    exit 1
  fi
  return $rc
}
cd $R
step 900 $O/gpu_tests.log python3 -m pytest tests -m gpu -x -q || exit 1
step 200 $O/smoke.log python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" || exit 1
# the driver's command, four ramps of the look-ahead (RAMP0 fronts before the first emit), twice each, alternating
for rep in 1 2; do
  for r0 in 3 5 7 4; do
    GHF_BENCH_RAMP0=$r0 step 200 $O/bench20_ramp${r0}_$rep.json python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs
  done
done
cd /tmp
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
SQ2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
for KIND in zipf sym16; do
  i=0
  for PMC in "FETCH_SIZE" "WRITE_SIZE" "$SQ1" "$SQ2"; do
    i=$((i+1))
    step 300 $O/pmc_${KIND}_p$i.log rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $O/pmc_${KIND}_p$i -- python3 $R/bench.py --kind $KIND --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-configs
  done
done
find $O -name '*kernel_trace.csv' -size +1M -delete
find $O -name '*.db' -delete
cd $R
step 150 $O/soak_cabi.log python3 scratch/soak.py 75 4041
step 150 $O/soak_k6.log python3 scratch/k6_soak.py 75 4042
echo done
