#!/bin/bash
# usage (GPU box): scratch/pmc_one.sh <tag> <lib|-> <mib> <kind>   -- SQ counters of K1 / K5 / K7 alone (scratch/prof_one.py), two passes
TAG=$1; LIB=$2; MIB=$3; KIND=$4
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_$TAG
cd /tmp
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $R/gpurun_out/pmc_$TAG/p$i -- python3 $R/scratch/prof_one.py $LIB $MIB $KIND > $R/gpurun_out/pmc_$TAG/p$i.log 2>&1 || echo "pass $i failed"
done
cd $R && python3 scratch/pmc_summary.py $TAG > gpurun_out/pmc_$TAG/summary.txt; grep -A18 "k_decode" gpurun_out/pmc_$TAG/summary.txt
