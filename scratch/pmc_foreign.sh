#!/bin/bash
# PMC passes of the side-car-less decode (scratch/foreign_prof.py) per input kind: what bounds the K6 kernels.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_foreign
mkdir -p $O
cd /tmp
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
SQ2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
for KIND in uniform zipf sym16; do
  i=0
  for PMC in "$SQ1" "$SQ2"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $O/${KIND}_p$i -- python3 $R/scratch/foreign_prof.py $KIND > $O/${KIND}_p$i.log 2>&1 || echo "pmc $KIND pass $i failed"
  done
done
find $O -name '*kernel_trace.csv' -size +1M -delete
find $O -name '*.db' -delete
