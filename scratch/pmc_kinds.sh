#!/bin/bash
# PMC passes of the 256 MiB bench for the other inputs (zipf, sym16): what bounds K5 / K7 there.  gpurun_out/pmc_kinds/
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_kinds
mkdir -p $O
cd /tmp
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
SQ2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
for KIND in zipf sym16; do
  i=0
  for PMC in "FETCH_SIZE" "WRITE_SIZE" "$SQ1" "$SQ2"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $O/${KIND}_p$i -- python3 $R/bench.py --kind $KIND --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-configs > $O/${KIND}_p$i.log 2>&1 || echo "pmc $KIND pass $i failed"
  done
done
find $O -name '*kernel_trace.csv' -size +1M -delete
find $O -name '*.db' -delete
