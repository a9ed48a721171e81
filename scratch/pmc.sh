#!/bin/bash
# usage: scratch/pmc.sh <tag> -- collects PMC passes for the bench kernels (run on the GPU box via gpurun)
TAG=$1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $R/gpurun_out/pmc_${TAG}/p$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify $BENCH_ARGS > $R/gpurun_out/pmc_${TAG}/p$i.log 2>&1 || echo "pass $i failed"
done
ls -R $R/gpurun_out/pmc_${TAG} | head -40
