#!/bin/bash
# Build-container side of round 3's ablations: textual patches of a COPY of the kernel sources ->
# scratch/exp/libghf_<name>.so (the product sources carry no experiment switches).  usage: scratch/ablate_r3.sh
set -e
R=$(cd $(dirname $0)/.. && pwd)
P=$R/golden-huffman_amd
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None -I$R/include"
rm -rf $R/scratch/exp; mkdir -p $R/scratch/exp
build() {  # name, sed for ghf_decode.hip, sed for ghf_emit.hip, sed for ghf_kernels.hip
  T=$(mktemp -d /tmp/ghf_ab.XXXX)
  cp $P/csrc/*.hip $P/csrc/*.h $T/
  [ -n "$2" ] && sed -i -e "$2" $T/ghf_decode.hip
  [ -n "$3" ] && sed -i -e "$3" $T/ghf_emit.hip
  [ -n "$4" ] && sed -i -e "$4" $T/ghf_kernels.hip
  /opt/rocm/bin/hipcc $FLAGS -I$T -shared -o $R/scratch/exp/libghf_$1.so $T/ghf_kernels.hip $T/ghf_emit.hip $T/ghf_decode.hip $T/ghf_api.hip $T/ghf_comm.hip -ldl 2>&1 | grep -v "argument unused" || true
  rm -rf $T
  echo built $1
}
# K7's bookkeeping: the bare memory skeleton (scratch/membench3.hip) takes 0.092 / 1.61 ms, K7 without its decode 0.117 / 1.75.
# Every build below keeps every load and store of the kernel in place (same addresses: nothing can fault), the uniform-bytes
# decoder replaced by "the bytes a lane decodes are its bit offset" (nodecode) and then ONE piece of bookkeeping removed:
#   notable   the 64 KiB table replication of the prologue (unused without the decoder)
#   static    groups by static stride instead of ticket counters
#   linear    staging without byte swaps and without the padded layout's address arithmetic
# (the K7 bookkeeping builds of round 3 -- nodecode, nodecode_static, static, 64 / 128 ticket classes -- are in git history and in
#  profiles/r03/experiments/membench3_k7_skeleton.txt; a "ticket ranges" build ABORTED in the pipelined bench and is not kept)
build base "" "" "" &
wait
ls $R/scratch/exp/
