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
# K1 at 256 MiB: 64 us where a pure read takes 45 -- what do the final reduction and the LDS atomics cost?  (measured once:
# base 0.0648, without the final global atomics 0.0631, without ANY ds_add 0.0606 ms -- profiles/r03/experiments/k1_ablate_256MiB.log.
# The builds leave the histogram empty, so everything behind K1 works on garbage: the no-atomics build FAULTED in a later
# kernel at 4 GiB.  Do not run such a build through ablate_run.py again; time K1 alone.)
build base "" "" "" &
wait
ls $R/scratch/exp/
