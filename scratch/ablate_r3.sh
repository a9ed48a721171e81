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
# cache-policy variants of the pipelined bench (scratch/bench_with_lib.py)
build base "" "" "" &
build k7_span_plain 's|      R\[k\] = load_stream(base + (o + 16u <= lim ? o : 0u));|      R[k] = *reinterpret_cast<const uint4*>(base + (o + 16u <= lim ? o : 0u));|' "" "" &
build k7_store_plain 's|            store_stream(og + r \* 1024, \*reinterpret_cast<const uint4\*>(tile + sl \* 16 + piece \* 4));|            *reinterpret_cast<uint4*>(og + r * 1024) = *reinterpret_cast<const uint4*>(tile + sl * 16 + piece * 4);|' "" "" &
wait
build k5_store_nt "" 's|      W.out_units\[W.unit_base + j\] = v;|      store_stream(\&W.out_units[W.unit_base + j], v);|' "" &
build k1_load_plain "" "" 's|load_stream(vptr(|*(vptr(|g' &
build k5_load_nt "" 's|            buf\[j\] = pv\[nx \* 64\];|            buf[j] = load_stream(pv + nx * 64);|' "" &
wait
ls $R/scratch/exp/
