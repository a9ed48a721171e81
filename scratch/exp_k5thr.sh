#!/bin/bash
# experiment: where does K5's non-temporal store stop paying?  never / always
set -e
R=$(cd $(dirname $0)/.. && pwd)
P=$R/golden-huffman_amd
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None -I$R/include"
mkdir -p $R/scratch/exp
rm -f $R/scratch/exp/libghf_*.so
build() {
  T=$(mktemp -d /tmp/ghf_ab.XXXX)
  cp $P/csrc/*.hip $P/csrc/*.h $T/
  sed -i -e "s|W.stream = P.n <= (1ull << 30);|W.stream = $2;|" $T/ghf_emit.hip
  grep -q "W.stream = $2;" $T/ghf_emit.hip
  /opt/rocm/bin/hipcc $FLAGS -I$T -shared -o $R/scratch/exp/libghf_$1.so $T/ghf_kernels.hip $T/ghf_emit.hip $T/ghf_decode.hip $T/ghf_api.hip $T/ghf_comm.hip -ldl 2>&1 | grep -v "argument unused" || true
  rm -rf $T
  echo built $1
}
build never false &
build always true &
wait
