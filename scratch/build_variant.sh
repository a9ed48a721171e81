#!/bin/bash
# scratch/build_variant.sh <name> [extra hipcc flags...] -- the library built from the tree's sources with extra flags
# (e.g. -DGHF_CHUNKS_PER_SLOT=2) -> scratch/exp/libghf_<name>.so, for A/B runs with scratch/k_ab.py / bench_with_lib2.py
set -e
R=$(cd $(dirname $0)/.. && pwd)
N=$1; shift
mkdir -p $R/scratch/exp
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None \
  -I$R/include -I$R/golden-huffman_amd/csrc "$@" -shared -o $R/scratch/exp/libghf_$N.so \
  $R/golden-huffman_amd/csrc/ghf_kernels.hip $R/golden-huffman_amd/csrc/ghf_emit.hip $R/golden-huffman_amd/csrc/ghf_decode.hip \
  $R/golden-huffman_amd/csrc/ghf_api.hip $R/golden-huffman_amd/csrc/ghf_comm.hip -ldl 2>&1 | grep -v "argument unused" || true
echo built $R/scratch/exp/libghf_$N.so
