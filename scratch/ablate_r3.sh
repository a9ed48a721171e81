#!/bin/bash
# Build-container side of round 3's K5/K7 ablations: textual patches of a COPY of the kernel sources ->
# scratch/exp/libghf_<name>.so (the product sources carry no experiment switches).  usage: scratch/ablate_r3.sh
set -e
R=$(cd $(dirname $0)/.. && pwd)
P=$R/golden-huffman_amd
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None -I$R/include"
rm -rf $R/scratch/exp; mkdir -p $R/scratch/exp
build() {  # name, sed for ghf_decode.hip, sed for ghf_emit.hip, git rev to take ghf_emit.hip from, sed for ghf_internal.h
  T=$(mktemp -d /tmp/ghf_ab.XXXX)
  cp $P/csrc/*.hip $P/csrc/*.h $T/
  [ -n "$4" ] && git -C $R show $4:golden-huffman_amd/csrc/ghf_emit.hip > $T/ghf_emit.hip
  [ -n "$2" ] && sed -i -e "$2" $T/ghf_decode.hip
  [ -n "$3" ] && sed -i -e "$3" $T/ghf_emit.hip
  [ -n "$5" ] && sed -i -e "$5" $T/ghf_internal.h
  /opt/rocm/bin/hipcc $FLAGS -I$T -shared -o $R/scratch/exp/libghf_$1.so $T/ghf_kernels.hip $T/ghf_emit.hip $T/ghf_decode.hip $T/ghf_api.hip $T/ghf_comm.hip -ldl 2>&1 | grep -v "argument unused" || true
  rm -rf $T
  echo built $1
}
build base "" "" &
build emit_r02 "" "" 774c86c &
build emit_nosplit "" 's|constexpr bool kSplitLookup = true;|constexpr bool kSplitLookup = false;|' &
wait
build emit_occ8 "" 's|__launch_bounds__(kEmitThreads, 6)|__launch_bounds__(kEmitThreads, 8)|' "" 's|constexpr int kEmitThreads = 512; |constexpr int kEmitThreads = 1024;|; s|constexpr uint32_t kEmitSlots = 256 \* 3 \* 8;|constexpr uint32_t kEmitSlots = 256 * 2 * 16;|' &
build emit_occ8_nosplit "" 's|__launch_bounds__(kEmitThreads, 6)|__launch_bounds__(kEmitThreads, 8)|; s|constexpr bool kSplitLookup = true;|constexpr bool kSplitLookup = false;|' "" 's|constexpr int kEmitThreads = 512; |constexpr int kEmitThreads = 1024;|; s|constexpr uint32_t kEmitSlots = 256 \* 3 \* 8;|constexpr uint32_t kEmitSlots = 256 * 2 * 16;|' &
build emit_nostore "" 's|      W.out_units\[W.unit_base + j\] = v;|      if (v.x == 0x12345678u \&\& v.y == 0x9abcdef0u) W.out_units[W.unit_base + j] = v;|' &
wait
ls $R/scratch/exp/
