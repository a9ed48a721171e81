#!/bin/bash
# Build-container side of the K7/K5 ablations: textual patches of a COPY of the kernel sources -> scratch/exp/libghf_<name>.so
# (the product sources carry no experiment switches).  usage: scratch/ablate_build.sh
set -e
R=$(cd $(dirname $0)/.. && pwd)
P=$R/golden-huffman_amd
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None -I$R/include -I$P/csrc"
rm -rf $R/scratch/exp; mkdir -p $R/scratch/exp
build() {  # name, sed script for ghf_decode.hip, sed script for ghf_emit.hip
  T=$(mktemp -d /tmp/ghf_ab.XXXX)
  cp $P/csrc/*.hip $P/csrc/*.h $T/
  [ -n "$2" ] && sed -i -e "$2" $T/ghf_decode.hip
  [ -n "$3" ] && sed -i -e "$3" $T/ghf_emit.hip
  /opt/rocm/bin/hipcc $FLAGS -I$T -shared -o $R/scratch/exp/libghf_$1.so $T/ghf_kernels.hip $T/ghf_emit.hip $T/ghf_decode.hip $T/ghf_api.hip 2>&1 | grep -v "argument unused" || true
  rm -rf $T
  echo built $1
}
build base "" "" &
# K5: no unit stores (LDS work stays)
build emit_nostore "" 's|      W.out_units\[W.unit_base + j\] = v;|      if (v.x == 0x12345678u \&\& v.y == 0x9abcdef0u) W.out_units[W.unit_base + j] = v;|' &
# K5: no manual drain
build emit_nodrain "" 's|  if (DRAIN) __builtin_amdgcn_s_waitcnt(0x0F71);  // vmcnt(1)||' &
wait
# K5: tile loads replaced by arithmetic (no global reads in the main loop)
build emit_noload "" 's|        A = pv\[nx \* 64\];|        A = make_uint4(v.x * 2654435761u + 1u, v.y + 7u, v.z ^ v.x, v.w + 3u);|; s|        B = pv\[nx \* 64\];|        B = make_uint4(v.x * 2654435761u + 1u, v.y + 7u, v.z ^ v.x, v.w + 3u);|' &
# K5: no side-car stores
build emit_noseg "" 's|  if (seg_dst) \*seg_dst = seg_base + incl;  // side-car: where this lane.s segment ends, relative to its block||' &
wait
ls $R/scratch/exp/
